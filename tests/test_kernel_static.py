"""Static checks of the hand-written kernel source that need no GPU (hipcc cross-compiles gfx950 here).

The dominant kernel (csrc/ntt16_kernels.hip) issues its source loads with inline asm and waits for them with counted s_waitcnt: the
compiler does not know that those asm statements are loads, so a register spill or a copy that it places between a load and its wait would
read garbage silently.  tools/check_inflight.py walks the ISA of every kernel of that file with the in-order completion model of vmcnt and
flags any instruction that touches a destination register of a load that may still be in flight -- for the shipped build and for the
pipelined-loads experiment (MKHE_H16_PIPE: eleven source pairs in flight; with twelve the allocator does spill one, which the same check reports)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("flags,ok", [((), True), (("-DMKHE_H16_PIPE",), True), (("-DMKHE_H16_PIPE", "-DMKHE_H16_PIPE_P=12"), False)])
def test_no_instruction_touches_a_register_of_a_load_in_flight(flags, ok):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py"), *flags], capture_output=True, text=True, timeout=900)
    assert (r.returncode == 0) == ok, r.stdout[-1500:] + r.stderr[-500:]

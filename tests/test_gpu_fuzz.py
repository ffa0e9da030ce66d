"""Short, seeded runs of the random-shape checkers of tools/ (-m gpu): what the fixed cases of the other files do not reach.

* tools/fuzz_batch.py -- the lock-step entry points against the single-operation ones (device against device).  Its first run, in round 5, found
  mkhe_rotate_batch reading digits it had staged for the fused small-ring kernel as complete transforms: every fixed batch test ran on a ring where nothing is staged.
* tools/fuzz_parity.py -- Rotate / Conjugate / MulAndRelin (and mkbfv MulRelinNew) on random rings, levels, id sets, against the oracle.
* tools/fuzz_circuit.py -- random circuits on the mkckks Evaluator surface (rotate-and-add, lanes) against the oracle evaluator.

Each is a child process (the scripts are programs: `python3 tools/fuzz_*.py SECONDS SEED`), a dozen seconds each; the long runs are in profiles/r5_fuzz_*.txt."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seconds,seed,extra,least", [
    ("fuzz_batch.py", 8, 1201, [], 300),
    ("fuzz_parity.py", 12, 1202, ["wide"], 100),
    ("fuzz_circuit.py", 10, 1203, [], 100),
    ("fuzz_parity.py", 20, 1204, ["pn15"], 4),          # the headline ring with its full chain (seconds of oracle time per case): levels 0-13, 1-4 parties
])
def test_seeded_fuzz_run(script, seconds, seed, extra, least):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(seconds), str(seed)] + extra, capture_output=True, text=True, timeout=600)
    tail = out.stdout[-1500:] + out.stderr[-1500:]
    assert out.returncode == 0 and "MISMATCH" not in out.stdout, tail
    m = re.search(r"^# (\d+) (?:cases|operations|batched calls) in", out.stdout, flags=re.M)
    assert m and int(m.group(1)) >= least, tail

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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fused_f2_kernel_keeps_its_loads_in_flight_untouched():
    """ntt16_f2_kernel (round 6) holds key words and -- on the U-class path -- ten source pairs of the NEXT digit in flight across its product phase and
    the loop's back edge, all hand-issued: the same check on its ISA.  (With twelve pairs and four key pairs the allocator spills inside the digit
    loop, and on the balanced path it spills registers of loads in flight: that path keeps the two-group loads, and this test says if that changes.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, CHECK_SRC="ntt16_f2_kernels.hip"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    res = _kernel_resources("ntt16_f2_kernels.hip")
    k = [n for n in res if "ntt16_f2_kernel" in n]
    assert len(k) == 1 and res[k[0]]["vgpr_count"] <= 128, res          # one 1024-thread workgroup per CU: four waves per SIMD


def _kernel_resources(src):
    """name -> dict of the code-object metadata (.vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size) of every kernel of a csrc file"""
    import re
    csrc = os.path.join(ROOT, "mkhe-kklss_amd", "csrc")
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                        os.path.join(csrc, src), "-o", "-"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    out = {}
    for blk in r.stdout.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_h16_kernels_fit_eight_waves_per_simd_without_scratch():
    """every instantiation of the H16 forward kernel (N = 2^15, N = 2^14, the halves and the quarters of N = 2^16) keeps 16 coefficients per thread in at most 64
    VGPRs -- 8 waves per SIMD, two 1024-thread workgroups per CU -- and spills no vector register (scratch of 512 resident workgroups goes through HBM)"""
    res = _kernel_resources("ntt16_kernels.hip")
    names = [n for n in res if "ntt16_fwd" in n or "ntt14_fwd" in n]
    assert len(names) == 6, sorted(res)
    for n in names:
        # (the quarter kernel of N = 2^16 carries a third instantiation of the pass, the double-precision F class, whose 64-bit constants the
        # compiler keeps in vector registers: six spilled registers around one re-distribution are tolerated there)
        spills, scratch = (6, 32) if "ntt14_fwd_split" in n else (0, 0)
        assert res[n]["vgpr_count"] <= 64 and res[n]["vgpr_spill_count"] <= spills and res[n]["private_segment_fixed_size"] <= scratch, (n, res[n])
    # the inverse kernel of the family (round 3): same occupancy; a dozen spilled registers around the member sums of a merged launch are tolerated,
    # the 738 of its first balanced-path instantiation are not
    inv = [n for n in res if "ntt14_inv" in n]
    assert len(inv) == 1, sorted(res)
    assert res[inv[0]]["vgpr_count"] <= 64 and res[inv[0]]["vgpr_spill_count"] <= 16 and res[inv[0]]["private_segment_fixed_size"] <= 64, res[inv[0]]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_digit_spread_kernels_do_not_spill():
    res = _kernel_resources("poly_kernels.hip")
    names = [n for n in res if "decomp_spread" in n]
    assert len(names) == 3, sorted(res)                      # <0>, <1> and the radix-4 kernel
    for n in names:
        assert res[n]["vgpr_spill_count"] == 0 and res[n]["private_segment_fixed_size"] == 0, (n, res[n])
    r4 = [n for n in names if "spread4" in n][0]
    assert res[r4]["vgpr_count"] <= 128, res[r4]              # four waves per SIMD


def _isa(src, *flags):
    csrc = os.path.join(ROOT, "mkhe-kklss_amd", "csrc")
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", *flags, "-S", "--cuda-device-only",
                        os.path.join(csrc, src), "-o", "-"], capture_output=True, text=True, timeout=900)
    return r


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_m0_is_only_touched_by_the_addtid_groups():
    """ds_write_addtid_b32 takes its base from M0, which the asm blocks of ntt16_kernels.hip set themselves (addtid_write8); hipcc answers the `m0`
    clobber with "reserved registers on the clobber list may not be preserved" (157 notes): the compiler promises nothing about M0 around the block.
    What the kernels need is weaker and checkable on the ISA: (1) every add-TID store sits in a group `s_mov_b32 m0, sN; s_nop 0; ds_write_addtid_b32 x 8`
    with nothing else in between (the write of M0 and its wait state are ours); (2) NO other instruction of these kernels reads or writes M0 -- so there
    is no compiler-held value in M0 that the blocks could destroy, and none that could leak into a block.  A compiler that starts using M0 in these
    kernels (LDS-DMA, s_movrel, interpolation ...) fails here instead of computing wrong limbs."""
    r = _isa("ntt16_kernels.hip")
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l.strip() for l in r.stdout.splitlines()]
    code = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    groups = stores = 0
    i = 0
    while i < len(code):
        op = code[i].split()[0]
        if op == "s_mov_b32" and code[i].split()[1].rstrip(",") == "m0":
            assert code[i + 1].startswith("s_nop 0"), code[i:i + 3]
            n = 0
            while code[i + 2 + n].startswith("ds_write_addtid_b32"):
                n += 1
            assert n == 8, code[i:i + 12]
            groups += 1; stores += n; i += 2 + n
            continue
        assert op != "ds_write_addtid_b32", "an add-TID store outside its group: " + code[i]
        toks = code[i].replace(",", " ").split()
        assert "m0" not in toks[1:], "M0 used outside the add-TID groups: " + code[i]
        i += 1
    assert groups >= 24 and stores == 8 * groups, (groups, stores)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_h32_kernel_fits_one_workgroup_per_cu_without_scratch():
    """ntt32_kernels.hip: 32 coefficients per thread + the pinned product temporaries v122..v127 in at most 128 VGPRs (4 waves per SIMD: the one
    1024-thread workgroup per CU), no spilled vector register, no scratch; its ablation switches do not compile without -DMKHE_ABLATION"""
    res = _kernel_resources("ntt32_kernels.hip")
    names = [n for n in res if "ntt32_fwd" in n]
    assert len(names) == 2, sorted(res)
    for n in names:
        assert res[n]["vgpr_count"] <= 128 and res[n]["vgpr_spill_count"] == 0 and res[n]["private_segment_fixed_size"] == 0, (n, res[n])
    r = _isa("ntt32_kernels.hip", "-DMKHE_H32_X_NOBFLY")
    assert r.returncode != 0 and "MKHE_ABLATION" in r.stderr
    assert _isa("ntt32_kernels.hip", "-DMKHE_H32_X_NOBFLY", "-DMKHE_ABLATION").returncode == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["ntt16_kernels.hip", "ntt32_kernels.hip", "ntt16_f2_kernels.hip"])
def test_forward_kernels_read_their_job_constants_with_scalar_loads(src):
    """Round 3 shipped `kb->sched[m]` (a byte of the kernel arguments, dynamic index) in the job walk of every H16-class forward kernel: a byte load
    is a VECTOR memory instruction (global_load_ubyte + v_readfirstlane), and the `s_waitcnt vmcnt(0)` the compiler has to put between the two waits
    for every result store of the previous job before the next one has requested a single word.  Worth 0.4 % when it was found (stage 0's counted
    waits stand behind those stores anyway), but a stall the source does not show: the constants of a job are wave-uniform and come through s_load --
    no sub-dword vector load may appear in these files, and the only dword vector loads are the data / twiddle loads (x2 and x4)."""
    r = _isa(src)
    assert r.returncode == 0, r.stderr[-1500:]
    bad = [l.strip() for l in r.stdout.splitlines() if l.strip().startswith(("global_load_ubyte", "global_load_sbyte", "global_load_ushort", "global_load_sshort",
                                                                             "global_load_dword ", "flat_load", "buffer_load"))]
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_h32_prefetch_experiment_keeps_its_registers_in_flight_untouched():
    """-DMKHE_H32_PREFETCH=1 (not the default: slower by 3 %) requests the next limb's words between the stores of the current one; the 64
    destination registers are in flight across the loop's back edge, invisible to the compiler.  tools/ntt32_inflight_check.py walks the ISA from
    the requests to the counted waits of stage 0: no instruction may touch a register before the wait that covers it, and no full vector-memory
    wait may precede the first counted one."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ntt32_inflight_check.py"), "-DMKHE_H32_PREFETCH=1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]

"""Diagnostic: per-wave phase timeline of the forward NTT kernel (mkhe_ntt_trace): where do a limb's cycles go?

    make -C mkhe-kklss_amd/csrc trace
    MKHE_LIB=$PWD/mkhe-kklss_amd/lib/libmkhe_hip_trace.so python tools/ntt_trace.py [polys]
(13 small primes of PN15QP880 -> one kernel class, polys*13 limbs; the stamps are compiled in only in the trace build)
"""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkrlwe
from mkhe_kklss_amd._abi import lib, check
pset = H.PN15QP880
params = mkrlwe.Parameters(15, pset["Q"][1:14], pset["P"], 2)
N = 1 << 15
count, limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 80, 13
rng = np.random.default_rng(0)
a = rng.integers(0, 1 << 53, (count, limbs, N), dtype=np.uint64)
src = mkrlwe.DeviceLimbs(params, count, limbs).upload(a)
dst = mkrlwe.DeviceLimbs(params, count, limbs)
njobs = count * limbs
words = njobs * 16 * 16
tr = mkrlwe.DeviceLimbs(params, (words + N - 1) // N, 1)
for _ in range(3): mkrlwe.ntt(params, src, dst)
check(lib().mkhe_ntt_trace(params.ctx, tr.devptr()))
mkrlwe.ntt(params, src, dst)
params.sync()
check(lib().mkhe_ntt_trace(params.ctx, None))
t = tr.download().reshape(-1)[:words].reshape(njobs, 16, 16).astype(np.int64)
st = t[:, :, :10]                         # shader-clock stamps
names = ["load wait", "reduce+phase0", "xchg A->B (barriers)", "phase1", "xchg B->C", "phase2", "normalise", "xchg C->B", "store issue"]
d = np.diff(st, axis=2)                   # [job][wave][9]
tot = st[:, :, 9] - st[:, :, 0]
print("limbs %d; per-wave total cycles: mean %.0f  min %.0f  max %.0f" % (njobs, tot.mean(), tot.min(), tot.max()))
for k, n in enumerate(names):
    print("  %-24s mean %8.0f cycles (%5.1f%%)   min %8.0f  max %8.0f" % (n, d[:, :, k].mean(), 100 * d[:, :, k].mean() / tot.mean(), d[:, :, k].min(), d[:, :, k].max()))
rt = (t[:, :, 13] - t[:, :, 12]) / 100.0
print("real time per wave per limb: mean %.1f us -> shader clock %.2f GHz" % (rt.mean(), tot.mean() / rt.mean() / 1e3))
# skew between the waves of one workgroup at the end of each phase
for k in (2, 3, 4, 6, 9):
    sk = st[:, :, k].max(axis=1) - st[:, :, k].min(axis=1)
    print("  wave skew at stamp %d: mean %.0f cycles" % (k, sk.mean()))
# gap between consecutive limbs of one workgroup (persistent loop): next start - previous end on the same HW slot
hw = t[:, 0, 14]
start, end = t[:, 0, 12], t[:, :, 13].max(axis=1)
span = (end.max() - start.min()) / 100.0
print("kernel span %.1f us for %d limbs -> %.2f us per limb chip-wide" % (span, njobs, span / njobs))

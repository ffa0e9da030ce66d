"""Diagnostic: per-wave phase timeline of ntt32_fwd_kernel (trace build: make -C mkhe-kklss_amd/csrc trace;
MKHE_NTT32=1 MKHE_LIB=.../libmkhe_hip_trace.so python tools/ntt32_trace.py [parties])."""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import lib, check
p = H.PN15QP880
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, level = 1 << 15, 13
rng = np.random.default_rng(0)
ids = ["u%d" % i for i in range(k)]
host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
ct = mkckks.NewCiphertext(params, ids, level, p["scale"]).upload(host)
ev = mkckks.NewEvaluator(params)
njobs = k * 224
words = njobs * 16 * 32
tr = mkrlwe.DeviceLimbs(params, (words + N - 1) // N, 1)
for _ in range(200): ev.HoistedForm(ct)
check(lib().mkhe_ntt_trace(params.ctx, tr.devptr()))
ev.HoistedForm(ct)
params.sync()
check(lib().mkhe_ntt_trace(params.ctx, None))
t = tr.download().reshape(-1)[:words].reshape(njobs, 16, 32).astype(np.int64)
names = ["loads + stage 0", "phase A (4 stages)", "xchg A->B (4 barriers)", "phase B", "xchg B->C", "phase C", "normalise", "xchg C->E", "store issue"]
mods = t[:, 0, 15]
for cls, sel in (("U-class limbs", mods != 0), ("all limbs", mods >= 0)):
    st = t[sel][:, :, 0:10]
    d = np.diff(st, axis=2)
    tot = st[:, :, 9] - st[:, :, 0]
    print("%s (%d): per-wave cycles per limb mean %.0f  min %.0f  max %.0f" % (cls, sel.sum(), tot.mean(), tot.min(), tot.max()))
    for i, n in enumerate(names):
        print("  %-26s mean %8.0f (%5.1f%%)  min %7.0f  max %8.0f" % (n, d[:, :, i].mean(), 100 * d[:, :, i].mean() / tot.mean(), d[:, :, i].min(), d[:, :, i].max()))
rt = (t[:, :, 13] - t[:, :, 12]) / 100.0
cyc = t[:, :, 9] - t[:, :, 0]
print("real time per wave per limb: mean %.1f us; shader clock %.2f GHz" % (rt.mean(), (cyc / (rt * 1e3)).mean()))
start, end = t[:, 0, 12], t[:, :, 13].max(axis=1)
print("kernel span %.1f us for %d limbs" % ((end.max() - start.min()) / 100.0, njobs))
# one workgroup's timeline: wave 0 vs wave 15 offsets at each stamp for its first limbs
blk = t[:, 0, 14]
j0 = np.where(blk == 5)[0]
j0 = j0[np.argsort(t[j0, 0, 12])]
for j in j0[:3]:
    base = t[j, :, 0].min()
    print("workgroup 5, job %d (modulus %d): stamps relative to the first wave's start, waves 0 / 5 / 10 / 15" % (j, mods[j]))
    for w in (0, 5, 10, 15):
        print("   wave %2d: %s" % (w, " ".join("%6d" % (t[j, w, i] - base) for i in range(10))))

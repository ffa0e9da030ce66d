"""Diagnostic: per-wave phase timeline of ntt16_fwd_kernel (trace build: make -C mkhe-kklss_amd/csrc trace;
MKHE_LIB=.../libmkhe_hip_trace.so python tools/ntt16_trace.py [parties]).  Where do a pass's cycles go, and do the two
workgroups of a CU overlap?"""
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
for _ in range(3): ev.HoistedForm(ct)
check(lib().mkhe_ntt_trace(params.ctx, tr.devptr()))
ev.HoistedForm(ct)
params.sync()
check(lib().mkhe_ntt_trace(params.ctx, None))
t = tr.download().reshape(-1)[:words].reshape(njobs, 16, 32).astype(np.int64)
names = ["loads + stage 0", "phase A", "xchg A->B (4 barriers)", "phase B (+ C twiddle requests)", "xchg B->C", "phase C (+ D requests)", "xchg C->D",
         "phase D", "normalise", "xchg D->E", "store issue"]
for h in (0, 1):
    st = t[:, :, 16 * h: 16 * h + 12]
    d = np.diff(st, axis=2)
    tot = st[:, :, 11] - st[:, :, 0]
    print("pass %d: per-wave cycles mean %.0f  min %.0f  max %.0f" % (h, tot.mean(), tot.min(), tot.max()))
    for i, n in enumerate(names):
        print("  %-32s mean %8.0f (%5.1f%%)  min %7.0f  max %8.0f" % (n, d[:, :, i].mean(), 100 * d[:, :, i].mean() / tot.mean(), d[:, :, i].min(), d[:, :, i].max()))
gap = t[:, :, 16] - t[:, :, 11]
print("between the passes (store drain wait + reload issue): mean %.0f cycles" % gap.mean())
rt = (t[:, :, 28] - t[:, :, 12]) / 100.0
cyc = t[:, :, 16 + 11] - t[:, :, 0]
print("real time per wave per limb: mean %.1f us; shader clock %.2f GHz" % (rt.mean(), (cyc / (rt * 1e3)).mean()))
hw = t[:, 0, 13]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7      # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13
start, end = t[:, 0, 12], t[:, :, 28].max(axis=1)
print("kernel span %.1f us for %d limbs" % ((end.max() - start.min()) / 100.0, njobs))
blk = t[:, 0, 14]
# which workgroups were resident at the same time on the same (se, sh, cu) -- within one XCD only; XCC_ID is not in HW_ID
key = se * 32 + sh * 16 + cu
first = {}
for j in np.argsort(start):
    first.setdefault(int(blk[j]), (int(key[j]), int(start[j])))
from collections import Counter
c = Counter(k for k, _ in first.values())
print("distinct (se,sh,cu) keys %d; workgroups per key at start: %s" % (len(c), dict(Counter(c.values()))))

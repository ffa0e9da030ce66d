"""Diagnostic: timeline of forward-NTT workgroups (start/end per limb, CU placement)."""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkrlwe
from mkhe_kklss_amd._abi import lib, check
pset = H.PN15QP880
params = mkrlwe.Parameters(15, pset["Q"][1:14], pset["P"], 2)     # 13 small primes only -> one kernel class
N = 1 << 15
count, limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 80, 13
rng = np.random.default_rng(0)
a = rng.integers(0, 1 << 53, (count, limbs, N), dtype=np.uint64)
src = mkrlwe.DeviceLimbs(params, count, limbs).upload(a)
dst = mkrlwe.DeviceLimbs(params, count, limbs)
njobs = count * limbs
tr = mkrlwe.DeviceLimbs(params, 1, 1)     # N words >= 4*njobs
assert 4 * njobs <= N
for _ in range(3): mkrlwe.ntt(params, src, dst)
check(lib().mkhe_ntt_trace(params.ctx, tr.devptr()))
mkrlwe.ntt(params, src, dst)
params.sync()
check(lib().mkhe_ntt_trace(params.ctx, None))
t = tr.download().reshape(-1)[: 4 * njobs].reshape(njobs, 4).astype(np.int64)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # us
dur = end - start
hw = t[:, 2]; xcc = (hw >> 20) & 0xf   # not the XCC id, only used to count placements
clk = t[:, 3] / np.maximum(dur, 1e-9) / 1e3
print('shader clock during a limb (s_memtime ticks / us): mean %.2f GHz  min %.2f  max %.2f' % (clk.mean(), clk.min(), clk.max()))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
place = hw & 0xfffff00
print("jobs", njobs, "kernel span %.1f us" % end.max(), "mean limb %.1f us  min %.1f  max %.1f" % (dur.mean(), dur.min(), dur.max()))
print("distinct placements (xcc,se,sh,cu):", len(set(place.tolist())), " distinct xcc:", sorted(set(xcc.tolist())))
# concurrency over time
ts = np.linspace(0, end.max(), 40)
print("concurrent limbs over time:", [int(((start <= x) & (end > x)).sum()) for x in ts])
# per placement: jobs and gaps
order = np.argsort(start)
gaps = []
byp = {}
for j in order:
    byp.setdefault(int(place[j]), []).append(j)
for p, js in byp.items():
    for a_, b_ in zip(js[:-1], js[1:]):
        gaps.append(start[b_] - end[a_])
gaps = np.array(gaps)
print("per-CU gap between consecutive limbs: mean %.2f us, median %.2f, max %.2f (n=%d)" % (gaps.mean(), np.median(gaps), gaps.max(), len(gaps)))

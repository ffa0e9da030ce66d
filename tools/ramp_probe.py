"""Diagnostic: what does the first ~80 ms of MulRelin steps warm up?  Times 5-step windows of the headline MulRelin after (a) nothing,
(b) N crs_expand launches (HBM-write-bound kernels on other buffers), (c) N MulRelin steps.  python tools/ramp_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness as H
from mkhe_kklss_amd import mkrlwe, mkckks
from mkhe_kklss_amd._abi import check, lib
p = H.PN15QP880
k = 4
names = ["user%d" % i for i in range(k)]
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
level = len(p["Q"]) - 1
rng = np.random.default_rng(1)
N = 1 << p["logN"]
def ct():
    return np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, 7, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
params.AddCRS(-1, seed=7)
ct0 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(ct())
ct1 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(ct())
ev = mkckks.NewEvaluator(params)
spare = mkrlwe.RelinearizationKey(params, "spare")
def window(n=5):
    params.sync(); t0 = time.perf_counter()
    for _ in range(n): r = ev.MulRelinNew(ct0, ct1, rlk)
    params.sync(); return (time.perf_counter() - t0) * 1e3 / n
for _ in range(2): ev.MulRelinNew(ct0, ct1, rlk)          # allocations, code objects
params.sync(); time.sleep(1.0)
print("after 1 s idle:            %.3f ms/step" % window())
time.sleep(1.0)
for _ in range(3000): check(lib().mkhe_crs_expand(params.ctx, 7, 5, spare.Value[0].h))      # ~100 ms of another kernel
print("after 3000 crs_expand:     %.3f ms/step" % window())
time.sleep(1.0)
for _ in range(100): ev.MulRelinNew(ct0, ct1, rlk)
print("after 100 MulRelin:        %.3f ms/step" % window())
print("next windows:              %s" % ["%.3f" % window() for _ in range(4)])
time.sleep(0.2)
print("after 0.2 s idle:          %.3f ms/step" % window())

"""Does replaying the MulRelin step as ONE hipGraph submission change its duration?  (round 3 experiment)
PN15QP880, 4 parties, device-expanded keys; eager issue vs graph replay, 300 steps each after 100 untimed ones."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import check, lib
p = H.PN15QP880
k = 4
names = ["u%d" % i for i in range(k)]
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
level = len(p["Q"]) - 1
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, 7, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
params.AddCRS(-1, seed=7)
rng = np.random.default_rng(1)
N = 1 << p["logN"]
host = lambda: np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
ct0 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(host())
ct1 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(host())
ev = mkckks.NewEvaluator(params)
def timed(fn, n=300, w=100):
    for _ in range(w): fn()
    params.sync(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    params.sync(); return (time.perf_counter() - t0) / n * 1e3
ref = ev.MulRelinNew(ct0, ct1, rlk).download()
print("eager  %.4f ms/step" % timed(lambda: ev.MulRelinNew(ct0, ct1, rlk)))
with params.Capture() as g:
    out = ev.MulRelinNew(ct0, ct1, rlk)
print("graph  %.4f ms/step" % timed(g.launch))
print("graph result identical:", bool((out.download() == ref).all()))
print("eager  %.4f ms/step" % timed(lambda: ev.MulRelinNew(ct0, ct1, rlk)))

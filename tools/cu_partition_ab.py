"""Experiment: MulRelin/s on PN15QP880, 4 parties, with 1-4 evaluators in flight (forked contexts, round robin), the contexts' streams
either on the whole chip or on disjoint CU sets (library built with -DMKHE_CU_PARTITION_EXPERIMENT: tools/build_variant.sh cupart engine -DMKHE_CU_PARTITION_EXPERIMENT;
MKHE_LIB=.../var_cupart/lib.so MKHE_CU_PARTS=2 [MKHE_CU_PART_MODE=1] python3 tools/cu_partition_ab.py)."""
import sys, os, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import lib, check
p = H.PN15QP880
k = int(os.environ.get("PARTIES", "4"))
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
N, level = 1 << 15, len(p["Q"]) - 1
rng = np.random.default_rng(0)
names = ["u%d" % i for i in range(k)]
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, 7, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
params.AddCRS(-1, seed=7)
def ct():
    h = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
    return mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h)
c0, c1 = ct(), ct()
evs = [mkckks.NewEvaluator(params)]
for _ in range(3):
    evs.append(evs[0].Fork())
ref = evs[0].MulRelinNew(c0, c1, rlk).download()
print("# MKHE_CU_PARTS=%s MKHE_CU_PART_MODE=%s parties %d" % (os.environ.get("MKHE_CU_PARTS"), os.environ.get("MKHE_CU_PART_MODE"), k))
for nfl in (1, 2, 3, 4, 2, 1):
    use = evs[:nfl]
    outs = [e.MulRelinNew(c0, c1, rlk) for e in use]
    same = all((o.download() == ref).all() for o in outs)
    for _ in range(100):
        for e in use: e.MulRelinNew(c0, c1, rlk)
    for e in use: e.params.sync()
    best = 0.0
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(100):
            for e in use: e.MulRelinNew(c0, c1, rlk)
        for e in use: e.params.sync()
        best = max(best, 100 * nfl / (time.perf_counter() - t))
    print("in flight %d: %8.1f MulRelin/s  (results identical: %s)" % (nfl, best, same), flush=True)

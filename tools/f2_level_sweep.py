"""Diagnostic: MulRelinNew / RotateNew per second on PN15QP880 by parties and level (device-resident operands, random residues, keys expanded on the device).
Run under the diagnostic library with MKHE_F2_FUSED set (0 unfused, 1 planned grid, 2 whole chip or nothing) to compare launch sets:  MKHE_LIB=.../libmkhe_hip_switches.so MKHE_F2_FUSED=0 python3 tools/f2_level_sweep.py"""
import sys, os, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import lib, check
p = H.PN15QP880
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
N = 1 << 15
rng = np.random.default_rng(0)
PARTIES = [int(x) for x in os.environ.get("PARTIES", "1 2 3 4").split()]
KMAX = max(PARTIES)
names = ["u%d" % i for i in range(KMAX)]
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, 7, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
params.AddCRS(-1, seed=7)
ev = mkckks.NewEvaluator(params)
levels = [int(x) for x in os.environ.get("LEVELS", "1 3 5 7 9 11 13").split()]
print("# MKHE_F2_FUSED=%s" % os.environ.get("MKHE_F2_FUSED"))
print("# parties level MulRelinNew/s")
for k in PARTIES:
    for level in levels:
        def ct():
            h = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"][:level + 1]]) for _ in range(1 + k)])
            return mkckks.NewCiphertext(params, names[:k], level, p["scale"]).upload(h)
        c0, c1 = ct(), ct()
        for _ in range(8): ev.MulRelinNew(c0, c1, rlk)
        params.sync()
        best = 0.0
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(40): ev.MulRelinNew(c0, c1, rlk)
            params.sync()
            best = max(best, 40 / (time.perf_counter() - t))
        print("%d %2d %9.1f" % (k, level, best), flush=True)

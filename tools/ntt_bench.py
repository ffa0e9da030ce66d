"""NTT kernel throughput vs batch size (limbs per launch), PN15QP880 moduli, N = 2^15."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import harness as H
from mkhe_kklss_amd import mkrlwe
pset = H.PN15QP880
logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
params = mkrlwe.Parameters(logN, pset["Q"], pset["P"], 2)
N = 1 << logN
limbs = 14
rng = np.random.default_rng(0)
COUNTS = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else (16, 18, 36, 73, 146, 292)
for count in COUNTS:
    nl = count * limbs
    a = rng.integers(0, 1 << 53, (count, limbs, N), dtype=np.uint64)
    src = mkrlwe.DeviceLimbs(params, count, limbs).upload(a)
    dst = mkrlwe.DeviceLimbs(params, count, limbs)
    for inverse in (False, True):
        for _ in range(3): mkrlwe.ntt(params, src, dst, inverse=inverse)
        params.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps): mkrlwe.ntt(params, src, dst, inverse=inverse)
        params.sync()
        dt = (time.perf_counter() - t0) / reps
        print("logN %d %s limbs %5d  %8.1f us/launch  %6.3f us/limb  %7.1f GB/s (16N B/limb)  %.1f us per limb-CU" % (
            logN, "inv" if inverse else "fwd", nl, dt * 1e6, dt * 1e6 / nl, nl * 16 * N / dt / 1e9, dt * 1e6 / max(1, -(-nl // 256))), flush=True)
    del src, dst

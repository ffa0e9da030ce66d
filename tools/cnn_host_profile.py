"""cProfile of the host side of one encrypted-CNN inference (where do the ~3.6 ms of issue time go?)"""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness_cnn as HC                      # noqa: E402
from mkhe_kklss_amd import cnn, mkckks, mkrlwe   # noqa: E402

p = HC.PN14QP433
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"])
params.GenDefaultCRS(seed=1)
for r in HC.ROTS:
    params.AddCRS(r, seed=1)
kgen = mkrlwe.NewKeyGenerator(params, mkrlwe.HostSampler(np.random.default_rng(1), insecure_test_only=True))
rlkSet, rtkSet = mkrlwe.RelinearizationKeySet(params), mkrlwe.RotationKeySet()
for id in ("dataOwner", "modelOwner"):
    sk = kgen.GenSecretKey(id)
    rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, kgen.GenSecretKey(id)))
    for r in HC.ROTS + [1 << i for i in range(p["logN"] - 1)]:
        rtkSet.AddRotationKey(kgen.GenRotationKey(r, sk))
rng = np.random.default_rng(2)
level, N = len(p["Q"]) - 1, 1 << p["logN"]


def ct(id):
    host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(2)])
    return mkckks.NewCiphertext(params, [id], level, p["scale"]).upload(host)


ctImage, ctK = ct("dataOwner"), [ct("modelOwner") for _ in range(4)]
ctFC1, ctFC2, ctB1, ctB2 = [ct("modelOwner") for _ in range(8)], ct("modelOwner"), ct("modelOwner"), ct("modelOwner")
pt = mkrlwe.DeviceLimbs(params, 1, level - 3).upload(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"][: level - 3]])[None])
ev = mkckks.NewEvaluator(params)
forks = [ev.Fork() for _ in range(7)]
hoisted = (ev.HoistedForm(ctImage), [ev.HoistedForm(c) for c in ctK], [ev.HoistedForm(c) for c in ctFC1])
run = lambda: cnn.Inference(ev, rlkSet, rtkSet, ctImage, ctK, ctFC1, ctFC2, ctB1, ctB2, pt, p["scale"], hoisted=hoisted, forks=forks)
for _ in range(3):
    run()
params.sync()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    run()
pr.disable()
params.sync()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)

"""Probe: B independent encrypted inferences issued from B host threads, each through its own forked engine context (own stream, shared keys):
inferences per second against B = 1.  python tools/cnn_threads_probe.py [parties] [steps]"""
import os, sys, time, threading
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness_cnn as HC
from mkhe_kklss_amd import cnn, mkckks, mkrlwe
from mkhe_kklss_amd._abi import check, lib
parties = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = HC.PN14QP433
owners = (dict(image="dataOwner", kernels="modelOwner", fc1="modelOwner", fc2="modelOwner") if parties <= 2 else
          dict(image="dataOwner", kernels="convOwner", fc1="fc1Owner", fc2="fc2Owner"))
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
params.GenDefaultCRS(seed=1)
for r in HC.ROTS:
    params.AddCRS(r, seed=1)
kgen = mkrlwe.NewKeyGenerator(params, mkrlwe.HostSampler(np.random.default_rng(1), insecure_test_only=True))
rlkSet, rtkSet = mkrlwe.RelinearizationKeySet(params), mkrlwe.RotationKeySet()
for id in sorted(set(owners.values())):
    sk = kgen.GenSecretKey(id)
    rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, kgen.GenSecretKey(id)))
    for r in HC.ROTS + [1 << i for i in range(p["logN"] - 1)]:
        rtkSet.AddRotationKey(kgen.GenRotationKey(r, sk))
params.sync()
rng = np.random.default_rng(2)
level, N = len(p["Q"]) - 1, 1 << p["logN"]
def ct(id):
    host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(2)])
    return mkckks.NewCiphertext(params, [id], level, p["scale"]).upload(host)
ctKernels = [ct(owners["kernels"]) for _ in range(4)]
ctFC1, ctFC2, ctB1, ctB2 = [ct(owners["fc1"]) for _ in range(8)], ct(owners["fc2"]), ct(owners["fc1"]), ct(owners["fc2"])
ptMask = mkrlwe.DeviceLimbs(params, 1, level - 3).upload(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"][: level - 3]])[None])
ev0 = mkckks.NewEvaluator(params)
hk, hf = [ev0.HoistedForm(c) for c in ctKernels], [ev0.HoistedForm(c) for c in ctFC1]
params.sync()
def run(B, forks_per):
    evs = [ev0.Fork() for _ in range(B)]
    fk = [[e.Fork() for _ in range(forks_per)] for e in evs]
    for e in evs + [f for l in fk for f in l]:
        check(lib().mkhe_set_overlap(e.params.ctx, 0))
    imgs = [ct(owners["image"]) for _ in range(B)]
    hi = [ev0.HoistedForm(c) for c in imgs]
    params.sync()
    outs = [None] * B
    def work(b, n):
        for _ in range(n):
            outs[b] = cnn.Inference(evs[b], rlkSet, rtkSet, imgs[b], ctKernels, ctFC1, ctFC2, ctB1, ctB2, ptMask, p["scale"], hoisted=(hi[b], hk, hf), forks=fk[b] or None)
        evs[b].params.sync()
    for phase, n in (("warm", 3), ("timed", steps)):
        th = [threading.Thread(target=work, args=(b, n)) for b in range(B)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    print("B = %2d threads x %d forks: %7.1f inferences/s (%.2f ms per inference per thread)" % (B, forks_per, B * steps / dt, dt * 1e3 / steps), flush=True)
    return outs
for B, f in ((1, 7), (1, 0), (2, 0), (4, 0), (4, 3), (8, 0), (8, 1)):
    run(B, f)

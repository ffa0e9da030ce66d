"""Probe: wall time, summed kernel time (HIP events per kernel class) and launches per step of a batched cnn inference.  python tools/cnn_batch_probe.py [B] [parties] [steps]"""
import os, sys, time, ctypes as C
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness_cnn as HC
from mkhe_kklss_amd import cnn, mkckks, mkrlwe
from mkhe_kklss_amd._abi import check, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
parties = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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
rng = np.random.default_rng(2)
level, N = len(p["Q"]) - 1, 1 << p["logN"]
def ct(id):
    host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(2)])
    return mkckks.NewCiphertext(params, [id], level, p["scale"]).upload(host)
ctKernels = [ct(owners["kernels"]) for _ in range(4)]
ctFC1, ctFC2, ctB1, ctB2 = [ct(owners["fc1"]) for _ in range(8)], ct(owners["fc2"]), ct(owners["fc1"]), ct(owners["fc2"])
ptMask = mkrlwe.DeviceLimbs(params, 1, level - 3).upload(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"][: level - 3]])[None])
ev = mkckks.NewEvaluator(params)
check(lib().mkhe_set_overlap(params.ctx, 0))
hk, hf = [ev.HoistedForm(c) for c in ctKernels], [ev.HoistedForm(c) for c in ctFC1]
if B > 1:
    E = mkckks.BatchEvaluator(params, B)
    img = mkckks.BatchCiphertext([ct(owners["image"]) for _ in range(B)])
else:
    E, img = ev, ct(owners["image"])
hi = E.HoistedForm(img)
def run():
    return cnn.Inference(E, rlkSet, rtkSet, img, ctKernels, ctFC1, ctFC2, ctB1, ctB2, ptMask, p["scale"], hoisted=(hi, hk, hf))
for _ in range(3): run()
params.sync()
t0 = time.perf_counter()
for _ in range(steps): run()
ti = time.perf_counter() - t0
params.sync()
dt = time.perf_counter() - t0
print("B = %d: %.2f ms per step wall (%.2f ms host issue), %.1f inferences/s" % (B, dt * 1e3 / steps, ti * 1e3 / steps, B * steps / dt))
ncls = lib().mkhe_prof_nclass()
names = [lib().mkhe_prof_name(i).decode().split("  ")[0] for i in range(ncls)]
check(lib().mkhe_prof_enable(params.ctx, 1))
for _ in range(steps): run()
ms = (C.c_double * ncls)(); cnt = (C.c_long * ncls)(); byt = (C.c_double * ncls)()
check(lib().mkhe_prof_collect(params.ctx, ms, cnt, byt))
check(lib().mkhe_prof_enable(params.ctx, 0))
print("kernel time per step %.2f ms over %d profiled launch groups per step" % (sum(ms) / steps, sum(cnt) / steps))
for i in sorted(range(ncls), key=lambda i: -ms[i]):
    if cnt[i]: print("   %-34s %6.1f groups/step  %7.3f ms/step" % (names[i], cnt[i] / steps, ms[i] / steps))

import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
from mkhe_kklss_amd import mkrlwe
Q = [0xfffffffff6a0001, 0x3fffffffd60001, 0x3fffffffca0001]
P = [0x7ffffffffe70001, 0x7ffffffffe10001]
rng = np.random.default_rng(7)
for logN in (10, 11, 12, 13, 14, 15):
    N = 1 << logN
    params = mkrlwe.Parameters(logN, Q, P, 2)
    ring = O.Ring(logN, Q + P)
    for i in range(5): assert params.Psi(i) == ring.psi(i)
    count = 3
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q + P]) for _ in range(count)])
    src = mkrlwe.DeviceLimbs(params, count, 5).upload(a)
    dst = mkrlwe.DeviceLimbs(params, count, 5)
    mkrlwe.ntt(params, src, dst)
    got = dst.download()
    ref = np.stack([np.stack([ring.ntt(i, a[c][i]) for i in range(5)]) for c in range(count)])
    ok_f = (got == ref).all()
    back = mkrlwe.DeviceLimbs(params, count, 5)
    mkrlwe.ntt(params, dst, back, inverse=True)
    ok_i = (back.download() == a).all()
    mkrlwe.ntt(params, dst, back, inverse=True, lazy=True)
    lz = back.download()
    ok_l = all(((lz[c][i] % np.uint64(q)) == a[c][i]).all() and (lz[c][i] < 2 * q).all() for c in range(count) for i, q in enumerate(Q + P))
    print("logN", logN, "fwd", ok_f, "inv", ok_i, "lazy", ok_l, flush=True)
    if not ok_f:
        bad = np.argwhere(got != ref); print(" mismatches", len(bad), bad[:5])

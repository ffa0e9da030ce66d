"""Determinism under load (-m gpu): the same little circuit -- MulRelinNew, RotateNew, AddNew, ConjugateNew on resident
ciphertexts, side-stream overlap on, a forked context doing the same concurrently -- evaluated a few hundred times must give
the first iteration's bits every time.  Guards the stream-ordered buffer pools, the side-stream fork / join logic and the
cross-context ordering against races (every kernel is deterministic, so any difference is an ordering bug)."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pname", ["N13_q6", "N14_q6", "N15_q4"])
def test_repeated_circuit_is_deterministic(pname):
    from mkhe_kklss_amd import mkckks, mkrlwe
    from mkhe_kklss_amd._abi import check, lib
    # (N14_q6: the ring whose engine-internal Decompose launches run through ext_fused_lds_kernel -- Rotate, Conjugate and step F2 of the circuit below)
    pset = {"N13_q6": H.small_ckks(13, 6), "N14_q6": H.small_ckks(14, 6), "N15_q4": H.small_ckks(15, 4)}[pname]
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"])
    names = ["a", "b", "c"]
    level = len(pset["Q"]) - 1
    rng = np.random.default_rng(3)
    N = 1 << pset["logN"]
    host = lambda: np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in pset["Q"]]) for _ in range(1 + len(names))])
    ct0 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(host())
    ct1 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(host())
    rlk, rks, cks = mkrlwe.RelinearizationKeySet(params), mkrlwe.RotationKeySet(), mkrlwe.ConjugationKeySet()
    seed = 99
    for i, n in enumerate(names):                      # uniform key material written on the device
        key = mkrlwe.RelinearizationKey(params, n)
        for j in range(3):
            check(lib().mkhe_crs_expand(params.ctx, seed, 100 + 3 * i + j, key.Value[j].h))
        rlk.AddRelinearizationKey(key)
        rk = mkrlwe.RotationKey(params, 1, n)
        check(lib().mkhe_crs_expand(params.ctx, seed, 200 + i, rk.Value.h))
        rks.AddRotationKey(rk)
        ck = mkrlwe.ConjugationKey(params, n)
        check(lib().mkhe_crs_expand(params.ctx, seed, 300 + i, ck.Value.h))
        cks.AddConjugationKey(ck)
    for idx in (-1, 1, -2):
        params.AddCRS(idx, seed=seed)
    ev = mkckks.NewEvaluator(params)
    fork = ev.Fork()

    def circuit(e):
        r = e.MulRelinNew(ct0, ct1, rlk)
        s = e.AddNew(e.RotateNew(r, 1, rks), r)
        return e.AddNew(e.ConjugateNew(s, cks), s)

    ref = circuit(ev).download()
    iters = 150 if pset["logN"] <= 13 else 100 if pset["logN"] == 14 else 40
    for it in range(iters):
        fork.params.wait_for(params)
        a = circuit(ev)
        b = circuit(fork)
        params.wait_for(fork.params)
        if it % 10 == 9 or it == iters - 1:
            assert (a.download() == ref).all(), "main context, iteration %d" % it
            assert (b.download() == ref).all(), "forked context, iteration %d" % it
    params.sync()
    fork.params.sync()


def test_buffer_freed_under_a_foreign_reader_is_not_reused_early():
    """A ciphertext created on the main context is read by work queued on a forked context and destroyed right away; a new
    ciphertext of the same shape (the pool hands the same buffer back) is then overwritten on the main stream.  The fork's
    results must be those of the original operand: pool_alloc has to order the main stream behind the fork's readers
    (Context::seq_ / synced_ bookkeeping, csrc/engine.hip), although the application itself never joined the two."""
    from mkhe_kklss_amd import mkckks, mkrlwe
    from mkhe_kklss_amd._abi import check, lib
    pset = H.small_ckks(15, 4)
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"])
    names = ["a", "b"]
    level = len(pset["Q"]) - 1
    rng = np.random.default_rng(11)
    N = 1 << pset["logN"]
    host = lambda: np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in pset["Q"]]) for _ in range(1 + len(names))])
    rlk = mkrlwe.RelinearizationKeySet(params)
    for i, n in enumerate(names):
        key = mkrlwe.RelinearizationKey(params, n)
        for j in range(3):
            check(lib().mkhe_crs_expand(params.ctx, 5, 10 + 3 * i + j, key.Value[j].h))
        rlk.AddRelinearizationKey(key)
    params.AddCRS(-1, seed=5)
    ev = mkckks.NewEvaluator(params)
    fork = ev.Fork()
    hx, hy, hz = host(), host(), host()
    ct1 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(hy)
    x = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(hx)
    ref = ev.MulRelinNew(x, ct1, rlk).download()
    for rep in range(6):
        fork.params.wait_for(params)
        outs = [fork.MulRelinNew(x, ct1, rlk) for _ in range(6)]          # milliseconds of queued readers of x on the fork's stream
        ptr = x.devptr()
        x.__del__()                                                          # back into the main context's pool, no join
        held = []                                                            # the pool hands out its oldest matching entry first
        for _ in range(16):
            held.append(mkckks.NewCiphertext(params, names, level, pset["scale"]))
            if held[-1].devptr() == ptr:
                break
        z = held[-1]
        assert z.devptr() == ptr, "the pool is expected to hand the freed buffer back"
        z.upload(hz)                                                         # overwrites it on the main stream
        for o in outs:
            assert (o.download() == ref).all(), "repetition %d: a reader on the forked context saw the buffer's next contents" % rep
        for h in held:
            h.__del__()
        x = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(hx)
    params.sync()
    fork.params.sync()


def test_failed_context_creation_leaves_no_trace():
    """mkhe_ctx_create that throws AFTER the streams exist (a supplied psi that is not a primitive 2N-th root; an alpha the
    decomposer tables do not cover) must not leave a half-built context in the per-device registry that the pools of the other
    contexts walk (csrc/engine.hip Context::Context / registry_add): afterwards handles of a good context on the same device --
    and of a fork of it -- are created, used and freed as usual, and the failed call did not disturb a split-phase sequence
    (MKHE_TRY resets the thread's last-context pointer before anything can throw)."""
    from mkhe_kklss_amd import mkckks, mkrlwe
    from mkhe_kklss_amd._abi import MkheError
    pset = H.small_ckks(12, 4)
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"])
    names = ["a", "b"]
    level = len(pset["Q"]) - 1
    rng = np.random.default_rng(3)
    N = 1 << pset["logN"]
    host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in pset["Q"]]) for _ in range(1 + len(names))])
    ct = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(host)
    for _ in range(3):
        with pytest.raises(MkheError, match="psi"):
            mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], psiQ=[2] * len(pset["Q"]))
        with pytest.raises(MkheError, match="PCount/gamma"):            # alpha = 7: thrown while the decomposer tables are built
            mkrlwe.Parameters(12, H.PN16_Q[:6], H.PN16_P + H.PN15QP880["P"] + H.PN14QP439["P"][:1], gamma=1)
    ev = mkckks.NewEvaluator(params)
    fork = ev.Fork()
    for it in range(8):                                   # pool_free / pool_alloc walk the registry on every one of these
        t = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(host)
        s = ev.AddNew(ct, t)
        fork.params.wait_for(params)
        r = fork.AddNew(s, t)
        params.wait_for(fork.params)
        exp = host.copy()
        for l, q in enumerate(pset["Q"]):
            exp[:, l] = (3 * host[:, l].astype(object) % q).astype(np.uint64)
        assert (r.download() == exp).all()
        for h in (t, s, r):
            h.__del__()
    fork.params.sync()
    params.sync()

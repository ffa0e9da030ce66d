"""Replays the reference's own property tests on this path with seeded, VALID keys:

  mkrlwe_test.go:456-505   ExternalProduct      log2(sum|err|) <= 10 + logN
  mkrlwe_test.go:507-610   Decompose (max / min level)          <= 10 + logN
  mkckks_test.go:201-226   encrypt/decrypt precision            <= -log2(scale)+logSlots+8
  mkckks_test.go:320-362   MulRelin of a k-party sum, squared   <= -log2(scale)+logSlots+12
  mkckks_test.go:415-505   rotations (disabled in the driver, :126-130)  same +8 bound
  mkckks_test.go:507-550   conjugation

CPU half (always): on the oracle.  GPU half (-m gpu): the same scenarios through the C ABI, plus
bit-equality of the device result with the oracle's on the valid inputs.
"""
import numpy as np
import pytest

import harness as H
from scenario import Scenario

PSET_CPU = H.small_ckks(11, 4)


def _max_log2_err(a, b):
    d = np.abs(np.asarray(a) - np.asarray(b))
    return float(np.log2(max(d.real.max(), d.imag.max(), 1e-300)))


@pytest.fixture(scope="module")
def sc():
    return Scenario(PSET_CPU, parties=2, seed=5, rotations=(1, 4), conj=True)


def test_encrypt_decrypt_precision(sc):
    for n in sc.names:
        z = sc.message(-1 - 1j, 1 + 1j)
        c0, c1 = sc.encrypt(z, n)
        ct = np.stack([c0, c1])
        out = sc.decrypt_decode([n], ct, sc.scale)
        assert _max_log2_err(out, z) <= sc.precision_bound(8)


def test_external_product_noise(sc):
    """<h(c), g*s + e>_P ~ c*s"""
    ks, kg = sc.ks, sc.kg
    n = sc.names[0]
    level = sc.level
    z = sc.message(-1 - 1j, 1 + 1j)
    c0, _ = sc.encrypt(z, n)
    sg = kg.gen_switching_key(sc.sk[n])
    for lvl in (level, 0):
        a = c0[: lvl + 1]
        t = ks.external_product(lvl, a, sg)
        err = np.empty_like(t)
        for j in range(lvl + 1):
            cs = ks.ringQ.intt(j, ks.ringQ.mul(j, ks.ringQ.ntt(j, a[j]), sc.sk[n][j]))
            err[j] = ks.ringQ.sub(j, t[j], cs)
        assert H.log2_inner_sum(err, sc.Q[: lvl + 1]) <= 10 + sc.logN


def test_decompose_inner_product_noise(sc):
    """sum_i h_i(c) (.) (g_i s + e_i), ModDown  ~ c*s   (mkrlwe_test.go:507-610)"""
    ks, kg = sc.ks, sc.kg
    n = sc.names[1]
    z = sc.message(-1 - 1j, 1 + 1j)
    c0, _ = sc.encrypt(z, n)
    sg = kg.gen_switching_key(sc.sk[n])
    for lvl in (sc.level, 0):
        a = c0[: lvl + 1]
        t = ks.external_product_hoisted(lvl, ks.decompose(lvl, a), sg)
        err = np.empty_like(t)
        for j in range(lvl + 1):
            cs = ks.ringQ.intt(j, ks.ringQ.mul(j, ks.ringQ.ntt(j, a[j]), sc.sk[n][j]))
            err[j] = ks.ringQ.sub(j, t[j], cs)
        assert H.log2_inner_sum(err, sc.Q[: lvl + 1]) <= 10 + sc.logN


def _mulrelin_oracle(sc, ct):
    k = len(sc.names)
    ids = list(range(k))
    rl = {i: sc.rlk[n] for i, n in enumerate(sc.names)}
    _, out = sc.ks.mul_and_relin(sc.level, ids, ct, ids, ct, rl, sc.kg.CRS[-1])
    nb, scale = sc.ks.ckks_nb_rescales(sc.level, sc.scale * sc.scale, sc.scale)
    out = np.stack([sc.ks.ringQ.div_round_last_many(out[s], nb)[0] for s in range(1 + k)])
    return out, scale


def test_mulrelin_precision_oracle(sc):
    """testEvaluatorMul: square of the sum of k ciphertexts"""
    k = len(sc.names)
    zs = {n: sc.message(complex(0.1 / k, 1.0 / k), complex(0.1 / k, 1.0 / k) + 0) for n in sc.names}
    zs = {n: np.full(sc.N // 2, complex(0.1 / k, 1.0 / k)) + sc.rng.uniform(-0.05, 0.05, sc.N // 2) for n in sc.names}
    ct = sc.sum_ciphertext(zs)
    out, scale = _mulrelin_oracle(sc, ct)
    want = sum(zs.values()) ** 2
    got = sc.decrypt_decode(sc.names, out, scale)
    assert _max_log2_err(got, want) <= sc.precision_bound(12)


def test_rotation_and_conjugation_oracle(sc):
    ks = sc.ks
    zs = {n: sc.message(-1 - 1j, 1 + 1j) for n in sc.names}
    ct = sc.sum_ciphertext(zs)
    tot = sum(zs.values())
    for rot in (1, 4):
        galEl = pow(5, rot, 2 * sc.N)
        out = ks.rotate(sc.level, galEl, [0, 1], ct, [sc.rk[n][rot] for n in sc.names], sc.kg.CRS[rot])
        got = sc.decrypt_decode(sc.names, out, sc.scale)
        assert _max_log2_err(got, np.roll(tot, -rot)) <= sc.precision_bound(8)
    out = ks.conjugate(sc.level, 2 * sc.N - 1, [0, 1], ct, [sc.ck[n] for n in sc.names], sc.kg.CRS[-2])
    got = sc.decrypt_decode(sc.names, out, sc.scale)
    assert _max_log2_err(got, np.conj(tot)) <= sc.precision_bound(8)


# ------------------------------------------------------------------ the same on the device
@pytest.fixture(scope="module")
def dev(sc):
    from mkhe_kklss_amd import mkrlwe, mkckks
    params = mkckks.Parameters(sc.logN, sc.Q, sc.P, sc.scale)
    for idx, host in sc.kg.CRS.items():
        params.AddCRS(idx, host)
    rlk = mkrlwe.RelinearizationKeySet(params)
    rks, cks = mkrlwe.RotationKeySet(), mkrlwe.ConjugationKeySet()
    for n in sc.names:
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *sc.rlk[n]))
        for rot, k in sc.rk[n].items():
            rks.AddRotationKey(mkrlwe.RotationKey(params, rot, n, k))
        cks.AddConjugationKey(mkrlwe.ConjugationKey(params, n, sc.ck[n]))
    return dict(params=params, rlk=rlk, rks=rks, cks=cks, ev=mkckks.NewEvaluator(params), mkckks=mkckks)


@pytest.mark.gpu
def test_mulrelin_precision_device(sc, dev):
    k = len(sc.names)
    zs = {n: np.full(sc.N // 2, complex(0.1 / k, 1.0 / k)) + sc.rng.uniform(-0.05, 0.05, sc.N // 2) for n in sc.names}
    ct = sc.sum_ciphertext(zs)
    d = dev["mkckks"].NewCiphertext(dev["params"], sc.names, sc.level, sc.scale).upload(ct)
    res = dev["ev"].MulRelinNew(d, d, dev["rlk"])
    got_ct = res.download()
    want = sum(zs.values()) ** 2
    got = sc.decrypt_decode(res.ids, got_ct, res.Scale)
    assert _max_log2_err(got, want) <= sc.precision_bound(12)
    ref, scale = _mulrelin_oracle(sc, ct)
    assert res.Scale == scale and (got_ct == ref).all()


@pytest.mark.gpu
def test_rotation_and_conjugation_device(sc, dev):
    zs = {n: sc.message(-1 - 1j, 1 + 1j) for n in sc.names}
    ct = sc.sum_ciphertext(zs)
    tot = sum(zs.values())
    d = dev["mkckks"].NewCiphertext(dev["params"], sc.names, sc.level, sc.scale).upload(ct)
    ev = dev["ev"]
    for rot in (1, 4, 5):          # 5 = 1 + 4 goes through the power-of-two decomposition (evaluator.go:516-523)
        res = ev.RotateNew(d, rot, dev["rks"])
        got = sc.decrypt_decode(res.ids, res.download(), res.Scale)
        assert _max_log2_err(got, np.roll(tot, -rot)) <= sc.precision_bound(8)
    hoisted = ev.HoistedForm(d)
    res = ev.RotateHoistedNew(d, 4, hoisted, dev["rks"])
    assert (res.download() == ev.RotateNew(d, 4, dev["rks"]).download()).all()
    res = ev.ConjugateNew(d, dev["cks"])
    got = sc.decrypt_decode(res.ids, res.download(), res.Scale)
    assert _max_log2_err(got, np.conj(tot)) <= sc.precision_bound(8)
    with pytest.raises(Exception, match="precomputed rotation keys"):
        ev.RotateHoistedNew(d, 5, hoisted, dev["rks"])


@pytest.mark.gpu
def test_caller_chain_device():
    """A caller on top of the path (what cnn/ does with the evaluator): encrypted inner product of a
    2-party vector with plaintext weights -- MulPtxtNew (incl. Rescale), rotate-and-sum over RotateNew / AddNew, a
    MultByConst and a SubNew -- entirely on resident ciphertexts; the decrypted result must match numpy within the
    CKKS precision bound the reference uses for one multiplication."""
    from mkhe_kklss_amd import mkrlwe, mkckks
    pset = H.small_ckks(11, 4)
    rots = (1, 2, 4, 8)
    sc = Scenario(pset, parties=2, seed=23, rotations=rots)
    params = mkckks.Parameters(sc.logN, sc.Q, sc.P, sc.scale)
    for idx, host in sc.kg.CRS.items():
        params.AddCRS(idx, host)
    rks = mkrlwe.RotationKeySet()
    for n in sc.names:
        for rot, k in sc.rk[n].items():
            rks.AddRotationKey(mkrlwe.RotationKey(params, rot, n, k))
    ev = mkckks.NewEvaluator(params)
    n = sc.N // 2
    zs = {nm: sc.rng.uniform(-1, 1, n) + 0j for nm in sc.names}
    x = sum(zs.values())
    w = sc.rng.uniform(-1, 1, n) + 0j
    ct = mkckks.NewCiphertext(params, sc.names, sc.level, sc.scale).upload(sc.sum_ciphertext(zs))
    pt = sc.enc.encode(w, sc.scale, sc.Q)
    prod = ev.MulPtxtNew(ct, pt, sc.scale)                       # slot-wise x * w, rescaled
    assert prod.Level() == sc.level - 1
    acc = prod
    for rot in rots:                                              # 16-slot partial sums by rotate-and-sum
        acc = ev.AddNew(acc, ev.RotateNew(acc, rot, rks))
    tripled = mkckks.NewCiphertext(params, sc.names, acc.Level(), acc.Scale)
    ev.MultByConst(acc, 3, tripled)
    res = ev.SubNew(tripled, acc)                                 # 3*acc - acc = 2*acc
    got = sc.decrypt_decode(res.ids, res.download(), res.Scale)
    xw = x * w
    want = 2 * sum(np.roll(xw, -s) for s in range(16))
    assert _max_log2_err(got, want) <= sc.precision_bound(12) + 5      # 32 summed terms: 5 more bits

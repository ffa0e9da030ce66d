"""MK-BFV parity tests (-m gpu): the HIP path through the C ABI (mkhe_bfv_*) vs the CPU oracle, bit-exact,
plus the reference's exact-decrypt property (mkbfv_test.go:365-401) replayed on the device.

Building blocks follow mkbfv/basis_extension.go (ModUpQtoR, Rescale, Quantize) and mkbfv/keyswitch*.go
(DecomposeBFV, ExternalProductBFVHoisted); the whole is mkbfv.Evaluator.MulRelinNew (evaluator.go:78-140).
"""
import numpy as np
import pytest

import harness as H
import harness_bfv as HB

pytestmark = pytest.mark.gpu

SETS = {
    "N10_q3": HB.small_bfv(10, 3),
    "N12_q4big": HB.small_bfv(12, 4, big=True),      # mixes 54- and 55-bit primes like the reference chain
    "N13_q2": HB.small_bfv(13, 2),
    "N11_q14": dict(HB.BFV_PN15QP880, logN=11),       # the full 14+14+2 prime chain at a small degree
}


class BfvPair:
    def __init__(self, pset, seed=0):
        from mkhe_kklss_amd import mkbfv, mkrlwe
        self.mk, self.mkb = mkrlwe, mkbfv
        self.pset = pset
        self.bfv = HB.make_bfv(pset)
        self.params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])
        self.ev = mkbfv.NewEvaluator(self.params)
        self.rng = np.random.default_rng(seed)
        self.N, self.nq = 1 << pset["logN"], len(pset["Q"])
        self.Q, self.QMul, self.P = pset["Q"], pset["QMul"], pset["P"]

    def swk(self):
        host = H.uniform_swk(self.rng, self.bfv.ks)
        return host, self.mk.SwitchingKey(self.params, host)

    def ct(self, names):
        host = H.uniform_ct(self.rng, self.bfv.ks, len(names), self.nq)
        return host, self.mkb.NewCiphertext(self.params, names).upload(host)

    def rlk_set(self, names):
        host, dev = {}, self.mkb.NewRelinearizationKeyKeySet(self.params)
        for n in names:
            ks = [H.uniform_swk(self.rng, self.bfv.ks) for _ in range(5)]
            host[n] = tuple(ks)
            dev.AddRelinearizationKey(self.mkb.RelinearizationKey(self.params, n, *ks))
        return host, dev


@pytest.fixture(scope="module", params=list(SETS))
def bp(request):
    return BfvPair(SETS[request.param], seed=hash(request.param) & 0xffff)


def _polys_q(bp, count):
    x = np.stack([H.uniform_poly(bp.rng, bp.Q, bp.N) for _ in range(count)])
    # coefficients on the integer boundaries of the float64 correction index
    x[0, :, 0] = 0
    x[0, :, 1] = np.array([q - 1 for q in bp.Q], dtype=np.uint64)
    x[0, :, 2] = 1
    return x


def test_modup_q_to_r(bp):
    x = _polys_q(bp, 3)
    src = bp.mk.DeviceLimbs(bp.params, 3, bp.nq).upload(x)
    dst = bp.mkb.PolyR(bp.params, 3)
    bp.ev.conv.ModUpQtoR(src, dst)
    got = dst.download()
    for c in range(3):
        assert (got[c] == bp.bfv.modup_q_to_r(x[c])).all()


def test_rescale(bp):
    x = _polys_q(bp, 3)
    src = bp.mk.DeviceLimbs(bp.params, 3, bp.nq).upload(x)
    dst = bp.mkb.PolyR(bp.params, 3)
    bp.ev.conv.Rescale(src, dst)
    got = dst.download()
    for c in range(3):
        assert (got[c] == bp.bfv.rescale(x[c])).all()


def test_ntt_r_and_quantize(bp):
    R = bp.Q + bp.QMul
    y = np.stack([np.stack([H.uniform_poly(bp.rng, [m], bp.N)[0] for m in R]) for _ in range(2)])
    src = bp.mkb.PolyR(bp.params, 2).upload(y)
    f = bp.mkb.PolyR(bp.params, 2)
    from mkhe_kklss_amd._abi import check, lib
    check(lib().mkhe_bfv_ntt_r(bp.params.ctx, src.devptr(), f.devptr(), 2, 0))
    yn = f.download()
    for c in range(2):
        assert (yn[c] == bp.bfv.ntt_r(y[c])).all()
    out = bp.mk.DeviceLimbs(bp.params, 2, bp.nq)
    bp.ev.conv.Quantize(f, out, bp.params.T())
    got = out.download()
    for c in range(2):
        assert (got[c] == bp.bfv.quantize(yn[c])).all()
    assert (f.download() == yn).all()          # the input is left untouched


@pytest.mark.parametrize("kind", ["modup", "rescale"])
def test_decompose_bfv(bp, kind):
    """DecomposeBFV on the lazy outputs of ModUpQtoR / Rescale (QMul resp. Q limbs up to ~3x their modulus)"""
    x = _polys_q(bp, 1)[0]
    ar = bp.bfv.modup_q_to_r(x) if kind == "modup" else bp.bfv.rescale(x)
    src = bp.mkb.PolyR(bp.params, 1).upload(ar[None])
    ad1, ad2 = bp.mk.NewSwitchingKey(bp.params), bp.mk.NewSwitchingKey(bp.params)
    bp.ev.ksw.DecomposeBFV(src, ad1, ad2)
    e1, e2 = bp.bfv.decompose(ar)
    assert (ad1.download() == e1).all()
    assert (ad2.download() == e2).all()


def test_external_product_bfv_hoisted(bp):
    (h1, d1), (h2, d2), (g1, k1), (g2, k2) = bp.swk(), bp.swk(), bp.swk(), bp.swk()
    c = bp.mk.DeviceLimbs(bp.params, 1, bp.nq)
    bp.ev.ksw.ExternalProductBFVHoisted(d1, d2, k1, k2, c)
    assert (c.download()[0] == bp.bfv.external_product_hoisted(h1, h2, g1, g2)).all()


@pytest.mark.parametrize("kind", ["modup", "rescale"])
def test_external_product_bfv_non_hoisted(bp, kind):
    """ExternalProductBFV (keyswitch.go:83-113): decomposition inside; equals DecomposeBFV + ExternalProductBFVHoisted"""
    x = _polys_q(bp, 1)[0]
    ar = bp.bfv.modup_q_to_r(x) if kind == "modup" else bp.bfv.rescale(x)
    src = bp.mkb.PolyR(bp.params, 1).upload(ar[None])
    (g1, k1), (g2, k2) = bp.swk(), bp.swk()
    c = bp.mk.DeviceLimbs(bp.params, 1, bp.nq)
    bp.ev.ksw.ExternalProductBFV(src, k1, k2, c)
    assert (c.download()[0] == bp.bfv.external_product(ar, g1, g2)).all()


CASES = [
    (["a"], ["a"]),
    (["a", "b"], ["a", "b"]),
    (["a"], ["b"]),
    (["a", "b"], ["b", "c"]),
    (["a", "b", "c"], ["a", "b", "c"]),
    ([], []),
    ([], ["a"]),
]


@pytest.mark.parametrize("ids0,ids1", CASES)
def test_mul_relin_new(bp, ids0, ids1):
    if bp.nq >= 14 and len(set(ids0) | set(ids1)) > 2:
        pytest.skip("large chain: covered with fewer parties")
    names = sorted(set(ids0) | set(ids1))
    idx = {n: i for i, n in enumerate(names)}
    h0, c0 = bp.ct(ids0)
    h1, c1 = bp.ct(ids1)
    rlk_h, rlk_d = bp.rlk_set(names)
    u_h, u_d = bp.swk()
    bp.params.CRS[-1] = u_d
    out = bp.ev.MulRelinNew(c0, c1, rlk_d)
    ido, ref = bp.bfv.mul_relin_new([idx[i] for i in ids0], h0, [idx[i] for i in ids1], h1,
                                    {idx[n]: rlk_h[n] for n in names}, u_h)
    assert out.ids == [names[i] for i in ido]
    assert (out.download() == ref).all()
    # same call again (pools and cached hoisted slots are reused) and the square of one handle
    assert (bp.ev.MulRelinNew(c0, c1, rlk_d).download() == ref).all()
    sq = bp.ev.MulRelinNew(c0, c0, rlk_d)
    _, ref2 = bp.bfv.mul_relin_new([idx[i] for i in ids0], h0, [idx[i] for i in ids0], h0,
                                   {idx[n]: rlk_h[n] for n in names}, u_h)
    assert (sq.download() == ref2).all()
    # the reference's non-hoisted twin (Evaluator.mulRelin -> KeySwitcher.MulAndRelinBFV, mkbfv/keyswitch.go:115-251) on its own
    # device path (mkhe_bfv_mul_relin_unhoisted): the same ciphertext bit for bit, also for the square and after the hoisted path has
    # left its cached slots behind
    un = bp.ev.mulRelin(c0, c1, rlk_d)
    assert un.ids == out.ids and (un.download() == ref).all()
    assert (bp.ev.mulRelin(c0, c0, rlk_d).download() == ref2).all()
    assert (bp.ev.MulRelinNew(c0, c1, rlk_d).download() == ref).all()


def test_missing_rlk_raises(bp):
    from mkhe_kklss_amd._abi import MkheError
    h0, c0 = bp.ct(["a", "b"])
    rlk_h, rlk_d = bp.rlk_set(["a"])
    u_h, u_d = bp.swk()
    bp.params.CRS[-1] = u_d
    with pytest.raises(MkheError, match="cannot GetRelinearizationKey"):
        bp.ev.MulRelinNew(c0, c0, rlk_d)
    with pytest.raises(MkheError, match="cannot GetRelinearizationKey"):
        bp.ev.mulRelin(c0, c0, rlk_d)


def test_add_sub(bp):
    """mkbfv/evaluator.go:27-76 on overlapping id sets"""
    h0, c0 = bp.ct(["a", "b"])
    h1, c1 = bp.ct(["b", "c"])
    add, sub = bp.ev.AddNew(c0, c1).download(), bp.ev.SubNew(c0, c1).download()
    rq = bp.bfv.ringQ
    f = lambda fn, x, y: np.stack([getattr(rq, fn)(j, x[j], y[j]) for j in range(bp.nq)])
    assert (add[0] == f("add", h0[0], h1[0])).all() and (sub[0] == f("sub", h0[0], h1[0])).all()
    assert (add[1] == h0[1]).all() and (sub[1] == h0[1]).all()                               # "a": only op0
    assert (add[2] == f("add", h0[2], h1[1])).all() and (sub[2] == f("sub", h0[2], h1[1])).all()   # "b": both
    assert (add[3] == h1[2]).all()                                                           # "c": only op1
    assert (sub[3] == np.stack([rq.neg(j, h1[2][j]) for j in range(bp.nq)])).all()


def test_exact_decrypt_property_on_device():
    """mkbfv_test.go:365-401 with valid keys: (sum_i Enc(m_i))^2 decrypts to (sum m_i)^2 exactly,
    and the device ciphertext equals the oracle's bit for bit."""
    from mkhe_kklss_amd import mkbfv
    pset = HB.small_bfv(11, 3)
    sc = HB.BFVScenario(pset, parties=3, seed=21)
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])
    ev = mkbfv.NewEvaluator(params)
    names = ["user%d" % i for i in sc.ids]
    rlk = mkbfv.NewRelinearizationKeyKeySet(params)
    for i in sc.ids:
        rlk.AddRelinearizationKey(mkbfv.RelinearizationKey(params, names[i], *sc.rlk[i]))
    params.AddCRS(-1, sc.u)
    msgs = {i: sc.message(0, 2) for i in sc.ids}
    cts = []
    for i in sc.ids:
        cts.append(mkbfv.NewCiphertext(params, [names[i]]).upload(sc.fresh_ct(msgs[i], i)))
    ct = cts[0]
    for c in cts[1:]:
        ct = ev.AddNew(ct, c)
    tot = sum(msgs.values())
    assert (sc.decrypt(sc.ids, ct.download()) == tot).all()
    res = ev.MulRelinNew(ct, ct, rlk)
    got = res.download()
    assert (sc.decrypt(sc.ids, got) == HB.negacyclic_mul_mod_t(tot, tot, sc.bfv.T)).all()
    _, ref = sc.bfv.mul_relin_new(sc.ids, ct.download(), sc.ids, ct.download(), sc.rlk, sc.u)
    assert (got == ref).all()


def test_rotate_on_bfv_context(bp):
    """mkbfv Rotate = mkrlwe.KeySwitcher.Rotate on coefficient-domain ciphertexts (evaluator.go:150-180)"""
    names = ["a", "b"]
    h, c = bp.ct(names)
    rk_h, rk_d = [], bp.mk.RotationKeySet()
    for n in names:
        hh, dd = bp.swk()
        rk_h.append(hh)
        rk_d.AddRotationKey(bp.mk.RotationKey(bp.params, 3, n, hh))
    crs_h, crs_d = bp.swk()
    bp.params.CRS[3] = crs_d
    out = bp.ev.RotateNew(c, 3, rk_d)
    galEl = pow(5, 3, 2 * bp.N)
    ref = bp.bfv.ks.rotate(bp.nq - 1, galEl, [0, 1], h, rk_h, crs_h)
    assert (out.download() == ref).all()


def test_forked_context_same_results(bp):
    """a forked engine context (own stream and pools, shared keys / ciphertexts) evaluates MulRelinNew to the same bits"""
    if bp.nq >= 14:
        pytest.skip("covered on the small chains")
    names = ["a", "b"]
    h0, c0 = bp.ct(names)
    h1, c1 = bp.ct(names)
    rlk_h, rlk_d = bp.rlk_set(names)
    u_h, u_d = bp.swk()
    bp.params.CRS[-1] = u_d
    ref = bp.ev.MulRelinNew(c0, c1, rlk_d).download()
    fork = bp.mkb.NewEvaluator(bp.params.Fork())
    fork.params.wait_for(bp.params)
    got = fork.MulRelinNew(c0, c1, rlk_d)
    bp.params.wait_for(fork.params)
    assert (got.download() == ref).all()
    fork.params.sync()

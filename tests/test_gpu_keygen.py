"""Key generation and CRS expansion on the device (-m gpu) through the C ABI
(mkhe_keygen_*, mkhe_bfv_keygen_*, mkhe_crs_expand) against the oracle restatement of mkrlwe/keygen.go,
mkbfv/keygen.go and params.go:16-99 on the same samples, bit-exact; plus the reference's exact-decryption property
(mkbfv_test.go:365-401) replayed with keys and CRS that never existed on the host."""
import numpy as np
import pytest

import harness as H
import harness_bfv as HB
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SETS = {
    "N10_q3": H.small_ckks(10, 3),
    "N12_a2_q5": H.small_alpha2(12, 5),
    "N13_q14": dict(H.PN15QP880, logN=13),       # the 14 + 2 chain: past the 128-limb threshold of the split NTT path
    "N16_a2_q3": H.small_alpha2(16, 3),          # N = 2^16: split NTT
}


class Pair:
    def __init__(self, pset, seed):
        from mkhe_kklss_amd import mkrlwe
        self.mk = mkrlwe
        self.pset = pset
        self.params = mkrlwe.Parameters(pset["logN"], pset["Q"], pset["P"], pset.get("gamma", 2))
        self.ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], pset.get("gamma", 2))
        self.okg = O.KeyGen(self.ks)
        self.rng = np.random.default_rng(seed)
        self.N, self.beta = self.ks.N, self.ks.beta_max
        self.kgen = mkrlwe.NewKeyGenerator(self.params, mkrlwe.HostSampler(np.random.default_rng(seed + 1), insecure_test_only=True))

    def ternary(self):
        return self.rng.choice(np.array([-1, 0, 0, 1], dtype=np.int32), self.N)

    def gauss(self, *count):
        return np.clip(np.rint(self.rng.normal(0, 3.2, count + (self.N,))), -19, 19).astype(np.int32)


@pytest.fixture(scope="module", params=list(SETS))
def pr(request):
    return Pair(SETS[request.param], seed=hash(request.param) & 0xfff)


@pytest.mark.parametrize("idx", [-1, 0, 4])
def test_crs_expand(pr, idx):
    crs = pr.params.AddCRS(idx, seed=0xC0FFEE + 5)
    assert (crs.download() == pr.okg.crs_expand(0xC0FFEE + 5, idx)).all()


def test_secret_key(pr):
    s = pr.ternary()
    s[:4] = [0, 1, -1, -1]
    sk = pr.kgen.GenSecretKey("a", s)
    assert (sk.Value.download()[0] == pr.okg.gen_secret_key(s)).all()
    g = pr.gauss()
    assert (pr.kgen.GenSecretKeyGaussian("a", g).Value.download()[0] == pr.okg.gen_secret_key(g)).all()


def test_switching_and_public_key(pr):
    s, e = pr.ternary(), pr.gauss(pr.beta)
    sk = pr.kgen.GenSecretKey("a", s)
    swk = pr.mk.NewSwitchingKey(pr.params)
    pr.kgen.GenSwitchingKey(sk, swk, e)
    sk_h = pr.okg.gen_secret_key(s)
    assert (swk.download() == pr.okg.gen_switching_key(sk_h, e)).all()
    a = pr.params.AddCRS(0)
    e1 = pr.gauss(1)
    pk = pr.kgen.GenPublicKey(sk, e1)
    assert (pk.Value.download() == pr.okg.gen_public_key(sk_h, e1[0], a.download()[0])).all()


def test_relin_key(pr):
    s, r, e = pr.ternary(), pr.ternary(), pr.gauss(3, pr.beta)
    a, u = pr.params.AddCRS(0), pr.params.AddCRS(-1)
    sk, rk = pr.kgen.GenSecretKey("a", s), pr.kgen.GenSecretKey("a", r)
    rlk = pr.kgen.GenRelinearizationKey(sk, rk, e)
    ref = pr.okg.gen_relin_key(pr.okg.gen_secret_key(s), pr.okg.gen_secret_key(r), e, a.download(), u.download())
    for got, exp in zip(rlk.Value, ref):
        assert (got.download() == exp).all()


@pytest.mark.parametrize("rot", [1, 8, -3])
def test_rotation_key(pr, rot):
    s, e = pr.ternary(), pr.gauss(pr.beta)
    crs = pr.params.AddCRS(rot)
    sk = pr.kgen.GenSecretKey("a", s)
    rk = pr.kgen.GenRotationKey(rot, sk, e)
    galEl = pow(5, rot % (pr.N // 2), 2 * pr.N)
    assert rk.RotIdx == rot % (pr.N // 2)
    assert (rk.Value.download() == pr.okg.gen_rotation_key(galEl, pr.okg.gen_secret_key(s), e, crs.download())).all()


def test_conjugation_key(pr):
    s, e = pr.ternary(), pr.gauss(pr.beta)
    crs = pr.params.AddCRS(-2)
    sk = pr.kgen.GenSecretKey("a", s)
    ck = pr.kgen.GenConjugationKey(sk, e)
    assert (ck.Value.download() == pr.okg.gen_conjugation_key(pr.okg.gen_secret_key(s), e, crs.download())).all()


def test_missing_crs_raises(pr):
    from mkhe_kklss_amd._abi import MkheError
    sk = pr.kgen.GenSecretKey("a")
    with pytest.raises(MkheError, match="CRS for given rot idx is not generated"):
        pr.kgen.GenRotationKey(12345, sk)
    with pytest.raises(MkheError, match="expected samples of shape"):
        pr.kgen.GenSecretKey("a", np.zeros(5, dtype=np.int32))


def test_sampler_statistics(pr):
    """the host sampler draws what lattigo's samplers draw: ternary with P(0) = 1/2, rounded Gaussian sigma 3.2, |e| <= 19"""
    smp = pr.mk.HostSampler(np.random.default_rng(3), insecure_test_only=True)
    t = smp.ternary(1 << 16)
    assert set(np.unique(t)) == {-1, 0, 1} and abs((t == 0).mean() - 0.5) < 0.02 and abs((t == 1).mean() - 0.25) < 0.02
    g = smp.gaussian(4, 1 << 14)
    assert g.dtype == np.int32 and np.abs(g).max() <= 19 and abs(g.std() - 3.2) < 0.1 and abs(g.mean()) < 0.05


@pytest.mark.parametrize("name", ["N10_q3", "N12_q4big"])
def test_bfv_relin_key(name):
    from mkhe_kklss_amd import mkbfv, mkrlwe
    pset = {"N10_q3": HB.small_bfv(10, 3), "N12_q4big": HB.small_bfv(12, 4, big=True)}[name]
    bfv = HB.make_bfv(pset)
    okg = O.KeyGen(bfv.ks)
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])
    kgen = mkbfv.NewKeyGenerator(params)
    rng = np.random.default_rng(11)
    N, beta = bfv.ks.N, bfv.ks.beta_max
    s, r = (rng.choice(np.array([-1, 0, 0, 1], dtype=np.int32), N) for _ in range(2))
    e = np.clip(np.rint(rng.normal(0, 3.2, (5, beta, N))), -19, 19).astype(np.int32)
    a1, a2, u = params.AddCRS(0), params.AddCRS(-3), params.AddCRS(-1)
    sk, rk = kgen.GenSecretKey("a", s), kgen.GenSecretKey("a", r)
    rlk = kgen.GenRelinearizationKey(sk, rk, e)
    # the gadget scalars against the harness' own big-integer computation
    hk = HB.BFVKeyGen(bfv, 1)
    for which, moduli in enumerate((bfv.Q, bfv.QMul)):
        QQMul = hk.Qprod * hk.QMulprod
        for i in range(beta):
            Gi = QQMul // moduli[i]
            G = (Gi * hk.T * pow(Gi % moduli[i], -1, moduli[i]) * hk.Pprod) // hk.QMulprod
            assert [int(x) for x in kgen.gadget(which)[i]] == [G % m for m in hk.QP]
    ref = okg.bfv_gen_relin_key(okg.gen_secret_key(s), okg.gen_secret_key(r), kgen.gadget(0), kgen.gadget(1), e,
                                a1.download(), a2.download(), u.download())
    V = rlk.Value
    got = (V[0].Value[0], V[1].Value[0], V[0].Value[1], V[1].Value[1], V[0].Value[2])
    for g, x in zip(got, ref):
        assert (g.download() == x).all()
    swk1, swk2 = mkrlwe.NewSwitchingKey(params), mkrlwe.NewSwitchingKey(params)
    kgen.GenBFVSwitchingKey(sk, swk1, swk2, e[2:4])
    assert (swk1.download() == okg.bfv_gen_switching_key(okg.gen_secret_key(s), kgen.gadget(0), e[2])).all()
    assert (swk2.download() == okg.bfv_gen_switching_key(okg.gen_secret_key(s), kgen.gadget(1), e[3])).all()


def test_bfv_exact_decrypt_with_device_generated_keys():
    """mkbfv_test.go:365-401: CRS expanded and all keys generated on the device (own random samples); fresh
    encryptions under the device-made public keys; (sum Enc(m_i))^2 evaluated on the device decrypts exactly."""
    from mkhe_kklss_amd import mkbfv
    pset = HB.small_bfv(11, 3)
    sc = HB.BFVScenario(pset, parties=3, seed=5)                 # host side: encoder, encryptor, decryptor
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])
    params.GenDefaultCRS(seed=77)
    kgen, ev = mkbfv.NewKeyGenerator(params), mkbfv.NewEvaluator(params)
    names = ["user%d" % i for i in sc.ids]
    rlkSet = mkbfv.NewRelinearizationKeyKeySet(params)
    for i in sc.ids:
        sk, pk = kgen.GenKeyPair(names[i])
        r = kgen.GenSecretKey(names[i])
        rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, r))
        sc.sk[i] = sk.Value.download()[0]
        pkh = pk.Value.download()
        sc.pk[i] = (pkh[0], pkh[1])
    msgs = {i: sc.message(0, 2) for i in sc.ids}
    ct = None
    for i in sc.ids:
        c = mkbfv.NewCiphertext(params, [names[i]]).upload(sc.fresh_ct(msgs[i], i))
        ct = c if ct is None else ev.AddNew(ct, c)
    tot = sum(msgs.values())
    assert (sc.decrypt(sc.ids, ct.download()) == tot).all()
    res = ev.MulRelinNew(ct, ct, rlkSet)
    assert (sc.decrypt(sc.ids, res.download()) == HB.negacyclic_mul_mod_t(tot, tot, sc.bfv.T)).all()

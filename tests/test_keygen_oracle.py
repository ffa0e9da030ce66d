"""Key generation / CRS expansion oracle (oracle/ora_keygen.c, restating mkrlwe/keygen.go, mkbfv/keygen.go,
mkrlwe/params.go:16-99 with the random samples as inputs) against

  * the published Philox4x32-10 known-answer vectors (Random123 kat_vectors),
  * the independent Python key generator of the test harness (tests/harness.py KeyGen, harness_bfv.py BFVKeyGen: whole-array
    numpy composition, automorphisms applied to the small secret in the coefficient domain, Python big integers for
    the BFV gadget) fed with the same samples -- the keys that harness produces are the ones the reference's property
    tests (noise bounds, exact BFV decryption) are replayed with elsewhere in this suite.
"""
import numpy as np
import pytest

import harness as H
import harness_bfv as HB
from oracle import oracle as O


def test_philox_known_answers():
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, out in kat:
        assert O.philox4x32_10(ctr, key) == out


class Recording(H.KeyGen):
    """harness KeyGen that records every Gaussian draw, in order"""

    def __init__(self, ks, seed):
        super().__init__(ks, seed)
        self.drawn = []

    def gaussian(self):
        e = super().gaussian()
        self.drawn.append(e.copy())
        return e

    def take(self):
        e, self.drawn = np.stack(self.drawn), []
        return e


class RecordingBFV(HB.BFVKeyGen):
    def __init__(self, bfv, seed):
        super().__init__(bfv, seed)
        self.drawn = []

    def gaussian(self):
        e = H.KeyGen.gaussian(self)
        self.drawn.append(e.copy())
        return e

    take = Recording.take


PSETS = [H.small_ckks(10, 3), H.small_alpha2(10, 5)]


@pytest.fixture(scope="module", params=range(len(PSETS)))
def kg(request):
    pset = PSETS[request.param]
    ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], pset.get("gamma", 2))
    return ks, Recording(ks, seed=5 + request.param), O.KeyGen(ks)


def test_crs_expand(kg):
    ks, _, okg = kg
    crs = okg.crs_expand(0x4D4B4845, -1)
    QP = ks.Q + ks.P
    # every limb canonical; the stored value is MForm(sample) and the sample stream is what ora_crs_sample defines
    for j, q in enumerate(QP):
        assert int(crs[:, j].max()) < q
    r, i = (ks.ringQ, 1)
    plain = r.invmform(i, crs[1, 1])
    for w in (0, 1, 77, ks.N - 1):
        assert int(plain[w]) == okg.crs_sample(0x4D4B4845, -1, 1 * ks.m + 1, w, QP[1])
    # different idx / seed give different polynomials; the same arguments the same
    assert (okg.crs_expand(0x4D4B4845, -1) == crs).all()
    assert (okg.crs_expand(0x4D4B4845, 0) != crs).mean() > 0.99
    assert (okg.crs_expand(0x4D4B4846, -1) != crs).mean() > 0.99
    # uniformity, coarse: mean of sample/q near 1/2, top bit populated on both sides
    u = plain.astype(np.float64) / QP[1]
    assert abs(u.mean() - 0.5) < 0.05 and 0.2 < (u > 0.5).mean() < 0.8


def test_crs_sample_rejects():
    """a modulus just above a power of two rejects about half of the candidates: still in range, deterministic"""
    okg = O.KeyGen.__new__(O.KeyGen)
    q = (1 << 20) + 7
    vals = [okg.crs_sample(1, 2, 3, w, q) for w in range(4000)]
    assert max(vals) < q and len(set(vals)) > 3900
    assert abs(np.mean(vals) / q - 0.5) < 0.03


def test_secret_and_error(kg):
    ks, hk, okg = kg
    sk_h, s = hk.gen_secret_key()
    assert (okg.gen_secret_key(s) == sk_h).all()
    e = hk.gaussian()
    assert (okg.gen_gaussian_error(e) == hk.small_to_qp(e)).all()
    hk.take()


def test_switching_and_public_key(kg):
    ks, hk, okg = kg
    sk, s = hk.gen_secret_key()
    swk_h = hk.gen_switching_key(sk)
    assert (okg.gen_switching_key(sk, hk.take()) == swk_h).all()
    a = hk.add_crs(0)
    pk0, pk1 = hk.gen_public_key(sk)
    pk = okg.gen_public_key(sk, hk.take()[0], a[0])
    assert (pk[0] == pk0).all() and (pk[1] == pk1).all()


def test_relin_key(kg):
    ks, hk, okg = kg
    sk, _ = hk.gen_secret_key()
    r, _ = hk.gen_secret_key()
    a, u = hk.add_crs(0), hk.add_crs(-1)
    b, d, v = hk.gen_relin_key(sk, r)
    ob, od, ov = okg.gen_relin_key(sk, r, hk.take().reshape(3, ks.beta_max, ks.N), a, u)
    assert (ob == b).all() and (od == d).all() and (ov == v).all()


@pytest.mark.parametrize("rot", [1, 4, 37])
def test_rotation_key(kg, rot):
    ks, hk, okg = kg
    sk, s = hk.gen_secret_key()
    hk.add_crs(rot)
    rk = hk.gen_rotation_key(rot, sk, s)
    galEl = pow(5, rot, 2 * ks.N)
    assert (okg.gen_rotation_key(galEl, sk, hk.take(), hk.CRS[rot]) == rk).all()
    # the NTT-domain index permutation alone against the coefficient-domain automorphism of the small secret
    for g in (galEl, 2 * ks.N - 1):
        exp = hk._apply("mform", hk.small_to_qp(hk.automorphism_sk(s, g)))
        assert (okg.permute_ntt_qp(g, sk) == exp).all()


def test_conjugation_key(kg):
    ks, hk, okg = kg
    sk, s = hk.gen_secret_key()
    hk.add_crs(-2)
    ck = hk.gen_conjugation_key(sk, s)
    assert (okg.gen_conjugation_key(sk, hk.take(), hk.CRS[-2]) == ck).all()


def bfv_gadget_residues(hk, moduli):
    """residues of the big-integer gadget scalars Gi (mkbfv/keygen.go:104-116,137-149)"""
    QQMul = hk.Qprod * hk.QMulprod
    g = np.zeros((hk.ks.beta_max, hk.ks.m), dtype=np.uint64)
    for i in range(hk.ks.beta_max):
        qi = moduli[i]
        Gi = QQMul // qi
        Ti = pow(Gi % qi, -1, qi)
        G = (Gi * hk.T * Ti * hk.Pprod) // hk.QMulprod
        g[i] = [G % m for m in hk.QP]
    return g


def test_bfv_relin_key():
    pset = HB.small_bfv(10, 3)
    bfv = HB.make_bfv(pset)
    hk = RecordingBFV(bfv, seed=9)
    okg = O.KeyGen(bfv.ks)
    sk, _ = hk.gen_secret_key()
    r, _ = hk.gen_secret_key()
    a1, a2, u = hk.add_crs(0), hk.add_crs(-3), hk.add_crs(-1)
    ref = hk.gen_relin_key_bfv(sk, r)
    beta, N = bfv.ks.beta_max, bfv.ks.N
    drawn = hk.take()
    # harness order: (b1_i, b2_i) interleaved, d1, d2, v  ->  oracle layout [b1, b2, d1, d2, v][beta][N]
    e = np.concatenate([drawn[0:2 * beta:2], drawn[1:2 * beta:2], drawn[2 * beta:]]).reshape(5, beta, N)
    g1, g2 = bfv_gadget_residues(hk, bfv.Q), bfv_gadget_residues(hk, bfv.QMul)
    got = okg.bfv_gen_relin_key(sk, r, g1, g2, e, a1, a2, u)
    for x, y in zip(got, ref):
        assert (x == y).all()

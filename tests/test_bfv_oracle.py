"""MK-BFV oracle checks (CPU): the C restatement of mkbfv/{basis_extension,keyswitch,keyswitch_hoisted,
evaluator}.go against an independent big-integer model and against the reference's own property tests

  mkbfv_test.go:282-301  testEncAndDec        decrypt(encrypt(m)) == m            (exact)
  mkbfv_test.go:365-401  testEvaluatorMul     (sum_i ct_i)^2 decrypts to (sum m_i)^2 (exact)

replayed with seeded valid keys (tests/harness_bfv.py).
"""
import numpy as np
import pytest

import harness as H
import harness_bfv as HB
from oracle import pymodel as M

PSET = HB.small_bfv(10, 3)


@pytest.fixture(scope="module")
def sc():
    return HB.BFVScenario(PSET, parties=2, seed=11)


def _crt_cols(poly, moduli, idx):
    return [M.crt([int(poly[l][k]) for l in range(len(moduli))], moduli)[0] for k in idx]


def test_modup_q_to_r_literal_vs_model():
    """ModUpQtoR = copy + literal modUpExact Q -> QMul (lazy representatives included)"""
    bfv = HB.make_bfv(PSET)
    rng = np.random.default_rng(1)
    N = bfv.N
    x = H.uniform_poly(rng, bfv.Q, N)
    # plant coefficients whose float64 correction index sits on an integer boundary (x = 0, x = Q-1)
    x[:, 0] = 0
    x[:, 1] = np.array([q - 1 for q in bfv.Q], dtype=np.uint64)
    r = bfv.modup_q_to_r(x)
    assert (r[: bfv.nq] == x).all()
    for k in list(range(8)) + [N - 1]:
        exp, _ = M.modup_literal([int(x[l][k]) for l in range(bfv.nq)], bfv.Q, bfv.QMul)
        assert [int(r[bfv.nq + j][k]) for j in range(bfv.nq)] == exp, k


def test_rescale_and_quantize_semantics():
    """Rescale: x -> ~round(QMul/Q * x) over R; Quantize: NTT_R(x) -> ~round(t/QMul * x) over Q.
    The fast base conversions are exact up to the well-known +-1 (float correction index) slack."""
    bfv = HB.make_bfv(PSET)
    rng = np.random.default_rng(2)
    N, nq = bfv.N, bfv.nq
    Qp, QMp = 1, 1
    for q in bfv.Q:
        Qp *= q
    for q in bfv.QMul:
        QMp *= q
    R = bfv.Q + bfv.QMul
    idx = list(range(6))
    # -- Rescale
    x = H.uniform_poly(rng, bfv.Q, N)
    r = bfv.rescale(x)
    xv = _crt_cols(x, bfv.Q, idx)
    for n, k in enumerate(idx):
        got = M.crt([int(r[l][k]) % R[l] for l in range(2 * nq)], R)[0]
        cen = xv[n]                                    # [x]_Q in [0, Q)
        # ModDownQPtoP of (x*QMul mod Q, 0): (0 - [x*QMul]_Q)/Q mod QMul = floor(x*QMul/Q)
        want = (cen * QMp) // Qp
        # the value is defined modulo QMul only; the Q part is its lift (ModUpPtoQ)
        gm = M.crt([int(r[nq + l][k]) for l in range(nq)], bfv.QMul)[0]
        assert min((gm - want) % QMp, (want - gm) % QMp) <= 2, k
        lift = [int(r[l][k]) % bfv.Q[l] for l in range(nq)]
        assert any(lift == [(gm + e * QMp) % q for q in bfv.Q] for e in (0, 1, -1, 2)), k
        assert got % QMp == gm
    # -- Quantize
    y = np.stack([H.uniform_poly(rng, [m], N)[0] for m in R])
    qz = bfv.quantize(bfv.ntt_r(y))
    yv = _crt_cols(y, R, idx)
    for n, k in enumerate(idx):
        got = M.crt([int(qz[l][k]) for l in range(nq)], bfv.Q)[0]
        ty = (yv[n] * bfv.T) % (Qp * QMp)
        want = (ty // QMp) % Qp                        # ((ty mod Q) - [ty]_QMul)/QMul
        assert min((got - want) % Qp, (want - got) % Qp) <= 2, k


def test_decompose_bfv_is_digit_spread():
    bfv = HB.make_bfv(PSET)
    rng = np.random.default_rng(3)
    ar = np.stack([H.uniform_poly(rng, [m], bfv.N)[0] for m in bfv.Q + bfv.QMul])
    ad1, ad2 = bfv.decompose(ar)
    ks = bfv.ks
    for d in range(2 * bfv.nq):
        got = ad1[d] if d < bfv.nq else ad2[d - bfv.nq]
        for j, q in enumerate(bfv.Q + bfv.P):
            ring, i = (ks.ringQ, j) if j < bfv.nq else (ks.ringP, j - bfv.nq)
            assert (got[j] == ring.ntt(i, ar[d])).all()


def test_enc_dec_exact(sc):
    for i in sc.ids:
        m = sc.message(-(sc.bfv.T // 4), sc.bfv.T // 4)
        ct = sc.fresh_ct(m, i)
        assert (sc.decrypt([i], ct) == m).all()


@pytest.mark.parametrize("hoisted", [True, False])
def test_mul_relin_of_sum_squared_exact(sc, hoisted):
    """mkbfv_test.go:365-401"""
    msgs = {i: sc.message(0, 2) for i in sc.ids}
    ct = sc.sum_ct(msgs)
    tot = sum(msgs.values())
    assert (sc.decrypt(sc.ids, ct) == tot).all()
    ids_out, out = sc.bfv.mul_relin_new(sc.ids, ct, sc.ids, ct, sc.rlk, sc.u, hoisted=hoisted)
    assert ids_out == sc.ids
    want = HB.negacyclic_mul_mod_t(tot, tot, sc.bfv.T)
    assert (sc.decrypt(ids_out, out) == want).all()


def test_mul_relin_disjoint_id_sets_exact(sc):
    """ct of party 0 times ct of party 1: the union id set, every tensor branch of keyswitch_hoisted.go:138-166"""
    m0, m1 = sc.message(-3, 4), sc.message(-3, 4)
    ct0, ct1 = sc.fresh_ct(m0, 0), sc.fresh_ct(m1, 1)
    ids_out, out = sc.bfv.mul_relin_new([0], ct0, [1], ct1, sc.rlk, sc.u)
    assert ids_out == [0, 1]
    assert (sc.decrypt(ids_out, out) == HB.negacyclic_mul_mod_t(m0, m1, sc.bfv.T)).all()
    ids_out, out2 = sc.bfv.mul_relin_new([0], ct0, [1], ct1, sc.rlk, sc.u, hoisted=False)
    assert (out == out2).all()


def test_hoisted_equals_plain_on_uniform_inputs():
    """MulAndRelinBFVHoisted == MulAndRelinBFV bit for bit (uniform, non-key inputs)"""
    bfv = HB.make_bfv(PSET)
    d = HB.uniform_bfv_inputs(PSET, 2, 7)
    _, a = bfv.mul_relin_new([0, 1], d["op0"], [0, 1], d["op1"], d["rlk"], d["u"], hoisted=True)
    _, b = bfv.mul_relin_new([0, 1], d["op0"], [0, 1], d["op1"], d["rlk"], d["u"], hoisted=False)
    assert (a == b).all()
    for l, q in enumerate(bfv.Q):
        assert (a[:, l] < q).all()


# ---------------------------------------------------------------- C oracle vs the independent Python model
MODEL_SET = dict(logN=4, Q=HB.BFV_PN15QP880["Q"][:2], QMul=HB.BFV_PN15QP880["QMul"][:2], P=HB.BFV_PN15QP880["P"], T=65537)


def _model_pair():
    # the reference's primes are = 1 mod 2^16 >= 2N for every N <= 2^15, so they serve N = 16 as well
    return HB.make_bfv(MODEL_SET), M.BfvModel(4, MODEL_SET["Q"], MODEL_SET["QMul"], MODEL_SET["P"], MODEL_SET["T"])


def test_conversions_vs_model():
    bfv, mdl = _model_pair()
    rng = np.random.default_rng(31)
    for _ in range(4):
        x = H.uniform_poly(rng, bfv.Q, bfv.N)
        assert bfv.modup_q_to_r(x).tolist() == mdl.modup_q_to_r(x)
        assert bfv.rescale(x).tolist() == mdl.rescale(x)
        y = np.stack([H.uniform_poly(rng, [m], bfv.N)[0] for m in bfv.Q + bfv.QMul])
        assert bfv.quantize(bfv.ntt_r(y)).tolist() == mdl.quantize_coeff(y)


@pytest.mark.parametrize("ids0,ids1", [([0, 1], [0, 1]), ([0], [1]), ([0, 1], [1, 2])])
def test_mul_relin_new_vs_model(ids0, ids1):
    bfv, mdl = _model_pair()
    rng = np.random.default_rng(hash((tuple(ids0), tuple(ids1))) & 0xffff)
    swk = lambda: np.stack([H.uniform_poly(rng, bfv.Q + bfv.P, bfv.N) for _ in range(bfv.nq)])
    ct = lambda n: np.stack([H.uniform_poly(rng, bfv.Q, bfv.N) for _ in range(1 + n)])
    op0, op1 = ct(len(ids0)), ct(len(ids1))
    rlk = {i: tuple(swk() for _ in range(5)) for i in sorted(set(ids0) | set(ids1))}
    u = swk()
    ido, out = bfv.mul_relin_new(ids0, op0, ids1, op1, rlk, u)
    idm, outm = mdl.mul_relin_new(ids0, op0, ids1, op1, rlk, u)
    assert ido == idm
    assert out.tolist() == outm

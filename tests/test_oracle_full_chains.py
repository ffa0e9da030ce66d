"""CPU: the C oracle against the independent big-integer model on the FULL prime chains of every parameter set the reference
ships (test_oracle_vs_model.py pins the pair on the first three Q primes only) -- PN15QP880 (14 + 2),
PN14QP439 (6 + 2), PN16QP1761 (34 + 4, alpha = 2), the mkbfv chains (Q, QMul, P) and the cnn chain PN14QP433 (7 + 2), all at
N = 16 (the primes are = 1 mod 2^16 or 2^17, so every one of them supports the small ring): per modulus the lattigo root rule,
the NTT against its definition, the NttPsi / NttPsiInv table layout, Montgomery constants, and per chain Rescale, ModDown
(exact floor division), Decompose, ExternalProduct and one full MulAndRelin.

Parity against the Go reference itself stays unpinned (no Go toolchain, no lattigo sources, no reference-held vectors:
SURVEY.md 8c); this file removes "only a few moduli were ever checked" from the list of reasons."""
import numpy as np
import pytest

import harness as H
import harness_bfv as HB
import harness_cnn as HC
from oracle import oracle as O, pymodel as M

LOGN = 4
N = 1 << LOGN

CHAINS = {
    "PN15QP880": (H.PN15QP880["Q"], H.PN15QP880["P"]),
    "PN14QP439": (H.PN14QP439["Q"], H.PN14QP439["P"]),
    "PN16QP1761": (H.PN16QP1761["Q"], H.PN16QP1761["P"]),
    "BFV_PN15_Q": (HB.BFV_PN15QP880["Q"], HB.BFV_PN15QP880["P"]),
    "BFV_PN15_QMul": (HB.BFV_PN15QP880["QMul"], HB.BFV_PN15QP880["P"]),
    "BFV_PN14_Q": (HB.BFV_PN14QP439["Q"], HB.BFV_PN14QP439["P"]),
    "BFV_PN14_QMul": (HB.BFV_PN14QP439["QMul"], HB.BFV_PN14QP439["P"]),
    "cnn_PN14QP433": (HC.PN14QP433["Q"], HC.PN14QP433["P"]),
}
NATIVE_LOGN = {"PN15QP880": 15, "PN14QP439": 14, "PN16QP1761": 16, "BFV_PN15_Q": 15, "BFV_PN15_QMul": 15, "BFV_PN14_Q": 14,
               "BFV_PN14_QMul": 14, "cnn_PN14QP433": 14}


def rnd_poly(rng, mods):
    return np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods])


def to_int(p):
    return [[int(v) for v in l] for l in p]


@pytest.mark.parametrize("name", list(CHAINS))
def test_every_modulus_root_ntt_tables_montgomery(name):
    Q, P = CHAINS[name]
    mods = Q + P
    rng = np.random.default_rng(len(name))
    # the root the engine and the oracle use at the set's own ring degree is lattigo's (first generator from g = 3)
    big = O.Ring(NATIVE_LOGN[name], mods)
    for i, q in enumerate(mods):
        assert big.psi(i) == M.find_psi(q, 1 << NATIVE_LOGN[name])
        assert pow(big.psi(i), 1 << NATIVE_LOGN[name], q) == q - 1
    r = O.Ring(LOGN, mods)
    for i, q in enumerate(mods):
        psi = r.psi(i)
        assert psi == M.find_psi(q, N)
        a = rng.integers(0, q, N, dtype=np.uint64)
        A = r.ntt(i, a)
        assert [int(v) for v in A] == M.ntt_def([int(v) for v in a], q, psi, LOGN)
        assert (r.intt(i, A) == a).all()
        lz = r.intt_lazy(i, A)
        assert (lz < 2 * q).all() and ((lz % np.uint64(q)) == a).all()
        t, ti = r.psi_table(i), r.psi_table(i, inverse=True)
        for j in range(N):
            assert int(t[j]) == M.mform(pow(psi, M.bitrev(j, LOGN), q), q)
            assert int(ti[j]) == M.mform(pow(psi, -M.bitrev(j, LOGN), q), q)
        b = rng.integers(0, q, N, dtype=np.uint64)
        Rinv = pow(M.R, -1, q)
        assert [int(v) for v in r.mul(i, a, b)] == [int(x) * int(y) * Rinv % q for x, y in zip(a, b)]
        assert [int(v) for v in r.mform(i, a)] == [M.mform(int(x), q) for x in a]
        assert [int(v) for v in r.invmform(i, a)] == [M.invmform(int(x), q) for x in a]


@pytest.mark.parametrize("name", list(CHAINS))
def test_rescale_and_moddown_over_the_whole_chain(name):
    Q, P = CHAINS[name]
    rng = np.random.default_rng(7 + len(name))
    r = O.Ring(LOGN, Q)
    p = rnd_poly(rng, Q)
    cur, mods = p, list(Q)
    for _ in range(min(3, len(Q) - 1)):                      # three successive rescales: every RescaleParams row that they touch
        out, _ = O.Ring(LOGN, mods).div_round_last_many(cur, 1)
        for k in range(N):
            assert [int(out[l][k]) for l in range(len(mods) - 1)] == M.div_round_last([int(cur[l][k]) for l in range(len(mods))], mods)
        cur, mods = out, mods[:-1]
    ra, rb = O.Ring(LOGN, Q), O.Ring(LOGN, P)
    fbe = O.BasisExtender(ra, rb)
    xq, xp = rnd_poly(rng, Q), rnd_poly(rng, P)
    out = fbe.moddown_ab2a(xq, xp)
    for k in range(N):
        assert [int(out[j][k]) for j in range(len(Q))] == M.moddown_exact([int(xq[j][k]) for j in range(len(Q))],
                                                                          [int(xp[j][k]) for j in range(len(P))], Q, P)


@pytest.mark.parametrize("name", ["PN15QP880", "PN14QP439", "PN16QP1761", "BFV_PN15_Q", "cnn_PN14QP433"])
def test_keyswitch_over_the_whole_chain(name):
    """Decompose / ExternalProduct at the maximum and a middle level, and one 2-party MulAndRelin, with every prime of the chain."""
    Q, P = CHAINS[name]
    rng = np.random.default_rng(11 + len(name))
    ks = O.KeySwitcher(LOGN, Q, P, 2)
    mdl = M.Model(LOGN, Q, P, 2)
    assert ks.alpha == mdl.alpha and ks.beta_max == mdl.beta(len(Q) - 1)

    def rnd_swk():
        s = ks.new_swk()
        for i in range(ks.beta_max):
            s[i] = rnd_poly(rng, Q + P)
        return s

    for level in (len(Q) - 1, len(Q) // 2):
        a = rnd_poly(rng, Q)
        h, hm = ks.decompose(level, a), mdl.decompose(a, level)
        for i in range(ks.beta(level)):
            for j in mdl.limb_index(level):
                assert [int(v) for v in h[i][j]] == hm[i][j]
        bg = rnd_swk()
        c = ks.external_product(level, a, bg)
        assert to_int(c) == mdl.external_product(to_int(a), bg, level)
    level = len(Q) - 1
    ids0, ids1 = [0, 1], [1]
    op0 = np.stack([rnd_poly(rng, Q) for _ in range(3)])
    op1 = np.stack([rnd_poly(rng, Q) for _ in range(2)])
    rlk = {i: (rnd_swk(), rnd_swk(), rnd_swk()) for i in range(2)}
    u = rnd_swk()
    ido, out = ks.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u)
    ido2, outm = mdl.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u)
    assert ido == ido2 and [to_int(p) for p in out] == outm


@pytest.mark.parametrize("pset_name", ["BFV_PN15QP880", "BFV_PN14QP439"])
def test_bfv_full_chain_vs_model(pset_name):
    """mkbfv on the full Q / QMul / P chains at N = 16: ModUpQtoR, Rescale, Quantize and one MulRelinNew against BfvModel
    (literal modUpExact, schoolbook tensor over R, double gadget) -- every QMul and 55-bit tail prime takes part."""
    src = getattr(HB, pset_name)
    pset = dict(logN=LOGN, Q=src["Q"], QMul=src["QMul"], P=src["P"], T=src["T"])
    bfv = HB.make_bfv(pset)
    mdl = M.BfvModel(LOGN, pset["Q"], pset["QMul"], pset["P"], pset["T"])
    rng = np.random.default_rng(len(pset_name))
    x = H.uniform_poly(rng, bfv.Q, bfv.N)
    assert bfv.modup_q_to_r(x).tolist() == mdl.modup_q_to_r(x)
    assert bfv.rescale(x).tolist() == mdl.rescale(x)
    y = np.stack([H.uniform_poly(rng, [m], bfv.N)[0] for m in bfv.Q + bfv.QMul])
    assert bfv.quantize(bfv.ntt_r(y)).tolist() == mdl.quantize_coeff(y)
    swk = lambda: np.stack([H.uniform_poly(rng, bfv.Q + bfv.P, bfv.N) for _ in range(bfv.nq)])
    ct = lambda n: np.stack([H.uniform_poly(rng, bfv.Q, bfv.N) for _ in range(1 + n)])
    ids0, ids1 = [0, 1], [1]
    op0, op1 = ct(2), ct(1)
    rlk = {i: tuple(swk() for _ in range(5)) for i in (0, 1)}
    u = swk()
    ido, out = bfv.mul_relin_new(ids0, op0, ids1, op1, rlk, u)
    idm, outm = mdl.mul_relin_new(ids0, op0, ids1, op1, rlk, u)
    assert ido == idm and out.tolist() == outm

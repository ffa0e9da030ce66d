"""CPU: the C oracle against the independent Python big-integer model (oracle/pymodel.py).

The model states every operation by its mathematical definition (NTT as polynomial evaluation,
ModDown as exact floor division after CRT reconstruction, negacyclic schoolbook products); the
oracle restates the reference's limb-level algorithms.  Agreement on random inputs pins the oracle
to the mathematics; the literal float64 / lazy-representative paths are pinned separately.
"""
import numpy as np
import pytest

import harness as H
from oracle import oracle as O, pymodel as M

Q3 = H.PN15QP880["Q"][:3]
P2 = H.PN15QP880["P"]
LOGN = 4
N = 1 << LOGN


def rnd_poly(rng, mods):
    return np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods])


def to_int(p):
    return [[int(v) for v in l] for l in p]


def test_psi_rule_matches_for_all_shipped_primes():
    """lattigo NewRing root choice (g scans from 3) for every prime of the reference's sets."""
    for pset in (H.PN15QP880, H.PN14QP439):
        logN = pset["logN"]
        r = O.Ring(logN, pset["Q"] + pset["P"])
        for i, q in enumerate(pset["Q"] + pset["P"]):
            psi = r.psi(i)
            assert psi == M.find_psi(q, 1 << logN)
            assert pow(psi, 1 << logN, q) == q - 1


def test_ntt_is_the_definition():
    rng = np.random.default_rng(1)
    r = O.Ring(LOGN, Q3 + P2)
    for i, q in enumerate(Q3 + P2):
        a = rng.integers(0, q, N, dtype=np.uint64)
        A = r.ntt(i, a)
        assert [int(v) for v in A] == M.ntt_def([int(v) for v in a], q, r.psi(i), LOGN)
        assert (r.intt(i, A) == a).all()
        lz = r.intt_lazy(i, A)
        assert (lz < 2 * q).all() and ((lz % np.uint64(q)) == a).all()
        big = rng.integers(0, 1 << 60, N, dtype=np.uint64)          # unreduced digit copies
        assert [int(v) for v in r.ntt(i, big)] == M.ntt_def([int(v) % q for v in big], q, r.psi(i), LOGN)


def test_ntt_tables_follow_lattigo_layout():
    """NttPsi[j] = psi^bitrev(j) * 2^64 mod q ; NttPsiInv likewise."""
    r = O.Ring(LOGN, Q3)
    for i, q in enumerate(Q3):
        psi = r.psi(i)
        t, ti = r.psi_table(i), r.psi_table(i, inverse=True)
        for j in range(N):
            assert int(t[j]) == M.mform(pow(psi, M.bitrev(j, LOGN), q), q)
            assert int(ti[j]) == M.mform(pow(psi, -M.bitrev(j, LOGN), q), q)


def test_pointwise_ops():
    rng = np.random.default_rng(2)
    r = O.Ring(LOGN, Q3)
    for i, q in enumerate(Q3):
        a, b, z = (rng.integers(0, q, N, dtype=np.uint64) for _ in range(3))
        Rinv = pow(M.R, -1, q)
        assert [int(v) for v in r.mul(i, a, b)] == [int(x) * int(y) * Rinv % q for x, y in zip(a, b)]
        assert [int(v) for v in r.mul_add(i, a, b, z)] == [(int(w) + int(x) * int(y) * Rinv) % q for x, y, w in zip(a, b, z)]
        assert [int(v) for v in r.mform(i, a)] == [M.mform(int(x), q) for x in a]
        assert [int(v) for v in r.invmform(i, a)] == [M.invmform(int(x), q) for x in a]
        assert [int(v) for v in r.mul_scalar(i, a, 12345678901234567)] == [int(x) * 12345678901234567 % q for x in a]
        assert int(r.neg(i, np.zeros(N, dtype=np.uint64))[0]) == q        # Neg maps 0 -> q (App. A.2)


@pytest.mark.parametrize("pset", ["alpha1", "alpha2"])
def test_decompose_extprod_mulrelin(pset):
    rng = np.random.default_rng(3)
    if pset == "alpha1":
        Qs, Ps = Q3, P2
    else:
        Qs, Ps = H.PN16_Q[:5], H.PN16_P
    ks = O.KeySwitcher(LOGN, Qs, Ps, 2)
    mdl = M.Model(LOGN, Qs, Ps, 2)
    assert ks.alpha == mdl.alpha

    def rnd_swk():
        s = ks.new_swk()
        for i in range(ks.beta_max):
            s[i] = rnd_poly(rng, Qs + Ps)
        return s

    for level in range(len(Qs) - 1, -1, -1):
        a = rnd_poly(rng, Qs)
        h, hm = ks.decompose(level, a), mdl.decompose(a, level)
        for i in range(ks.beta(level)):
            for j in mdl.limb_index(level):
                assert [int(v) for v in h[i][j]] == hm[i][j]
        bg = rnd_swk()
        c = ks.external_product(level, a, bg)
        assert to_int(c) == mdl.external_product(to_int(a), bg, level)
        assert (ks.external_product_hoisted(level, h, bg) == c).all()
    ids0, ids1 = [0, 1], [1, 2]
    op0 = np.stack([rnd_poly(rng, Qs) for _ in range(3)])
    op1 = np.stack([rnd_poly(rng, Qs) for _ in range(3)])
    rlk = {i: (rnd_swk(), rnd_swk(), rnd_swk()) for i in range(3)}
    u = rnd_swk()
    for level in (len(Qs) - 1, 1):
        ido, out = ks.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u)
        ido2, outm = mdl.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u)
        assert ido == ido2 and [to_int(p) for p in out] == outm
        h0 = {i: ks.decompose(level, op0[1 + a]) for a, i in enumerate(ids0)}
        h1 = {i: ks.decompose(level, op1[1 + a]) for a, i in enumerate(ids1)}
        assert (ks.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u, h0, h1)[1] == out).all()


def test_decompose_split_literal_lazy_representatives():
    """alpha = 2: DecomposeAndSplit writes multSum's un-reduced representative (App. D-2a)."""
    rng = np.random.default_rng(4)
    Qs, Ps = H.PN16_Q[:5], H.PN16_P
    ks = O.KeySwitcher(LOGN, Qs, Ps, 2)
    a = rnd_poly(rng, Qs)
    for level in (4, 2, 1):
        oq, op = ks.decompose_and_split(level, 0, a)
        for k in range(N):
            lit, _ = M.modup_literal([int(a[0][k]), int(a[1][k])], Qs[:2], Qs + Ps)
            assert [int(oq[j][k]) for j in range(level + 1)] == lit[: level + 1]
            assert [int(op[j][k]) for j in range(4)] == lit[len(Qs):]


def test_modup_float_index_edge_cases():
    """reconstructRNS: v = uint64(sum float64(y_i)/float64(p_i)); exercise residues at the ends of
    the range where the float sum sits next to an integer."""
    ra, rb = O.Ring(LOGN, Q3), O.Ring(LOGN, P2)
    fbe = O.BasisExtender(ra, rb)
    Pp = P2[0] * P2[1]
    vals = [0, 1, 2, Pp - 1, Pp - 2, P2[0], P2[1], Pp // 2, Pp // 2 + 1, (Pp * 3) // 4] + [0] * (N - 10)
    pb = np.stack([np.array([v % p for v in vals], dtype=np.uint64) for p in P2])
    lifted = fbe.modup_b2a(pb)
    for k in range(N):
        lit, v = M.modup_literal([int(pb[0][k]), int(pb[1][k])], P2, Q3)
        assert [int(lifted[j][k]) for j in range(3)] == lit
        # mathematically the lift is the value itself, up to one multiple of P when the float index
        # lands on the wrong side of an integer
        errs = {tuple((x - vals[k] - e * Pp) % q for x, q in zip(lit, Q3)) for e in (-1, 0, 1)}
        assert (0, 0, 0) in errs


def test_moddown_is_exact_floor_division():
    rng = np.random.default_rng(5)
    ra, rb = O.Ring(LOGN, Q3), O.Ring(LOGN, P2)
    fbe = O.BasisExtender(ra, rb)
    xq, xp = rnd_poly(rng, Q3), rnd_poly(rng, P2)
    out = fbe.moddown_ab2a(xq, xp)
    for k in range(N):
        assert [int(out[j][k]) for j in range(3)] == M.moddown_exact([int(xq[j][k]) for j in range(3)], [int(xp[j][k]) for j in range(2)], Q3, P2)


def test_rescale_and_permute():
    rng = np.random.default_rng(6)
    Qs = H.PN16_Q[:5]
    r = O.Ring(LOGN, Qs)
    p = rnd_poly(rng, Qs)
    out, mutated = r.div_round_last_many(p, 1)
    for k in range(N):
        assert [int(out[l][k]) for l in range(4)] == M.div_round_last([int(p[l][k]) for l in range(5)], Qs)
    h = (Qs[4] - 1) >> 1
    assert [int(v) for v in mutated[4]] == [(int(x) + h) % Qs[4] for x in p[4]]      # lattigo mutates the last limb
    out2, _ = r.div_round_last_many(p, 2)
    for k in range(N):
        assert [int(out2[l][k]) for l in range(3)] == M.div_round_last(M.div_round_last([int(p[l][k]) for l in range(5)], Qs), Qs[:4])
    g = pow(5, 3, 2 * N)
    pp = r.permute(g, p)
    for l in range(5):
        assert [int(v) for v in pp[l]] == M.automorphism([int(v) for v in p[l]], g, Qs[l], LOGN)

"""Parity tests proper (-m gpu): HIP path through the C ABI vs the CPU oracle, bit-exact.

Mirrors the reference's own tests of this path -- mkrlwe_test.go:456-679 (Decompose,
ExternalProduct), mkckks_test.go:320-362 (MulRelin) -- with seeded inputs.  The reference asserts
noise bounds; here the stronger statement "identical uint64 outputs to the restated algorithm"
is checked on uniform inputs (arithmetic does not depend on inputs being valid encryptions).
"""
import numpy as np
import pytest

import harness as H
from gpu_common import Pair, oracle_mul_and_relin

pytestmark = pytest.mark.gpu

SETS = {
    "N10_q3": H.small_ckks(10, 3),
    "N11_q2": H.small_ckks(11, 2),
    "N12_q4": H.small_ckks(12, 4),
    "N13_q3": H.small_ckks(13, 3),
    "N14_pn14": H.PN14QP439,
    # alpha = 2 (4 special primes, gamma = 2): CRT-reconstructed gadget digits (basis_extension.go:471-533);
    # odd limb counts end in a one-limb digit (copy path) at the maximum level
    "N11_a2_q5": H.small_alpha2(11, 5),
    "N12_a2_q4": H.small_alpha2(12, 4),
    # N = 2^16 (BASELINE.json configs[3] shape, head of the reference's PN16 chain): split NTT
    "N16_a2_q3": H.small_alpha2(16, 3),
}


@pytest.fixture(scope="module", params=list(SETS))
def pair(request):
    return Pair(SETS[request.param], seed=hash(request.param) & 0xffff)


def test_ntt_roundtrip_and_oracle(pair):
    """lattigo ring.NTT / InvNTT / InvNTTLazy on every modulus of Q and P."""
    mk, ks = pair.mk, pair.ks
    mods = pair.Q + pair.P
    a = np.stack([H.uniform_poly(pair.rng, mods, pair.N) for _ in range(2)])
    src = mk.DeviceLimbs(pair.params, 2, len(mods)).upload(a)
    dst = mk.DeviceLimbs(pair.params, 2, len(mods))
    mk.ntt(pair.params, src, dst)
    got = dst.download()
    for c in range(2):
        for j in range(len(mods)):
            r, i = (ks.ringQ, j) if j < len(pair.Q) else (ks.ringP, j - len(pair.Q))
            assert (got[c][j] == r.ntt(i, a[c][j])).all()
    back = mk.DeviceLimbs(pair.params, 2, len(mods))
    mk.ntt(pair.params, dst, back, inverse=True)
    assert (back.download() == a).all()
    mk.ntt(pair.params, dst, back, inverse=True, lazy=True)
    lz = back.download()
    for j, q in enumerate(mods):
        assert (lz[:, j] < 2 * q).all() and ((lz[:, j] % np.uint64(q)) == a[:, j]).all()


def test_ntt_unreduced_input(pair):
    """forward NTT of values that are not reduced mod q (the digit spread of Decompose copies
    60-bit limbs under 54-bit moduli; lattigo's final BRedAdd makes the output canonical)."""
    mk, ks = pair.mk, pair.ks
    nq = len(pair.Q)
    a = np.stack([np.stack([pair.rng.integers(0, 4 * q, pair.N, dtype=np.uint64) for q in pair.Q])])
    src = mk.DeviceLimbs(pair.params, 1, nq).upload(a)
    dst = mk.DeviceLimbs(pair.params, 1, nq)
    mk.ntt(pair.params, src, dst)
    got = dst.download()
    for j in range(nq):
        assert (got[0][j] == ks.ringQ.ntt(j, a[0][j])).all()


@pytest.mark.parametrize("drop", [0, 1])
def test_decompose(pair, drop):
    """KeySwitcher.Decompose (mkrlwe_test.go:507-610: max level and a lower level)."""
    level = pair.maxlevel - drop
    if level < 0:
        pytest.skip("single limb")
    host, dev = pair.ct(["u0"], pair.maxlevel)
    ad = pair.mk.NewSwitchingKey(pair.params)
    pair.ksw.Decompose(level, dev, "u0", ad)
    ref = pair.ks.decompose(level, host[1][: level + 1])
    got = ad.download()
    beta = pair.ks.beta(level)
    act = list(range(level + 1)) + [len(pair.Q) + j for j in range(len(pair.P))]
    assert (got[:beta][:, act] == ref[:beta][:, act]).all()


def test_decompose_ntt_input(pair):
    """Decompose with a.IsNTT == true (keyswitch.go:55-61)."""
    level = pair.maxlevel
    host, dev = pair.ct([], level)
    ad = pair.mk.NewSwitchingKey(pair.params)
    pair.ksw.Decompose(level, dev, "0", ad, is_ntt=True)
    ref = pair.ks.decompose(level, host[0], is_ntt=True)
    assert (ad.download() == ref).all()


@pytest.mark.parametrize("drop", [0, 1])
def test_external_product(pair, drop):
    """ExternalProduct and ExternalProductHoisted (mkrlwe_test.go:456-505)."""
    level = pair.maxlevel - drop
    if level < 0:
        pytest.skip("single limb")
    host, dev = pair.ct(["u0"], pair.maxlevel)
    bg_h, bg_d = pair.swk()
    out = pair.mk.NewCiphertext(pair.params, ["u0"], level)
    pair.ksw.ExternalProduct(level, dev, "u0", bg_d, out, "0")
    ref = pair.ks.external_product(level, host[1][: level + 1], bg_h)
    assert (out.download()[0] == ref).all()
    ad = pair.mk.NewSwitchingKey(pair.params)
    pair.ksw.Decompose(level, dev, "u0", ad)
    pair.ksw.ExternalProductHoisted(level, ad, bg_d, out, "u0")
    assert (out.download()[1] == ref).all()


CASES = [
    # (ids0, ids1, level drop of out, input extra limbs, hoisted)
    (["a", "b"], ["a", "b"], 0, 0, False),
    (["a", "b"], ["a", "b"], 0, 0, True),
    (["a", "b"], ["b", "c"], 0, 0, False),
    (["a"], ["b"], 0, 0, True),
    (["a", "b", "c"], ["b"], 1, 1, False),
    (["a", "b"], ["a", "b"], 1, 0, True),
    # degenerate id sets (a ciphertext may consist of c_0 only) and a larger party count
    ([], [], 0, 0, False),
    ([], ["a"], 0, 0, False),
    (["a"], [], 0, 0, True),
    (["p%d" % i for i in range(7)], ["p%d" % i for i in range(3, 9)], 0, 0, False),
    # five, six and eight parties in op0: x as the by-product of step F1 in its wide form (ext_inner_xwide_kernel; seven is the case above)
    (["p%d" % i for i in range(5)], ["p0", "p5"], 0, 0, True),
    (["p%d" % i for i in range(6)], ["p%d" % i for i in range(6)], 1, 0, False),
    (["p%d" % i for i in range(8)], ["p%d" % i for i in range(2, 6)], 0, 0, False),
    (["p%d" % i for i in range(10)], ["p1", "p9"], 0, 0, False),
    (["p%d" % i for i in range(13)], ["p%d" % i for i in range(13)], 0, 0, True),
    (["p%d" % i for i in range(16)], ["p3"], 1, 0, False),
    # as many parties in op1 as in op0, up to four: y computed inside the F1 kernel (ext_inner_xy_kernel<1..4>, round 4) -- equal, overlapping and
    # disjoint id sets (one and two parties are the cases at the top)
    (["p0", "p1", "p2"], ["p0", "p1", "p2"], 0, 0, False),
    (["p0", "p1", "p2", "p3"], ["p0", "p1", "p2", "p3"], 0, 0, True),
    (["p0", "p1", "p2"], ["p1", "p2", "p3"], 1, 0, True),
    (["p0", "p1", "p2", "p3"], ["p4", "p5", "p6", "p7"], 0, 0, False),
    # ... and five to eight parties per operand (ext_inner_xy_wide_kernel<5..8>; six equal parties is a case above)
    (["p%d" % i for i in range(5)], ["p%d" % i for i in range(3, 8)], 0, 0, False),
    (["p%d" % i for i in range(7)], ["p%d" % i for i in range(7)], 1, 0, True),
    (["p%d" % i for i in range(8)], ["p%d" % i for i in range(8)], 0, 0, False),
]


@pytest.mark.parametrize("ids0,ids1,drop,extra,hoisted", CASES)
def test_mul_and_relin(pair, ids0, ids1, drop, extra, hoisted):
    """KeySwitcher.MulAndRelin / MulAndRelinHoisted on k-party operands, union id sets, level
    taken from ctOut (keyswitch_hoisted.go:46), inputs possibly at a higher level (App. D-7)."""
    mk = pair.mk
    level = pair.maxlevel - drop
    if level < 0 or level + extra > pair.maxlevel:
        pytest.skip("not enough limbs")
    names = sorted(set(ids0) | set(ids1))
    in_limbs = level + 1 + extra
    h0, d0 = pair.ct(ids0, level, in_limbs)
    h1, d1 = pair.ct(ids1, level, in_limbs)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    out = mk.NewCiphertext(pair.params, names, level)
    if hoisted:
        hh0, hh1 = mk.NewHoistedCiphertext(), mk.NewHoistedCiphertext()
        for i in sorted(ids0):
            hh0.Value[i] = mk.NewSwitchingKey(pair.params)
            pair.ksw.Decompose(level, d0, i, hh0.Value[i])
        for i in sorted(ids1):
            hh1.Value[i] = mk.NewSwitchingKey(pair.params)
            pair.ksw.Decompose(level, d1, i, hh1.Value[i])
        pair.ksw.MulAndRelinHoisted(d0, d1, hh0, hh1, rlk_d, out)
    else:
        pair.ksw.MulAndRelin(d0, d1, rlk_d, out)
    ido, ref = oracle_mul_and_relin(pair, level, sorted(ids0), h0, sorted(ids1), h1, rlk_h, u_h, names)
    assert ido == out.ids
    assert (out.download() == ref).all()
    # the fused MulAndRelin + Rescale entry (mkhe_mul_relin_rescale; mkckks/evaluator.go:558-581 always rescales right after): the
    # DivRoundByLastModulus rides on the last ModDown's store where every output slot is written once (up to four products per
    # destination), a pooled temporary + mkhe_rescale otherwise -- either way Rescale(MulAndRelin) of the oracle, bit for bit
    if level >= 1:
        res = mk.NewCiphertext(pair.params, names, level - 1)
        if hoisted:
            pair.ksw.MulAndRelinHoisted(d0, d1, hh0, hh1, rlk_d, res, rescaled=True)
        else:
            pair.ksw.MulAndRelinHoisted(d0, d1, None, None, rlk_d, res, rescaled=True)
        ref_r = np.stack([pair.ks.ringQ.div_round_last_many(ref[s], 1)[0] for s in range(1 + len(names))])
        assert (res.download() == ref_r).all()
        assert (out.download() == ref).all()          # (and the un-rescaled product of the first call is untouched)


def test_mul_and_relin_square(pair):
    """op0 == op1 (mkckks MulRelinNew square case, evaluator.go:419-427; TestCKKS squares)."""
    mk = pair.mk
    level = pair.maxlevel
    names = ["a", "b"]
    h0, d0 = pair.ct(names, level)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.MulAndRelin(d0, d0, rlk_d, out)
    _, ref = oracle_mul_and_relin(pair, level, names, h0, names, h0, rlk_h, u_h, names)
    assert (out.download() == ref).all()


def test_mul_and_relin_errors(pair):
    """the reference panics on level mismatch / missing key (keyswitch_hoisted.go:48-54, keys.go:190-198)."""
    mk = pair.mk
    if pair.maxlevel < 1:
        pytest.skip("single limb")
    _, d0 = pair.ct(["a"], pair.maxlevel - 1)
    rlk_h, rlk_d = pair.rlk_set(["a"])
    pair.params.AddCRS(-1, H.uniform_swk(pair.rng, pair.ks))
    out = mk.NewCiphertext(pair.params, ["a"], pair.maxlevel)
    with pytest.raises(mk.MkheError, match="different levels"):
        pair.ksw.MulAndRelin(d0, d0, rlk_d, out)
    out2 = mk.NewCiphertext(pair.params, ["a"], pair.maxlevel - 1)
    with pytest.raises(mk.MkheError, match="no relinearization key"):
        pair.ksw.MulAndRelin(d0, d0, mk.RelinearizationKeySet(pair.params), out2)


@pytest.mark.parametrize("hoisted", [False, True])
def test_rotate(pair, hoisted):
    """KeySwitcher.Rotate / RotateHoisted incl. the signed coefficient permutation (keyswitch.go:267-296)."""
    mk = pair.mk
    level = pair.maxlevel
    names = ["a", "b"]
    rot = 3
    h, d = pair.ct(names, level)
    crs_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(rot, crs_h)
    rkset = mk.RotationKeySet()
    rk_h = []
    for i in names:
        k = H.uniform_swk(pair.rng, pair.ks)
        rk_h.append(k)
        rkset.AddRotationKey(mk.RotationKey(pair.params, rot, i, k))
    out = mk.NewCiphertext(pair.params, names, level)
    if hoisted:
        hh = mk.NewHoistedCiphertext()
        for i in names:
            hh.Value[i] = mk.NewSwitchingKey(pair.params)
            pair.ksw.Decompose(level, d, i, hh.Value[i])
        pair.ksw.RotateHoisted(d, rot, hh, rkset, out)
    else:
        pair.ksw.Rotate(d, rot, rkset, out)
    galEl = pow(5, rot, 2 * pair.N)
    ref = pair.ks.rotate(level, galEl, [0, 1], h, rk_h, crs_h)
    assert (out.download() == ref).all()


@pytest.fixture(scope="module")
def pn14():
    return Pair(H.PN14QP439, seed=1414)


@pytest.mark.parametrize("parties,level", [(1, 0), (1, 5), (2, 1), (3, 2), (4, 0), (4, 3), (4, 4), (4, 5)])
def test_small_ring_rotate_and_conjugate_every_digit_count(pn14, parties, level):
    """Rotate / Conjugate of a ciphertext without a hoisted form on the N = 2^14 ring, where the engine leaves its own digits after the cross stages and
    ext_fused_lds_kernel finishes the transform, multiplies and (two digit groups: level >= 1) runs the inverse sub-transforms in the same kernel
    (csrc/ntt_kernels.h, ExtFusedArgs): one to six gadget digits (one, two and three rounds of the kernel, the single-group form at level 0), one to four
    parties (merged destinations of one to four members; 4 parties at level 5 is 192 limbs -- above the limit, the unfused path), input above the
    output's level.  Bit for bit against the oracle."""
    pair, mk = pn14, pn14.mk
    names = ["p%d" % i for i in range(parties)]
    rot = 5
    in_level = min(level + 1, pair.maxlevel)
    h, d = pair.ct(names, in_level)
    crs_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(rot, crs_h)
    rkset, rk_h = mk.RotationKeySet(), []
    for i in names:
        k = H.uniform_swk(pair.rng, pair.ks)
        rk_h.append(k)
        rkset.AddRotationKey(mk.RotationKey(pair.params, rot, i, k))
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.Rotate(d, rot, rkset, out)
    ref = pair.ks.rotate(level, pow(5, rot, 2 * pair.N), list(range(parties)), h, rk_h, crs_h)
    assert (out.download() == ref).all()
    if parties in (1, 3, 4) and level in (0, 2, 4):
        ccrs = H.uniform_swk(pair.rng, pair.ks)
        pair.params.AddCRS(-2, ccrs)
        ckset, ck_h = mk.ConjugationKeySet(), []
        for i in names:
            k = H.uniform_swk(pair.rng, pair.ks)
            ck_h.append(k)
            ckset.AddConjugationKey(mk.ConjugationKey(pair.params, i, k))
        out = mk.NewCiphertext(pair.params, names, level)
        pair.ksw.Conjugate(d, ckset, out)
        ref = pair.ks.conjugate(level, 2 * pair.N - 1, list(range(parties)), h, ck_h, ccrs)
        assert (out.download() == ref).all()


@pytest.mark.parametrize("parties,level", [(4, 1), (4, 2), (4, 4), (2, 0), (3, 3)])
def test_small_ring_mul_and_relin_staged_f2(pn14, parties, level):
    """MulAndRelin on the N = 2^14 ring below the top level: the digits of the t_i (step F2) are left after the cross stages and the tail batch finishes
    them inside its product kernel, beside the precomputed step-E products and the NTT-domain tensor term (Context::mr_finish_head / _tail)."""
    pair, mk = pn14, pn14.mk
    names = ["p%d" % i for i in range(parties)]
    h0, d0 = pair.ct(names, level)
    h1, d1 = pair.ct(names, level)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.MulAndRelin(d0, d1, rlk_d, out)
    ido, ref = oracle_mul_and_relin(pair, level, names, h0, names, h1, rlk_h, u_h, names)
    assert ido == out.ids and (out.download() == ref).all()


@pytest.mark.parametrize("nP,gamma,logN", [(1, 1, 11), (3, 3, 11), (3, 1, 11), (1, 1, 14), (3, 3, 14)])
def test_merged_products_other_special_prime_counts(nP, gamma, logN):
    """The merged external products (one inverse NTT of the summed Q limbs, one ModDown tail per destination: csrc/poly_kernels.hip
    moddown_merged_kernel<nP>) with one and three special primes -- alpha = 1 and alpha = 3 -- on a 3-party MulAndRelin with equal id sets
    (destinations with three and two products) and a Rotate (keyswitch_hoisted.go:44-179, keyswitch.go:234-298)."""
    from gpu_common import Pair, oracle_mul_and_relin
    # (logN = 14, alpha = 1: the fused sub-transform + product kernel of the small ring with one and three special primes -- ext_fused_lds_kernel)
    pset = dict(logN=logN, Q=H.PN16_Q[:4], P=H.PN16_P[:nP], scale=float(1 << 45))
    pr = Pair(pset, seed=100 * nP + gamma, gamma=gamma)
    mk, level = pr.mk, pr.maxlevel
    names = ["a", "b", "c"]
    h0, d0 = pr.ct(names, level)
    h1, d1 = pr.ct(names, level)
    rlk_h, rlk_d = pr.rlk_set(names)
    u_h = H.uniform_swk(pr.rng, pr.ks)
    pr.params.AddCRS(-1, u_h)
    out = mk.NewCiphertext(pr.params, names, level)
    pr.ksw.MulAndRelin(d0, d1, rlk_d, out)
    ido, ref = oracle_mul_and_relin(pr, level, names, h0, names, h1, rlk_h, u_h, names)
    assert ido == names and (out.download() == ref).all()
    rot = 5
    crs_h = H.uniform_swk(pr.rng, pr.ks)
    pr.params.AddCRS(rot, crs_h)
    rkset, rk_h = mk.RotationKeySet(), []
    for i in names:
        k = H.uniform_swk(pr.rng, pr.ks)
        rk_h.append(k)
        rkset.AddRotationKey(mk.RotationKey(pr.params, rot, i, k))
    rout = mk.NewCiphertext(pr.params, names, level)
    pr.ksw.Rotate(d0, rot, rkset, rout)
    assert (rout.download() == pr.ks.rotate(level, pow(5, rot, 2 * pr.N), [0, 1, 2], h0, rk_h, crs_h)).all()
    pr.params.close()


def test_conjugate(pair):
    """KeySwitcher.Conjugate (keyswitch.go:302-332)."""
    mk = pair.mk
    level = pair.maxlevel
    names = ["a", "b"]
    h, d = pair.ct(names, level)
    crs_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-2, crs_h)
    ckset = mk.ConjugationKeySet()
    ck_h = []
    for i in names:
        k = H.uniform_swk(pair.rng, pair.ks)
        ck_h.append(k)
        ckset.AddConjugationKey(mk.ConjugationKey(pair.params, i, k))
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.Conjugate(d, ckset, out)
    ref = pair.ks.conjugate(level, 2 * pair.N - 1, [0, 1], h, ck_h, crs_h)
    assert (out.download() == ref).all()


@pytest.mark.parametrize("nb", [1, 2])
def test_rescale(pair, nb):
    """lattigo DivRoundByLastModulusManyLvl on every poly (mkckks/evaluator.go:385-391)."""
    from mkhe_kklss_amd._abi import check, lib
    mk = pair.mk
    level = pair.maxlevel
    if level < nb:
        pytest.skip("not enough limbs")
    h, d = pair.ct(["a", "b"], level)
    out = mk.NewCiphertext(pair.params, ["a", "b"], level - nb)
    check(lib().mkhe_rescale(pair.params.ctx, d.h, nb, out.h))
    got = out.download()
    for s in range(3):
        ref, _ = pair.ks.ringQ.div_round_last_many(h[s], nb)
        assert (got[s] == ref).all()
    assert (d.download() == h).all()      # input untouched (documented deviation from lattigo's in-place +h)


def test_ckks_add_sub(pair):
    """mkckks.Evaluator.AddNew / SubNew (evaluator.go:316-357): union id set, operands at different levels"""
    from mkhe_kklss_amd import mkckks
    if pair.maxlevel < 1:
        pytest.skip("single limb")
    params = pair.params
    ev = mkckks.Evaluator.__new__(mkckks.Evaluator)
    ev.params, ev.ksw = params, pair.ksw
    params._scale = 2.0 ** 40
    h0 = H.uniform_ct(pair.rng, pair.ks, 2, pair.maxlevel + 1)
    h1 = H.uniform_ct(pair.rng, pair.ks, 2, pair.maxlevel)
    c0 = mkckks.NewCiphertext(params, ["a", "b"], pair.maxlevel, 2.0 ** 40).upload(h0)
    c1 = mkckks.NewCiphertext(params, ["b", "c"], pair.maxlevel - 1, 2.0 ** 40).upload(h1)
    add, sub = ev.AddNew(c0, c1), ev.SubNew(c0, c1)
    L = pair.maxlevel
    assert add.Level() == L - 1 and add.ids == ["a", "b", "c"]
    rq = pair.ks.ringQ
    f = lambda fn, x, y: np.stack([getattr(rq, fn)(j, x[j], y[j]) for j in range(L)])
    ga, gs = add.download(), sub.download()
    assert (ga[0] == f("add", h0[0], h1[0])).all() and (gs[0] == f("sub", h0[0], h1[0])).all()
    assert (ga[1] == h0[1][:L]).all() and (gs[1] == h0[1][:L]).all()
    assert (ga[2] == f("add", h0[2], h1[1])).all() and (gs[2] == f("sub", h0[2], h1[1])).all()
    assert (ga[3] == h1[2]).all() and (gs[3] == np.stack([rq.neg(j, h1[2][j]) for j in range(L)])).all()
    # scale matching (evaluator.go:270-292): the operand with the smaller scale is multiplied by floor(ratio) first
    h2 = H.uniform_ct(pair.rng, pair.ks, 1, pair.maxlevel + 1)
    c2 = mkckks.NewCiphertext(params, ["a"], pair.maxlevel, 2.0 ** 45 * 1.5).upload(h2)
    ratio = int((2.0 ** 45 * 1.5) // 2.0 ** 40)                       # 48
    for (x, y), fn in (((c0, c2), "add"), ((c2, c0), "sub")):
        r = (ev.AddNew if fn == "add" else ev.SubNew)(x, y)
        assert r.Scale == 2.0 ** 45 * 1.5 and r.Level() == pair.maxlevel and r.ids == ["a", "b"]
        g = r.download()
        Lf = pair.maxlevel + 1
        sc = lambda p: np.stack([rq.mul_scalar(j, p[j], ratio) for j in range(Lf)])        # the scaled c0
        ff = lambda a_, b_: np.stack([getattr(rq, fn)(j, a_[j], b_[j]) for j in range(Lf)])
        if fn == "add":
            assert (g[0] == ff(sc(h0[0]), h2[0])).all() and (g[1] == ff(sc(h0[1]), h2[1])).all() and (g[2] == sc(h0[2])).all()
        else:
            assert (g[0] == ff(h2[0], sc(h0[0]))).all() and (g[1] == ff(h2[1], sc(h0[1]))).all()
            assert (g[2] == np.stack([rq.neg(j, sc(h0[2])[j]) for j in range(Lf)])).all()


@pytest.mark.parametrize("constant", [3, -2.5, 0.75 + 1.25j, -4 + 0j])
def test_ckks_mult_by_const(pair, constant):
    """mkckks.Evaluator.MultByConst (evaluator.go:117-199): integer, fractional, complex constants"""
    from mkhe_kklss_amd import mkckks
    params = pair.params
    ev = mkckks.Evaluator.__new__(mkckks.Evaluator)
    ev.params, ev.ksw = params, pair.ksw
    L = pair.maxlevel + 1
    h = H.uniform_ct(pair.rng, pair.ks, 2, L)
    c0 = mkckks.NewCiphertext(params, ["a", "b"], pair.maxlevel, 2.0 ** 40).upload(h)
    out = mkckks.NewCiphertext(params, ["a", "b"], pair.maxlevel, 1.0)
    ev.MultByConst(c0, constant, out)
    got = out.download()
    cre, cim = (constant.real, constant.imag) if isinstance(constant, complex) else (float(constant), 0.0)
    frac = any(c != 0 and c != int(c) for c in (cre, cim))
    scale = float(pair.Q[pair.maxlevel]) if frac else 1.0
    assert out.Scale == 2.0 ** 40 * scale
    N = pair.N
    for l, q in enumerate(pair.Q):
        sre = mkckks.scaleUpExact(cre, scale, q) if cre != 0 else 0
        sim = (mkckks.scaleUpExact(cim, scale, q) * pow(pair.ks.ringQ.psi(l), N // 2, q)) % q if cim != 0 else 0
        c1, c2 = (sre + sim) % q, (sre - sim) % q
        for s in range(3):
            x = [int(v) for v in h[s][l]]
            assert [int(v) for v in got[s][l][: N // 2]] == [(v * c1) % q for v in x[: N // 2]]
            assert [int(v) for v in got[s][l][N // 2:]] == [(v * c2) % q for v in x[N // 2:]]


def test_ckks_mul_ptxt_and_drop_level(pair):
    """MulPtxtNew (evaluator.go:465-481, product in the NTT domain + Rescale) and DropLevelNew (:96-114)"""
    from mkhe_kklss_amd import mkckks
    if pair.maxlevel < 1:
        pytest.skip("single limb")
    params, ks = pair.params, pair.ks
    ev = mkckks.Evaluator.__new__(mkckks.Evaluator)
    ev.params, ev.ksw = params, pair.ksw
    scale = float(pair.Q[pair.maxlevel])
    params.Scale = lambda: scale          # Pair holds mkrlwe.Parameters; mkckks.Parameters adds the default scale
    L = pair.maxlevel + 1
    h = H.uniform_ct(pair.rng, ks, 1, L)
    pt = H.uniform_poly(pair.rng, pair.Q, pair.N)
    ct = mkckks.NewCiphertext(params, ["a"], pair.maxlevel, scale).upload(h)
    res = ev.MulPtxtNew(ct, pt, scale)
    rq = ks.ringQ
    prod = np.stack([np.stack([rq.intt(l, rq.mul(l, rq.ntt(l, h[s][l]), rq.mform(l, rq.ntt(l, pt[l])))) for l in range(L)]) for s in range(2)])
    nb, sc = ks.ckks_nb_rescales(pair.maxlevel, scale * scale, scale)
    assert nb >= 1 and res.Level() == pair.maxlevel - nb and res.Scale == sc
    ref = np.stack([rq.div_round_last_many(prod[s], nb)[0] for s in range(2)])
    assert (res.download() == ref).all()
    d = ev.DropLevelNew(ct, 1)
    assert d.Level() == pair.maxlevel - 1 and (d.download() == h[:, : L - 1]).all()


def test_overlap_switch_does_not_change_results(pair):
    """mkhe_set_overlap(0): every kernel on the main stream (the configuration of the per-kernel timings)"""
    from mkhe_kklss_amd._abi import check, lib
    mk = pair.mk
    level = pair.maxlevel
    names = ["a", "b", "c"]
    h0, d0 = pair.ct(names, level)
    h1, d1 = pair.ct(names, level)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    _, ref = oracle_mul_and_relin(pair, level, names, h0, names, h1, rlk_h, u_h, names)
    try:
        for on in (0, 1, 0):
            check(lib().mkhe_set_overlap(pair.params.ctx, on))
            out = mk.NewCiphertext(pair.params, names, level)
            pair.ksw.MulAndRelin(d0, d1, rlk_d, out)
            assert (out.download() == ref).all()
    finally:
        check(lib().mkhe_set_overlap(pair.params.ctx, 1))


def test_ntt_word_boundary_patterns(pair):
    """inputs whose 32-bit words sit on the boundaries of the signed-digit split used by the NTT kernels
    (low word 0x80000000 / 0x7fffffff / 0xffffffff / 0, values up to 4q - 1), forward and inverse"""
    mk, ks = pair.mk, pair.ks
    nq = len(pair.Q)
    N = pair.N
    pats = np.array([0x80000000, 0x7fffffff, 0xffffffff, 0, 1, 0x100000000, 0x17fffffff, 0x180000000], dtype=np.uint64)
    a = np.empty((1, nq, N), dtype=np.uint64)
    for j, q in enumerate(pair.Q):
        hi = pair.rng.integers(0, (4 * q) >> 32, N, dtype=np.uint64) << np.uint64(32)
        v = hi | pats[pair.rng.integers(0, len(pats), N)]
        v = np.where(v >= np.uint64(4 * q), v % np.uint64(q), v)
        v[:4] = [4 * q - 1, q - 1, q, 2 * q + 0x80000000]
        a[0, j] = v
    src = mk.DeviceLimbs(pair.params, 1, nq).upload(a)
    dst = mk.DeviceLimbs(pair.params, 1, nq)
    mk.ntt(pair.params, src, dst)
    got = dst.download()
    for j in range(nq):
        assert (got[0][j] == ks.ringQ.ntt(j, a[0][j])).all()
    # inverse on canonical inputs with the same word patterns
    b = np.stack([a[0, j] % np.uint64(q) for j, q in enumerate(pair.Q)])[None]
    src.upload(b)
    mk.ntt(pair.params, src, dst, inverse=True)
    got = dst.download()
    for j in range(nq):
        assert (got[0][j] == ks.ringQ.intt(j, b[0][j])).all()


def test_limb_pointer_entry_points(pair):
    """the upload / download entry points the cgo shim uses (ring.Poly.Coeffs is [][]uint64: one pointer per limb):
    mkhe_swk_upload_limbs, mkhe_ct_upload_poly_limbs, mkhe_ct_download_poly_limbs"""
    import ctypes as C
    from mkhe_kklss_amd._abi import check, lib
    mk = pair.mk
    N, nq, m = pair.N, len(pair.Q), len(pair.Q) + len(pair.P)
    host = H.uniform_swk(pair.rng, pair.ks)
    limbs = [np.ascontiguousarray(host[i, j]) for i in range(host.shape[0]) for j in range(m)]       # separate heap objects
    ptrs = (C.c_void_p * len(limbs))(*[l.ctypes.data for l in limbs])
    swk = mk.NewSwitchingKey(pair.params)
    check(lib().mkhe_swk_upload_limbs(pair.params.ctx, swk.h, ptrs, host.shape[0]))
    assert (swk.download() == host).all()
    ct = mk.NewCiphertext(pair.params, ["a", "b"], pair.maxlevel)
    polys = H.uniform_ct(pair.rng, pair.ks, 2, nq)
    for slot in range(3):
        pl = [np.ascontiguousarray(polys[slot, l]) for l in range(nq)]
        pp = (C.c_void_p * nq)(*[l.ctypes.data for l in pl])
        check(lib().mkhe_ct_upload_poly_limbs(pair.params.ctx, ct.h, slot, pp))
    assert (ct.download() == polys).all()
    out = [np.zeros(N, dtype=np.uint64) for _ in range(nq)]
    po = (C.c_void_p * nq)(*[l.ctypes.data for l in out])
    check(lib().mkhe_ct_download_poly_limbs(pair.params.ctx, ct.h, 2, po))
    assert (np.stack(out) == polys[2]).all()


def test_ckks_rescale_new_and_errors(pair):
    """mkckks.Evaluator.RescaleNew (evaluator.go:359-414): scale loop on the host, division on the device, the error cases"""
    from mkhe_kklss_amd import mkckks
    if pair.maxlevel < 2:
        pytest.skip("needs three limbs")
    params = pair.params
    ev = mkckks.Evaluator.__new__(mkckks.Evaluator)
    ev.params, ev.ksw = params, pair.ksw
    L = pair.maxlevel + 1
    h = H.uniform_ct(pair.rng, pair.ks, 1, L)
    scale = float(pair.Q[pair.maxlevel]) * float(pair.Q[pair.maxlevel - 1]) * 2.0 ** 30
    ct = mkckks.NewCiphertext(params, ["a"], pair.maxlevel, scale).upload(h)
    res = ev.RescaleNew(ct, 2.0 ** 30)
    nb, sc = pair.ks.ckks_nb_rescales(pair.maxlevel, scale, 2.0 ** 30)
    assert nb == 2 and res.Level() == pair.maxlevel - 2 and res.Scale == sc
    ref = np.stack([pair.ks.ringQ.div_round_last_many(h[s], nb)[0] for s in range(2)])
    assert (res.download() == ref).all()
    with pytest.raises(pair.mk.MkheError, match="minScale is 0"):
        ev.RescaleNew(ct, 0)
    low = mkckks.NewCiphertext(params, ["a"], 0, scale)
    with pytest.raises(pair.mk.MkheError, match="already at level 0"):
        ev.RescaleNew(low, 2.0 ** 30)
    ct.Scale = 0.0
    with pytest.raises(pair.mk.MkheError, match="scale is 0"):
        ev.RescaleNew(ct, 2.0 ** 30)


def test_null_arguments_are_errors_not_crashes():
    """the C ABI reports null contexts / handles through its error channel (the reference panics; it never segfaults)"""
    import ctypes as C
    from mkhe_kklss_amd._abi import lib
    L = lib()
    assert L.mkhe_ctx_sync(None) != 0 and b"null context" in L.mkhe_last_error()
    assert L.mkhe_ctx_n(None) == 0 and L.mkhe_ct_limbs(None) == 0
    h = C.c_void_p()
    assert L.mkhe_swk_create(None, C.byref(h)) != 0
    pset = H.small_ckks(10, 2)
    from mkhe_kklss_amd import mkrlwe
    params = mkrlwe.Parameters(pset["logN"], pset["Q"], pset["P"])
    assert L.mkhe_swk_upload(params.ctx, None, None) != 0 and b"null argument" in L.mkhe_last_error()
    assert L.mkhe_ct_download(params.ctx, None, None) != 0
    assert L.mkhe_rescale(params.ctx, None, 1, None) != 0
    assert L.mkhe_crs_expand(params.ctx, 1, 0, None) != 0


def test_split_finish_matches_finish_and_checks_its_order():
    """mkhe_mr_finish == mkhe_mr_finish_head; mkhe_mr_finish_tail (the party-sharded path runs the head while x is still being
    all-reduced): same bits as the one-call form and as the oracle; a tail without its head is an error, not a crash"""
    from mkhe_kklss_amd import mkrlwe
    from mkhe_kklss_amd._abi import check, handle_array, lib
    pset = H.small_ckks(12, 3)
    pair = Pair(pset, seed=77)
    names = ["a", "b", "c"]
    level = pair.maxlevel
    h0, d0 = pair.ct(names, level, level + 1)
    h1, d1 = pair.ct(names, level, level + 1)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    _, ref = oracle_mul_and_relin(pair, level, names, h0, names, h1, rlk_h, u_h, names)
    L, ctx = lib(), pair.params.ctx
    keys = [rlk_d.Value[n] for n in names]
    b1 = handle_array([k.Value[0].h for k in keys]); dd = handle_array([k.Value[1].h for k in keys]); v0 = handle_array([k.Value[2].h for k in keys])
    outs = []
    for split in (False, True):
        out = mkrlwe.NewCiphertext(pair.params, names, level)
        x, y = mkrlwe.NewSwitchingKey(pair.params), mkrlwe.NewSwitchingKey(pair.params)
        check(L.mkhe_mr_partial(ctx, d0.h, d1.h, None, None, b1, dd, 1, out.h, x.h, y.h))
        check(L.mkhe_swk_fold(ctx, x.h, level, 1)); check(L.mkhe_swk_fold(ctx, y.h, level, 1))
        if split:
            assert L.mkhe_mr_finish_tail(ctx, d0.h, d1.h, x.h, v0, pair.params.CRS[-1].h, out.h) != 0
            assert b"without mr_finish_head" in L.mkhe_last_error()
            check(L.mkhe_mr_partial(ctx, d0.h, d1.h, None, None, b1, dd, 1, out.h, x.h, y.h))      # (the error reset the plan)
            check(L.mkhe_swk_fold(ctx, x.h, level, 1)); check(L.mkhe_swk_fold(ctx, y.h, level, 1))
            check(L.mkhe_mr_finish_head(ctx, d0.h, d1.h, y.h, out.h))
            check(L.mkhe_mr_finish_tail(ctx, d0.h, d1.h, x.h, v0, pair.params.CRS[-1].h, out.h))
        else:
            check(L.mkhe_mr_finish(ctx, d0.h, d1.h, x.h, y.h, v0, pair.params.CRS[-1].h, out.h))
        outs.append(out.download())
    assert (outs[0] == ref).all() and (outs[1] == ref).all()

"""Headline configurations under -m gpu: BASELINE.json configs[1] and configs[2] at FULL size, device vs oracle bit for bit.

* configs[1]: mkckks PN15QP880 (N = 2^15, 14 Q + 2 P primes), 4 parties: MulRelinNew = hoisting of both operands +
  MulAndRelinHoisted + Rescale -- the timed region of mkckks/mkckks_benchmark_test.go:57-84;
* the Decompose-fused forward NTT of that ring at every launch size class of csrc/ntt_kernels.hip (1792, 896, 672, 448
  and 224 limbs, and a lower level), so that the mixed-class persistent launch, the per-class kernels and the
  split / low-latency paths are each compared with the oracle (mkrlwe/keyswitch.go:49-73);
* RotateHoisted with 4 parties at PN15QP880 (mkrlwe/keyswitch_hoisted.go:183-247);
* configs[2]: mkbfv PN15QP880 (14 Q + 14 QMul + 2 P primes), MulRelinNew with 2 and 4 parties
  (mkbfv/mkbfv_bench_test.go:10-64);
* configs[3] ring: PN16QP1761 with 8 parties at the maximum level, one hoisted Rotate and MulAndRelin[Hoisted] (mkrlwe/mkrlwe_test.go:22-35).
"""
import numpy as np
import pytest

import harness as H
from gpu_common import Pair

pytestmark = pytest.mark.gpu


def _swk(pset, rng, beta=None):
    """uniform residues, switching-key shaped uint64[beta][nQ+nP][N]"""
    Q, P, N = pset["Q"], pset["P"], 1 << pset["logN"]
    beta = len(Q) if beta is None else beta
    out = np.empty((beta, len(Q) + len(P), N), dtype=np.uint64)
    for j, q in enumerate(Q + P):
        out[:, j] = rng.integers(0, q, (beta, N), dtype=np.uint64)
    return out


def _ct(pset, rng, n, limbs):
    N = 1 << pset["logN"]
    out = np.empty((1 + n, limbs, N), dtype=np.uint64)
    for l in range(limbs):
        out[:, l] = rng.integers(0, pset["Q"][l], (1 + n, N), dtype=np.uint64)
    return out


@pytest.fixture(scope="module")
def pn15():
    from oracle import oracle as O
    from mkhe_kklss_amd import mkckks
    p = H.PN15QP880
    ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
    params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
    return dict(pset=p, ks=ks, params=params, mk=mkckks, rng=np.random.default_rng(0x15880))


def test_pn15_four_party_mulrelin_new(pn15):
    """BASELINE.json configs[1]: the benchmark's timed region (hoist + MulAndRelinHoisted + Rescale), k = 4."""
    from mkhe_kklss_amd import mkrlwe
    p, ks, params, mk, rng = (pn15[k] for k in ("pset", "ks", "params", "mk", "rng"))
    k = 4
    names = ["user%d" % i for i in range(k)]
    level = len(p["Q"]) - 1
    h0, h1 = _ct(p, rng, k, level + 1), _ct(p, rng, k, level + 1)
    rlk_h, rlk = {}, mkrlwe.RelinearizationKeySet(params)
    for i, n in enumerate(names):
        rlk_h[i] = tuple(_swk(p, rng) for _ in range(3))
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *rlk_h[i]))
    u = _swk(p, rng)
    params.AddCRS(-1, u)
    ct0 = mk.NewCiphertext(params, names, level, p["scale"]).upload(h0)
    ct1 = mk.NewCiphertext(params, names, level, p["scale"]).upload(h1)
    ev = mk.NewEvaluator(params)
    res = ev.MulRelinNew(ct0, ct1, rlk)
    ids = list(range(k))
    _, ref = ks.mul_and_relin(level, ids, h0, ids, h1, rlk_h, u)
    nb, _ = ks.ckks_nb_rescales(level, p["scale"] * p["scale"], p["scale"])
    assert nb == 1 and res.Level() == level - 1
    ref = np.stack([ks.ringQ.div_round_last_many(ref[s], nb)[0] for s in range(1 + k)])
    got = res.download()
    assert got.shape == ref.shape and (got == ref).all()
    # the hoisted entry point on precomputed hoisted forms gives the same ciphertext
    res2 = ev.MulRelinHoistedNew(ct0, ct1, ev.HoistedForm(ct0), ev.HoistedForm(ct1), rlk)
    assert (res2.download() == ref).all()
    # and a second evaluation through the same (now warm) pools is identical: no stale scratch
    assert (ev.MulRelinNew(ct0, ct1, rlk).download() == ref).all()


def _mulrelin_case(fix, k0, k1=None, level=None, hoisted=False, ids1=None):
    """One MulRelinNew (hoist + MulAndRelinHoisted + Rescale) of a k0-party by k1-party ciphertext pair at `level` against the oracle, bit for bit."""
    from mkhe_kklss_amd import mkrlwe
    p, ks, params, mk, rng = (fix[k] for k in ("pset", "ks", "params", "mk", "rng"))
    k1 = k0 if k1 is None else k1
    level = len(p["Q"]) - 1 if level is None else level
    ids0 = list(range(k0))
    ids1 = list(range(k1)) if ids1 is None else ids1
    allids = sorted(set(ids0) | set(ids1))
    names = {i: "user%d" % i for i in allids}
    h0, h1 = _ct(p, rng, k0, level + 1), _ct(p, rng, len(ids1), level + 1)
    rlk_h, rlk = {}, mkrlwe.RelinearizationKeySet(params)
    for i in allids:
        rlk_h[i] = tuple(_swk(p, rng) for _ in range(3))
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, names[i], *rlk_h[i]))
    u = _swk(p, rng)
    params.AddCRS(-1, u)
    ct0 = mk.NewCiphertext(params, [names[i] for i in ids0], level, p["scale"]).upload(h0)
    ct1 = mk.NewCiphertext(params, [names[i] for i in ids1], level, p["scale"]).upload(h1)
    ev = mk.NewEvaluator(params)
    res = ev.MulRelinHoistedNew(ct0, ct1, ev.HoistedForm(ct0), ev.HoistedForm(ct1), rlk) if hoisted else ev.MulRelinNew(ct0, ct1, rlk)
    oids, ref = ks.mul_and_relin(level, ids0, h0, ids1, h1, rlk_h, u)
    nb, _ = ks.ckks_nb_rescales(level, p["scale"] * p["scale"], p["scale"])
    assert res.Level() == level - nb
    ref = np.stack([ks.ringQ.div_round_last_many(ref[s], nb)[0] for s in range(ref.shape[0])])
    got = res.download()
    assert got.shape == ref.shape and (got == ref).all()
    assert (ev.MulRelinNew(ct0, ct1, rlk).download() == ref).all()      # warm pools, same bits


def test_pn15_two_party_mulrelin_new(pn15):
    """BASELINE.json configs[0] at its own size: the reference benchmark's 2-party MulRelin (mkckks/mkckks_benchmark_test.go:40-44,57-84) on PN15QP880."""
    _mulrelin_case(pn15, 2)


# Step F2 inside the Decompose NTT of the t_i (round 6, csrc/ntt16_f2_kernels.hip): the launches whose schedule differs -- four parties (two runs per
# group), five (a workgroup's run crosses groups: parts of unequal length), eight (one group per workgroup: no parts), lower levels (fewer digits and
# limb slots, other cuts), op1 with fewer parties than op0 (step E inside the F1 kernel still), hoisted forms supplied by the caller
# -- and the shapes whose passes do not fill the chip twice (one to three parties, low levels): the grid is planned (f2_plan_schedule: fewer workgroups
# than CUs, up to eight parts per product, groups cut unevenly)
@pytest.mark.parametrize("k0,k1,level,hoisted", [(4, 4, 9, False), (4, 4, 1, False), (4, 2, 13, False), (5, 5, 13, False), (8, 8, 13, False), (6, 6, 5, False), (4, 4, 13, True), (7, 7, 2, False),
                                                 (1, 1, 13, False), (2, 2, 13, False), (3, 3, 13, False), (1, 1, 0, False), (2, 2, 6, False), (3, 1, 4, False), (4, 4, 6, False),
                                                 (3, 3, 9, True), (1, 3, 11, False), (2, 1, 3, False)])
def test_pn15_fused_f2_shapes(pn15, k0, k1, level, hoisted):
    _mulrelin_case(pn15, k0, k1, level, hoisted)


def test_pn15_batch_runs_in_flight(pn15):
    """mkhe_mul_relin_batch at N = 2^15 (round 6): B evaluations through the single-operation path on the context and two internal ones, round robin -- five
    inputs (lanes 0 1 2 0 1), four parties, top level, twice (the internal contexts are reused), then a dependent batch on the outputs; every output equal to
    the single evaluation's (which the tests above hold against the oracle)."""
    from mkhe_kklss_amd import mkrlwe
    p, params, mk, rng = (pn15[k] for k in ("pset", "params", "mk", "rng"))
    k, B, level = 4, 5, len(p["Q"]) - 1
    names = ["user%d" % i for i in range(k)]
    rlk = mkrlwe.RelinearizationKeySet(params)
    for n in names:
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *(_swk(p, rng) for _ in range(3))))
    params.AddCRS(-1, _swk(p, rng))
    new = lambda: mk.NewCiphertext(params, names, level, p["scale"]).upload(_ct(p, rng, k, level + 1))
    b0, b1 = mk.BatchCiphertext([new() for _ in range(B)]), mk.BatchCiphertext([new() for _ in range(B)])
    ev, bev = mk.NewEvaluator(params), mk.BatchEvaluator(params, B)
    ref = [ev.MulRelinNew(b0.cts[i], b1.cts[i], rlk).download() for i in range(B)]
    for _ in range(2):
        out = bev.MulRelinNew(b0, b1, rlk)
        for i in range(B):
            assert (out.cts[i].download() == ref[i]).all(), i
    hh = bev.HoistedForm(out)
    sq = bev.MulRelinHoistedNew(out, out, hh, hh, rlk)              # consumes what the lanes wrote: ordered behind them on the caller's stream
    for i in range(B):
        assert (sq.cts[i].download() == ev.MulRelinNew(out.cts[i], out.cts[i], rlk).download()).all(), i


@pytest.mark.parametrize("parties,drop", [(8, 0), (4, 0), (3, 0), (2, 0), (1, 0), (3, 2), (5, 1)])
def test_pn15_hoisted_form_launch_classes(pn15, parties, drop):
    """HoistedForm of n components = ONE Decompose launch of n * beta * (level + 1 + nP) limbs:
    1792 / 896 / 672 (mixed-class persistent kernel), 448 / 224 (one kernel per modulus class), and lower levels
    (3 * 12 * 14 = 504 limbs just under, 5 * 13 * 15 = 975 over the threshold)."""
    p, ks, params, mk, rng = (pn15[k] for k in ("pset", "ks", "params", "mk", "rng"))
    level = len(p["Q"]) - 1 - drop
    names = ["p%d" % i for i in range(parties)]
    h = _ct(p, rng, parties, level + 1)
    ct = mk.NewCiphertext(params, names, level, p["scale"]).upload(h)
    hoisted = mk.NewEvaluator(params).HoistedForm(ct)
    beta = ks.beta(level)
    act = list(range(level + 1)) + [len(p["Q"]) + j for j in range(len(p["P"]))]
    for i, n in enumerate(names):
        ref = ks.decompose(level, h[1 + i])
        got = hoisted.Value[n].download()
        assert (got[:beta][:, act] == ref[:beta][:, act]).all(), "party %d" % i


def test_pn15_pinned_forward_kernel_choice(pn15):
    """mkhe_ctx_set_ntt_choice pins the forward kernel of N = 2^15 launches where two apply (a bench line can be repeated with
    the kernels it records): 0 = two-pass, 1 = single-pass, for every shape of the context; same bits, and mkhe_ntt_choice reports the pin."""
    from mkhe_kklss_amd._abi import check, lib
    p, ks, params, mk, rng = (pn15[k] for k in ("pset", "ks", "params", "mk", "rng"))
    level, parties = len(p["Q"]) - 1, 3                       # 3 * 14 * 16 = 672 limbs: both kernels apply
    names = ["p%d" % i for i in range(parties)]
    h = _ct(p, rng, parties, level + 1)
    ct = mk.NewCiphertext(params, names, level, p["scale"]).upload(h)
    ref = ks.decompose(level, h[1])
    try:
        for choice in (0, 1, 0):
            check(lib().mkhe_ctx_set_ntt_choice(params.ctx, 0, 1, choice))
            got = mk.NewEvaluator(params).HoistedForm(ct).Value["p0"].download()
            assert (got == ref).all(), choice
            assert lib().mkhe_ntt_choice(params.ctx, 672, 1) == choice
        check(lib().mkhe_ctx_set_ntt_choice(params.ctx, 672, 1, 1))      # one shape
        assert lib().mkhe_ntt_choice(params.ctx, 672, 1) == 1
        assert lib().mkhe_ctx_set_ntt_choice(params.ctx, 0, 1, 7) != 0    # bad value: an error, not a crash
    finally:
        check(lib().mkhe_ctx_set_ntt_choice(params.ctx, 0, 1, -1))        # back to measuring for the tests that follow


def test_n15_modulus_outside_the_h16_ranges():
    """A small-class modulus (31q < 2^62) that the one-round product of csrc/ntt16_kernels.hip does not cover (48q >= 2^62: its
    never-reduced values grow by q per stage): such a context keeps the round-1 forward kernels for its large Decompose
    launches (Context::h16_gap_), and those still agree with the oracle."""
    from oracle import oracle as O
    from mkhe_kklss_amd import mkckks
    base = H.PN15QP880
    pset = dict(logN=15, Q=[0x1fffffffffc0001] + base["Q"][1:9], P=base["P"], scale=base["scale"])
    assert (1 << 62) // 48 <= pset["Q"][0] < (1 << 62) // 31
    ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=0)
    rng = np.random.default_rng(57)
    level = len(pset["Q"]) - 1
    names = ["a", "b"]
    h = _ct(pset, rng, 2, level + 1)
    ct = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(h)
    hoisted = mkckks.NewEvaluator(params).HoistedForm(ct)          # 2 * 9 * 11 = 198 limbs: an H16 launch in any other context
    for i, n in enumerate(names):
        assert (hoisted.Value[n].download() == ks.decompose(level, h[1 + i])).all(), "party %d" % i
    params.close()


def test_pn15_rotate_hoisted_four_parties(pn15):
    from mkhe_kklss_amd import mkrlwe
    p, ks, params, mk, rng = (pn15[k] for k in ("pset", "ks", "params", "mk", "rng"))
    k, rot = 4, 3
    names = ["r%d" % i for i in range(k)]
    level = len(p["Q"]) - 1
    h = _ct(p, rng, k, level + 1)
    ct = mk.NewCiphertext(params, names, level, p["scale"]).upload(h)
    crs = _swk(p, rng)
    params.AddCRS(rot, crs)
    keys = [_swk(p, rng) for _ in range(k)]
    rks = mkrlwe.RotationKeySet()
    for n, key in zip(names, keys):
        rks.AddRotationKey(mkrlwe.RotationKey(params, rot, n, key))
    ev = mk.NewEvaluator(params)
    ref = ks.rotate(level, pow(5, rot, 2 << p["logN"]), list(range(k)), h, keys, crs)
    assert (ev.RotateHoistedNew(ct, rot, ev.HoistedForm(ct), rks).download() == ref).all()
    assert (ev.RotateNew(ct, rot, rks).download() == ref).all()


@pytest.mark.parametrize("parties", [2, 4])
def test_bfv_pn15_mulrelin_new(parties):
    """BASELINE.json configs[2]: mkbfv MulRelinNew (ModUpQtoR + Rescale + DecomposeBFV + MulAndRelinBFVHoisted + Quantize)."""
    import harness_bfv as HB
    from mkhe_kklss_amd import mkbfv
    pset = HB.BFV_PN15QP880
    names = ["user%d" % i for i in range(parties)]
    data = HB.uniform_bfv_inputs(pset, parties, 0xBF15 + parties)
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"], device=0)
    ct0 = mkbfv.NewCiphertext(params, names).upload(data["op0"])
    ct1 = mkbfv.NewCiphertext(params, names).upload(data["op1"])
    rlk = mkbfv.NewRelinearizationKeyKeySet(params)
    for i, n in enumerate(names):
        rlk.AddRelinearizationKey(mkbfv.RelinearizationKey(params, n, *data["rlk"][i]))
    params.AddCRS(-1, data["u"])
    res = mkbfv.NewEvaluator(params).MulRelinNew(ct0, ct1, rlk)
    ids = list(range(parties))
    _, ref = HB.make_bfv(pset).mul_relin_new(ids, data["op0"], ids, data["op1"], data["rlk"], data["u"])
    got = res.download()
    assert got.shape == ref.shape and (got == ref).all()
    if parties == 2:      # the non-hoisted twin at full size (mkbfv/keyswitch.go:115-251; its own device path)
        assert (mkbfv.NewEvaluator(params).mulRelin(ct0, ct1, rlk).download() == ref).all()
    params.close()


def test_pn16_rotate_hoisted_eight_parties():
    """BASELINE.json configs[3] ring at its party count: PN16QP1761 (N = 2^16, alpha = 2, beta = 17), 8 parties."""
    pair = Pair(H.PN16QP1761, seed=168)
    mk, p, rng = pair.mk, H.PN16QP1761, pair.rng
    k, rot = 8, 5
    names = ["u%d" % i for i in range(k)]
    level = pair.maxlevel
    h = _ct(p, rng, k, level + 1)
    ct = mk.NewCiphertext(pair.params, names, level).upload(h)
    beta = pair.ks.beta_max
    crs = _swk(p, rng, beta)
    pair.params.AddCRS(rot, crs)
    keys = [_swk(p, rng, beta) for _ in range(k)]
    rks = mk.RotationKeySet()
    for n, key in zip(names, keys):
        rks.AddRotationKey(mk.RotationKey(pair.params, rot, n, key))
    hh = mk.NewHoistedCiphertext()
    for n in names:
        hh.Value[n] = mk.NewSwitchingKey(pair.params)
        pair.ksw.Decompose(level, ct, n, hh.Value[n])
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.RotateHoisted(ct, rot, hh, rks, out)
    ref = pair.ks.rotate(level, pow(5, rot, 2 * pair.N), list(range(k)), h, keys, crs)
    assert (out.download() == ref).all()


def test_pn16_mul_and_relin_eight_parties():
    """BASELINE.json configs[3], the MulRelin half: PN16QP1761 (N = 2^16, 34 + 4 primes, alpha = 2, beta = 17) at the maximum level with
    8 parties, MulAndRelin and MulAndRelinHoisted against the oracle (mkrlwe/keyswitch.go:122-230, keyswitch_hoisted.go:44-179).  Eight
    parties drive what two (tests/test_gpu_fullsize.py) do not: two virtual items per output slot in the merged external products, the
    x inner product as its own launch (more than four parties) and decomp_spread's fused first stage over 16 operand components.
    The 24 key polynomials (8.1 GB) are three random arrays rotated by a party-specific offset along the coefficient axis -- distinct
    data per party and per key at memcpy speed; the oracle runs its limb loops on 8 threads (same integers, tests/test_oracle_vs_model)."""
    from oracle import oracle as O
    pair = Pair(H.PN16QP1761, seed=1688)
    mk, p, rng = pair.mk, H.PN16QP1761, pair.rng
    k = 8
    names = ["u%d" % i for i in range(k)]
    level = pair.maxlevel
    beta = pair.ks.beta_max
    assert pair.params.Alpha() == 2 and pair.params.Beta(level) == 17 == beta
    h0, h1 = _ct(p, rng, k, level + 1), _ct(p, rng, k, level + 1)
    base = [_swk(p, rng, beta) for _ in range(3)]
    u = _swk(p, rng, beta)
    pair.params.AddCRS(-1, u)
    rlk_h, rlk = {}, mk.RelinearizationKeySet(pair.params)
    for i, n in enumerate(names):
        rlk_h[i] = tuple(np.ascontiguousarray(np.roll(base[j], 997 * i + 131 * j + 1, axis=2)) for j in range(3))
        rlk.AddRelinearizationKey(mk.RelinearizationKey(pair.params, n, *rlk_h[i]))
    ct0 = mk.NewCiphertext(pair.params, names, level).upload(h0)
    ct1 = mk.NewCiphertext(pair.params, names, level).upload(h1)
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.MulAndRelin(ct0, ct1, rlk, out)
    got = out.download()
    O.set_threads(8)
    try:
        ids = list(range(k))
        _, ref = pair.ks.mul_and_relin(level, ids, h0, ids, h1, rlk_h, u)
    finally:
        O.set_threads(1)
    assert got.shape == ref.shape and (got == ref).all()
    for l, q in enumerate(pair.Q):
        assert (got[:, l] < q).all()
    # the hoisted entry point on hoisted forms computed beforehand (keyswitch_hoisted.go:44-179)
    hh = []
    for ct in (ct0, ct1):
        h = mk.NewHoistedCiphertext()
        for n in names:
            h.Value[n] = mk.NewSwitchingKey(pair.params)
            pair.ksw.Decompose(level, ct, n, h.Value[n])
        hh.append(h)
    out2 = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.MulAndRelinHoisted(ct0, ct1, hh[0], hh[1], rlk, out2)
    assert (out2.download() == ref).all()
    pair.params.close()


# ---------------------------------------------------------------- PN14QP439: the first set of the reference's benchmark (mkckks_benchmark_test.go:13)
@pytest.fixture(scope="module")
def pn14():
    from oracle import oracle as O
    from mkhe_kklss_amd import mkckks
    p = H.PN14QP439
    ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
    params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
    return dict(pset=p, ks=ks, params=params, mk=mkckks, rng=np.random.default_rng(0x14439))


@pytest.mark.parametrize("parties,drop", [(8, 0), (4, 0), (3, 0), (6, 2), (2, 0)])
def test_pn14_hoisted_form_launch_classes(pn14, parties, drop):
    """N = 2^14: Decompose launches of >= 128 limbs run the one-pass instantiation of the H16 kernel (csrc/ntt16_kernels.hip
    ntt14_fwd_kernel<true>: 8 x 6 x 8 = 384, 192, 144 limbs, a lower level 6 x 4 x 6 = 144), smaller ones (2 x 6 x 8 = 96) the round-1
    kernels; the 59-bit head prime and the two 60-bit special primes take the balanced path with partial reductions, the 52-bit primes the
    U class.  Canonical digits against the oracle (mkrlwe/keyswitch.go:49-73)."""
    p, ks, params, mk, rng = (pn14[k] for k in ("pset", "ks", "params", "mk", "rng"))
    level = len(p["Q"]) - 1 - drop
    names = ["p%d" % i for i in range(parties)]
    h = _ct(p, rng, parties, level + 1)
    ct = mk.NewCiphertext(params, names, level, p["scale"]).upload(h)
    hoisted = mk.NewEvaluator(params).HoistedForm(ct)
    beta = ks.beta(level)
    act = list(range(level + 1)) + [len(p["Q"]) + j for j in range(len(p["P"]))]
    for i, n in enumerate(names):
        ref = ks.decompose(level, h[1 + i])
        got = hoisted.Value[n].download()
        assert (got[:beta][:, act] == ref[:beta][:, act]).all(), "party %d" % i


def test_pn14_four_party_mulrelin_new(pn14):
    """the benchmark's timed region at PN14QP439 with 4 parties (engine-internal digits: un-normalised U-class outputs)"""
    from mkhe_kklss_amd import mkrlwe
    p, ks, params, mk, rng = (pn14[k] for k in ("pset", "ks", "params", "mk", "rng"))
    k = 4
    names = ["user%d" % i for i in range(k)]
    level = len(p["Q"]) - 1
    h0, h1 = _ct(p, rng, k, level + 1), _ct(p, rng, k, level + 1)
    rlk_h, rlk = {}, mkrlwe.RelinearizationKeySet(params)
    for i, n in enumerate(names):
        rlk_h[i] = tuple(_swk(p, rng) for _ in range(3))
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *rlk_h[i]))
    u = _swk(p, rng)
    params.AddCRS(-1, u)
    ct0 = mk.NewCiphertext(params, names, level, p["scale"]).upload(h0)
    ct1 = mk.NewCiphertext(params, names, level, p["scale"]).upload(h1)
    ev = mk.NewEvaluator(params)
    res = ev.MulRelinNew(ct0, ct1, rlk)
    ids = list(range(k))
    _, ref = ks.mul_and_relin(level, ids, h0, ids, h1, rlk_h, u)
    nb, _ = ks.ckks_nb_rescales(level, p["scale"] * p["scale"], p["scale"])
    ref = np.stack([ks.ringQ.div_round_last_many(ref[s], nb)[0] for s in range(1 + k)])
    assert res.Level() == level - nb and (res.download() == ref).all()
    assert (ev.MulRelinNew(ct0, ct1, rlk).download() == ref).all()


def test_pn14_two_party_mulrelin_new(pn14):
    """BASELINE.json configs[0]: the reference benchmark's first case -- PN14QP439, 2 parties (mkckks/mkckks_benchmark_test.go:13,40-44)."""
    _mulrelin_case(pn14, 2)


@pytest.mark.parametrize("which", ["pn14", "pn15"])
def test_plain_forward_ntt_of_many_limbs(which, pn14, pn15):
    """ring.NTT of 20 polynomials at once (160 / 320 limbs): the H16 kernels without the Decompose reduction (ntt14_fwd_kernel<false>,
    ntt16_fwd_kernel<false>), out of place and in place (N = 2^15: the parked-half path), against the oracle on every modulus."""
    from mkhe_kklss_amd import mkrlwe
    f = pn14 if which == "pn14" else pn15
    p, ks, params, rng = f["pset"], f["ks"], f["params"], f["rng"]
    mods = p["Q"] + p["P"]
    N = 1 << p["logN"]
    cnt = 20
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods]) for _ in range(cnt)])
    src = mkrlwe.DeviceLimbs(params, cnt, len(mods)).upload(a)
    dst = mkrlwe.DeviceLimbs(params, cnt, len(mods))
    mkrlwe.ntt(params, src, dst)
    got = dst.download()
    mkrlwe.ntt(params, src, src)
    inplace = src.download()
    for c in (0, 7, cnt - 1):
        for j in range(len(mods)):
            r, i = (ks.ringQ, j) if j < len(p["Q"]) else (ks.ringP, j - len(p["Q"]))
            ref = r.ntt(i, a[c][j])
            assert (got[c][j] == ref).all() and (inplace[c][j] == ref).all(), (c, j)
    assert (got == inplace).all()


@pytest.mark.parametrize("which,cnt", [("pn14", 70), ("pn15", 20), ("pn16", 4)])
def test_plain_inverse_ntt_of_many_limbs(which, cnt, pn14, pn15):
    """ring.InvNTT / InvNTTLazy of many polynomials at once (560 limbs of 2^14, 320 of 2^15, 152 of 2^16): launches that fill the chip run the
    one-pass 2^14-point sub-transforms of the H16-class inverse kernel (ntt14_inv_kernel: whole limbs, halves + ntt_split_inv_kernel, quarters +
    ntt_pass4_inv_kernel), out of place and in place, against the oracle on every modulus and as the inverse of the forward transform."""
    from mkhe_kklss_amd import mkrlwe
    if which == "pn16":
        from oracle import oracle as O
        from mkhe_kklss_amd import mkckks
        p = H.PN16QP1761
        params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
        ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
        rng = np.random.default_rng(1616)
    else:
        f = pn14 if which == "pn14" else pn15
        p, ks, params, rng = f["pset"], f["ks"], f["params"], f["rng"]
    mods = p["Q"] + p["P"]
    N = 1 << p["logN"]
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods]) for _ in range(cnt)])
    src = mkrlwe.DeviceLimbs(params, cnt, len(mods)).upload(a)
    dst = mkrlwe.DeviceLimbs(params, cnt, len(mods))
    mkrlwe.ntt(params, src, dst, inverse=True)
    got = dst.download()
    mkrlwe.ntt(params, src, dst, inverse=True, lazy=True)
    lazy = dst.download()
    for c in (0, cnt - 1):
        for j in range(len(mods)):
            r, i = (ks.ringQ, j) if j < len(p["Q"]) else (ks.ringP, j - len(p["Q"]))
            ref = r.intt(i, a[c][j])
            assert (got[c][j] == ref).all(), (c, j)
    for j, q in enumerate(mods):                     # InvNTTLazy: the same residues, below 2q
        assert (lazy[:, j] < 2 * q).all() and (lazy[:, j] % q == got[:, j]).all(), j
    mkrlwe.ntt(params, src, src, inverse=True)       # in place
    assert (src.download() == got).all()
    mkrlwe.ntt(params, src, src)                     # and back
    assert (src.download() == a).all()

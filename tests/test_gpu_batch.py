"""B inputs in lock step (mkhe_*_batch, mkckks.BatchEvaluator; -m gpu): every output of a batched call equals the single-operation entry
point's on the same input bit for bit -- which the other GPU tests pin to the oracle -- and one case per operation is compared with the oracle
directly.  Shapes: hoisted and engine-hoisted operands, squares, an operand broadcast to every input (the model ciphertexts of cnn), batches
that overflow one launch's item lists (more than 64 external products: several launches per step), and the whole cnn inference on B images."""
import numpy as np
import pytest

import harness as H
import harness_cnn as HC

pytestmark = pytest.mark.gpu


class Case:
    def __init__(self, pset, names, seed):
        from oracle import oracle as O
        from mkhe_kklss_amd import mkckks, mkrlwe
        self.mkckks, self.mkrlwe, self.p, self.names = mkckks, mkrlwe, pset, names
        self.ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
        self.params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=0)
        self.rng = np.random.default_rng(seed)
        self.level = len(pset["Q"]) - 1
        self.rlk_h, self.rlk = {}, mkrlwe.RelinearizationKeySet(self.params)
        for n in names:
            k = tuple(H.uniform_swk(self.rng, self.ks) for _ in range(3))
            self.rlk_h[n] = k
            self.rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(self.params, n, *k))
        self.u_h = H.uniform_swk(self.rng, self.ks)
        self.params.AddCRS(-1, self.u_h)
        self.rots = [1, 5]
        self.rk_h, self.rtk, self.crs_h = {}, mkrlwe.RotationKeySet(), {}
        for r in self.rots:
            self.crs_h[r] = H.uniform_swk(self.rng, self.ks)
            self.params.AddCRS(r, self.crs_h[r])
            for n in names:
                self.rk_h[(n, r)] = H.uniform_swk(self.rng, self.ks)
                self.rtk.AddRotationKey(mkrlwe.RotationKey(self.params, r, n, self.rk_h[(n, r)]))
        self.ev = mkckks.NewEvaluator(self.params)

    def ct(self, ids, level=None):
        level = self.level if level is None else level
        h = H.uniform_ct(self.rng, self.ks, len(ids), level + 1)
        return h, self.mkckks.NewCiphertext(self.params, ids, level, self.p["scale"]).upload(h)

    def batch(self, ids, B, level=None):
        pairs = [self.ct(ids, level) for _ in range(B)]
        return [h for h, _ in pairs], self.mkckks.BatchCiphertext([c for _, c in pairs])


@pytest.fixture(scope="module", params=["N12_q4", "N14_pn14", "N15_q4"])
def case(request):
    # (N15: mkhe_mul_relin_batch runs its B evaluations IN FLIGHT there -- the single-operation path on the context and two internal ones, round robin
    # (csrc/batch.hip, round 6) -- instead of in lock step; everything else of this file is lock step on every ring)
    # (N14: the ring on which engine-internal Decompose launches go through ext_fused_lds_kernel -- round 5: mkhe_rotate_batch staged its digits there and
    # then read them as full transforms; every test of this file passed on N = 2^12, tools/fuzz_batch.py found it)
    c = Case({"N12_q4": H.small_ckks(12, 4), "N14_pn14": H.PN14QP439, "N15_q4": H.small_ckks(15, 4)}[request.param], ["a", "b", "c", "d"], 77)
    if request.param == "N15_q4":
        from mkhe_kklss_amd._abi import lib, check
        check(lib().mkhe_ctx_set_batch_lanes(c.params.ctx, 0))       # (by default only launch sets of 1536 hoisted limbs and more: four parties on the full chain)
    return c


@pytest.mark.parametrize("ids0,ids1,B", [(["a", "b"], ["a", "b"], 3), (["a"], ["b", "c"], 2), (["a", "b", "c", "d"], ["a", "b", "c", "d"], 7), ([], ["a"], 2), (["a", "b"], [], 2)])
@pytest.mark.parametrize("hoisted", [False, True])
def test_mul_relin_batch(case, ids0, ids1, B, hoisted):
    bev = case.mkckks.BatchEvaluator(case.params, B)
    h0, b0 = case.batch(ids0, B)
    h1, b1 = case.batch(ids1, B)
    hh0 = bev.HoistedForm(b0) if hoisted else None
    hh1 = bev.HoistedForm(b1) if hoisted else None
    out = bev.MulRelinHoistedNew(b0, b1, hh0, hh1, case.rlk)
    assert len(out) == B and out.ids == sorted(set(ids0) | set(ids1))
    for k in range(B):
        ref = case.ev.MulRelinNew(b0.cts[k], b1.cts[k], case.rlk)
        assert out.cts[k].Level() == ref.Level() and out.Scale == ref.Scale
        assert (out.cts[k].download() == ref.download()).all(), k
    # input 0 against the oracle itself
    names = out.ids
    idx = {n: i for i, n in enumerate(names)}
    _, o = case.ks.mul_and_relin(case.level, [idx[i] for i in ids0], h0[0], [idx[i] for i in ids1], h1[0], {idx[n]: case.rlk_h[n] for n in names}, case.u_h)
    nb, _ = case.ks.ckks_nb_rescales(case.level, case.p["scale"] ** 2, case.p["scale"])
    o = np.stack([case.ks.ringQ.div_round_last_many(o[s], nb)[0] for s in range(o.shape[0])])
    assert (out.cts[0].download() == o).all()


def test_square_and_broadcast_operand(case):
    B = 4
    bev = case.mkckks.BatchEvaluator(case.params, B)
    _, b0 = case.batch(["a", "c"], B)
    hb = bev.HoistedForm(b0)
    sq = bev.MulRelinHoistedNew(b0, b0, hb, hb, case.rlk)                     # squares: op1 is op0
    _, model = case.ct(["b"])                                                 # one ciphertext for every input (cnn's kernels / weights)
    hm = case.ev.HoistedForm(model)
    pr = bev.MulRelinHoistedNew(b0, model, hb, hm, case.rlk)
    pr2 = bev.MulRelinNew(b0, model, case.rlk)
    for k in range(B):
        assert (sq.cts[k].download() == case.ev.MulRelinNew(b0.cts[k], b0.cts[k], case.rlk).download()).all()
        ref = case.ev.MulRelinNew(b0.cts[k], model, case.rlk).download()
        assert (pr.cts[k].download() == ref).all() and (pr2.cts[k].download() == ref).all()
    s = bev.AddNew(pr, pr2)                                                   # batch + batch
    t = bev.SubNew(sq, bev.AddNew(sq, sq))
    for k in range(B):
        assert (s.cts[k].download() == case.ev.AddNew(pr.cts[k], pr2.cts[k]).download()).all()
        assert (t.cts[k].download() == case.ev.SubNew(sq.cts[k], case.ev.AddNew(sq.cts[k], sq.cts[k])).download()).all()
    # a batch plus ONE ciphertext with other ids (cnn's bias): the union of the id sets, components copied
    _, bias = case.ct(["d"], level=sq.Level())
    w = bev.AddNew(sq, bias)
    for k in range(B):
        assert (w.cts[k].download() == case.ev.AddNew(sq.cts[k], bias).download()).all()


@pytest.mark.parametrize("ids,B", [(["a", "b"], 3), (["a", "b", "c", "d"], 9), (["c"], 2)])
@pytest.mark.parametrize("hoisted", [False, True])
def test_rotate_batch(case, ids, B, hoisted):
    bev = case.mkckks.BatchEvaluator(case.params, B)
    hs, b = case.batch(ids, B)
    hb = bev.HoistedForm(b) if hoisted else None
    for r in case.rots:
        out = bev.RotateHoistedNew(b, r, hb, case.rtk) if hoisted else bev.RotateNew(b, r, case.rtk)
        for k in range(B):
            assert (out.cts[k].download() == case.ev.RotateNew(b.cts[k], r, case.rtk).download()).all(), (r, k)
        o = case.ks.rotate(case.level, pow(5, r, 2 << case.p["logN"]), list(range(len(ids))), hs[B - 1], [case.rk_h[(n, r)] for n in sorted(ids)], case.crs_h[r])
        assert (out.cts[B - 1].download() == o).all()
    # 6 = 5 + 1 is not a CRS index: RotateNew decomposes it (evaluator.go:516-523) -- here 6 = 2 + 4 has no keys, so use what exists: rotation by 0
    z = bev.RotateNew(b, 0, case.rtk)
    for k in range(B):
        assert (z.cts[k].download() == b.cts[k].download()).all()


def test_hoisted_form_batch_matches_decompose(case):
    B = 5
    bev = case.mkckks.BatchEvaluator(case.params, B)
    hs, b = case.batch(["a", "d"], B, level=2)
    hb = bev.HoistedForm(b)
    beta = case.ks.beta(2)
    act = [0, 1, 2] + [len(case.p["Q"]) + j for j in range(len(case.p["P"]))]
    for k in range(B):
        for i, n in enumerate(["a", "d"]):
            ref = case.ks.decompose(2, hs[k][1 + i])
            got = hb.hoisted[k].Value[n].download()
            assert (got[:beta][:, act] == ref[:beta][:, act]).all(), (k, n)


def test_batch_shape_errors(case):
    from mkhe_kklss_amd._abi import MkheError
    bev = case.mkckks.BatchEvaluator(case.params, 2)
    _, c0 = case.ct(["a"])
    _, c1 = case.ct(["b"])
    with pytest.raises(MkheError):
        case.mkckks.BatchCiphertext([c0, c1])
    _, b = case.batch(["a"], 2)
    with pytest.raises(MkheError):
        bev.RotateNew(b, 5, case.mkrlwe.RotationKeySet())                     # no rotation key
    out = bev.RotateNew(b, 1, case.rtk)                                       # the context is usable after the error
    assert (out.cts[1].download() == case.ev.RotateNew(b.cts[1], 1, case.rtk).download()).all()


TWO = dict(image="dataOwner", kernels="modelOwner", fc1="modelOwner", fc2="modelOwner")
FOUR = dict(image="dataOwner", kernels="convOwner", fc1="fc1Owner", fc2="fc2Owner")


@pytest.mark.parametrize("owners,B", [(TWO, 3), (FOUR, 8)], ids=["2party_B3", "4party_B8"])
def test_cnn_inference_on_a_batch_of_images(owners, B):
    """cnn/cnn.go on B images at once: every image's output ciphertext equals its own single-image inference bit for bit, and image 0 decrypts
    to the plaintext network's logits"""
    from mkhe_kklss_amd import cnn, mkckks
    sc = HC.CnnScenario(owners, seed=11)
    models = [HC.synthetic_model(20 + k) for k in range(B)]
    m0 = sc.encrypt_model(models[0])                                          # the model: one for all images
    imgs = [sc.encrypt(HC.pack_image(m), owners["image"]) for m in models]
    pt, pt_scale = sc.mask_plaintext(sc.level - 4)
    bev = mkckks.BatchEvaluator(sc.params, B)
    out = cnn.Inference(bev, sc.rlkSet, sc.rtkSet, mkckks.BatchCiphertext(imgs), m0["ctKernels"], m0["ctFC1"], m0["ctFC2"], m0["ctB1"], m0["ctB2"], pt, pt_scale)
    assert len(out) == B and out.Level() == 0
    for k in range(B):
        ref = cnn.Inference(sc.eval, sc.rlkSet, sc.rtkSet, imgs[k], m0["ctKernels"], m0["ctFC1"], m0["ctFC2"], m0["ctB1"], m0["ctB2"], pt, pt_scale)
        assert out.cts[k].Scale == ref.Scale and (out.cts[k].download() == ref.download()).all(), k
    got = sc.decrypt(out.cts[0])[:HC.NCLS]
    exp = HC.plain_forward(models[0])
    assert int(np.argmax(got.real)) == int(np.argmax(exp)) and np.abs(got.real - exp).max() < 1e-3 * max(1.0, np.abs(exp).max())


# ---------------------------------------------------------------- round 5
def test_batch_rejects_an_output_that_is_another_items_input(case):
    """The items of a batch run as one launch set, in no order: output k aliasing input j != k is an error (include/mkhe.h), not a race."""
    from mkhe_kklss_amd._abi import MkheError
    B = 3
    bev = case.mkckks.BatchEvaluator(case.params, B)
    _, b0 = case.batch(["a", "b"], B)
    _, b1 = case.batch(["a", "b"], B)
    crossed = case.mkckks.BatchCiphertext([b0.cts[1], b0.cts[2], b0.cts[0]])          # out[k] = op0[k + 1]
    with pytest.raises(MkheError, match="input . of another item"):
        from mkhe_kklss_amd._abi import check, lib
        check(lib().mkhe_ct_binary_batch(case.params.ctx, 0, B, bev._h(b0), bev._h(b1), bev._h(crossed)))
    rk = [case.rtk.GetRotationKey(i, 1).Value.h for i in b0.ids]
    from mkhe_kklss_amd._abi import handle_array
    with pytest.raises(MkheError, match="input . of another item"):
        check(lib().mkhe_rotate_batch(case.params.ctx, case.params.GaloisElementForColumnRotationBy(1), B, bev._h(b0), None, handle_array(rk), case.params.CRS[1].h, bev._h(crossed)))
    # in place per item stays legal for the elementwise entry: a[k] += b[k]
    ref = [(case.ev.AddNew(b0.cts[k], b1.cts[k])).download() for k in range(B)]
    check(lib().mkhe_ct_binary_batch(case.params.ctx, 0, B, bev._h(b0), bev._h(b1), bev._h(b0)))
    for k in range(B):
        assert (b0.cts[k].download() == ref[k]).all()


def test_batch_mul_relin_without_the_fused_rescale_matches_the_single_evaluator(case):
    """BatchEvaluator mirrors Evaluator.MulRelinHoistedNew's branch: with fuse_rescale off the product is formed at its level and one
    mkhe_rescale(nb) per input follows (it used to fold the first Rescale regardless)."""
    B = 2
    bev = case.mkckks.BatchEvaluator(case.params, B)
    _, b0 = case.batch(["a", "b"], B)
    _, b1 = case.batch(["b", "c"], B)
    fused = bev.MulRelinNew(b0, b1, case.rlk)
    old = bev.ev.fuse_rescale
    bev.ev.fuse_rescale = False
    try:
        plain = bev.MulRelinNew(b0, b1, case.rlk)
        single = [bev.ev.MulRelinNew(b0.cts[k], b1.cts[k], case.rlk) for k in range(B)]
    finally:
        bev.ev.fuse_rescale = old
    for k in range(B):
        assert plain.cts[k].Level() == fused.cts[k].Level() == single[k].Level()
        assert (plain.cts[k].download() == fused.cts[k].download()).all() and (plain.cts[k].download() == single[k].download()).all()


def test_pool_statistics_and_trim(case):
    """The stream-ordered pools are bounded per device and can be handed back (mkhe_pool_trim; the engine does it itself when hipMalloc fails)."""
    from mkhe_kklss_amd._abi import check, lib
    ctx = case.params.ctx
    cts = [case.ct(["a", "b"])[1] for _ in range(4)]
    want = [c.download() for c in cts]
    del cts[2:]                                            # two handles go back to the pool
    import gc; gc.collect()
    assert lib().mkhe_pool_held_bytes(ctx) > 0
    check(lib().mkhe_pool_trim(ctx))
    assert lib().mkhe_pool_held_bytes(ctx) == 0
    assert (cts[0].download() == want[0]).all() and (cts[1].download() == want[1]).all()
    _, again = case.ct(["a", "b"])                         # allocation after a trim
    assert again.download().shape == want[0].shape
    assert lib().mkhe_pool_held_bytes(None) == -1 and lib().mkhe_ntt_choice(None, 1792, 1) == -2


# ---------------------------------------------------------------- round 5: lanes with their own rotation, rotate-and-add in one pass
@pytest.mark.parametrize("ids,level", [(["a", "b"], None), (["a", "b", "c", "d"], 2), (["c"], 1), (["a", "b", "c"], 0)])
@pytest.mark.parametrize("hoisted", [False, True])
def test_rotate_multi_each_lane_its_own_rotation(case, ids, level, hoisted):
    """mkhe_rotate_multi: lane b rotates by rots[b] with the keys of THAT rotation; every lane against the oracle's Rotate (keyswitch_hoisted.go:183-247)"""
    rots = [1, 5, 5, 1, 1]
    B = len(rots)
    level = case.level if level is None else level
    bev = case.mkckks.BatchEvaluator(case.params, B)
    hs, b = case.batch(ids, B, level)
    hb = bev.HoistedForm(b) if hoisted else None
    out = bev._rotate_multi(b, rots, hb, case.rtk, None)
    for k, r in enumerate(rots):
        o = case.ks.rotate(level, pow(5, r, 2 << case.p["logN"]), list(range(len(ids))), hs[k], [case.rk_h[(n, r)] for n in sorted(ids)], case.crs_h[r])
        assert (out.cts[k].download() == o).all(), (k, r)
    # one input broadcast to every lane (cnn: the image / the FC1 input), hoisted once
    h1, c1 = case.ct(ids, level)
    hh = case.ev.HoistedForm(c1) if hoisted else None
    out = bev.RotateHoistedNew(c1, rots, hh, case.rtk) if hoisted else bev._rotate_multi(c1, rots, None, case.rtk, None)
    for k, r in enumerate(rots):
        o = case.ks.rotate(level, pow(5, r, 2 << case.p["logN"]), list(range(len(ids))), h1, [case.rk_h[(n, r)] for n in sorted(ids)], case.crs_h[r])
        assert (out.cts[k].download() == o).all(), (k, r)


@pytest.mark.parametrize("ids,level", [(["a", "b"], None), (["a", "b", "c", "d"], 1), (["d"], 0)])
def test_rotate_and_add_is_the_two_calls(case, ids, level):
    """AddNew(ct, RotateNew(ct, r)) in one engine call (the add on the rotation's store; cnn/cnn.go:33-37): against the oracle's Rotate followed by
    ring.Add, for one ciphertext, for a batch, and with an addend that is NOT the rotated ciphertext; a rotated zero coefficient with a sign flip is
    stored as q by the permutation (keyswitch.go:290) and must still add up canonically"""
    from mkhe_kklss_amd._abi import check, lib, handle_array, MkheError
    import ctypes as C
    level = case.level if level is None else level
    N, Q = 1 << case.p["logN"], case.p["Q"]
    h, c = case.ct(ids, level)
    def ref(hin, hadd, r):
        o = case.ks.rotate(level, pow(5, r, 2 * N), list(range(len(ids))), hin, [case.rk_h[(n, r)] for n in sorted(ids)], case.crs_h[r])
        return np.stack([np.stack([case.ks.ringQ.add(j, hadd[s, j], o[s, j]) for j in range(level + 1)]) for s in range(o.shape[0])])
    for r in case.rots:
        got = case.ev.RotateAndAddNew(c, r, case.rtk)
        assert (got.download() == ref(h, h, r)).all(), r
        two = case.ev.AddNew(c, case.ev.RotateNew(c, r, case.rtk))
        assert (got.download() == two.download()).all() and got.Scale == two.Scale and got.Level() == two.Level()
    assert (case.ev.RotateAndAddNew(c, 0, case.rtk).download() == case.ev.AddNew(c, c).download()).all()
    # batch, and a different addend per lane
    B = 3
    bev = case.mkckks.BatchEvaluator(case.params, B)
    hs, b = case.batch(ids, B, level)
    ha, a = case.batch(ids, B, level)
    got = bev.RotateAndAddNew(b, 5, case.rtk)
    oth = bev._rotate_multi(b, [5, 1, 5], None, case.rtk, a)
    for k in range(B):
        assert (got.cts[k].download() == ref(hs[k], hs[k], 5)).all(), k
        assert (oth.cts[k].download() == ref(hs[k], ha[k], [5, 1, 5][k])).all(), k
    if level < case.level:
        # an addend (and an input) one level ABOVE the output: only the first level + 1 limbs take part (mkrlwe/keyswitch_hoisted.go:183-190: the level is ctOut's)
        hu, up = case.batch(ids, B, level + 1)
        hv, add_up = case.batch(ids, B, level + 1)
        out = bev._new(set(ids), level, case.p["scale"])
        rk = [case.rtk.GetRotationKey(i, r).Value.h for r in [1, 5, 1] for i in up.ids]
        galv = (C.c_uint64 * B)(*[case.params.GaloisElementForColumnRotationBy(r) for r in [1, 5, 1]])
        check(lib().mkhe_rotate_multi(case.params.ctx, B, galv, bev._h(up), None, handle_array(rk), handle_array([case.params.CRS[r].h for r in [1, 5, 1]]), bev._h(add_up), bev._h(out)))
        for k in range(B):
            assert (out.cts[k].download() == ref(np.ascontiguousarray(hu[k][:, :level + 1]), np.ascontiguousarray(hv[k][:, :level + 1]), [1, 5, 1][k])).all(), k
    # errors: an output that is an addend; an addend with other ids
    rk = [case.rtk.GetRotationKey(i, 1).Value.h for i in c.ids]
    gal = (C.c_uint64 * 1)(case.params.GaloisElementForColumnRotationBy(1))
    out = case.mkckks.NewCiphertext(case.params, ids, level, case.p["scale"])
    with pytest.raises(MkheError, match="aliases an addend"):
        check(lib().mkhe_rotate_multi(case.params.ctx, 1, gal, handle_array([c.h]), None, handle_array(rk), handle_array([case.params.CRS[1].h]), handle_array([out.h]), handle_array([out.h])))
    _, other = case.ct(["a", "c"] if ids != ["a", "c"] else ["b"], level)
    with pytest.raises(MkheError, match="addend must carry the ids"):
        check(lib().mkhe_rotate_multi(case.params.ctx, 1, gal, handle_array([c.h]), None, handle_array(rk), handle_array([case.params.CRS[1].h]), handle_array([other.h]), handle_array([out.h])))


@pytest.mark.parametrize("n,ids,level", [(2, ["a", "b"], None), (8, ["a", "b", "c"], 2), (19, ["d"], 1), (1, ["a"], 0)])
def test_sum_of_ciphertexts_is_the_add_chain(case, n, ids, level):
    """mkhe_ct_sum: out = cts[0] + ... + cts[n-1] in one launch (19 > 16: partial sums), against the oracle's ring.Add chain and the AddNew chain"""
    level = case.level if level is None else level
    pairs = [case.ct(ids, level) for _ in range(n)]
    got = case.ev.SumNew([c for _, c in pairs])
    ref = pairs[0][0].copy()
    for h, _ in pairs[1:]:
        ref = np.stack([np.stack([case.ks.ringQ.add(j, ref[s, j], h[s, j]) for j in range(level + 1)]) for s in range(ref.shape[0])])
    assert (got.download() == ref).all()
    chain = pairs[0][1]
    for _, c in pairs[1:]:
        chain = case.ev.AddNew(chain, c)
    assert (got.download() == chain.download()).all() and got.Scale == chain.Scale
    # summands of different shapes fall back to the chain (id union, scale matching): same as AddNew
    _, other = case.ct(["a", "d"] if ids != ["a", "d"] else ["b"], level)
    mixed = case.ev.SumNew([pairs[0][1], other])
    assert (mixed.download() == case.ev.AddNew(pairs[0][1], other).download()).all()

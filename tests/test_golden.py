"""Golden fixtures (tests/golden/*.npz, produced by the independent Python model -- see
tests/golden/make_golden.py): the C oracle must reproduce them on CPU; the HIP NTT must reproduce
the size-2^10 vector on the GPU."""
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", ["alpha1_n16", "alpha2_n16"])
def test_oracle_reproduces_golden(name):
    g = load(name)
    logN, Qs, Ps = int(g["logN"]), [int(q) for q in g["Q"]], [int(p) for p in g["P"]]
    ks = O.KeySwitcher(logN, Qs, Ps, 2)
    nq = len(Qs)
    for j in range(nq + len(Ps)):
        r, i = (ks.ringQ, j) if j < nq else (ks.ringP, j - nq)
        assert r.psi(i) == int(g["psi"][j])
        assert (r.ntt(i, g["ntt_in"][j]) == g["ntt_out"][j]).all()
    for level in (nq - 1, max(0, nq - 2)):
        beta = ks.beta(level)
        act = list(range(level + 1)) + [nq + j for j in range(len(Ps))]
        h = ks.decompose(level, g["dec_in"])
        assert (h[:beta][:, act] == g["dec_out_l%d" % level][:beta][:, act]).all()
        assert (ks.external_product(level, g["dec_in"], g["bg"]) == g["ext_out_l%d" % level]).all()
    rlk = {i: (g["rlk%d_b" % i], g["rlk%d_d" % i], g["rlk%d_v" % i]) for i in range(3)}
    for level in (nq - 1, 1):
        ido, out = ks.mul_and_relin(level, list(g["ids0"]), g["op0"], list(g["ids1"]), g["op1"], rlk, g["crs_u"])
        assert ido == list(g["mr_ids_l%d" % level])
        assert (out == g["mr_out_l%d" % level]).all()
    out, _ = ks.ringQ.div_round_last_many(g["dec_in"], 1)
    assert (out == g["rescale_out"]).all()
    assert (ks.ringQ.permute(int(g["galEl"]), g["dec_in"]) == g["perm_out"]).all()


def test_oracle_ntt_n1024_golden():
    g = load("ntt_n1024")
    mods = [int(q) for q in g["mods"]]
    r = O.Ring(10, mods)
    for j in range(len(mods)):
        assert r.psi(j) == int(g["psi"][j])
        assert (r.ntt(j, g["ntt_in"][j]) == g["ntt_out"][j]).all()


@pytest.mark.gpu
def test_gpu_ntt_n1024_golden():
    from mkhe_kklss_amd import mkrlwe
    g = load("ntt_n1024")
    mods = [int(q) for q in g["mods"]]
    params = mkrlwe.Parameters(10, mods[:2], mods[2:], gamma=1)
    for j in range(3):
        assert params.Psi(j) == int(g["psi"][j])
    src = mkrlwe.DeviceLimbs(params, 1, 3).upload(g["ntt_in"][None])
    dst = mkrlwe.DeviceLimbs(params, 1, 3)
    mkrlwe.ntt(params, src, dst)
    assert (dst.download()[0] == g["ntt_out"]).all()
    mkrlwe.ntt(params, dst, src, inverse=True)
    assert (src.download()[0] == g["ntt_in"]).all()


def _bfv(g):
    return O.BFV(int(g["logN"]), [int(q) for q in g["Q"]], [int(q) for q in g["QMul"]], [int(p) for p in g["P"]], int(g["T"]))


def test_oracle_reproduces_bfv_golden():
    g = load("bfv_n16")
    bfv = _bfv(g)
    rlk = {i: tuple(g["rlk%d_%s" % (i, nm)] for nm in ("b1", "b2", "d1", "d2", "v")) for i in range(3)}
    ido, out = bfv.mul_relin_new(list(g["ids0"]), g["op0"], list(g["ids1"]), g["op1"], rlk, g["crs_u"])
    assert ido == list(g["mr_ids"]) and (out == g["mr_out"]).all()
    assert (bfv.modup_q_to_r(g["conv_in"]) == g["modup_out"]).all()
    assert (bfv.rescale(g["conv_in"]) == g["rescale_out"]).all()
    c = load("bfv_conv_n1024")
    bfv = _bfv(c)
    assert (bfv.modup_q_to_r(c["conv_in"]) == c["modup_out"]).all()
    assert (bfv.rescale(c["conv_in"]) == c["rescale_out"]).all()
    assert (bfv.ntt_r(c["r_in"]) == c["r_ntt"]).all()
    assert (bfv.quantize(c["r_ntt"]) == c["quantize_out"]).all()


@pytest.mark.gpu
def test_gpu_bfv_conversions_golden():
    from mkhe_kklss_amd import mkbfv, mkrlwe
    c = load("bfv_conv_n1024")
    params = mkbfv.Parameters(int(c["logN"]), [int(q) for q in c["Q"]], [int(q) for q in c["QMul"]], [int(p) for p in c["P"]], int(c["T"]))
    conv = mkbfv.FastBasisExtender(params)
    nq = len(c["Q"])
    src = mkrlwe.DeviceLimbs(params, 1, nq).upload(c["conv_in"][None])
    r = mkbfv.PolyR(params, 1)
    conv.ModUpQtoR(src, r)
    assert (r.download()[0] == c["modup_out"]).all()
    conv.Rescale(src, r)
    assert (r.download()[0] == c["rescale_out"]).all()
    rn = mkbfv.PolyR(params, 1).upload(c["r_ntt"][None])
    out = mkrlwe.DeviceLimbs(params, 1, nq)
    conv.Quantize(rn, out, params.T())
    assert (out.download()[0] == c["quantize_out"]).all()


def test_crs_expansion_golden():
    """the CRS stream definition (Philox4x32-10 words, mask-and-reject, MForm) pinned by a fixture computed in plain Python integers"""
    g = load("crs_n16")
    ks = O.KeySwitcher(int(g["logN"]), [int(q) for q in g["Q"]], [int(p) for p in g["P"]], 1)
    assert (O.KeyGen(ks).crs_expand(int(g["seed"]), int(g["idx"])) == g["crs"]).all()

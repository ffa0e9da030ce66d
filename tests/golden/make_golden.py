#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run from the repo root:  python tests/golden/make_golden.py

The reference ships no golden vectors and cannot run here (no Go toolchain, lattigo not
vendored), so these fixtures are DATA produced by the independent Python big-integer model
(oracle/pymodel.py: NTT by definition, exact-CRT ModDown, schoolbook products) for small rings, plus
one size-2^10 single-limb NTT for the GPU kernels (whose minimum ring degree is 2^10).
Inputs are seeded; outputs are whatever the model computes -- nothing here is copied from the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H                     # noqa: E402
from oracle import pymodel as M         # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rnd_poly(rng, mods, N):
    return np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods])


def u64(x):
    return np.array(x, dtype=np.uint64)


def small_case(name, logN, Qs, Ps, seed):
    N = 1 << logN
    rng = np.random.default_rng(seed)
    mdl = M.Model(logN, Qs, Ps, 2)
    beta, m = mdl.beta(len(Qs) - 1), len(Qs) + len(Ps)
    d = dict(logN=logN, Q=u64(Qs), P=u64(Ps), psi=u64(mdl.psi))
    # NTT of one poly per modulus
    a = rnd_poly(rng, Qs + Ps, N)
    d["ntt_in"] = a
    d["ntt_out"] = u64([M.ntt_def([int(v) for v in a[j]], q, mdl.psi[j], logN) for j, q in enumerate(Qs + Ps)])
    # decompose + external product at two levels
    poly = rnd_poly(rng, Qs, N)
    bg = np.stack([rnd_poly(rng, Qs + Ps, N) for _ in range(beta)])
    d["dec_in"], d["bg"] = poly, bg
    for level in (len(Qs) - 1, max(0, len(Qs) - 2)):
        h = mdl.decompose(poly, level)
        arr = np.zeros((beta, m, N), dtype=np.uint64)
        for i in range(mdl.beta(level)):
            for j in mdl.limb_index(level):
                arr[i, j] = u64(h[i][j])
        d["dec_out_l%d" % level] = arr
        d["ext_out_l%d" % level] = u64(mdl.external_product([[int(v) for v in l] for l in poly], bg, level))
    # MulAndRelin: op0 ids {0,1}, op1 ids {1,2}
    ids0, ids1 = [0, 1], [1, 2]
    op0 = np.stack([rnd_poly(rng, Qs, N) for _ in range(3)])
    op1 = np.stack([rnd_poly(rng, Qs, N) for _ in range(3)])
    rlk = {i: tuple(np.stack([rnd_poly(rng, Qs + Ps, N) for _ in range(beta)]) for _ in range(3)) for i in range(3)}
    u = np.stack([rnd_poly(rng, Qs + Ps, N) for _ in range(beta)])
    d.update(op0=op0, op1=op1, crs_u=u, ids0=np.array(ids0), ids1=np.array(ids1))
    for i in range(3):
        d["rlk%d_b" % i], d["rlk%d_d" % i], d["rlk%d_v" % i] = rlk[i]
    for level in (len(Qs) - 1, 1):
        ido, out = mdl.mul_and_relin(level, ids0, op0, ids1, op1, rlk, u)
        d["mr_ids_l%d" % level] = np.array(ido)
        d["mr_out_l%d" % level] = u64(out)
    # rescale + automorphism
    d["rescale_out"] = u64([[M.div_round_last([int(poly[l][k]) for l in range(len(Qs))], Qs)[j] for k in range(N)]
                            for j in range(len(Qs) - 1)])
    g = pow(5, 3, 2 * N)
    d["galEl"] = np.uint64(g)
    d["perm_out"] = u64([M.automorphism([int(v) for v in poly[l]], g, Qs[l], logN) for l in range(len(Qs))])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name)


def ntt_1024():
    logN, N = 10, 1024
    rng = np.random.default_rng(77)
    mods = [H.PN15QP880["Q"][0], H.PN15QP880["Q"][1], H.PN15QP880["P"][0]]
    psi = [M.find_psi(q, N) for q in mods]
    a = rnd_poly(rng, mods, N)
    out = u64([M.ntt_def([int(v) for v in a[j]], q, psi[j], logN) for j, q in enumerate(mods)])
    np.savez_compressed(os.path.join(OUT, "ntt_n1024.npz"), logN=logN, mods=u64(mods), psi=u64(psi), ntt_in=a, ntt_out=out)
    print("wrote ntt_n1024")


def bfv_case():
    """mkbfv: full MulRelinNew at N = 16 by the independent model (oracle/pymodel.py BfvModel)"""
    import harness_bfv as HB
    logN, N = 4, 16
    Qs, QMs, Ps, T = HB.BFV_PN15QP880["Q"][:2], HB.BFV_PN15QP880["QMul"][:2], HB.BFV_PN15QP880["P"], 65537
    rng = np.random.default_rng(21)
    mdl = M.BfvModel(logN, Qs, QMs, Ps, T)
    swk = lambda: np.stack([rnd_poly(rng, Qs + Ps, N) for _ in range(len(Qs))])
    ids0, ids1 = [0, 1], [1, 2]
    op0 = np.stack([rnd_poly(rng, Qs, N) for _ in range(3)])
    op1 = np.stack([rnd_poly(rng, Qs, N) for _ in range(3)])
    rlk = {i: tuple(swk() for _ in range(5)) for i in range(3)}
    u = swk()
    d = dict(logN=logN, Q=u64(Qs), QMul=u64(QMs), P=u64(Ps), T=np.uint64(T), op0=op0, op1=op1, crs_u=u,
             ids0=np.array(ids0), ids1=np.array(ids1))
    for i in range(3):
        for k, nm in enumerate(("b1", "b2", "d1", "d2", "v")):
            d["rlk%d_%s" % (i, nm)] = rlk[i][k]
    ido, out = mdl.mul_relin_new(ids0, op0, ids1, op1, rlk, u)
    d["mr_ids"], d["mr_out"] = np.array(ido), u64(out)
    x = rnd_poly(rng, Qs, N)
    d["conv_in"], d["modup_out"], d["rescale_out"] = x, u64(mdl.modup_q_to_r(x)), u64(mdl.rescale(x))
    np.savez_compressed(os.path.join(OUT, "bfv_n16.npz"), **d)
    print("wrote bfv_n16")


def bfv_conv_1024():
    """mkbfv base conversions at the device's minimum ring degree: ModUpQtoR, Rescale, ringR NTT + Quantize"""
    import harness_bfv as HB
    logN, N = 10, 1024
    Qs, QMs, Ps, T = HB.BFV_PN15QP880["Q"][:2], HB.BFV_PN15QP880["QMul"][:2], HB.BFV_PN15QP880["P"], 65537
    rng = np.random.default_rng(22)
    mdl = M.BfvModel(logN, Qs, QMs, Ps, T)
    x = rnd_poly(rng, Qs, N)
    x[:, 0] = 0
    x[:, 1] = u64([q - 1 for q in Qs])
    y = rnd_poly(rng, Qs + QMs, N)
    yn = u64([M.ntt_def([int(v) for v in y[l]], m, mdl.psiR[l], logN) for l, m in enumerate(Qs + QMs)])
    np.savez_compressed(os.path.join(OUT, "bfv_conv_n1024.npz"), logN=logN, Q=u64(Qs), QMul=u64(QMs), P=u64(Ps), T=np.uint64(T),
                        conv_in=x, modup_out=u64(mdl.modup_q_to_r(x)), rescale_out=u64(mdl.rescale(x)),
                        r_in=y, r_ntt=yn, quantize_out=u64(mdl.quantize_coeff(y)))
    print("wrote bfv_conv_n1024")


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11) in plain Python integers, independent of the C oracle"""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return c


def crs_case():
    """CRS expansion (include/mkhe.h mkhe_crs_expand): uniform sample by mask-and-reject over the Philox words, then MForm"""
    logN, Qs, Ps, seed, idx = 4, H.PN15QP880["Q"][:3], H.PN15QP880["P"][:1], 0x4D4B4845, -1
    N, mods = 1 << logN, Qs + Ps
    beta, m = len(Qs), len(mods)
    out = np.zeros((beta, m, N), dtype=np.uint64)
    for d in range(beta):
        for j, q in enumerate(mods):
            mask = (1 << q.bit_length()) - 1
            for w in range(N):
                block = 0
                while True:
                    o = philox4x32_10([w, d * m + j, idx & 0xffffffff, block], [seed & 0xffffffff, seed >> 32])
                    c = [((o[1] << 32) | o[0]) & mask, ((o[3] << 32) | o[2]) & mask]
                    hit = [x for x in c if x < q]
                    if hit:
                        out[d, j, w] = (hit[0] << 64) % q          # MForm
                        break
                    block += 1
    np.savez_compressed(os.path.join(OUT, "crs_n16.npz"), logN=logN, Q=u64(Qs), P=u64(Ps), seed=np.uint64(seed), idx=np.int64(idx), crs=out)
    print("wrote crs_n16")


if __name__ == "__main__":
    crs_case()
    bfv_case()
    bfv_conv_1024()
    small_case("alpha1_n16", 4, H.PN15QP880["Q"][:3], H.PN15QP880["P"], 11)
    small_case("alpha2_n16", 4, H.PN16_Q[:5], H.PN16_P, 12)
    ntt_1024()

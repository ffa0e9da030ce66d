"""Seeded MK-BFV test harness (TEST INFRASTRUCTURE): parameter sets, the BFV relinearization key
generation, encryption and decryption restated from the reference so mkbfv_test.go's exact-decrypt
property tests can be replayed on the oracle and on the device.

Restates (CPU, through the oracle's ring ops):
  mkbfv/mkbfv_test.go:29-112   PN15QP880 / PN14QP439 literals (Q, QMul, P, T)
  mkbfv/keygen.go:25-91        GenRelinearizationKey: (b1, b2, d1, d2, v)
  mkbfv/keygen.go:94-161       GenBFVSwitchingKey: gadgets floor(t*P*(QQMul/q_i)*[(QQMul/q_i)^-1]_{q_i} / QMul)
  mkbfv/encryptor.go:27-37     EncryptMsg  (plaintext = round(Q/T * m), coefficient domain)
  mkbfv/decryptor.go:31-56     Decrypt     (round(T/Q * phase) mod T)
Messages are plain polynomials over Z_T (the reference packs them with lattigo's batch encoder,
client-side code outside the evaluated path); products are negacyclic products mod T.
"""
import numpy as np

import harness as H
from oracle import oracle as O

BFV_PN15QP880 = dict(
    logN=15,
    Q=[0x3fffffffd60001, 0x3fffffff6d0001, 0x3fffffff550001, 0x3fffffff360001, 0x3fffffff000001,
       0x3ffffffef40001, 0x3ffffffed30001, 0x3ffffffe970001, 0x3ffffffe800001, 0x3ffffffe410001,
       0x7fffffffe90001, 0x7fffffffbd0001, 0x7fffffffaa0001, 0x7fffffff9f0001],
    QMul=[0x3fffffffca0001, 0x3fffffff5d0001, 0x3fffffff390001, 0x3fffffff2a0001, 0x3ffffffefa0001,
          0x3ffffffed70001, 0x3ffffffeaa0001, 0x3ffffffe920001, 0x3ffffffe790001, 0x3ffffffe320001,
          0x7fffffffbf0001, 0x7fffffffba0001, 0x7fffffffa50001, 0x7fffffff7e0001],
    P=[0xffffffffffc0001, 0xfffffffff840001], T=65537)
BFV_PN14QP439 = dict(
    logN=14,
    Q=[0x1fffffffe30001, 0x1fffffffd10001, 0x1fffffffbf0001, 0x1fffffffb60001, 0x1fffffff920001, 0x3fffffffd60001],
    QMul=[0x1fffffffd80001, 0x1fffffffc50001, 0x1fffffffb90001, 0x1fffffffa50001, 0x1fffffff900001, 0x3fffffffca0001],
    P=[0xffffffffffc0001, 0xfffffffff840001], T=65537)


def small_bfv(logN, nq=3, big=False):
    """reduced-size set with the reference's own primes (all = 1 mod 2^16)."""
    src = BFV_PN15QP880
    if big:     # include the 55-bit tail of the Q / QMul chains
        sel = list(range(nq - 1)) + [10]
    else:
        sel = list(range(nq))
    return dict(logN=logN, Q=[src["Q"][i] for i in sel], QMul=[src["QMul"][i] for i in sel], P=src["P"], T=src["T"])


def make_bfv(pset):
    return O.BFV(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])


def uniform_bfv_inputs(pset, parties, seed):
    """uniform residues: two ciphertexts, (b1,b2,d1,d2,v) per party, CRS u -- SURVEY.md 8d style inputs"""
    rng = np.random.default_rng(seed)
    Q, P, N = pset["Q"], pset["P"], 1 << pset["logN"]
    swk = lambda: np.stack([H.uniform_poly(rng, Q + P, N) for _ in range(len(Q))])
    ct = lambda: np.stack([H.uniform_poly(rng, Q, N) for _ in range(1 + parties)])
    return dict(op0=ct(), op1=ct(), rlk={i: tuple(swk() for _ in range(5)) for i in range(parties)}, u=swk())


class BFVKeyGen(H.KeyGen):
    def __init__(self, bfv, seed=1):
        super().__init__(bfv.ks, seed)
        self.bfv = bfv
        self.T = bfv.T
        self.Qprod = 1
        for q in bfv.Q:
            self.Qprod *= q
        self.QMulprod = 1
        for q in bfv.QMul:
            self.QMulprod *= q

    def _scalar(self, poly, G):
        out = np.empty_like(poly)
        for j in range(poly.shape[0]):
            r, i = self._ring(j)
            out[j] = r.mul_scalar(i, poly[j], G % self.QP[j])
        return out

    def gen_bfv_switching_key(self, sk):
        """(swk1, swk2) = G_i * s + e in MForm, G over the Q digits resp. the QMul digits (keygen.go:94-161)"""
        ks = self.ks
        QQMul = self.Qprod * self.QMulprod
        s_plain = self._apply("invmform", sk)
        out = []
        for moduli in (self.bfv.Q, self.bfv.QMul):
            swk = np.empty((ks.beta_max, ks.m, self.N), dtype=np.uint64)
            for i in range(ks.beta_max):
                qi = moduli[i]
                Gi = QQMul // qi
                Ti = pow(Gi % qi, -1, qi)
                G = (Gi * self.T * Ti * self.Pprod) // self.QMulprod
                t = self._scalar(s_plain, G)
                t = self._bin("add", t, self.gen_gaussian_error())
                swk[i] = self._apply("mform", t)
            out.append(swk)
        return out

    def gen_relin_key_bfv(self, sk, r):
        """(b1, b2, d1, d2, v)  (keygen.go:25-91); CRS[0], CRS[-3], CRS[-1] must exist"""
        ks = self.ks
        a1, a2, u = self.CRS[0], self.CRS[-3], self.CRS[-1]
        b1, b2 = np.empty_like(a1), np.empty_like(a1)
        for i in range(ks.beta_max):
            for a, b in ((a1, b1), (a2, b2)):
                t = self._apply("invmform", self._mul(a[i], sk))
                b[i] = self._apply("mform", self._bin("sub", self.gen_gaussian_error(), t))
        d1, d2 = self.gen_bfv_switching_key(sk)
        for i in range(ks.beta_max):
            d1[i] = self._mul_sub(a1[i], r, d1[i])
            d2[i] = self._mul_sub(a2[i], r, d2[i])
        v = self.gen_switching_key(r)
        for i in range(ks.beta_max):
            v[i] = self._apply("reduce", self._apply("neg", self._mul_add(u[i], sk, v[i])))
        return b1, b2, d1, d2, v

    # ---- messages: polynomials over Z_T with centred coefficients
    def encode(self, m):
        """round(Q/T * m) per coefficient, RNS over Q (bfv ScaleUp)"""
        Q, T = self.Qprod, self.T
        coeffs = [((int(c) % T) * Q + T // 2) // T for c in m]
        return H.int_poly_to_rns(coeffs, self.bfv.Q)

    def decode(self, poly):
        """round(T/Q * x) mod T, centred"""
        c, Q = H.crt_center(poly, self.bfv.Q)
        T = self.T
        out = []
        for v in c:
            m = ((v * T * 2 + Q) // (2 * Q)) % T
            out.append(m - T if m > T // 2 else m)
        return np.array(out, dtype=np.int64)


def negacyclic_mul_mod_t(a, b, T):
    """(a * b mod X^N + 1) mod T, centred; a, b small int64 arrays"""
    N = len(a)
    full = np.convolve(a.astype(np.int64), b.astype(np.int64))
    res = full[:N].copy()
    res[: N - 1] -= full[N:]
    res %= T
    return np.where(res > T // 2, res - T, res)


class BFVScenario:
    """k parties with valid keys over a (small) BFV parameter set."""

    def __init__(self, pset, parties=2, seed=3):
        self.pset = pset
        self.bfv = make_bfv(pset)
        self.kg = BFVKeyGen(self.bfv, seed)
        self.ids = list(range(parties))
        for idx in (0, -1, -3):
            self.kg.add_crs(idx)
        self.sk, self.pk, self.rlk = {}, {}, {}
        for i in self.ids:
            self.sk[i], _ = self.kg.gen_secret_key()
            r, _ = self.kg.gen_secret_key()
            self.pk[i] = self.kg.gen_public_key(self.sk[i])
            self.rlk[i] = self.kg.gen_relin_key_bfv(self.sk[i], r)
        self.u = self.kg.CRS[-1]
        self.level = len(pset["Q"]) - 1
        self.N = 1 << pset["logN"]

    def message(self, lo, hi):
        return self.kg.rng.integers(lo, hi, self.N).astype(np.int64)

    def encrypt(self, m, i):
        return self.kg.encrypt(self.kg.encode(m), self.pk[i], self.level)

    def fresh_ct(self, m, i):
        """ciphertext over the id set {i}: uint64[2][nQ][N]"""
        c0, c1 = self.encrypt(m, i)
        return np.stack([c0, c1])

    def sum_ct(self, msgs):
        """sum over all parties of fresh encryptions: ids = all, uint64[1+k][nQ][N]"""
        ks = self.bfv.ks
        k = len(self.ids)
        ct = np.zeros((1 + k, self.level + 1, self.N), dtype=np.uint64)
        for i in self.ids:
            c0, c1 = self.encrypt(msgs[i], i)
            for j in range(self.level + 1):
                ct[0][j] = ks.ringQ.add(j, ct[0][j], c0[j])
            ct[1 + i] = c1
        return ct

    def decrypt(self, ids, ct):
        vals = {"0": ct[0]}
        for a, i in enumerate(ids):
            vals[i] = ct[1 + a]
        return self.kg.decode(self.kg.decrypt(vals, self.sk))

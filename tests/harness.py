"""Seeded test harness (TEST INFRASTRUCTURE): parameter sets, key generation, encryption,
decryption and CKKS encoding restated from the reference so its property tests can be replayed.

Restates (CPU, through the oracle's ring ops):
  mkrlwe/params.go:16-61        CRS generation (uniform polys, Montgomery form)
  mkrlwe/keygen.go:44-55        genSecretKeyFromSampler
  mkrlwe/keygen.go:88-109       GenPublicKey
  mkrlwe/keygen.go:137-187      GenRelinearizationKey  (b, d, v)
  mkrlwe/keygen.go:190-229      GenRotationKey
  mkrlwe/keygen.go:240-268      GenConjugationKey
  mkrlwe/keygen.go:270-327      GenSwitchingKey (gadget P * CRT idempotent)
  mkrlwe/encryptor.go:55-118    Encrypt (coefficient-domain branch)
  mkrlwe/decryptor.go:26-66     PartialDecrypt / Decrypt
The reference draws unseeded randomness; here everything is seeded.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

# ---------------------------------------------------------------- parameter sets
PN15QP880 = dict(  # mkckks/mkckks_test.go:51-72
    logN=15,
    Q=[0xfffffffff6a0001, 0x3fffffffd60001, 0x3fffffffca0001, 0x3fffffff6d0001, 0x3fffffff5d0001,
       0x3fffffff550001, 0x3fffffff390001, 0x3fffffff360001, 0x3fffffff2a0001, 0x3fffffff000001,
       0x3ffffffefa0001, 0x3ffffffef40001, 0x3ffffffed70001, 0x3ffffffed30001],
    P=[0x7ffffffffe70001, 0x7ffffffffe10001], scale=float(1 << 54))
PN14QP439 = dict(  # mkckks/mkckks_test.go:73-90
    logN=14,
    Q=[0x7ffffffffe70001, 0xffffffff00001, 0xfffffffe40001, 0xfffffffe20001, 0xfffffffbe0001, 0xfffffffa60001],
    P=[0xffffffffffc0001, 0xfffffffff840001], scale=float(1 << 52))
PN16_Q = [0x80000000080001, 0x2000000a0001, 0x2000000e0001, 0x1fffffc20001, 0x200000440001, 0x200000500001,
          0x200000620001, 0x1fffff980001, 0x2000006a0001, 0x1fffff7e0001]   # head of mkrlwe_test.go:22-35
PN16_P = [0x80000000440001, 0x7fffffffba0001, 0x80000000500001, 0x7fffffffaa0001]
PN16QP1761 = dict(  # mkrlwe/mkrlwe_test.go:22-35 (55 + 33 x 45 bit Q, 4 x 55 bit P): BASELINE.json configs[3]
    logN=16,
    Q=PN16_Q + [0x200000860001, 0x200000a60001, 0x200000aa0001, 0x200000b20001, 0x200000c80001, 0x1fffff360001,
                0x200000e20001, 0x1fffff060001, 0x200000fe0001, 0x1ffffede0001, 0x1ffffeca0001, 0x1ffffeb40001,
                0x200001520001, 0x1ffffe760001, 0x2000019a0001, 0x1ffffe640001, 0x200001a00001, 0x1ffffe520001,
                0x200001e80001, 0x1ffffe0c0001, 0x1ffffdee0001, 0x200002480001, 0x1ffffdb60001, 0x200002560001],
    P=PN16_P, scale=float(1 << 45))


def small_ckks(logN, nq=4, scale_bits=54):
    """reduced-size set with the reference's own primes (all are = 1 mod 2^16)."""
    return dict(logN=logN, Q=PN15QP880["Q"][:nq], P=PN15QP880["P"], scale=float(1 << scale_bits))


def small_alpha2(logN, nq=5):
    """4 special primes, gamma = 2 -> alpha = 2 (CRT-reconstruction decomposer path)."""
    return dict(logN=logN, Q=PN16_Q[:nq], P=PN16_P, scale=float(1 << 45))


# ---------------------------------------------------------------- random material
def uniform_poly(rng, moduli, N):
    return np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in moduli])


def uniform_swk(rng, ks):
    """uniform switching key / hoisted-digit shaped array uint64[betaMax][nQ+nP][N]."""
    out = np.empty((ks.beta_max, ks.m, ks.N), dtype=np.uint64)
    for i in range(ks.beta_max):
        out[i] = uniform_poly(rng, ks.Q + ks.P, ks.N)
    return out


def uniform_ct(rng, ks, n, limbs):
    return np.stack([uniform_poly(rng, ks.Q[:limbs], ks.N) for _ in range(1 + n)])


# ---------------------------------------------------------------- key generation (valid keys)
class KeyGen:
    SIGMA = 3.2  # rlwe.DefaultSigma

    def __init__(self, ks, seed=1):
        self.ks = ks
        self.rng = np.random.default_rng(seed)
        self.N = ks.N
        self.QP = ks.Q + ks.P
        self.nq, self.np_ = len(ks.Q), len(ks.P)
        self.CRS = {}
        self.Pprod = 1
        for p in ks.P:
            self.Pprod *= p

    def _ring(self, j):
        return (self.ks.ringQ, j) if j < self.nq else (self.ks.ringP, j - self.nq)

    def _apply(self, fn, poly, *args):
        out = np.empty_like(poly)
        for j in range(poly.shape[0]):
            r, i = self._ring(j)
            out[j] = getattr(r, fn)(i, poly[j], *[a[j] if isinstance(a, np.ndarray) else a for a in args])
        return out

    def small_to_qp(self, s):
        """signed small coefficients -> PolyQP (coefficient domain) -> NTT"""
        poly = np.stack([np.where(s < 0, q - (-s).astype(np.uint64), s.astype(np.uint64)).astype(np.uint64) for q in self.QP])
        return self._apply("ntt", poly)

    def gaussian(self):
        e = np.rint(self.rng.normal(0, self.SIGMA, self.N)).astype(np.int64)
        return np.clip(e, -19, 19)

    def ternary(self):
        return self.rng.choice(np.array([-1, 0, 0, 1], dtype=np.int64), self.N)

    def add_crs(self, idx):
        """uniform, read as already being in Montgomery form (params.go:49-57)"""
        self.CRS[idx] = np.stack([uniform_poly(self.rng, self.QP, self.N) for _ in range(self.ks.beta_max)])
        return self.CRS[idx]

    def gen_secret_key(self):
        """NTT + Montgomery form PolyQP (keygen.go:44-55)"""
        s = self.ternary()
        return self._apply("mform", self.small_to_qp(s)), s

    def gen_gaussian_error(self):
        return self.small_to_qp(self.gaussian())

    def gen_switching_key(self, sk):
        """g*s + e in MForm (keygen.go:270-327)"""
        ks = self.ks
        swk = np.empty((ks.beta_max, ks.m, self.N), dtype=np.uint64)
        ps = np.stack([ks.ringQ.mul_scalar(j, sk[j], self.Pprod % ks.Q[j]) for j in range(self.nq)])
        for i in range(ks.beta_max):
            e = self._apply("mform", self.gen_gaussian_error())
            for j in range(ks.alpha):
                idx = i * ks.alpha + j
                if idx >= self.nq:
                    break
                e[idx] = ks.ringQ.add(idx, e[idx], ps[idx])
            swk[i] = e
        return swk

    def gen_public_key(self, sk):
        """pk0 = -a*s + e (NTT, non-Montgomery), pk1 = a = CRS[0][0] (keygen.go:88-109)"""
        a = self.CRS[0][0]
        pk0 = self.gen_gaussian_error()
        out = np.empty_like(pk0)
        for j in range(self.ks.m):
            r, i = self._ring(j)
            out[j] = r.mul_sub(i, sk[j], a[j], pk0[j])
        return out, a.copy()

    def gen_relin_key(self, sk, r):
        """(b, d, v) (keygen.go:137-187)"""
        ks = self.ks
        a, u = self.CRS[0], self.CRS[-1]
        b = np.empty_like(a)
        for i in range(ks.beta_max):
            t = self._apply("invmform", self._mul(a[i], sk))
            e = self.gen_gaussian_error()
            b[i] = self._apply("mform", self._bin("sub", e, t))
        d = self.gen_switching_key(sk)
        for i in range(ks.beta_max):
            d[i] = self._mul_sub(a[i], r, d[i])
        v = self.gen_switching_key(r)
        for i in range(ks.beta_max):
            v[i] = self._apply("neg", self._mul_add(u[i], sk, v[i]))
            v[i] = self._apply("reduce", v[i])      # Neg maps 0 -> q; canonicalise (same residue class)
        return b, d, v

    def _mul(self, x, y):
        out = np.empty_like(x)
        for j in range(x.shape[0]):
            r, i = self._ring(j)
            out[j] = r.mul(i, x[j], y[j])
        return out

    def _mul_add(self, x, y, z):
        out = np.empty_like(x)
        for j in range(x.shape[0]):
            r, i = self._ring(j)
            out[j] = r.mul_add(i, x[j], y[j], z[j])
        return out

    def _mul_sub(self, x, y, z):
        out = np.empty_like(x)
        for j in range(x.shape[0]):
            r, i = self._ring(j)
            out[j] = r.mul_sub(i, x[j], y[j], z[j])
        return out

    def _bin(self, fn, x, y):
        out = np.empty_like(x)
        for j in range(x.shape[0]):
            r, i = self._ring(j)
            out[j] = getattr(r, fn)(i, x[j], y[j])
        return out

    def automorphism_sk(self, s_small, galEl):
        """sigma_galEl applied to the small secret (coefficient domain, signed)"""
        N = self.N
        out = np.zeros(N, dtype=np.int64)
        idx = (np.arange(N, dtype=np.int64) * galEl)
        pos = idx & (N - 1)
        sgn = (idx >> (N.bit_length() - 1)) & 1
        out[pos] = np.where(sgn == 1, -s_small, s_small)
        return out

    def gen_rotation_key(self, rotidx, sk, s_small):
        """rk = -a * sigma^-1(s) + g*s + e (keygen.go:190-229)"""
        N2 = 2 * self.N
        galEl = pow(5, rotidx, N2)
        galInv = pow(galEl, -1, N2)
        sk_out = self._apply("mform", self.small_to_qp(self.automorphism_sk(s_small, galInv)))
        rk = self.gen_switching_key(sk)
        a = self.CRS[rotidx]
        for i in range(self.ks.beta_max):
            rk[i] = self._mul_sub(a[i], sk_out, rk[i])
        return rk

    def gen_conjugation_key(self, sk, s_small):
        """ck = -a*s + g*sigma(s) + e (keygen.go:240-268)"""
        galEl = 2 * self.N - 1
        sk_out = self._apply("mform", self.small_to_qp(self.automorphism_sk(s_small, galEl)))
        ck = self.gen_switching_key(sk_out)
        a = self.CRS[-2]
        for i in range(self.ks.beta_max):
            ck[i] = self._mul_sub(a[i], sk, ck[i])
        return ck

    # ---- encryption / decryption (coefficient-domain ciphertexts, encryptor.go:95-112)
    def encrypt(self, pt, pk, level):
        """pt: uint64[level+1][N] coefficient domain.  Returns (c0, c1) coefficient domain."""
        ks = self.ks
        u = self.ternary()
        c0 = np.empty((level + 1, self.N), dtype=np.uint64)
        c1 = np.empty_like(c0)
        e0, e1 = self.gaussian(), self.gaussian()
        for j in range(level + 1):
            q = ks.Q[j]
            un = ks.ringQ.mform(j, ks.ringQ.ntt(j, np.where(u < 0, q - 1, u).astype(np.uint64)))
            t0 = ks.ringQ.intt(j, ks.ringQ.mul(j, un, pk[0][j]))
            t1 = ks.ringQ.intt(j, ks.ringQ.mul(j, un, pk[1][j]))
            e0q = np.where(e0 < 0, q - (-e0).astype(np.uint64), e0.astype(np.uint64)).astype(np.uint64)
            e1q = np.where(e1 < 0, q - (-e1).astype(np.uint64), e1.astype(np.uint64)).astype(np.uint64)
            c0[j] = ks.ringQ.add(j, ks.ringQ.add(j, t0, e0q), pt[j])
            c1[j] = ks.ringQ.add(j, t1, e1q)
        return c0, c1

    def decrypt(self, ct_values, sks):
        """ct_values: {"0": poly, id: poly}; sks: {id: sk PolyQP (NTT, MForm)}.  Returns poly (canonical)."""
        ks = self.ks
        c0 = ct_values["0"].copy()
        level = c0.shape[0] - 1
        for id, c in ct_values.items():
            if id == "0":
                continue
            if id not in sks:
                raise KeyError("Cannot Decrypt: there is a missing secretkey")
            for j in range(level + 1):
                t = ks.ringQ.intt(j, ks.ringQ.mul(j, ks.ringQ.ntt(j, c[j]), sks[id][j]))
                c0[j] = ks.ringQ.add(j, c0[j], t)
        return np.stack([ks.ringQ.reduce(j, c0[j]) for j in range(level + 1)])


# ---------------------------------------------------------------- plaintext helpers
def crt_center(poly, moduli):
    """RNS poly [limbs][N] -> list of centred Python ints."""
    Qp = 1
    for q in moduli:
        Qp *= q
    out = []
    for k in range(poly.shape[1]):
        x = 0
        for l, q in enumerate(moduli):
            Mi = Qp // q
            x += int(poly[l][k]) * Mi * pow(Mi, -1, q)
        x %= Qp
        out.append(x - Qp if x > Qp // 2 else x)
    return out, Qp


def log2_inner_sum(poly, moduli):
    """mkrlwe_test.go:92-155 log2OfInnerSum: bit length of sum |coeff| (centred, exact)."""
    c, _ = crt_center(poly, moduli)
    return sum(abs(v) for v in c).bit_length()


def int_poly_to_rns(coeffs, moduli):
    return np.stack([np.array([int(c) % q for c in coeffs], dtype=np.uint64) for q in moduli])


class CKKSEncoder:
    """Canonical-embedding encoder (lattigo ckks.Encoder semantics, client side only):
    slot j <-> evaluation at zeta^(5^j), zeta = exp(i*pi/N)."""

    def __init__(self, logN):
        self.N = 1 << logN
        self.n = self.N // 2
        N = self.N
        rot = np.array([pow(5, j, 2 * N) for j in range(self.n)], dtype=np.int64)
        k = np.arange(N, dtype=np.int64)
        ang = np.pi * ((rot[:, None] * k[None, :]) % (2 * N)) / N
        self.E = np.exp(1j * ang)              # [n][N]: zeta_j^k

    def encode(self, z, scale, moduli):
        z = np.asarray(z, dtype=np.complex128)
        m = (2.0 / self.N) * np.real(np.conj(self.E).T @ z)
        coeffs = [int(round(float(v) * scale)) for v in m]
        return int_poly_to_rns(coeffs, moduli)

    def decode(self, poly, scale, moduli):
        c, _ = crt_center(poly, moduli)
        m = np.array([float(v) for v in c]) / scale
        return self.E @ m

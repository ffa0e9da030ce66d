"""ctypes binding of the CPU ORACLE (oracle/_build/libmkhe_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path (mkhe-kklss_amd/) never imports this module.

The oracle restates the reference Go path (see oracle/ora_mkrlwe.h for file:line citations).
PARITY UNPINNED vs Go: the reference ships no golden vectors and Go is not available.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MKHE_ORACLE_LIB: another build of the same sources (tests/test_sanitizers.py: the ASan + UBSan build, oracle/Makefile SAN=1)
_LIB_PATH = os.environ.get("MKHE_ORACLE_LIB") or os.path.join(_HERE, "_build", "libmkhe_oracle.so")

u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int)
pp = C.POINTER(C.c_void_p)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ora_ring_new.restype = C.c_void_p
        L.ora_ring_new.argtypes = [C.c_int, u64p, C.c_int, u64p]
        L.ora_ring_free.argtypes = [C.c_void_p]
        L.ora_ring_psi.restype = C.c_uint64
        L.ora_ring_psi.argtypes = [C.c_void_p, C.c_int]
        L.ora_ring_qinv.restype = C.c_uint64
        L.ora_ring_qinv.argtypes = [C.c_void_p, C.c_int]
        L.ora_ring_psi_table.restype = u64p
        L.ora_ring_psi_table.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ora_primitive_root.restype = C.c_uint64
        L.ora_primitive_root.argtypes = [C.c_uint64]
        for name in ("ora_ntt", "ora_intt", "ora_intt_lazy", "ora_limb_mform", "ora_limb_invmform",
                     "ora_limb_neg", "ora_limb_reduce"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        for name in ("ora_limb_mul", "ora_limb_mul_add", "ora_limb_mul_sub", "ora_limb_add", "ora_limb_sub"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.ora_limb_mul_scalar.argtypes = [C.c_void_p, C.c_int, u64p, C.c_uint64, u64p]
        L.ora_permute.argtypes = [C.c_void_p, C.c_int, C.c_uint64, u64p, u64p]
        L.ora_div_round_last_many.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p]
        L.ora_fbe_new.restype = C.c_void_p
        L.ora_fbe_new.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_fbe_free.argtypes = [C.c_void_p]
        L.ora_fbe_modup_a2b.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p]
        L.ora_fbe_modup_b2a.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p]
        L.ora_fbe_moddown_ab2a.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p, u64p]
        L.ora_fbe_moddown_ab2b.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p, u64p]
        L.ora_ks_new.restype = C.c_void_p
        L.ora_ks_new.argtypes = [C.c_int, u64p, C.c_int, u64p, C.c_int, C.c_int, u64p, u64p]
        L.ora_ks_free.argtypes = [C.c_void_p]
        L.ora_ks_alpha.argtypes = [C.c_void_p]
        L.ora_ks_beta.argtypes = [C.c_void_p, C.c_int]
        L.ora_ks_swk_words.restype = C.c_size_t
        L.ora_ks_swk_words.argtypes = [C.c_void_p]
        L.ora_ks_ringq.restype = C.c_void_p
        L.ora_ks_ringq.argtypes = [C.c_void_p]
        L.ora_ks_ringp.restype = C.c_void_p
        L.ora_ks_ringp.argtypes = [C.c_void_p]
        L.ora_decompose_and_split.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p, u64p, u64p]
        L.ora_decompose.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p]
        L.ora_external_product.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p, u64p]
        L.ora_external_product_hoisted.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.ora_mul_and_relin.argtypes = [C.c_void_p, C.c_int,
                                        C.c_int, i32p, u64p, C.c_int,
                                        C.c_int, i32p, u64p, C.c_int,
                                        pp, pp, pp, pp, pp, u64p,
                                        C.c_int, i32p, u64p]
        L.ora_mr_xy.argtypes = [C.c_void_p, C.c_int,
                                C.c_int, i32p, u64p, C.c_int, C.c_int, i32p, u64p, C.c_int,
                                pp, pp, pp, pp, u64p, u64p, C.c_int]
        L.ora_mr_finish.argtypes = [C.c_void_p, C.c_int,
                                    C.c_int, i32p, u64p, C.c_int, C.c_int, i32p, u64p, C.c_int,
                                    pp, pp, u64p, u64p, pp, u64p, C.c_int, C.c_int, i32p, u64p]
        L.ora_rotate.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, i32p, u64p, C.c_int, pp, pp, u64p, u64p]
        L.ora_conjugate.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, i32p, u64p, C.c_int, pp, u64p, u64p]
        L.ora_ckks_nb_rescales.restype = C.c_int
        L.ora_ckks_nb_rescales.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double]
        L.ora_bfv_new.restype = C.c_void_p
        L.ora_bfv_new.argtypes = [C.c_int, u64p, u64p, C.c_int, u64p, C.c_int, C.c_int, C.c_uint64]
        L.ora_bfv_free.argtypes = [C.c_void_p]
        L.ora_bfv_ks.restype = C.c_void_p
        L.ora_bfv_ks.argtypes = [C.c_void_p]
        L.ora_bfv_ringqmul.restype = C.c_void_p
        L.ora_bfv_ringqmul.argtypes = [C.c_void_p]
        for name in ("ora_bfv_modup_q_to_r", "ora_bfv_rescale", "ora_bfv_quantize"):
            getattr(L, name).argtypes = [C.c_void_p, u64p, u64p]
        L.ora_bfv_decompose.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.ora_bfv_external_product.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p]
        L.ora_bfv_external_product_hoisted.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, u64p]
        L.ora_bfv_mul_and_relin.argtypes = [C.c_void_p, C.c_int, i32p, u64p, C.c_int, i32p, u64p,
                                            pp, pp, pp, pp, pp, pp, pp, pp, pp, u64p, C.c_int, i32p, u64p]
        L.ora_bfv_mul_relin_new.argtypes = [C.c_void_p, C.c_int, i32p, u64p, C.c_int, i32p, u64p,
                                            pp, pp, pp, pp, pp, u64p, C.c_int, C.c_int, i32p, u64p]
        s32p = C.POINTER(C.c_int32)
        L.ora_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ora_crs_sample.restype = C.c_uint64
        L.ora_crs_sample.argtypes = [C.c_uint64, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.ora_crs_expand.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, u64p]
        for name in ("ora_small_to_qp", "ora_gen_secret_key", "ora_gen_gaussian_error"):
            getattr(L, name).argtypes = [C.c_void_p, s32p, u64p]
        L.ora_gen_switching_key.argtypes = [C.c_void_p, u64p, s32p, u64p]
        L.ora_gen_public_key.argtypes = [C.c_void_p, u64p, s32p, u64p, u64p]
        L.ora_gen_relin_key.argtypes = [C.c_void_p, u64p, u64p, s32p, u64p, u64p, u64p, u64p, u64p]
        L.ora_permute_ntt_qp.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.ora_gen_rotation_key.argtypes = [C.c_void_p, C.c_uint64, u64p, s32p, u64p, u64p]
        L.ora_gen_conjugation_key.argtypes = [C.c_void_p, u64p, s32p, u64p, u64p]
        L.ora_bfv_gen_switching_key.argtypes = [C.c_void_p, u64p, u64p, s32p, u64p]
        L.ora_bfv_gen_relin_key.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, s32p, u64p, u64p, u64p,
                                            u64p, u64p, u64p, u64p, u64p]
        L.ora_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def set_threads(n):
    """limb-level OpenMP threads of the oracle's hot loops (1 = the single-goroutine reference)"""
    lib().ora_set_threads(int(n))


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _u64arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def _i32(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.int32))
    return a, a.ctypes.data_as(i32p)


def _ptrs(arrs, n):
    """array of n void* from dict/list of numpy arrays (None -> NULL)."""
    if arrs is None:
        return None, None
    T = C.c_void_p * n
    keep = []
    vals = []
    for i in range(n):
        a = arrs.get(i) if isinstance(arrs, dict) else (arrs[i] if i < len(arrs) else None)
        if a is None:
            vals.append(None)
        else:
            assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
            keep.append(a)
            vals.append(a.ctypes.data)
    return T(*vals), keep


class Ring:
    """lattigo ring.Ring restated (N, moduli, NTT tables)."""

    def __init__(self, logN, moduli, psi=None, _handle=None, _owner=None):
        self.logN, self.N = logN, 1 << logN
        self.moduli = [int(q) for q in moduli]
        self._owner = _owner
        if _handle is not None:
            self.h = _handle
            self._own = False
        else:
            m = _u64arr(self.moduli)
            ps = _u64arr(psi) if psi is not None else None
            self.h = lib().ora_ring_new(logN, _p(m), len(self.moduli), _p(ps) if ps is not None else None)
            self._own = True
            if not self.h:
                raise ValueError("ora_ring_new failed")

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().ora_ring_free(self.h)
            self.h = None

    def psi(self, i):
        return int(lib().ora_ring_psi(self.h, i))

    def psi_table(self, i, inverse=False):
        ptr = lib().ora_ring_psi_table(self.h, i, 1 if inverse else 0)
        return np.ctypeslib.as_array(ptr, shape=(self.N,)).copy()

    def _un(self, fn, i, a):
        a = _u64arr(a)
        z = np.empty_like(a)
        getattr(lib(), fn)(self.h, i, _p(a), _p(z))
        return z

    def _bin(self, fn, i, a, b):
        a, b = _u64arr(a), _u64arr(b)
        z = np.empty_like(a)
        getattr(lib(), fn)(self.h, i, _p(a), _p(b), _p(z))
        return z

    def ntt(self, i, a): return self._un("ora_ntt", i, a)
    def intt(self, i, a): return self._un("ora_intt", i, a)
    def intt_lazy(self, i, a): return self._un("ora_intt_lazy", i, a)
    def mform(self, i, a): return self._un("ora_limb_mform", i, a)
    def invmform(self, i, a): return self._un("ora_limb_invmform", i, a)
    def neg(self, i, a): return self._un("ora_limb_neg", i, a)
    def reduce(self, i, a): return self._un("ora_limb_reduce", i, a)
    def mul(self, i, a, b): return self._bin("ora_limb_mul", i, a, b)
    def add(self, i, a, b): return self._bin("ora_limb_add", i, a, b)
    def sub(self, i, a, b): return self._bin("ora_limb_sub", i, a, b)

    def mul_add(self, i, a, b, z):
        a, b = _u64arr(a), _u64arr(b)
        z = _u64arr(z).copy()
        lib().ora_limb_mul_add(self.h, i, _p(a), _p(b), _p(z))
        return z

    def mul_sub(self, i, a, b, z):
        a, b = _u64arr(a), _u64arr(b)
        z = _u64arr(z).copy()
        lib().ora_limb_mul_sub(self.h, i, _p(a), _p(b), _p(z))
        return z

    def mul_scalar(self, i, a, s):
        a = _u64arr(a)
        z = np.empty_like(a)
        lib().ora_limb_mul_scalar(self.h, i, _p(a), int(s), _p(z))
        return z

    # poly level ([level+1][N])
    def ntt_poly(self, p):
        return np.stack([self.ntt(i, p[i]) for i in range(p.shape[0])])

    def intt_poly(self, p):
        return np.stack([self.intt(i, p[i]) for i in range(p.shape[0])])

    def permute(self, galEl, p):
        p = _u64arr(p)
        out = np.empty_like(p)
        lib().ora_permute(self.h, p.shape[0] - 1, int(galEl), _p(p), _p(out))
        return out

    def div_round_last_many(self, p, nb):
        """returns (out[level+1-nb][N], mutated_in)"""
        p = _u64arr(p).copy()
        level = p.shape[0] - 1
        out = np.zeros_like(p)
        lib().ora_div_round_last_many(self.h, level, nb, _p(p), _p(out))
        return out[: level + 1 - nb].copy(), p


class BasisExtender:
    """mkrlwe.FastBasisExtender between ring A and ring B."""

    def __init__(self, ra, rb):
        self.ra, self.rb = ra, rb
        self.h = lib().ora_fbe_new(ra.h, rb.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_fbe_free(self.h)
            self.h = None

    def modup_a2b(self, pa, levelB=None):
        pa = _u64arr(pa)
        levelB = self.rb_level(levelB)
        out = np.empty((levelB + 1, self.ra.N), dtype=np.uint64)
        lib().ora_fbe_modup_a2b(self.h, pa.shape[0] - 1, levelB, _p(pa), _p(out))
        return out

    def modup_b2a(self, pb, levelA=None):
        pb = _u64arr(pb)
        levelA = len(self.ra.moduli) - 1 if levelA is None else levelA
        out = np.empty((levelA + 1, self.ra.N), dtype=np.uint64)
        lib().ora_fbe_modup_b2a(self.h, pb.shape[0] - 1, levelA, _p(pb), _p(out))
        return out

    def rb_level(self, l):
        return len(self.rb.moduli) - 1 if l is None else l

    def moddown_ab2a(self, pa, pb):
        pa, pb = _u64arr(pa), _u64arr(pb)
        out = np.empty_like(pa)
        lib().ora_fbe_moddown_ab2a(self.h, pa.shape[0] - 1, pb.shape[0] - 1, _p(pa), _p(pb), _p(out))
        return out

    def moddown_ab2b(self, pa, pb):
        pa, pb = _u64arr(pa), _u64arr(pb)
        out = np.empty_like(pb)
        lib().ora_fbe_moddown_ab2b(self.h, pa.shape[0] - 1, pb.shape[0] - 1, _p(pa), _p(pb), _p(out))
        return out


class KeySwitcher:
    """mkrlwe.KeySwitcher restated.  Switching keys are uint64[betaMax][nQ+nP][N]."""

    def __init__(self, logN, Q, P, gamma=2, psiQ=None, psiP=None, _handle=None, _owner=None):
        self.logN, self.N = logN, 1 << logN
        self.Q, self.P, self.gamma = [int(q) for q in Q], [int(p) for p in P], gamma
        q, p = _u64arr(self.Q), _u64arr(self.P)
        pq = _u64arr(psiQ) if psiQ is not None else None
        ppp = _u64arr(psiP) if psiP is not None else None
        self._owner = _owner            # a BFV object that owns the handle
        if _handle is not None:
            self.h = _handle
        else:
            self.h = lib().ora_ks_new(logN, _p(q), len(self.Q), _p(p), len(self.P), gamma,
                                      _p(pq) if pq is not None else None, _p(ppp) if ppp is not None else None)
        if not self.h:
            raise ValueError("ora_ks_new failed")
        self.alpha = lib().ora_ks_alpha(self.h)
        self.beta_max = lib().ora_ks_beta(self.h, len(self.Q) - 1)
        self.ringQ = Ring(logN, self.Q, _handle=lib().ora_ks_ringq(self.h), _owner=self)
        self.ringP = Ring(logN, self.P, _handle=lib().ora_ks_ringp(self.h), _owner=self)
        self.m = len(self.Q) + len(self.P)

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "_owner", None) is None:
            lib().ora_ks_free(self.h)
        self.h = None

    def beta(self, level):
        return lib().ora_ks_beta(self.h, level)

    def new_swk(self):
        return np.zeros((self.beta_max, self.m, self.N), dtype=np.uint64)

    def decompose_and_split(self, level, digit, a):
        a = _u64arr(a)
        outq = np.zeros((len(self.Q), self.N), dtype=np.uint64)
        outp = np.zeros((len(self.P), self.N), dtype=np.uint64)
        lib().ora_decompose_and_split(self.h, level, len(self.P) - 1, self.alpha, digit, self.gamma,
                                      _p(a), _p(outq), _p(outp))
        return outq[: level + 1], outp

    def decompose(self, level, a, is_ntt=False):
        a = _u64arr(a)
        ad = self.new_swk()
        lib().ora_decompose(self.h, level, 1 if is_ntt else 0, _p(a), _p(ad))
        return ad

    def external_product(self, level, a, bg, is_ntt=False):
        a, bg = _u64arr(a), _u64arr(bg)
        c = np.empty((level + 1, self.N), dtype=np.uint64)
        lib().ora_external_product(self.h, level, 1 if is_ntt else 0, _p(a), _p(bg), _p(c))
        return c

    def external_product_hoisted(self, level, ah, bg):
        ah, bg = _u64arr(ah), _u64arr(bg)
        c = np.empty((level + 1, self.N), dtype=np.uint64)
        lib().ora_external_product_hoisted(self.h, level, _p(ah), _p(bg), _p(c))
        return c

    def mul_and_relin(self, level, ids0, op0, ids1, op1, rlk, crs_u, hoist0=None, hoist1=None):
        """op0/op1: uint64[1+n][limbs][N]; rlk: {id: (b, d, v)}; hoist*: {id: swk} or None.
        Returns (ids_out, out[1+nout][level+1][N])."""
        op0, op1, crs_u = _u64arr(op0), _u64arr(op1), _u64arr(crs_u)
        ids_out = sorted(set(ids0) | set(ids1))
        npar = max(ids_out + [0]) + 1
        out = np.zeros((1 + len(ids_out), level + 1, self.N), dtype=np.uint64)
        a0, p0 = _i32(ids0)
        a1, p1 = _i32(ids1)
        ao, po = _i32(ids_out)
        rb, k1 = _ptrs({i: rlk[i][0] for i in rlk}, npar)
        rd, k2 = _ptrs({i: rlk[i][1] for i in rlk}, npar)
        rv, k3 = _ptrs({i: rlk[i][2] for i in rlk}, npar)
        h0, k4 = _ptrs(hoist0, npar)
        h1, k5 = _ptrs(hoist1, npar)
        lib().ora_mul_and_relin(self.h, level, len(ids0), p0, _p(op0), op0.shape[1],
                                len(ids1), p1, _p(op1), op1.shape[1],
                                h0, h1, rb, rd, rv, _p(crs_u), len(ids_out), po, _p(out))
        return ids_out, out

    def mr_xy(self, level, ids0, op0, ids1, op1, rlk, mform, hoist0=None, hoist1=None):
        """steps A-C: returns (x, y) switching-key shaped arrays; mform=False: canonical partial sums"""
        op0, op1 = _u64arr(op0), _u64arr(op1)
        npar = max(list(ids0) + list(ids1) + [0]) + 1
        x, y = self.new_swk(), self.new_swk()
        a0, p0 = _i32(ids0)
        a1, p1 = _i32(ids1)
        rb, k1 = _ptrs({i: rlk[i][0] for i in rlk}, npar)
        rd, k2 = _ptrs({i: rlk[i][1] for i in rlk}, npar)
        h0, k4 = _ptrs(hoist0, npar)
        h1, k5 = _ptrs(hoist1, npar)
        lib().ora_mr_xy(self.h, level, len(ids0), p0, _p(op0), op0.shape[1], len(ids1), p1, _p(op1), op1.shape[1],
                        h0, h1, rb, rd, _p(x), _p(y), 1 if mform else 0)
        return x, y

    def mr_finish(self, level, ids0, op0, ids1, op1, x, y, rlk, crs_u, with_c0, hoist0=None, hoist1=None):
        """steps D-F with x, y in Montgomery form -> (ids_out, out)"""
        op0, op1, crs_u, x, y = _u64arr(op0), _u64arr(op1), _u64arr(crs_u), _u64arr(x), _u64arr(y)
        ids_out = sorted(set(ids0) | set(ids1))
        npar = max(ids_out + [0]) + 1
        out = np.zeros((1 + len(ids_out), level + 1, self.N), dtype=np.uint64)
        a0, p0 = _i32(ids0)
        a1, p1 = _i32(ids1)
        ao, po = _i32(ids_out)
        rv, k3 = _ptrs({i: rlk[i][2] for i in rlk}, npar)
        h0, k4 = _ptrs(hoist0, npar)
        h1, k5 = _ptrs(hoist1, npar)
        lib().ora_mr_finish(self.h, level, len(ids0), p0, _p(op0), op0.shape[1], len(ids1), p1, _p(op1), op1.shape[1],
                            h0, h1, _p(x), _p(y), rv, _p(crs_u), 1 if with_c0 else 0, len(ids_out), po, _p(out))
        return ids_out, out

    def rotate(self, level, galEl, ids, ct, rk, crs, hoist=None):
        """rk, hoist: lists aligned with ids."""
        ct, crs = _u64arr(ct), _u64arr(crs)
        out = np.zeros((1 + len(ids), level + 1, self.N), dtype=np.uint64)
        a, p = _i32(ids)
        rkp, k1 = _ptrs(list(rk), len(ids))
        hp, k2 = _ptrs(list(hoist) if hoist is not None else None, len(ids))
        lib().ora_rotate(self.h, level, int(galEl), len(ids), p, _p(ct), ct.shape[1], hp, rkp, _p(crs), _p(out))
        return out

    def conjugate(self, level, galEl, ids, ct, ck, crs):
        ct, crs = _u64arr(ct), _u64arr(crs)
        out = np.zeros((1 + len(ids), level + 1, self.N), dtype=np.uint64)
        a, p = _i32(ids)
        ckp, k1 = _ptrs(list(ck), len(ids))
        lib().ora_conjugate(self.h, level, int(galEl), len(ids), p, _p(ct), ct.shape[1], ckp, _p(crs), _p(out))
        return out

    def ckks_nb_rescales(self, level, scale, min_scale):
        s = C.c_double(scale)
        nb = lib().ora_ckks_nb_rescales(self.ringQ.h, level, C.byref(s), float(min_scale))
        return nb, s.value


class BFV:
    """mkbfv Evaluator / KeySwitcher / FastBasisExtender restated (alpha = 1, maximum level).
    rlk of a party = (b1, b2, d1, d2, v): Value[0].Value[0], Value[1].Value[0], Value[0].Value[1],
    Value[1].Value[1], Value[0].Value[2] of mkbfv.RelinearizationKey (keys.go:6-9, keygen.go:41-83)."""

    def __init__(self, logN, Q, QMul, P, T, gamma=2):
        self.logN, self.N = logN, 1 << logN
        self.Q, self.QMul, self.P, self.T = [int(q) for q in Q], [int(q) for q in QMul], [int(p) for p in P], int(T)
        assert len(self.Q) == len(self.QMul)
        q, qm, p = _u64arr(self.Q), _u64arr(self.QMul), _u64arr(self.P)
        self.h = lib().ora_bfv_new(logN, _p(q), _p(qm), len(self.Q), _p(p), len(self.P), gamma, self.T)
        if not self.h:
            raise ValueError("ora_bfv_new failed (alpha must be 1)")
        self.ks = KeySwitcher(logN, self.Q, self.P, gamma, _handle=lib().ora_bfv_ks(self.h), _owner=self)
        self.ringQ, self.ringP = self.ks.ringQ, self.ks.ringP
        self.ringQMul = Ring(logN, self.QMul, _handle=lib().ora_bfv_ringqmul(self.h), _owner=self)
        self.nq = len(self.Q)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_bfv_free(self.h)
            self.h = None

    def _conv(self, fn, a, out_limbs):
        a = _u64arr(a)
        out = np.empty((out_limbs, self.N), dtype=np.uint64)
        getattr(lib(), fn)(self.h, _p(a), _p(out))
        return out

    def modup_q_to_r(self, polyq): return self._conv("ora_bfv_modup_q_to_r", polyq, 2 * self.nq)
    def rescale(self, polyq): return self._conv("ora_bfv_rescale", polyq, 2 * self.nq)
    def quantize(self, polyr_ntt): return self._conv("ora_bfv_quantize", polyr_ntt, self.nq)

    def ntt_r(self, polyr):
        return np.stack([self.ringQ.ntt(j, polyr[j]) if j < self.nq else self.ringQMul.ntt(j - self.nq, polyr[j])
                         for j in range(2 * self.nq)])

    def decompose(self, ar):
        ar = _u64arr(ar)
        ad1, ad2 = self.ks.new_swk(), self.ks.new_swk()
        lib().ora_bfv_decompose(self.h, _p(ar), _p(ad1), _p(ad2))
        return ad1, ad2

    def external_product(self, ar, bg1, bg2):
        ar, bg1, bg2 = _u64arr(ar), _u64arr(bg1), _u64arr(bg2)
        c = np.empty((self.nq, self.N), dtype=np.uint64)
        lib().ora_bfv_external_product(self.h, _p(ar), _p(bg1), _p(bg2), _p(c))
        return c

    def external_product_hoisted(self, ah1, ah2, bg1, bg2):
        ah1, ah2, bg1, bg2 = _u64arr(ah1), _u64arr(ah2), _u64arr(bg1), _u64arr(bg2)
        c = np.empty((self.nq, self.N), dtype=np.uint64)
        lib().ora_bfv_external_product_hoisted(self.h, _p(ah1), _p(ah2), _p(bg1), _p(bg2), _p(c))
        return c

    def _rlk_ptrs(self, rlk, npar):
        out, keep = [], []
        for k in range(5):
            p, kp = _ptrs({i: rlk[i][k] for i in rlk}, npar)
            out.append(p)
            keep.append(kp)
        return out, keep

    def mul_and_relin(self, ids0, op0r, ids1, op1r, rlk, crs_u, hoist0=None, hoist1=None):
        """MulAndRelinBFV[Hoisted] on R-basis operands uint64[1+n][2nQ][N]; hoist*: {id: (ad1, ad2)} or None."""
        op0r, op1r, crs_u = _u64arr(op0r), _u64arr(op1r), _u64arr(crs_u)
        ids_out = sorted(set(ids0) | set(ids1))
        npar = max(ids_out + [0]) + 1
        out = np.zeros((1 + len(ids_out), self.nq, self.N), dtype=np.uint64)
        a0, p0 = _i32(ids0)
        a1, p1 = _i32(ids1)
        ao, po = _i32(ids_out)
        rl, keep = self._rlk_ptrs(rlk, npar)
        hp, hk = [None] * 4, []
        if hoist0 is not None or hoist1 is not None:
            assert hoist0 is not None and hoist1 is not None
            for k, (h, j) in enumerate(((hoist0, 0), (hoist0, 1), (hoist1, 0), (hoist1, 1))):
                hp[k], kk = _ptrs({i: h[i][j] for i in h}, npar)
                hk.append(kk)
        lib().ora_bfv_mul_and_relin(self.h, len(ids0), p0, _p(op0r), len(ids1), p1, _p(op1r),
                                    hp[0], hp[1], hp[2], hp[3], rl[0], rl[1], rl[2], rl[3], rl[4],
                                    _p(crs_u), len(ids_out), po, _p(out))
        return ids_out, out

    def mul_relin_new(self, ids0, op0, ids1, op1, rlk, crs_u, hoisted=True):
        """Evaluator.MulRelinNew on Q-basis ciphertexts uint64[1+n][nQ][N] -> (ids_out, out)."""
        op0, op1, crs_u = _u64arr(op0), _u64arr(op1), _u64arr(crs_u)
        ids_out = sorted(set(ids0) | set(ids1))
        npar = max(ids_out + [0]) + 1
        out = np.zeros((1 + len(ids_out), self.nq, self.N), dtype=np.uint64)
        a0, p0 = _i32(ids0)
        a1, p1 = _i32(ids1)
        ao, po = _i32(ids_out)
        rl, keep = self._rlk_ptrs(rlk, npar)
        lib().ora_bfv_mul_relin_new(self.h, len(ids0), p0, _p(op0), len(ids1), p1, _p(op1),
                                    rl[0], rl[1], rl[2], rl[3], rl[4], _p(crs_u), 1 if hoisted else 0,
                                    len(ids_out), po, _p(out))
        return ids_out, out


def _s32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return list(o)


class KeyGen:
    """mkrlwe.KeyGenerator / mkbfv.KeyGenerator and CRS generation restated with the random samples as inputs
    (oracle/ora_keygen.h).  ks: KeySwitcher (for BFV: BFV(...).ks).  Secret keys are PolyQP uint64[nQ+nP][N]."""

    def __init__(self, ks):
        self.ks = ks

    def _qp(self):
        return np.zeros((self.ks.m, self.ks.N), dtype=np.uint64)

    def crs_sample(self, seed, idx, row, coeff, q):
        return int(lib().ora_crs_sample(seed, idx, row, coeff, q))

    def crs_expand(self, seed, idx):
        out = self.ks.new_swk()
        lib().ora_crs_expand(self.ks.h, seed, idx, _p(out))
        return out

    def small_to_qp(self, s):
        a, p = _s32(s); out = self._qp()
        lib().ora_small_to_qp(self.ks.h, p, _p(out))
        return out

    def gen_secret_key(self, s):
        a, p = _s32(s); out = self._qp()
        lib().ora_gen_secret_key(self.ks.h, p, _p(out))
        return out

    def gen_gaussian_error(self, e):
        a, p = _s32(e); out = self._qp()
        lib().ora_gen_gaussian_error(self.ks.h, p, _p(out))
        return out

    def gen_switching_key(self, sk, e):
        a, p = _s32(e); out = self.ks.new_swk()
        lib().ora_gen_switching_key(self.ks.h, _p(_u64arr(sk)), p, _p(out))
        return out

    def gen_public_key(self, sk, e, crs_a):
        a, p = _s32(e); out = np.zeros((2, self.ks.m, self.ks.N), dtype=np.uint64)
        lib().ora_gen_public_key(self.ks.h, _p(_u64arr(sk)), p, _p(_u64arr(crs_a)), _p(out))
        return out

    def gen_relin_key(self, sk, r, e, crs_a, crs_u):
        a, p = _s32(e); b, d, v = self.ks.new_swk(), self.ks.new_swk(), self.ks.new_swk()
        lib().ora_gen_relin_key(self.ks.h, _p(_u64arr(sk)), _p(_u64arr(r)), p, _p(_u64arr(crs_a)), _p(_u64arr(crs_u)), _p(b), _p(d), _p(v))
        return b, d, v

    def permute_ntt_qp(self, galEl, poly):
        out = self._qp()
        lib().ora_permute_ntt_qp(self.ks.h, galEl, _p(_u64arr(poly)), _p(out))
        return out

    def gen_rotation_key(self, galEl, sk, e, crs):
        a, p = _s32(e); out = self.ks.new_swk()
        lib().ora_gen_rotation_key(self.ks.h, galEl, _p(_u64arr(sk)), p, _p(_u64arr(crs)), _p(out))
        return out

    def gen_conjugation_key(self, sk, e, crs):
        a, p = _s32(e); out = self.ks.new_swk()
        lib().ora_gen_conjugation_key(self.ks.h, _p(_u64arr(sk)), p, _p(_u64arr(crs)), _p(out))
        return out

    def bfv_gen_switching_key(self, sk, g, e):
        a, p = _s32(e); out = self.ks.new_swk()
        lib().ora_bfv_gen_switching_key(self.ks.h, _p(_u64arr(sk)), _p(_u64arr(g)), p, _p(out))
        return out

    def bfv_gen_relin_key(self, sk, r, g1, g2, e, crs_a1, crs_a2, crs_u):
        a, p = _s32(e); o = [self.ks.new_swk() for _ in range(5)]
        lib().ora_bfv_gen_relin_key(self.ks.h, _p(_u64arr(sk)), _p(_u64arr(r)), _p(_u64arr(g1)), _p(_u64arr(g2)), p,
                                    _p(_u64arr(crs_a1)), _p(_u64arr(crs_a2)), _p(_u64arr(crs_u)), *[_p(x) for x in o])
        return tuple(o)

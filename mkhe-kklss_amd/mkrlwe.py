"""Host-side mirror of the reference package `mkrlwe` for the accelerated path.

Same names, argument meaning and error behaviour as the Go types (reference file:line cited per
item); all polynomial data lives in HBM behind C-ABI handles (include/mkhe.h).  The Go reference
panics on misuse; here the same conditions raise `MkheError`/`KeyError` with the reference's text.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._abi import MkheError, check, handle_array, lib

GALOIS_GEN = 5  # lattigo rlwe.GaloisGen


class Parameters:
    """mkrlwe.Parameters (mkrlwe/params.go:8-12): ring parameters + CRS map + gamma, plus the
    engine context (= NewKeySwitcher state, keyswitch.go:33-47)."""

    def __init__(self, logN, Q, P, gamma=2, psiQ=None, psiP=None, device=0):
        self.logN, self._N = int(logN), 1 << int(logN)
        self.Q, self.P, self.gamma = [int(q) for q in Q], [int(p) for p in P], int(gamma)
        self.device = int(device)
        self._psi = (psiQ, psiP)
        self.ctx = self._create_context(psiQ, psiP)
        self.CRS = {}                      # idx -> SwitchingKey   (params.go:37-46)
        self._ids = {}                     # party id string -> dense int for the C ABI

    def _create_context(self, psiQ, psiP):
        q = np.asarray(self.Q, dtype=np.uint64)
        p = np.asarray(self.P, dtype=np.uint64)
        pq = np.asarray(psiQ, dtype=np.uint64) if psiQ is not None else None
        pp = np.asarray(psiP, dtype=np.uint64) if psiP is not None else None
        h = C.c_void_p()
        check(lib().mkhe_ctx_create(C.byref(h), self.logN, q.ctypes.data_as(_abi.u64p), len(self.Q),
                                    p.ctypes.data_as(_abi.u64p), len(self.P), self.gamma,
                                    pq.ctypes.data_as(_abi.u64p) if pq is not None else None,
                                    pp.ctypes.data_as(_abi.u64p) if pp is not None else None, self.device))
        return h

    def close(self):
        """destroys the engine context.  The CRS handles point back at this object (a reference cycle whose finalizers run in
        arbitrary order), so they are released here first; handles that outlive the context are freed by their own
        finalizer through the context-less path of mkhe_*_destroy (plain hipFree)."""
        if getattr(self, "ctx", None):
            if getattr(self, "_parent", None) is None:          # a fork shares its parent's CRS dictionary
                for key in list(getattr(self, "CRS", {}).values()):
                    key.__del__()
                self.CRS.clear()
            lib().mkhe_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # rlwe.Parameters accessors used by the reference
    def N(self): return self._N
    def LogN(self): return self.logN
    def QCount(self): return len(self.Q)
    def PCount(self): return len(self.P)
    def MaxLevel(self): return len(self.Q) - 1
    def Gamma(self): return self.gamma
    def Alpha(self): return self.PCount() // self.gamma                              # params.go:63-65
    def Beta(self, levelQ): return int(math.ceil((levelQ + 1) / self.Alpha()))       # params.go:67-71
    def SwkShape(self): return (self.Beta(self.MaxLevel()), self.QCount() + self.PCount(), self._N)

    def GaloisElementForColumnRotationBy(self, k):
        return pow(GALOIS_GEN, k % (2 * self._N), 2 * self._N)

    def GaloisElementForRowRotation(self):
        return 2 * self._N - 1

    def Psi(self, i):
        return int(lib().mkhe_ctx_psi(self.ctx, i))

    CRS_SEED = 0x4D4B4845          # default public seed of the device-side CRS expansion

    def AddCRS(self, idx, host_swk=None, seed=None):
        """params.AddCRS / NewParameters CRS slots (params.go:37-61,77-99).  With host_swk the uniform polys are
        sampled by the caller (NTT + Montgomery form) and uploaded once; without, CRS[idx] is expanded on the
        device from the public `seed` (mkhe_crs_expand: Philox4x32-10 + mask-and-reject, then MForm) -- every party
        that uses the same seed holds the same CRS and nothing is transferred."""
        if host_swk is not None:
            self.CRS[idx] = SwitchingKey(self, host_swk)
        else:
            self.CRS[idx] = SwitchingKey(self)
            check(lib().mkhe_crs_expand(self.ctx, self.CRS_SEED if seed is None else int(seed), int(idx), self.CRS[idx].h))
        return self.CRS[idx]

    def GenDefaultCRS(self, seed=None):
        """the CRS list NewParameters creates (params.go:37-46): 0, -1 (relin), -2 (conj), -3, -4 (BFV relin) and the
        power-of-two rotations"""
        for idx in [0, -1, -2, -3, -4] + [1 << i for i in range(self.logN - 1)]:
            self.AddCRS(idx, seed=seed)

    def party_index(self, pid):
        if pid == "0":
            raise MkheError("Cannot IDSet Add : 0 cannot be used")              # idset.go:12-16
        if pid not in self._ids:
            self._ids[pid] = len(self._ids)
        return self._ids[pid]

    def sync(self):
        check(lib().mkhe_ctx_sync(self.ctx))

    def Fork(self):
        """A second engine context over the same ring (own stream and scratch pools) that shares this object's CRS map,
        party ids and every key / ciphertext handle.  Operations issued through different forks run concurrently on
        the GPU; order them with wait_for.  (No reference counterpart: the Go evaluator is single-threaded.)"""
        import copy
        f = copy.copy(self)                 # same Q / P lists, same CRS and id dictionaries (shared objects)
        f.ctx = self._create_context(*self._psi)
        f._parent = self                    # the CRS handles belong to the parent's context: keep it alive
        return f

    def Capture(self):
        """context manager recording every engine call issued through this context (and through forks ordered after it and
        joined back) into a HIP graph: `with params.Capture() as g: ...calls...`, then g.launch() replays them with one
        submission.  Objects created inside the block are kept alive by the graph (their buffers are its temporaries)."""
        return Graph(self)

    def wait_for(self, other):
        """work issued through this context from now on starts after everything issued through `other` so far"""
        check(lib().mkhe_ctx_wait_for(self.ctx, other.ctx))

    def stream(self):
        return lib().mkhe_ctx_stream(self.ctx)


class Graph:
    """mkhe_capture_* / mkhe_graph_launch (include/mkhe.h): a captured sequence of engine calls."""

    def __init__(self, params):
        self.params, self.h, self.keep = params, None, []

    def __enter__(self):
        import gc
        # no destructor of an unrelated device object may run inside the capture window (hipFree / stream destruction are
        # device-wide operations): collect pending cyclic garbage now and keep the collector off until the capture ends
        gc.collect()
        self._gc = gc.isenabled()
        gc.disable()
        check(lib().mkhe_capture_begin(self.params.ctx))
        _live_graphs.append(self)
        return self

    def __exit__(self, et, ev, tb):
        import gc
        _live_graphs.remove(self)
        h = C.c_void_p()
        rc = lib().mkhe_capture_end(self.params.ctx, C.byref(h))
        if self._gc:
            gc.enable()
        if et is None:
            check(rc)
            self.h = h
        return False

    def launch(self):
        check(lib().mkhe_graph_launch(self.params.ctx, self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.params.sync()
                lib().mkhe_graph_destroy(self.h)
                self.h = None
        except Exception:
            pass


_live_graphs = []      # captures in progress: device objects created meanwhile are pinned to them (never freed before the graph)


def _pin(obj):
    for g in _live_graphs:
        g.keep.append(obj)


class SwitchingKey:
    """mkrlwe.SwitchingKey (keys.go:23-25): []rlwe.PolyQP of Beta(maxLevel) digits, resident in HBM.
    Host layout uint64[beta][nQ+nP][N]."""

    def __init__(self, params, host=None, zero=True):
        """zero=False: no zero fill (the key is about to be written by an engine call or an upload)"""
        self.params = params
        h = C.c_void_p()
        create = lib().mkhe_swk_create if (zero and host is None) else lib().mkhe_swk_create_uninit
        check(create(params.ctx, C.byref(h)))
        self.h = h
        _pin(self)
        if host is not None:
            self.upload(host)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        if host.shape != self.params.SwkShape():
            raise MkheError("SwitchingKey: expected shape %r, got %r" % (self.params.SwkShape(), host.shape))
        check(lib().mkhe_swk_upload(self.params.ctx, self.h, host.ctypes.data_as(_abi.u64p)))

    def download(self):
        out = np.empty(self.params.SwkShape(), dtype=np.uint64)
        check(lib().mkhe_swk_download(self.params.ctx, self.h, out.ctypes.data_as(_abi.u64p)))
        return out

    def devptr(self):
        return lib().mkhe_swk_devptr(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None) and getattr(self, "_block", None) is None:
                lib().mkhe_swk_destroy(self.params.ctx, self.h)          # ctx None (context already closed): plain hipFree
                self.h = None
        except Exception:
            pass


def NewSwitchingKey(params):
    """keys.go:245-255"""
    return SwitchingKey(params)


class RelinearizationKey:
    """keys.go:34-37: Value = (b, d, v)."""

    def __init__(self, params, id, b=None, d=None, v=None):
        self.ID = id
        self.Value = [SwitchingKey(params, b), SwitchingKey(params, d), SwitchingKey(params, v)]


class RelinearizationKeySet:
    """keys.go:53-57,165-198"""

    def __init__(self, params):
        self.params = params
        self.Value = {}

    def AddRelinearizationKey(self, rlk):
        self.Value[rlk.ID] = rlk

    def DelRelinearizationKey(self, id):
        self.Value.pop(id, None)

    def GetRelinearizationKey(self, id):
        if id not in self.Value:
            raise MkheError("cannot GetRelinearizationKey: there is no relinearization key with given id")
        return self.Value[id]


class RotationKey:
    """keys.go:40-44"""

    def __init__(self, params, rotidx, id, value=None):
        self.ID, self.RotIdx = id, int(rotidx)
        self.Value = SwitchingKey(params, value)


class RotationKeySet:
    """keys.go:60-62,128-162"""

    def __init__(self):
        self.Value = {}

    def AddRotationKey(self, rk):
        self.Value.setdefault(rk.ID, {})[rk.RotIdx] = rk

    def GetRotationKey(self, id, rotidx):
        if id not in self.Value or rotidx not in self.Value[id]:
            raise MkheError("cannot GetRotationKeys: there is no rotation key with given id")
        return self.Value[id][rotidx]


class ConjugationKey:
    """keys.go:47-50"""

    def __init__(self, params, id, value=None):
        self.ID = id
        self.Value = SwitchingKey(params, value)


class ConjugationKeySet:
    """keys.go:65-67,200-228"""

    def __init__(self):
        self.Value = {}

    def AddConjugationKey(self, ck):
        self.Value[ck.ID] = ck

    def GetConjugationKey(self, id):
        if id not in self.Value:
            raise MkheError("cannot GetConjugationKey: there is no conjugation key with given id")
        return self.Value[id]


class HoistedCiphertext:
    """elements.go:5-15: map id -> SwitchingKey holding h(c_id)."""

    def __init__(self):
        self.Value = {}


def NewHoistedCiphertext():
    return HoistedCiphertext()


class Ciphertext:
    """mkrlwe.Ciphertext (elements.go:17-33): Value["0"] plus one poly per party id, all at the same
    level, coefficient domain.  Device layout uint64[1+n][level+1][N]; `ids` fixes the slot order."""

    def __init__(self, params, idset, level, zero=True):
        """zero=False: no zero fill (the ciphertext is about to be the ctOut of an engine call, which writes all of it)"""
        self.params = params
        self.ids = sorted(idset)
        for i in self.ids:
            params.party_index(i)
        self._level = int(level)
        arr = np.asarray([params.party_index(i) for i in self.ids], dtype=np.int32)
        h = C.c_void_p()
        create = lib().mkhe_ct_create if zero else lib().mkhe_ct_create_uninit
        check(create(params.ctx, len(self.ids), arr.ctypes.data_as(_abi.i32p), level + 1, C.byref(h)))
        self.h = h
        _pin(self)

    def IDSet(self):
        return set(self.ids)

    def Level(self):
        return self._level

    def slot(self, id):
        return 0 if id == "0" else 1 + self.ids.index(id)

    def shape(self):
        return (1 + len(self.ids), self._level + 1, self.params.N())

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        if host.shape != self.shape():
            raise MkheError("Ciphertext: expected shape %r, got %r" % (self.shape(), host.shape))
        check(lib().mkhe_ct_upload(self.params.ctx, self.h, host.ctypes.data_as(_abi.u64p)))
        return self

    def download(self):
        out = np.empty(self.shape(), dtype=np.uint64)
        check(lib().mkhe_ct_download(self.params.ctx, self.h, out.ctypes.data_as(_abi.u64p)))
        return out

    def set_values(self, value):
        """value: {"0": poly, id: poly} with polys uint64[level+1][N] (Go: ct.Value[id].Coeffs)."""
        host = np.zeros(self.shape(), dtype=np.uint64)
        for k, v in value.items():
            host[self.slot(k)] = np.asarray(v, dtype=np.uint64)[: self._level + 1]
        return self.upload(host)

    def values(self):
        host = self.download()
        out = {"0": host[0]}
        for a, i in enumerate(self.ids):
            out[i] = host[1 + a]
        return out

    def devptr(self):
        return lib().mkhe_ct_devptr(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None) and getattr(self, "_block", None) is None:
                lib().mkhe_ct_destroy(self.params.ctx, self.h)           # ctx None (context already closed): plain hipFree
                self.h = None
        except Exception:
            pass


def NewCiphertext(params, idset, level):
    """elements.go:22-33"""
    return Ciphertext(params, idset, level)


class _HandleBlock:
    """the handles of one mkhe_ct_create_batch / mkhe_swk_create_batch call: views of one pooled block, destroyed by ONE call when the last Python
    view is gone (every view keeps a reference to this object)"""

    def __init__(self, params, arr, count, destroy):
        self.params, self.arr, self.count, self.destroy = params, arr, count, destroy

    def __del__(self):
        try:
            if self.arr is not None:
                self.destroy(self.params.ctx, self.count, self.arr)
                self.arr = None
        except Exception:
            pass


def batch_ciphertexts(cls, params, idset, level, B, **attrs):
    """B uninitialised ciphertexts of one shape as views of one block (mkhe_ct_create_batch); cls: Ciphertext or a subclass, attrs: extra attributes"""
    ids = sorted(idset)
    arr_ids = np.asarray([params.party_index(i) for i in ids], dtype=np.int32)
    hs = (C.c_void_p * B)()
    check(lib().mkhe_ct_create_batch(params.ctx, B, len(ids), arr_ids.ctypes.data_as(_abi.i32p), level + 1, hs))
    block = _HandleBlock(params, hs, B, lib().mkhe_ct_destroy_batch)
    _pin(block)
    out = []
    for k in range(B):
        c = object.__new__(cls)
        c.params, c.ids, c._level, c.h, c._block = params, ids, int(level), C.c_void_p(hs[k]), block
        for a, v in attrs.items():
            setattr(c, a, v)
        out.append(c)
    return out


def batch_switching_keys(params, count):
    """count uninitialised switching keys / hoisted-digit vectors as views of one block (mkhe_swk_create_batch)"""
    hs = (C.c_void_p * count)()
    check(lib().mkhe_swk_create_batch(params.ctx, count, hs))
    block = _HandleBlock(params, hs, count, lib().mkhe_swk_destroy_batch)
    _pin(block)
    out = []
    for k in range(count):
        s = object.__new__(SwitchingKey)
        s.params, s.h, s._block = params, C.c_void_p(hs[k]), block
        out.append(s)
    return out


class KeySwitcher:
    """mkrlwe.KeySwitcher (keyswitch.go:8-47).  The scratch pools of the reference (ks.Pool,
    swkPool1-3, polyQPool) are engine-internal device buffers owned by the context."""

    def __init__(self, params):
        self.Parameters = params
        self.ctx = params.ctx

    # -- Decompose (keyswitch.go:49-73)
    def Decompose(self, levelQ, ct, id, ad, is_ntt=False):
        check(lib().mkhe_decompose(self.ctx, levelQ, 1 if is_ntt else 0, ct.h, ct.slot(id), ad.h))

    # -- ExternalProduct (keyswitch.go:79-118): c <- ModDown(<h(a), bg>), a = ct.Value[id]
    def ExternalProduct(self, levelQ, ct, id, bg, out, out_id, is_ntt=False):
        check(lib().mkhe_external_product(self.ctx, levelQ, 1 if is_ntt else 0, ct.h, ct.slot(id), bg.h,
                                          out.h, out.slot(out_id)))

    # -- ExternalProductHoisted (keyswitch_hoisted.go:10-40)
    def ExternalProductHoisted(self, levelQ, aHoisted, bg, out, out_id):
        check(lib().mkhe_external_product_hoisted(self.ctx, levelQ, aHoisted.h, bg.h, out.h, out.slot(out_id)))

    # -- MulAndRelin (keyswitch.go:122-230)
    def MulAndRelin(self, op0, op1, rlkSet, ctOut):
        self.MulAndRelinHoisted(op0, op1, None, None, rlkSet, ctOut)

    # -- MulAndRelinHoisted (keyswitch_hoisted.go:44-179)
    def MulAndRelinHoisted(self, op0, op1, op0Hoisted, op1Hoisted, rlkSet, ctOut, rescaled=False):
        """rescaled=True (no reference counterpart at this level; mkckks.Evaluator.mulRelinHoisted, evaluator.go:558-581, is the caller):
        ctOut is one level below the product and receives Rescale(MulAndRelin(..)) in one engine call (mkhe_mul_relin_rescale)"""
        level = ctOut.Level() + (1 if rescaled else 0)
        if op0.Level() < level:
            raise MkheError("Cannot MulAndRelin: op0 and op1 have different levels")        # :48-50
        if op1.Level() < level:
            raise MkheError("Cannot MulAndRelin: op0 and op1 have different levels")
        params = self.Parameters
        if -1 not in params.CRS:
            raise MkheError("mkhe: CRS[-1] (u) has not been uploaded")
        d0 = [rlkSet.GetRelinearizationKey(i).Value[1].h for i in op0.ids]
        v0 = [rlkSet.GetRelinearizationKey(i).Value[2].h for i in op0.ids]
        b1 = [rlkSet.GetRelinearizationKey(i).Value[0].h for i in op1.ids]
        h0 = [op0Hoisted.Value[i].h for i in op0.ids] if op0Hoisted is not None else None
        if op1Hoisted is op0Hoisted and op1 is op0:
            h1 = h0
        else:
            h1 = [op1Hoisted.Value[i].h for i in op1.ids] if op1Hoisted is not None else None
        a_h0 = handle_array(h0)
        a_h1 = a_h0 if h1 is h0 else handle_array(h1)
        fn = lib().mkhe_mul_relin_rescale if rescaled else lib().mkhe_mul_and_relin
        check(fn(self.ctx, op0.h, op1.h, a_h0, a_h1, handle_array(b1), handle_array(d0),
                 handle_array(v0), params.CRS[-1].h, ctOut.h))

    def _rotidx(self, rotidx):
        n2 = self.Parameters.N() // 2
        while rotidx < 0:
            rotidx += n2                                                                     # keyswitch.go:246-249
        return rotidx

    # -- Rotate (keyswitch.go:234-298)
    def Rotate(self, ctIn, rotidx, rkSet, ctOut):
        self.RotateHoisted(ctIn, rotidx, None, rkSet, ctOut)

    # -- RotateHoisted (keyswitch_hoisted.go:183-247)
    def RotateHoisted(self, ctIn, rotidx, ctInHoisted, rkSet, ctOut):
        params = self.Parameters
        if ctIn.Level() < ctOut.Level():
            raise MkheError("Cannot Rotate: ctIn and ctOut have different levels")
        rotidx = self._rotidx(rotidx)
        if rotidx not in params.CRS:
            raise MkheError("mkhe: no CRS for rotation index %d" % rotidx)
        rk = [rkSet.GetRotationKey(i, rotidx).Value.h for i in ctIn.ids]
        hs = [ctInHoisted.Value[i].h for i in ctIn.ids] if ctInHoisted is not None else None
        galEl = params.GaloisElementForColumnRotationBy(rotidx)
        check(lib().mkhe_rotate(self.ctx, galEl, ctIn.h, handle_array(hs), handle_array(rk), params.CRS[rotidx].h, ctOut.h))

    # -- Conjugate (keyswitch.go:302-332)
    def Conjugate(self, ctIn, ckSet, ctOut):
        params = self.Parameters
        if ctIn.Level() < ctOut.Level():
            raise MkheError("Cannot Conjugate: ctIn and ctOut have different levels")
        ck = [ckSet.GetConjugationKey(i).Value.h for i in ctIn.ids]
        check(lib().mkhe_conjugate(self.ctx, params.GaloisElementForRowRotation(), ctIn.h, handle_array(ck),
                                   params.CRS[-2].h, ctOut.h))


def NewKeySwitcher(params):
    return KeySwitcher(params)


# ---- key generation
class SecretKey:
    """mkrlwe.SecretKey (keys.go:9-12): Value = PolyQP (NTT, Montgomery form) resident on the device."""

    def __init__(self, params, id):
        self.ID = id
        self.Value = DeviceLimbs(params, 1, params.QCount() + params.PCount())


class PublicKey:
    """mkrlwe.PublicKey (keys.go:15-18): Value = [2]PolyQP, (-a*s + e, a)."""

    def __init__(self, params, id):
        self.ID = id
        self.Value = DeviceLimbs(params, 2, params.QCount() + params.PCount())


class HostSampler:
    """The small-norm samples of lattigo's ring.TernarySampler / ring.GaussianSampler, drawn on the HOST: secret
    randomness never comes from the GPU.

    Default (no argument): every draw is fed by os.urandom -- the kernel CSPRNG, the counterpart of lattigo's keyed
    blake2b XOF (utils.NewPRNG, keygen.go:26).  Uniform 64-bit words become 53-bit uniforms; ternary values come from
    one uniform each, Gaussians from Box-Muller pairs, rounded, and redrawn while |.| > 6 sigma like lattigo's sampler.

    `rng` (a numpy Generator) replaces the entropy source for REPRODUCIBLE TESTS AND BENCHMARKS ONLY and must be
    acknowledged with insecure_test_only=True: numpy's generators are not cryptographic."""

    def __init__(self, rng=None, sigma=3.2, insecure_test_only=False):
        if rng is not None and not insecure_test_only:
            raise MkheError("HostSampler: a numpy Generator is not a cryptographic source -- pass insecure_test_only=True "
                            "(tests / benchmarks), or no rng at all for os.urandom")
        self.rng = rng
        self.sigma, self.bound = float(sigma), int(6 * float(sigma))        # rlwe.DefaultSigma, keygen.go:36

    def _uniform(self, n):
        """n uniforms in [0, 1) with 53 random bits each"""
        if self.rng is not None:
            return self.rng.random(n)
        import os
        w = np.frombuffer(os.urandom(8 * n), dtype=np.uint64)
        return (w >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def _normal(self, n):
        if self.rng is not None:
            return self.rng.normal(0.0, self.sigma, n)
        m = (n + 1) // 2
        u1, u2 = 1.0 - self._uniform(m), self._uniform(m)                   # u1 in (0, 1]
        r = np.sqrt(-2.0 * np.log(u1)) * self.sigma
        return np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]

    def ternary(self, N, p=0.5):
        """0 with probability p, +-1 with (1-p)/2 each (GenSecretKeyWithDistrib keygen.go:69-76)"""
        u = self._uniform(N)
        return np.where(u < p, 0, np.where(u < p + (1 - p) / 2, 1, -1)).astype(np.int32)

    def gaussian(self, count, N):
        """round(N(0, sigma)), resampled while |.| > bound"""
        e = np.rint(self._normal(count * N)).reshape(count, N)
        bad = np.abs(e) > self.bound
        while bad.any():
            e[bad] = np.rint(self._normal(int(bad.sum())))
            bad = np.abs(e) > self.bound
        return e.astype(np.int32)


def _s32(a, shape):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.shape != tuple(shape):
        raise MkheError("keygen: expected samples of shape %r, got %r" % (tuple(shape), a.shape))
    return a, a.ctypes.data_as(_abi.s32p)


class KeyGenerator:
    """mkrlwe.KeyGenerator (keygen.go:13-40) on the device.  Every Gen* method takes the samples it would draw as an
    optional argument (`s` / `e`, int32) -- given, the result is a deterministic function of them (parity tests);
    omitted, they come from `sampler`."""

    def __init__(self, params, sampler=None):
        self.params = params
        self.sampler = sampler if sampler is not None else HostSampler()

    def _beta(self):
        return self.params.Beta(self.params.MaxLevel())

    def _errors(self, e, count):
        N = self.params.N()
        if e is None:
            e = self.sampler.gaussian(int(np.prod(count)), N)
        return _s32(np.asarray(e).reshape(tuple(np.atleast_1d(count)) + (N,)), tuple(np.atleast_1d(count)) + (N,))

    def GenSecretKey(self, id, s=None):
        """keygen.go:58-60 (ternary, P(0) = 1/2) -> genSecretKeyFromSampler :44-55"""
        return self.GenSecretKeyWithDistrib(0.5, id, s)

    def GenSecretKeyWithDistrib(self, p, id, s=None):
        """keygen.go:69-76"""
        if s is None:
            s = self.sampler.ternary(self.params.N(), p)
        a, ptr = _s32(s, (self.params.N(),))
        sk = SecretKey(self.params, id)
        check(lib().mkhe_keygen_secret(self.params.ctx, ptr, sk.Value.devptr()))
        return sk

    def GenSecretKeyGaussian(self, id, s=None):
        """keygen.go:63-65"""
        if s is None:
            s = self.sampler.gaussian(1, self.params.N())[0]
        return self.GenSecretKeyWithDistrib(0.0, id, s)

    def GenPublicKey(self, sk, e=None):
        """keygen.go:88-109"""
        if 0 not in self.params.CRS:
            raise MkheError("cannot GenPublicKey: CRS[0] is not generated")
        a, ptr = self._errors(e, 1)
        pk = PublicKey(self.params, sk.ID)
        check(lib().mkhe_keygen_public_key(self.params.ctx, sk.Value.devptr(), ptr, self.params.CRS[0].h, pk.Value.devptr()))
        return pk

    def GenKeyPair(self, id):
        """keygen.go:112-115"""
        sk = self.GenSecretKey(id)
        return sk, self.GenPublicKey(sk)

    def GenSwitchingKey(self, skIn, swk, e=None):
        """keygen.go:270-327: swk <- g*skIn + e in MForm"""
        a, ptr = self._errors(e, self._beta())
        check(lib().mkhe_keygen_switching_key(self.params.ctx, skIn.Value.devptr(), ptr, swk.h))

    def GenRelinearizationKey(self, sk, r, e=None):
        """keygen.go:137-187; e: [3][beta][N] for b, d, v"""
        params = self.params
        if params.PCount() == 0:
            raise MkheError("modulus P is empty")
        a, ptr = self._errors(e, (3, self._beta()))
        rlk = RelinearizationKey(params, sk.ID)
        check(lib().mkhe_keygen_relin_key(params.ctx, sk.Value.devptr(), r.Value.devptr(), ptr, params.CRS[0].h, params.CRS[-1].h,
                                          rlk.Value[0].h, rlk.Value[1].h, rlk.Value[2].h))
        return rlk

    def GenRotationKey(self, rotidx, sk, e=None):
        """keygen.go:190-229"""
        params = self.params
        if rotidx not in params.CRS:
            raise MkheError("Cannot GenRotationKey: CRS for given rot idx is not generated")
        crs = params.CRS[rotidx]
        while rotidx < 0:
            rotidx += params.N() // 2
        a, ptr = self._errors(e, self._beta())
        rk = RotationKey(params, rotidx, sk.ID)
        check(lib().mkhe_keygen_rotation_key(params.ctx, params.GaloisElementForColumnRotationBy(rotidx), sk.Value.devptr(), ptr,
                                             crs.h, rk.Value.h))
        return rk

    def GenDefaultRotationKeys(self, sk, rtkSet):
        """keygen.go:232-237"""
        rotidx = 1
        while rotidx < self.params.N() // 2:
            rtkSet.AddRotationKey(self.GenRotationKey(rotidx, sk))
            rotidx *= 2

    def GenConjugationKey(self, sk, e=None):
        """keygen.go:240-268"""
        params = self.params
        a, ptr = self._errors(e, self._beta())
        ck = ConjugationKey(params, sk.ID)
        check(lib().mkhe_keygen_conjugation_key(params.ctx, sk.Value.devptr(), ptr, params.CRS[-2].h, ck.Value.h))
        return ck


def NewKeyGenerator(params, sampler=None):
    return KeyGenerator(params, sampler)


# ---- raw device buffers for ring-level calls (tests / bench of the NTT kernel)
class DeviceLimbs:
    """uint64[count][limbs][N] raw device buffer (mkhe_buf_*)."""

    def __init__(self, params, count, limbs):
        self.params, self.count, self.limbs = params, count, limbs
        self.words = count * limbs * params.N()
        d = C.c_void_p()
        check(lib().mkhe_buf_alloc(params.ctx, self.words, C.byref(d)))
        self.d = d
        _pin(self)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        assert host.shape == (self.count, self.limbs, self.params.N())
        check(lib().mkhe_buf_upload(self.params.ctx, self.d, host.ctypes.data_as(_abi.u64p), self.words))
        return self

    def download(self):
        out = np.empty((self.count, self.limbs, self.params.N()), dtype=np.uint64)
        check(lib().mkhe_buf_download(self.params.ctx, self.d, out.ctypes.data_as(_abi.u64p), self.words))
        return out

    def devptr(self):
        return self.d

    def __del__(self):
        try:
            if getattr(self, "d", None) and self.params.ctx:
                lib().mkhe_buf_free(self.params.ctx, self.d)
                self.d = None
        except Exception:
            pass


def ntt(params, src, dst, mod_base=0, inverse=False, lazy=False):
    """ring.NTTLvl / InvNTTLvl / InvNTTLazyLvl on DeviceLimbs buffers (limb l under modulus mod_base+l)."""
    assert src.count == dst.count and src.limbs == dst.limbs
    check(lib().mkhe_ntt(params.ctx, src.devptr(), dst.devptr(), src.count, src.limbs, mod_base,
                         1 if inverse else 0, 1 if lazy else 0))

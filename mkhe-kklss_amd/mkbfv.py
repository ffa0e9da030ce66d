"""Host-side mirror of the reference package `mkbfv` (Evaluator hot methods).

Same names and argument meaning as mkbfv/{params,keys,elements,evaluator,basis_extension}.go; all
polynomial work runs on the device through the C ABI (include/mkhe.h, mkhe_bfv_* and mkhe_ct_*).
BFV ciphertexts live at the maximum level in the coefficient domain (elements.go:9-11).
"""
import ctypes as C
import math

import numpy as np

from . import _abi, mkrlwe
from ._abi import MkheError, check, handle_array, lib


class Parameters(mkrlwe.Parameters):
    """mkbfv.Parameters (params.go:21-76): mkrlwe parameters over (Q, P) with gamma = 2, plus ring QMul,
    ring R = Q || QMul and the plaintext modulus T."""

    def __init__(self, logN, Q, QMul, P, T, device=0):
        if len(Q) != len(QMul):
            raise MkheError("cannot NewParametersFromLiteral: length of Q & QMul is not equal")     # params.go:30-32
        self.QMul = [int(q) for q in QMul]
        self._T = int(T)
        super().__init__(logN, Q, P, gamma=2, device=device)

    def _create_context(self, psiQ, psiP):
        q = np.asarray(self.Q, dtype=np.uint64)
        qm = np.asarray(self.QMul, dtype=np.uint64)
        p = np.asarray(self.P, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().mkhe_ctx_create_bfv(C.byref(h), self.logN, q.ctypes.data_as(_abi.u64p), qm.ctypes.data_as(_abi.u64p),
                                        len(self.Q), p.ctypes.data_as(_abi.u64p), len(self.P), self.gamma,
                                        self._T, self.device))
        return h

    def T(self): return self._T
    def RCount(self): return 2 * len(self.Q)


class Ciphertext(mkrlwe.Ciphertext):
    """mkbfv.Ciphertext (elements.go:5-11): always at params.MaxLevel()."""

    def __init__(self, params, idset, zero=True):
        super().__init__(params, idset, params.MaxLevel(), zero)


def NewCiphertext(params, idset, zero=True):
    return Ciphertext(params, idset, zero)


class RelinearizationKey:
    """mkbfv.RelinearizationKey (keys.go:6-9,23-31): Value[0], Value[1] = two mkrlwe relinearization keys
    (b1, d1, v) and (b2, d2, -) for the Q resp. QMul gadget (keygen.go:41-83)."""

    def __init__(self, params, id, b1=None, b2=None, d1=None, d2=None, v=None):
        self.ID = id
        self.Value = [mkrlwe.RelinearizationKey(params, id, b1, d1, v), mkrlwe.RelinearizationKey(params, id, b2, d2, None)]


class RelinearizationKeySet:
    """keys.go:11-21,33-82.  PolyRPool / HoistPool of the reference are engine-internal device buffers."""

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


def NewRelinearizationKeyKeySet(params):
    return RelinearizationKeySet(params)


class KeyGenerator(mkrlwe.KeyGenerator):
    """mkbfv.KeyGenerator (mkbfv/keygen.go:9-21): the mkrlwe generator plus the BFV relinearization key."""

    def gadget(self, which):
        """residues [beta][nQ+nP] of the big-integer gadget scalars Gi of GenBFVSwitchingKey (keygen.go:104-116 for the
        Q digits, which = 0; :137-149 for the QMul digits, which = 1)"""
        params = self.params
        Q, QMul, P = math.prod(params.Q), math.prod(params.QMul), math.prod(params.P)
        moduli = params.Q if which == 0 else params.QMul
        alpha, beta = params.Alpha(), params.Beta(params.MaxLevel())
        g = np.zeros((beta, params.QCount() + params.PCount()), dtype=np.uint64)
        for i in range(beta):
            Qi = math.prod(moduli[i * alpha:(i + 1) * alpha])
            Gi = (Q * QMul) // Qi
            Ti = pow(Gi % Qi, -1, Qi)
            Gi = (Gi * params.T() * Ti * P) // QMul
            g[i] = [Gi % m for m in params.Q + params.P]
        return g

    def GenBFVSwitchingKey(self, sk, swk1, swk2, e=None):
        """keygen.go:91-162; e: [2][beta][N]"""
        a, ptr = self._errors(e, (2, self._beta()))
        for which, swk in enumerate((swk1, swk2)):
            g = self.gadget(which)
            check(lib().mkhe_bfv_keygen_switching_key(self.params.ctx, sk.Value.devptr(), g.ctypes.data_as(_abi.u64p),
                                                      a[which].ctypes.data_as(_abi.s32p), swk.h))

    def GenRelinearizationKey(self, sk, r, e=None):
        """mkbfv/keygen.go:24-88; e: [5][beta][N] for b1, b2, d1, d2, v"""
        params = self.params
        a, ptr = self._errors(e, (5, self._beta()))
        rlk = RelinearizationKey(params, sk.ID)
        g1, g2 = self.gadget(0), self.gadget(1)
        V = rlk.Value
        check(lib().mkhe_bfv_keygen_relin_key(params.ctx, sk.Value.devptr(), r.Value.devptr(), g1.ctypes.data_as(_abi.u64p),
                                              g2.ctypes.data_as(_abi.u64p), ptr, params.CRS[0].h, params.CRS[-3].h, params.CRS[-1].h,
                                              V[0].Value[0].h, V[1].Value[0].h, V[0].Value[1].h, V[1].Value[1].h, V[0].Value[2].h))
        return rlk


def NewKeyGenerator(params, sampler=None):
    return KeyGenerator(params, sampler)


class PolyR(mkrlwe.DeviceLimbs):
    """`count` polynomials over ring R (uint64[count][2nQ][N]) resident on the device."""

    def __init__(self, params, count=1):
        super().__init__(params, count, params.RCount())


class FastBasisExtender:
    """mkbfv.FastBasisExtender (basis_extension.go:7-47) on device buffers."""

    def __init__(self, params):
        self.params = params

    def ModUpQtoR(self, polyQ, polyR):
        """basis_extension.go:49-64; polyQ: DeviceLimbs [count][nQ][N], polyR: PolyR"""
        check(lib().mkhe_bfv_modup_q_to_r(self.params.ctx, polyQ.devptr(), polyR.devptr(), polyQ.count))

    def Rescale(self, polyQ, polyR):
        """basis_extension.go:82-96"""
        check(lib().mkhe_bfv_rescale(self.params.ctx, polyQ.devptr(), polyR.devptr(), polyQ.count))

    def Quantize(self, polyR, polyQ, t=None):
        """basis_extension.go:66-80; polyR in the NTT domain; t must be params.T()"""
        if t is not None and int(t) != self.params.T():
            raise MkheError("mkhe: Quantize scalar must be the context's plaintext modulus")
        check(lib().mkhe_bfv_quantize(self.params.ctx, polyR.devptr(), polyQ.devptr(), polyR.count))


class KeySwitcher(mkrlwe.KeySwitcher):
    """mkbfv.KeySwitcher (keyswitch.go:7-65): the mkrlwe key switcher over (Q, P) plus the BFV gadget calls."""

    def DecomposeBFV(self, polyR, ad1, ad2, index=0):
        """keyswitch.go:67-90: polynomial `index` of a PolyR buffer -> (ad1, ad2)"""
        off = index * polyR.limbs * self.Parameters.N() * 8
        check(lib().mkhe_bfv_decompose(self.ctx, C.c_void_p(polyR.devptr().value + off), ad1.h, ad2.h))

    def ExternalProductBFV(self, polyR, bg1, bg2, c, index=0):
        """keyswitch.go:83-113 (non-hoisted: the decomposition happens inside); polynomial `index` of a PolyR buffer;
        c: DeviceLimbs [1][nQ][N]"""
        off = index * polyR.limbs * self.Parameters.N() * 8
        check(lib().mkhe_bfv_external_product(self.ctx, C.c_void_p(polyR.devptr().value + off), bg1.h, bg2.h, c.devptr()))

    def ExternalProductBFVHoisted(self, aHoisted1, aHoisted2, bg1, bg2, c):
        """keyswitch_hoisted.go:6-34; c: DeviceLimbs [1][nQ][N]"""
        check(lib().mkhe_bfv_external_product_hoisted(self.ctx, aHoisted1.h, aHoisted2.h, bg1.h, bg2.h, c.devptr()))


def NewKeySwitcher(params):
    return KeySwitcher(params)


class Evaluator:
    """mkbfv.Evaluator (evaluator.go:7-20)."""

    def __init__(self, params):
        self.params = params
        self.ksw = KeySwitcher(params)
        self.conv = FastBasisExtender(params)

    def newCiphertextBinary(self, op0, op1):
        """evaluator.go:22-25"""
        return NewCiphertext(self.params, op0.IDSet() | op1.IDSet())

    def AddNew(self, op0, op1):
        """evaluator.go:44-52"""
        ctOut = self.newCiphertextBinary(op0, op1)
        check(lib().mkhe_ct_add(self.params.ctx, op0.h, op1.h, ctOut.h))
        return ctOut

    def SubNew(self, op0, op1):
        """evaluator.go:54-76"""
        ctOut = self.newCiphertextBinary(op0, op1)
        check(lib().mkhe_ct_sub(self.params.ctx, op0.h, op1.h, ctOut.h))
        return ctOut

    def mulRelin(self, op0, op1, rlkSet):
        """evaluator.go:95-113 -> KeySwitcher.MulAndRelinBFV (keyswitch.go:115-251): the non-hoisted twin of mulRelinHoisted, on its
        own device path (mkhe_bfv_mul_relin_unhoisted: the reference's order and pool discipline, every party component decomposed
        twice into one pair of pool vectors); the same ciphertext bit for bit."""
        return self.MulRelinNew(op0, op1, rlkSet, hoisted=False)

    def MulRelinNew(self, op0, op1, rlkSet, hoisted=True):
        """evaluator.go:78-82 -> mulRelinHoisted (:118-140)"""
        params = self.params
        if -1 not in params.CRS:
            raise MkheError("mkhe: CRS[-1] (u) has not been uploaded")
        ctOut = NewCiphertext(params, op0.IDSet() | op1.IDSet(), zero=False)        # every limb is written by the engine
        k0 = [rlkSet.GetRelinearizationKey(i) for i in op0.ids]
        k1 = [rlkSet.GetRelinearizationKey(i) for i in op1.ids]
        b1 = [k.Value[0].Value[0].h for k in k1]
        b2 = [k.Value[1].Value[0].h for k in k1]
        d1 = [k.Value[0].Value[1].h for k in k0]
        d2 = [k.Value[1].Value[1].h for k in k0]
        v = [k.Value[0].Value[2].h for k in k0]
        fn = lib().mkhe_bfv_mul_relin if hoisted else lib().mkhe_bfv_mul_relin_unhoisted
        check(fn(params.ctx, op0.h, op1.h, handle_array(b1), handle_array(b2), handle_array(d1),
                 handle_array(d2), handle_array(v), params.CRS[-1].h, ctOut.h))
        return ctOut

    def RotateNew(self, ct0, rotidx, rkSet):
        """evaluator.go:142-180"""
        n2 = self.params.N() // 2
        rotidx %= n2
        ctOut = NewCiphertext(self.params, ct0.IDSet())
        if rotidx == 0:
            check(lib().mkhe_ct_copy(self.params.ctx, ct0.h, ctOut.h))
            return ctOut
        if rotidx in self.params.CRS:
            self.ksw.Rotate(ct0, rotidx, rkSet, ctOut)
            return ctOut
        ctTmp, k = ct0, 1
        while rotidx > 0:
            if rotidx % 2:
                nxt = NewCiphertext(self.params, ct0.IDSet())
                self.ksw.Rotate(ctTmp, k, rkSet, nxt)
                ctTmp = nxt
            rotidx //= 2
            k *= 2
        return ctTmp

    def ConjugateNew(self, ct0, ckSet):
        """evaluator.go:182-192"""
        ctOut = NewCiphertext(self.params, ct0.IDSet())
        self.ksw.Conjugate(ct0, ckSet, ctOut)
        return ctOut


def NewEvaluator(params):
    return Evaluator(params)

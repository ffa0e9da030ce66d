"""Host-side mirror of the reference package `mkckks` (Evaluator hot methods).

Orchestration and float64 scale bookkeeping stay on the host exactly as in
mkckks/evaluator.go:359-443,543-617; all polynomial work runs on the device via mkrlwe.KeySwitcher.
"""
from . import mkrlwe
from ._abi import MkheError, check, lib


class Parameters(mkrlwe.Parameters):
    """mkckks.Parameters (mkckks/params.go:11-24): mkrlwe parameters with gamma = 2 + default scale."""

    def __init__(self, logN, Q, P, scale, logSlots=None, psiQ=None, psiP=None, device=0):
        super().__init__(logN, Q, P, gamma=2, psiQ=psiQ, psiP=psiP, device=device)
        self._scale = float(scale)
        self.logSlots = logN - 1 if logSlots is None else logSlots

    def Scale(self): return self._scale
    def LogSlots(self): return self.logSlots


class Ciphertext(mkrlwe.Ciphertext):
    """mkckks.Ciphertext (mkckks/elements.go:5-17): mkrlwe.Ciphertext + Scale."""

    def __init__(self, params, idset, level, scale):
        super().__init__(params, idset, level)
        self.Scale = float(scale)

    def ScalingFactor(self):
        return self.Scale


def NewCiphertext(params, idset, level, scale):
    return Ciphertext(params, idset, level, scale)


class Evaluator:
    """mkckks.Evaluator (mkckks/evaluator.go:13-39)."""

    def __init__(self, params):
        self.params = params
        self.ksw = mkrlwe.NewKeySwitcher(params)

    def newCiphertextBinary(self, op0, op1):
        """evaluator.go:306-313"""
        return NewCiphertext(self.params, op0.IDSet() | op1.IDSet(), min(op0.Level(), op1.Level()),
                             max(op0.ScalingFactor(), op1.ScalingFactor()))

    # ---- AddNew / SubNew (evaluator.go:316-357 -> evaluateInPlace :200-304)
    def _binary(self, op0, op1, fn):
        s0, s1 = op0.ScalingFactor(), op1.ScalingFactor()
        # the reference first multiplies the operand with the smaller scale by floor(ratio) when that is > 1
        # (MultByConst, :214-281); that branch is not on the accelerated path
        if (s0 > s1 and s0 // s1 > 1) or (s1 > s0 and s1 // s0 > 1):
            raise MkheError("mkhe: Add/Sub of ciphertexts whose scales differ by a factor > 1 is not on the device path")
        ctOut = self.newCiphertextBinary(op0, op1)
        check(fn(self.params.ctx, op0.h, op1.h, ctOut.h))
        return ctOut

    def AddNew(self, op0, op1):
        return self._binary(op0, op1, lib().mkhe_ct_add)

    def SubNew(self, op0, op1):
        return self._binary(op0, op1, lib().mkhe_ct_sub)

    # ---- Rescale (evaluator.go:359-398)
    def nbRescales(self, ctIn, minScale):
        Q = self.params.Q
        scale, nb = ctIn.Scale, 0
        while ctIn.Level() - nb >= 0 and scale / float(Q[ctIn.Level() - nb]) >= minScale / 2:
            scale /= float(Q[ctIn.Level() - nb])
            nb += 1
        return nb, scale

    def RescaleNew(self, ct0, threshold):
        if threshold <= 0:
            raise MkheError("cannot Rescale: minScale is 0")
        if ct0.Scale == 0:
            raise MkheError("cannot Rescale: ciphertext scale is 0")
        if ct0.Level() == 0:
            raise MkheError("cannot Rescale: input Ciphertext already at level 0")
        nb, scale = self.nbRescales(ct0, threshold)
        out = NewCiphertext(self.params, ct0.IDSet(), ct0.Level() - nb, scale)
        check(lib().mkhe_rescale(self.params.ctx, ct0.h, nb, out.h))
        return out

    # ---- HoistedForm (evaluator.go:543-553)
    def HoistedForm(self, ct):
        h = mkrlwe.NewHoistedCiphertext()
        for id in ct.ids:
            h.Value[id] = mkrlwe.NewSwitchingKey(self.params)
            self.ksw.Decompose(ct.Level(), ct, id, h.Value[id])
        return h

    # ---- MulRelinNew (evaluator.go:416-443): hoisting of both operands happens inside the engine
    def MulRelinNew(self, op0, op1, rlkSet):
        return self.MulRelinHoistedNew(op0, op1, None, None, rlkSet)

    # ---- MulRelinHoistedNew / mulRelinHoisted (evaluator.go:558-581)
    def MulRelinHoistedNew(self, op0, op1, op0Hoisted, op1Hoisted, rlkSet):
        ctOut = self.newCiphertextBinary(op0, op1)
        ctOut.Scale = op0.ScalingFactor() * op1.ScalingFactor()
        self.ksw.MulAndRelinHoisted(op0, op1, op0Hoisted, op1Hoisted, rlkSet, ctOut)
        nb, scale = self.nbRescales(ctOut, self.params.Scale())
        if nb == 0 or ctOut.Level() == 0:
            return ctOut
        res = NewCiphertext(self.params, ctOut.IDSet(), ctOut.Level() - nb, scale)
        check(lib().mkhe_rescale(self.params.ctx, ctOut.h, nb, res.h))
        return res

    def _norm_rot(self, rotidx):
        n2 = self.params.N() // 2
        return rotidx % n2

    # ---- RotateNew (evaluator.go:485-525)
    def RotateNew(self, ct0, rotidx, rkSet):
        rotidx = self._norm_rot(rotidx)
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
        if rotidx == 0:
            ctOut.upload(ct0.download())
            return ctOut
        if rotidx in self.params.CRS:
            self.ksw.Rotate(ct0, rotidx, rkSet, ctOut)
            return ctOut
        ctTmp, k = ct0, 1
        while rotidx > 0:                                   # power-of-two decomposition, :516-523
            if rotidx % 2:
                nxt = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
                self.ksw.Rotate(ctTmp, k, rkSet, nxt)
                ctTmp = nxt
            rotidx //= 2
            k *= 2
        return ctTmp

    # ---- RotateHoistedNew (evaluator.go:585-617)
    def RotateHoistedNew(self, ct0, rotidx, ct0Hoisted, rkSet):
        rotidx = self._norm_rot(rotidx)
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
        if rotidx == 0:
            ctOut.upload(ct0.download())
            return ctOut
        if rotidx not in self.params.CRS:
            raise MkheError("Hoisted rotation only works for precomputed rotation keys")
        self.ksw.RotateHoisted(ct0, rotidx, ct0Hoisted, rkSet, ctOut)
        return ctOut

    # ---- ConjugateNew (evaluator.go:527-541)
    def ConjugateNew(self, ct0, ckSet):
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
        self.ksw.Conjugate(ct0, ckSet, ctOut)
        return ctOut


def NewEvaluator(params):
    return Evaluator(params)

"""Host-side mirror of the reference package `mkckks` (Evaluator hot methods).

Orchestration and float64 scale bookkeeping stay on the host exactly as in
mkckks/evaluator.go:359-443,543-617; all polynomial work runs on the device via mkrlwe.KeySwitcher.
"""
import ctypes as C
import math

import numpy as np

from . import _abi, mkrlwe
from ._abi import MkheError, check, handle_array, lib


def scaleUpExact(value, n, q):
    """mkckks/utils.go:59-86: round(|value| * n) mod q through a 53-bit big.Float (= float64 arithmetic), negated
    for value < 0 (q - 0 = q is kept, like the reference)."""
    x = float(-n * value) if value < 0 else float(n * value)
    res = int(x + 0.5) % q
    return q - res if value < 0 else res


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

    def __init__(self, params, idset, level, scale, zero=True):
        super().__init__(params, idset, level, zero)
        self.Scale = float(scale)

    def ScalingFactor(self):
        return self.Scale


def NewCiphertext(params, idset, level, scale, zero=True):
    return Ciphertext(params, idset, level, scale, zero)


class Evaluator:
    """mkckks.Evaluator (mkckks/evaluator.go:13-39)."""

    def __init__(self, params):
        self.params = params
        self.ksw = mkrlwe.NewKeySwitcher(params)
        self.fuse_rescale = True          # MulRelin[Hoisted]New: the single Rescale inside the engine call (False: two calls, for A/B tests)

    def Fork(self):
        """an evaluator on a forked context (mkrlwe.Parameters.Fork): same keys and ciphertexts, its own stream"""
        return Evaluator(self.params.Fork())

    def newCiphertextBinary(self, op0, op1):
        """evaluator.go:306-313"""
        return NewCiphertext(self.params, op0.IDSet() | op1.IDSet(), min(op0.Level(), op1.Level()),
                             max(op0.ScalingFactor(), op1.ScalingFactor()), zero=False)      # every limb is written by the engine call that follows

    # ---- AddNew / SubNew (evaluator.go:316-357 -> evaluateInPlace :200-304)
    def _binary(self, op0, op1, fn):
        s0, s1 = op0.ScalingFactor(), op1.ScalingFactor()
        # scale matching (evaluateInPlace :270-292, the branch for a fresh ctOut): the operand with the smaller scale is
        # first multiplied by floor(ratio) when that is > 1 -- MultByConst with an integer-valued float64, i.e. constant scale 1
        # -- into a pool element; the result carries max(s0, s1) like the reference's (the residual factor is not tracked)
        if s1 > s0 and math.floor(s1 / s0) > 1:
            tmp = NewCiphertext(self.params, op0.ids, op0.Level(), s0, zero=False)
            self.MultByConst(op0, float(math.floor(s1 / s0)), tmp)
            op0 = tmp
        elif s0 > s1 and math.floor(s0 / s1) > 1:
            tmp = NewCiphertext(self.params, op1.ids, op1.Level(), s1, zero=False)
            self.MultByConst(op1, float(math.floor(s0 / s1)), tmp)
            op1 = tmp
        ctOut = self.newCiphertextBinary(op0, op1)
        ctOut.Scale = max(s0, s1)
        check(fn(self.params.ctx, op0.h, op1.h, ctOut.h))
        return ctOut

    def AddNew(self, op0, op1):
        return self._binary(op0, op1, lib().mkhe_ct_add)

    def SubNew(self, op0, op1):
        return self._binary(op0, op1, lib().mkhe_ct_sub)

    # ---- getConstAndScale (evaluator.go:40-94)
    def getConstAndScale(self, level, constant):
        scale = 1.0
        if isinstance(constant, complex):
            cReal, cImag = constant.real, constant.imag
            for c in (cReal, cImag):
                if c != 0 and c - float(int(c)) != 0:
                    scale = float(self.params.Q[level])
        elif isinstance(constant, float):
            cReal, cImag = constant, 0.0
            if cReal != 0 and cReal - float(int(cReal)) != 0:
                scale = float(self.params.Q[level])
        else:
            cReal, cImag = float(int(constant)), 0.0
        return cReal, cImag, scale

    # ---- MultByConst (evaluator.go:117-199): constants per limb on the host, the products on the device
    def MultByConst(self, ct0, constant, ctOut):
        params = self.params
        level = min(ct0.Level(), ctOut.Level())
        cReal, cImag, scale = self.getConstAndScale(level, constant)
        first = np.zeros(ctOut.Level() + 1, dtype=np.uint64)
        second = np.zeros(ctOut.Level() + 1, dtype=np.uint64)
        R = 1 << 64
        for i in range(level + 1):
            qi = params.Q[i]
            sReal = scaleUpExact(cReal, scale, qi) if cReal != 0 else 0
            sConst, sImag = sReal, 0
            if cImag != 0:
                # MRed(scaleUpExact(cImag), NttPsi[i][1]): NttPsi[i][1] = psi^(N/2) * R, so the product is plain
                sImag = (scaleUpExact(cImag, scale, qi) * pow(params.Psi(i), params.N() // 2, qi)) % qi
                sConst = (sConst + sImag) % qi if sConst + sImag >= qi else sConst + sImag            # CRed
            first[i] = (sConst % qi) * R % qi                                                       # MForm
            c2 = sConst
            if cImag != 0:
                c2 = sReal + (qi - sImag)
                c2 = c2 - qi if c2 >= qi else c2                                                     # CRed
            second[i] = (c2 % qi) * R % qi
        check(lib().mkhe_ct_mul_const(params.ctx, ct0.h, first.ctypes.data_as(_abi.u64p), second.ctypes.data_as(_abi.u64p), ctOut.h))
        ctOut.Scale = ct0.Scale * scale

    # ---- DropLevelNew (evaluator.go:96-114): keep the first level+1-levels limbs of every component
    def DropLevelNew(self, ct0, levels):
        out = NewCiphertext(self.params, ct0.IDSet(), ct0.Level() - levels, ct0.Scale, zero=False)
        one = np.array([(1 << 64) % q for q in self.params.Q[: out.Level() + 1]], dtype=np.uint64)      # MForm(1): x * 1
        check(lib().mkhe_ct_mul_const(self.params.ctx, ct0.h, one.ctypes.data_as(_abi.u64p), one.ctypes.data_as(_abi.u64p), out.h))
        return out

    # ---- MulPtxtNew (evaluator.go:465-481); pt: host polynomial uint64[level+1][N] (coefficient domain) + its scale
    def MulPtxtNew(self, ct, pt_value, pt_scale):
        params = self.params
        level = ct.Level()
        ctOut = NewCiphertext(params, ct.IDSet(), level, ct.Scale * float(pt_scale), zero=False)
        if isinstance(pt_value, mkrlwe.DeviceLimbs):          # already resident (uploaded once by the caller; required inside a graph capture)
            pt = pt_value
            if pt.limbs != level + 1:
                raise MkheError("MulPtxtNew: the resident plaintext must have level + 1 limbs")
        else:
            pt = mkrlwe.DeviceLimbs(params, 1, level + 1).upload(np.ascontiguousarray(pt_value, dtype=np.uint64)[None, : level + 1])
        check(lib().mkhe_ct_mul_ptxt(params.ctx, ct.h, pt.devptr(), ctOut.h))
        if ctOut.Level() == 0:                 # eval.Rescale returns an error there; MulPtxtNew ignores it (:480)
            return ctOut
        nb, scale = self.nbRescales(ctOut, params.Scale())
        if nb == 0:
            return ctOut
        res = NewCiphertext(params, ctOut.IDSet(), level - nb, scale, zero=False)
        check(lib().mkhe_rescale(params.ctx, ctOut.h, nb, res.h))
        return res

    # ---- Rescale (evaluator.go:359-398)
    def nbRescales(self, ctIn, minScale):
        return self._nb_rescales(ctIn.Level(), ctIn.Scale, minScale)

    def _nb_rescales(self, level, scale, minScale):
        Q = self.params.Q
        nb = 0
        while level - nb >= 0 and scale / float(Q[level - nb]) >= minScale / 2:
            scale /= float(Q[level - nb])
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
        out = NewCiphertext(self.params, ct0.IDSet(), ct0.Level() - nb, scale, zero=False)
        check(lib().mkhe_rescale(self.params.ctx, ct0.h, nb, out.h))
        return out

    # ---- HoistedForm (evaluator.go:543-553)
    def HoistedForm(self, ct):
        h = mkrlwe.NewHoistedCiphertext()
        for id in ct.ids:
            h.Value[id] = mkrlwe.SwitchingKey(self.params, zero=False)          # every digit the level uses is written below
        check(lib().mkhe_hoisted_form(self.params.ctx, ct.Level(), ct.h, handle_array([h.Value[id].h for id in ct.ids])))   # one batched launch
        return h

    # ---- MulRelinNew (evaluator.go:416-443): hoisting of both operands happens inside the engine
    def MulRelinNew(self, op0, op1, rlkSet):
        return self.MulRelinHoistedNew(op0, op1, None, None, rlkSet)

    # ---- MulRelinHoistedNew / mulRelinHoisted (evaluator.go:558-581)
    def MulRelinHoistedNew(self, op0, op1, op0Hoisted, op1Hoisted, rlkSet):
        level, prod_scale = min(op0.Level(), op1.Level()), op0.ScalingFactor() * op1.ScalingFactor()
        # the number of rescales depends on scales and moduli only (evaluator.go:359-398): the usual single one is folded into the
        # engine call (mkhe_mul_relin_rescale: the DivRoundByLastModulus rides on the last ModDown's store), bit-identical to the two calls
        nb1, scale1 = self._nb_rescales(level, prod_scale, self.params.Scale())
        if nb1 == 1 and level >= 1 and self.fuse_rescale:
            res = NewCiphertext(self.params, op0.IDSet() | op1.IDSet(), level - 1, scale1, zero=False)
            self.ksw.MulAndRelinHoisted(op0, op1, op0Hoisted, op1Hoisted, rlkSet, res, rescaled=True)
            return res
        ctOut = NewCiphertext(self.params, op0.IDSet() | op1.IDSet(), level, prod_scale, zero=False)      # newCiphertextBinary, :306-313
        self.ksw.MulAndRelinHoisted(op0, op1, op0Hoisted, op1Hoisted, rlkSet, ctOut)
        nb, scale = self.nbRescales(ctOut, self.params.Scale())
        if nb == 0 or ctOut.Level() == 0:
            return ctOut
        res = NewCiphertext(self.params, ctOut.IDSet(), ctOut.Level() - nb, scale, zero=False)
        check(lib().mkhe_rescale(self.params.ctx, ctOut.h, nb, res.h))
        return res

    def _norm_rot(self, rotidx):
        n2 = self.params.N() // 2
        return rotidx % n2

    # ---- RotateNew (evaluator.go:485-525)
    def RotateNew(self, ct0, rotidx, rkSet):
        rotidx = self._norm_rot(rotidx)
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
        if rotidx == 0:
            check(lib().mkhe_ct_copy(self.params.ctx, ct0.h, ctOut.h))
            return ctOut
        if rotidx in self.params.CRS:
            self.ksw.Rotate(ct0, rotidx, rkSet, ctOut)
            return ctOut
        ctTmp, k = ct0, 1
        while rotidx > 0:                                   # power-of-two decomposition, :516-523
            if rotidx % 2:
                nxt = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
                self.ksw.Rotate(ctTmp, k, rkSet, nxt)
                ctTmp = nxt
            rotidx //= 2
            k *= 2
        return ctTmp

    # ---- RotateHoistedNew (evaluator.go:585-617)
    def RotateHoistedNew(self, ct0, rotidx, ct0Hoisted, rkSet):
        rotidx = self._norm_rot(rotidx)
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
        if rotidx == 0:
            check(lib().mkhe_ct_copy(self.params.ctx, ct0.h, ctOut.h))
            return ctOut
        if rotidx not in self.params.CRS:
            raise MkheError("Hoisted rotation only works for precomputed rotation keys")
        self.ksw.RotateHoisted(ct0, rotidx, ct0Hoisted, rkSet, ctOut)
        return ctOut

    # ---- ConjugateNew (evaluator.go:527-541)
    def ConjugateNew(self, ct0, ckSet):
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
        self.ksw.Conjugate(ct0, ckSet, ctOut)
        return ctOut

    # ---- AddNew(ct0, RotateNew(ct0, rotidx, rkSet)): the step every log-sum of cnn/cnn.go repeats (:33-37,64-67,83-86,90-93) as ONE engine call --
    # the ring.Add rides on the store of the rotation's ModDown (mkhe_rotate_multi with post_add).  Same integers as the two calls.
    def RotateAndAddNew(self, ct0, rotidx, rkSet):
        rotidx = self._norm_rot(rotidx)
        if rotidx == 0:
            return self.AddNew(ct0, self.RotateNew(ct0, 0, rkSet))
        ctTmp, k, steps = ct0, 1, []
        if rotidx in self.params.CRS:
            steps = [rotidx]
        else:
            r = rotidx
            while r > 0:                                    # power-of-two decomposition, :516-523: the addition joins the LAST rotation
                if r % 2:
                    steps.append(k)
                r //= 2
                k *= 2
        for s in steps[:-1]:
            nxt = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
            self.ksw.Rotate(ctTmp, s, rkSet, nxt)
            ctTmp = nxt
        last = steps[-1]
        if last not in self.params.CRS:
            raise MkheError("mkhe: no CRS for rotation index %d" % last)
        ctOut = NewCiphertext(self.params, ct0.IDSet(), ct0.Level(), ct0.Scale, zero=False)
        rk = [rkSet.GetRotationKey(i, last).Value.h for i in ct0.ids]
        gal = (C.c_uint64 * 1)(self.params.GaloisElementForColumnRotationBy(last))
        check(lib().mkhe_rotate_multi(self.params.ctx, 1, gal, handle_array([ctTmp.h]), None, handle_array(rk), handle_array([self.params.CRS[last].h]),
                                      handle_array([ct0.h]), handle_array([ctOut.h])))
        return ctOut

    # ---- out = cts[0]; for c in cts[1:]: out = AddNew(out, c) -- the sum over the products of a layer (cnn/cnn.go:19-30,58-62) -- as ONE launch
    # when the summands have one shape and one scale (then every AddNew is a plain ring.Add per component); the chain itself otherwise.
    def SumNew(self, cts):
        cts = list(cts)
        c0 = cts[0]
        if len(cts) == 1 or any(c.ids != c0.ids or c.Level() != c0.Level() or c.Scale != c0.Scale for c in cts):
            out = c0
            for c in cts[1:]:
                out = self.AddNew(out, c)
            return out
        out = NewCiphertext(self.params, c0.IDSet(), c0.Level(), c0.Scale, zero=False)
        check(lib().mkhe_ct_sum(self.params.ctx, len(cts), handle_array([c.h for c in cts]), out.h))
        return out

    def Lanes(self, n):
        """n independent operations of one shape on THIS context as one launch set (a BatchEvaluator over the same stream): how the device runs the
        independent chains of cnn.Convolution / FC1Layer -- small kernels do not overlap each other on this chip, lanes make them one kernel"""
        hit = self.__dict__.setdefault("_lanes", {}).get(n)
        if hit is None:
            hit = self._lanes[n] = BatchEvaluator(self.params, n, ev=self)
        return hit


def NewEvaluator(params):
    return Evaluator(params)


# ---------------------------------------------------------------- B inputs in lock step (round 4; include/mkhe.h "B independent operations")
class BatchCiphertext:
    """B ciphertexts of one shape (same ids, level, scale): what a BatchEvaluator consumes and returns.  cts[b] are ordinary Ciphertexts."""

    def __init__(self, cts):
        self.cts = list(cts)
        c0 = self.cts[0]
        for c in self.cts:
            if c.ids != c0.ids or c.Level() != c0.Level() or c.Scale != c0.Scale:
                raise MkheError("BatchCiphertext: the ciphertexts of a batch must have one shape and scale")
        self.ids, self.Scale, self.params = c0.ids, c0.Scale, c0.params
        self._harr = handle_array([c.h for c in self.cts])          # (built once: every batched call passes it)

    def __len__(self): return len(self.cts)
    def IDSet(self): return set(self.ids)
    def Level(self): return self.cts[0].Level()
    def ScalingFactor(self): return self.Scale
    def download(self): return np.stack([c.download() for c in self.cts])


class BatchHoisted:
    """per input the hoisted forms of its party components (mkrlwe.HoistedCiphertext each)"""

    def __init__(self, hoisted):
        self.hoisted = list(hoisted)


class BatchEvaluator:
    """The mkckks.Evaluator surface on B inputs at a time: every method takes BatchCiphertexts (or, for an operand that is the same for every
    input -- the model of cnn -- a plain Ciphertext / HoistedCiphertext, which is broadcast) and issues ONE launch set for the B operations
    (mkhe_*_batch).  mkhe_kklss_amd.cnn runs on it unchanged.  Scale bookkeeping is the single-input evaluator's (one shape, one scale)."""

    def __init__(self, params, B, ev=None):
        self.params, self.B = params, int(B)
        self.ev = ev if ev is not None else Evaluator(params)
        self._bcast = {}                       # broadcast operands (the model ciphertexts of cnn): their (void*)[B], keyed by object

    def Fork(self):
        """a batch evaluator on a forked context (own stream, shared keys and ciphertexts): independent chains of a circuit overlap on the GPU"""
        return BatchEvaluator(self.params.Fork(), self.B)

    def Lanes(self, n):
        """n independent operations PER INPUT as one launch set: a BatchEvaluator of B * n items on the same context, lane-major (item j * B + b = lane j
        of input b).  The independent chains of cnn.Convolution / FC1Layer on B images: one rotation / hoisting / MulRelin launch set for all of them."""
        hit = self.__dict__.setdefault("_lanes", {}).get(n)
        if hit is None:
            hit = self._lanes[n] = BatchEvaluator(self.params, self.B * n, ev=self.ev)
        return hit

    def SumNew(self, cts):
        """out = cts[0]; for c in cts[1:]: out = AddNew(out, c), on batches (one batched Add per summand)"""
        out = cts[0]
        for c in cts[1:]:
            out = self.AddNew(out, c)
        return out

    # -- helpers
    def _cts(self, op):
        return op.cts if isinstance(op, BatchCiphertext) else [op] * self.B

    def _new(self, like_ids, level, scale):
        # (one block and one create / destroy call for the B outputs instead of B of each)
        return BatchCiphertext(mkrlwe.batch_ciphertexts(Ciphertext, self.params, like_ids, level, self.B, Scale=float(scale)))

    def _h(self, op):
        """the (void*)[B] of an operand: cached on a BatchCiphertext, built (and cached per evaluator) for a broadcast ciphertext"""
        if isinstance(op, BatchCiphertext):
            return op._harr
        if isinstance(op, list):
            return handle_array([c.h for c in op])
        key = id(op)
        hit = self._bcast.get(key)
        if hit is None or hit[0] is not op:
            if len(self._bcast) > 256:
                self._bcast.clear()
            hit = self._bcast[key] = (op, handle_array([op.h] * self.B))
        return hit[1]

    def _hoists(self, hoisted, ops):
        """flat [b * n + a] handle list, or None"""
        if hoisted is None:
            return None
        ids = tuple(ops[0].ids)
        cache = hoisted.__dict__.setdefault("_flat", {})
        arr = cache.get((ids, self.B))
        if arr is None:
            hs = hoisted.hoisted if isinstance(hoisted, BatchHoisted) else [hoisted] * self.B
            arr = cache[(ids, self.B)] = handle_array([hs[b].Value[i].h for b in range(self.B) for i in ids])
        return arr

    # -- HoistedForm (evaluator.go:543-553)
    def HoistedForm(self, ct):
        if not isinstance(ct, BatchCiphertext):
            return self.ev.HoistedForm(ct)
        if not ct.ids:                                            # no party component: nothing to hoist
            return BatchHoisted([mkrlwe.NewHoistedCiphertext() for _ in ct.cts])
        hs, keys, k = [], mkrlwe.batch_switching_keys(self.params, self.B * len(ct.ids)), 0
        for c in ct.cts:
            h = mkrlwe.NewHoistedCiphertext()
            for id in c.ids:
                h.Value[id] = keys[k]; k += 1
            hs.append(h)
        check(lib().mkhe_hoisted_form_batch(self.params.ctx, ct.Level(), self.B, self._h(ct),
                                            handle_array([hs[b].Value[id].h for b in range(self.B) for id in ct.cts[b].ids])))
        return BatchHoisted(hs)

    # -- AddNew / SubNew (evaluator.go:316-357)
    def _binary(self, op0, op1, opcode):
        a, b = self._cts(op0), self._cts(op1)
        s0, s1 = a[0].ScalingFactor(), b[0].ScalingFactor()
        if (s1 > s0 and math.floor(s1 / s0) > 1) or (s0 > s1 and math.floor(s0 / s1) > 1):
            # scale matching multiplies one operand by a constant first (evaluateInPlace :270-292): rare in the circuits this class serves -- per input
            fn = self.ev.AddNew if opcode == 0 else self.ev.SubNew
            return BatchCiphertext([fn(a[k], b[k]) for k in range(self.B)])
        out = self._new(a[0].IDSet() | b[0].IDSet(), min(a[0].Level(), b[0].Level()), max(s0, s1))
        check(lib().mkhe_ct_binary_batch(self.params.ctx, opcode, self.B, self._h(op0), self._h(op1), self._h(out)))
        return out

    def AddNew(self, op0, op1): return self._binary(op0, op1, 0)
    def SubNew(self, op0, op1): return self._binary(op0, op1, 1)

    # -- MulRelin[Hoisted]New (evaluator.go:416-443,558-581)
    def MulRelinNew(self, op0, op1, rlkSet):
        return self.MulRelinHoistedNew(op0, op1, None, None, rlkSet)

    def MulRelinHoistedNew(self, op0, op1, op0Hoisted, op1Hoisted, rlkSet):
        a, b = self._cts(op0), self._cts(op1)
        params = self.params
        level, prod_scale = min(a[0].Level(), b[0].Level()), a[0].ScalingFactor() * b[0].ScalingFactor()
        if -1 not in params.CRS:
            raise MkheError("mkhe: CRS[-1] (u) has not been uploaded")
        nb1, scale1 = self.ev._nb_rescales(level, prod_scale, params.Scale())
        # the single evaluator's branch (Evaluator.MulRelinHoistedNew): the Rescale is folded into the engine call only when it is the usual single
        # one and fuse_rescale is set; otherwise the product is formed at its level and ONE mkhe_rescale(nb) per input follows, as there
        rescale = nb1 == 1 and level >= 1 and self.ev.fuse_rescale
        ids = a[0].IDSet() | b[0].IDSet()
        out = self._new(ids, level - 1 if rescale else level, prod_scale / float(params.Q[level]) if rescale else prod_scale)
        d0 = [rlkSet.GetRelinearizationKey(i).Value[1].h for i in a[0].ids]
        v0 = [rlkSet.GetRelinearizationKey(i).Value[2].h for i in a[0].ids]
        b1 = [rlkSet.GetRelinearizationKey(i).Value[0].h for i in b[0].ids]
        check(lib().mkhe_mul_relin_batch(params.ctx, self.B, self._h(op0), self._h(op1), self._hoists(op0Hoisted, a), self._hoists(op1Hoisted, b),
                                         handle_array(b1), handle_array(d0), handle_array(v0), params.CRS[-1].h, 1 if rescale else 0, self._h(out)))
        if rescale:
            return out
        nb, scale = self.ev.nbRescales(out.cts[0], params.Scale())
        if nb == 0 or out.cts[0].Level() == 0:
            return out
        res = self._new(ids, out.cts[0].Level() - nb, scale)
        for c, r in zip(out.cts, res.cts):              # (nb != 1 or fuse_rescale off: not in the circuits of the reference -- per input)
            check(lib().mkhe_rescale(params.ctx, c.h, nb, r.h))
        return res

    # -- RotateNew / RotateHoistedNew (evaluator.go:485-525,585-617)
    def _rotate(self, ct, rotidx, hoisted, rkSet):
        params = self.params
        cts = self._cts(ct)
        out = self._new(cts[0].IDSet(), cts[0].Level(), cts[0].Scale)
        rk = [rkSet.GetRotationKey(i, rotidx).Value.h for i in cts[0].ids]
        check(lib().mkhe_rotate_batch(params.ctx, params.GaloisElementForColumnRotationBy(rotidx), self.B, self._h(ct), self._hoists(hoisted, cts),
                                      handle_array(rk), params.CRS[rotidx].h, self._h(out)))
        return out

    def RotateNew(self, ct, rotidx, rkSet):
        rotidx = self.ev._norm_rot(rotidx)
        if rotidx == 0:
            return BatchCiphertext([self.ev.RotateNew(c, 0, rkSet) for c in self._cts(ct)])
        if rotidx in self.params.CRS:
            return self._rotate(ct, rotidx, None, rkSet)
        tmp, k = ct, 1
        while rotidx > 0:                                   # power-of-two decomposition, :516-523
            if rotidx % 2:
                tmp = self._rotate(tmp, k, None, rkSet)
            rotidx //= 2
            k *= 2
        return tmp

    def RotateHoistedNew(self, ct, rotidx, ctHoisted, rkSet):
        if isinstance(rotidx, (list, tuple)):
            return self._rotate_multi(ct, [self.ev._norm_rot(r) for r in rotidx], ctHoisted, rkSet, None)
        rotidx = self.ev._norm_rot(rotidx)
        if rotidx == 0:
            return BatchCiphertext([self.ev.RotateNew(c, 0, rkSet) for c in self._cts(ct)])
        if rotidx not in self.params.CRS:
            raise MkheError("Hoisted rotation only works for precomputed rotation keys")
        return self._rotate(ct, rotidx, ctHoisted, rkSet)

    def _rotate_multi(self, ct, rots, hoisted, rkSet, post):
        """input b rotated by rots[b] (every index non-zero and with a CRS), each with its own keys: one launch set (mkhe_rotate_multi);
        post: BatchCiphertext / Ciphertext added to the rotated ciphertexts on the store, or None"""
        params = self.params
        cts = self._cts(ct)
        if len(rots) != self.B:
            raise MkheError("BatchEvaluator: one rotation index per input")
        for r in rots:
            if r == 0 or r not in params.CRS:
                raise MkheError("Hoisted rotation only works for precomputed rotation keys")
        out = self._new(cts[0].IDSet(), cts[0].Level(), cts[0].Scale)
        rk = [rkSet.GetRotationKey(i, r).Value.h for r in rots for i in cts[0].ids]
        gal = (C.c_uint64 * self.B)(*[params.GaloisElementForColumnRotationBy(r) for r in rots])
        check(lib().mkhe_rotate_multi(params.ctx, self.B, gal, self._h(ct), self._hoists(hoisted, cts), handle_array(rk),
                                      handle_array([params.CRS[r].h for r in rots]), self._h(post) if post is not None else None, self._h(out)))
        return out

    def RotateAndAddNew(self, ct, rotidx, rkSet):
        """AddNew(ct, RotateNew(ct, rotidx, rkSet)) for every input, the addition on the store of the rotation (Evaluator.RotateAndAddNew)"""
        rotidx = self.ev._norm_rot(rotidx)
        if rotidx == 0 or rotidx not in self.params.CRS:
            return self.AddNew(ct, self.RotateNew(ct, rotidx, rkSet))
        return self._rotate_multi(ct, [rotidx] * self.B, None, rkSet, ct)

    # -- MulPtxtNew (evaluator.go:465-481)
    def MulPtxtNew(self, ct, pt_value, pt_scale):
        params = self.params
        cts = self._cts(ct)
        level, scale = cts[0].Level(), cts[0].Scale * float(pt_scale)
        if isinstance(pt_value, mkrlwe.DeviceLimbs):
            pt = pt_value
            if pt.limbs != level + 1:
                raise MkheError("MulPtxtNew: the resident plaintext must have level + 1 limbs")
        else:
            pt = mkrlwe.DeviceLimbs(params, 1, level + 1).upload(np.ascontiguousarray(pt_value, dtype=np.uint64)[None, : level + 1])
        nb, rscale = (0, scale) if level == 0 else self.ev._nb_rescales(level, scale, params.Scale())
        out = self._new(cts[0].IDSet(), level - nb, rscale if nb else scale)
        check(lib().mkhe_ct_mul_ptxt_batch(params.ctx, self.B, self._h(ct), pt.devptr(), nb, self._h(out)))
        return out

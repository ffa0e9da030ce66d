"""TEST INFRASTRUCTURE: the mkckks.Evaluator surface that cnn/cnn.go uses, on HOST ciphertexts with the CPU oracle doing all polynomial work.

`mkhe_kklss_amd.cnn.Convolution / FC1Layer / FC2Layer / Inference` are duck-typed over their evaluator, so the same circuit runs on this class:
that is the CPU baseline of the cnn bench line (bench.py --scheme cnn: one inference on one host thread, then limb-parallel) and the
bit-for-bit check of the device inference against the oracle.  Only tests/ and bench.py's cpu_baseline leg import this file.

Restates the host-side logic of mkckks/evaluator.go (scale matching :270-292, nbRescales :376-384, power-of-two rotations :516-523) like
mkhe_kklss_amd/mkckks.py does; the polynomial work is oracle/ora_*.c."""
import math

import numpy as np


class OCt:
    """host ciphertext: ids (sorted), uint64[1 + n][level + 1][N], Scale"""

    def __init__(self, ids, host, scale):
        self.ids, self.host, self.Scale = sorted(ids), np.ascontiguousarray(host, dtype=np.uint64), float(scale)

    def Level(self): return self.host.shape[1] - 1
    def IDSet(self): return set(self.ids)
    def ScalingFactor(self): return self.Scale
    def download(self): return self.host


class OHoisted:
    def __init__(self):
        self.Value = {}


class OracleEvaluator:
    def __init__(self, ks, Q, default_scale, rlk_host, rk_host, crs_host, logN):
        """rlk_host: {id: (b, d, v)}; rk_host: {(id, rotidx): swk}; crs_host: {rotidx: swk} (with -1 = u); all uint64[beta][m][N]"""
        self.ks, self.Q, self.scale0, self.rlk, self.rk, self.crs, self.N = ks, list(Q), float(default_scale), rlk_host, rk_host, crs_host, 1 << logN
        self.params = None

    # ---- host bookkeeping (mkckks/evaluator.go:376-384)
    def _nb_rescales(self, level, scale, min_scale):
        nb = 0
        while level - nb >= 0 and scale / float(self.Q[level - nb]) >= min_scale / 2:
            scale /= float(self.Q[level - nb])
            nb += 1
        return nb, scale

    def _rescale(self, host, scale):
        level = host.shape[1] - 1
        if level == 0:
            return host, scale
        nb, s = self._nb_rescales(level, scale, self.scale0)
        if nb == 0:
            return host, scale
        return np.stack([self.ks.ringQ.div_round_last_many(host[k], nb)[0] for k in range(host.shape[0])]), s

    def _galois(self, rotidx):
        return pow(5, rotidx, 2 * self.N)

    # ---- AddNew (evaluator.go:316-327 -> evaluateInPlace :200-304)
    def _mult_int(self, ct, c):
        out = np.empty_like(ct.host)
        for k in range(ct.host.shape[0]):
            for j in range(ct.host.shape[1]):
                out[k, j] = self.ks.ringQ.mul_scalar(j, ct.host[k, j], int(c) % self.Q[j])
        return OCt(ct.ids, out, ct.Scale)

    def AddNew(self, op0, op1):
        s0, s1 = op0.Scale, op1.Scale
        if s1 > s0 and math.floor(s1 / s0) > 1:
            op0 = self._mult_int(op0, math.floor(s1 / s0))
        elif s0 > s1 and math.floor(s0 / s1) > 1:
            op1 = self._mult_int(op1, math.floor(s0 / s1))
        ids = sorted(op0.IDSet() | op1.IDSet())
        level = min(op0.Level(), op1.Level())
        out = np.zeros((1 + len(ids), level + 1, self.N), dtype=np.uint64)
        def comp(ct, id):
            if id == "0":
                return ct.host[0]
            return ct.host[1 + ct.ids.index(id)] if id in ct.ids else None
        for s, id in enumerate(["0"] + ids):
            a, b = comp(op0, id), comp(op1, id)
            for j in range(level + 1):
                out[s, j] = a[j] if b is None else (b[j] if a is None else self.ks.ringQ.add(j, a[j], b[j]))
        return OCt(ids, out, max(s0, s1))

    # ---- HoistedForm (evaluator.go:543-553)
    def HoistedForm(self, ct):
        h = OHoisted()
        for i, id in enumerate(ct.ids):
            h.Value[id] = self.ks.decompose(ct.Level(), ct.host[1 + i])
        return h

    # ---- MulRelin[Hoisted]New (evaluator.go:416-443,558-581)
    def MulRelinNew(self, op0, op1, rlkSet):
        return self.MulRelinHoistedNew(op0, op1, None, None, rlkSet)

    def MulRelinHoistedNew(self, op0, op1, h0, h1, rlkSet):
        ids = sorted(op0.IDSet() | op1.IDSet())
        idx = {id: i for i, id in enumerate(ids)}
        level = min(op0.Level(), op1.Level())
        rl = {idx[id]: self.rlk[id] for id in ids}
        hh0 = {idx[id]: h0.Value[id] for id in op0.ids} if h0 is not None else None
        hh1 = {idx[id]: h1.Value[id] for id in op1.ids} if h1 is not None else None
        ido, out = self.ks.mul_and_relin(level, [idx[i] for i in op0.ids], op0.host, [idx[i] for i in op1.ids], op1.host, rl, self.crs[-1], hh0, hh1)
        assert ido == list(range(len(ids)))
        out, scale = self._rescale(out, op0.Scale * op1.Scale)
        return OCt(ids, out, scale)

    # ---- RotateNew / RotateHoistedNew (evaluator.go:485-525,585-617)
    def _rotate(self, ct, rotidx, hoisted):
        n = len(ct.ids)
        rk = [self.rk[(id, rotidx)] for id in ct.ids]
        hs = [hoisted.Value[id] for id in ct.ids] if hoisted is not None else None
        out = self.ks.rotate(ct.Level(), self._galois(rotidx), list(range(n)), ct.host, rk, self.crs[rotidx], hs)
        return OCt(ct.ids, out, ct.Scale)

    def RotateNew(self, ct, rotidx, rkSet):
        rotidx %= self.N // 2
        if rotidx == 0:
            return OCt(ct.ids, ct.host.copy(), ct.Scale)
        if rotidx in self.crs:
            return self._rotate(ct, rotidx, None)
        tmp, k = ct, 1
        while rotidx > 0:
            if rotidx % 2:
                tmp = self._rotate(tmp, k, None)
            rotidx //= 2
            k *= 2
        return tmp

    def RotateHoistedNew(self, ct, rotidx, hoisted, rkSet):
        rotidx %= self.N // 2
        if rotidx == 0:
            return OCt(ct.ids, ct.host.copy(), ct.Scale)
        return self._rotate(ct, rotidx, hoisted)

    # ---- MulPtxtNew (evaluator.go:465-481); pt_value: host polynomial uint64[>= level + 1][N], coefficient domain
    def MulPtxtNew(self, ct, pt_value, pt_scale):
        level, r = ct.Level(), self.ks.ringQ
        pt = [r.mform(j, r.ntt(j, np.ascontiguousarray(pt_value[j]))) for j in range(level + 1)]
        out = np.empty_like(ct.host)
        for k in range(ct.host.shape[0]):
            for j in range(level + 1):
                out[k, j] = r.intt(j, r.mul(j, r.ntt(j, ct.host[k, j]), pt[j]))
        out, scale = self._rescale(out, ct.Scale * float(pt_scale))
        return OCt(ct.ids, out, scale)

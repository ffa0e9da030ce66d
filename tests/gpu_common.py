"""Shared helpers of the -m gpu parity tests: build the same random material on both sides
(CPU oracle <-> HIP engine through the C ABI)."""
import numpy as np

import harness as H
from oracle import oracle as O


class Pair:
    """One parameter set instantiated on the oracle (ks) and on the device (params)."""

    def __init__(self, pset, seed=0, gamma=2):
        from mkhe_kklss_amd import mkrlwe
        self.mk = mkrlwe
        self.pset = pset
        self.logN, self.N = pset["logN"], 1 << pset["logN"]
        self.Q, self.P = pset["Q"], pset["P"]
        self.ks = O.KeySwitcher(self.logN, self.Q, self.P, gamma)
        self.params = mkrlwe.Parameters(self.logN, self.Q, self.P, gamma)
        self.ksw = mkrlwe.NewKeySwitcher(self.params)
        self.rng = np.random.default_rng(seed)
        self.maxlevel = len(self.Q) - 1

    def swk(self):
        host = H.uniform_swk(self.rng, self.ks)
        return host, self.mk.SwitchingKey(self.params, host)

    def ct(self, ids, level, limbs=None):
        limbs = level + 1 if limbs is None else limbs
        host = H.uniform_ct(self.rng, self.ks, len(ids), limbs)
        dev = self.mk.NewCiphertext(self.params, ids, limbs - 1).upload(host)
        return host, dev

    def rlk_set(self, ids):
        host = {}
        dev = self.mk.RelinearizationKeySet(self.params)
        for i in ids:
            b, d, v = (H.uniform_swk(self.rng, self.ks) for _ in range(3))
            host[i] = (b, d, v)
            dev.AddRelinearizationKey(self.mk.RelinearizationKey(self.params, i, b, d, v))
        return host, dev


def oracle_mul_and_relin(pair, level, ids0, op0, ids1, op1, rlk_host, u_host, names):
    """ids are strings on the device side; the oracle wants dense ints."""
    idx = {n: k for k, n in enumerate(names)}
    rl = {idx[i]: rlk_host[i] for i in rlk_host}
    ido, out = pair.ks.mul_and_relin(level, [idx[i] for i in ids0], op0, [idx[i] for i in ids1], op1, rl, u_host)
    return [names[i] for i in ido], out

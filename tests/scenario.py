"""Valid (seeded) multi-key CKKS material for replaying the reference's property tests
(mkckks_test.go genTestParams :135-179, newTestVectors :181-199) on the oracle and on the device."""
import numpy as np

import harness as H
from oracle import oracle as O


class Scenario:
    def __init__(self, pset, parties, seed=1, rotations=(), conj=False):
        self.pset = pset
        self.logN, self.N = pset["logN"], 1 << pset["logN"]
        self.Q, self.P, self.scale = pset["Q"], pset["P"], pset["scale"]
        self.level = len(self.Q) - 1
        self.ks = O.KeySwitcher(self.logN, self.Q, self.P, 2)
        self.kg = H.KeyGen(self.ks, seed)
        self.names = ["user%d" % i for i in range(parties)]
        self.kg.add_crs(0)
        self.kg.add_crs(-1)
        if conj:
            self.kg.add_crs(-2)
        for r in rotations:
            self.kg.add_crs(r)
        self.sk, self.sk_small, self.pk, self.rlk, self.rk, self.ck = {}, {}, {}, {}, {}, {}
        for n in self.names:
            sk, s = self.kg.gen_secret_key()
            r, _ = self.kg.gen_secret_key()
            self.sk[n], self.sk_small[n] = sk, s
            self.pk[n] = self.kg.gen_public_key(sk)
            self.rlk[n] = self.kg.gen_relin_key(sk, r)
            self.rk[n] = {rot: self.kg.gen_rotation_key(rot, sk, s) for rot in rotations}
            if conj:
                self.ck[n] = self.kg.gen_conjugation_key(sk, s)
        self.enc = H.CKKSEncoder(self.logN)
        self.rng = np.random.default_rng(seed + 1000)

    def message(self, lo, hi):
        n = self.N // 2
        return self.rng.uniform(lo.real, hi.real, n) + 1j * self.rng.uniform(lo.imag, hi.imag, n)

    def encrypt(self, z, name, level=None):
        """-> (c0, c1) coefficient domain, party `name`"""
        level = self.level if level is None else level
        pt = self.enc.encode(z, self.scale, self.Q[: level + 1])
        return self.kg.encrypt(pt, self.pk[name], level)

    def sum_ciphertext(self, zs):
        """sum of single-party encryptions of zs[name] -> (names, host ct [1+k][L][N])"""
        L = self.level + 1
        ct = np.zeros((1 + len(self.names), L, self.N), dtype=np.uint64)
        for a, n in enumerate(self.names):
            c0, c1 = self.encrypt(zs[n], n)
            for j in range(L):
                ct[0][j] = self.ks.ringQ.add(j, ct[0][j], c0[j])
            ct[1 + a] = c1
        return ct

    def decrypt_decode(self, names, ct, scale):
        level = ct.shape[1] - 1
        vals = {"0": ct[0]}
        for a, n in enumerate(names):
            vals[n] = ct[1 + a]
        sks = {n: self.sk[n][: len(self.Q)] for n in names}
        pt = self.kg.decrypt(vals, sks)
        return self.enc.decode(pt, scale, self.Q[: level + 1])

    def precision_bound(self, extra):
        """mkckks_test.go:221,357: log2|delta| <= -log2(scale) + logSlots + extra"""
        return -np.log2(self.scale) + (self.logN - 1) + extra

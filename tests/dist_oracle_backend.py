"""TEST-ONLY backend for mkhe_kklss_amd.dist.ShardedMulRelin: local arithmetic done by the CPU oracle,
so the sharding + collective pattern can be exercised with gloo on CPU (no GPU in this container)."""
import numpy as np

from mkhe_kklss_amd.dist import assign_parties, assign_units


class OracleShardBackend:
    def __init__(self, ks, names, rank, world, op0_host, op1_host, rlk_host, crs_u, level, torch):
        self.ks, self.names, self.level, self.torch = ks, list(names), level, torch
        ids0, ids1 = assign_units(self.names, world)[rank]
        self.idx = {n: i for i, n in enumerate(self.names)}
        self.ids0, self.ids1, self.with_c0 = [self.idx[n] for n in ids0], [self.idx[n] for n in ids1], rank == 0
        sl = lambda host, ids: np.ascontiguousarray(np.stack([host[0]] + [host[1 + i] for i in ids]))
        self.op0, self.op1 = sl(op0_host, self.ids0), sl(op1_host, self.ids1)
        self.rlk = {self.idx[n]: rlk_host[n] for n in rlk_host}
        self.u = crs_u
        self.act = list(range(level + 1)) + [len(ks.Q) + j for j in range(len(ks.P))]

    def _ring(self, j):
        nq = len(self.ks.Q)
        return (self.ks.ringQ, j) if j < nq else (self.ks.ringP, j - nq)

    def partial_xy(self):
        x, y = self.ks.mr_xy(self.level, self.ids0, self.op0, self.ids1, self.op1, self.rlk, mform=False)
        self.x, self.y = x, y
        self.tx, self.ty = self.torch.from_numpy(x.view(np.int64)), self.torch.from_numpy(y.view(np.int64))
        return self.tx, self.ty

    def before_collective(self): pass
    def after_collective(self): pass

    def fold_xy(self):
        for buf in (self.x, self.y):
            for i in range(self.ks.beta(self.level)):
                for j in self.act:
                    r, k = self._ring(j)
                    buf[i][j] = r.mform(k, r.reduce(k, buf[i][j]))

    def finish(self):
        ido, out = self.ks.mr_finish(self.level, self.ids0, self.op0, self.ids1, self.op1, self.x, self.y,
                                     self.rlk, self.u, self.with_c0)
        full = np.zeros((1 + len(self.names), self.level + 1, self.ks.N), dtype=np.uint64)
        full[0] = out[0]
        for a, i in enumerate(ido):
            full[1 + i] = out[1 + a]
        self.full = full
        self.tfull = self.torch.from_numpy(full.view(np.int64))
        return self.tfull

    def fold_out(self):
        for s in range(self.full.shape[0]):
            for j in range(self.level + 1):
                self.full[s][j] = self.ks.ringQ.reduce(j, self.full[s][j])


class OracleRotateBackend:
    """TEST-ONLY backend of mkhe_kklss_amd.dist.ShardedRotate on the CPU oracle."""

    def __init__(self, ks, names, rank, world, ct_host, rk_host, crs, galEl, level, torch):
        self.ks, self.names, self.level, self.torch, self.galEl = ks, list(names), level, torch, galEl
        self.ids = [self.names.index(n) for n in assign_parties(self.names, world)[rank]]
        self.with_c0 = rank == 0
        self.ct, self.rk, self.crs = ct_host, rk_host, crs

    def partial(self):
        ks, L = self.ks, self.level + 1
        full = np.zeros((1 + len(self.names), L, ks.N), dtype=np.uint64)
        if self.with_c0:
            full[0] = self.ct[0]
        for i in self.ids:
            a = self.ct[1 + i]
            e = ks.external_product(self.level, a, self.rk[self.names[i]])
            for j in range(L):
                full[0][j] = ks.ringQ.add(j, full[0][j], e[j])
            full[1 + i] = ks.external_product(self.level, a, self.crs)
        self.full = full
        return self.torch.from_numpy(full.view(np.int64))

    def before_collective(self): pass
    def after_collective(self): pass

    def finish(self):
        ks = self.ks
        out = np.empty_like(self.full)
        for s in range(self.full.shape[0]):
            red = np.stack([ks.ringQ.reduce(j, self.full[s][j]) for j in range(self.level + 1)])
            out[s] = ks.ringQ.permute(self.galEl, red)
        self.out = out
        return out

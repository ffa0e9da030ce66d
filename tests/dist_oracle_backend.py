"""TEST-ONLY backend for mkhe_kklss_amd.dist.ShardedMulRelin: local arithmetic done by the CPU oracle,
so the sharding + collective pattern can be exercised with gloo on CPU (no GPU in this container)."""
import numpy as np

from mkhe_kklss_amd.dist import assign_moduli, assign_parties, assign_units


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

    # ---- the mesh form of the x / y exchange (ShardedMulRelin._mesh_*): this rank folds its slice from the pieces it received
    def limb_words(self):
        return self.ks.N

    def fold_pieces(self, which, recv, npieces, piece_words, first_word, nwords):
        buf = self.x if which == "x" else self.y
        N, mtot = self.ks.N, buf.shape[1]
        pieces = recv.numpy().view(np.uint64).reshape(npieces, piece_words)
        flat = buf.reshape(-1)
        for i in range(nwords // N):
            l = first_word // N + i
            d, j = divmod(l, mtot)
            if d >= self.ks.beta(self.level) or j not in self.act:
                continue
            r, k = self._ring(j)
            tot = pieces[:, i * N:(i + 1) * N].sum(axis=0, dtype=np.uint64)
            flat[l * N:(l + 1) * N] = r.mform(k, r.reduce(k, tot))

    def finish_head(self):
        pass                                                  # (the oracle has no split finish: everything happens in finish_tail)

    def finish_tail(self):
        return self.finish()

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


class OracleLimbBackend:
    """TEST-ONLY backend of mkhe_kklss_amd.dist.LimbShardedMulRelin for the gloo tests: the CPU oracle evaluates the whole
    MulAndRelin on every rank and the backend hands LimbShardedMulRelin exactly the slices a rank would own (zeros
    elsewhere), so that the exchange pattern -- all-reduces of disjoint slices, four rounds -- is exercised end to end:
    rounds 1-3 carry a known pattern cut by modulus ownership, round 4 the real output limbs."""

    def __init__(self, ks, names, rank, world, op0, op1, rlk, crs_u, level, torch):
        self.ks, self.names, self.level, self.torch = ks, list(names), level, torch
        nq, np_ = len(ks.Q), len(ks.P)
        self.owned = set(assign_moduli(nq, np_, world)[rank])
        ids = list(range(len(names)))
        _, self.ref = ks.mul_and_relin(level, ids, op0, ids, op1, {i: rlk[n] for i, n in enumerate(names)}, crs_u)
        self.nq, self.np_ = nq, np_
        self.pattern = {}

    def _mask(self, arr, first_mod):
        """arr [..., limbs, N]: keep the limbs whose modulus index first_mod + l is owned"""
        out = np.zeros_like(arr)
        for l in range(arr.shape[-2]):
            if first_mod + l in self.owned:
                out[..., l, :] = arr[..., l, :]
        return out

    def phase(self, ph):
        k, N, L = len(self.names), self.ks.N, self.level + 1
        if ph == 4:
            self.out = self._mask(self.ref, 0)
            return self.torch.from_numpy(self.out.view(np.int64).reshape(-1))
        shape = {1: (k, self.np_, N), 2: (k, L, N), 3: (3 * k, self.np_, N)}[ph]
        full = (np.arange(int(np.prod(shape)), dtype=np.uint64) * np.uint64(2654435761 + ph)).reshape(shape)
        self.pattern[ph] = full
        self.sent = self._mask(full, self.nq if ph in (1, 3) else 0)
        return self.torch.from_numpy(self.sent.view(np.int64).reshape(-1))

    def before_collective(self): pass
    def after_collective(self): pass

    def result(self):
        assert (self.sent.reshape(-1) == self.pattern[3].reshape(-1)).all()      # round 3 arrived complete
        return self.out


class OracleBfvShardBackend:
    """TEST-ONLY backend of mkhe_kklss_amd.dist.ShardedBfvMulRelin: the sharded algorithm restated with the oracle's mkbfv
    building blocks (ModUpQtoR / Rescale / Quantize / DecomposeBFV / ExternalProductBFVHoisted / Decompose /
    ExternalProductHoisted, mkbfv/keyswitch_hoisted.go:36-206) and its ring operations for the linear parts."""

    def __init__(self, bfv, names, rank, world, op0_host, op1_host, rlk_host, crs_u, torch):
        self.bfv, self.names, self.torch = bfv, list(names), torch
        self.idx = {n: i for i, n in enumerate(self.names)}
        self.ids = [self.idx[n] for n in assign_parties(self.names, world)[rank]]
        self.with_c0 = rank == 0
        self.op0 = [op0_host[0]] + [op0_host[1 + i] for i in self.ids]
        self.op1 = [op1_host[0]] + [op1_host[1 + i] for i in self.ids]
        self.rlk = {self.idx[n]: rlk_host[n] for n in rlk_host}
        self.u = crs_u
        self.level = bfv.nq - 1

    def _rings(self, m):                       # limbs of a switching key: Q then P
        nq = self.bfv.nq
        return [(self.bfv.ringQ, j) for j in range(nq)] + [(self.bfv.ringP, j) for j in range(m - nq)]

    def _ringr(self, j):
        return (self.bfv.ringQ, j) if j < self.bfv.nq else (self.bfv.ringQMul, j - self.bfv.nq)

    def _mac(self, acc, key, h):               # acc += key (.) h over all digits and limbs (canonical partial sums)
        for i in range(acc.shape[0]):
            for jj, (r, j) in enumerate(self._rings(acc.shape[1])):
                acc[i][jj] = r.mul_add(j, key[i][jj], h[i][jj], acc[i][jj])

    def partial_xy(self):
        bfv = self.bfv
        self.r0 = [bfv.modup_q_to_r(p) for p in self.op0]
        self.r1 = [bfv.rescale(p) for p in self.op1]
        self.h0 = [bfv.decompose(p) for p in self.r0[1:]]
        self.h1 = [bfv.decompose(p) for p in self.r1[1:]]
        z = lambda: np.zeros_like(bfv.ks.new_swk())
        self.x1, self.x2, self.y1, self.y2 = z(), z(), z(), z()
        for a, i in enumerate(self.ids):
            b1, b2, d1, d2, _ = self.rlk[i]
            self._mac(self.x1, d1, self.h0[a][0]); self._mac(self.x2, d2, self.h0[a][1])
            self._mac(self.y1, b1, self.h1[a][0]); self._mac(self.y2, b2, self.h1[a][1])
        self.t = [self.torch.from_numpy(v.view(np.int64)) for v in (self.x1, self.x2, self.y1, self.y2)]
        return self.t

    def before_collective(self): pass
    def after_collective(self): pass

    def fold_xy(self):
        for buf in (self.x1, self.x2, self.y1, self.y2):
            for i in range(buf.shape[0]):
                for jj, (r, j) in enumerate(self._rings(buf.shape[1])):
                    buf[i][jj] = r.mform(j, r.reduce(j, buf[i][jj]))

    def finish(self):
        bfv, nq, N = self.bfv, self.bfv.nq, self.bfv.N
        mulr = lambda a, b: np.stack([self._ringr(j)[0].mul(self._ringr(j)[1], a[j], b[j]) for j in range(2 * nq)])
        addr = lambda a, b: np.stack([self._ringr(j)[0].add(self._ringr(j)[1], a[j], b[j]) for j in range(2 * nq)])
        mformr = lambda a: np.stack([self._ringr(j)[0].mform(self._ringr(j)[1], a[j]) for j in range(2 * nq)])
        f0 = [bfv.ntt_r(p) for p in self.r0]
        f1 = [bfv.ntt_r(p) for p in self.r1]
        p1 = mformr(f0[0])                                         # MForm(NTT(op0_0))
        p2m = mformr(f1[0])                                        # MForm(NTT(op1_0))
        out = np.zeros((1 + len(self.ids), nq, N), dtype=np.uint64)
        if self.with_c0:
            out[0] = bfv.quantize(mulr(p1, f1[0]))
        for a in range(len(self.ids)):
            out[1 + a] = bfv.quantize(addr(mulr(p1, f1[1 + a]), mulr(p2m, f0[1 + a])))
        addq = lambda a, b: np.stack([bfv.ringQ.add(j, a[j], b[j]) for j in range(nq)])
        for a, i in enumerate(self.ids):
            out[1 + a] = addq(out[1 + a], bfv.external_product_hoisted(self.h1[a][0], self.h1[a][1], self.x1, self.x2))
        for a, i in enumerate(self.ids):
            t = bfv.external_product_hoisted(self.h0[a][0], self.h0[a][1], self.y1, self.y2)
            ht = bfv.ks.decompose(self.level, t)
            out[0] = addq(out[0], bfv.ks.external_product_hoisted(self.level, ht, self.rlk[i][4]))
            out[1 + a] = addq(out[1 + a], bfv.ks.external_product_hoisted(self.level, ht, self.u))
        full = np.zeros((1 + len(self.names), nq, N), dtype=np.uint64)
        full[0] = out[0]
        for a, i in enumerate(self.ids):
            full[1 + i] = out[1 + a]
        self.full = full
        self.tfull = self.torch.from_numpy(full.view(np.int64))
        return self.tfull

    def fold_out(self):
        for s in range(self.full.shape[0]):
            for j in range(self.bfv.nq):
                self.full[s][j] = self.bfv.ringQ.reduce(j, self.full[s][j])

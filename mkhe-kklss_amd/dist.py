"""Party-sharded MulAndRelin over torch.distributed (one process per GPU; backend "nccl" = RCCL over
xGMI on MI355X, "gloo" in the CPU tests).

The reference is single-process; the sharding follows the algorithm's own structure
(keyswitch_hoisted.go:79-117,147-178).  Work is cut into 2k half-party units
(party i, side 0) = hoist c0_i, d_i-term of x, step F of party i      (needs y)
(party i, side 1) = hoist c1_i, b_i-term of y, step E of party i      (needs x)
and every rank evaluates the reference algorithm on the sub-ciphertexts made of c_0 and the
components of its units.  Exchange steps (the only collectives):
  1. x_part, y_part: uint64 sums of canonical residues, then fold + MForm -- as reduce-scatter + all-gather over disjoint limb slices where the
     buffer divides (round 6, `mesh`: one all-to-all of slices, every rank sums, folds and MForm's ITS slice -- 1 / world of the fold work, per-link
     payload buffer / world on the point-to-point xGMI mesh, SURVEY.md 8e(2) -- then an all-gather of the folded slices), else an all-reduce
  2. all-reduce(out_0) + all-gather(out_i)    out_0 is a sum over ranks; out_i comes from party i's owner (ranks that own whole
                                              parties in equal numbers; otherwise one all-reduce of the whole ciphertext)
All sums are exact (ranks * q < 2^63) and order independent, so the result is bit-identical to the
single-device evaluation.

Ordering of the collectives (every Hip*Backend, `sync`): "stream" (default under RCCL) makes the engine's own stream torch's
current stream (torch.cuda.ExternalStream) for every collective and every torch copy, so RCCL orders itself against the
engine's kernels with events and the host never waits inside a step; the all-reduces of y and x are issued asynchronously
back to back, the engine waits for y alone, runs the part of the finish that needs only y (F1 and the Decompose of the t_i:
mkhe_mr_finish_head) while x is still being reduced, then waits for x (mkhe_mr_finish_tail).  "host" drains the engine stream
before every collective and waits for it afterwards (gloo, which moves device tensors through the host).
"""
import contextlib
import numpy as np


def assign_units(names, world):
    """-> list over ranks of (ids0_local, ids1_local).  Units are chunked evenly in the order
    (p0,0),(p0,1),(p1,0),... so for world <= k (k % world == 0) a rank owns whole parties."""
    units = [(n, s) for n in names for s in (0, 1)]
    out = [([], []) for _ in range(world)]
    for j, (n, s) in enumerate(units):
        r = j * world // len(units)
        out[r][s].append(n)
    return out


class _TorchOnEngineStream:
    """mixin of the Hip*Backends: torch work (collectives, slot copies) ordered against the engine's kernels"""

    def _init_sync(self, params, torch, dev, sync):
        self.ext = torch.cuda.ExternalStream(int(params.stream()), device=dev)
        self.sync = sync

    def set_sync(self, sync):
        self.sync = sync

    @contextlib.contextmanager
    def torch_section(self):
        if self.sync == "stream":
            with self.torch.cuda.stream(self.ext):           # enqueued on the engine stream: no host synchronisation
                yield
        else:
            self.params.sync()                                # engine stream -> host; torch runs on its own stream
            yield
            self.torch.cuda.current_stream().synchronize()

    # (the pre-round-2 names, still used by the emulated-rank tests)
    def before_collective(self):
        self.params.sync()

    def after_collective(self):
        self.torch.cuda.current_stream().synchronize()


@contextlib.contextmanager
def _section(backend):
    sec = getattr(backend, "torch_section", None)
    if sec is not None:
        with sec():
            yield
    else:
        backend.before_collective()
        yield
        backend.after_collective()


class ShardedMulRelin:
    """Orchestrates one party-sharded MulAndRelinHoisted.  `backend` does the local arithmetic
    (HipShardBackend below on the GPU); `dist` is torch.distributed (or None for world size 1).
    force_collectives: issue the collectives even at world size 1 (exercises the stream plumbing on one GPU)."""

    def __init__(self, backend, dist=None, group=None, force_collectives=False, mesh=True):
        self.b, self.dist, self.group, self.force, self.mesh = backend, dist, group, force_collectives, mesh
        self._recv = {}
        self.used_mesh = False                # the last run exchanged x and y as reduce-scatter + all-gather

    def _active(self):
        return self.dist is not None and (self.force or self.dist.get_world_size(self.group) > 1)

    # ---- x / y on the mesh: reduce-scatter (all-to-all of slices + the rank's own fold) and all-gather
    def _mesh_ok(self, t):
        """the buffer divides into one slice of whole limbs per rank, and the backend can fold a slice from pieces"""
        if not (self.mesh and self._active() and hasattr(self.b, "fold_pieces")):
            return False
        world = self.dist.get_world_size(self.group)
        return t.numel() % (world * self.b.limb_words()) == 0

    def _mesh_scatter(self, t, name, async_op=False):
        """all-to-all: piece p of the receive buffer = rank p's slice `rank` of its partial sum"""
        recv = self._recv.get(name)
        if recv is None or recv.numel() != t.numel() or recv.device != t.device:
            recv = self._recv[name] = t.new_empty(t.numel())
        return recv, self.dist.all_to_all_single(recv, t.view(-1), group=self.group, async_op=async_op)

    def _mesh_finish(self, t, recv, which):
        """this rank's slice: sum of the pieces, fold, MForm (into its place in t); then every rank's folded slice to everybody"""
        world, rank = self.dist.get_world_size(self.group), self.dist.get_rank(self.group)
        per = t.numel() // world
        self.b.fold_pieces(which, recv, world, per, rank * per, per)
        flat = t.view(-1)
        with _section(self.b):
            self.dist.all_gather_into_tensor(flat, flat[rank * per:(rank + 1) * per].clone(), group=self.group)

    def _all_reduce(self, t):
        if self._active():
            with _section(self.b):
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def run(self):
        b = self.b
        x, y = b.partial_xy()                 # torch int64 views of the rank's partial sums
        self.used_mesh = False
        if self._mesh_ok(x) and hasattr(b, "finish_head"):
            self.used_mesh = True
            # y first (F1 -> Decompose(t_i), the long chain); x's slices travel while the head runs
            stream = getattr(b, "sync", "host") == "stream"
            with _section(b):
                ry, wy = self._mesh_scatter(y, "y", async_op=stream)
                rx, wx = self._mesh_scatter(x, "x", async_op=stream)
                if stream:
                    wy.wait()                 # blocks the engine stream, not the host
            self._mesh_finish(y, ry, "y")
            b.finish_head()
            if stream:
                with b.torch_section():
                    wx.wait()
            self._mesh_finish(x, rx, "x")
            full = b.finish_tail()
        elif self._active() and getattr(b, "sync", "host") == "stream" and hasattr(b, "finish_head"):
            # y is needed first (F1 -> Decompose(t_i), the long chain), x only by step E: both reductions are started now, the
            # engine stream waits for y, and x is reduced under F1 / Decompose
            dist = self.dist
            with b.torch_section():
                wy = dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                wx = dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                wy.wait()                     # blocks the engine stream, not the host
            b.fold_y()
            b.finish_head()
            with b.torch_section():
                wx.wait()
            b.fold_x()
            full = b.finish_tail()
        else:
            self._all_reduce(x)
            self._all_reduce(y)
            b.fold_xy()                       # x, y <- MForm(sum mod q)
            full = b.finish()                 # torch int64 view [1+k][L][N]: own contributions, zeros elsewhere
        self._exchange_out(full)
        b.fold_out()                          # every slot <- sum mod q
        return full

    def _exchange_out(self, full):
        """out_0 is a sum over the ranks (keyswitch_hoisted.go:147-178: every party adds <h(t_i), v_i> to it): all-reduce of
        that one slot.  out_i comes from the rank that owns party i when the ranks own whole parties in equal numbers: then
        the slots are all-gathered (each rank sends only its own), otherwise (half-party units: two ranks add into the same
        slot) the whole ciphertext is all-reduced."""
        dist = self.dist
        if not self._active():
            return
        world, k = dist.get_world_size(self.group), full.shape[0] - 1
        if k == 0 or k % world != 0:
            self._all_reduce(full)
            return
        c, rank = k // world, dist.get_rank(self.group)
        with _section(self.b):
            dist.all_reduce(full[0], op=dist.ReduceOp.SUM, group=self.group)
            chunks = [full[1 + r * c: 1 + (r + 1) * c] for r in range(world)]
            dist.all_gather(chunks, chunks[rank].clone(), group=self.group)   # the input is copied: it is also one of the output views


def assign_parties(names, world):
    """whole parties per rank for the rotation: -> list over ranks of id lists (contiguous chunks)"""
    out = [[] for _ in range(world)]
    for j, n in enumerate(names):
        out[j * world // len(names)].append(n)
    return out


class ShardedRotate:
    """Party-sharded Rotate[Hoisted] (keyswitch.go:234-298, keyswitch_hoisted.go:183-247): the 2k external products
    are independent per party.  Rank r evaluates them for its parties (rank 0 also carries c_0); one all-reduce of the
    un-permuted ciphertext (slot 0 is a sum over ranks, slot i comes from its owner, zeros elsewhere), then the signed
    permutation on the folded result -- after the reduction, so that the reference's "q - 0 = q" representative of
    keyswitch.go:290 comes out exactly as on one device."""

    def __init__(self, backend, dist=None, group=None, force_collectives=False):
        self.b, self.dist, self.group, self.force = backend, dist, group, force_collectives

    def run(self):
        b = self.b
        full = b.partial()                    # torch int64 view [1+k][L][N]
        if self.dist is not None and (self.force or self.dist.get_world_size(self.group) > 1):
            with _section(b):
                self.dist.all_reduce(full, op=self.dist.ReduceOp.SUM, group=self.group)
        return b.finish()                     # fold + permutation -> the rotated ciphertext


class _DevView:
    """exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)"""

    def __init__(self, ptr, nwords):
        self.__cuda_array_interface__ = dict(shape=(int(nwords),), typestr="<i8", data=(int(ptr), False), version=3, strides=None)


class HipShardBackend(_TorchOnEngineStream):
    """Local arithmetic of one rank on its MI355X through the C ABI (include/mkhe.h, mkhe_mr_*)."""

    def __init__(self, params, names, rank, world, op0_host, op1_host, rlk_host, level, torch, device_index, sync="host"):
        """op*_host: full ciphertexts uint64[1+k][L][N]; rlk_host: {name: (b, d, v)} for (at least) the
        parties this rank needs.  Only the rank's own components and keys are uploaded.  sync: see the module docstring."""
        from . import mkrlwe
        import ctypes as C
        from ._abi import check, lib
        self.params, self.names, self.level, self.torch = params, list(names), level, torch
        self.C, self.check, self.lib, self.mk = C, check, lib, mkrlwe
        ids0, ids1 = assign_units(self.names, world)[rank]
        self.ids0, self.ids1, self.with_c0 = ids0, ids1, rank == 0
        sl = lambda host, ids: np.ascontiguousarray(np.stack([host[0]] + [host[1 + self.names.index(n)] for n in ids]))
        self.op0 = mkrlwe.NewCiphertext(params, ids0, level).upload(sl(op0_host, ids0))
        self.op1 = mkrlwe.NewCiphertext(params, ids1, level).upload(sl(op1_host, ids1))
        # key material: host arrays (uploaded here) or resident SwitchingKey handles (e.g. written by mkhe_crs_expand)
        as_key = lambda k_: k_ if isinstance(k_, mkrlwe.SwitchingKey) else mkrlwe.SwitchingKey(params, k_)
        self.keys = {n: [as_key(rlk_host[n][j]) if need else None
                         for j, need in enumerate((n in ids1, n in ids0, n in ids0))]
                     for n in set(ids0) | set(ids1)}
        self.out_ids = sorted(set(ids0) | set(ids1))
        self.out = mkrlwe.NewCiphertext(params, self.out_ids, level)
        self.full = mkrlwe.NewCiphertext(params, self.names, level)
        self.x, self.y = mkrlwe.NewSwitchingKey(params), mkrlwe.NewSwitchingKey(params)
        dev = torch.device("cuda", device_index)
        words = int(lib().mkhe_ctx_swk_words(params.ctx))
        self.tx = torch.as_tensor(_DevView(self.x.devptr(), words), device=dev)
        self.ty = torch.as_tensor(_DevView(self.y.devptr(), words), device=dev)
        N, L = params.N(), level + 1
        self.tfull = torch.as_tensor(_DevView(self.full.devptr(), (1 + len(self.names)) * L * N), device=dev).view(1 + len(self.names), L, N)
        self.tout = torch.as_tensor(_DevView(self.out.devptr(), (1 + len(self.out_ids)) * L * N), device=dev).view(1 + len(self.out_ids), L, N)
        self._init_sync(params, torch, dev, sync)

    def _arr(self, hs):
        from ._abi import handle_array
        return handle_array(hs)

    def partial_xy(self):
        b1 = [self.keys[n][0].h for n in self.op1.ids]
        d0 = [self.keys[n][1].h for n in self.op0.ids]
        self.check(self.lib().mkhe_mr_partial(self.params.ctx, self.op0.h, self.op1.h, None, None,
                                              self._arr(b1), self._arr(d0), 1 if self.with_c0 else 0, self.out.h, self.x.h, self.y.h))
        return self.tx, self.ty

    def fold_x(self):
        self.check(self.lib().mkhe_swk_fold(self.params.ctx, self.x.h, self.level, 1))

    def fold_y(self):
        self.check(self.lib().mkhe_swk_fold(self.params.ctx, self.y.h, self.level, 1))

    def fold_xy(self):
        self.fold_x()
        self.fold_y()

    def limb_words(self):
        return self.params.N()

    def fold_pieces(self, which, recv, npieces, piece_words, first_word, nwords):
        """x or y: the slice [first_word, first_word + nwords) <- MForm(sum over the pieces of `recv` mod q) (mkhe_swk_fold_pieces)"""
        N = self.params.N()
        dst = (self.x if which == "x" else self.y).devptr() + 8 * first_word
        self.check(self.lib().mkhe_swk_fold_pieces(self.params.ctx, self.C.c_void_p(recv.data_ptr()), npieces, piece_words, first_word // N, nwords // N,
                                                   self.level, 1, self.C.c_void_p(dst)))

    def _spread_out(self):
        """the rank's slots of `out` into the full-width ciphertext (zeros elsewhere)"""
        with self.torch_section():
            self.tfull.zero_()
            self.tfull[0].copy_(self.tout[0])
            for a, n in enumerate(self.out_ids):
                self.tfull[1 + self.names.index(n)].copy_(self.tout[1 + a])
        return self.tfull

    def finish(self):
        v0 = [self.keys[n][2].h for n in self.op0.ids]
        self.check(self.lib().mkhe_mr_finish(self.params.ctx, self.op0.h, self.op1.h, self.x.h, self.y.h,
                                             self._arr(v0), self.params.CRS[-1].h, self.out.h))
        return self._spread_out()

    def finish_head(self):
        self.check(self.lib().mkhe_mr_finish_head(self.params.ctx, self.op0.h, self.op1.h, self.y.h, self.out.h))

    def finish_tail(self):
        v0 = [self.keys[n][2].h for n in self.op0.ids]
        self.check(self.lib().mkhe_mr_finish_tail(self.params.ctx, self.op0.h, self.op1.h, self.x.h,
                                                  self._arr(v0), self.params.CRS[-1].h, self.out.h))
        return self._spread_out()

    def fold_out(self):
        self.check(self.lib().mkhe_ct_fold(self.params.ctx, self.full.h))


class HipRotateBackend(_TorchOnEngineStream):
    """Local arithmetic of one rank for ShardedRotate through the C ABI (mkhe_rotate_partial, mkhe_ct_fold,
    mkhe_ct_automorphism)."""

    def __init__(self, params, names, rank, world, ct_host, rk_host, crs, rotidx, level, torch, device_index, hoisted=False, sync="host"):
        """ct_host: uint64[1+k][L][N]; rk_host: {name: rotation key array or resident SwitchingKey} for (at least) this
        rank's parties; crs: the device SwitchingKey params.CRS[rotidx]; hoisted: RotateHoisted (keyswitch_hoisted.go:183-247)
        on hoisted forms of the rank's components computed once here, instead of Rotate (keyswitch.go:234-298)."""
        from . import mkrlwe
        from ._abi import check, handle_array, lib
        self.params, self.names, self.level, self.torch = params, list(names), level, torch
        self.check, self.lib, self.harr = check, lib, handle_array
        self.ids = assign_parties(self.names, world)[rank]
        self.with_c0 = rank == 0
        sub = np.ascontiguousarray(np.stack([ct_host[0]] + [ct_host[1 + self.names.index(n)] for n in self.ids]))
        self.sub = mkrlwe.NewCiphertext(params, self.ids, level).upload(sub)
        self.part = mkrlwe.NewCiphertext(params, self.ids, level)
        self.keys = [rk_host[n] if isinstance(rk_host[n], mkrlwe.SwitchingKey) else mkrlwe.SwitchingKey(params, rk_host[n]) for n in self.ids]
        self.crs = crs
        self.hoist = None
        if hoisted and self.ids:
            self.hoist = [mkrlwe.SwitchingKey(params, zero=False) for _ in self.ids]
            check(lib().mkhe_hoisted_form(params.ctx, level, self.sub.h, handle_array([h.h for h in self.hoist])))
        self.galEl = params.GaloisElementForColumnRotationBy(rotidx)
        self.full = mkrlwe.NewCiphertext(params, self.names, level)
        self.out = mkrlwe.NewCiphertext(params, self.names, level)
        dev = torch.device("cuda", device_index)
        N, L = params.N(), level + 1
        self.tfull = torch.as_tensor(_DevView(self.full.devptr(), (1 + len(self.names)) * L * N), device=dev).view(1 + len(self.names), L, N)
        self.tpart = torch.as_tensor(_DevView(self.part.devptr(), (1 + len(self.ids)) * L * N), device=dev).view(1 + len(self.ids), L, N)
        self._init_sync(params, torch, dev, sync)

    def partial(self):
        self.check(self.lib().mkhe_rotate_partial(self.params.ctx, self.sub.h, self.harr([h.h for h in self.hoist]) if self.hoist else None,
                                                  self.harr([k.h for k in self.keys]),
                                                  self.crs.h, 1 if self.with_c0 else 0, self.part.h))
        with self.torch_section():
            self.tfull.zero_()
            self.tfull[0].copy_(self.tpart[0])
            for a, n in enumerate(self.ids):
                self.tfull[1 + self.names.index(n)].copy_(self.tpart[1 + a])
        return self.tfull

    def finish(self):
        self.check(self.lib().mkhe_ct_fold(self.params.ctx, self.full.h))
        self.check(self.lib().mkhe_ct_automorphism(self.params.ctx, self.galEl, self.full.h, self.out.h))
        return self.out


# ---------------------------------------------------------------------------------------------------------------------
# Limb sharding: every rank holds the full operands (ciphertext polynomials are small) and owns a subset of the RNS
# moduli.  NTTs, inner products and ModDown outputs are computed for the owned moduli only, for ALL parties, so the
# per-party sums x, y -- the 2 x 56 MiB that party sharding has to all-reduce -- never leave the rank.  What crosses
# the links per step (PN15QP880, k = 4): the P limbs of the external products (2 + 6 MiB), t_i (14 MiB) and the
# output ciphertext (17.5 MiB).  All exchanges are all-reduces of disjoint slices (zeros elsewhere): exact and order
# independent, so the result is bit-identical to the single-device evaluation.
def assign_moduli(nq, np_, world):
    """-> list over ranks of modulus-index lists (Q then P), round-robin so every rank gets Q limbs and the two
    modulus classes (small / big primes) are spread evenly"""
    out = [[] for _ in range(world)]
    for j in range(nq + np_):
        out[j % world].append(j)
    return out


class LimbShardedMulRelin:
    """Orchestrates one limb-sharded MulAndRelin: four engine phases around three exchanges + the output exchange.
    force_collectives: issue the all-reduces even at world size 1 (exercises the stream plumbing on one GPU)."""

    def __init__(self, backend, dist=None, group=None, force_collectives=False):
        self.b, self.dist, self.group, self.force = backend, dist, group, force_collectives

    def _all_reduce(self, t):
        if self.dist is not None and (self.force or self.dist.get_world_size(self.group) > 1):
            if hasattr(self.b, "all_reduce"):
                self.b.all_reduce(self.dist, t, self.group)          # stream-ordered, no host synchronisation
                return
            self.b.before_collective()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.b.after_collective()

    def run(self):
        b = self.b
        for phase in (1, 2, 3, 4):
            t = b.phase(phase)                 # torch int64 view of what has to be summed over the ranks
            if t.numel():
                self._all_reduce(t)
        return b.result()


class HipLimbBackend:
    """One rank of LimbShardedMulRelin on its MI355X through the C ABI (mkhe_ctx_set_owned, mkhe_lsh_phase)."""

    def __init__(self, params, names, rank, world, op0_host, op1_host, rlk_host, level, torch, device_index, sync="stream"):
        """full operands uint64[1+k][L][N] and the keys {name: (b, d, v)} of ALL parties (only the owned limbs of the
        keys are ever read).  sync = "stream": the collectives are enqueued with the engine's stream as torch's current
        stream (torch.cuda.ExternalStream), so RCCL orders itself against the engine's kernels with events and the host
        never waits inside a step; "host": drain the engine stream, all-reduce on torch's stream, wait for it."""
        import ctypes as C
        from . import mkrlwe, _abi
        from ._abi import check, handle_array, lib
        self.params, self.names, self.level, self.torch = params, list(names), level, torch
        self.C, self.check, self.lib, self.harr = C, check, lib, handle_array
        self.owned = assign_moduli(len(params.Q), len(params.P), world)[rank]
        arr = np.asarray(self.owned, dtype=np.int32)
        check(lib().mkhe_ctx_set_owned(params.ctx, arr.ctypes.data_as(_abi.i32p), len(arr)))
        self.op0 = mkrlwe.NewCiphertext(params, self.names, level).upload(op0_host)
        self.op1 = mkrlwe.NewCiphertext(params, self.names, level).upload(op1_host)
        self.keys = {n: [mkrlwe.SwitchingKey(params, rlk_host[n][j]) for j in range(3)] for n in self.names}
        self.out = mkrlwe.NewCiphertext(params, self.names, level)
        k, N, L = len(self.names), params.N(), level + 1
        self.stage_words = max((3 * k) * len(params.P) * N, k * L * N, 1)
        self.stage = mkrlwe.DeviceLimbs(params, 1, -(-self.stage_words // N))
        dev = torch.device("cuda", device_index)
        self.tstage = torch.as_tensor(_DevView(self.stage.devptr().value, self.stage.words), device=dev)
        self.tout = torch.as_tensor(_DevView(self.out.devptr(), (1 + k) * L * N), device=dev)
        self.b1 = self.harr([self.keys[n][0].h for n in self.names])
        self.d0 = self.harr([self.keys[n][1].h for n in self.names])
        self.v0 = self.harr([self.keys[n][2].h for n in self.names])
        self.ext = torch.cuda.ExternalStream(int(params.stream()), device=dev)
        self.set_sync(sync)

    def set_sync(self, sync):
        """"stream": collectives ordered on the engine stream; "host": through the host (see __init__)"""
        self.sync = sync
        if sync == "stream":
            self.all_reduce = self._all_reduce_on_engine_stream
        elif "all_reduce" in self.__dict__:
            del self.all_reduce

    def _all_reduce_on_engine_stream(self, dist, t, group):
        with self.torch.cuda.stream(self.ext):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def phase(self, ph):
        w = self.C.c_size_t(0)
        first, third = ph == 1, ph == 3
        self.check(self.lib().mkhe_lsh_phase(self.params.ctx, ph, self.op0.h, self.op1.h,
                                             self.b1 if first else None, self.d0 if first else None,
                                             self.v0 if third else None, self.params.CRS[-1].h if third else None,
                                             self.out.h, self.stage.devptr(), self.C.byref(w)))
        return self.tout[: w.value] if ph == 4 else self.tstage[: w.value]

    def before_collective(self):
        self.params.sync()                         # engine stream -> host; RCCL runs on torch's stream

    def after_collective(self):
        self.torch.cuda.current_stream().synchronize()

    def result(self):
        return self.out


# ---------------------------------------------------------------------------------------------------------------------
# mkbfv: MulRelinNew sharded by WHOLE parties (mkbfv/keyswitch_hoisted.go:36-206).  Quantize rounds, so the two tensor terms
# of an output slot (op0_0 * op1_j + op0_j * op1_0) must be added on one rank before it: the half-party units of the CKKS
# sharding do not work here, a rank holds both components of every party it owns (world <= k).  The linear parts are
# exchanged exactly as in ShardedMulRelin: all-reduce of the four partial sums x1, x2, y1, y2 (Q and QMul gadgets), then
# all-reduce(out_0) + all-gather(out_i).
class ShardedBfvMulRelin(ShardedMulRelin):
    def run(self):
        b = self.b
        parts = b.partial_xy()                # four torch int64 views: x1, x2, y1, y2
        for t in parts:
            self._all_reduce(t)
        b.fold_xy()
        full = b.finish()
        self._exchange_out(full)
        b.fold_out()
        return full


class HipBfvShardBackend(_TorchOnEngineStream):
    """Local arithmetic of one rank of ShardedBfvMulRelin through the C ABI (mkhe_bfv_mr_partial / mkhe_swk_fold /
    mkhe_bfv_mr_finish / mkhe_ct_fold).  rlk_host: {name: (b1, b2, d1, d2, v)} for (at least) the rank's parties."""

    def __init__(self, params, names, rank, world, op0_host, op1_host, rlk_host, torch, device_index, sync="host"):
        from . import mkbfv, mkrlwe
        from ._abi import check, handle_array, lib
        self.params, self.names, self.torch = params, list(names), torch
        self.check, self.lib, self.harr = check, lib, handle_array
        if world > len(self.names):
            raise ValueError("BFV sharding needs whole parties: world <= number of parties")
        self.ids = assign_parties(self.names, world)[rank]
        self.with_c0 = rank == 0
        sl = lambda host: np.ascontiguousarray(np.stack([host[0]] + [host[1 + self.names.index(n)] for n in self.ids]))
        self.op0 = mkbfv.NewCiphertext(params, self.ids).upload(sl(op0_host))
        self.op1 = mkbfv.NewCiphertext(params, self.ids).upload(sl(op1_host))
        as_key = lambda k_: k_ if isinstance(k_, mkrlwe.SwitchingKey) else mkrlwe.SwitchingKey(params, k_)
        self.keys = {n: [as_key(k_) for k_ in rlk_host[n]] for n in self.ids}
        self.out = mkbfv.NewCiphertext(params, self.ids)
        self.full = mkbfv.NewCiphertext(params, self.names)
        self.xy = [mkrlwe.NewSwitchingKey(params) for _ in range(4)]
        dev = torch.device("cuda", device_index)
        words = int(lib().mkhe_ctx_swk_words(params.ctx))
        self.txy = [torch.as_tensor(_DevView(s.devptr(), words), device=dev) for s in self.xy]
        N, L = params.N(), params.MaxLevel() + 1
        self.level = params.MaxLevel()
        self.tfull = torch.as_tensor(_DevView(self.full.devptr(), (1 + len(self.names)) * L * N), device=dev).view(1 + len(self.names), L, N)
        self.tout = torch.as_tensor(_DevView(self.out.devptr(), (1 + len(self.ids)) * L * N), device=dev).view(1 + len(self.ids), L, N)
        self._init_sync(params, torch, dev, sync)

    def _k(self, j):
        return self.harr([self.keys[n][j].h for n in self.ids])

    def partial_xy(self):
        x1, x2, y1, y2 = self.xy
        self.check(self.lib().mkhe_bfv_mr_partial(self.params.ctx, self.op0.h, self.op1.h, self._k(0), self._k(1), self._k(2), self._k(3),
                                                  1 if self.with_c0 else 0, self.out.h, x1.h, x2.h, y1.h, y2.h))
        return self.txy

    def fold_xy(self):
        for s in self.xy:
            self.check(self.lib().mkhe_swk_fold(self.params.ctx, s.h, self.level, 1))

    def finish(self):
        x1, x2, y1, y2 = self.xy
        self.check(self.lib().mkhe_bfv_mr_finish(self.params.ctx, self.op0.h, self.op1.h, x1.h, x2.h, y1.h, y2.h, self._k(4),
                                                 self.params.CRS[-1].h, self.out.h))
        with self.torch_section():
            self.tfull.zero_()
            self.tfull[0].copy_(self.tout[0])
            for a, n in enumerate(self.ids):
                self.tfull[1 + self.names.index(n)].copy_(self.tout[1 + a])
        return self.tfull

    def fold_out(self):
        self.check(self.lib().mkhe_ct_fold(self.params.ctx, self.full.h))

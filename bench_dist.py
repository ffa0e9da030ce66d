"""bench.py, N > 1 leg: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).

The SAME k-party MulRelin as the single-GPU bench is evaluated sharded over the ranks (mkhe_kklss_amd/dist.py):
strong scaling, value = MulRelin/s of the whole job.
  --shard limb  (default) every rank owns a subset of the RNS moduli and evaluates all parties there; x, y stay local;
                exchanged per step: P limbs of the external products, t_i, the output ciphertext (~40 MB at k = 4)
  --shard party the paper's structure: half-party units; all-reduce of the x and y partial sums (beta*(nQ+nP)*N words
                each) and of the output ciphertext (~135 MB at k = 4)
"""
import os
import sys
import time

import numpy as np


def run_distributed(args):
    import torch
    import torch.distributed as dist
    import harness as H
    from bench import synth_party_keys, synth_cts, synth_swk
    from mkhe_kklss_amd import mkckks
    from mkhe_kklss_amd._abi import check, lib
    from mkhe_kklss_amd.dist import HipLimbBackend, HipShardBackend, LimbShardedMulRelin, ShardedMulRelin, assign_units

    # a plain `python bench.py --force-dist` (no launcher): single-rank rendezvous on the loopback interface
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend_name = os.environ.get("MKHE_DIST_BACKEND", "nccl")          # "gloo" + MKHE_DIST_ONE_DEVICE=1: functional test of
    if os.environ.get("MKHE_DIST_ONE_DEVICE"):                          # the N > 1 path on a single-GPU box (timings meaningless)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if backend_name == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend_name, rank=rank, world_size=world)
    pset = H.PN15QP880 if args.params == "PN15QP880" else H.PN14QP439
    k = args.parties
    names = ["user%d" % i for i in range(k)]
    level = len(pset["Q"]) - 1
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=local_rank)
    op0, op1 = synth_cts(pset, k, args.seed)
    params.AddCRS(-1, synth_swk(pset, np.random.default_rng(args.seed + 7)))
    nwx = int(lib().mkhe_ctx_swk_words(params.ctx))
    Nn, L, npp = 1 << pset["logN"], level + 1, len(pset["P"])
    if args.shard == "limb":
        # every rank: full operands and (the owned limbs of) every party's keys; x, y stay local
        rlk = {n: synth_party_keys(pset, names.index(n), args.seed) for n in names}
        backend = HipLimbBackend(params, names, rank, world, op0, op1, rlk, level, torch, local_rank,
                                 sync="stream" if args.dist_sync == "auto" else args.dist_sync)
        smr = LimbShardedMulRelin(backend, dist, force_collectives=bool(os.environ.get("MKHE_FORCE_COLLECTIVES")))
        full = backend.out
        exchanged = 8 * Nn * (k * npp + k * L + 3 * k * npp + (k + 1) * L)
        sharding = "RNS limbs (every rank: all parties, its moduli), see mkhe_kklss_amd/dist.py LimbShardedMulRelin"
    else:
        ids0, ids1 = assign_units(names, world)[rank]
        rlk = {n: synth_party_keys(pset, names.index(n), args.seed) for n in set(ids0) | set(ids1)}
        backend = HipShardBackend(params, names, rank, world, op0, op1, rlk, level, torch, local_rank)
        smr = ShardedMulRelin(backend, dist)
        full = backend.full
        exchanged = 8 * (2 * nwx + (k + 1) * L * Nn)
        sharding = "half-party units, see mkhe_kklss_amd/dist.py ShardedMulRelin"
    del rlk
    res = mkckks.NewCiphertext(params, names, level - 1, pset["scale"])

    def step():
        smr.run()
        check(lib().mkhe_rescale(params.ctx, full.h, 1, res.h))

    for _ in range(args.warmup):
        step()
    params.sync(); torch.cuda.synchronize(); dist.barrier()
    sync_mode = getattr(backend, "sync", "host")
    if args.shard == "limb" and args.dist_sync == "auto":
        # untimed probe: three steps with each way of ordering the collectives, every rank adopts the faster one
        probe = {}
        for mode in ("stream", "host"):
            backend.set_sync(mode)
            step()
            params.sync(); torch.cuda.synchronize(); dist.barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            params.sync(); torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            probe[mode] = float(tt.item())
        sync_mode = min(probe, key=probe.get)
        backend.set_sync(sync_mode)
        step()
        params.sync(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    params.sync(); torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    out = None
    if rank == 0:
        out = dict(metric="mkckks_mulrelin_per_sec", value=args.steps / dt, unit="MulRelin/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps, higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="u64", data="synthetic",
                   config=dict(workload="mkckks %d-party MulRelin (hoist + MulAndRelinHoisted + Rescale), %s N=2^%d, %d Q + %d P limbs, "
                                        "sharded over %d GPUs" % (k, args.params, pset["logN"], len(pset["Q"]), len(pset["P"]), world),
                               parties=k, params=args.params, seed=args.seed, sharding=sharding, collective_ordering=sync_mode,
                               allreduce_bytes_per_step=exchanged),
                   roofline=None, cpu_baseline=None)
    dist.barrier()
    dist.destroy_process_group()
    return out

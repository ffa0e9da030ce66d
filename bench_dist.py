"""bench.py, N > 1 leg: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).

The SAME k-party MulRelin as the single-GPU bench is evaluated sharded over the ranks (mkhe_kklss_amd/dist.py): strong
scaling, value = MulRelin/s of the whole job.  Legs (all timed with the barrier / synchronise / max-over-ranks contract):
  party  (the headline `value`; SURVEY.md 8e, BASELINE.json north_star): GPU g owns whole parties (half-party units when
         there are more GPUs than parties) with their keys; all-reduce of the x and y partial sums, all-reduce of out_0 and
         all-gather of the out_i
  limb   (secondary, alpha = 1 rings only): every rank owns a subset of the RNS moduli and evaluates all parties there;
         x, y stay local; exchanged: P limbs of the external products, t_i, the output ciphertext
  rotate (secondary): hoisted Rotate of the k-party ciphertext, parties sharded, one all-reduce
--params PN16QP1761 --parties 8 is BASELINE.json configs[3] (keys written on the device by the CRS expander).
"""
import os
import sys
import time

import numpy as np


def _timed(dist, torch, params, fn, steps, warmup):
    """W untimed + K timed calls of fn between barrier + synchronise on both sides; MAX over ranks; -> seconds"""
    for _ in range(warmup):
        fn()
    params.sync(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    params.sync(); torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt.item())


def _init(torch, dist):
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
    return rank, world, local_rank


def _party_sync(args, dist):
    """ordering of the collectives of the party-sharded legs (mkhe_kklss_amd/dist.py): on the engine's stream under RCCL,
    through the host with gloo (which moves device tensors through the host) or when --dist-sync host asks for it"""
    return "stream" if dist.get_backend() == "nccl" and getattr(args, "dist_sync", "auto") != "host" else "host"


def run_distributed_bfv(args):
    """--scheme bfv at N > 1 (BASELINE.json configs[2] sharded): mkbfv MulRelinNew with whole parties per rank
    (mkhe_kklss_amd/dist.py ShardedBfvMulRelin); needs N <= parties"""
    import torch
    import torch.distributed as dist
    import harness_bfv as HB
    from mkhe_kklss_amd import mkbfv
    from mkhe_kklss_amd.dist import HipBfvShardBackend, ShardedBfvMulRelin, assign_parties
    rank, world, local_rank = _init(torch, dist)
    pset = HB.BFV_PN15QP880 if args.params == "PN15QP880" else HB.BFV_PN14QP439
    k = args.parties
    names = ["user%d" % i for i in range(k)]
    data = HB.uniform_bfv_inputs(pset, k, args.seed)
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"], device=local_rank)
    params.AddCRS(-1, data["u"])
    mine = assign_parties(names, world)[rank]
    rlk = {n: data["rlk"][names.index(n)] for n in mine}
    b = HipBfvShardBackend(params, names, rank, world, data["op0"], data["op1"], rlk, torch, local_rank, sync=_party_sync(args, dist))
    del data, rlk
    smr = ShardedBfvMulRelin(b, dist)
    dt = _timed(dist, torch, params, smr.run, args.steps, args.warmup)
    from bench import roofline_leg
    roofline = roofline_leg(args, params, smr.run, pset["logN"], "%s-bfv k=%d sharded over %d" % (args.params, k, world))     # every rank: the steps contain collectives
    out = None
    if rank == 0:
        out = dict(metric="mkbfv_mulrelin_per_sec", value=args.steps / dt, unit="MulRelin/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps, higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="u64", data="synthetic",
                   config=dict(workload="mkbfv %d-party MulRelinNew, %s, whole parties sharded over %d GPUs (RCCL all-reduce of x1, x2, "
                                        "y1, y2, out_0; all-gather of out_i)" % (k, args.params, world), parties=k, params=args.params,
                               seed=args.seed, rccl_ranks=dist.get_world_size()),
                   roofline=roofline, cpu_baseline=None)
    dist.barrier()
    dist.destroy_process_group()
    return out


def run_replicas_cnn(args):
    """--scheme cnn at N > 1 (BASELINE.json configs[4]): the encrypted inference is one latency-bound circuit that does not shard;
    every GPU runs its own inference (independent replicas, no data-path collective): weak scaling, value = inferences/s of the node"""
    import torch
    import torch.distributed as dist
    from bench import run_cnn
    rank, world, local_rank = _init(torch, dist)
    args.device = local_rank
    dist.barrier()
    one = run_cnn(args)
    t = torch.tensor([one["ms_per_step"]], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = None
    if rank == 0:
        ms = float(t.item())
        out = dict(one, value=world * 1e3 / ms, ms_per_step=ms, n_gpus=world, scaling="weak")            # roofline: rank 0's replica
        out["config"] = dict(one["config"], replicas=world, sharding="independent replicas: one inference per GPU, no collective")
    dist.barrier()
    dist.destroy_process_group()
    return out


def run_distributed(args):
    import torch
    import torch.distributed as dist
    import harness as H
    from bench import synth_party_keys, synth_cts, synth_swk
    from mkhe_kklss_amd import mkckks, mkrlwe
    from mkhe_kklss_amd._abi import check, lib
    from mkhe_kklss_amd.dist import (HipLimbBackend, HipRotateBackend, HipShardBackend, LimbShardedMulRelin, ShardedMulRelin,
                                     ShardedRotate, assign_parties, assign_units)

    rank, world, local_rank = _init(torch, dist)
    pset = {"PN15QP880": H.PN15QP880, "PN14QP439": H.PN14QP439, "PN16QP1761": H.PN16QP1761}[args.params]
    k = args.parties
    names = ["user%d" % i for i in range(k)]
    level = len(pset["Q"]) - 1
    device_keys = args.device_keys or args.params == "PN16QP1761"
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=local_rank)
    op0, op1 = synth_cts(pset, k, args.seed)
    if device_keys:
        params.AddCRS(-1, seed=args.seed)
    else:
        params.AddCRS(-1, synth_swk(pset, np.random.default_rng(args.seed + 7)))
    nwx = int(lib().mkhe_ctx_swk_words(params.ctx))
    Nn, L, npp = 1 << pset["logN"], level + 1, len(pset["P"])

    def party_keys(n):
        """(b, d, v) of one party: host arrays, or device handles filled by the CRS expander (same on every rank: the seed and
        the index are public)"""
        i = names.index(n)
        if not device_keys:
            return synth_party_keys(pset, i, args.seed)
        keys = []
        for j in range(3):
            key = mkrlwe.SwitchingKey(params, zero=False)
            check(lib().mkhe_crs_expand(params.ctx, args.seed, 1000 + 3 * i + j, key.h))
            keys.append(key)
        return tuple(keys)

    legs = {}
    res = mkckks.NewCiphertext(params, names, level - 1, pset["scale"])
    # ---- party sharding (headline)
    ids0, ids1 = assign_units(names, world)[rank]
    rlk = {n: party_keys(n) for n in sorted(set(ids0) | set(ids1))}
    psync = _party_sync(args, dist)
    backend = HipShardBackend(params, names, rank, world, op0, op1, rlk, level, torch, local_rank, sync=psync)
    smr = ShardedMulRelin(backend, dist, mesh=os.environ.get("MKHE_DIST_MESH", "1") != "0")      # (0: x / y as all-reduces, the exchange of rounds 3-5)
    def step_party():
        smr.run()
        check(lib().mkhe_rescale(params.ctx, backend.full.h, 1, res.h))
    dt = _timed(dist, torch, params, step_party, args.steps, args.warmup)
    legs["party"] = dict(mulrelin_per_sec=args.steps / dt, ms_per_step=dt * 1e3 / args.steps, collective_ordering=psync,
                         x_y_exchange="reduce-scatter (all-to-all of limb slices + the rank's own fold) + all-gather" if smr.used_mesh else "all-reduce",
                         exchanged_bytes_per_step=8 * (2 * nwx + (1 + (k // world if k % world == 0 else k)) * L * Nn))
    # per-kernel HIP-event leg on every rank (the steps contain collectives), rank 0's figures are reported: the dominant kernel of a
    # rank's share of the work (its launch sizes shrink with N: a "sharded k=.. N=.." workload tag keeps the single-GPU PMC traffic out)
    from bench import roofline_leg
    roofline = roofline_leg(args, params, step_party, pset["logN"], "%s k=%d sharded over %d" % (args.params, k, world))
    # ---- self-validation + replica leg (round 3).  Every rank builds the full key set and evaluates the SAME MulRelin on its own GPU
    # (the single-GPU path of bench.py): rank 0 compares the sharded result with it byte for byte (config.matches_single_gpu) and, with
    # host key material, with the CPU oracle (cpu_baseline.bit_exact_vs_gpu); timed under the same contract, the N independent
    # evaluations are the replica leg (weak scaling: N MulRelins per step, no data-path collective) reported beside the strong-scaling
    # `value` -- DESIGN.md section 7 says which one the link model expects to win.
    sharded = res.download() if rank == 0 else None
    rlk_all = mkrlwe.RelinearizationKeySet(params)
    host_keys = {}
    for n in names:
        kb = party_keys(n)
        if device_keys:
            key = mkrlwe.RelinearizationKey.__new__(mkrlwe.RelinearizationKey)      # wraps the expanded handles, nothing allocated
            key.ID, key.Value = n, list(kb)
            rlk_all.AddRelinearizationKey(key)
        else:
            host_keys[names.index(n)] = kb
            rlk_all.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *kb))
    ct0 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(op0)
    ct1 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(op1)
    ev1 = mkckks.NewEvaluator(params)
    single = ev1.MulRelinNew(ct0, ct1, rlk_all)
    matches = bool(rank != 0 or (sharded.shape == tuple(single.download().shape) and (sharded == single.download()).all()))
    def step_replica():
        return ev1.MulRelinNew(ct0, ct1, rlk_all)
    dt = _timed(dist, torch, params, step_replica, args.steps, args.warmup)
    legs["replicas"] = dict(mulrelin_per_sec=world * args.steps / dt, ms_per_step=dt * 1e3 / args.steps, scaling="weak",
                            note="every GPU evaluates its own %d-party MulRelin with all keys resident: N per step, no collective" % k)
    cpu = None
    if rank == 0 and not args.no_cpu and not device_keys:
        from oracle import oracle as O
        ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
        ids = list(range(k))
        u_h = synth_swk(pset, np.random.default_rng(args.seed + 7))
        nth = min(os.cpu_count() or 1, len(pset["Q"]) + len(pset["P"]))
        O.set_threads(nth)
        t0 = time.perf_counter()
        _, ref = ks.mul_and_relin(level, ids, op0, ids, op1, host_keys, u_h)
        ref = np.stack([ks.ringQ.div_round_last_many(ref[s_], 1)[0] for s_ in range(1 + k)])
        cdt = time.perf_counter() - t0
        O.set_threads(1)
        cpu = dict(value=1.0 / cdt, unit="MulRelin/s", cores=nth, kind="port",
                   sample="1 full %d-party MulRelin (%s), oracle with its limb loops on %d host threads of rank 0, %.1f s" % (k, args.params, nth, cdt),
                   bit_exact_vs_gpu=bool((sharded == ref).all()))
    del smr, backend, rlk, rlk_all, host_keys, ev1, single, ct0, ct1
    # ---- hoisted Rotate, parties sharded (BASELINE.json configs[3]: "MulRelin + hoisted Rotate")
    rot = 1
    mine = assign_parties(names, world)[rank]
    if device_keys:
        params.AddCRS(rot, seed=args.seed)
        rk = {}
        for n in mine:
            key = mkrlwe.SwitchingKey(params, zero=False)
            check(lib().mkhe_crs_expand(params.ctx, args.seed, 2000 + names.index(n), key.h))
            rk[n] = key
    else:
        params.AddCRS(rot, synth_swk(pset, np.random.default_rng(args.seed + 99)))
        rk = {n: synth_swk(pset, np.random.default_rng(args.seed + 100 + names.index(n))) for n in mine}
    rb = HipRotateBackend(params, names, rank, world, op0, rk, params.CRS[rot], rot, level, torch, local_rank, hoisted=True, sync=psync)
    srot = ShardedRotate(rb, dist)
    dt = _timed(dist, torch, params, srot.run, args.steps, args.warmup)
    legs["rotate_hoisted"] = dict(rotate_per_sec=args.steps / dt, ms_per_step=dt * 1e3 / args.steps,
                                  exchanged_bytes_per_step=8 * (k + 1) * L * Nn)
    del srot, rb, rk
    # ---- limb sharding (secondary; alpha = 1 only: with alpha >= 2 a gadget digit spans several moduli)
    sync_mode = None
    if params.Alpha() == 1 and not device_keys and not args.no_limb_leg:
        rlk = {n: synth_party_keys(pset, names.index(n), args.seed) for n in names}
        lb = HipLimbBackend(params, names, rank, world, op0, op1, rlk, level, torch, local_rank,
                            sync="stream" if args.dist_sync == "auto" else args.dist_sync)
        del rlk
        lsm = LimbShardedMulRelin(lb, dist, force_collectives=bool(os.environ.get("MKHE_FORCE_COLLECTIVES")))
        def step_limb():
            lsm.run()
            check(lib().mkhe_rescale(params.ctx, lb.out.h, 1, res.h))
        sync_mode = lb.sync
        if args.dist_sync == "auto":
            # untimed probe: three steps with each way of ordering the collectives, every rank adopts the faster one
            probe = {}
            for mode in ("stream", "host"):
                lb.set_sync(mode)
                probe[mode] = _timed(dist, torch, params, step_limb, 3, 1)
            sync_mode = min(probe, key=probe.get)
            lb.set_sync(sync_mode)
        dt = _timed(dist, torch, params, step_limb, args.steps, args.warmup)
        legs["limb"] = dict(mulrelin_per_sec=args.steps / dt, ms_per_step=dt * 1e3 / args.steps, collective_ordering=sync_mode,
                            exchanged_bytes_per_step=8 * Nn * (k * npp + k * L + 3 * k * npp + (k + 1) * L))
    out = None
    if rank == 0:
        v = legs["party"]
        out = dict(metric="mkckks_mulrelin_per_sec", value=v["mulrelin_per_sec"], unit="MulRelin/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=v["ms_per_step"], higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="u64", data="synthetic",
                   config=dict(workload="mkckks %d-party MulRelin (hoist + MulAndRelinHoisted + Rescale), %s N=2^%d, %d Q + %d P limbs, "
                                        "parties sharded over %d GPUs (RCCL all-reduce of x, y, out_0; all-gather of out_i)"
                                        % (k, args.params, pset["logN"], len(pset["Q"]), len(pset["P"]), world),
                               parties=k, params=args.params, seed=args.seed, rccl_ranks=dist.get_world_size(),
                               key_material="device" if device_keys else "host",
                               sharding="parties (mkhe_kklss_amd/dist.py ShardedMulRelin); secondary legs below", legs=legs,
                               matches_single_gpu=matches),
                   roofline=roofline, cpu_baseline=cpu)
    dist.barrier()
    dist.destroy_process_group()
    return out

#!/usr/bin/env python3
"""bench.py -- MK-CKKS MulRelin/sec on MI355X (BASELINE.json metric) + NTT roofline + CPU baseline.

A "step" is one mkckks.Evaluator.MulRelinNew on synthetic k-party ciphertexts:
hoisting of both operands (Decompose), KeySwitcher.MulAndRelinHoisted and Rescale -- the timed
region of the reference benchmark (mkckks/mkckks_benchmark_test.go:78-82), all inputs and keys
resident in HBM when the clock starts.

  python bench.py --gpus 1 --steps K --warmup W          (default workload = configs[1])
  python bench.py --gpus N ...                           (spawns N ranks itself: torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the same, launched from outside)

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec



def synth_swk(pset, rng):
    """uniform residues per limb, switching-key shaped uint64[beta][nQ+nP][N] (alpha = 1)"""
    Q, P, N = pset["Q"], pset["P"], 1 << pset["logN"]
    out = np.empty((len(Q), len(Q) + len(P), N), dtype=np.uint64)
    for i in range(len(Q)):
        for j, q in enumerate(Q + P):
            out[i, j] = rng.integers(0, q, N, dtype=np.uint64)
    return out


def synth_party_keys(pset, party_index, seed):
    """(b, d, v) of one party; seeded per party so every rank can regenerate just what it needs"""
    rng = np.random.default_rng(seed + 1000 * (party_index + 1))
    return tuple(synth_swk(pset, rng) for _ in range(3))


def synth_cts(pset, parties, seed):
    rng = np.random.default_rng(seed)
    Q, N = pset["Q"], 1 << pset["logN"]
    ct = lambda: np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q]) for _ in range(1 + parties)])
    return ct(), ct()


def synth_inputs(pset, parties, seed):
    """uniform residues per limb: ciphertexts, rlk (b,d,v) per party, CRS u."""
    op0, op1 = synth_cts(pset, parties, seed)
    return dict(op0=op0, op1=op1, rlk=[synth_party_keys(pset, i, seed) for i in range(parties)],
                u=synth_swk(pset, np.random.default_rng(seed + 7)))


def run_bfv(args):
    """--scheme bfv: one mkbfv.Evaluator.MulRelinNew (BASELINE.json configs[3] shape: k-party, PN15QP880 BFV chain,
    ModUpQtoR / Rescale / DecomposeBFV / MulAndRelinBFVHoisted) per step.  Secondary line, same JSON contract."""
    import harness_bfv as HB
    from mkhe_kklss_amd import mkbfv
    from mkhe_kklss_amd._abi import check, lib

    pset = HB.BFV_PN15QP880 if args.params == "PN15QP880" else HB.BFV_PN14QP439
    k = args.parties
    names = ["user%d" % i for i in range(k)]
    data = HB.uniform_bfv_inputs(pset, k, args.seed)
    params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"], device=0)
    ct0 = mkbfv.NewCiphertext(params, names).upload(data["op0"])
    ct1 = mkbfv.NewCiphertext(params, names).upload(data["op1"])
    rlk = mkbfv.NewRelinearizationKeyKeySet(params)
    for i, n in enumerate(names):
        rlk.AddRelinearizationKey(mkbfv.RelinearizationKey(params, n, *data["rlk"][i]))
    params.AddCRS(-1, data["u"])
    ev = mkbfv.NewEvaluator(params)
    if os.environ.get("MKHE_NO_OVERLAP"):
        check(lib().mkhe_set_overlap(params.ctx, 0))
    step = lambda: ev.MulRelinNew(ct0, ct1, rlk)
    # same order of legs as run_single: a cold-start figure first, then a long window, then the contract's W + K steps on settled clocks
    extras = {}
    for _ in range(args.warmup):
        res = step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    params.sync()
    extras["mulrelin_per_sec_cold_start"] = args.steps / (time.perf_counter() - t0)
    extras["timing_protocol"] = ("legs in this order: cold start (W warm-up + K timed steps right after the host-side set-up: mulrelin_per_sec_cold_start, the figure "
                                 "rounds 1-2 reported as value) -> secondary legs -> 60 untimed + 120 timed steps (steady state) -> W + K = the timed region of `value` -> K steps under HIP events (roofline)")
    if not getattr(args, "no_extras", False):
        un = lambda: ev.mulRelin(ct0, ct1, rlk)         # the non-hoisted twin (mkbfv/keyswitch.go:115-251) on its own device path
        for _ in range(2):
            un()
        params.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            un()
        params.sync()
        extras["mulrelin_unhoisted_per_sec"] = args.steps / (time.perf_counter() - t0)
    for _ in range(60):
        step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(120):
        step()
    params.sync()
    extras["mulrelin_per_sec_steady_state"] = 120 / (time.perf_counter() - t0)
    for _ in range(args.warmup):
        res = step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    params.sync()
    dt = time.perf_counter() - t0
    roofline = roofline_leg(args, params, step, pset["logN"], "%s-bfv k=%d" % (args.params, k))
    cpu = None
    if not args.no_cpu:
        bfv = HB.make_bfv(pset)
        ids = list(range(k))
        reps = max(1, args.cpu_reps)
        t0 = time.perf_counter()
        for _ in range(reps):
            _, ref = bfv.mul_relin_new(ids, data["op0"], ids, data["op1"], data["rlk"], data["u"])
        cdt = time.perf_counter() - t0
        cpu = dict(value=reps / cdt, unit="MulRelin/s", cores=1, kind="port",
                   sample="%d full %d-party BFV MulRelin (%s) on 1 host thread, %.1f s" % (reps, k, args.params, cdt),
                   bit_exact_vs_gpu=bool((res.download() == ref).all()))
    return dict(metric="mkbfv_mulrelin_per_sec", value=args.steps / dt, unit="MulRelin/s", n_gpus=1, steps=args.steps,
                warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps, higher_is_better=True, scaling="strong",
                vs_baseline=None, dtype="u64", data="synthetic",
                config=dict(workload="mkbfv %d-party MulRelinNew (ModUpQtoR + Rescale + DecomposeBFV + MulAndRelinBFVHoisted), %s N=2^%d, %d Q + %d QMul + %d P limbs"
                            % (k, args.params, pset["logN"], len(pset["Q"]), len(pset["QMul"]), len(pset["P"])),
                            parties=k, params=args.params, seed=args.seed, **extras),
                roofline=roofline, cpu_baseline=cpu)


id_builtin = id


def run_cnn(args):
    """--scheme cnn: the reference's BenchmarkCNN (cnn/cnn_bench_test.go:11-78): one encrypted inference = Convolution,
    square, FC1, square, FC2 on PN14QP433; --parties 2 = the reference's dataOwner / modelOwner, 4 = one owner per layer
    (BASELINE.json configs[4]).  Keys and CRS are generated on the device; ciphertext limbs are uniform (timing does
    not depend on the values; encrypted == plaintext is tests/test_gpu_cnn.py).  Secondary line, same JSON contract."""
    import harness_cnn as HC
    from mkhe_kklss_amd import cnn, mkckks, mkrlwe
    from mkhe_kklss_amd._abi import check, lib
    p = HC.PN14QP433
    owners = (dict(image="dataOwner", kernels="modelOwner", fc1="modelOwner", fc2="modelOwner") if args.parties <= 2 else
              dict(image="dataOwner", kernels="convOwner", fc1="fc1Owner", fc2="fc2Owner"))
    params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=getattr(args, "device", 0))
    params.GenDefaultCRS(seed=args.seed)
    for r in HC.ROTS:
        params.AddCRS(r, seed=args.seed)
    kgen = mkrlwe.NewKeyGenerator(params, mkrlwe.HostSampler(np.random.default_rng(args.seed), insecure_test_only=True))
    rlkSet, rtkSet = mkrlwe.RelinearizationKeySet(params), mkrlwe.RotationKeySet()
    t0 = time.perf_counter()
    for id in sorted(set(owners.values())):
        sk = kgen.GenSecretKey(id)
        rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, kgen.GenSecretKey(id)))
        for r in HC.ROTS + [1 << i for i in range(p["logN"] - 1)]:
            rtkSet.AddRotationKey(kgen.GenRotationKey(r, sk))
    params.sync()
    keygen_s = time.perf_counter() - t0
    rng = np.random.default_rng(args.seed)
    level, N = len(p["Q"]) - 1, 1 << p["logN"]
    host_cts = {}
    def ct(id):
        host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(2)])
        c = mkckks.NewCiphertext(params, [id], level, p["scale"]).upload(host)
        host_cts[id_of(c)] = (id, host)                 # kept for the CPU-baseline leg (the oracle runs the same inference on the same inputs)
        return c
    id_of = id_builtin
    ctImage, ctKernels = ct(owners["image"]), [ct(owners["kernels"]) for _ in range(4)]
    ctFC1, ctFC2, ctB1, ctB2 = [ct(owners["fc1"]) for _ in range(8)], ct(owners["fc2"]), ct(owners["fc1"]), ct(owners["fc2"])
    ptMask_host = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"][: level - 3]])
    ptMask = mkrlwe.DeviceLimbs(params, 1, level - 3).upload(ptMask_host[None])            # resident plaintext (uploaded once)
    ev = mkckks.NewEvaluator(params)
    forks = [ev.Fork() for _ in range(max(0, args.forks))]       # extra engine contexts: the independent chains of a layer overlap
    if os.environ.get("MKHE_CNN_INTRA_OVERLAP", "0") == "0":
        # the side streams inside one operation (x / y / tensor overlap) pay off for the large launches of the headline workload;
        # here every launch is tiny, the forks provide the concurrency, and the extra event record / wait calls only cost host time
        from mkhe_kklss_amd._abi import check, lib
        for e in [ev] + forks:
            check(lib().mkhe_set_overlap(e.params.ctx, 0))
    hoisted = (ev.HoistedForm(ctImage), [ev.HoistedForm(c) for c in ctKernels], [ev.HoistedForm(c) for c in ctFC1])     # precomputation, as in the reference
    B = max(1, args.batch)
    if B > 1:
        # B images in lock step: the model ciphertexts are broadcast, every evaluator call is one launch set for the B images (no forks: the
        # batch is the concurrency)
        bev = mkckks.BatchEvaluator(params, B)
        images = [ctImage] + [ct(owners["image"]) for _ in range(B - 1)]
        bImage = mkckks.BatchCiphertext(images)
        bhoisted = (bev.HoistedForm(bImage), hoisted[1], hoisted[2])
    layer_ms = {}
    def timed(name, fn):
        params.sync(); t = time.perf_counter(); out = fn(); params.sync()
        layer_ms[name] = layer_ms.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return out
    def inference(record, E=None, img=None, hst=None, fk="default"):
        E, img, hst, fk = E or ev, img or ctImage, hst or hoisted, forks if fk == "default" else fk
        T = timed if record else (lambda name, fn: fn())
        convOut = T("Convolution", lambda: cnn.Convolution(E, rlkSet, rtkSet, img, hst[0], ctKernels, hst[1], fk))
        sq1 = T("Square1", lambda: (lambda h: (E.MulRelinHoistedNew(convOut, convOut, h, h, rlkSet)))(E.HoistedForm(convOut)))
        sq1h = T("Square1", lambda: E.HoistedForm(sq1))
        fc1 = T("FC1", lambda: cnn.FC1Layer(E, rlkSet, rtkSet, sq1, sq1h, ctFC1, hst[2], ctB1, fk))
        sq2 = T("Square2", lambda: (lambda h: E.MulRelinHoistedNew(fc1, fc1, h, h, rlkSet))(E.HoistedForm(fc1)))
        return T("FC2", lambda: cnn.FC2Layer(E, rlkSet, rtkSet, sq2, ctFC2, ctB2, ptMask, p["scale"]))
    single_inference = inference
    batch_check = None
    if B > 1:
        bforks = [bev.Fork() for _ in range(max(0, args.batch_forks))]
        for e in [bev] + bforks:
            if os.environ.get("MKHE_CNN_INTRA_OVERLAP", "0") == "0":
                check(lib().mkhe_set_overlap(e.params.ctx, 0))
        inference = lambda record: single_inference(record, bev, bImage, bhoisted, bforks or None)
    for _ in range(max(1, args.warmup)):
        out = inference(False)
    params.sync()
    graph = None
    if args.graph:
        # the whole inference recorded once into a HIP graph (the fork streams become parallel branches) and replayed with
        # one submission per step: the inputs are the resident ciphertext handles, a new image is uploaded into ctImage
        from mkhe_kklss_amd._abi import MkheError
        try:
            with params.Capture() as graph:
                out = inference(False)
            for _ in range(max(1, args.warmup)):
                graph.launch()
            params.sync()
        except MkheError as e:
            print("graph capture unavailable, issuing eagerly: %s" % e, file=sys.stderr)
            graph = None
    t0 = time.perf_counter()
    issue = 0.0
    for _ in range(args.steps):
        ti = time.perf_counter()
        if graph is not None:
            graph.launch()
        else:
            out = inference(False)
        issue += time.perf_counter() - ti
    params.sync()
    dt = time.perf_counter() - t0
    for _ in range(args.steps):                      # per-layer figures: a second, untimed-for-`value` pass with a sync per layer
        inference(True)
    roofline = roofline_leg(args, params, lambda: inference(False), p["logN"], "cnn PN14QP433 k=%d" % len(set(owners.values())))
    if B > 1:
        # every image of the batch against its own single-image inference (bit for bit); `out` becomes image 0's ciphertext for the oracle leg below
        bout = out
        same_b = []
        for k in range(B):
            ref_k = single_inference(False, ev, images[k], (ev.HoistedForm(images[k]), hoisted[1], hoisted[2]), forks)
            same_b.append(bool(ref_k.Scale == bout.cts[k].Scale and (ref_k.download() == bout.cts[k].download()).all()))
        batch_check = dict(images=B, identical_to_single_image_inference=all(same_b))
        out = bout.cts[0]
    # ---- CPU baseline (cnn/cnn_bench_test.go:12-75 on the host): the SAME inference -- same circuit (mkhe_kklss_amd/cnn.py is duck-typed over its
    # evaluator), same keys (downloaded once), same input limbs -- on the oracle through tests/oracle_evaluator.py, one host thread, then with the
    # oracle's limb loops spread over the cores; its output ciphertext is compared with the device's bit for bit
    cpu = None
    if not args.no_cpu:
        from oracle import oracle as O
        import oracle_evaluator as OE
        ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
        parties = sorted(set(owners.values()))
        rots = sorted(set(HC.ROTS + [1 << i for i in range(p["logN"] - 1)]))
        rlk_h = {id: tuple(rlkSet.GetRelinearizationKey(id).Value[j].download() for j in range(3)) for id in parties}
        rk_h = {(id, r): rtkSet.GetRotationKey(id, r).Value.download() for id in parties for r in rots}
        crs_h = {r: params.CRS[r].download() for r in rots + [-1] if r in params.CRS}
        oev = OE.OracleEvaluator(ks, p["Q"], p["scale"], rlk_h, rk_h, crs_h, p["logN"])
        H_ = lambda c: OE.OCt([host_cts[id_of(c)][0]], host_cts[id_of(c)][1], p["scale"])
        oI, oK, oF1, oF2, oB1, oB2 = H_(ctImage), [H_(c) for c in ctKernels], [H_(c) for c in ctFC1], H_(ctFC2), H_(ctB1), H_(ctB2)
        ohoist = (oev.HoistedForm(oI), [oev.HoistedForm(c) for c in oK], [oev.HoistedForm(c) for c in oF1])           # precomputation, outside the timer like the device's
        def cpu_inference():
            convOut = cnn.Convolution(oev, None, None, oI, ohoist[0], oK, ohoist[1], None)
            sq1 = (lambda h: oev.MulRelinHoistedNew(convOut, convOut, h, h, None))(oev.HoistedForm(convOut))
            fc1 = cnn.FC1Layer(oev, None, None, sq1, oev.HoistedForm(sq1), oF1, ohoist[2], oB1, None)
            sq2 = (lambda h: oev.MulRelinHoistedNew(fc1, fc1, h, h, None))(oev.HoistedForm(fc1))
            return cnn.FC2Layer(oev, None, None, sq2, oF2, oB2, ptMask_host, p["scale"])
        O.set_threads(1)
        t0 = time.perf_counter(); ref = cpu_inference(); cdt = time.perf_counter() - t0
        got = out.download()
        same = bool(got.shape == ref.host.shape and (got == ref.host).all() and out.Level() == ref.Level() and out.Scale == ref.Scale)
        nth = min(os.cpu_count() or 1, len(p["Q"]) + len(p["P"]))
        O.set_threads(nth)
        cpu_inference()
        t0 = time.perf_counter(); ref_mt = cpu_inference(); mdt = time.perf_counter() - t0
        O.set_threads(1)
        cpu = dict(value=1.0 / cdt, unit="inference/s", cores=1, kind="port",
                   sample="1 full encrypted inference (Convolution + square + FC1 + square + FC2, %d parties, PN14QP433) on 1 host thread, %.1f s" % (len(parties), cdt),
                   bit_exact_vs_gpu=same, value_limb_parallel=1.0 / mdt, cores_limb_parallel=nth,
                   limb_parallel_identical=bool((ref_mt.host == ref.host).all()))
        del rlk_h, rk_h, crs_h
    return dict(metric="cnn_inference_per_sec", value=B * args.steps / dt, unit="inference/s", n_gpus=1, steps=args.steps,
                warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps, higher_is_better=True, scaling="strong",
                vs_baseline=None, dtype="u64", data="synthetic",
                config=dict(workload="cnn encrypted inference (Convolution + square + FC1 + square + FC2, cnn/cnn.go), PN14QP433 N=2^14, "
                                     "7 Q + 2 P limbs, %d parties" % len(set(owners.values())),
                            parties=len(set(owners.values())), params="PN14QP433", seed=args.seed, batch=B, batch_check=batch_check,
                            forks=len(forks) if B == 1 else 0, chains=("forks" if forks else "lanes") if B == 1 else ("forks of the batch evaluator" if args.batch_forks > 0 else "lanes of the batch's launch sets"),
                            launch_groups_per_inference=(sum(k["launches_per_step"] for k in roofline["kernels"].values()) / B) if roofline and roofline.get("kernels") else None,
                            hip_graph=graph is not None, host_issue_ms=issue * 1e3 / args.steps, layer_ms={k: v / args.steps for k, v in layer_ms.items()}, out_level=out.Level(),
                            keygen_s=keygen_s, keys_generated=len(set(owners.values())) * (3 + len(HC.ROTS) + p["logN"] - 1)),
                roofline=roofline, cpu_baseline=cpu)


def csrc_digest():
    """sha256 over the kernel / engine sources (csrc/*.hip, *.h, name and contents, sorted): ties profiles/traffic.json to the code it profiled"""
    import glob, hashlib
    d = os.path.join(ROOT, "mkhe-kklss_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def roofline_leg(args, params, step, logN, workload, extra=None):
    """per-kernel HIP-event timing on the context stream, same steps again (side-stream overlap off: each kernel
    then runs alone, so its duration is the kernel's own and comparable with the rocprofv3 kernel trace taken with
    MKHE_NO_OVERLAP=1; `value` is measured with overlap on)"""
    from mkhe_kklss_amd._abi import check, lib
    ncls = lib().mkhe_prof_nclass()
    check(lib().mkhe_set_overlap(params.ctx, 0))
    check(lib().mkhe_prof_enable(params.ctx, 1))
    for _ in range(args.steps):
        step()
    ms = (C.c_double * ncls)()
    cnt = (C.c_long * ncls)()
    byt = (C.c_double * ncls)()
    check(lib().mkhe_prof_collect(params.ctx, ms, cnt, byt))
    check(lib().mkhe_prof_enable(params.ctx, 0))
    check(lib().mkhe_set_overlap(params.ctx, 0 if os.environ.get("MKHE_NO_OVERLAP") else 1))
    names_k = [lib().mkhe_prof_name(i).decode().replace("<N,", "<%d," % logN).replace("<N>", "<%d>" % logN)
               for i in range(ncls)]
    kernels = {}
    for i in range(ncls):
        if cnt[i]:
            kernels[names_k[i]] = dict(launches_per_step=cnt[i] / args.steps, ms_per_step=ms[i] / args.steps,
                                       avg_launch_us=1e3 * ms[i] / cnt[i],
                                       achieved_GBs=byt[i] / (ms[i] * 1e-3) / 1e9)
    # The Decompose-fused forward NTT of N = 2^15 has two forms (two-pass ntt16_fwd_kernel<true>, single-pass ntt32_fwd_kernel<true>: same bits) and the
    # engine settles on one PER LAUNCH SHAPE, so one MulRelin may run the 1792-limb launch on one and the 896-limb launch on the other: for the choice
    # of the dominant kernel -- and for its roofline figures -- the two classes are ONE kernel (times, launches and algorithmic bytes added up)
    dec = [i for i in range(ncls) if cnt[i] and names_k[i].split()[0] in ("ntt16_fwd_kernel<true>", "ntt32_fwd_kernel<true>")]
    ms_l, cnt_l, byt_l = list(ms), list(cnt), list(byt)
    if len(dec) == 2:
        a, b = sorted(dec, key=lambda i: -ms[i])
        ms_l[a] += ms[b]; cnt_l[a] += cnt[b]; byt_l[a] += byt[b]
        ms_l[b] = 0.0
    di = max((i for i in range(ncls) if cnt[i]), key=lambda i: ms_l[i])
    dom = names_k[di]
    if len(dec) == 2 and di in dec:
        dom = " + ".join("%s (%d of %d launches)" % (names_k[i].split()[0], cnt[i], cnt_l[di]) for i in sorted(dec, key=lambda i: -ms[i])) + "  (Decompose NTT, the form chosen per launch shape)"
    ms_d, cnt_d, byt_d = ms_l[di], cnt_l[di], byt_l[di]
    achieved = byt_d / (ms_d * 1e-3) / 1e9
    # HBM traffic: PMC counters cannot be read from inside this process; the figures come from the committed rocprofv3 --pmc
    # summary of the clean profile command (profiles/traffic.json, tools/profile_round.sh + tools/traffic_from_pmc.py:
    # (2*FETCH_SIZE + WRITE_SIZE) KB per launch).  They are attached only when the profiled run had the launch
    # pattern measured here: same workload, same kernel sources (csrc_sha256), and per kernel the recorded call count equals
    # launches_per_step * (steps the recorded command ran) -- otherwise `traffic` stays null.
    traffic, tnote = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        tk = tj.get("kernels", {})
        stale = tj.get("csrc_sha256") != csrc_digest()      # the PMC passes profiled other kernel sources than the ones that run here
        def pmc_bytes(name):
            key = name.split()[0]
            rec = tk.get(key) or tk.get(key.replace("[_batch]", "_batch"))
            if stale or tj.get("workload") != workload or rec is None or tj.get("steps") is None:
                return None
            # steps the recorded command ran: cold-start leg (W + K), steady-state leg (100 + 200), timed region (W + K), HIP-event leg (K)
            expect = kernels[name]["launches_per_step"] * (2 * (tj["warmup"] + tj["steps"]) + 300 + tj["steps"])
            return rec["hbm_bytes_per_launch"] if abs(rec["launches"] - expect) < 0.5 else None
        if dom in kernels:
            traffic = pmc_bytes(dom)
        elif len(dec) == 2 and not stale and tj.get("workload") == workload and tj.get("steps") is not None:
            # the Decompose NTT split over its two forms (one launch shape each): the recorded per-shape figures of the two kernels, when each of
            # them ran its shape once per step in the recorded command too
            big, small = sorted(dec, key=lambda i: -(byt[i] / cnt[i]))
            rb, rs = (tk.get(names_k[big].split()[0]) or {}).get("large"), (tk.get(names_k[small].split()[0]) or {}).get("small")
            nsteps = 2 * (tj["warmup"] + tj["steps"]) + 300 + tj["steps"]
            if rb and rs and cnt[big] == cnt[small] == args.steps and rb["launches"] == nsteps and rs["launches"] == nsteps:
                traffic = 0.5 * (rb["hbm_bytes_per_launch"] + rs["hbm_bytes_per_launch"])
        for name, k in kernels.items():
            hb = pmc_bytes(name)
            if hb is not None:
                k["hbm_bytes_per_launch_pmc"] = hb
                k["hbm_GBs_pmc"] = hb / (k["avg_launch_us"] * 1e-6) / 1e9          # what the kernel really moves: never above the peak
        if traffic is None:
            tnote = ("profiles/traffic.json was recorded for other kernel sources (csrc_sha256 differs): re-run tools/profile_round.sh" if stale
                     else "profiles/traffic.json was recorded for another workload or launch pattern")
    out = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
               frac=achieved / HBM_PEAK_GBS, traffic=traffic,
               alg_bytes_per_launch=byt_d / cnt_d, avg_launch_us=1e3 * ms_d / cnt_d,
               kernels=kernels, launches_per_step=sum(v["launches_per_step"] for v in kernels.values()))      # (engine kernels of this rank: collectives and torch copies not counted)
    if tnote:
        out["traffic_note"] = tnote
    # no byte model may claim more than the chip moves: an algorithmic figure above the HBM peak means the model counts re-reads that
    # the kernel does not make (round 2: ext_inner_group_kernel's shared operand was charged once per item)
    over = sorted(n for n, k in kernels.items() if k["achieved_GBs"] > HBM_PEAK_GBS)
    out["kernels_over_peak"] = over
    if over:
        print("bench.py: WARNING: algorithmic GB/s above the HBM peak for %s -- fix the byte model in csrc/engine.hip" % ", ".join(over), file=sys.stderr)
    if "true>" in dom.split()[0] and "fwd" in dom and extra and extra.get("decompose"):
        # the Decompose-fused NTT reads each source limb once (compulsorily; the re-reads by the nQ+nP workgroups that spread it
        # are cache hits) and writes beta * (level + 1 + nP) limbs per component: SURVEY.md 8(d) "Decompose" row, beside the
        # 16*N-per-limb NTT figure that `achieved` is computed from
        N = 1 << logN
        limbs = byt_d / cnt_d / (16.0 * N)
        comps = limbs / extra["decompose"]["limbs_per_component"]
        comp_bytes = 8.0 * N * (comps * extra["decompose"]["source_limbs_per_component"] + limbs)
        out["compulsory_bytes_per_launch"] = comp_bytes
        out["frac_compulsory"] = comp_bytes / (out["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    return out


def valu_bound_leg(reps=300):
    """roofline.valu_bound ("secondary ceiling to report"): the Decompose NTT of N = 2^15 with its loads, stores and
    LDS exchanges compiled out -- butterflies, reductions, address arithmetic, barriers only (lib/libmkhe_hip_bflyonly.so: WRONG results on purpose,
    timing only) -- beside the shipped kernel, both back to back on the launch shapes of the 4-party MulRelin (tools/ntt16_bench.py in child processes,
    HIP events on the context stream), both forward kernels.  What the vector ALU alone needs is the floor no byte-level change can move."""
    import re
    import subprocess
    libdir = os.path.join(ROOT, "mkhe-kklss_amd", "lib")
    bf = os.path.join(libdir, "libmkhe_hip_bflyonly.so")
    if not os.path.exists(bf) or os.environ.get("MKHE_LIB"):
        return None
    out = {}
    try:
        for kern, mode in (("ntt32_fwd_kernel<true>", "1"), ("ntt16_fwd_kernel<true>", "0")):
            rec = {}
            for tag, env in (("shipped_us", {}), ("butterflies_only_us", {"MKHE_LIB": bf})):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ntt16_bench.py"), str(reps)], capture_output=True, text=True, timeout=300,
                                   env=dict(os.environ, MKHE_NTT32=mode, **env))
                us = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"limbs\s+(\d+)\s+\S+\s+launches\s+\d+\s+([0-9.]+) us/launch", r.stdout)}
                if 1792 not in us or 896 not in us:
                    return dict(error="tools/ntt16_bench.py gave no timing (%s)" % (r.stderr.strip().splitlines() or ["no output"])[-1][:200])
                rec[tag] = {"1792_limbs": us[1792], "896_limbs": us[896]}
            rec["butterflies_share"] = {k: rec["butterflies_only_us"][k] / rec["shipped_us"][k] for k in rec["shipped_us"]}
            # the roofline fraction (16 N bytes per limb over 8 TB/s) the kernel would reach if its memory side cost nothing
            rec["frac_if_alu_only"] = {k: 16.0 * 32768 * int(k.split("_")[0]) / (rec["butterflies_only_us"][k] * 1e-6) / 1e9 / HBM_PEAK_GBS for k in rec["shipped_us"]}
            out[kern] = rec
    except Exception as e:                                  # a diagnostic leg: never fails the bench line
        return dict(error=str(e)[:200])
    out["how"] = "%d launches back to back per shape (clock settled under load); butterflies_only = loads, stores, LDS exchanges compiled out" % reps
    return out


def run_single(args):
    import harness as H
    from mkhe_kklss_amd import mkrlwe, mkckks
    from mkhe_kklss_amd._abi import check, lib

    pset = {"PN15QP880": H.PN15QP880, "PN14QP439": H.PN14QP439, "PN16QP1761": H.PN16QP1761}[args.params]
    k = args.parties
    names = ["user%d" % i for i in range(k)]
    params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=0)
    level = len(pset["Q"]) - 1
    device_keys = args.device_keys or args.params == "PN16QP1761"
    rlk = mkrlwe.RelinearizationKeySet(params)
    if device_keys:
        # uniform key material written by the engine's own CRS expander (mkhe_crs_expand) instead of 1 GB per party
        # of host random numbers: same distribution, nothing to upload; no host copy, hence no CPU-oracle comparison
        op0, op1 = synth_cts(pset, k, args.seed)
        data = dict(op0=op0, op1=op1)
        for i, n in enumerate(names):
            key = mkrlwe.RelinearizationKey(params, n)
            for j in range(3):
                check(lib().mkhe_crs_expand(params.ctx, args.seed, 1000 + 3 * i + j, key.Value[j].h))
            rlk.AddRelinearizationKey(key)
        params.AddCRS(-1, seed=args.seed)
        device_check = not args.no_cpu          # the oracle leg of a device-keys run: see device_keys_check below
        args.no_cpu = True
    else:
        device_check = False
        data = synth_inputs(pset, k, args.seed)
        for n, (b, d, v) in zip(names, data["rlk"]):
            rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, b, d, v))
        params.AddCRS(-1, data["u"])
    ct0 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(data["op0"])
    ct1 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(data["op1"])
    ev = mkckks.NewEvaluator(params)
    if os.environ.get("MKHE_NO_OVERLAP"):
        check(lib().mkhe_set_overlap(params.ctx, 0))

    def step():
        return ev.MulRelinNew(ct0, ct1, rlk)
    B = max(1, args.batch)
    if B > 1:
        # B MulRelin in lock step (mkhe_mul_relin_batch): input 0 is the pair above, the others are further uniform ciphertexts of the same shape
        bev = mkckks.BatchEvaluator(params, B)
        rngb = np.random.default_rng(args.seed + 7)
        def more():
            h = np.stack([np.stack([rngb.integers(0, q, 1 << pset["logN"], dtype=np.uint64) for q in pset["Q"]]) for _ in range(1 + k)])
            return mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(h)
        b0 = mkckks.BatchCiphertext([ct0] + [more() for _ in range(B - 1)])
        b1 = mkckks.BatchCiphertext([ct1] + [more() for _ in range(B - 1)])
        single_step = step
        def step():
            return bev.MulRelinNew(b0, b1, rlk)

    # ---- order of the legs (round 3).  After an idle phase (the host-side set-up above is one) this GPU takes about 150 ms of load to settle its
    # clocks (tools/ramp_probe.py: a 5-step window runs at 1.10 ms per step right after 0.2 s of idleness and at 0.92 ms 120 steps later), and
    # W + K = 23 steps last 22 ms.  Rounds 1 and 2 timed `value` first, i.e. on the ramp; that figure is still measured first and reported as
    # config.mulrelin_per_sec_cold_start.  The secondary legs then run BEFORE the timed region instead of after it, so that the contract's
    # W warm-up + K timed steps run on settled clocks like any step of a running service; nothing is skipped or shortened in the timed region.
    extras = {}
    for _ in range(args.warmup):
        res = step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    params.sync()
    extras["mulrelin_per_sec_cold_start"] = args.steps / (time.perf_counter() - t0)
    extras["timing_protocol"] = ("legs in this order: cold start (W warm-up + K timed steps right after the host-side set-up: mulrelin_per_sec_cold_start, the figure "
                                 "rounds 1-2 reported as value) -> secondary legs (Rotate / Conjugate / key generation) -> 100 untimed + 200 timed steps "
                                 "(steady state) -> W + K = the timed region of `value` -> K steps under HIP events (roofline) -> mkhe_mul_relin_batch with B = 2, 4 (40 untimed + 100 timed calls each) -> two / three "
                                 "evaluations in flight on forked contexts (100 untimed + 100 timed rounds each)")

    # ---- secondary figure: hoisted rotation of the same k-party ciphertext, hoisting included / excluded
    if not args.no_extras:
        rot = 1
        rng = np.random.default_rng(args.seed + 99)
        rks = mkrlwe.RotationKeySet()
        if device_keys:
            params.AddCRS(rot, seed=args.seed)
            for i, n in enumerate(names):
                rk = mkrlwe.RotationKey(params, rot, n)
                check(lib().mkhe_crs_expand(params.ctx, args.seed, 2000 + i, rk.Value.h))
                rks.AddRotationKey(rk)
        else:
            params.AddCRS(rot, synth_swk(pset, rng))
            for n in names:
                rks.AddRotationKey(mkrlwe.RotationKey(params, rot, n, synth_swk(pset, rng)))
        hh = ev.HoistedForm(ct0)
        cks = mkrlwe.ConjugationKeySet()
        for i, n in enumerate(names):
            ck = mkrlwe.ConjugationKey(params, n)
            check(lib().mkhe_crs_expand(params.ctx, args.seed, 3000 + i, ck.Value.h))      # uniform key material written on the device
            cks.AddConjugationKey(ck)
        params.AddCRS(-2, seed=args.seed)
        for fn, key in ((lambda: ev.RotateNew(ct0, rot, rks), "rotate_per_sec"),
                        (lambda: ev.RotateHoistedNew(ct0, rot, hh, rks), "rotate_hoisted_per_sec"),
                        (lambda: ev.ConjugateNew(ct0, cks), "conjugate_per_sec")):
            for _ in range(3):
                fn()
            params.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            params.sync()
            extras[key] = args.steps / (time.perf_counter() - t0)



        # ---- one party's relinearization key generated on the device (samples drawn on the host beforehand,
        # their upload included) and one CRS expanded from the public seed instead of uploaded
        kgen = mkrlwe.NewKeyGenerator(params, mkrlwe.HostSampler(np.random.default_rng(args.seed + 5), insecure_test_only=True))
        params.AddCRS(0)
        sk, r = kgen.GenSecretKey("user0"), kgen.GenSecretKey("user0")
        e = kgen.sampler.gaussian(3 * params.Beta(level), params.N())
        for name, fn in (("relin_keygen_per_sec", lambda: kgen.GenRelinearizationKey(sk, r, e)),
                         ("crs_expand_per_sec", lambda: params.AddCRS(7))):
            for _ in range(3):
                fn()
            params.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            params.sync()
            extras[name] = args.steps / (time.perf_counter() - t0)

    # ---- the same single-stream MulRelin over a long window (200 steps after 100 untimed ones), then the contract's timed region
    for _ in range(100):
        step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        res_s = step()
    params.sync()
    extras["mulrelin_per_sec_steady_state"] = 200 / (time.perf_counter() - t0)
    del res_s

    for _ in range(args.warmup):
        res = step()
    params.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    params.sync()
    dt = time.perf_counter() - t0
    ms_per_step = dt * 1e3 / args.steps
    value = B * args.steps / dt
    if B > 1:
        same_b = [bool((res.cts[i].download() == ev.MulRelinNew(b0.cts[i], b1.cts[i], rlk).download()).all()) for i in range(B)]
        extras["batch"] = B
        extras["batch_check"] = dict(inputs=B, identical_to_single_mulrelin=all(same_b))
        extras["mulrelin_per_sec_single_input_same_run"] = None
        for _ in range(20):
            single_step()
        params.sync()
        t1 = time.perf_counter()
        for _ in range(100):
            single_step()
        params.sync()
        extras["mulrelin_per_sec_single_input_same_run"] = 100 / (time.perf_counter() - t1)
        res = res.cts[0]

    beta = params.Beta(level)
    if pset["logN"] == 15 and params.Alpha() == 1:
        # N = 2^15: two forward kernels apply to the Decompose launches (same bits).  MKHE_NTT32 = 2 (default): per launch shape the engine times a
        # block of launches of each kernel inside this workload (after its first 64 launches) and keeps the faster one; 0 / 1 force one
        from mkhe_kklss_amd._abi import lib as _lib
        per_comp = beta * (level + 1 + len(pset["P"]))
        mode = os.environ.get("MKHE_NTT32", "2")
        extras["ntt_kernel_choice"] = dict(MKHE_NTT32=mode)
        if mode == "2":
            extras["ntt_kernel_choice"].update({str(n * per_comp): {1: "ntt32_fwd_kernel (single pass)", 0: "ntt16_fwd_kernel (two passes)", -1: "undecided"}[
                _lib().mkhe_ntt_choice(params.ctx, n * per_comp, 1)] for n in (2 * k, k)})
    roofline = roofline_leg(args, params, step, pset["logN"], "%s k=%d" % (args.params, k),
                            extra=dict(decompose=dict(limbs_per_component=beta * (level + 1 + len(pset["P"])), source_limbs_per_component=level + 1))
                            if params.Alpha() == 1 else None)
    if args.params == "PN15QP880" and not args.no_extras and roofline is not None:
        params.sync()
        vb = valu_bound_leg()
        if vb is not None:
            roofline["valu_bound"] = vb
            # the dominant kernel OUTSIDE the step: the same launch shape back to back (clock settled under its own load), both forms -- beside `frac`,
            # which is measured inside the MulRelin where the streaming kernels leave clock headroom under the power cap
            try:
                n_limbs = int(round(roofline["alg_bytes_per_launch"] / (16.0 * (1 << pset["logN"]))))
                key = "%d_limbs" % n_limbs
                b2b = {kern: 16.0 * (1 << pset["logN"]) * n_limbs / (rec["shipped_us"][key] * 1e-6) / 1e9 / HBM_PEAK_GBS
                       for kern, rec in vb.items() if isinstance(rec, dict) and "shipped_us" in rec and key in rec["shipped_us"]}
                if b2b:
                    roofline["frac_back_to_back"] = b2b
            except Exception:
                pass

    # ---- B MulRelin through mkhe_mul_relin_batch on THIS ring: on PN14QP439 in lock step (the latency-bound launches of a step serve B
    # inputs at once); on PN15QP880 with four parties IN FLIGHT since round 6 (csrc/batch.hip: the single-operation path on the context and two internal
    # ones, round robin, joined at the end of the call).  Throughput of a service that has B independent products at hand; `value` stays the single-input
    # rate.  Every output is checked against the single-input result.  (Among the last legs, like the in-flight figures below: steady clocks.)
    if B == 1 and not args.no_extras and not device_keys and args.params in ("PN15QP880", "PN14QP439"):
        ref_single = step().download()
        for Bb in (2, 4):
            bevb = mkckks.BatchEvaluator(params, Bb)
            bb0, bb1 = mkckks.BatchCiphertext([ct0] * Bb), mkckks.BatchCiphertext([ct1] * Bb)
            for _ in range(40):                  # (the internal contexts of the in-flight form settle their own kernel choice in their first launches)
                outb = bevb.MulRelinNew(bb0, bb1, rlk)
            params.sync()
            t0 = time.perf_counter()
            for _ in range(100):
                outb = bevb.MulRelinNew(bb0, bb1, rlk)
            params.sync()
            extras["mulrelin_per_sec_batch%d" % Bb] = Bb * 100 / (time.perf_counter() - t0)
            extras["batch%d_identical_to_single" % Bb] = bool(all((c.download() == ref_single).all() for c in outb.cts))
            del outb, bb0, bb1, bevb

    # ---- throughput with two / three independent MulRelin in flight (forked engine contexts: same keys and ciphertexts, a stream pair each): the
    # latency-bound stretches of one evaluation (small inverse NTTs, ModDowns) run beside the other's kernels.  Measured here, in the steady state, as
    # the LAST leg on the device (rounds 4-5 measured it among the first legs, clocks not yet settled, and read it as a loss; in front of the HIP-event
    # leg it leaves the package hotter than the timed region does: the Decompose NTT then reads 250-260 us where the rocprofv3 trace says 227)
    if B == 1 and not args.no_extras:
        evs = [ev, ev.Fork(), ev.Fork()]
        for nfl in (2, 3):
            use = evs[:nfl]
            for _ in range(100):                 # (a forked context settles its own kernel choice in its first hundred launches)
                for e in use:
                    e.MulRelinNew(ct0, ct1, rlk)
            for e in use:
                e.params.sync()
            t1 = time.perf_counter()
            for _ in range(100):
                outs = [e.MulRelinNew(ct0, ct1, rlk) for e in use]
            for e in use:
                e.params.sync()
            extras["mulrelin_per_sec_%s_in_flight" % {2: "two", 3: "three"}[nfl]] = 100 * nfl / (time.perf_counter() - t1)
            extras["in_flight_identical_to_single"] = bool(all((o.download() == res.download()).all() for o in outs))
            del outs
        del evs
        params.sync()

    # ---- device-expanded keys (PN16QP1761: 7.9 GB of key material that never exists on the host) still get an oracle check: the keys of the
    # first two parties and the CRS u are regenerated on the host from the same public seed (oracle/ora_keygen.c restates the Philox
    # expander; 7 x 340 MB instead of 25), and the engine's MulRelinNew of the two-party sub-ciphertexts -- same context, same resident
    # keys as the timed region -- is compared with the oracle bit for bit.  (The eight-party evaluation against the oracle with host keys
    # is tests/test_gpu_headline.py::test_pn16_mul_and_relin_eight_parties.)
    if device_check:
        from oracle import oracle as O
        kc = min(2, k)
        ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
        kg = O.KeyGen(ks)
        tc0 = time.perf_counter()
        sub = [0] + list(range(1, 1 + kc))
        h0, h1 = np.ascontiguousarray(data["op0"][sub]), np.ascontiguousarray(data["op1"][sub])
        c0 = mkckks.NewCiphertext(params, names[:kc], level, pset["scale"]).upload(h0)
        c1 = mkckks.NewCiphertext(params, names[:kc], level, pset["scale"]).upload(h1)
        got = ev.MulRelinNew(c0, c1, rlk).download()
        rl = {i: tuple(kg.crs_expand(args.seed, 1000 + 3 * i + j) for j in range(3)) for i in range(kc)}
        u_h = kg.crs_expand(args.seed, -1)
        ids = list(range(kc))
        nb, _ = ks.ckks_nb_rescales(level, pset["scale"] * pset["scale"], pset["scale"])
        # the same oracle run is the CPU baseline of this line: ONE thread like the single-goroutine reference, on the
        # sub-problem whose keys exist on the host -- kc parties, stated in `sample`, no extrapolation to the k parties of the timed GPU workload
        O.set_threads(1)
        tc1 = time.perf_counter()
        _, ref = ks.mul_and_relin(level, ids, h0, ids, h1, rl, u_h)
        ref = np.stack([ks.ringQ.div_round_last_many(ref[s_], nb)[0] for s_ in range(1 + kc)])
        cdt = time.perf_counter() - tc1
        nth = min(os.cpu_count() or 1, 16)
        O.set_threads(nth)
        tc2 = time.perf_counter()
        _, ref_mt = ks.mul_and_relin(level, ids, h0, ids, h1, rl, u_h)
        ref_mt = np.stack([ks.ringQ.div_round_last_many(ref_mt[s_], nb)[0] for s_ in range(1 + kc)])
        mdt = time.perf_counter() - tc2
        O.set_threads(1)
        same = bool(got.shape == ref.shape and (got == ref).all())
        extras["device_keys_check"] = dict(parties=kc, bit_exact_vs_oracle=same,
                                           keys_regenerated_on_host=3 * kc + 1, seconds=time.perf_counter() - tc0)
        # the GPU on the SAME kc-party sub-problem (same context and resident keys), so that the two figures of this object are comparable
        params.sync(); tg = time.perf_counter()
        for _ in range(10):
            ev.MulRelinNew(c0, c1, rlk)
        params.sync(); gdt = (time.perf_counter() - tg) / 10
        device_cpu = dict(value=1.0 / cdt, unit="MulRelin/s", cores=1, kind="port", parties=kc,
                          sample="1 full %d-party MulRelin (%s; the timed GPU workload has %d parties -- the keys of the first %d exist on the host too, "
                                 "regenerated from the public seed; no extrapolation) on 1 host thread, %.1f s" % (kc, args.params, k, kc, cdt),
                          bit_exact_vs_gpu=same, value_limb_parallel=1.0 / mdt, cores_limb_parallel=nth,
                          limb_parallel_identical=bool((ref_mt == ref).all()), gpu_same_subproblem_per_sec=1.0 / gdt)
        del rl, u_h, c0, c1

    # ---- CPU baseline: the oracle (single-thread C restatement of the Go path) on the same inputs
    cpu = device_cpu if device_check else None
    if not args.no_cpu:
        from oracle import oracle as O
        ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
        ids = list(range(k))
        rl = {i: data["rlk"][i] for i in ids}
        reps = max(1, args.cpu_reps)
        t0 = time.perf_counter()
        for _ in range(reps):
            _, ref = ks.mul_and_relin(level, ids, data["op0"], ids, data["op1"], rl, data["u"])
            ref = np.stack([ks.ringQ.div_round_last_many(ref[s], 1)[0] for s in range(1 + k)])
        cdt = time.perf_counter() - t0
        same = bool((res.download() == ref).all())
        # the same oracle with its limb loops (NTTs, products, MForm) spread over the host cores with OpenMP: a fairer ceiling than
        # the single goroutine of the reference; identical results
        nth = min(os.cpu_count() or 1, len(pset["Q"]) + len(pset["P"]))      # one thread per limb at most (the loops have 14..16 iterations)
        O.set_threads(nth)
        ks.mul_and_relin(level, ids, data["op0"], ids, data["op1"], rl, data["u"])          # thread pool warm-up
        t0 = time.perf_counter()
        _, ref_mt = ks.mul_and_relin(level, ids, data["op0"], ids, data["op1"], rl, data["u"])
        ref_mt = np.stack([ks.ringQ.div_round_last_many(ref_mt[s], 1)[0] for s in range(1 + k)])
        mdt = time.perf_counter() - t0
        O.set_threads(1)
        cpu = dict(value=reps / cdt, unit="MulRelin/s", cores=1, kind="port",
                   sample="%d full %d-party MulRelin (%s) on 1 host thread, %.1f s" % (reps, k, args.params, cdt),
                   bit_exact_vs_gpu=same, value_limb_parallel=1.0 / mdt, cores_limb_parallel=nth,
                   limb_parallel_identical=bool((ref_mt == ref).all()))
    return dict(metric="mkckks_mulrelin_per_sec", value=value, unit="MulRelin/s", n_gpus=1, steps=args.steps,
                warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="strong",
                vs_baseline=None, dtype="u64", data="synthetic",
                config=dict(workload="mkckks %d-party MulRelin (hoist + MulAndRelinHoisted + Rescale), %s N=2^%d, %d Q + %d P limbs"
                            % (k, args.params, pset["logN"], len(pset["Q"]), len(pset["P"])),
                            parties=k, params=args.params, seed=args.seed, key_material="device" if device_keys else "host", **extras),
                roofline=roofline, cpu_baseline=cpu)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--parties", type=int, default=4)
    ap.add_argument("--params", default="PN15QP880", choices=["PN15QP880", "PN14QP439", "PN16QP1761"],
                    help="PN15QP880 = BASELINE.json configs[1] (default); PN16QP1761 = the configs[3] ring (N = 2^16, 34 + 4 primes, "
                         "alpha = 2) on ONE GPU, keys written on the device (implies --device-keys, single GPU only)")
    ap.add_argument("--forks", type=int, default=0,
                    help="--scheme cnn: extra engine contexts through which the independent chains of a layer are issued (rounds 1-4: 7); "
                         "0 (default since round 5) = the chains are LANES of one launch set on one context (mkhe_kklss_amd/cnn.py, Evaluator.Lanes)")
    ap.add_argument("--batch", type=int, default=1,
                    help="B inputs in lock step (mkhe_*_batch entry points, mkckks.BatchEvaluator): --scheme cnn evaluates B images per step, "
                         "--scheme ckks B MulRelin per step; value counts inputs (inferences / MulRelin per second), every output is compared "
                         "with the B = 1 evaluation of the same input")
    ap.add_argument("--batch-forks", type=int, default=0, help="--scheme cnn --batch B: forked batch evaluators for the independent chains of a layer (round 4: 3); "
                                                                "0 (default since round 5): the chains are lanes of the batch's launch sets (B x n items)")
    ap.add_argument("--graph", type=int, default=0,
                    help="--scheme cnn: 1 = replay the inference from a captured HIP graph (falls back to eager issue when the loaded HIP "
                         "runtime cannot capture), 0 = issue every call eagerly (default: 3.9 ms per inference in every run; replays "
                         "came out at 3.6 or 4.5 ms depending on the run)")
    ap.add_argument("--device-keys", action="store_true",
                    help="fill keys / CRS with the engine's CRS expander instead of host random numbers (no CPU-oracle check)")
    ap.add_argument("--seed", type=int, default=0x4D4B4845)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=4)
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path even at world size 1 (testing)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary Rotate / RotateHoisted figures")
    ap.add_argument("--no-limb-leg", action="store_true",
                    help="N > 1: skip the secondary limb-sharded MulRelin figure (the headline value is always the party-sharded one)")
    ap.add_argument("--dist-sync", default="auto", choices=["auto", "stream", "host"],
                    help="N > 1, limb sharding: order the collectives on the engine's stream or through the host "
                         "(auto: an untimed probe of both after the warm-up picks the faster one)")
    ap.add_argument("--scheme", default="ckks", choices=["ckks", "bfv", "cnn"],
                    help="ckks = BASELINE.json headline metric (default); bfv = the mkbfv MulRelin line; cnn = one encrypted "
                         "CNN inference per step (cnn/cnn.go on PN14QP433; --parties 2 or 4) -- both single GPU")
    args = ap.parse_args()
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the image exports it, a
    # stripped environment may not -- set before anything initialises the HIP runtime, inherited by the ranks we launch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N rank processes, one per GPU, started BEFORE anything in this
        # process has touched the GPU; rank 0 of the children prints the JSON line on the inherited stdout
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # the ranks' stdout carries library banners (RCCL / gloo) besides rank 0's result: pass everything but the JSON line on to
        # stderr, so that stdout is the ONE line the contract asks for
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        result = None
        for line in proc.stdout:
            if line.startswith('{"metric"'):
                result = line
            else:
                sys.stderr.write(line)
        rc = proc.wait()
        if result is not None:
            sys.stdout.write(result)
            sys.stdout.flush()
        sys.exit(rc)
    json_fd = None
    if world > 1:
        # a rank of a multi-process run: the collective library prints its banner on stdout from every rank.  Everything written to
        # fd 1 from here on (C level included) goes to stderr; rank 0's JSON line is written to the real stdout at the end.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    if args.gpus > 1 or world > 1 or args.force_dist:
        import bench_dist
        out = (bench_dist.run_distributed_bfv if args.scheme == "bfv" else bench_dist.run_replicas_cnn if args.scheme == "cnn"
               else bench_dist.run_distributed)(args)
    elif args.scheme == "bfv":
        out = run_bfv(args)
    elif args.scheme == "cnn":
        out = run_cnn(args)
    else:
        out = run_single(args)
    if out is not None:
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled files of one round:  python tools/write_profiles_readme.py r4"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
J = lambda n: json.load(open(P + tag + "_" + n + ".json")) if os.path.exists(P + tag + "_" + n + ".json") else None
pl, no, ov, bf, bfp, pn, pn_no, c2, c4, p14 = (J(n) for n in ("bench_plain", "bench_noovl", "bench_ovl", "bench_bfv", "bench_bfv_plain", "bench_pn16",
                                                          "bench_pn16_noovl", "bench_cnn2", "bench_cnn4", "bench_pn14"))
tr = json.load(open(P + "traffic.json"))
stats = {r["Name"]: r for r in csv.DictReader(open(P + tag + "_kernel_stats_noovl.csv"))}


def st(name):
    for k, v in stats.items():
        if name in k:
            return float(v["AverageNs"]) / 1e3, int(v["Calls"]), float(v["MinNs"]) / 1e3, float(v["MaxNs"]) / 1e3
    return (0, 0, 0, 0)


def steps_of(j):
    """steps a bench.py command ran (round 3 order of legs): cold start (W + K), steady state (100 + 200), timed region (W + K), HIP-event leg (K)"""
    return 2 * (j["warmup"] + j["steps"]) + 300 + j["steps"]


def pmc_rec(key):
    tk = tr.get("kernels", {})
    return tk.get(key) or tk.get(key.replace("[_batch]", "_batch"))


def table(j, with_pmc):
    rows = []
    for k, v in sorted(j["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        rec = pmc_rec(k.split()[0]) if with_pmc else None
        pmc = "–"
        if rec and tr.get("steps") is not None and abs(rec["launches"] - v["launches_per_step"] * (2 * (tr["warmup"] + tr["steps"]) + 300 + tr["steps"])) < 0.5:
            pmc = "%.0f" % (rec["hbm_bytes_per_launch"] / (v["avg_launch_us"] * 1e-6) / 1e9)
        rows.append("| `%s` | %.1f | %.1f | %.3f | %.0f | %s |" % (k.split("  ")[0], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"], v["achieved_GBs"], pmc))
    return "\n".join(rows)


def whole_step():
    tot, cov_ms, all_ms = 0.0, 0.0, 0.0
    for k, v in pl["roofline"]["kernels"].items():
        rec = pmc_rec(k.split()[0])
        all_ms += v["ms_per_step"]
        if rec and abs(rec["launches"] - v["launches_per_step"] * (2 * (tr["warmup"] + tr["steps"]) + 300 + tr["steps"])) < 0.5:
            tot += rec["hbm_bytes_per_launch"] * v["launches_per_step"]; cov_ms += v["ms_per_step"]
    return tot, cov_ms / all_ms


def abl(name, limbs=1792, kernel="ntt32"):
    """us per launch of a variant in the ablation table (the file has one part per kernel: '## ntt32_fwd_kernel ...', '## ntt16_fwd_kernel ...')"""
    txt = open(P + tag + "_ntt16_ablation.txt").read()
    parts = re.split(r"^## ", txt, flags=re.M)
    txt = next((x for x in parts if x.startswith(kernel)), txt)
    m = re.search(r"== %s .*?\n(?:.*\n)*?limbs +%d .*? ([0-9.]+) us/launch" % (re.escape(name), limbs), txt)
    return float(m.group(1)) if m else float("nan")


R, Rn = pl["roofline"], no["roofline"]
DOM = "ntt32_fwd_kernel<true>" if "ntt32" in pl["roofline"].get("kernel", "") else "ntt16_fwd_kernel<true>"
ALT = "ntt32_fwd_kernel<true>"
H16K = "ntt16_fwd_kernel<true>"
avg, calls, mn, mx = st(DOM)
OTHER = H16K if DOM == ALT else ALT
oavg, ocalls, omn, omx = st(OTHER)
expect = 2 * steps_of(no)
dom = pmc_rec(DOM) or {}
cb = pl["cpu_baseline"]
C = pl["config"]
tot_bytes, cov = whole_step()
power = [l for l in open(P + tag + "_power_probe.txt").read().split("\n") if l.startswith("sample under load")]
pw = sorted(float(re.search(r"Power \(W\): ([0-9.]+)", l).group(1)) for l in power)
ck = sorted(float(re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l).group(1)) for l in power)
ub = open(P + tag + "_ubench.txt").read()
cb8, cb16, c2b8 = J("bench_cnn4_batch8"), J("bench_cnn4_batch16"), J("bench_cnn2_batch8")
plh = J("bench_plain_h32")
pl16 = J("bench_plain_h16")
alt = pmc_rec(ALT) or {}
h16rec = pmc_rec(H16K) or {}
ctx_txt = open(P + tag + "_ntt_in_context.txt").read() if os.path.exists(P + tag + "_ntt_in_context.txt") else ""
pn14b = [json.loads(l) for l in open(P + tag + "_pn14_batch.jsonl") if l.strip().startswith("{")] if os.path.exists(P + tag + "_pn14_batch.jsonl") else []
sizes = open(P + tag + "_ntt16_launch_sizes.txt").read()
def size_us(kernel_part, limbs):
    parts = re.split(r"^== ", sizes, flags=re.M)
    part = next((x for x in parts if kernel_part in x.split("\n")[0]), "")
    m = re.search(r"limbs +%d .*? ([0-9.]+) us/launch" % limbs, part)
    return float(m.group(1)) if m else float("nan")
bf2 = re.search(r"two-butterfly asm block.*", ub)
m31 = re.search(r"mm31 .*?8 waves/SIMD\s+[0-9.]+ ms\s+clock ([0-9.]+) GHz\s+([0-9.]+) ns .*?\(\s*([0-9.]+) cyc\)", ub)
m30 = re.search(r"mm30u .*?8 waves/SIMD\s+[0-9.]+ ms\s+clock ([0-9.]+) GHz\s+([0-9.]+) ns .*?\(\s*([0-9.]+) cyc\)", ub)
def ctx_line(name):
    m = re.search(r"^%s launches \d+ 1792 limbs: median ([0-9.]+) us \(min ([0-9.]+)\)\s+896 limbs: median ([0-9.]+) us \(min ([0-9.]+)\)\s+mean of both ([0-9.]+)" % name, ctx_txt, re.M)
    return tuple(float(x) for x in m.groups()) if m else (float("nan"),) * 5


c16, c32, c16b, c32b = ctx_line("h16"), ctx_line("h32"), ctx_line("h16_again"), ctx_line("h32_again")
Rh = plh["roofline"] if plh else {}
def _cyc(label, waves):
    m = re.search(label + r"\s+waves/SIMD\s+" + waves + r".*?([0-9.]+) cycles", ub)
    return m.group(1) if m else "?"
cyc_c1, cyc_h1, cyc_c4, cyc_h4 = _cyc(r"compiler's form \(nops\)", r"1\.0"), _cyc("subtract before add", r"1\.0"), _cyc(r"compiler's form \(nops\)", r"4\.0"), _cyc("subtract before add", r"4\.0")
rd = re.findall(r"^(.*?)\s+(?:streams|rows)\s+(\d+)\s+[0-9.]+ MB per launch\s+[0-9.]+ us\s+([0-9.]+) GB/s", ub, re.M)
rd_strided = [float(g) for n, s_, g in rd if "stream" in n and "nontemporal" in n]
rd_contig = [float(g) for n, s_, g in rd if "contiguous" in n and "nontemporal" in n]
nanp = (float('nan'), float('nan'))
plh_value, plh_frac = (plh["value"], Rh.get("frac", float('nan'))) if plh else nanp
pl16_value, pl16_frac = (pl16["value"], pl16["roofline"].get("frac", float('nan'))) if pl16 else nanp
rds = (min(rd_strided), max(rd_strided)) if rd_strided else nanp
rdc = (min(rd_contig), max(rd_contig)) if rd_contig else nanp
def round5_block():
    """what round 5 added to the record (the sections below keep their structure from round 4)"""
    out = ["## Round 5 at a glance", ""]
    vb = (pl["roofline"].get("valu_bound") or {})
    if vb and "error" not in vb:
        for kern in ("ntt32_fwd_kernel<true>", "ntt16_fwd_kernel<true>"):
            r = vb.get(kern)
            if r:
                out.append("* `roofline.valu_bound` in the bench line (`lib/libmkhe_hip_bflyonly.so`: the kernel with loads, stores and LDS exchanges compiled out, 300 launches back to back, "
                           "same process tree as the bench): `%s` shipped %.0f / %.0f µs (1792 / 896 limbs), **butterflies only %.0f / %.0f µs** = %.2f / %.2f of the shipped time; "
                           "with a free memory side the kernel would stand at %.2f / %.2f of the 16·N roofline." % (
                               kern, r["shipped_us"]["1792_limbs"], r["shipped_us"]["896_limbs"], r["butterflies_only_us"]["1792_limbs"], r["butterflies_only_us"]["896_limbs"],
                               r["butterflies_share"]["1792_limbs"], r["butterflies_share"]["896_limbs"], r["frac_if_alu_only"]["1792_limbs"], r["frac_if_alu_only"]["896_limbs"]))
        out.append("  The vector ALU alone is three quarters of the kernel: 0.50–0.54 of the HBM roofline is where a 64-bit modular butterfly of 12 VALU instructions puts this part, "
                   "whatever the bytes do (DESIGN.md §4.1; no rewrite of the phase structure was attempted this round).")
    if C.get("mulrelin_per_sec_batch2"):
        out.append("* B MulRelin in lock step on the headline ring (`config.mulrelin_per_sec_batch2/4`, every output identical to the single-input result: %s / %s): **%.0f / %.0f MulRelin/s** "
                   "against %.0f for one input at a time in the same run — the step is the Decompose NTT and two streaming launches at their ceilings, batching has no idle time to fill "
                   "(on PN14QP439 it has: see below)." % (C.get("batch2_identical_to_single"), C.get("batch4_identical_to_single"), C["mulrelin_per_sec_batch2"], C["mulrelin_per_sec_batch4"], pl["value"]))
    if c4:
        cc = c4["config"]
        out.append("* cnn (4 parties): **%.0f inferences/s, %.2f ms per image** (round 4: 337 / 2.97 ms; 2.15 ms before the fused kernel below) — the independent rotate → hoist → MulRelin chains of Convolution / FC1 run as LANES of one launch "
                   "set on one context (`mkhe_rotate_multi`: each lane its own Galois element and keys; `Evaluator.Lanes`), every `AddNew(x, RotateNew(x, r))` is one engine call (the add on the ModDown's "
                   "store) and the sums over a layer's products one launch (`mkhe_ct_sum`).  Kernel trace, same box (`r5_cnn4_lanes_trace_summary.txt` / `_forks_`): **213 kernels per inference instead of 447**; "
                   "the trace also shows why forks never helped: with 7 forked contexts every kernel ran ALONE (\"alone µs\" = total µs for every kernel) — small kernels of different streams do not overlap on this chip.  "
                   "Per layer (ms, a sync per layer): %s; host issue %.2f ms." % (c4["value"], c4["ms_per_step"], ", ".join("%s %.2f" % (k, v) for k, v in cc.get("layer_ms", {}).items()), cc.get("host_issue_ms", 0)))
        out.append("* ... and, second half of round 5, **one kernel from the forward sub-transforms through the products to the inverse sub-transforms** for the digits the engine decomposes for its own use "
                   "(`ext_fused_lds_kernel`, DESIGN.md §8: rotations, conjugations, step F2; launches of up to 150 limbs): 2.14 → 1.94 ms per image with 4 parties, 1.94 → 1.84 with 2, 213 → 191 kernels per inference "
                   "(`r5_fused_ab.txt`: off / forward half / both halves alternating in one call, the limit in limbs, the digit-group counts that spill; `r5_cnn4_fused_trace_summary.txt`).  "
                   "684 GPU tests green with the path forced onto every launch it can take, with and without its inverse half (`r5_switch_matrix.txt`, sets `round5_fused_*`).")
    out.append("* Measured and NOT kept (`r5_fuse_pass_ab.txt`, one call, switches library): the 2 / 3 cross stages of the small N = 2^14 NTTs as a dot product at the load (fused forward sub-transform kernel; "
               "inverse pass inside the ModDown kernels) — 627 GPU tests green with it, every line slower (cnn 449 → 392, PN14QP439 6072 → 5418, headline 1364 → 1336): 8 products and 8 loads per word cost more than the launch they save.  "
               "Thresholds of the small-launch forms (`MKHE_NTT_LDS11_MAX`, `MKHE_NTT14_MIN`): the defaults stand (± 1 %).")
    out.append("* Also measured and not kept, each in one call on the switches library: the tensor chain forked off behind the F1 kernel (`r5_tensor_late_ab.txt`: 1316–1335 → 1243–1258 MulRelin/s); "
               "small N = 2^15 launches as sixteen 2^11-point sub-transforms behind a radix-16 pass (`r5_pass16_ab.txt`: inverse launch 42 → 51 µs — those launches sum up to five members at their load, ≈ 100 MB); "
               "the streaming launches reading the limb slots last-written-first for the Infinity Cache (no effect); for the fused small-ring kernel: three / four digit groups, two digits per group and round, "
               "2^10-point blocks behind a radix-16 pass (`r5_fused_ab.txt`).")
    out.append("* PN16QP1761 line: `cpu_baseline` = the oracle on the 2-party sub-problem whose keys exist on the host (no extrapolation), `gpu_same_subproblem_per_sec` beside it (`r5_bench_pn16.json`).")
    out.append("* The product library reads two environment variables (`MKHE_NTT32`, `MKHE_POOL_GB`); every A/B switch lives in `libmkhe_hip_switches.so` (`r5_switch_matrix.txt`: the GPU suite per switch set on that build).")
    return "\n".join(out) + "\n"


txt = f'''# profiles/ — measured on MI355X (gfx950), round 5

{round5_block()}

Distilled by `tools/collect_profiles.py` from ONE `gpurun` call of `tools/profile_round.sh` (the commands are in that script; build = the commit that carries these files); this file is written by
`tools/write_profiles_readme.py`.  Files of earlier rounds (`r1*` … `r4*`) are kept for comparison.

Every profiled pass runs the same command, `MKHE_NO_OVERLAP=1 python3 bench.py --steps K --warmup W --no-cpu --no-extras` (every kernel alone on the main stream).  `bench.py` runs its legs in this order: cold-start figure (W + K steps),
300 steps of the steady-state leg, the timed region (W + K), the HIP-event leg (K) — so the dominant kernel, the Decompose-fused forward NTT `{DOM}` (twice per MulRelin: 1792 limbs for the hoisting of the 8 operand components, 896 limbs for the 4 intermediate t_i),
is launched exactly **2 × (2·(W + K) + 300 + K)** times per command.

| file | command |
|---|---|
| `{tag}_bench_plain.json` | `python3 bench.py` (no flags, no profiler: what the driver runs); `{tag}_bench_plain_h32.json` / `_h16.json`: the same with `MKHE_NTT32=1` / `=0` (the forward kernel forced), same call |
| `{tag}_kernel_stats_noovl.csv`, `{tag}_bench_noovl.json` | `MKHE_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras` (expected calls of the forward NTT: 2 × {steps_of(no)} = {expect}; recorded: {calls} + {ocalls} on the two kernels) |
| `{tag}_kernel_stats_ovl.csv`, `{tag}_bench_ovl.json` | the same with the side-stream overlap on |
| `traffic.json` | two passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of the same command with `--steps 6 --warmup 2` with the two-pass kernel forced (`MKHE_NTT32=0`: 2 × {2 * (2 + 6) + 300 + 6} = {2 * (2 * 8 + 306)} calls; recorded: {h16rec.get("launches", "?")}) and the same two passes with `MKHE_NTT32=1` (recorded: {alt.get("launches", "?")}); tied to the kernel sources by `csrc_sha256` |
| `{tag}_sq_counters.txt` | three `--pmc` passes (SQ wave / wait / instruction counters, LDS, L2 hit rate) with `--steps 4 --warmup 2` |
| `{tag}_ntt16_isa.txt`, `{tag}_ntt32_isa.txt` | `tools/ntt16_isa.py`, `tools/ntt32_isa.py`: instruction counts of the two forward kernels from the gfx950 ISA, one butterfly (one two-butterfly asm block) verbatim, code-object records |
| `{tag}_ubench.txt` | `tools/ubench/bfly30u_rate`, `bfly31_rate`, `valu_rate`, `bfly_asm_rate` (the butterfly as the compiler emits it beside the hand-scheduled two-butterfly block), and `read_bw` (round 4: what a kernel that ONLY reads reaches on this part, in the access pattern of the streaming kernels) |
| `{tag}_power_probe.txt` | `tools/power_probe.sh`: rocm-smi package power and shader clock beside 30 000 back-to-back launches of the dominant kernel |
| `{tag}_ntt16_ablation.txt` | `tools/ntt16_variants.sh`: both forward kernels re-built without their memory streams / LDS exchanges / butterflies (wrong results on purpose) and with the experiments that were not kept, 1500 launches each (steady state) |
| `{tag}_ntt16_launch_sizes.txt` | `tools/ntt16_bench.py 1500`: the Decompose NTT at 1792 / 896 / 448 / 224 limbs back to back, `MKHE_NTT32=0` and `=1`, same call |
| `{tag}_ntt_in_context.txt` | `tools/trace_ntt_in_context.sh`: rocprofv3 kernel trace of the bench command, the 1792- and 896-limb launches INSIDE the MulRelin apart, both kernels, twice |
| `{tag}_kernel_stats_bfv.csv`, `{tag}_bench_bfv*.json` | `bench.py --scheme bfv` under the profiler (overlap off) and plain |
| `{tag}_bench_pn16*.json`, `{tag}_kernel_stats_pn16.csv` | `bench.py --params PN16QP1761 --parties 8` (configs[3] ring on one GPU; the plain run carries `config.device_keys_check`) |
| `{tag}_bench_cnn2/4.json`, `{tag}_bench_cnn4_batch8/16.json`, `{tag}_bench_cnn2_batch8.json` | `bench.py --scheme cnn --parties 2/4 [--batch B]` (the unbatched lines carry `cpu_baseline`) |
| `{tag}_bench_pn14.json`, `{tag}_pn14_batch.jsonl`, `{tag}_party_sweep.jsonl` | secondary workloads; `bench.py --params PN14QP439 --batch 4/8/16` |
| `{tag}_dist_5ranks.json` | `MKHE_DIST_BACKEND=gloo MKHE_DIST_ONE_DEVICE=1 python bench.py --gpus 5 --steps 2 --warmup 1`: the N > 1 path with all five ranks on the one device of the box (functional: `matches_single_gpu`; timings meaningless; DESIGN.md §7) |
| `{tag}_gputests.txt`, `{tag}_switch_matrix.txt` | `pytest -m gpu` with the defaults, and once per switch set of `tools/switch_matrix.sh` (the single-pass kernel forced onto every N = 2^15 launch, measured choice, thresholds at 1, earlier rounds' features off) |

## Headline (BASELINE.json configs[1]): mkckks 4-party MulRelin, PN15QP880, N = 2^15, 14 Q + 2 P limbs

* **{pl["value"]:.0f} MulRelin/s** ({pl["ms_per_step"]:.3f} ms per step: hoist both operands + MulAndRelinHoisted + Rescale; round 4: 1310–1370 by box — the step itself did not change in round 5 —, round 3: 1157–1207, round 2: 1038, round 1: 763), bit-exact against the oracle on the same inputs in this very run
  (`cpu_baseline.bit_exact_vs_gpu = {cb["bit_exact_vs_gpu"]}`); CPU oracle on the GPU box's host: {cb["value"]:.2f} MulRelin/s on 1 thread, {cb.get("value_limb_parallel", 0):.2f} with its limb loops on {cb.get("cores_limb_parallel", "?")} threads.
* Same run: cold start **{C.get("mulrelin_per_sec_cold_start", 0):.0f}/s**, 200 steps after 100 untimed ones {C.get("mulrelin_per_sec_steady_state", 0):.0f}/s, two MulRelin in flight through forked contexts {C.get("mulrelin_per_sec_two_in_flight", 0):.0f}/s.
* under `rocprofv3 --kernel-trace`, overlap off: {no["value"]:.0f} MulRelin/s ({no["ms_per_step"]:.3f} ms); overlap on: {ov["value"]:.0f} MulRelin/s ({ov["ms_per_step"]:.3f} ms).
* **Box to box** the figures move by ± 3 %, and round 4 met two kinds of parts — or states of a part: the configured cap reads 1400 W on both — that differ in what the forward NTT kernels do inside the MulRelin (below): on most the single-pass
  kernel is ahead (`MKHE_NTT32=1` against `=0`, same call: 1230 MulRelin/s / `roofline.frac` 0.530 against 1232 / 0.519; 1214 / 0.532 against 1192–1199 / 0.505–0.508; 1214 / 0.521 against 1207 / 0.508), on some it is throttled (1162 / 0.463 against 1180 / 0.492; 1151 / 0.465 against 1187 / 0.512).
  (Those pairs predate the F1 fusion below, which added ≈ 100 MulRelin/s to both columns.)  Default `bench.py` with the final library on five other boxes: 1337 MulRelin/s / 0.528 and 1353 / 0.515 (1792-limb shape on the single-pass kernel, 896-limb shape on the two-pass one), 1335 / 0.498 (the same choice), 1313 / 0.491 and 1298 / 0.493 (two-pass for both).
  The engine measures which kernel gives the shorter operation in the process at hand (`MKHE_NTT32=2`, default) — this set: **{C.get("ntt_kernel_choice")}**; forced in the same call: single-pass {plh_value:.0f} / {plh_frac:.3f}, two-pass {pl16_value:.0f} / {pl16_frac:.3f}.

Per kernel class, HIP events inside `bench.py` (roofline leg, overlap off), per step.  "algorithmic GB/s" is the byte model of DESIGN.md §4 (`roofline.kernels_over_peak` = {R.get("kernels_over_peak")});
"PMC GB/s" is what the kernel really moved through the L2's memory side, (2·FETCH_SIZE + WRITE_SIZE) from `traffic.json` over the same launch pattern:

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(pl, True)}

Whole step: the kernels with a PMC column ({100 * cov:.0f} % of the kernel time) move **{tot_bytes / 1e9:.2f} GB per MulRelin** through HBM (round 3: 2.76, round 2: 3.40 GB) = {tot_bytes / 1e9 / pl["ms_per_step"]:.2f} TB/s averaged over the {pl["ms_per_step"]:.3f} ms step.  Round 4 removed a launch and three passes from it (below): the two streaming launches left
read every key once (b, d: 470 MB; v, u: 294 MB) and every hoisted digit once (h(c0), h(c1): 470 MB; h(t): 235 MB) — the compulsory bytes of the algorithm.

### Dominant kernel: the Decompose-fused forward NTT — `{DOM}` in this set (DESIGN.md §4.1)

* HIP-event average inside `bench.py`: **{R["avg_launch_us"]:.1f} µs per launch** (plain run), {Rn["avg_launch_us"]:.1f} µs in the profiled run; rocprofv3 kernel-trace of that profiled run: `{DOM}` **{avg:.1f} µs** over {calls} calls
  (min {mn:.0f}, max {mx:.0f} µs), `{OTHER}` {oavg:.1f} µs over {ocalls} calls (the engine times a block of launches of each form per launch shape before it settles, and may settle on different forms for the 1792- and the 896-limb shape): {calls} + {ocalls} = {calls + ocalls} of the expected {expect}, **{(avg * calls + oavg * ocalls) / max(1, calls + ocalls):.1f} µs** over all of them.  Round 3: 172–176 µs, round 2: 207 µs, round 1 (`ntt_fwd_kernel<15,2,true>`): 297.5 µs.
* algorithmic bytes per launch {R["alg_bytes_per_launch"] / 1e6:.1f} MB (16·N B per limb-NTT × (1792 + 896)/2 limbs) ⇒ **{R["achieved"]:.0f} GB/s = {R["frac"]:.3f} of the 8 TB/s HBM peak** (round 3: 0.49–0.52 by box, round 2: 0.42, round 1: 0.296).
  By the compulsory bytes of the fused Decompose ({R.get("compulsory_bytes_per_launch", 0) / 1e6:.0f} MB per average launch) it is {R.get("frac_compulsory", 0):.3f}.
* HBM traffic from the PMC passes (the kernel forced in them): two-pass kernel {h16rec.get("hbm_bytes_per_launch", 0) / 1e6:.1f} MB per launch (FETCH_SIZE {h16rec.get("fetch_size_kb", 0) / 1e3:.1f} MB ×2 + WRITE_SIZE {h16rec.get("write_size_kb", 0) / 1e3:.1f} MB) =
  {h16rec.get("hbm_bytes_per_launch", 0) / R["alg_bytes_per_launch"]:.2f}× the algorithmic bytes; single-pass kernel {alt.get("hbm_bytes_per_launch", 0) / 1e6:.1f} MB = {alt.get("hbm_bytes_per_launch", 0) / R["alg_bytes_per_launch"]:.2f}× (the second pass's re-read of the source is gone).  Written: the {R["alg_bytes_per_launch"] / 2e6:.0f} MB of results and nothing else (no spilled VGPR, no scratch: `{tag}_ntt16_isa.txt`, `{tag}_ntt32_isa.txt`).
* Power (`{tag}_power_probe.txt`, 30 000 launches back to back): {pw[len(pw) // 2]:.0f} W (median of the samples under load) at {ck[len(ck) // 2] / 1e3:.2f} GHz.

### The two-pass kernel `ntt16_fwd_kernel<true>` (`MKHE_NTT32=0`)

* **A vector byte load in its job walk** (and in every other H16-class forward kernel since round 3): `kb->sched[m]`, a BYTE of the kernel arguments under a dynamic index, compiles to `global_load_ubyte` + `s_waitcnt vmcnt(0)` + `v_readfirstlane`, and the wait
  stands behind every result store of the previous job.  It is a scalar dword load now (`tests/test_kernel_static.py` refuses sub-dword and dword vector loads in these files) and worth 0–2 % — same-call A/B in `{tag}_ntt16_ablation.txt`
  (`round3_schedule_byte_load`): {abl("round3_schedule_byte_load", kernel="ntt16"):.1f} against {abl("shipped", kernel="ntt16"):.1f} / {abl("shipped_again", kernel="ntt16"):.1f} µs here, {abl("round3_schedule_byte_load"):.1f} against {abl("shipped"):.1f} / {abl("shipped_again"):.1f} on the single-pass kernel; 245.1 against 244.0 and 231.2 against 230.4 in another call.  (Two calls in which this was the only code difference were 7 % apart: box to box.)
* **Steady-state ablation** (`{tag}_ntt16_ablation.txt`, 1792 limbs, µs per launch, second part of the file): shipped {abl("shipped", kernel="ntt16"):.0f}, the vector-ALU side alone {abl("no_mem_no_xchg", kernel="ntt16"):.0f}, all butterflies removed {abl("no_bfly", kernel="ntt16"):.0f}.
* `{tag}_ubench.txt`: the bare butterfly on `mm31` {m31.group(3) if m31 else "?"} cycles per wave at {m31.group(1) if m31 else "?"} GHz, on `mm30u` **{m30.group(3) if m30 else "?"} cycles at {m30.group(1) if m30 else "?"} GHz**; `{tag}_ntt16_isa.txt`: 13.6 VALU instructions per butterfly in the U-class pass body.

### The single-pass kernel `{ALT}` (`MKHE_NTT32=1`; DESIGN.md §4.1)

One 1024-thread workgroup per CU holds a whole limb (32 coefficients per thread): no stage repeated, every source word loaded once, one cross-wave exchange per limb, 12 VALU instructions per butterfly in every stage.

* **Back to back it is the faster kernel on every part** (`{tag}_ntt16_launch_sizes.txt`, 1500 launches, same call): 1792 limbs **{size_us("MKHE_NTT32=1", 1792):.1f} µs** against {size_us("MKHE_NTT32=0", 1792):.1f}; 896 limbs {size_us("MKHE_NTT32=1", 896):.1f} against {size_us("MKHE_NTT32=0", 896):.1f} (one workgroup per CU deals whole limbs: four rounds for 3.5 rounds of work).
  On a 1400 W part of this round: 229.3 / 120.4 against 241.9 / 120.9 — VERDICT r3's 235 µs for 1792 limbs is met there, its 117 µs for 896 limbs is not.
* **Inside the MulRelin it depends on the part** (`{tag}_ntt_in_context.txt`, medians of the second half of a bench run, µs, 1792 / 896 limbs / mean): two-pass {c16[0]:.1f} / {c16[2]:.1f} / {c16[4]:.1f} and {c16b[0]:.1f} / {c16b[2]:.1f} / {c16b[4]:.1f},
  single-pass {c32[0]:.1f} / {c32[2]:.1f} / {c32[4]:.1f} and {c32b[0]:.1f} / {c32b[2]:.1f} / {c32b[4]:.1f}.  On a 1400 W part (one call, before this set): single-pass 211.4 / 113.4 / 162.5 (**0.542 of the roofline**), two-pass 226.0 / 112.9 / 169.6 (0.519);
  on a part capped at 1255 W: single-pass 237.2 / 126.4 / 181.9 (0.484), two-pass 226.9 / 114.6 / 171.1 (0.515) — the two-pass kernel keeps its in-context time under the lower cap, the single-pass kernel (one workgroup per CU, every wave of a CU in the same phase) does not.
  `bench.py` in the same calls: 1230 MulRelin/s / 0.530 against 1232 / 0.519, 1214 / 0.532 against 1192–1199 / 0.505–0.508 (good parts); 1162 / 0.463 against 1180 / 0.492, 1151 / 0.465 against 1187 / 0.512 (throttling parts: they sit at 1255 W and 2.09 GHz under the kernel
  where the others reach 1400 W and 2.2–2.3 GHz, with the same configured cap; frequent in the last hours of the round).  **The engine therefore measures** (`MKHE_NTT32=2`, default): per launch shape, after its first 64 launches, a block of launches on each kernel; the period from one launch of the shape
  to the next on the GPU's clock (the whole operation) is the sample, the medians decide (`config.ntt_kernel_choice` in the bench line).  On throttling parts it settled on the two-pass kernel in every run (1196–1206 MulRelin/s / 0.50–0.51).
* **Steady-state ablation** (`{tag}_ntt16_ablation.txt`, first part, 1792 limbs, µs per launch): shipped {abl("shipped"):.0f}; no result stores {abl("no_stores"):.0f}, no source loads {abl("no_source_loads"):.0f}, no twiddle loads {abl("no_twiddle_loads"):.0f}, no LDS exchanges {abl("no_exchanges"):.0f};
  the vector-ALU side alone **{abl("butterflies_only"):.0f}**; all butterflies removed **{abl("no_butterflies"):.0f}**; butterflies and twiddle loads removed {abl("no_butterflies_no_twiddles"):.0f} (the data stream alone: {16 * 32768 * 1792 / abl("no_butterflies_no_twiddles") / 8e6:.2f} of the roofline).
  Not kept, same table: no phase priorities {abl("no_phase_priorities"):.0f}, one priority set {abl("one_priority_set"):.0f}, eight twiddle pairs in flight {abl("ring_8"):.0f}, one butterfly per asm block {abl("one_butterfly_per_asm_block"):.0f},
  the next limb's source words requested between this limb's stores (`MKHE_H32_PREFETCH=1`: 64 registers in flight across the loop's back edge, checked on the ISA by `tools/ntt32_inflight_check.py`) {abl("prefetch_next_limb"):.0f}.
* `{tag}_ubench.txt`: the U-class butterfly as one lone wave issues it — compiler's form {cyc_c1} cycles,
  the hand-scheduled two-butterfly block {cyc_h1}; with four waves per SIMD (the kernel's occupancy)
  {cyc_c4} against {cyc_h4}: 12 VALU instructions per butterfly are ≈ 49 cycles of a SIMD, 240 butterflies per thread and limb ≈ 47 000 of the ≈ 75 000 cycles a limb takes.

### Streaming kernels

**Round 4: y and step E inside the F1 kernel.**  Rounds 1–3 ran the 4-party MulAndRelin's linear algebra as three streaming launches: y = Σ_j b_j ⊙ h(c1_j) (`inner_product_kernel`: 528 MB), F1 + x (the t_i = ⟨h(c0_i), y⟩ and, as a by-product,
x = Σ_i d_i ⊙ h(c0_i): 587 MB), and the E / F2 batch (⟨h(c1_j), x⟩, ⟨h(t_i), v_i⟩, ⟨h(t_i), u⟩: 822 MB).  All of it is pointwise in the coefficient: the thread that forms the F1 products at a coefficient needs y there and nowhere else, and once it holds x[d] and
h(c1_j)[d] (which it loaded for y) step E costs it G more accumulators.  `ext_inner_xy_kernel<G0, G1, E>` (one to four parties per operand, single device; five to eight in both: `ext_inner_xy_wide_kernel`) reads the four operand families once — h(c0), d, h(c1), b: 16 sixteen-byte loads per digit and thread — and
writes the t_i and the E products; y and x are never stored, the h(c1_j) are not read a second time, and the tail batch finds its E items precomputed in its c1 slots (`ExtItem::pre`).  Same operations on the same values: the same integers (the whole GPU suite,
`MKHE_FUSE_Y=0` / `MKHE_FUSE_E=0` are the switches).  Same call: 1203–1213 → 1231–1238 MulRelin/s with y inside, → **1308–1318** with step E inside as well; `ext_inner_kernel` here: two launches of {R["kernels"]["ext_inner_kernel"]["avg_launch_us"]:.0f} µs on average.
The batched entry (`ext_inner_xy_batch_kernel<G0, G1>`, B inputs) computes x_b and y_b in the thread; step E there is still a tail item (cnn, whose MulRelins have 1 to 3 parties per operand: 330 → 338 inferences/s one image at a time, 978 → 1022 at B = 8, same call).  mkbfv runs the same kernel over its two gadgets (623–629 → 747–751 MulRelin/s, same call); five to eight parties per operand take `ext_inner_xy_wide_kernel` (PN16QP1761 with 8 parties 91–92 → 99.4 MulRelin/s, PN15QP880 with 8 parties 628 → 681).
Their algorithmic GB/s equal their PMC GB/s (the byte model charges every distinct operand once): ≈ 5.8 TB/s = 0.73 of the 8 TB/s spec.  `{tag}_ubench.txt` (`read_bw`, round 4) measures what a kernel that ONLY reads reaches on the same box: {rds[0]:.0f}–{rds[1]:.0f} GB/s in the pattern of these kernels (14–70 concurrent streams 4 MB apart, 16 bytes per lane), {rdc[0]:.0f}–{rdc[1]:.0f} GB/s with one contiguous region per workgroup (another box of the round: 5464–5875 and 6003–6075) — the streaming kernels are within 0–10 % of the read ceiling of their access pattern, not 27 % under a roofline; the contiguous pattern (digit-major tiles instead of [digit][modulus][N]) would be a re-layout of every hoisted form and key.  The ModDown launches and the small inverse NTTs are
launch-latency-bound; the Rescale no longer appears: it rides on the merged ModDown's store (`mkhe_mul_relin_rescale`, DESIGN.md §4.3).

## BASELINE.json configs[2]: mkbfv 4-party MulRelinNew, PN15QP880 BFV chain (14 Q + 14 QMul + 2 P)

* **{bfp["value"]:.0f} MulRelin/s** ({bfp["ms_per_step"]:.3f} ms per step; round 3: ≈ 620, round 2: 571, round 1: 435; round 4: y1, y2 and step E inside the F1 kernel), bit-exact against the oracle at full size (`tests/test_gpu_headline.py`, and `bench.py --scheme bfv` in every run: `bit_exact_vs_gpu = {bfp["cpu_baseline"]["bit_exact_vs_gpu"] if bfp.get("cpu_baseline") else "n/a"}`);
  cold start {bfp["config"].get("mulrelin_per_sec_cold_start", 0):.0f}/s; the reference's non-hoisted twin on its own device path (`mkhe_bfv_mul_relin_unhoisted`: every component decomposed twice, no batching) {bfp["config"].get("mulrelin_unhoisted_per_sec", 0):.0f}/s; under the profiler with overlap off {bf["value"]:.0f}/s.

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(bf, False)}
'''
sweep = P + tag + "_party_sweep.jsonl"
if os.path.exists(sweep):
    rows = [json.loads(l) for l in open(sweep) if l.strip().startswith("{")]
    if rows:
        txt += "\n## Party count (the metric is \"MulRelin/sec at n parties\"): PN15QP880, one GPU\n\n`python3 bench.py --parties k --no-cpu --device-keys --steps 20 --warmup 3`:\n\n| parties | MulRelin/s | ms per MulRelin | cold start /s | Rotate/s | RotateHoisted/s | Conjugate/s |\n|---|---|---|---|---|---|---|\n"
        for r in rows:
            c = r["config"]
            txt += "| %d | %.0f | %.3f | %.0f | %.0f | %.0f | %.0f |\n" % (c["parties"], r["value"], r["ms_per_step"], c.get("mulrelin_per_sec_cold_start", 0), c.get("rotate_per_sec", 0), c.get("rotate_hoisted_per_sec", 0), c.get("conjugate_per_sec", 0))
if p14:
    r14 = p14["roofline"]
    txt += ("\n## PN14QP439 (N = 2^14, 6 + 2 limbs), the first set of the reference's benchmark (`mkckks_benchmark_test.go:13`)\n\n4 parties: **%.0f MulRelin/s** (%.3f ms; round 2: 3567).  Its Decompose NTT runs the one-pass N = 2^14 instantiation of the H16 kernel "
            "(`ntt14_fwd_kernel<true>`, both modulus classes in one launch): %.1f µs per launch, %.2f of the roofline (round 2: two launches of the round-1 kernel per class, 0.11).\n" % (p14["value"], p14["ms_per_step"], r14["avg_launch_us"], r14["frac"]))
    if pn14b:
        txt += ("B inputs in lock step (`bench.py --params PN14QP439 --batch B`, round 4: `mkhe_mul_relin_batch`, every output identical to the single-input MulRelin): "
                + ", ".join("B = %d: **%.0f MulRelin/s** (%.2fx of the %.0f/s that one input at a time reaches in the same process, check %s)" % (
                    r["config"]["batch"], r["value"], r["value"] / r["config"]["mulrelin_per_sec_single_input_same_run"], r["config"]["mulrelin_per_sec_single_input_same_run"],
                    r["config"]["batch_check"]["identical_to_single_mulrelin"]) for r in pn14b)
                + ".  By B = 8 the step is GPU-bound (its kernels' own times add up to the step); VERDICT r3 asked for 1.5x at B = 4.\n")
if pn:
    c = pn["config"]
    txt += f'''
## BASELINE.json configs[3] ring on ONE GPU: 8-party MulRelin + hoisted Rotate, PN16QP1761 (N = 2^16, 34 Q + 4 P primes, α = 2, β = 17)

`python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2`: **{pn["value"]:.1f} MulRelin/s** ({pn["ms_per_step"]:.2f} ms per step; round 3: 89–95, round 2: 68.5, round 1: 53.8), Rotate {c.get("rotate_per_sec", 0):.0f}/s, RotateHoisted {c.get("rotate_hoisted_per_sec", 0):.0f}/s.
Round 4: y and step E inside the F1 kernel (`ext_inner_xy_wide_kernel<8>`: 91–92 → 99.4 MulRelin/s in one call; y was a 1.0 ms `inner_product_kernel<8>` launch).  Round 3 changed its Decompose (DESIGN.md §4 "N = 2^16"): `decomp_spread4_kernel` reconstructs the two-limb digits on one-round radix-2^30 products and applies the first TWO stages of the forward NTT before it stores, and the four 2^14-point
sub-transforms of every limb are single in-place passes of the H16 kernel (`ntt14_fwd_split_kernel`).  Per Decompose launch (8 components x 17 digits x 38 moduli + the x / y digits = 8058 limbs of 2^16 words = 4.2 GB): 0.99 + 2.62 ms
(cross-half stage only, two-pass 2^15-point sub-transforms out of place: 4.2 GB written by the spread, 8.4 GB read and 4.2 GB written by the NTT = 4.8 TB/s, HBM-bound) became 0.83 + 2.23 ms (4.2 + 4.2 + 4.2 GB; the spread stores at 5.1 TB/s, the NTT is bound by its
butterflies under the power cap like the N = 2^15 kernel) -- 75.5 → 81.1 MulRelin/s on one box, same call (`MKHE_SPREAD_RADIX4=0` is the A/B switch).  `r3_pn16_traffic.txt` has the PMC bytes of both paths (`tools/pn16_traffic.sh`): the largest
Decompose launch (both operands: 16 components, 5.29 GB of digits) reads 5.46 GB and writes 5.29 GB in `ntt14_fwd_split_kernel` where `ntt16_fwd_split_kernel` read 11.4 GB; the whole step moves ≈ 53 GB through HBM = 4.5 TB/s over its 11.7 ms -- this configuration is
memory-bound as a whole (`ext_inner_kernel` 2 × 6.6 GB, `inner_product_kernel<8>` 2 × 5.6 GB, the Decompose pair 2 × 12.3 GB).  The inverse launches run `ntt14_inv_kernel` + `ntt_pass4_inv_kernel` since the end of round 3 (DESIGN.md §4): 0.79 → 0.51 ms per step; x comes out of step F1 for up to sixteen parties (`ext_inner_xwide_kernel`): one `inner_product_kernel<8>` launch
and its 5.6 GB gone (87.7 → 90.4 MulRelin/s in one call; the traffic table above was recorded before these two); the 33 moduli below 2^45.67 run double-precision butterflies in `ntt14_fwd_split_kernel` (F class, +1.5–3 %).
Per kernel class (HIP events, overlap off, per step):

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(pn, False)}

The keys (7.9 GB) are written on the device by the CRS expander; `config.device_keys_check` = {c.get("device_keys_check")}: the keys of the first two parties and the CRS regenerated on the host from the same seed, the engine's two-party MulRelinNew on the resident keys against the oracle.
Bit-exactness at this ring with 8 parties against the oracle with host keys: `tests/test_gpu_headline.py::test_pn16_mul_and_relin_eight_parties` (MulAndRelin and MulAndRelinHoisted, maximum level) and `::test_pn16_rotate_hoisted_eight_parties`.
The party-sharded N > 1 run of this configuration is `python3 bench.py --gpus 8 --params PN16QP1761 --parties 8` (DESIGN.md §7 has the link model; not measurable on the single-GPU boxes of this pool).
'''
if c2 and c4:
    cb4 = c4.get("cpu_baseline") or {}
    txt += f"""
## BASELINE.json configs[4] caller on one GPU: encrypted CNN inference (cnn/cnn.go), PN14QP433

`python3 bench.py --scheme cnn --parties 2|4 --steps 20 --warmup 3`: **{c2["value"]:.0f}** inferences/s with 2 parties ({c2["ms_per_step"]:.2f} ms), **{c4["value"]:.0f}** with 4 ({c4["ms_per_step"]:.2f} ms) (round 4: 360 / 337 through forked contexts — round 5 runs the independent chains as lanes of one launch set, see the top of this file —, round 3: 350 / 343, round 2: 276 / 268).
`cpu_baseline` (round 4): the SAME inference — same circuit, keys, model and image — on the CPU oracle through `tests/oracle_evaluator.py`, one host thread: {cb4.get("value", float("nan")):.3f} inferences/s ({cb4.get("sample", "")}),
{cb4.get("value_limb_parallel", float("nan")):.2f} with the oracle's limb loops on {cb4.get("cores_limb_parallel", "?")} threads; its output ciphertext equals the device's bit for bit (`bit_exact_vs_gpu = {cb4.get("bit_exact_vs_gpu")}`).
"""
    if cb8:
        txt += ("**B images in lock step** (`--batch B`, `mkckks.BatchEvaluator`: `cnn.Inference` unchanged on batched ciphertexts; every image's output equals its own single-image inference bit for bit, `config.batch_check`): "
                "4 parties B = 8 **%.0f inferences/s** (%.2f ms per step of 8 images, host issue %.1f ms; check %s)" % (cb8["value"], cb8["ms_per_step"], cb8["config"]["host_issue_ms"], cb8["config"]["batch_check"]["identical_to_single_image_inference"]))
        if cb16:
            txt += ", B = 16 **%.0f**" % cb16["value"]
        if c2b8:
            txt += "; 2 parties B = 8 **%.0f**" % c2b8["value"]
        txt += (" — %.1fx the single-image rate of the same box (VERDICT r3: 3x of 343).  Replaying the batched inference from a HIP graph changes nothing (970 against 961 at B = 8: `hipGraphLaunch` spends on the host what the eager issue does).\n" % (cb8["value"] / c4["value"]))
    txt += "`--gpus N` runs N independent replicas.  Encrypted == plaintext logits, and the device's output ciphertext == the oracle evaluator's: `tests/test_gpu_cnn.py`.\n"
# a figure that is missing from the tracked files must fail here, not print as "nan" (round 5 shipped three paragraphs of "nan" from an empty table)
import re as _re
bad = [l[:140] for l in txt.split("\n") if _re.search(r"(?<![A-Za-z])nan(?![A-Za-z])", l)]
if bad:
    sys.exit("tools/write_profiles_readme.py: %d line(s) of profiles/README.md would carry a missing figure (nan):\n  " % len(bad) + "\n  ".join(bad[:8]))
open(P + "README.md", "w").write(txt)
print("profiles/README.md written")

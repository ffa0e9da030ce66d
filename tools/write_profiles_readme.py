#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled files of one round:  python tools/write_profiles_readme.py r3"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
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


def abl(name, limbs=1792):
    """us per launch of a variant in the ablation table"""
    txt = open(P + tag + "_ntt16_ablation.txt").read()
    m = re.search(r"== %s .*?\n(?:.*\n)*?limbs +%d .*? ([0-9.]+) us/launch" % (re.escape(name), limbs), txt)
    return float(m.group(1)) if m else float("nan")


R, Rn = pl["roofline"], no["roofline"]
avg, calls, mn, mx = st("ntt16_fwd_kernel<true>")
expect = 2 * steps_of(no)
dom = pmc_rec("ntt16_fwd_kernel<true>") or {}
cb = pl["cpu_baseline"]
C = pl["config"]
tot_bytes, cov = whole_step()
power = [l for l in open(P + tag + "_power_probe.txt").read().split("\n") if l.startswith("sample under load")]
pw = sorted(float(re.search(r"Power \(W\): ([0-9.]+)", l).group(1)) for l in power)
ck = sorted(float(re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l).group(1)) for l in power)
ub = open(P + tag + "_ubench.txt").read()
m31 = re.search(r"mm31 .*?8 waves/SIMD\s+[0-9.]+ ms\s+clock ([0-9.]+) GHz\s+([0-9.]+) ns .*?\(\s*([0-9.]+) cyc\)", ub)
m30 = re.search(r"mm30u .*?8 waves/SIMD\s+[0-9.]+ ms\s+clock ([0-9.]+) GHz\s+([0-9.]+) ns .*?\(\s*([0-9.]+) cyc\)", ub)
txt = f'''# profiles/ — measured on MI355X (gfx950), round 3

Distilled by `tools/collect_profiles.py` from ONE `gpurun` call of `tools/profile_round.sh` (the commands are in that script; build = the commit that carries these files); this file is written by
`tools/write_profiles_readme.py`.  Files of earlier rounds (`r1*`, `r2*`, `r3*`) are kept for comparison.

Every profiled pass runs the same command, `MKHE_NO_OVERLAP=1 python3 bench.py --steps K --warmup W --no-cpu --no-extras` (every kernel alone on the main stream).  Since round 3 `bench.py` runs its legs in this order: cold-start figure (W + K steps),
300 steps of the steady-state leg, the timed region (W + K), the HIP-event leg (K) — so the dominant kernel, the Decompose-fused forward NTT `ntt16_fwd_kernel<true>` (twice per MulRelin: 1792 limbs for the hoisting of the 8 operand components, 896 limbs for the 4 intermediate t_i),
is launched exactly **2 × (2·(W + K) + 300 + K)** times per command.

| file | command |
|---|---|
| `{tag}_bench_plain.json` | `python3 bench.py` (no flags, no profiler: what the driver runs) |
| `{tag}_kernel_stats_noovl.csv`, `{tag}_bench_noovl.json` | `MKHE_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras` (expected calls of the dominant kernel: 2 × {steps_of(no)} = {expect}; recorded: {calls}) |
| `{tag}_kernel_stats_ovl.csv`, `{tag}_bench_ovl.json` | the same with the side-stream overlap on |
| `traffic.json` | two passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of the same command with `--steps 6 --warmup 2` (dominant kernel: 2 × {2 * (2 + 6) + 300 + 6} = {2 * (2 * 8 + 306)} calls; recorded: {dom.get("launches", "?")}); tied to the kernel sources by `csrc_sha256` |
| `{tag}_sq_counters.txt` | three `--pmc` passes (SQ wave / wait / instruction counters, LDS, L2 hit rate) with `--steps 4 --warmup 2` |
| `{tag}_ntt16_isa.txt` | `tools/ntt16_isa.py`: instruction counts of the dominant kernel's U-class pass body from the gfx950 ISA, one butterfly verbatim, code-object record |
| `{tag}_ubench.txt` | `tools/ubench/bfly30u_rate` (the round-3 butterfly against round 2's, bare, with a correctness check against the host), `bfly31_rate`, `valu_rate` |
| `{tag}_power_probe.txt` | `tools/power_probe.sh`: rocm-smi package power and shader clock beside 30 000 back-to-back launches of the dominant kernel |
| `{tag}_ntt16_ablation.txt` | `tools/ntt16_variants.sh`: the dominant kernel re-built without its memory streams / LDS exchanges / butterflies (wrong results on purpose) and with the experiments that were not kept, 1500 launches each (steady state) |
| `{tag}_ntt16_launch_sizes.txt` | `tools/ntt16_bench.py 1500`: the Decompose NTT at 1792 / 896 / 448 / 224 limbs |
| `{tag}_kernel_stats_bfv.csv`, `{tag}_bench_bfv*.json` | `bench.py --scheme bfv` under the profiler (overlap off) and plain |
| `{tag}_pn16_traffic.txt` | `tools/pn16_traffic.sh` + `tools/pn16_traffic.py`: FETCH_SIZE / WRITE_SIZE per kernel of the PN16QP1761 8-party step, radix-4 Decompose beside `MKHE_SPREAD_RADIX4=0` |
| `{tag}_bench_pn16*.json`, `{tag}_kernel_stats_pn16.csv` | `bench.py --params PN16QP1761 --parties 8` (configs[3] ring on one GPU; the plain run carries `config.device_keys_check`) |
| `{tag}_bench_cnn2/4.json`, `{tag}_bench_pn14.json`, `{tag}_party_sweep.jsonl` | secondary workloads |

## Headline (BASELINE.json configs[1]): mkckks 4-party MulRelin, PN15QP880, N = 2^15, 14 Q + 2 P limbs

* **{pl["value"]:.0f} MulRelin/s** ({pl["ms_per_step"]:.3f} ms per step: hoist both operands + MulAndRelinHoisted + Rescale; round 2: 1038, round 1: 763), bit-exact against the oracle on the same inputs in this very run
  (`cpu_baseline.bit_exact_vs_gpu = {cb["bit_exact_vs_gpu"]}`); CPU oracle on the GPU box's host: {cb["value"]:.2f} MulRelin/s on 1 thread, {cb.get("value_limb_parallel", 0):.2f} with its limb loops on {cb.get("cores_limb_parallel", "?")} threads.
* `value` is timed on settled clocks since round 3 (the secondary legs run before the timed region, DESIGN.md §6).  Same run: cold start (what rounds 1 and 2 reported) **{C.get("mulrelin_per_sec_cold_start", 0):.0f}/s**, 200 steps after 100 untimed ones
  {C.get("mulrelin_per_sec_steady_state", 0):.0f}/s, two MulRelin in flight through forked contexts {C.get("mulrelin_per_sec_two_in_flight", 0):.0f}/s.
* under `rocprofv3 --kernel-trace`, overlap off: {no["value"]:.0f} MulRelin/s ({no["ms_per_step"]:.3f} ms); overlap on: {ov["value"]:.0f} MulRelin/s ({ov["ms_per_step"]:.3f} ms).
* **Box to box** the figures move by ± 2 % with the clock a part sustains at the 1400 W cap: the same kernel, default `bench.py`, on the boxes gpurun dealt on the last day of the round gave
  1185 MulRelin/s / `roofline.frac` 0.498 (2.09 GHz under the NTT kernel), 1197 / 0.496 (2.13 GHz), 1212 / 0.510 (2.16 GHz), 1199 / 0.512 (2.14 GHz), 1216 / 0.515, 1206 / 0.501, 1217 / 0.510, 1175 / 0.494, 1232 / 0.520, 1209 / 0.502, 1164 / 0.488 and 1174 / 0.491 (this set; the last four with the smaller LDS sub-transforms of the small launches -- on the two slowest boxes the HBM-bound `ext_inner_kernel` is 5 % slower as well, 0.2615 against 0.2493 ms per step: the parts differ in more than the core clock); on the third box the library of the commit before ran
  1206 / 0.510 in the same call -- kernel comparisons in this repository are therefore made inside one gpurun call (`MKHE_LIB=.../libmkhe_prev.so` beside the new build, or the A/B switches of DESIGN.md §6), never across calls.

Per kernel class, HIP events inside `bench.py` (roofline leg, overlap off), per step.  "algorithmic GB/s" is the byte model of DESIGN.md §4 (`roofline.kernels_over_peak` = {R.get("kernels_over_peak")}: no model claims more than the chip moves any more);
"PMC GB/s" is what the kernel really moved through the L2's memory side, (2·FETCH_SIZE + WRITE_SIZE) from `traffic.json` over the same launch pattern:

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(pl, True)}

Whole step: the kernels with a PMC column ({100 * cov:.0f} % of the kernel time) move **{tot_bytes / 1e9:.2f} GB per MulRelin** through HBM (round 2: 3.40 GB) = {tot_bytes / 1e9 / pl["ms_per_step"]:.2f} TB/s averaged over the {pl["ms_per_step"]:.3f} ms step.  The three streaming launches
are at their compulsory bytes (y: 528 MB, F1 + x: 587 MB, E / F2: 822 MB); what remains above the 1.75 GB of SURVEY.md §8(d)'s compulsory model is the second read of one operand's hoisted digits (y needs h(c1), step E needs it again once x exists — x and y depend on
each other's operands, one of the two is read twice), the x / y round trip, and the Decompose NTT's re-reads (below).

### Dominant kernel `ntt16_fwd_kernel<true>` (DESIGN.md §3 "Round 3", §4)

* HIP-event average inside `bench.py`: **{R["avg_launch_us"]:.1f} µs per launch** (plain run), {Rn["avg_launch_us"]:.1f} µs in the profiled run; rocprofv3 kernel-trace average of that profiled run: **{avg:.1f} µs** over {calls} calls
  (min {mn:.0f} = the 896-limb launches, max {mx:.0f} µs).  Round 2: 207 µs, round 1 (`ntt_fwd_kernel<15,2,true>`): 297.5 µs.
* algorithmic bytes per launch {R["alg_bytes_per_launch"] / 1e6:.1f} MB (16·N B per limb-NTT × (1792 + 896)/2 limbs) ⇒ **{R["achieved"]:.0f} GB/s = {R["frac"]:.3f} of the 8 TB/s HBM peak** (round 2: 0.42, round 1: 0.296).
  By the compulsory bytes of the fused Decompose ({R.get("compulsory_bytes_per_launch", 0) / 1e6:.0f} MB per average launch) it is {R.get("frac_compulsory", 0):.3f}.
* HBM traffic from the PMC passes: {dom.get("hbm_bytes_per_launch", 0) / 1e6:.1f} MB per launch (FETCH_SIZE {dom.get("fetch_size_kb", 0) / 1e3:.1f} MB ×2 + WRITE_SIZE {dom.get("write_size_kb", 0) / 1e3:.1f} MB) =
  {dom.get("hbm_bytes_per_launch", 0) / R["alg_bytes_per_launch"]:.2f}× the algorithmic bytes.  Written: the {R["alg_bytes_per_launch"] / 2e6:.0f} MB of results and nothing else (no spilled VGPR, no scratch: `{tag}_ntt16_isa.txt`; round 2 wrote 403 MB, the first build of round 3 396 MB).  Read: the source limbs in both passes
  (each is spread under 16 moduli) and the twiddle pairs — an XCD's 4 MiB L2 holds neither its 14 source limbs (3.5 MB) plus the tables of the 4–5 moduli in flight (1.5 MB), so part of the re-reads come from the Infinity Cache.
* **It runs at the package power cap** (`{tag}_power_probe.txt`): {pw[len(pw) // 2]:.0f} W (median of the samples under load; cap 1400 W, idle 236 W) at {ck[len(ck) // 2] / 1e3:.2f} GHz.  After an idle phase the clocks need ≈ 150 ms to settle: ten back-to-back launches (round 2's tables)
  measure 355 µs for 1792 limbs, 1500 launches **{abl("shipped"):.0f} µs** = {abl("shipped") / 1792:.3f} µs per limb ({R["avg_launch_us"] / 1344:.3f} µs per limb inside the MulRelin, whose launches alternate with memory-bound kernels).
* **Steady-state ablation of the shipped kernel** (`{tag}_ntt16_ablation.txt`, 1792 limbs, µs per launch): shipped {abl("shipped"):.0f}; no result stores {abl("no_store"):.0f}, no source loads {abl("no_src"):.0f}, no per-lane twiddle loads {abl("no_tw"):.0f},
  no LDS exchanges {abl("no_xchg"):.0f}; no memory stream at all {abl("no_mem"):.0f}; neither memory nor exchanges — the vector-ALU side alone — **{abl("no_mem_no_xchg"):.0f}** ({16 * 32768 * 1792 / abl("no_mem_no_xchg") / 8e6:.2f} of the roofline); all butterflies removed — the memory / LDS side alone — **{abl("no_bfly"):.0f}**;
  exchanges and barriers alone {abl("skeleton"):.0f}.  0.50 of the roofline would be 235 µs: above both floors, but only with 87 % of either side hidden under the other; the kernel hides about two thirds.  Experiments on hiding more, same table:
  one memory round trip per pass (`pipelined_loads`) {abl("pipelined_loads"):.0f}, parking instead of recomputing stage 0 (`force_park`) {abl("force_park"):.0f}, phase D on the one-round product (`phase_d_one_round`, 10 spilled VGPRs) {abl("phase_d_one_round"):.0f}.
  LDS writes as plain `ds_write_b32` (`plain_lds_writes`) {abl("plain_lds_writes"):.0f}, no raised wave priority in front of the cross-wave barriers (`no_priority`) {abl("no_priority"):.0f}, the last LDS reads of a re-distribution flowing into the next phase (`flowing_lds_reads`) {abl("flowing_lds_reads"):.0f}.
  What round 3 changed: the U class off (`MKHE_H16_UCLASS=0`) {abl("U class off"):.0f}, the round-2 reduction schedule for the 59/60-bit primes {abl("round-2 reduction schedule"):.0f}.
* `{tag}_ubench.txt`: the bare butterfly on `mm31` (round 2) {m31.group(3) if m31 else "?"} cycles per wave at {m31.group(1) if m31 else "?"} GHz = {m31.group(2) if m31 else "?"} ns, on `mm30u` (round 3: unsigned low data digit, no fix-up instructions) **{m30.group(3) if m30 else "?"} cycles at {m30.group(1) if m30 else "?"} GHz = {m30.group(2) if m30 else "?"} ns**
  (65 536 products checked against the host first).  `{tag}_ntt16_isa.txt`: 13.6 VALU instructions per butterfly in the U-class pass body of the shipped code object (round 2: 17.7).

### Streaming kernels

`inner_product_kernel<4>` {R["kernels"]["inner_product_kernel"]["avg_launch_us"]:.0f} µs per launch, `ext_inner_kernel` {R["kernels"]["ext_inner_kernel"]["avg_launch_us"]:.0f} µs (two launches: F1 + x, and the E / F2 batch): 16-byte lanes, non-temporal loads for every operand read once per launch, items that share a key computed by one thread.
Their algorithmic GB/s now equal their PMC GB/s (the byte model charges every distinct operand once): ≈ 5.8 TB/s = 0.73 of the 8 TB/s spec, 0.92 of the 6.29 TB/s that MI355X_MICROARCH.md measures for a float4 copy.  The ModDown launches and the small inverse NTTs are
launch-latency-bound; the Rescale no longer appears: it rides on the merged ModDown's store (`mkhe_mul_relin_rescale`, DESIGN.md §4 "Fused Rescale").

## BASELINE.json configs[2]: mkbfv 4-party MulRelinNew, PN15QP880 BFV chain (14 Q + 14 QMul + 2 P)

* **{bfp["value"]:.0f} MulRelin/s** ({bfp["ms_per_step"]:.3f} ms per step; round 2: 571, round 1: 435), bit-exact against the oracle at full size (`tests/test_gpu_headline.py`, and `bench.py --scheme bfv` in every run: `bit_exact_vs_gpu = {bfp["cpu_baseline"]["bit_exact_vs_gpu"] if bfp.get("cpu_baseline") else "n/a"}`);
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
if pn:
    c = pn["config"]
    txt += f'''
## BASELINE.json configs[3] ring on ONE GPU: 8-party MulRelin + hoisted Rotate, PN16QP1761 (N = 2^16, 34 Q + 4 P primes, α = 2, β = 17)

`python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2`: **{pn["value"]:.1f} MulRelin/s** ({pn["ms_per_step"]:.2f} ms per step; round 2: 68.5, round 1: 53.8), Rotate {c.get("rotate_per_sec", 0):.0f}/s, RotateHoisted {c.get("rotate_hoisted_per_sec", 0):.0f}/s.
Round 3 changed its Decompose (DESIGN.md §4 "N = 2^16"): `decomp_spread4_kernel` reconstructs the two-limb digits on one-round radix-2^30 products and applies the first TWO stages of the forward NTT before it stores, and the four 2^14-point
sub-transforms of every limb are single in-place passes of the H16 kernel (`ntt14_fwd_split_kernel`).  Per Decompose launch (8 components x 17 digits x 38 moduli + the x / y digits = 8058 limbs of 2^16 words = 4.2 GB): 0.99 + 2.62 ms
(cross-half stage only, two-pass 2^15-point sub-transforms out of place: 4.2 GB written by the spread, 8.4 GB read and 4.2 GB written by the NTT = 4.8 TB/s, HBM-bound) became 0.83 + 2.23 ms (4.2 + 4.2 + 4.2 GB; the spread stores at 5.1 TB/s, the NTT is bound by its
butterflies under the power cap like the N = 2^15 kernel) -- 75.5 → 81.1 MulRelin/s on one box, same call (`MKHE_SPREAD_RADIX4=0` is the A/B switch).  `{tag}_pn16_traffic.txt` has the PMC bytes of both paths (`tools/pn16_traffic.sh`): the largest
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
    txt += f'''
## BASELINE.json configs[4] caller on one GPU: encrypted CNN inference (cnn/cnn.go), PN14QP433

`python3 bench.py --scheme cnn --parties 2|4 --steps 20 --warmup 3`: **{c2["value"]:.0f}** inferences/s with 2 parties ({c2["ms_per_step"]:.2f} ms), **{c4["value"]:.0f}** with 4 ({c4["ms_per_step"]:.2f} ms) (round 2: 276 / 268).
`--gpus N` runs N independent replicas.  Encrypted == plaintext logits, synthetic model and the reference's trained weights: `tests/test_gpu_cnn.py`.
'''
open(P + "README.md", "w").write(txt)
print("profiles/README.md written")

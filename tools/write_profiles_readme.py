#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled files of one round:  python tools/write_profiles_readme.py r4"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
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
DOM = "ntt32_fwd_kernel<true>"
avg, calls, mn, mx = st(DOM)
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
txt = f'''# profiles/ — measured on MI355X (gfx950), round 4

Distilled by `tools/collect_profiles.py` from ONE `gpurun` call of `tools/profile_round.sh` (the commands are in that script; build = the commit that carries these files); this file is written by
`tools/write_profiles_readme.py`.  Files of earlier rounds (`r1*`, `r2*`, `r3*`) are kept for comparison.

Every profiled pass runs the same command, `MKHE_NO_OVERLAP=1 python3 bench.py --steps K --warmup W --no-cpu --no-extras` (every kernel alone on the main stream).  `bench.py` runs its legs in this order: cold-start figure (W + K steps),
300 steps of the steady-state leg, the timed region (W + K), the HIP-event leg (K) — so the dominant kernel, the Decompose-fused forward NTT `{DOM}` (twice per MulRelin: 1792 limbs for the hoisting of the 8 operand components, 896 limbs for the 4 intermediate t_i),
is launched exactly **2 × (2·(W + K) + 300 + K)** times per command.

| file | command |
|---|---|
| `{tag}_bench_plain.json` | `python3 bench.py` (no flags, no profiler: what the driver runs) |
| `{tag}_kernel_stats_noovl.csv`, `{tag}_bench_noovl.json` | `MKHE_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras` (expected calls of the dominant kernel: 2 × {steps_of(no)} = {expect}; recorded: {calls}) |
| `{tag}_kernel_stats_ovl.csv`, `{tag}_bench_ovl.json` | the same with the side-stream overlap on |
| `traffic.json` | two passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of the same command with `--steps 6 --warmup 2` (dominant kernel: 2 × {2 * (2 + 6) + 300 + 6} = {2 * (2 * 8 + 306)} calls; recorded: {dom.get("launches", "?")}); tied to the kernel sources by `csrc_sha256` |
| `{tag}_sq_counters.txt` | three `--pmc` passes (SQ wave / wait / instruction counters, LDS, L2 hit rate) with `--steps 4 --warmup 2` |
| `{tag}_ntt32_isa.txt` | `tools/ntt32_isa.py`: instruction counts of the dominant kernel from the gfx950 ISA, one two-butterfly asm block verbatim, code-object record (`{tag}_ntt16_isa.txt`: the same for the H16 kernel's pass body) |
| `{tag}_ubench.txt` | `tools/ubench/bfly30u_rate`, `bfly31_rate`, `valu_rate`, and `bfly_asm_rate` (round 4: the butterfly as the compiler emits it beside the hand-scheduled two-butterfly block, 1 / 4 / 8 waves per SIMD) |
| `{tag}_power_probe.txt` | `tools/power_probe.sh`: rocm-smi package power and shader clock beside 30 000 back-to-back launches of the dominant kernel |
| `{tag}_ntt16_ablation.txt` | `tools/ntt16_variants.sh`: the single-pass kernel (and, second part, the H16 kernel) re-built without its memory streams / LDS exchanges / butterflies (wrong results on purpose) and with the experiments that were not kept, 1500 launches each (steady state) |
| `{tag}_ntt16_launch_sizes.txt` | `tools/ntt16_bench.py 1500`: the Decompose NTT at 1792 / 896 / 448 / 224 limbs, default build and `MKHE_NTT32=0`, same call |
| `{tag}_kernel_stats_bfv.csv`, `{tag}_bench_bfv*.json` | `bench.py --scheme bfv` under the profiler (overlap off) and plain |
| `{tag}_bench_pn16*.json`, `{tag}_kernel_stats_pn16.csv` | `bench.py --params PN16QP1761 --parties 8` (configs[3] ring on one GPU; the plain run carries `config.device_keys_check`) |
| `{tag}_bench_cnn2/4.json`, `{tag}_bench_cnn4_batch8/16.json`, `{tag}_bench_cnn2_batch8.json` | `bench.py --scheme cnn --parties 2/4 [--batch B]` (the unbatched lines carry `cpu_baseline`) |
| `{tag}_bench_pn14.json`, `{tag}_pn14_batch.jsonl`, `{tag}_party_sweep.jsonl` | secondary workloads; `bench.py --params PN14QP439 --batch 4/8/16` |
| `{tag}_dist_5ranks.json` | `bench.py --gpus 5` with all ranks on the one device of the box (DESIGN.md §7) |
| `{tag}_gputests*.txt`, `{tag}_switch_matrix.txt` | `pytest -m gpu` with the defaults and with the single-pass kernel forced onto every N = 2^15 launch (`MKHE_NTT32_MIN=1`) |

## Headline (BASELINE.json configs[1]): mkckks 4-party MulRelin, PN15QP880, N = 2^15, 14 Q + 2 P limbs

* **{pl["value"]:.0f} MulRelin/s** ({pl["ms_per_step"]:.3f} ms per step: hoist both operands + MulAndRelinHoisted + Rescale; round 3: 1157–1207, round 2: 1038, round 1: 763), bit-exact against the oracle on the same inputs in this very run
  (`cpu_baseline.bit_exact_vs_gpu = {cb["bit_exact_vs_gpu"]}`); CPU oracle on the GPU box's host: {cb["value"]:.2f} MulRelin/s on 1 thread, {cb.get("value_limb_parallel", 0):.2f} with its limb loops on {cb.get("cores_limb_parallel", "?")} threads.
* Same run: cold start **{C.get("mulrelin_per_sec_cold_start", 0):.0f}/s**, 200 steps after 100 untimed ones {C.get("mulrelin_per_sec_steady_state", 0):.0f}/s, two MulRelin in flight through forked contexts {C.get("mulrelin_per_sec_two_in_flight", 0):.0f}/s.
* under `rocprofv3 --kernel-trace`, overlap off: {no["value"]:.0f} MulRelin/s ({no["ms_per_step"]:.3f} ms); overlap on: {ov["value"]:.0f} MulRelin/s ({ov["ms_per_step"]:.3f} ms).
* **Box to box** the figures move by ± 2 % (round 3's list: 1164–1232 MulRelin/s, `roofline.frac` 0.488–0.520 for one build).  Round 4's A/B runs, each pair inside ONE gpurun call (`MKHE_NTT32=0` is the switch): single-pass kernel 1213 / 0.525, 1218 / 0.531, 1215 / 0.527
  against 1185 / 0.508, 1207 / 0.515, 1188 / 0.510 for the H16 kernel; this set: {pl["value"]:.0f} / {R["frac"]:.3f}.

Per kernel class, HIP events inside `bench.py` (roofline leg, overlap off), per step.  "algorithmic GB/s" is the byte model of DESIGN.md §4 (`roofline.kernels_over_peak` = {R.get("kernels_over_peak")});
"PMC GB/s" is what the kernel really moved through the L2's memory side, (2·FETCH_SIZE + WRITE_SIZE) from `traffic.json` over the same launch pattern:

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(pl, True)}

Whole step: the kernels with a PMC column ({100 * cov:.0f} % of the kernel time) move **{tot_bytes / 1e9:.2f} GB per MulRelin** through HBM (round 3: 2.76, round 2: 3.40 GB) = {tot_bytes / 1e9 / pl["ms_per_step"]:.2f} TB/s averaged over the {pl["ms_per_step"]:.3f} ms step.  The three streaming launches
are at their compulsory bytes (y: 528 MB, F1 + x: 587 MB, E / F2: 822 MB).

### Dominant kernel `{DOM}` (DESIGN.md §4 "The single-pass kernel")

* HIP-event average inside `bench.py`: **{R["avg_launch_us"]:.1f} µs per launch** (plain run), {Rn["avg_launch_us"]:.1f} µs in the profiled run; rocprofv3 kernel-trace average of that profiled run: **{avg:.1f} µs** over {calls} calls
  (min {mn:.0f} = the 896-limb launches, max {mx:.0f} µs).  Round 3 (`ntt16_fwd_kernel<true>`): 172–176 µs, round 2: 207 µs, round 1 (`ntt_fwd_kernel<15,2,true>`): 297.5 µs.
* algorithmic bytes per launch {R["alg_bytes_per_launch"] / 1e6:.1f} MB (16·N B per limb-NTT × (1792 + 896)/2 limbs) ⇒ **{R["achieved"]:.0f} GB/s = {R["frac"]:.3f} of the 8 TB/s HBM peak** (round 3: 0.50–0.51, round 2: 0.42, round 1: 0.296).
  By the compulsory bytes of the fused Decompose ({R.get("compulsory_bytes_per_launch", 0) / 1e6:.0f} MB per average launch) it is {R.get("frac_compulsory", 0):.3f}.
* HBM traffic from the PMC passes: {dom.get("hbm_bytes_per_launch", 0) / 1e6:.1f} MB per launch (FETCH_SIZE {dom.get("fetch_size_kb", 0) / 1e3:.1f} MB ×2 + WRITE_SIZE {dom.get("write_size_kb", 0) / 1e3:.1f} MB) =
  {dom.get("hbm_bytes_per_launch", 0) / R["alg_bytes_per_launch"]:.2f}× the algorithmic bytes (H16, round 3: 0.93×: the second pass's re-read of the source is gone).  Written: the {R["alg_bytes_per_launch"] / 2e6:.0f} MB of results and nothing else (no scratch: `{tag}_ntt32_isa.txt`).  Read: every source limb once per modulus it is spread under (mostly L2 hits) and 512 KB of twiddle pairs per limb-NTT —
  as many bytes as the limb itself moves; they come from the L2 / Infinity Cache (16 moduli × 1 MB of tables).
* Back to back (`{tag}_ntt16_launch_sizes.txt`, 1500 launches, same call): 1792 limbs **{size_us("default", 1792):.1f} µs** against {size_us("MKHE_NTT32=0", 1792):.1f} for the H16 kernel; 896 limbs {size_us("default", 896):.1f} against {size_us("MKHE_NTT32=0", 896):.1f} (one workgroup per CU deals whole limbs: four rounds for 3.5 rounds of work);
  448 and 224 limbs stay on the H16 kernel ({size_us("default", 448):.1f} / {size_us("default", 224):.1f} µs).  VERDICT r3's 235 / 117 µs are not reached.
* Power (`{tag}_power_probe.txt`): {pw[len(pw) // 2]:.0f} W (median of the samples under load; cap 1400 W) at {ck[len(ck) // 2] / 1e3:.2f} GHz — the H16 kernel sat at the cap at 2.09 GHz.
* **Steady-state ablation** (`{tag}_ntt16_ablation.txt`, 1792 limbs, µs per launch): shipped {abl("shipped"):.0f}; no result stores {abl("no_stores"):.0f}, no source loads {abl("no_source_loads"):.0f}, no twiddle loads {abl("no_twiddle_loads"):.0f}, no LDS exchanges {abl("no_exchanges"):.0f};
  the vector-ALU side alone (no loads, stores, exchanges) **{abl("butterflies_only"):.0f}**; all butterflies removed **{abl("no_butterflies"):.0f}**; butterflies and twiddle loads removed {abl("no_butterflies_no_twiddles"):.0f} (the data stream alone: {16 * 32768 * 1792 / abl("no_butterflies_no_twiddles") / 8e6:.2f} of the roofline).
  Both sides are as long as the whole: the kernel no longer waits for one of them, it is as long as its butterflies AND as long as its memory instructions with their waits — the twiddle pairs are half of the second (per limb: 31 per-lane 16-byte loads + 31 `ds_read_b128` per thread, against 32 + 32 eight-byte data accesses).
  Not kept, same table: no phase priorities {abl("no_phase_priorities"):.0f}, one priority set {abl("one_priority_set"):.0f}, eight twiddle pairs in flight {abl("ring_8"):.0f}, one butterfly per asm block {abl("one_butterfly_per_asm_block"):.0f}.
  Second part of the file, the H16 kernel in the same call: shipped {abl("shipped", kernel="ntt16"):.0f}, vector-ALU side alone {abl("no_mem_no_xchg", kernel="ntt16"):.0f}, no butterflies {abl("no_bfly", kernel="ntt16"):.0f}.
* `{tag}_ubench.txt`: the U-class butterfly as one lone wave issues it — compiler's form {re.search(r"compiler's form \(nops\)\s+waves/SIMD\s+1\.0.*?([0-9.]+) cycles", ub).group(1) if re.search(r"compiler's form \(nops\)\s+waves/SIMD\s+1\.0.*?([0-9.]+) cycles", ub) else "?"} cycles,
  the hand-scheduled two-butterfly block {re.search(r"subtract before add\s+waves/SIMD\s+1\.0.*?([0-9.]+) cycles", ub).group(1) if re.search(r"subtract before add\s+waves/SIMD\s+1\.0.*?([0-9.]+) cycles", ub) else "?"}; with four waves per SIMD (the kernel's occupancy)
  {re.search(r"compiler's form \(nops\)\s+waves/SIMD\s+4\.0.*?([0-9.]+) cycles", ub).group(1) if re.search(r"compiler's form \(nops\)\s+waves/SIMD\s+4\.0.*?([0-9.]+) cycles", ub) else "?"} against {re.search(r"subtract before add\s+waves/SIMD\s+4\.0.*?([0-9.]+) cycles", ub).group(1) if re.search(r"subtract before add\s+waves/SIMD\s+4\.0.*?([0-9.]+) cycles", ub) else "?"}: 12 VALU instructions per butterfly are ≈ 49 cycles of a SIMD, 240 butterflies per thread and limb ≈ 47 000 of the ≈ 80 000 cycles a limb takes.

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

#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled files of one round:  python tools/write_profiles_readme.py r6
Every figure comes from a tracked file of profiles/; a figure that is missing stops the script (no "nan" in the README).
(The round-5 README, with its narrative of rounds 1-5, is profiles/README_r5.md; its generator tools/write_profiles_readme_r5.py.)"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"


def need(path):
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        sys.exit("tools/write_profiles_readme.py: %s is missing or empty" % path)
    return path


def J(n, required=True):
    f = P + tag + "_" + n + ".json"
    if not os.path.exists(f):
        if required:
            sys.exit("tools/write_profiles_readme.py: %s is missing" % f)
        return None
    return json.loads(open(f).read().strip().splitlines()[-1])


pl, no, ov = J("bench_plain"), J("bench_noovl"), J("bench_ovl")
plh, pl16 = J("bench_plain_h32"), J("bench_plain_h16")
tr = json.load(open(need(P + "traffic.json")))
stats = {r["Name"]: r for r in csv.DictReader(open(need(P + tag + "_kernel_stats_noovl.csv")))}
R, C, cb = pl["roofline"], pl["config"], pl["cpu_baseline"]
for k in ("frac", "traffic", "frac_back_to_back", "valu_bound", "launches_per_step"):
    if R.get(k) is None:
        sys.exit("tools/write_profiles_readme.py: roofline.%s is null in %s_bench_plain.json (record the bench lines AFTER traffic.json: tools/collect_profiles.py says how)" % (k, tag))


def steps_of(j):
    return 2 * (j["warmup"] + j["steps"]) + 300 + j["steps"]


def st(name):
    for k, v in stats.items():
        if name in k:
            return float(v["AverageNs"]) / 1e3, int(v["Calls"]), float(v["MinNs"]) / 1e3, float(v["MaxNs"]) / 1e3
    sys.exit("tools/write_profiles_readme.py: no kernel named %s in %s_kernel_stats_noovl.csv" % (name, tag))


def pmc_rec(key):
    tk = tr.get("kernels", {})
    return tk.get(key) or tk.get(key.replace("[_batch]", "_batch"))


def table(j):
    rows = []
    for k, v in sorted(j["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        pmc = "%.0f MB, %.0f GB/s" % (v["hbm_bytes_per_launch_pmc"] / 1e6, v["hbm_GBs_pmc"]) if "hbm_GBs_pmc" in v else "–"
        rows.append("| `%s` | %.0f | %.1f | %.1f | %.0f | %s |" % (k.split("  ")[0], v["launches_per_step"], v["avg_launch_us"], 1e3 * v["ms_per_step"], v["achieved_GBs"], pmc))
    return "\n".join(rows)


def step_bytes(j):
    tot, cov_ms, all_ms = 0.0, 0.0, 0.0
    for k, v in j["roofline"]["kernels"].items():
        all_ms += v["ms_per_step"]
        if "hbm_bytes_per_launch_pmc" in v:
            tot += v["hbm_bytes_per_launch_pmc"] * v["launches_per_step"]; cov_ms += v["ms_per_step"]
    return tot, cov_ms / all_ms


nsteps = 2 * (tr["warmup"] + tr["steps"]) + 300 + tr["steps"]
all_bytes = 0.0
for k, v in tr["kernels"].items():
    if "<" not in k and any(x.startswith(k + "<") for x in tr["kernels"]):
        continue                                     # (the un-templated alias of a templated name)
    if k in ("ntt32_fwd_kernel<true>", "ntt16_fwd_kernel<true>") and k != R["kernel"].split()[0]:
        continue                                     # (the other form of the dominant kernel, from the second pair of passes)
    all_bytes += v["launches"] / nsteps * v["hbm_bytes_per_launch"]
m_unf = re.search(r"per step: ([0-9.]+) GB", open(need(P + tag + "_pmc_unfused.txt")).read())
if not m_unf:
    sys.exit("tools/write_profiles_readme.py: no per-step figure in %s_pmc_unfused.txt" % tag)
unf_bytes = float(m_unf.group(1))
f2k = [v for k, v in pl["roofline"]["kernels"].items() if k.startswith("ntt16_f2_kernel")][0]
f2_pmc, f2_alg = f2k["hbm_bytes_per_launch_pmc"], f2k["achieved_GBs"] * 1e9 * f2k["avg_launch_us"] * 1e-6
tail_us = sum(1e3 * v["ms_per_step"] for k, v in pl["roofline"]["kernels"].items() if k.split("<")[0].split()[0] in ("ntt_inv_kernel", "moddown[_batch]_kernel", "tensor_kernel", "ntt_fwd_kernel"))
DOM = R["kernel"].split()[0]
davg, dcalls, dmn, dmx = st(DOM.split("<")[0])
f2avg, f2calls, f2mn, f2mx = st("ntt16_f2_kernel")
xyavg, xycalls, _, _ = st("ext_inner_xy_kernel")
tot_bytes, cov = step_bytes(pl)
vb = R["valu_bound"]
b2b = R["frac_back_to_back"]

# ---- the fused / unfused A/B and the variants of the fused kernel
ab_txt = open(need(P + tag + "_f2_fused_ab.txt")).read()
ab = re.findall(r"== MKHE_F2_FUSED=(\d) \(switches library\)\n([0-9.]+) MulRelin/s ([0-9.]+) cold (\{.*\})", ab_txt)
if len(ab) < 4:
    sys.exit("tools/write_profiles_readme.py: %s_f2_fused_ab.txt holds %d of 6 runs" % (tag, len(ab)))
fused = [(float(v), float(c), eval(k)) for m, v, c, k in ab if m == "1"]
unfused = [(float(v), float(c), eval(k)) for m, v, c, k in ab if m == "0"]
mean = lambda xs: sum(xs) / len(xs)
var_txt = open(need(P + tag + "_f2_variants.txt")).read()
variants = re.findall(r"== (\S+)\s+([0-9.]+) MulRelin/s\s+ntt16_f2_kernel\s+([0-9.]+) us/launch\s+\((.*)\)", var_txt)
if len(variants) < 8:
    sys.exit("tools/write_profiles_readme.py: %s_f2_variants.txt holds %d variants (variant builds failed on the GPU box?)" % (tag, len(variants)))

# ---- ablation tables of the two forward kernels
abl_txt = open(need(P + tag + "_ntt16_ablation.txt")).read()
def abl(name, limbs=1792, kernel="ntt32"):
    parts = re.split(r"^## ", abl_txt, flags=re.M)
    txt = next((x for x in parts if x.startswith(kernel)), None)
    m = re.search(r"== %s .*?\n(?:.*\n)*?limbs +%d .*? ([0-9.]+) us/launch" % (re.escape(name), limbs), txt or "")
    if not m:
        sys.exit("tools/write_profiles_readme.py: no '%s' (%s, %d limbs) in %s_ntt16_ablation.txt" % (name, kernel, limbs, tag))
    return float(m.group(1))

power = [l for l in open(need(P + tag + "_power_probe.txt")).read().split("\n") if l.startswith("sample under load")]
pw = sorted(float(re.search(r"Power \(W\): ([0-9.]+)", l).group(1)) for l in power)
ck = sorted(float(re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l).group(1)) for l in power)
if not pw:
    sys.exit("tools/write_profiles_readme.py: no samples in %s_power_probe.txt" % tag)

bf, bfp, pn, c2, c4, p14 = J("bench_bfv"), J("bench_bfv_plain"), J("bench_pn16"), J("bench_cnn2"), J("bench_cnn4"), J("bench_pn14")
p2 = J("bench_2party")
cb8, cb16 = J("bench_cnn4_batch8", False), J("bench_cnn4_batch16", False)
pn14b = [json.loads(l) for l in open(P + tag + "_pn14_batch.jsonl") if l.strip().startswith("{")] if os.path.exists(P + tag + "_pn14_batch.jsonl") else []
sweep = [json.loads(l) for l in open(P + tag + "_party_sweep.jsonl") if l.strip().startswith("{")] if os.path.exists(P + tag + "_party_sweep.jsonl") else []
extra_files = [("%s_gputests.txt" % tag, "`python -m pytest tests -m gpu` on the final sources"), ("%s_switch_matrix.txt" % tag, "`tools/switch_matrix.sh`: the GPU suite under seven switch sets of the diagnostic library"),
               ("%s_fuzz_parity.txt" % tag, "`tools/fuzz_parity.py` long runs (random rings / levels / id sets against the oracle)"), ("%s_f2_trace.txt" % tag, "`tools/f2_trace.py` (trace build): per-wave phase timeline of `ntt16_f2_kernel` and the shader clock inside it"),
               ("%s_sq_counters.txt" % tag, "SQ / TCC counters per kernel (three `--pmc` passes of the same bench command)"), ("%s_dist_5ranks.json" % tag, "`MKHE_DIST_BACKEND=gloo MKHE_DIST_ONE_DEVICE=1 python bench.py --gpus 5 --steps 2 --warmup 1`: the N > 1 path with five ranks on the one device of the box (functional only)"),
               ("%s_pmc_unfused.txt" % tag, "the same two PMC passes with `MKHE_F2_FUSED=0` (diagnostic library): HBM bytes per step of the round-5 launch set on the same box"),
               ("%s_f2_plan_sweep.txt" % tag, "`tools/f2_level_sweep.py` + `f2_sweep_table.py`: MulRelinNew per second over 1-6 parties x levels 0-13, planned grid of `ntt16_f2_kernel` against the unfused launches, alternating -- the data behind `F2_RUN_COST` / `F2_SLACK` (DESIGN.md section 4.6)"),
               ("%s_f2_plan_ab.txt" % tag, "`tools/f2_plan_ab.sh`: `bench.py --parties 1..4`, mkbfv and cnn with the planned grid against the first version's rule (one workgroup per CU or nothing)"),
               ("%s_fuzz_pn15_planned.txt" % tag, "`tools/fuzz_parity.py 240 6101 pn15` on the planned-grid sources: every (parties, level) shape of MulRelin drawn at least four times"),
               ("%s_fuzz_circuit.txt" % tag, "`tools/fuzz_circuit.py` on the final sources"), ("%s_fuzz_batch.txt" % tag, "`tools/fuzz_batch.py` on the final sources"),
               ("%s_batch_lanes.txt" % tag, "`tools/batch_lanes_ab.sh`: `mkhe_mul_relin_batch` on the headline ring, B evaluations in flight against B in lock step against one at a time"),
               ("%s_cu_partition.txt" % tag, "`tools/cu_partition_ab.py`: one to four evaluations in flight on forked contexts, whole chip and disjoint CU sets (experiment)"),
               ("%s_auto_lanes_ab.txt" % tag, "experiment: the single MulRelin entry handing its evaluations to internal contexts by itself (built, green, removed)"),
               ("%s_party_latency.txt" % tag, "per-launch time of the latency-bound [inverse NTT, ModDown] pair by party count: what per-party pipelining would launch (DESIGN.md section 10)"),
               ("%s_f1_tiles_ab.txt" % tag, "experiment: digit-major tiles for the operands of the F1 kernel (built, green, removed)")]

txt = f"""# profiles/ — rocprofv3 evidence, round 6 (`{tag}_*`)

Everything here was produced on one MI355X per `gpurun` call by `tools/profile_round.sh <run> part1 | part2 | part3 | part4` (or the script a file names) and distilled by `tools/collect_profiles.py`; this file is
generated (`tools/write_profiles_readme.py {tag}`: a figure that is missing from the tracked files stops the generator — round 5 printed not-a-number for an empty table).
Earlier rounds: `README_r5.md` (its narrative of rounds 1–5 stands, except where this file says otherwise) and the `r1a_ … r5_` files.  Boxes differ by ± 3 %
(parts that throttle at 1255 W by more): compare only inside one file or one call.

## The bench line (`{tag}_bench_plain.json` = one plain `python bench.py`)

**{pl["value"]:.1f} MulRelin/s** ({pl["ms_per_step"]:.4f} ms per step; right after the host-side set-up: {C["mulrelin_per_sec_cold_start"]:.1f}), workload `{C["workload"]}`, bit-exact against the oracle in the same run
(`cpu_baseline.bit_exact_vs_gpu = {cb["bit_exact_vs_gpu"]}`; CPU oracle {cb["value"]:.3f} MulRelin/s on one thread, {cb["value_limb_parallel"]:.2f} on {cb["cores_limb_parallel"]}).
{R["launches_per_step"]:.0f} engine kernels per step (round 5: 11; the second Decompose launch and the streaming F2 launch became one).
Independent evaluations in flight on forked contexts, same run, steady state: two {C["mulrelin_per_sec_two_in_flight"]:.1f}, three {C["mulrelin_per_sec_three_in_flight"]:.1f} MulRelin/s;
through `mkhe_mul_relin_batch` (in flight on internal contexts at this launch size since round 6, joined at the end of each call; `{tag}_batch_lanes.txt`): B = 2 {C["mulrelin_per_sec_batch2"]:.1f}, B = 4 {C["mulrelin_per_sec_batch4"]:.1f}
(rounds 4-5 measured this among the first legs, clocks not settled, and read it as a loss; `{tag}_cu_partition.txt`: one to four in flight, and the same on disjoint CU sets -- a loss).

`roofline`: dominant kernel `{R["kernel"].split("  ")[0]}` — {R["alg_bytes_per_launch"] / 1e6:.1f} MB algorithmic (16·N bytes × {R["alg_bytes_per_launch"] / (16 * 32768):.0f} limbs) in {R["avg_launch_us"]:.1f} µs = {R["achieved"]:.0f} GB/s,
**frac {R["frac"]:.3f}** inside the step; **back to back** (`frac_back_to_back`, the `valu_bound` leg's shipped timings of the same launch shape): {", ".join("`%s` %.3f" % (k.split("<")[0], v) for k, v in b2b.items())};
`traffic` {R["traffic"] / 1e6:.0f} MB per launch by the PMC counters (`traffic.json`: (2·FETCH_SIZE + WRITE_SIZE)·1024, separate `--pmc` passes of `bench.py --steps 6 --warmup 2 --no-cpu --no-extras`) = {R["traffic"] / (R["avg_launch_us"] * 1e-6) / 1e9 / 8000:.3f} of the peak;
compulsory bytes (every source limb once, every digit limb once) {R["compulsory_bytes_per_launch"] / 1e6:.0f} MB = {R["frac_compulsory"]:.3f}.
`valu_bound` (the kernel with loads, stores and LDS exchanges compiled out, 300 launches back to back): """ + "; ".join(
    "`%s` shipped %.0f µs, butterflies only %.0f µs (%.2f of it; %.2f of the roofline if the memory side were free)" % (k.split("<")[0], v["shipped_us"]["1792_limbs"], v["butterflies_only_us"]["1792_limbs"],
                                                                                                            v["butterflies_share"]["1792_limbs"], v["frac_if_alu_only"]["1792_limbs"])
    for k, v in vb.items() if isinstance(v, dict) and "shipped_us" in v) + f""".
rocprofv3 cross-check (`{tag}_kernel_stats_noovl.csv`, `--kernel-trace --stats` of `MKHE_NO_OVERLAP=1 bench.py --steps 20 --warmup 3 --no-cpu --no-extras`): `{DOM.split("<")[0]}` {dcalls} calls, average {davg:.1f} µs
(min {dmn:.1f}, max {dmx:.1f}); `ntt16_f2_kernel` {f2calls} calls, average {f2avg:.1f} µs; `ext_inner_xy_kernel<4,4,true>` {xycalls} calls, average {xyavg:.1f} µs.
The two forms of the dominant kernel forced: `MKHE_NTT32=1` {plh["value"]:.1f} MulRelin/s (frac {plh["roofline"]["frac"]:.3f}), `MKHE_NTT32=0` {pl16["value"]:.1f} (frac {pl16["roofline"]["frac"]:.3f}); this run's own choice: `{C.get("ntt_kernel_choice")}`.

Per kernel class (HIP events on the context stream, one kernel at a time; algorithmic GB/s by the byte models of DESIGN.md §4; PMC = bytes per launch and GB/s by the counters where the launch pattern of `traffic.json` matches):

| kernel class | launches / step | µs / launch | µs / step | alg. GB/s | PMC |
|---|---|---|---|---|---|
{table(pl)}

HBM bytes per step by the counters, every kernel of the step (`traffic.json`: launches ÷ the {nsteps} steps of the profiled command × bytes per launch): **{all_bytes / 1e9:.2f} GB**;
the round-5 launch set on the same box, same passes (`{tag}_pmc_unfused.txt`, `MKHE_F2_FUSED=0`): {unf_bytes:.2f} GB.  (VERDICT r5 asked for ≤ 2.25 GB: not reached — `ntt16_f2_kernel` moves {f2_pmc / 1e6:.0f} MB
per launch where its byte model says {f2_alg / 1e6:.0f}: every t_i limb is fetched once per XCD, and the CRS u by more than one of the four parties' workgroups.)
The latency-bound tail (inverse NTTs, ModDowns, tensor, small forward NTTs): {tail_us:.0f} µs per step (VERDICT r5 asked for ≤ 130: not reached, and not attacked this round beyond the fused kernel).

## Step F2 inside the Decompose NTT of the t_i (`ntt16_f2_kernel`, DESIGN.md §4.6)

Same call, same box, alternating (`{tag}_f2_fused_ab.txt`; diagnostic library, `MKHE_F2_FUSED` = 1 / 0; `bench.py --no-cpu --no-extras`):

| | MulRelin/s (three runs) | cold start | Decompose NTT µs / step | streaming inner products µs / step | `ntt16_f2_kernel` | inverse NTT µs / step |
|---|---|---|---|---|---|---|
| fused (shipped) | {" / ".join("%.1f" % v for v, c, k in fused)} | {" / ".join("%.1f" % c for v, c, k in fused)} | {mean([sum(x for n, x in k.items() if n.startswith(("ntt16_fwd_kernel<true>", "ntt32_fwd_kernel<true>"))) for v, c, k in fused]):.1f} | {mean([k.get("ext_inner_kernel", 0) for v, c, k in fused]):.1f} | {mean([k.get("ntt16_f2_kernel", 0) for v, c, k in fused]):.1f} | {mean([k.get("ntt_inv_kernel<15>", 0) for v, c, k in fused]):.1f} |
| round-5 launch set | {" / ".join("%.1f" % v for v, c, k in unfused)} | {" / ".join("%.1f" % c for v, c, k in unfused)} | {mean([sum(x for n, x in k.items() if n.startswith(("ntt16_fwd_kernel<true>", "ntt32_fwd_kernel<true>"))) for v, c, k in unfused]):.1f} | {mean([k.get("ext_inner_kernel", 0) for v, c, k in unfused]):.1f} | – | {mean([k.get("ntt_inv_kernel<15>", 0) for v, c, k in unfused]):.1f} |

**+ {100 * (mean([v for v, c, k in fused]) / mean([v for v, c, k in unfused]) - 1):.1f} % MulRelin/s, + {100 * (mean([c for v, c, k in fused]) / mean([c for v, c, k in unfused]) - 1):.1f} % from a cold start** (VERDICT r5 priced ≥ 1440 MulRelin/s: not reached — DESIGN.md §4.6 says why: the kernel sits at the package power cap,
{pw[len(pw) // 2]:.0f} W / {ck[len(ck) // 2]:.0f} MHz under the dominant kernel in `{tag}_power_probe.txt`, and everything that only hides latency moves nothing).

Variants of the kernel inside the MulRelin (`{tag}_f2_variants.txt`: one library per variant, built on the box; the `no_*` / `vector_alu_only` rows give WRONG results on purpose and change the
operands' bit patterns — and with them the power: they bound nothing in this regime):

| variant | MulRelin/s | `ntt16_f2_kernel` µs / launch | flags |
|---|---|---|---|
""" + "\n".join("| %s | %s | %s | `%s` |" % (n.replace("_", " "), v, u, f or "–") for n, v, u, f in variants) + f"""

## The Decompose NTT kernels back to back (`{tag}_ntt16_ablation.txt`, `{tag}_ntt16_launch_sizes.txt`)

1500 launches per variant (steady state), 1792 limbs, µs per launch.  Single-pass kernel (`MKHE_NTT32=1`): shipped {abl("shipped"):.0f}; no result stores {abl("no_stores"):.0f}, no source loads {abl("no_source_loads"):.0f},
no twiddle loads {abl("no_twiddle_loads"):.0f}, no LDS exchanges {abl("no_exchanges"):.0f}; butterflies only {abl("butterflies_only"):.0f}; all butterflies removed {abl("no_butterflies"):.0f}.
Two-pass kernel (`MKHE_NTT32=0`): shipped {abl("shipped", kernel="ntt16"):.0f}, the vector-ALU side alone {abl("no_mem_no_xchg", kernel="ntt16"):.0f}, all butterflies removed {abl("no_bfly", kernel="ntt16"):.0f}.
(Round 5's table was empty: its variant libraries were linked from a stale object list — `tools/ntt16_variants.sh` copies every object now and `tools/collect_profiles.py` refuses a table without timings.)

## Secondary lines

| file | line |
|---|---|
| `{tag}_bench_bfv_plain.json` | mkbfv 4-party MulRelinNew, PN15QP880 (14 + 14 + 2 limbs): **{bfp["value"]:.1f} MulRelin/s** ({bfp["ms_per_step"]:.3f} ms), dominant `{bfp["roofline"]["kernel"].split("  ")[0]}` at {bfp["roofline"]["frac"]:.3f}; CPU oracle {bfp["cpu_baseline"]["value"]:.3f}/s, bit-exact {bfp["cpu_baseline"]["bit_exact_vs_gpu"]} |
| `{tag}_bench_pn16.json` | mkckks 8-party MulRelin, PN16QP1761 (N = 2^16, 34 + 4 limbs, α = 2): **{pn["value"]:.1f} MulRelin/s** ({pn["ms_per_step"]:.2f} ms); CPU oracle on the 2-party sub-problem {pn["cpu_baseline"]["value"]:.3f}/s (`{pn["cpu_baseline"]["sample"][:60]}…`), bit-exact {pn["cpu_baseline"]["bit_exact_vs_gpu"]} |
| `{tag}_bench_2party.json` | mkckks 2-party MulRelin, PN15QP880 (BASELINE configs[0], the reference benchmark's own case, at the headline ring): **{p2["value"]:.0f} MulRelin/s** ({1e3 * p2["ms_per_step"]:.1f} µs; three in flight {p2["config"]["mulrelin_per_sec_three_in_flight"]:.0f}), dominant `{p2["roofline"]["kernel"].split("  ")[0]}` at {p2["roofline"]["frac"]:.2f}; CPU oracle {p2["cpu_baseline"]["value"]:.2f}/s on one thread, {p2["cpu_baseline"]["value_limb_parallel"]:.2f} on {p2["cpu_baseline"]["cores_limb_parallel"]}, bit-exact {p2["cpu_baseline"]["bit_exact_vs_gpu"]} |
| `{tag}_bench_pn14.json` | mkckks 4-party MulRelin, PN14QP439 (the reference benchmark's first set): **{p14["value"]:.0f} MulRelin/s** ({1e3 * p14["ms_per_step"]:.1f} µs), dominant `{p14["roofline"]["kernel"].split("  ")[0]}` at {p14["roofline"]["frac"]:.2f}; CPU oracle {p14["cpu_baseline"]["value"]:.2f}/s on one thread, {p14["cpu_baseline"]["value_limb_parallel"]:.2f} on {p14["cpu_baseline"]["cores_limb_parallel"]}, bit-exact {p14["cpu_baseline"]["bit_exact_vs_gpu"]} |
| `{tag}_bench_cnn4.json` / `_cnn2.json` | encrypted CNN inference, 4 / 2 parties: **{c4["value"]:.0f} / {c2["value"]:.0f} inferences/s** ({c4["ms_per_step"]:.2f} / {c2["ms_per_step"]:.2f} ms per image); CPU oracle {c4["cpu_baseline"]["value"]:.2f}/s, bit-exact {c4["cpu_baseline"]["bit_exact_vs_gpu"]} |
""" + ("| `%s_bench_cnn4_batch8.json` / `_batch16.json` | B images in lock step, 4 parties: **%.0f / %s inferences/s** |\n" % (tag, cb8["value"], ("%.0f" % cb16["value"]) if cb16 else "–") if cb8 else "") + (
    "| `%s_pn14_batch.jsonl` | PN14QP439, B = 4 / 8 / 16 inputs in lock step: %s MulRelin/s |\n" % (tag, " / ".join("%.0f" % j["value"] for j in pn14b)) if pn14b else "") + (
    "| `%s_party_sweep.jsonl` | PN15QP880, 1 / 2 / 4 / 8 / 16 parties (device-expanded keys): %s MulRelin/s |\n" % (tag, " / ".join("%.0f" % j["value"] for j in sweep)) if sweep else "") + f"""| `{tag}_bench_noovl.json` / `_ovl.json` | the headline under `rocprofv3 --kernel-trace --stats`, side stream off / on: {no["value"]:.1f} / {ov["value"]:.1f} MulRelin/s |

## Other files of this round

| file | what |
|---|---|
""" + "\n".join("| `%s` | %s |" % (f, d) for f, d in extra_files if os.path.exists(P + f)) + f"""
| `{tag}_kernel_stats_noovl.csv`, `_ovl.csv`, `_bfv.csv`, `_pn16.csv` | `rocprofv3 --kernel-trace --stats` summaries of the bench commands above |
| `{tag}_ntt16_isa.txt`, `{tag}_ntt32_isa.txt` | instruction counts of the two forward kernels from the ISA (no GPU) |
| `{tag}_ubench.txt`, `{tag}_power_probe.txt`, `{tag}_ntt_in_context.txt` | issue-rate / read-bandwidth micro-benchmarks; package power and clock under the dominant kernel; the Decompose launch inside the MulRelin, both forms |
| `traffic.json` | HBM bytes per launch from the PMC passes, tied to the kernel sources by `csrc_sha256` (bench.py attaches them only when the digest and the launch pattern match) |
"""
bad = [l[:160] for l in txt.split("\n") if re.search(r"(?<![A-Za-z])nan(?![A-Za-z])", l)]
if bad:
    sys.exit("tools/write_profiles_readme.py: %d line(s) of profiles/README.md would carry a missing figure (nan):\n  " % len(bad) + "\n  ".join(bad[:8]))
open(P + "README.md", "w").write(txt)
print("profiles/README.md written")

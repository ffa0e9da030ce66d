#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled round-2 files:  python tools/write_profiles_readme_r2.py r2"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
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


def table(j, traffic=None):
    """PMC column: bytes per launch from the traffic file recorded in the SAME profile run (not from what bench.py attached: the plain
    bench of a run starts before that run's traffic.json exists), over the avg launch time of this table; only for kernels whose
    recorded call count matches launches_per_step * (W + 2 K) of the PMC command"""
    rows = []
    tk = (traffic or {}).get("kernels", {})
    for k, v in j["roofline"]["kernels"].items():
        key = k.split()[0]
        rec = tk.get(key) or tk.get(key.replace("[_batch]", "_batch"))
        pmc = "–"
        if rec and traffic.get("steps") is not None and abs(rec["launches"] - v["launches_per_step"] * (traffic["warmup"] + 2 * traffic["steps"])) < 0.5:
            pmc = "%.0f" % (rec["hbm_bytes_per_launch"] / (v["avg_launch_us"] * 1e-6) / 1e9)
        rows.append("| `%s` | %.1f | %.1f | %.3f | %.0f | %s |" % (k.split("  ")[0], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"], v["achieved_GBs"], pmc))
    return "\n".join(rows)


def whole_step_text():
    """bytes one MulRelin moves through HBM by the PMC passes, against SURVEY.md 8(d)'s two models"""
    tk, tot, cov_ms, all_ms = tr["kernels"], 0.0, 0.0, 0.0
    for k, v in pl["roofline"]["kernels"].items():
        key = k.split()[0]
        rec = tk.get(key) or tk.get(key.replace("[_batch]", "_batch"))
        all_ms += v["ms_per_step"]
        if rec and tr.get("steps") is not None and abs(rec["launches"] - v["launches_per_step"] * (tr["warmup"] + 2 * tr["steps"])) < 0.5:
            tot += rec["hbm_bytes_per_launch"] * v["launches_per_step"]
            cov_ms += v["ms_per_step"]
    return ("Whole step (SURVEY.md §8d asks for the bytes moved against its two models): the kernels with a PMC column — %.0f %% of the kernel time — move **%.2f GB per MulRelin** through HBM "
            "= %.2f TB/s averaged over the %.3f ms step.  Compulsory model (every key read once: 13 × 56 MiB; the 12 hoisted digit vectors written and read: ≈ 0.9 GiB): 1.75 GB ⇒ 0.22 ms at 8 TB/s; staged "
            "(unfused) model: 7.7 GB ⇒ 0.97 ms.  The step sits between the two: the fused kernels (digit spread in the NTT load, batched inner products, tensor on the hoisted diagonal) remove "
            "%.0f %% of the staged model's traffic; what remains above the compulsory bytes is mostly the second read of the hoisted digits (inner product, then external product) and the source re-reads of the Decompose NTT.\n"
            % (100 * cov_ms / all_ms, tot / 1e9, tot / 1e9 / pl["ms_per_step"], pl["ms_per_step"], 100 * (1 - tot / 7.7e9)))


R, Rn = pl["roofline"], no["roofline"]
whole_step = whole_step_text()
def pmc_frac(k):
    v = R["kernels"].get(k, {}).get("hbm_GBs_pmc")
    return "%.2f" % (v / 8000.0) if v else "n/a"
dom_key = R["kernel"].split()[0]
avg, calls, mn, mx = st("ntt16_fwd_kernel<true>")
expect = 2 * (no["warmup"] + 2 * no["steps"])
dom = tr["kernels"].get(dom_key, {})
cb = pl["cpu_baseline"]
txt = f'''# profiles/ — measured on MI355X (gfx950), round 2

Distilled by `tools/collect_profiles_r2.py` from ONE `gpurun` call of `tools/profile_r2.sh` (the commands are in that script; build = the
commit that carries these files); this file is written by `tools/write_profiles_readme_r2.py`.  Round-1 files (`r1*`) are kept for comparison.

Every profiled pass runs the same command, `MKHE_NO_OVERLAP=1 python3 bench.py --steps K --warmup W --no-cpu --no-extras` (every kernel alone on
the main stream, no secondary workloads).  The dominant kernel — the Decompose-fused forward NTT `ntt16_fwd_kernel<true>` — is then launched exactly
**2 × (W + 2·K)** times: twice per MulRelin (1792 limbs: hoisting of the 8 operand components; 896 limbs: the 4 intermediate t_i), in the warm-up, the
timed loop and the HIP-event leg.

| file | command |
|---|---|
| `{tag}_bench_plain.json` | `python3 bench.py` (no flags, no profiler: what the driver runs) |
| `{tag}_kernel_stats_noovl.csv`, `{tag}_bench_noovl.json` | `MKHE_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras` (expected calls of the dominant kernel: 2 × (3 + 40) = {expect}; recorded: {calls}) |
| `{tag}_kernel_stats_ovl.csv`, `{tag}_bench_ovl.json` | the same with the side-stream overlap on (kernels that run concurrently stretch each other) |
| `traffic.json` | two passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of the same command with `--steps 6 --warmup 2` (dominant kernel: 2 × (2 + 12) = 28 calls; recorded: {dom.get("launches", "?")}); `tools/traffic_from_pmc.py`; `bench.py` attaches these figures only when workload and launch pattern match |
| `{tag}_sq_counters.txt` | four `--pmc` passes (SQ wave / wait / instruction counters, LDS, VMEM, L2 hit rate) of the same command with `--steps 4 --warmup 2` |
| `{tag}_ntt16_isa.txt` | `tools/ntt16_isa.py`: instruction counts of the dominant kernel's pass body from the gfx950 ISA, one butterfly verbatim, code-object record |
| `{tag}_ubench.txt` | `tools/ubench/bfly16_rate` (the kernel's butterfly, bare, at 1 / 2 / 4 / 8 waves per SIMD, one and two interleaved chains), `valu_rate`, `bfly_rate` |
| `{tag}_ntt16_phase_trace.txt` | `tools/ntt16_trace.py` on the trace build: shader-clock stamps per wave, pass and phase of the dominant kernel |
| `{tag}_ntt16_ablation.txt` | `tools/ntt16_ablation.sh`: the dominant kernel re-built without its butterflies / LDS exchanges / result stores / LDS stash (timing only), 1792 and 896 limbs back to back |
| `{tag}_ntt16_launch_sizes.txt` | `tools/ntt16_bench.py`: the Decompose NTT at 1792 / 896 / 448 / 224 limbs (canonical outputs) |
| `{tag}_kernel_stats_bfv.csv`, `{tag}_bench_bfv*.json` | `bench.py --scheme bfv` under the profiler (overlap off) and plain |
| `{tag}_bench_pn16*.json`, `{tag}_kernel_stats_pn16.csv` | `bench.py --params PN16QP1761 --parties 8` (configs[3] ring on one GPU) |
| `{tag}_bench_cnn2/4.json`, `{tag}_bench_pn14.json`, `{tag}_party_sweep.jsonl` | secondary workloads |

## Headline (BASELINE.json configs[1]): mkckks 4-party MulRelin, PN15QP880, N = 2^15, 14 Q + 2 P limbs

* **{pl["value"]:.0f} MulRelin/s** ({pl["ms_per_step"]:.3f} ms per step: hoist both operands + MulAndRelinHoisted + Rescale; round 1: 763/s, 1.310 ms), bit-exact against
  the oracle on the same inputs in this very run (`cpu_baseline.bit_exact_vs_gpu = {cb["bit_exact_vs_gpu"]}`); CPU oracle on the GPU box's host: {cb["value"]:.2f} MulRelin/s on 1 thread,
  {cb.get("value_limb_parallel", 0):.2f} with its limb loops on {cb.get("cores_limb_parallel", "?")} threads.
* steady state: **{pl["config"].get("mulrelin_per_sec_steady_state", 0):.0f} MulRelin/s** (200 steps after 100 untimed ones, same run).  The headline figure above is timed as the
  bench contract says — {pl["steps"]} steps after {pl["warmup"]} warm-up steps, right after the host-side set-up — and this GPU takes about 150 ms of load to settle its clocks: a 5-step
  window runs at ≈ 1.10 ms per step after 0.2 s of idleness and at ≈ 0.92 ms 120 steps later (`tools/ramp_probe.py`); two MulRelin in flight through forked contexts:
  {pl["config"].get("mulrelin_per_sec_two_in_flight", 0):.0f}/s.
* under `rocprofv3 --kernel-trace`, overlap off: {no["value"]:.0f} MulRelin/s ({no["ms_per_step"]:.3f} ms); overlap on: {ov["value"]:.0f} MulRelin/s ({ov["ms_per_step"]:.3f} ms).

Per kernel class, HIP events inside `bench.py` (roofline leg, overlap off), per step.  "algorithmic GB/s" is the byte model of DESIGN.md §4 (it counts
cache-served re-reads and may exceed the chip's peak); "PMC GB/s" is what the kernel really moved through the L2's memory side,
(2·FETCH_SIZE + WRITE_SIZE) from `traffic.json` over the same launch pattern — the figure to hold against the 8 TB/s peak for the streaming kernels:

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(pl, tr)}

(Launches of ≤ 128 limbs in the `ntt_fwd_kernel<15,·,false>` and `ntt_inv_kernel<15>` classes run as `ntt_pass4_fwd/inv_kernel` + `ntt_fwd/inv_lds_kernel`,
the low-latency path of DESIGN.md §4; that is what the rocprofv3 statistics list, so those classes have no PMC column.)

{whole_step}
### Dominant kernel `ntt16_fwd_kernel<true>` (DESIGN.md §4, "H16")

* HIP-event average inside `bench.py`: **{R["avg_launch_us"]:.1f} µs per launch** (plain run), {Rn["avg_launch_us"]:.1f} µs in the profiled run; rocprofv3 kernel-trace average of that
  profiled run: **{avg:.1f} µs** over {calls} calls (min {mn:.0f} = the 896-limb launches, max {mx:.0f} µs).  Round 1 (`ntt_fwd_kernel<15,2,true>`): 297.5 µs.
* algorithmic bytes per launch {R["alg_bytes_per_launch"] / 1e6:.1f} MB (16·N B per limb-NTT × (1792 + 896)/2 limbs) ⇒ **{R["achieved"]:.0f} GB/s = {R["frac"]:.3f} of the 8 TB/s HBM peak** (round 1: 0.296).
  By the compulsory bytes of the fused Decompose (read every source limb once, write every digit limb: {R.get("compulsory_bytes_per_launch", 0) / 1e6:.0f} MB per average launch) it is {R.get("frac_compulsory", 0):.3f}.
* HBM traffic from the PMC passes: {dom.get("hbm_bytes_per_launch", 0) / 1e6:.1f} MB per launch (FETCH_SIZE {dom.get("fetch_size_kb", 0) / 1e3:.1f} MB ×2 + WRITE_SIZE {dom.get("write_size_kb", 0) / 1e3:.1f} MB) =
  {dom.get("hbm_bytes_per_launch", 0) / R["alg_bytes_per_launch"]:.2f}× the algorithmic bytes.  Written: the {R["alg_bytes_per_launch"] / 2e6:.0f} MB of results and what is left of the register spills (28 B of scratch per
  lane: 6 spilled VGPRs; the first half of round 2 wrote 40–60 B per lane plus the parked upper halves of pass 0: 1005 MB per launch).  Read: the source limbs in both passes (each is spread
  under 16 moduli: most re-reads are cache hits), the twiddle pairs.
* **What bounds it** (DESIGN.md §3): its memory traffic and its butterflies, which overlap only partly.  Ablation builds at 1792 limbs back to back (throttled clock): whole kernel
  ≈ 360 µs, every butterfly removed 202 µs (245 µs before the parking store / reload and most of the scratch traffic were removed), scalar-twiddle butterflies removed 253 µs, per-lane ones
  268 µs.  `{tag}_ubench.txt` (`bfly31_rate`): the butterfly on the one-round product takes **60.7 cycles per wave at 2.10 GHz = 28.8 ns**, on the two-round product (`bfly16_rate`,
  phase D) 73.1 cycles at 1.98 GHz = 36.8 ns.  A limb is 2 × (104 one-round + 16 two-round) wave-butterflies on 16 waves / 4 SIMDs ⇒ **28.7 µs per limb and CU** before any load, store,
  exchange or twiddle fetch; the average launch has 1344 / 256 = 5.25 limbs per CU ⇒ ≥ 151 µs.  Measured: {R["avg_launch_us"]:.0f} µs = {R["avg_launch_us"] * 256 / 1344:.1f} µs per limb and CU, {28.7 * 1344 / 256 / R["avg_launch_us"]:.2f} of that floor.
  `{tag}_ntt16_isa.txt` counts the instructions of the shipped code object, `{tag}_ntt16_phase_trace.txt` stamps the phases per wave.
* `{tag}_sq_counters.txt`: `SQ_INSTS_VALU` per wave, `SQ_WAIT_INST_ANY` (waves ready but waiting for the vector ALU that another wave holds) vs `SQ_WAIT_ANY`
  (waves at barriers / waitcnt), `SQ_LDS_BANK_CONFLICT` = 0 for all four LDS layouts, L2 hit rate.

What changed against round 1 (DESIGN.md §3, §4 have the measurements): 16 coefficients per thread and 64 VGPRs ⇒ two workgroups per CU; scalar (SGPR) twiddles for the
nine stages whose twiddles are uniform per workgroup / wave; **the one-round product** (twiddle stored as the pair w·2^31, w·2^63 mod q: 8 + 1 instructions instead of 12) in stage 0 and
phases A, B, C; both modulus classes on signed never-reduced butterflies with a 7-instruction float-estimated partial reduction (round 1: Harvey butterflies, +50 % instructions, for the
59/60-bit primes); stage 0 recomputed in the second pass instead of parking half a limb in HBM; stage-0 results staged through the wave's own LDS region instead of scratch; single-word
LDS reads (no register re-pairing moves); non-temporal result stores; one instantiation.

### Streaming kernels

`inner_product_kernel<4>` {R["kernels"]["inner_product_kernel"]["avg_launch_us"]:.0f} µs per launch (round 1: 104; one launch per step since x comes out of step F1's kernel as a by-product), `ext_inner_kernel` {R["kernels"]["ext_inner_kernel"]["avg_launch_us"]:.0f} µs
(two launches: F1 + x, and the E / F2 batch): 16-byte lanes, unrolled term / digit loops, items that share a key computed by one thread, and **non-temporal loads for every operand that is
read once per launch** (keys, hoisted digits), so that x, y, the CRS and the twiddles stay cache-resident.  PMC GB/s in the table above: {pmc_frac("inner_product_kernel")} and {pmc_frac("ext_inner_kernel")} of the 8 TB/s spec —
MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy (6.0–6.1 TB/s for an in-order sweep of a large table), i.e. they run at what the memory system delivers; a block-size ×
chunks-per-thread sweep spans 85–90 µs (DESIGN.md §4, "Streaming kernels").  The ModDown launches and the small inverse NTTs are launch-latency-bound (`{tag}_sq_counters.txt`: `SQ_WAIT_ANY`
0.74 of the wave cycles), not bandwidth-bound; since the external products of one destination are merged (DESIGN.md §4, "Merged external products": Q limbs summed in the NTT domain at the load
of one inverse NTT, the tensor term folded in, one ModDown tail per destination) a 4-party MulRelin runs 158 instead of 326 inverse limb-NTTs.

## BASELINE.json configs[2]: mkbfv 4-party MulRelinNew, PN15QP880 BFV chain (14 Q + 14 QMul + 2 P)

* **{bfp["value"]:.0f} MulRelin/s** ({bfp["ms_per_step"]:.3f} ms per step; round 1: 435/s), bit-exact against the oracle at full size (`tests/test_gpu_headline.py`, and `bench.py --scheme bfv` in every run);
  under the profiler with overlap off {bf["value"]:.0f}/s.

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s | PMC GB/s |
|---|---|---|---|---|---|
{table(bf)}
'''
sweep = P + tag + "_party_sweep.jsonl"
if os.path.exists(sweep):
    rows = [json.loads(l) for l in open(sweep) if l.strip().startswith("{")]
    if rows:
        txt += "\n## Party count (the metric is \"MulRelin/sec at n parties\"): PN15QP880, one GPU\n\n`python3 bench.py --parties k --no-cpu --device-keys --steps 20 --warmup 3`:\n\n| parties | MulRelin/s | ms per MulRelin | Rotate/s | RotateHoisted/s | Conjugate/s |\n|---|---|---|---|---|---|\n"
        for r in rows:
            c = r["config"]
            txt += "| %d | %.0f | %.3f | %.0f | %.0f | %.0f |\n" % (c["parties"], r["value"], r["ms_per_step"], c.get("rotate_per_sec", 0), c.get("rotate_hoisted_per_sec", 0), c.get("conjugate_per_sec", 0))
if p14:
    txt += "\nPN14QP439 (N = 2^14, 6 + 2 limbs), 4 parties: %.0f MulRelin/s (%.3f ms).\n" % (p14["value"], p14["ms_per_step"])
if pn:
    c = pn["config"]
    txt += f'''
## BASELINE.json configs[3] ring on ONE GPU: 8-party MulRelin + hoisted Rotate, PN16QP1761 (N = 2^16, 34 Q + 4 P primes, α = 2, β = 17)

`python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2`: **{pn["value"]:.1f} MulRelin/s** ({pn["ms_per_step"]:.2f} ms per step; round 1: 53.8), Rotate {c.get("rotate_per_sec", 0):.0f}/s,
RotateHoisted {c.get("rotate_hoisted_per_sec", 0):.0f}/s.  Bit-exactness at this ring with 8 parties: `tests/test_gpu_headline.py::test_pn16_rotate_hoisted_eight_parties`, 2 parties: `tests/test_gpu_fullsize.py`.
The party-sharded N > 1 run of this configuration is `python3 bench.py --gpus 8 --params PN16QP1761 --parties 8` (not measurable on the single-GPU boxes of this pool).
'''
if c2 and c4:
    txt += f'''
## BASELINE.json configs[4] caller on one GPU: encrypted CNN inference (cnn/cnn.go), PN14QP433

`python3 bench.py --scheme cnn --parties 2|4 --steps 20 --warmup 3`: **{c2["value"]:.0f}** inferences/s with 2 parties ({c2["ms_per_step"]:.2f} ms), **{c4["value"]:.0f}** with 4 ({c4["ms_per_step"]:.2f} ms) (round 1: 277 / 268).
`--gpus N` runs N independent replicas.  Encrypted == plaintext logits, synthetic model and the reference's trained weights: `tests/test_gpu_cnn.py`.
'''
open(P + "README.md", "w").write(txt)
print("profiles/README.md written")

#!/usr/bin/env python3
"""Regenerates profiles/README.md from the distilled files of tools/collect_profiles.py:  python tools/write_profiles_readme.py r1f"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"
tag = sys.argv[1] if len(sys.argv) > 1 else "r1f"
pl = json.load(open(P + tag + "_bench_plain.json"))
no = json.load(open(P + tag + "_bench_noovl.json"))
bf = json.load(open(P + tag + "_bench_bfv.json"))
tr = json.load(open(P + "traffic.json"))


def table(j):
    rows = []
    for k, v in j["roofline"]["kernels"].items():
        rows.append("| `%s` | %.1f | %.1f | %.3f | %.0f |" % (k.split("  ")[0], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"], v["achieved_GBs"]))
    return "\n".join(rows)


stats = {r["Name"]: r for r in csv.DictReader(open(P + tag + "_kernel_stats_noovl.csv"))}


def st(name):
    for k, v in stats.items():
        if name in k:
            return float(v["AverageNs"]) / 1e3, int(v["Calls"])
    return (0, 0)


R = pl["roofline"]
dom_key = R["kernel"].split()[0]                       # e.g. ntt_fwd_kernel<15,2,true>
dom_prof = dom_key.replace(",", ", ")                   # spelling of the rocprofv3 kernel statistics
dom = tr["kernels"][dom_key]
txt = f'''# profiles/ — measured on MI355X (gfx950), round 1

All files are distilled by `tools/collect_profiles.py` from `gpurun` runs of `tools/profile_round.sh` /
`tools/profile_sq.sh` (commands inside those scripts; build = HEAD of this round); this file is written by
`tools/write_profiles_readme.py`.

| file | command |
|---|---|
| `{tag}_bench_plain.json` | `python3 bench.py --steps 20 --warmup 3` (no profiler) |
| `{tag}_kernel_stats_noovl.csv`, `{tag}_bench_noovl.json` | `MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu` |
| `{tag}_kernel_stats_ovl.csv`, `{tag}_bench_ovl.json` | same without `MKHE_NO_OVERLAP` (side-stream overlap on: kernels that run concurrently stretch each other) |
| `{tag}_kernel_stats_bfv.csv`, `{tag}_bench_bfv.json` | `MKHE_NO_OVERLAP=1 rocprofv3 ... -- python3 bench.py --scheme bfv --steps 10 --warmup 2 --no-cpu` |
| `traffic.json` | two passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (+ `--kernel-trace`) of `bench.py --steps 6 --warmup 2 --no-cpu`, `tools/traffic_from_pmc.py` |
| `{tag}_sq_counters.txt` | three passes of 8 SQ counters each (`tools/profile_sq.sh`) |
| `{tag}_step_timeline.txt` | `tools/step_timeline.py` over a `rocprofv3 --kernel-trace` of `bench.py --steps 6 --warmup 2 --no-cpu --no-extras` (overlap on): start, duration and stream interleaving of every kernel of one MulRelin; the GPU is busy 99 % of the step, i.e. the step is the sum of its dependent kernels |
| `{tag}_ubench.txt` | output of `tools/ubench/valu_rate`, `imul_rate`, `bfly_rate` on the same chip: instruction issue rates per class and the register-resident butterfly floor (85 cycles per wave-butterfly ⇒ 38–45 µs per 2^15-point limb and CU before any load, store, exchange or twiddle traffic) |

## Headline (BASELINE.json configs[1]): mkckks 4-party MulRelin, PN15QP880, N = 2^15, 14 Q + 2 P limbs

* **{pl["value"]:.0f} MulRelin/s** ({pl["ms_per_step"]:.3f} ms per step: hoist both operands + MulAndRelinHoisted + Rescale), bit-exact against the
  oracle on the same inputs (`cpu_baseline.bit_exact_vs_gpu = {pl["cpu_baseline"]["bit_exact_vs_gpu"]}`); CPU oracle, 1 thread: {pl["cpu_baseline"]["value"]:.2f} MulRelin/s.
* under `rocprofv3 --kernel-trace` with overlap off: {no["value"]:.0f} MulRelin/s ({no["ms_per_step"]:.3f} ms).

Per kernel class, HIP events inside `bench.py` (roofline leg, overlap off), per step:

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s |
|---|---|---|---|---|
{table(pl)}

(The classes are the engine's timing classes: launches of ≤ 128 limbs in the `ntt_fwd_kernel<15,·,false>` and `ntt_inv_kernel<15>`
classes run as `ntt_pass4_fwd/inv_kernel` + `ntt_fwd/inv_lds_kernel` — the low-latency path of DESIGN.md §4 — which is what the
rocprofv3 kernel statistics list under those names.)

Dominant kernel `{dom_key}` (Decompose-fused forward NTT; template argument 2 = the 54-bit primes (reduction-free signed
butterflies) and the 59/60-bit primes (Harvey butterflies) of a batch share ONE persistent launch, the class is looked up per limb;
launched separately the big-prime class only got the CUs when the other class's workgroups exited and then ran a ragged
second round: 0.664 → 0.600 ms per step for the two Decompose launches):
HIP-event average {R["avg_launch_us"]:.1f} µs per launch, rocprofv3 kernel-trace average {st(dom_prof)[0]:.1f} µs
({st(dom_prof)[1]} calls; the two launches per step have 1792 and 896 limbs);
algorithmic bytes per launch {R["alg_bytes_per_launch"] / 1e6:.1f} MB (16·N B per limb-NTT) ⇒ **{R["achieved"]:.0f} GB/s = {R["frac"]:.3f} of the 8 TB/s HBM peak**;
HBM traffic from the PMC passes {dom["hbm_bytes_per_launch"] / 1e6:.1f} MB per launch (FETCH_SIZE {dom["fetch_size_kb"] / 1e3:.1f} MB ×2 + WRITE_SIZE {dom["write_size_kb"] / 1e3:.1f} MB):
{dom["hbm_bytes_per_launch"] / R["alg_bytes_per_launch"]:.2f}× the algorithmic bytes.  The excess is on the write side (WRITE_SIZE is ≈ 1.5× the limb bytes written): register
spills of the 128-VGPR kernels (68–92 B of scratch per lane) are written through to memory; the read side is *below*
the algorithmic figure because the fused digit spread re-reads each source limb 13–16 times from L2.
(A diagnostic build whose time stamps doubled the scratch size to ≈ 200 B tripled WRITE_SIZE and ran 20 % slower —
the stamps are therefore compiled only into `make trace`.)

Why the NTT sits at a third of the HBM roofline and not higher: `{tag}_sq_counters.txt` — `SQ_WAIT_INST_ANY` (≈ 0.3–0.4 of
the wave cycles) is waves waiting for the vector ALU that another wave of the SIMD holds, `SQ_WAIT_ANY` (≈ 0.4) waves
parked at barriers / waitcnt while the others compute; LDS (`SQ_ACTIVE_INST_LDS` ≈ 0.015, bank conflicts 0) and VMEM issue
are negligible.  `tools/ubench/valu_rate.hip` on the same chip: every 64-bit, carry or multiply instruction
(`v_mad_u64_u32`, `v_mad_i64_i32`, `v_mul_lo_u32`, `v_lshl_add_u64`, `v_add_co/v_addc`, `v_ashrrev_i64`, even
`v_add3_u32`) issues at ≈ 1.8 ns per wave and SIMD, only plain 32-bit VOP1/VOP2 ops at ≈ 1.05 ns.  The butterfly went
from ≈ 26 such instructions + 12 moves (first kernels, 306 µs per launch) to 15 + 2 with the signed-digit Montgomery
product (233 µs): DESIGN.md §3 has the steps and what each bought.

The memory-streaming kernels are where HBM is the bound: `ext_inner_kernel` {R["kernels"]["ext_inner_kernel"]["achieved_GBs"]:.0f} GB/s and
`inner_product_kernel` {R["kernels"]["inner_product_kernel"]["achieved_GBs"]:.0f} GB/s of algorithmic bytes (PMC traffic {tr["kernels"]["ext_inner_kernel"]["hbm_bytes_per_launch"] / 1e6:.0f} MB and {tr["kernels"]["inner_product_kernel"]["hbm_bytes_per_launch"] / 1e6:.0f} MB per launch;
`ext_inner` re-reads x / y, shared by four items each, from L2).

## BASELINE.json configs[2]: mkbfv 4-party MulRelinNew, PN15QP880 BFV chain (14 Q + 14 QMul + 2 P)

* **{bf["value"]:.0f} MulRelin/s** ({bf["ms_per_step"]:.3f} ms per step: ModUpQtoR + Rescale + DecomposeBFV + MulAndRelinBFVHoisted) under the
  profiler with overlap off; bit-exact against the oracle at full size (checked by `bench.py --scheme bfv`, CPU oracle 0.27 MulRelin/s).

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s |
|---|---|---|---|---|
{table(bf)}

'''


def load(name):
    f = P + tag + "_" + name + ".json"
    return json.load(open(f)) if os.path.exists(f) else None


pn, c2, c4, bp = load("bench_pn16"), load("bench_cnn2"), load("bench_cnn4"), load("bench_bfv_plain")
if bp:
    txt += f"""
Without the profiler and with the side-stream overlap on: **{bp["value"]:.0f} MulRelin/s** ({bp["ms_per_step"]:.3f} ms), `{tag}_bench_bfv_plain.json`.
"""
sweep = P + tag + "_party_sweep.jsonl"
if os.path.exists(sweep):
    rows = [json.loads(l) for l in open(sweep) if l.strip()]
    p14 = load("bench_pn14")
    txt += f"""
## Party count (the metric is "MulRelin/sec at n parties"): PN15QP880, one GPU

`python3 bench.py --parties k --no-cpu --device-keys --steps 20 --warmup 3` for k = 1 ... 16 (`{tag}_party_sweep.jsonl`; key material written
on the device, 2.7 GB of relinearization keys at k = 16).  The cost per MulRelin is linear in the number of parties, as the
reference's construction promises (≈ 0.28 ms per party + 0.2 ms):

| parties | MulRelin/s | ms per MulRelin | Rotate/s | RotateHoisted/s | Conjugate/s |
|---|---|---|---|---|---|
""" + "\n".join("| %d | %.0f | %.3f | %.0f | %.0f | %.0f |" % (r["config"]["parties"], r["value"], r["ms_per_step"], r["config"]["rotate_per_sec"],
                                                              r["config"]["rotate_hoisted_per_sec"], r["config"]["conjugate_per_sec"]) for r in rows) + "\n"
    if p14:
        txt += f"""
The reference's second parameter set, PN14QP439 (N = 2^14, 7 + 2 limbs), 4 parties: {p14["value"]:.0f} MulRelin/s ({p14["ms_per_step"]:.3f} ms), `{tag}_bench_pn14.json`.
"""
if pn:
    e = pn["config"]
    txt += f"""
## BASELINE.json configs[3] ring on ONE GPU: mkckks 8-party MulRelin + hoisted Rotate, PN16QP1761 (N = 2^16, 34 Q + 4 P primes, α = 2, β = 17)

`python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2` (`{tag}_bench_pn16.json`; keys written on the device by
`mkhe_crs_expand`, 8.1 GB of relinearization keys; bit-exactness at this ring is `tests/test_gpu_fullsize.py`):

* **{pn["value"]:.1f} MulRelin/s** ({pn["ms_per_step"]:.2f} ms per step), Rotate {e["rotate_per_sec"]:.0f}/s, RotateHoisted {e["rotate_hoisted_per_sec"]:.0f}/s,
  relinearization-key generation {e["relin_keygen_per_sec"]:.0f} keys/s, CRS expansion {e["crs_expand_per_sec"]:.0f}/s.
* the N = 2^16 forward NTT = two 2^15-point register-resident sub-transforms per limb; its cross-half radix-2 stage is fused
  into the α = 2 digit spread (`decomp_spread_kernel`), so the streaming pass of the first build (45.4 MulRelin/s) is gone.

| kernel | launches/step | avg µs/launch | ms/step | algorithmic GB/s |
|---|---|---|---|---|
{table(pn)}
"""
if c2 and c4:
    txt += f"""
## BASELINE.json configs[4] caller on one GPU: encrypted CNN inference (cnn/cnn.go), PN14QP433 (N = 2^14, 7 Q + 2 P primes)

`python3 bench.py --scheme cnn --parties 2|4 --steps 20 --warmup 3` (`{tag}_bench_cnn2.json`, `{tag}_bench_cnn4.json`): one step =
Convolution + square + FC1 + square + FC2 = 12 MulRelin, 29 rotations, 17 HoistedForm, 38 additions, 1 MulPtxt on resident
ciphertexts; keys ({c2["config"]["keys_generated"]} resp. {c4["config"]["keys_generated"]} switching-key triples / rotation keys) and CRS generated on the device in {c2["config"]["keygen_s"]:.2f} s / {c4["config"]["keygen_s"]:.2f} s.
Encrypted == plaintext logits to 4e-6: `tests/test_gpu_cnn.py`.

| parties | inference/s | ms per inference | Convolution | Square1 | FC1 | Square2 | FC2 (ms, with a sync per layer) |
|---|---|---|---|---|---|---|---|
| 2 (dataOwner, modelOwner: the reference's setting) | **{c2["value"]:.0f}** | {c2["ms_per_step"]:.2f} | {c2["config"]["layer_ms"]["Convolution"]:.2f} | {c2["config"]["layer_ms"]["Square1"]:.2f} | {c2["config"]["layer_ms"]["FC1"]:.2f} | {c2["config"]["layer_ms"]["Square2"]:.2f} | {c2["config"]["layer_ms"]["FC2"]:.2f} |
| 4 (one owner per layer) | **{c4["value"]:.0f}** | {c4["ms_per_step"]:.2f} | {c4["config"]["layer_ms"]["Convolution"]:.2f} | {c4["config"]["layer_ms"]["Square1"]:.2f} | {c4["config"]["layer_ms"]["FC1"]:.2f} | {c4["config"]["layer_ms"]["Square2"]:.2f} | {c4["config"]["layer_ms"]["FC2"]:.2f} |

`value` is measured with 7 forked engine contexts (independent chains of a layer overlap; `--forks 0`: one stream) and eager
submission (`--graph 1` replays a captured HIP graph: same result, not faster).  The per-layer columns add a sync per layer.
This workload is latency-bound, not bandwidth-bound: ≈ 360 launches per inference, each over a few dozen limbs of
2^14 coefficients (a fraction of the 256 CUs), so the figure of merit is the per-launch latency (≈ 25–30 µs for an NTT launch):

| kernel (2 parties) | launches/step | avg µs/launch | ms/step | algorithmic GB/s |
|---|---|---|---|---|
{table(c2)}
"""
open(P + "README.md", "w").write(txt)
print("wrote profiles/README.md")

#!/usr/bin/env python3
"""Distil gpurun_out/<run>/ (tools/profile_round.sh) into the tracked profiles/ directory under the tag <tag>:
    python tools/collect_profiles.py r3a r3 "PN15QP880 k=4"
and regenerate profiles/README.md (tools/write_profiles_readme.py)."""
import collections, csv, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
G, P = os.path.join(ROOT, "gpurun_out", run), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
for d, name in (("stats_noovl", "kernel_stats_noovl"), ("stats_ovl", "kernel_stats_ovl"), ("stats_bfv", "kernel_stats_bfv"), ("stats_pn16", "kernel_stats_pn16")):
    f = os.path.join(G, d, "p_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_%s.csv" % (tag, name)))
for f in ("bench_plain", "bench_plain_h32", "bench_plain_h16", "bench_noovl", "bench_ovl", "bench_bfv", "bench_bfv_plain", "bench_pn16", "bench_pn16_noovl", "bench_cnn2", "bench_cnn4", "bench_pn14", "bench_2party",
          "bench_cnn4_batch8", "bench_cnn4_batch16", "bench_cnn2_batch8"):
    src = os.path.join(G, f + ".json")
    if os.path.exists(src) and os.path.getsize(src) > 10:
        shutil.copy(src, os.path.join(P, "%s_%s.json" % (tag, f)))
for f, t in (("party_sweep.jsonl", "party_sweep.jsonl"), ("ubench.txt", "ubench.txt"), ("ntt16_bench.txt", "ntt16_launch_sizes.txt"), ("power_probe.txt", "power_probe.txt"),
             ("ntt16_ablation.txt", "ntt16_ablation.txt"), ("ntt_in_context.txt", "ntt_in_context.txt"), ("pn14_batch.jsonl", "pn14_batch.jsonl"),
             ("f2_variants.txt", "f2_variants.txt"), ("f2_fused_ab.txt", "f2_fused_ab.txt"), ("pmc_unfused.txt", "pmc_unfused.txt"), ("f2_trace.txt", "f2_trace.txt"),
             ("dist_5ranks.json", "dist_5ranks.json")):
    src = os.path.join(G, f)
    if os.path.exists(src):
        txt = "\n".join(l for l in open(src).read().split("\n") if not l.startswith(("RCCL", "HIP version", "ROCm version", "Hostname", "Librccl", "/opt/amdgpu")))
        if f == "power_probe.txt":
            # one sample per line: rocm-smi prints every field on a line of its own and the probe joins them
            out_l = []
            for l in open(src).read().split("\n"):
                mc, mp = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l), re.search(r"Power \(W\): ([0-9.]+)", l)
                if mc and mp and "mclk" in l:
                    out_l.append("sample under load: sclk clock level: 1 (%sMhz)   Package Power (W): %s" % (mc.group(1), mp.group(1)))
                elif l.strip():
                    out_l.append(l[:200])
            txt = "\n".join(out_l) + "\n"
        open(os.path.join(P, "%s_%s" % (tag, t)), "w").write(txt)
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "traffic_from_pmc.py"),
                               os.path.join(G, "pmc_fetch"), os.path.join(G, "pmc_write"), workload, "6", "2"] +
                              ([os.path.join(G, "pmc_fetch_h32"), os.path.join(G, "pmc_write_h32")] if os.path.isdir(os.path.join(G, "pmc_fetch_h32")) else []))
open(os.path.join(P, "traffic.json"), "wb").write(out)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("sq_a", "sq_b", "sq_d"):
    cc = os.path.join(G, d, "p_counter_collection.csv")
    if not os.path.exists(cc):
        continue
    for r in csv.DictReader(open(cc)):
        m = re.search(r"mkhe::(?:h16::|h32::)?(\w+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            agg[m.group(1) + (m.group(2) or "").replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(P, "%s_sq_counters.txt" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc (3 passes) over `MKHE_NO_OVERLAP=1 python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras`\n")
    f.write("# mean per dispatch over all dispatches of the kernel; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves\n")
    for k in sorted(agg):
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        f.write("\n== %s   (%d dispatches)\n" % (k, len(next(iter(agg[k].values())))))
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        for n in sorted(c):
            f.write("   %-26s %14.5g   (%.3f of SQ_WAVE_CYCLES)\n" % (n, c[n], c[n] / wc))
        if c.get("SQ_WAVES"):
            f.write("   per wave: VALU %.0f  LDS %.0f  VMEM %.0f  SALU %.0f instructions\n" % tuple(
                c.get(x, 0) / c["SQ_WAVES"] for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU")))
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            f.write("   L2 hit rate %.3f\n" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
# the ablation table must hold numbers (round 5: its variant libraries did not link and nobody noticed)
_abl = os.path.join(P, "%s_ntt16_ablation.txt" % tag)
if os.path.exists(_abl) and len(re.findall(r"us/launch", open(_abl).read())) < 8:
    sys.exit("tools/collect_profiles.py: %s has no timings (variant builds failed on the GPU box?)" % _abl)
# ... and no tracked text file of the round may be a crashed script's output (round 6 carried an _ntt_in_context.txt of tracebacks for a while)
_bad = [x for x in sorted(os.listdir(P)) if x.startswith(tag + "_") and x.endswith((".txt", ".json", ".jsonl")) and "Traceback (most recent call last)" in open(os.path.join(P, x), errors="replace").read()]
if _bad:
    sys.exit("tools/collect_profiles.py: a script crashed into %s" % ", ".join(_bad))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ntt16_isa.py"), tag], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ntt32_isa.py"), tag], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_profiles_readme.py"), tag])
print("profiles/ updated:", sorted(x for x in os.listdir(P) if x.startswith(tag) or x in ("traffic.json", "README.md")))

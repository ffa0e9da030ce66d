#!/usr/bin/env python3
"""Distil one round's gpurun_out/<tag>/ (tools/profile_round.sh) and gpurun_out/<tag>_sq/ (tools/profile_sq.sh)
into the tracked profiles/ directory:

  profiles/<tag>_kernel_stats_noovl.csv   rocprofv3 --kernel-trace --stats, MKHE_NO_OVERLAP=1 (kernels run alone)
  profiles/<tag>_kernel_stats_ovl.csv     same command with the side-stream overlap on (the configuration `value` is measured in)
  profiles/<tag>_kernel_stats_bfv.csv     bench.py --scheme bfv
  profiles/<tag>_bench_*.json             the bench.py lines printed under the profiler / without it
  profiles/<tag>_sq_counters.txt          SQ counter means per kernel (VALU/LDS/VMEM instructions, wait buckets)
  profiles/traffic.json                   HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes

  python tools/collect_profiles.py r1f "PN15QP880 k=4"
"""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
for d, name in (("stats_noovl", "kernel_stats_noovl"), ("stats_ovl", "kernel_stats_ovl"), ("stats_bfv", "kernel_stats_bfv")):
    shutil.copy(os.path.join(G, d, "p_kernel_stats.csv"), os.path.join(P, "%s_%s.csv" % (tag, name)))
for f in ("bench_noovl", "bench_ovl", "bench_plain", "bench_bfv"):
    shutil.copy(os.path.join(G, f + ".json"), os.path.join(P, "%s_%s.json" % (tag, f)))
for d, name in (("stats_pn16", "kernel_stats_pn16"), ("stats_cnn", "kernel_stats_cnn")):
    if os.path.exists(os.path.join(G, d, "p_kernel_stats.csv")):
        shutil.copy(os.path.join(G, d, "p_kernel_stats.csv"), os.path.join(P, "%s_%s.csv" % (tag, name)))
for f in ("bench_pn16", "bench_pn16_noovl", "bench_cnn2", "bench_cnn4", "bench_cnn_noovl", "bench_bfv_plain"):
    if os.path.exists(os.path.join(G, f + ".json")):
        shutil.copy(os.path.join(G, f + ".json"), os.path.join(P, "%s_%s.json" % (tag, f)))
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "traffic_from_pmc.py"),
                               os.path.join(G, "pmc_fetch"), os.path.join(G, "pmc_write"), workload])
open(os.path.join(P, "traffic.json"), "wb").write(out)

S = G + "_sq"
if os.path.isdir(S):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in "abc":
        cc = os.path.join(S, d, "p_counter_collection.csv")
        if not os.path.exists(cc):
            continue
        for r in csv.DictReader(open(cc)):
            m = re.search(r"mkhe::(\w+)(<[^>]*>)?", r["Kernel_Name"])
            if m:
                agg[m.group(1) + (m.group(2) or "").replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(P, "%s_sq_counters.txt" % tag), "w") as f:
        f.write("# rocprofv3 --pmc (3 passes of 8 SQ counters) over `python3 bench.py --steps 4 --warmup 2 --no-cpu`, MKHE_NO_OVERLAP=1\n")
        f.write("# mean per dispatch over the second half of the dispatches; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles summed over waves\n")
        for k in sorted(agg):
            c = {n: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for n, v in agg[k].items()}
            f.write("\n== %s\n" % k)
            wc = c.get("SQ_WAVE_CYCLES", 0) or 1
            for n in sorted(c):
                f.write("   %-26s %14.5g   (%.3f of SQ_WAVE_CYCLES)\n" % (n, c[n], c[n] / wc))
            if c.get("SQ_WAVES"):
                f.write("   per wave: VALU %.0f  LDS %.0f  VMEM %.0f  SALU %.0f instructions\n" % tuple(
                    c.get(x, 0) / c["SQ_WAVES"] for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU")))
print("profiles/ updated:", sorted(os.listdir(P)))

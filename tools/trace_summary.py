#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace taken with the side streams / forks ON: what the GPU did in the second half of the run (settled clocks).
  rocprofv3 --output-format csv --kernel-trace -d gpurun_out/tl -o p -- python3 bench.py ... ; python3 tools/trace_summary.py gpurun_out/tl [steps-in-second-half]
Prints: the window, the time at least one kernel was running (union of the intervals), the idle gaps (count, total, the largest), and per kernel name the
launches, their total and mean duration and how much of their time no OTHER kernel was running (a kernel that is alone on the chip is on the critical path)."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)) if "mkhe" in r["Kernel_Name"]]
rows.sort()
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
mid = t_lo + (t_hi - t_lo) // 2
rows = [r for r in rows if r[0] >= mid]
t0, t1 = rows[0][0], max(r[1] for r in rows)
# sweep: coverage counts
ev = []
for s, e, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0; gaps = []
gap_start = None
for t, d in ev:
    if depth > 0: busy += t - last
    elif t > last: gaps.append(t - last)
    last = t; depth += d
print("window %.1f us   busy %.1f us (%.1f %%)   kernels %d   idle gaps: %d, total %.1f us, max %.1f us" %
      ((t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0), len(rows), len(gaps), sum(gaps) / 1e3, (max(gaps) if gaps else 0) / 1e3))
if steps:
    print("per step: window %.1f us, busy %.1f us, idle %.1f us, kernels %.1f" % ((t1 - t0) / 1e3 / steps, busy / 1e3 / steps, sum(gaps) / 1e3 / steps, len(rows) / steps))
# alone-time per kernel: time during which it is the only kernel running
bounds = sorted(set([r[0] for r in rows] + [r[1] for r in rows]))
import bisect
cover = [0] * (len(bounds) - 1)
idx = {t: i for i, t in enumerate(bounds)}
for s, e, n in rows:
    for i in range(idx[s], idx[e]): cover[i] += 1
tot = defaultdict(float); alone = defaultdict(float); cnt = defaultdict(int)
for s, e, n in rows:
    k = n.replace("mkhe::", "").replace("void ", "")
    k = k[:k.index("(")] if "(" in k else k
    tot[k] += e - s; cnt[k] += 1
    for i in range(idx[s], idx[e]):
        if cover[i] == 1: alone[k] += bounds[i + 1] - bounds[i]
print("%-52s %8s %10s %9s %10s" % ("kernel", "launches", "total us", "mean us", "alone us"))
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-52s %8d %10.1f %9.1f %10.1f" % (k[:52], cnt[k], tot[k] / 1e3, tot[k] / 1e3 / cnt[k], alone[k] / 1e3))

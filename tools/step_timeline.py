#!/usr/bin/env python3
"""Timeline of one MulRelin step from a rocprofv3 kernel trace (overlap on): start, duration, kernel, and the idle gaps.
  rocprofv3 --output-format csv --kernel-trace -d gpurun_out/tl -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras
  python3 tools/step_timeline.py gpurun_out/tl [marker-kernel (default tensor_kernel)]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mkhe" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "tensor_kernel"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
print(marker, "launches", len(idx))
a, b = idx[4], idx[5]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
busy, cur_e, gaps = 0, t0, []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_e:
        gaps.append((s - cur_e, r["Kernel_Name"][:40]))
    busy += max(0, e - max(s, cur_e))
    cur_e = max(cur_e, e)
print("window us %.1f  busy us %.1f  kernels %d" % ((cur_e - t0) / 1e3, busy / 1e3, len(seg)))
for r in seg:
    print("%8.1f %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                               r["Kernel_Name"].replace("mkhe::", "")[:70]))
print("gaps (us, before):", [(round(g / 1e3, 1), n) for g, n in gaps])

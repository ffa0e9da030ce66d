#!/usr/bin/env python3
"""Per-kernel HBM bytes of bench.py --params PN16QP1761 --parties 8 from the PMC passes of tools/pn16_traffic.sh, radix-4 Decompose (MKHE_SPREAD_RADIX4=1,
the default) beside the earlier path (=0).  (2 * FETCH_SIZE + WRITE_SIZE) KB per launch as in tools/traffic_from_pmc.py; launches of one kernel differ in
size (the Decompose launches of a MulRelin, the small tensor-step ones), so the table gives the mean over all launches and the largest launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_from_pmc import per_kernel

root = sys.argv[1]
for mode, title in ((1, "radix-4 spread + one-pass quarter sub-transforms (default)"), (0, "MKHE_SPREAD_RADIX4=0: cross-half stage in the spread, two-pass half sub-transforms out of place")):
    f = per_kernel(os.path.join(root, "fetch_%d" % mode), "FETCH_SIZE")
    w = per_kernel(os.path.join(root, "write_%d" % mode), "WRITE_SIZE")
    print("== %s" % title)
    print("%-34s %8s %14s %14s %14s   %s" % ("kernel", "launches", "read MB (2xF)", "written MB", "total MB", "largest launch: read / written MB"))
    rows = []
    for k in sorted(set(f) | set(w)):
        fm = 2 * sum(f.get(k, [0])) / max(1, len(f.get(k, []))) / 1e3
        wm = sum(w.get(k, [0])) / max(1, len(w.get(k, []))) / 1e3
        rows.append((fm + wm, k, len(f.get(k, [])), fm, wm, 2 * max(f.get(k, [0])) / 1e3, max(w.get(k, [0])) / 1e3))
    for tot, k, n, fm, wm, fx, wx in sorted(rows, reverse=True)[:10]:
        print("%-34s %8d %14.1f %14.1f %14.1f   %.0f / %.0f" % (k, n, fm, wm, tot, fx, wx))
    print()

#!/bin/bash
# kernel trace of a bench line: mean duration per (kernel, grid size) in the second half of the run:  bash tools/trace_by_grid.sh <tag> <bench.py args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --output-format csv --kernel-trace -d $O/tr -o p -- python3 bench.py "$@" --no-cpu > $O/bench.json 2> $O/bench.err
python3 - $O <<'PY' > $O/by_grid.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/tr/**/p_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mkhe" in r["Kernel_Name"]]
t = sorted(int(r["Start_Timestamp"]) for r in rows); mid = t[len(t) // 2]
agg = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) >= mid:
        name = r["Kernel_Name"].replace("mkhe::", "").replace("void ", "").split("(")[0]
        agg[(name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg, key=lambda k: -sum(agg[k])):
    v = agg[k]; print("%-46s grid %-8s wg %-5s  n %5d  mean %7.1f us  min %7.1f" % (k[0][:46], k[1], k[2], len(v), sum(v) / len(v), min(v)))
PY
find $O -name 'p_kernel_trace.csv' -delete; find $O -name '*agent_info*' -delete
cat $O/by_grid.txt

#!/bin/bash
# instruction-cache counters of the Decompose NTT launches (tools/ntt16_bench.py), H32 and H16
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
for v in 1 0; do
  O=$R/gpurun_out/ic_$v; mkdir -p $O
  MKHE_NTT32=$v rocprofv3 --output-format csv --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU -d $O -o p -- python3 tools/ntt16_bench.py 10 > $O/out.txt 2> $O/err.txt
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        if "ntt" not in k: continue
        agg[(k, r.get("Grid_Size",""))][r["Counter_Name"]] += float(r["Counter_Value"])
    for key, c in agg.items():
        print("NTT32=$v", key, " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
done

#!/bin/bash
# SQ counters of the Decompose NTT launches alone (tools/ntt16_bench.py): where do the wave cycles of the kernel go?
#   gpurun -- 'MKHE_NTT32=1 bash tools/sq_ntt.sh tag > gpurun_out/sq_tag.txt 2>&1'
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/a -o p -- python3 tools/ntt16_bench.py 10 > $O/a.txt 2> $O/a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $O/b -o p -- python3 tools/ntt16_bench.py 10 > $O/b.txt 2> $O/b.err
python3 - <<PY
import csv, glob, collections
for part in ("a", "b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % part, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "ntt" not in k: continue
            key = (k, r.get("Grid_Size", ""))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for key, c in agg.items():
            print(part, key, " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY

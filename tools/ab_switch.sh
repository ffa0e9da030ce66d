#!/bin/bash
# one gpurun call: an A/B of ONE switch of the diagnostic library on the headline bench, alternating, with the per-class kernel times of the HIP-event leg:
#   bash tools/ab_switch.sh <tag> <MKHE_NAME> <value> <value> ...      e.g.  bash tools/ab_switch.sh abrev MKHE_EXT_REV 0 1 0 1 0 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; NAME=$2; shift 2
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
i=0
for v in "$@"; do
  i=$((i+1))
  env $NAME=$v python3 bench.py --no-cpu --no-extras ${BENCH_ARGS} > $O/r$i.json 2> $O/r$i.err
  python3 - $O/r$i.json "$NAME=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
print("%-22s %7.1f %s  %.4f ms  | " % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"]) + "  ".join("%s %.1f" % (n.split("<")[0].split(" ")[0][:14], 1e3 * v["ms_per_step"]) for n, v in k.items()))
PY
done

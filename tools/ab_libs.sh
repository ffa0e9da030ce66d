#!/bin/bash
# one gpurun call: the headline bench (no CPU leg, no extras) once per library / environment, alternating as listed, with the per-class kernel times:
#   bash tools/ab_libs.sh <tag> "<name>[:ENV=VAL,...]" ...     name = shipped | switches | a variant of tools/build_variant.sh (build/var_<name>/lib.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
i=0
for v in "$@"; do
  i=$((i+1))
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  lib=$R/mkhe-kklss_amd/build/var_$name/lib.so
  [ "$name" = shipped ] && lib=$R/mkhe-kklss_amd/lib/libmkhe_hip.so
  [ "$name" = switches ] && lib=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
  ( for e in ${envs//,/ }; do export "$e"; done; MKHE_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu --no-extras ${BENCH_ARGS} > $O/r$i.json 2> $O/r$i.err )
  python3 - $O/r$i.json "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print("%-34s %7.1f %s  %.4f ms  | " % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"]) + "  ".join("%s %.1f" % (n.split("<")[0].split(" ")[0][:12], 1e3 * v["ms_per_step"]) for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

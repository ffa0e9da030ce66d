#!/bin/bash
# one gpurun call: ONE switch of the diagnostic library on the small-ring lines (cnn 4 / 2 parties, PN14QP439 MulRelin), alternating:
#   bash tools/ab_small.sh <tag> <MKHE_NAME> <value> <value> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; NAME=$2; shift 2
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
i=0
for v in "$@"; do
  i=$((i+1))
  env $NAME=$v python3 bench.py --scheme cnn --parties 4 --no-cpu > $O/cnn4_$i.json 2> $O/cnn4_$i.err
  env $NAME=$v python3 bench.py --scheme cnn --parties 2 --no-cpu > $O/cnn2_$i.json 2> $O/cnn2_$i.err
  env $NAME=$v python3 bench.py --params PN14QP439 --no-cpu > $O/pn14_$i.json 2> $O/pn14_$i.err
  python3 - $O $i "$NAME=$v" <<'PY'
import json, sys
o, i, tag = sys.argv[1:4]
out = []
for f in ("cnn4", "cnn2", "pn14"):
    d = json.load(open("%s/%s_%s.json" % (o, f, i))); out.append("%s %8.1f /s %.4f ms" % (f, d["value"], d["ms_per_step"]))
print("%-24s " % tag + "   ".join(out))
PY
done

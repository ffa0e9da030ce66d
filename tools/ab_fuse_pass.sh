#!/bin/bash
# one gpurun call: the cross stages at the load (round 5) on / off, same box: cnn (4 and 2 parties), PN14QP439, the headline (side chain only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-abfp}
mkdir -p $O
cd $R
SW=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
for v in on:1 off:0 on2:1; do
  n=${v%%:*}; f=${v#*:}
  MKHE_LIB=$SW MKHE_NTT_FUSE_PASS=$f python3 bench.py --scheme cnn --parties 4 --no-cpu > $O/cnn4_$n.json 2> $O/cnn4_$n.err
  MKHE_LIB=$SW MKHE_NTT_FUSE_PASS=$f python3 bench.py --params PN14QP439 --no-cpu > $O/pn14_$n.json 2> $O/pn14_$n.err
done
MKHE_LIB=$SW MKHE_NTT_FUSE_PASS=0 python3 bench.py --no-cpu --no-extras > $O/head_off.json 2> $O/head_off.err
MKHE_LIB=$SW MKHE_NTT_FUSE_PASS=1 python3 bench.py --no-cpu --no-extras > $O/head_on.json 2> $O/head_on.err
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-16s %9.1f %s  %.4f ms" % (os.path.basename(f)[:-5], d["value"], d["unit"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY

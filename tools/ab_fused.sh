#!/bin/bash
# one gpurun call: forward sub-transforms + inner products in one kernel on the small ring (ext_fused_lds_kernel): off (MKHE_EXT_FUSED_MAX=0), on, and the
# group count NG forced, switches library -- cnn 4 parties, cnn 2 parties, PN14QP439 MulRelin; alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-abfused}
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
run() { n=$1; shift
  env "$@" python3 bench.py --scheme cnn --parties 4 --no-cpu > $O/cnn4_$n.json 2> $O/cnn4_$n.err
  env "$@" python3 bench.py --scheme cnn --parties 2 --no-cpu > $O/cnn2_$n.json 2> $O/cnn2_$n.err
  env "$@" python3 bench.py --params PN14QP439 --no-cpu > $O/pn14_$n.json 2> $O/pn14_$n.err; }
run a_150 MKHE_UNUSED=1
run b_200 MKHE_EXT_FUSED_MAX=200
run c_300 MKHE_EXT_FUSED_MAX=300
run d_100 MKHE_EXT_FUSED_MAX=100
run e_150 MKHE_UNUSED=1
run f_200 MKHE_EXT_FUSED_MAX=200
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-14s %9.1f %s  %.4f ms" % (os.path.basename(f)[:-5], d["value"], d["unit"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY

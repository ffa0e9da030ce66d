#!/bin/bash
# one gpurun call: where the small-launch forms of the N = 2^14 ring hand over (switches library): 2^11-point LDS sub-transforms below MKHE_NTT_LDS11_MAX limbs,
# the H16-class one-pass kernels from MKHE_NTT14_MIN (forward) / MKHE_NTT14_INV_MIN (inverse) limbs up
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ablds}
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
run() { n=$1; shift; env "$@" python3 bench.py --scheme cnn --parties 4 --no-cpu > $O/cnn4_$n.json 2> $O/cnn4_$n.err; env "$@" python3 bench.py --params PN14QP439 --no-cpu > $O/pn14_$n.json 2> $O/pn14_$n.err; }
run base MKHE_UNUSED=1
run l128 MKHE_NTT_LDS11_MAX=128
run l256_m256 MKHE_NTT_LDS11_MAX=256 MKHE_NTT14_MIN=256 MKHE_NTT14_INV_MIN=256
run l512_m512 MKHE_NTT_LDS11_MAX=512 MKHE_NTT14_MIN=512 MKHE_NTT14_INV_MIN=512
run l128_m256 MKHE_NTT_LDS11_MAX=128 MKHE_NTT14_MIN=256 MKHE_NTT14_INV_MIN=256
run base2 MKHE_UNUSED=1
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-20s %9.1f %s  %.4f ms" % (os.path.basename(f)[:-5], d["value"], d["unit"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY

#!/bin/bash
# one gpurun call: the tensor chain forked off after the F1 kernel (MKHE_TENSOR_LATE=1) against at the start of the step (0), switches library, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-abtl}
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
MKHE_TENSOR_LATE=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -x -q -k "mul_and_relin or mulrelin or MulRelin" > $O/tests_late.txt 2>&1; tail -2 $O/tests_late.txt
for i in 1 2 3; do
  for v in 0 1; do
    MKHE_TENSOR_LATE=$v python3 bench.py --no-cpu --no-extras > $O/head_${v}_$i.json 2> $O/head_${v}_$i.err
  done
done
for v in 0 1; do MKHE_TENSOR_LATE=$v python3 bench.py --params PN14QP439 --no-cpu --no-extras > $O/pn14_$v.json 2> $O/pn14_$v.err; done
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-14s %9.1f %s  %.4f ms  frac %.3f" % (os.path.basename(f)[:-5], d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY

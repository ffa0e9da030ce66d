#!/bin/bash
# one gpurun call: limb slices per coefficient of the ModDown kernels (MKHE_MDM_BY / MKHE_MD_BY), switches library, headline bench under the HIP-event leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-abmd}
mkdir -p $O
cd $R
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
for v in 4 1 2 3 7 4; do
  MKHE_MDM_BY=$v MKHE_MD_BY=$v python3 bench.py --no-cpu --no-extras > $O/by_$v.json 2> $O/by_$v.err
  python3 - $O/by_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
md = [v for n, v in k.items() if n.startswith("moddown")][0]
print("by=%s  %7.1f MulRelin/s  %.4f ms   moddown: %.1f launches/step  %.1f us/launch  %.1f us/step" % (sys.argv[2], d["value"], d["ms_per_step"], md["launches_per_step"], md["avg_launch_us"], 1e3 * md["ms_per_step"]))
PY
done

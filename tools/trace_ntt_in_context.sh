#!/bin/bash
# kernel trace of the default bench (overlap off): durations of the Decompose NTT launches inside the MulRelin, 1792- and 896-limb launches apart, and the fused step-F2 launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ctx
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  ( for e in ${envs//,/ }; do export "$e"; done
    MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace -d $O/$name -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/$name.json 2> $O/$name.err )
  python3 - $O/$name/p_kernel_trace.csv $name <<'PY'
import csv, sys, statistics as st
allrows = list(csv.DictReader(open(sys.argv[1])))
def durs(pred):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in allrows if pred(r["Kernel_Name"])]
    return d[len(d) // 2:]                # the second half of the run: settled clocks
d = durs(lambda n: "ntt32_fwd_kernel" in n or "ntt16_fwd_kernel" in n)
f2 = durs(lambda n: "ntt16_f2_kernel" in n)
big = [x for x in d if x > 160]; small = [x for x in d if x <= 160]
def fmt(label, xs):
    return "%s: median %.1f us (min %.1f, %d launches)" % (label, st.median(xs), min(xs), len(xs)) if xs else "%s: no launch" % label
# (round 6: the 896-limb launch of the t_i is gone at four parties -- its transforms run inside ntt16_f2_kernel)
print(sys.argv[2], " ", fmt("1792 limbs", big), " ", fmt("896 limbs", small), " ", fmt("ntt16_f2_kernel", f2))
PY
done

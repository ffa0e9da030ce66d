#!/bin/bash
# kernel trace of the default bench (overlap off): durations of the Decompose NTT launches inside the MulRelin, 1792- and 896-limb launches apart
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
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntt32_fwd_kernel" in r["Kernel_Name"] or "ntt16_fwd_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
d = d[len(d) // 2:]                       # the second half of the run: settled clocks
big = [x for x in d if x > 160]; small = [x for x in d if x <= 160]
print(sys.argv[2], "launches", len(d), "1792 limbs: median %.1f us (min %.1f)" % (st.median(big), min(big)), " 896 limbs: median %.1f us (min %.1f)" % (st.median(small), min(small)),
      " mean of both %.1f" % ((st.mean(big) + st.mean(small)) / 2))
PY
done

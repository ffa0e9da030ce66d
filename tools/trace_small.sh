#!/bin/bash
# kernel trace of a small-ring bench line (second half of the run): per kernel launches, mean duration, time alone on the chip
#   bash tools/trace_small.sh <tag> <bench.py arguments ...>        (environment is passed through: MKHE_LIB=..., switches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --output-format csv --kernel-trace -d $O/tr -o p -- python3 bench.py "$@" --no-cpu > $O/bench.json 2> $O/bench.err
python3 tools/trace_summary.py $O/tr > $O/summary.txt 2>&1
find $O -name 'p_kernel_trace.csv' -delete; find $O -name '*agent_info*' -delete
cat $O/summary.txt

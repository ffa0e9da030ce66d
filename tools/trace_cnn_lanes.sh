#!/bin/bash
# kernel trace of the cnn inference with the chains as lanes (default) and as forks (rounds 1-4), same box: kernels per inference, busy time, gaps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-cnntl}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="rocprofv3 --output-format csv --kernel-trace"
$P -d $O/lanes -o p -- python3 bench.py --scheme cnn --parties 4 --steps 10 --warmup 3 --no-cpu > $O/lanes.json 2> $O/lanes.err
python3 tools/trace_summary.py $O/lanes > $O/lanes_summary.txt 2>&1
$P -d $O/forks -o p -- python3 bench.py --scheme cnn --parties 4 --forks 7 --steps 10 --warmup 3 --no-cpu > $O/forks.json 2> $O/forks.err
python3 tools/trace_summary.py $O/forks > $O/forks_summary.txt 2>&1
for d in lanes forks; do f=$(find $O/$d -name p_kernel_trace.csv | head -1); [ -n "$f" ] && gzip -f "$f"; done
find $O -name '*agent_info*' -delete 2>/dev/null

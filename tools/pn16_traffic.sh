#!/bin/bash
# HBM bytes per launch of the N = 2^16 Decompose kernels, radix-4 path against the earlier one (run on the GPU box):
#   gpurun --timeout 900 -- 'bash tools/pn16_traffic.sh pn16t'   then   python tools/pn16_traffic.py gpurun_out/pn16t > profiles/r3_pn16_traffic.txt
TAG=${1:-pn16t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
export MKHE_NO_OVERLAP=1
P="rocprofv3 --output-format csv --kernel-trace"
B="python3 bench.py --params PN16QP1761 --parties 8 --steps 2 --warmup 1 --no-cpu --no-extras"
for mode in 1 0; do
    export MKHE_SPREAD_RADIX4=$mode
    $P --pmc FETCH_SIZE -d $O/fetch_$mode -o p -- $B > $O/bench_fetch_$mode.json 2> $O/bench_fetch_$mode.err
    $P --pmc WRITE_SIZE -d $O/write_$mode -o p -- $B > $O/bench_write_$mode.json 2> $O/bench_write_$mode.err
    echo "mode $mode done"
done
find $O -name '*agent_info.csv' -delete
find $O -name '*kernel_trace.csv' -delete
du -sh $O

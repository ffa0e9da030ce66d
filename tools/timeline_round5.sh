#!/bin/bash
# one gpurun call: kernel traces with the side streams / forks on (what the GPU really does in a step) for the headline, PN14QP439 and cnn
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-tl5}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="rocprofv3 --output-format csv --kernel-trace"
$P -d $O/head -o p -- python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extras > $O/head.json 2> $O/head.err
python3 tools/trace_summary.py $O/head > $O/head_summary.txt 2>&1
python3 tools/step_timeline.py $O/head > $O/head_timeline.txt 2>&1
$P -d $O/pn14 -o p -- python3 bench.py --params PN14QP439 --steps 40 --warmup 5 --no-cpu --no-extras > $O/pn14.json 2> $O/pn14.err
python3 tools/trace_summary.py $O/pn14 > $O/pn14_summary.txt 2>&1
python3 tools/step_timeline.py $O/pn14 > $O/pn14_timeline.txt 2>&1
$P -d $O/cnn -o p -- python3 bench.py --scheme cnn --parties 4 --steps 10 --warmup 3 --no-cpu > $O/cnn.json 2> $O/cnn.err
python3 tools/trace_summary.py $O/cnn > $O/cnn_summary.txt 2>&1
rm -rf $O/*/*/*.db 2>/dev/null
# keep the traces small enough to come back (64 MiB limit): compress
for d in head pn14 cnn; do f=$(find $O/$d -name p_kernel_trace.csv | head -1); [ -n "$f" ] && gzip -f "$f"; done
find $O -name '*agent_info*' -delete 2>/dev/null
du -sh $O

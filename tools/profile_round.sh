#!/bin/bash
# Round profile set (run on the GPU box through gpurun): kernel-trace stats of bench.py with and without the
# side-stream overlap, and the two HBM-traffic PMC passes (separate runs, kernel-trace only, as the guide prescribes).
#   gpurun -- 'bash tools/profile_round.sh r1f'
TAG=${1:-r1x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_noovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $O/bench_noovl.json 2> $O/bench_noovl.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_ovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $O/bench_ovl.json 2> $O/bench_ovl.err
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_bfv -o p -- python3 bench.py --scheme bfv --steps 10 --warmup 2 --no-cpu > $O/bench_bfv.json 2> $O/bench_bfv.err
python3 bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
# secondary workloads: the configs[3] ring on one GPU (N = 2^16, 8 parties, keys written on the device) and the encrypted CNN
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_pn16 -o p -- python3 bench.py --params PN16QP1761 --parties 8 --steps 6 --warmup 2 > $O/bench_pn16_noovl.json 2> $O/bench_pn16_noovl.err
python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2 > $O/bench_pn16.json 2> $O/bench_pn16.err
MKHE_NO_OVERLAP=1 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_cnn -o p -- python3 bench.py --scheme cnn --parties 2 --steps 10 --warmup 2 > $O/bench_cnn_noovl.json 2> $O/bench_cnn_noovl.err
python3 bench.py --scheme cnn --parties 2 --steps 20 --warmup 3 > $O/bench_cnn2.json 2> $O/bench_cnn2.err
python3 bench.py --scheme cnn --parties 4 --steps 20 --warmup 3 > $O/bench_cnn4.json 2> $O/bench_cnn4.err
python3 bench.py --scheme bfv --steps 10 --warmup 2 > $O/bench_bfv_plain.json 2> $O/bench_bfv_plain.err
find $O/stats_pn16 $O/stats_cnn -name '*kernel_trace.csv' -delete
# keep the merge small: drop the per-dispatch traces of the stats runs, keep their *_stats.csv
find $O/stats_noovl $O/stats_ovl $O/stats_bfv -name '*kernel_trace.csv' -delete
du -sh $O
ls -R $O | head -50

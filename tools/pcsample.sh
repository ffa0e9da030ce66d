#!/bin/bash
# PC sampling of the headline bench (rocprofv3 beta feature): where do the waves of a kernel sit?  bash tools/pcsample.sh <tag> [method] [interval]
TAG=${1:-pcs}; M=${2:-host_trap}; I=${3:-1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
U=time; [ "$M" = stochastic ] && U=cycles
timeout -k 10 240 rocprofv3 --output-format csv --kernel-trace --pc-sampling-beta-enabled 1 --pc-sampling-unit $U --pc-sampling-method $M --pc-sampling-interval $I -d $O -o p -- python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extras > $O/bench.json 2> $O/err.txt
echo "rc=$?"; tail -3 $O/err.txt; ls -la $O $O/* | head -30

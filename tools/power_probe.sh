#!/bin/bash
# Is the dominant kernel power-bound?  Runs the 1792 / 896-limb Decompose launches back to back for a few seconds (tools/ntt16_bench.py with
# many repetitions) and samples rocm-smi (socket power, shader clock) beside it.
#   gpurun -- 'bash tools/power_probe.sh > gpurun_out/power_probe.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "Power|sclk|mclk|fclk|Max" | head -12
echo "--- under load"
python3 $R/tools/ntt16_bench.py ${REPS:-30000} > /tmp/pp_bench.txt 2>&1 &
BP=$!
sleep 5
for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
    sleep 0.5
done
wait $BP
grep -E "limbs +(1792|896) " /tmp/pp_bench.txt | cut -c1-120

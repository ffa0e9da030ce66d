#!/bin/bash
# bench.py --scheme cnn over batch sizes and fork counts (one JSON line each -> one summary line each)
#   gpurun -- 'bash tools/cnn_batch_sweep.sh "8:0 8:1 8:3 8:7 16:3" > gpurun_out/cnn_sweep.txt'
for bf in ${1:-"1:7 8:3"}; do
    B=${bf%%:*}; F=${bf#*:}
    python3 bench.py --scheme cnn --parties ${PARTIES:-4} --steps ${STEPS:-20} --warmup 3 --no-cpu --batch $B --batch-forks $F --forks ${FORKS:-7} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('cnn parties %d batch %2d forks %d: %7.1f inferences/s  %6.2f ms/step  host issue %.2f ms  check %s' % (c['parties'], c['batch'], $F if c['batch'] > 1 else c['forks'], d['value'], d['ms_per_step'], c['host_issue_ms'], c['batch_check']))"
done

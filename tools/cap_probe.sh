#!/bin/bash
# which forward kernel wins inside the MulRelin on THIS part, beside the part's power cap (one short gpurun call; run it several times)
echo "power cap: $(rocm-smi --showmaxpower 2>/dev/null | grep -i 'max' | head -1)"
for v in h16:MKHE_NTT32=0 h32:MKHE_NTT32=1 h16:MKHE_NTT32=0 h32:MKHE_NTT32=1; do
  name=${v%%:*}; e=${v#*:}
  env $e python3 bench.py --no-cpu --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$name', 'MulRelin/s %.0f' % d['value'], 'frac %.3f' % r['frac'], 'avg us %.1f' % r['avg_launch_us'])"
done
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | head -2

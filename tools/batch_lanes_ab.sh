#!/bin/bash
# A/B of mkhe_mul_relin_batch at N = 2^15: B evaluations in flight (the product library: this context and two internal ones, round robin) against B in
# lock step (rounds 4-5; variant library: tools/build_variant.sh nolanes batch -DMKHE_BATCH_LANES=0), beside one evaluation at a time.  Same call, same box.
#   gpurun -- 'bash tools/batch_lanes_ab.sh > gpurun_out/batch_lanes.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/mkhe-kklss_amd/build/var_nolanes/lib.so
[ -f $V ] || { echo "missing $V"; exit 2; }
one() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d['config']
print('   %-34s %8.1f MulRelin/s   (one input at a time, same run: %s)' % ('$2', d['value'], round(c.get('mulrelin_per_sec_single_input_same_run') or 0, 1)))"; }
for k in ${PARTIES:-4 2 1}; do
  for B in ${BATCHES:-2 3 4 6}; do
    echo "== parties $k  B = $B"
    python3 $R/bench.py --parties $k --batch $B --no-cpu --no-extras --steps 40 --warmup 5 > /tmp/bl_a.json 2>/tmp/bl.err || { tail -3 /tmp/bl.err; exit 1; }
    one /tmp/bl_a.json "in flight (product library)"
    MKHE_LIB=$V python3 $R/bench.py --parties $k --batch $B --no-cpu --no-extras --steps 40 --warmup 5 > /tmp/bl_b.json 2>/tmp/bl.err || { tail -3 /tmp/bl.err; exit 1; }
    one /tmp/bl_b.json "lock step (-DMKHE_BATCH_LANES=0)"
  done
done

#!/bin/bash
# A/B of two builds on ONE GPU box (boxes of the pool differ by several per cent): the working tree against an older commit
# checked out and built under _r1/ beforehand:   git archive <commit> | tar -x -C _r1 && make -C _r1/mkhe-kklss_amd/csrc && make -C _r1/oracle
#   gpurun -- 'bash tools/ab_r1_r2.sh'
for cfg in "" "--scheme bfv" "--params PN16QP1761 --parties 8 --steps 6 --warmup 2" "--params PN14QP439" "--parties 1 --device-keys" "--parties 2 --device-keys" "--parties 8 --device-keys" "--scheme cnn --parties 4"; do
  for w in R2 R1; do
    d=.; [ $w = R1 ] && d=_r1
    (cd $d && python bench.py $cfg --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('$w', '$cfg', '|', round(j['value'],1), j['unit'], round(j['ms_per_step'],3), 'rot', round(c.get('rotate_per_sec',0)), 'roth', round(c.get('rotate_hoisted_per_sec',0)), 'conj', round(c.get('conjugate_per_sec',0)))")
  done
done

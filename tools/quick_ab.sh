#!/bin/bash
# one gpurun call: launch-size table (default and MKHE_NTT32=0), the prefetch variant, then the headline / secondary bench lines
mkdir -p gpurun_out/ab
( echo "== default"; python3 tools/ntt16_bench.py 1500 | grep -E "limbs" | cut -c1-150
  echo "== MKHE_NTT32=0"; MKHE_NTT32=0 python3 tools/ntt16_bench.py 1500 | grep -E "limbs" | cut -c1-150 ) > gpurun_out/ab/sizes.txt 2>&1
REPS=1500 bash tools/run_variants.sh pf shipped > gpurun_out/ab/variants.txt 2>&1
python3 bench.py > gpurun_out/ab/bench_plain.json 2> gpurun_out/ab/bench_plain.err
MKHE_NTT32=0 python3 bench.py --no-cpu > gpurun_out/ab/bench_h16.json 2> gpurun_out/ab/bench_h16.err
python3 bench.py --params PN14QP439 --no-cpu --no-extras --steps 50 --warmup 5 > gpurun_out/ab/bench_pn14.json 2>/dev/null
python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2 --no-cpu > gpurun_out/ab/bench_pn16.json 2>/dev/null
python3 bench.py --scheme bfv --no-cpu > gpurun_out/ab/bench_bfv.json 2>/dev/null
python3 bench.py --scheme cnn --parties 4 --no-cpu --steps 20 --warmup 3 > gpurun_out/ab/bench_cnn4.json 2>/dev/null
python3 bench.py --scheme cnn --parties 4 --batch 8 --no-cpu --steps 20 --warmup 3 > gpurun_out/ab/bench_cnn4_b8.json 2>/dev/null
cat gpurun_out/ab/sizes.txt gpurun_out/ab/variants.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", r.get("frac"), (r.get("kernel") or "")[:28], "avg_us", r.get("avg_launch_us"))
PY

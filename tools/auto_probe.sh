#!/bin/bash
# the measured kernel choice (MKHE_NTT32=2, default) beside both forced settings: one gpurun call
mkdir -p gpurun_out/auto
for v in auto:MKHE_UNUSED=1 h32:MKHE_NTT32=1 h16:MKHE_NTT32=0 auto2:MKHE_UNUSED=1; do
  name=${v%%:*}; e=${v#*:}
  env $e python3 bench.py --no-cpu > gpurun_out/auto/$name.json 2> gpurun_out/auto/$name.err
done
env python3 bench.py --scheme bfv --no-cpu > gpurun_out/auto/bfv_auto.json 2> gpurun_out/auto/bfv_auto.err
MKHE_NTT32=0 python3 bench.py --scheme bfv --no-cpu > gpurun_out/auto/bfv_h16.json 2> gpurun_out/auto/bfv_h16.err
MKHE_NTT32=1 python3 bench.py --scheme bfv --no-cpu > gpurun_out/auto/bfv_h32.json 2> gpurun_out/auto/bfv_h32.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/auto/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "frac", round(r.get("frac") or 0, 4), (r.get("kernel") or "")[:24], "avg_us", round(r.get("avg_launch_us") or 0, 1), d["config"].get("ntt_kernel_choice"), "traffic", r.get("traffic"))
PY

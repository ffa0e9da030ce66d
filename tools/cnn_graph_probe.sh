#!/bin/bash
# batched cnn inference replayed from a HIP graph beside eager issue, and the per-class breakdown of the PN14QP439 batched MulRelin (one gpurun call)
mkdir -p gpurun_out/probe
for b in 8 16; do
  for g in 0 1; do
    timeout -k 10 300 python bench.py --scheme cnn --parties 4 --batch $b --graph $g --steps 20 --warmup 3 --no-cpu > gpurun_out/probe/cnn4_b${b}_g${g}.json 2> gpurun_out/probe/cnn4_b${b}_g${g}.err || echo "cnn b=$b g=$g failed"
  done
done
for b in 1 4 8; do
  timeout -k 10 300 python bench.py --params PN14QP439 --parties 4 --batch $b --steps 50 --warmup 5 --no-cpu --no-extras > gpurun_out/probe/pn14_b${b}.json 2> gpurun_out/probe/pn14_b${b}.err || echo "pn14 b=$b failed"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/probe/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    c = d["config"]
    print(f, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "graph", c.get("hip_graph"), "issue_ms", c.get("host_issue_ms"), c.get("batch_check"))
    if "pn14" in f:
        for k, v in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print("    %-50s %5.1f x %7.1f us = %.3f ms" % (k[:50], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]))
PY

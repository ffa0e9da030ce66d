#!/usr/bin/env python3
"""one line per kernel class of a bench.py JSON line read from stdin (diagnostics)"""
import json, sys
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    print("%.1f %s  %.3f ms/step" % (j["value"], j["unit"], j["ms_per_step"]))
    for k, v in j["roofline"]["kernels"].items():
        print("   %-44s %5.1f x %8.1f us = %7.3f ms  %6.0f GB/s" % (k[:44], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"], v["achieved_GBs"]))

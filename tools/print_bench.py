#!/usr/bin/env python3
"""print the essentials of bench.py JSON lines found in the given log files"""
import json
import sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            ee = {k: (round(v["value"] / 1e6, 3), round(v["mean_iters"], 2)) for k, v in j.get("early_exit", {}).items()}
            print("%s: %.4g frames/s (%.2f ms kernel, frac %.3f, traffic %s) early-exit %s layout %s" % (
                f, j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["roofline"]["traffic"], ee, j["config"]["layout"]))

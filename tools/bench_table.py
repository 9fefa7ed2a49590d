#!/usr/bin/env python3
"""Print value, ms/step and the per-kernel table of a bench.py JSON line.  Usage: tools/bench_table.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("%.0f %s  %.3f ms/step" % (d["value"], d["unit"], d["ms_per_step"]))
for k in d.get("kernels", []):
    print("%-32s %.3f" % (k["kernel"], k["ms_per_launch"]))

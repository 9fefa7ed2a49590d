#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (stdin): headline + per-kernel table."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  n_gpus={d['n_gpus']}")
for k in d.get("kernels", []):
    print(f"  {k['kernel']:28s} {k['ms_per_launch']:8.3f} ms  {k['alg_GBps']:8.0f} GB/s alg")
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"])

#!/usr/bin/env python3
"""A/B timing of library builds: runs bench.py's kernel pass (C3, no CPU leg, no extras) once per library given and prints
the per-kernel launch times side by side.  usage: python tools/ab_kernels.py [lib.so ...]   (the in-tree build comes first)
Variants that change results on purpose (timing experiments) fail bench.py's parity check; that is ignored here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [None] + sys.argv[1:]
rows = {}
steps = {}
for lib in libs:
    env = dict(os.environ)
    name = "in-tree" if lib is None else os.path.basename(lib)
    if lib:
        env["VSLAM_AMD_LIB"] = os.path.abspath(lib)
        env["VSLAM_BENCH_ALLOW_DEGENERATE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-pairs", "0",
                        "--cpu-all-cores-pairs", "0", "--no-extras"], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", p.stderr[-2000:])
        continue
    d = json.loads(line[-1])
    steps[name] = d["ms_per_step"]
    for k in d.get("kernels", []):
        rows.setdefault(k["kernel"], {})[name] = k["ms_per_launch"] * k["launches_per_step"]
names = list(steps)
print("%-28s" % "kernel", *["%16s" % n[:16] for n in names])
for k, v in sorted(rows.items(), key=lambda kv: -max(kv[1].values())):
    print("%-28s" % k, *["%16.4f" % v.get(n, float("nan")) for n in names])
print("%-28s" % "ms_per_step", *["%16.4f" % steps[n] for n in names])

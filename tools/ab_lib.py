#!/usr/bin/env python3
"""TRAP (round 5): pipelines made one after the other in ONE process do not get the same hardware queues -- three identical
ones ran at 2.917 / 2.669 / 2.749 ms per batch -- so this tool cannot compare anything that touches streams or their order of
creation; use tools/ab_proc.sh (fresh process per configuration) for that.

Step time of TWO BUILDS of the library in one process, alternating in short blocks (clock drift and box-to-box
differences cancel): each build runs the bench's arrangement -- a vslam_pipeline with N batches in flight -- on the hard and
the easy data at C3 (or C5 / C2).
    python tools/ab_lib.py tools/_ab/base.so vslam_amd/libvslam_amd.so [C3] [in_flight] [rounds]
A baseline build: `git stash; python -m vslam_amd.build; cp vslam_amd/libvslam_amd.so tools/_ab/base.so; git stash pop`."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import capi, shard, synth  # noqa: E402

specs = [a.split(":", 1) for a in sys.argv[1:3]]   # lib.so[:NAME=VALUE,...]: environment its contexts read when they are made
paths = [os.path.abspath(sp[0]) for sp in specs]
envs = [dict(kv.split("=", 1) for kv in sp[1].split(",")) if len(sp) > 1 else {} for sp in specs]
wl = sys.argv[3] if len(sys.argv) > 3 else "C3"
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
data = {"hard": synth.frames_torch_hard(0x5EED0002, P, w, h, dev), "easy": synth.frames_torch(0x5EED0002, P, w, h, dev)}
pipes = []
for p, e in zip(paths, envs):
    os.environ.update(e)
    pipes.append(capi.Pipeline(0, depth, lib=capi.load_library(p)))
    for k in e:
        os.environ.pop(k)
outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(depth)]
steps = 30 if wl != "C5" else 9


def run(pp, bgr, n):
    for i in range(n):
        t, c = pp.acquire()
        c.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=outs[i % depth])
        pp.commit(t)
    pp.drain()


res = {}
ref = {}
for rnd in range(rounds + 1):
    for kind, bgr in data.items():
        for which, pp in enumerate(pipes):
            run(pp, bgr, 2 * depth)
            t0 = time.perf_counter()
            run(pp, bgr, steps)
            ms = (time.perf_counter() - t0) / steps * 1e3
            if rnd:
                res.setdefault((kind, which), []).append(ms)
            sig = tuple(outs[0][k].cpu().numpy().tobytes() for k in ("n", "best", "F"))
            if ref.setdefault(kind, sig) != sig:
                print("OUTPUTS DIFFER between the builds on", kind, "data", flush=True)
for kind in data:
    a, b = np.array(res[(kind, 0)]), np.array(res[(kind, 1)])
    print(f"{wl} {kind}, {depth} in flight: A {a.mean():.4f} ms (+-{a.std():.4f})   B {b.mean():.4f} ms (+-{b.std():.4f})   "
          f"B - A = {b.mean() - a.mean():+.4f} ms ({(b.mean() / a.mean() - 1) * 100:+.2f} %)", flush=True)

#!/usr/bin/env python3
"""Per-kernel launch times (HIP events, one context, one batch after the other) of TWO BUILDS of the library in one process,
alternating.      python tools/ab_lib_kernels.py tools/_ab/base.so vslam_amd/libvslam_amd.so [C3] [hard|easy] [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import capi, shard, synth  # noqa: E402

# a build may carry environment settings that its contexts read when they are made: lib.so:NAME=VALUE,NAME=VALUE
specs = [a.split(":", 1) for a in sys.argv[1:3]]
paths = [os.path.abspath(sp[0]) for sp in specs]
envs = [dict(kv.split("=", 1) for kv in sp[1].split(",")) if len(sp) > 1 else {} for sp in specs]
wl = sys.argv[3] if len(sys.argv) > 3 else "C3"
kind = sys.argv[4] if len(sys.argv) > 4 else "hard"
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 5
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
bgr = (synth.frames_torch_hard if kind == "hard" else synth.frames_torch)(0x5EED0002, P, w, h, dev)
pipes = []
for p, e in zip(paths, envs):
    os.environ.update(e)
    pipes.append(capi.Pipeline(0, 1, lib=capi.load_library(p)))
    for k in e:
        os.environ.pop(k)
out = capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev)
acc = [{}, {}]
for rnd in range(rounds + 1):
    for which, pp in enumerate(pipes):
        c = pp.contexts[0]
        c.prof_enable(True)
        c.prof_reset()
        for _ in range(3):
            c.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=out)
        rep = c.prof_report()
        c.prof_enable(False)
        if rnd:
            for k, (ms, cnt) in rep.items():
                acc[which].setdefault(k, []).append(ms / 3)
print(f"{wl} {kind}: ms per step and kernel slot, A = {sys.argv[1]}, B = {sys.argv[2]}")
tot = [0.0, 0.0]
for k in sorted(acc[0], key=lambda k: -np.mean(acc[0][k])):
    a, b = np.mean(acc[0][k]), np.mean(acc[1].get(k, [float("nan")]))
    tot[0] += a
    tot[1] += b
    print(f"{k:28s} {a:8.4f} {b:8.4f} {b - a:+8.4f}")
print(f"{'sum':28s} {tot[0]:8.4f} {tot[1]:8.4f} {tot[1] - tot[0]:+8.4f}")

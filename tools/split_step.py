#!/usr/bin/env python3
"""Would splitting ONE batch over several contexts of the same device (each its own streams and workspaces, all started
together, the step ends when the last one has) beat one context running the whole batch?  The latency-bound stages of one
part could then fill the holes of another, as with whole batches in flight (tools/inflight.py), but inside a step.
Alternating blocks in one process (boxes differ by several per cent).     python tools/split_step.py [parts ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

w, h, K, H, P = 1280, 720, 2000, 4096, 256
dev = torch.device("cuda", 0)
bgr = synth.frames_torch(0x5EED0002, P, w, h, dev)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
configs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
state = {}
for n in configs:
    ctxs = [Context(0, use_torch_stream=False) for _ in range(n)]
    parts = []
    for r in range(n):
        lo, hi = shard.shard_range(P, r, n)
        parts.append((torch.cat([bgr[lo:hi], bgr[P + lo:P + hi]]).contiguous(), hi - lo, seeds[lo:hi].contiguous()))
    state[n] = (ctxs, parts, [None] * n)


def step(n):
    ctxs, parts, outs = state[n]
    for r in range(n):
        b, p, s = parts[r]
        outs[r] = ctxs[r].frontend_pairs(b, p, K, ca, sa, pat, s, H, 10.0, out=outs[r])
    for c in ctxs:
        c.synchronize()


for n in configs:
    for _ in range(3):
        step(n)
times = {n: [] for n in configs}
for block in range(6):
    for n in configs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step(n)
        times[n].append((time.perf_counter() - t0) / 8 * 1e3)
ref = state[configs[0]][2]
for n in configs:
    outs = state[n][2]
    best = torch.cat([o["best"] for o in outs])
    F = torch.cat([o["F"] for o in outs])
    ref_best = torch.cat([o["best"] for o in ref])
    ref_F = torch.cat([o["F"] for o in ref])
    same = torch.equal(best, ref_best) and torch.equal(F.view(torch.int32), ref_F.view(torch.int32))
    t = np.array(times[n])
    print(f"{n} part(s): median {np.median(t):.3f} ms / step (min {t.min():.3f}, max {t.max():.3f}); results equal to the first configuration's: {same}", flush=True)

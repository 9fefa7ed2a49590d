#!/usr/bin/env python3
"""Batches in flight, two ways, one process, alternating: free-running contexts (every batch queued at once, as
tools/inflight.py did in round 4) against the product's vslam_pipeline (acquire waits for the batch n tickets back), on
the easy and the hard data.      python tools/inflight_ab.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, capi, shard, synth  # noqa: E402

w, h, K, H, P = 1280, 720, 2000, 4096, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 36
dev = torch.device("cuda", 0)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
data = {"easy": synth.frames_torch(0x5EED0002, P, w, h, dev), "hard": synth.frames_torch_hard(0x5EED0002, P, w, h, dev)}
ctxs = [Context(0, use_torch_stream=False) for _ in range(4)]
pipes = {n: capi.Pipeline(0, n) for n in (1, 2, 3, 4)}
outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(4)]


def free(n, bgr):
    for i in range(2 * n):
        ctxs[i % n].frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ctxs[i % n].frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
    for c in ctxs[:n]:
        c.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def piped(n, bgr):
    pp = pipes[n]
    for i in range(2 * n):
        t, c = pp.acquire()
        c.frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
        pp.commit(t)
    pp.drain()
    t0 = time.perf_counter()
    for i in range(steps):
        t, c = pp.acquire()
        c.frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
        pp.commit(t)
    pp.drain()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(3):
    for kind, bgr in data.items():
        row = []
        for n in (1, 2, 3, 4):
            row.append(f"n={n}: free {free(n, bgr):.3f} pipe {piped(n, bgr):.3f}")
        print(kind, " | ".join(row), flush=True)

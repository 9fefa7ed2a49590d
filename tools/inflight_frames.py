#!/usr/bin/env python3
"""Does it matter whether the batches in flight read the SAME frames or each its own?  (bench.py gives every context its own
batch; the round-4 measurement shared one.)      python tools/inflight_frames.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import capi, shard, synth  # noqa: E402

w, h, K, H, P = 1280, 720, 2000, 4096, 256
steps = 36
dev = torch.device("cuda", 0)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
own = [synth.frames_torch_hard(0x5EED0002 + 7919 * s, P, w, h, dev) for s in range(3)]
copies = [own[0]] + [own[0].clone() for _ in range(2)]      # same content, different memory
pipe = capi.Pipeline(0, 3)
outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(3)]


def run(frames):
    for i in range(6):
        t, c = pipe.acquire()
        c.frontend_pairs(frames[i % 3], P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % 3])
        pipe.commit(t)
    pipe.drain()
    t0 = time.perf_counter()
    for i in range(steps):
        t, c = pipe.acquire()
        c.frontend_pairs(frames[i % 3], P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % 3])
        pipe.commit(t)
    pipe.drain()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(4):
    print(f"shared {run([own[0]] * 3):.3f}  copies {run(copies):.3f}  own content {run(own):.3f}  "
          f"each content alone: {run([own[0]] * 3):.3f} {run([own[1]] * 3):.3f} {run([own[2]] * 3):.3f}", flush=True)

#!/usr/bin/env python3
"""How much does a second (third) batch in flight buy?  N contexts (own streams + workspaces), full C3 batches handed to
them round-robin without waiting in between; time for K batches / K.      python tools/inflight.py [N ...]"""
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
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    ctxs = [Context(0, use_torch_stream=False) for _ in range(n)]
    outs = [None] * n
    for rep in range(2):
        steps = 24
        for i in range(2 * n):
            outs[i % n] = ctxs[i % n].frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            outs[i % n] = ctxs[i % n].frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0, out=outs[i % n])
        for c in ctxs:
            c.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print(f"in flight {n}: {dt:.3f} ms / batch = {P / dt:.1f} k pairs/s", flush=True)
    same = all(torch.equal(outs[0][k], o[k]) for o in outs[1:] for k in ("best", "n", "F", "matches"))
    print("  outputs of the contexts identical:", same, flush=True)
    del ctxs, outs

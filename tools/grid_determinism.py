#!/usr/bin/env python3
"""The grid ORB/FAST extractor on the same batch again and again, beside another context that keeps the chip busy with the
live path: every output of every run must equal the first run's, bit for bit, compared on the device.  What a race inside the
rebuilt extractor (LDS atomics of the FAST kernel, wave-wide selections, the half-wave patches of the descriptor kernel) or a
timing-dependent hazard would eventually produce.      python tools/grid_determinism.py [runs=2000] [frames=64]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vslam_amd import Context, shard, synth  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
P = frames // 2
bgr = synth.frames_torch_hard(0x5EED0003, P, 1280, 720, dev)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
a, b = Context(0, use_torch_stream=False), Context(0, use_torch_stream=False)
ref = a.extract_features_grid(bgr, 4, 4, pat, 8192)
a.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
out = {k: torch.zeros_like(v) for k, v in ref.items()}
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(3, 0, P).view(np.int32)).to(dev)
other = synth.frames_torch_hard(0x5EED0004, P, 1280, 720, dev)
bo = b.frontend_pairs(other, P, 2000, ca, sa, None, seeds, 4096, 10.0)
torch.cuda.synchronize()
t0, bad = time.time(), 0
for r in range(runs):
    if r % 3 == 0:
        b.frontend_pairs(other, P, 2000, ca, sa, None, seeds, 4096, 10.0, out=bo)   # company on the chip, on its own streams
    a.extract_features_grid(bgr, 4, 4, pat, 8192, out=out)
    a.synchronize()
    n = int(ref["n"].max())
    same = all(torch.equal(out[k][:, :n] if out[k].dim() > 1 else out[k], ref[k][:, :n] if ref[k].dim() > 1 else ref[k]) for k in ref)
    bad += 0 if same else 1
    if r % 500 == 499:
        print(f"{r + 1} runs, {bad} differing, {time.time() - t0:.0f} s", flush=True)
b.synchronize()
print(f"grid determinism: {runs} runs of {frames} frames ({runs * frames} frame extractions), {bad} differing from the first")
sys.exit(1 if bad else 0)

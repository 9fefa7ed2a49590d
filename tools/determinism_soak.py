#!/usr/bin/env python3
"""The same batch through the whole front-end N times: every output of every run must equal the first run's, bit for bit
(races, timing-dependent pipeline hazards and stale workspaces all show up as a difference sooner or later; the parity tests
compare single runs with the oracle).  Contexts alternate so that runs also overlap each other on the device.
usage: python tools/determinism_soak.py [runs=2000] [C3|C2|C5] [contexts=2]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import Context, shard, synth  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
bgr = synth.frames_torch(0x5EED0002, P, w, h, dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
ctxs = [Context(0, use_torch_stream=False) for _ in range(nctx)]
outs = [None] * nctx
ref = ctxs[0].frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0)
ctxs[0].synchronize()
ref = {k: v.clone() for k, v in ref.items()}
keys = sorted(ref)
bad, t0, last = 0, time.time(), time.time()
for i in range(runs):
    c = i % nctx
    if outs[c] is not None:      # the run handed to this context nctx runs ago
        ctxs[c].synchronize()
        for k in keys:
            if not torch.equal(outs[c][k], ref[k]):
                bad += 1
                print(f"run {i - nctx}: output '{k}' differs in {(outs[c][k] != ref[k]).sum().item()} elements", flush=True)
    outs[c] = ctxs[c].frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=outs[c])
    if time.time() - last > 30:
        last = time.time()
        print(f"{i + 1} runs, {bad} differences, {time.time() - t0:.0f} s", flush=True)
for c in range(nctx):
    ctxs[c].synchronize()
    for k in keys:
        if outs[c] is not None and not torch.equal(outs[c][k], ref[k]):
            bad += 1
print(f"determinism soak {wl}: {runs} runs on {nctx} context(s), {bad} differences, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

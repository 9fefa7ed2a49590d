#!/usr/bin/env python3
"""The same batch through a vslam_pipeline again and again (three batches in flight, the arrangement the bench runs): every
output of every run must equal a single-context reference run's, bit for bit, compared on the device.  What a race between
the contexts, a stale workspace or a timing-dependent hazard of the shared-chip arrangement would eventually produce.
    python tools/pipeline_determinism.py [runs=3000] [C3|C2|C5] [in_flight=3] [hard|easy]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import Context, capi, shard, synth  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kind = sys.argv[4] if len(sys.argv) > 4 else "hard"
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
bgr = (synth.frames_torch_hard if kind == "hard" else synth.frames_torch)(0x5EED0002, P, w, h, dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
one = Context(0)
ref = one.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0)
one.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
one.close()
pipe = capi.Pipeline(0, depth)
outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(depth)]
tickets = [None] * depth
bad = checked = 0
t0 = time.time()
last_print = t0


def check(o):
    global bad, checked
    checked += 1
    for key in ("n", "best", "F", "xy", "desc", "nodes"):
        if not torch.equal(o[key], ref[key]):
            bad += 1
            print(f"run {checked}: {key} differs", flush=True)
            return
    nb = ref["best"][:, 3]                      # inlier matches: the valid prefix of each pair's list
    idx = torch.arange(K, device=dev)[None, :] < nb[:, None]
    if not torch.equal(o["matches"][idx], ref["matches"][idx]):
        bad += 1
        print(f"run {checked}: matches differ", flush=True)


for i in range(runs):
    s = i % depth
    if tickets[s] is not None:
        pipe.wait(tickets[s])
        check(outs[s])
    tickets[s] = pipe.submit_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, outs[s])
    if time.time() - last_print > 30:
        last_print = time.time()
        print(f"  {i} batches submitted, {checked} compared, {bad} differing", flush=True)
for s in range(depth):
    if tickets[s] is not None:
        pipe.wait(tickets[s])
        check(outs[s])
print(f"{wl} {kind}, {depth} in flight: {checked} batches ({checked * P} frame pairs) compared with a single-context run, {bad} differing, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Step-time A/B inside ONE process: configurations (context options) alternate in short blocks of steps, so clock drift and
box-to-box differences cancel.  usage: python tools/ab_step.py [C3|C5] [rounds]
Prints mean and spread of the step time per configuration."""
import os
os.environ.setdefault('VSLAM_AMD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vslam_amd', 'libvslam_amd_exp.so'))   # the knobs exist in the EXPERIMENTS build only
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import Context, shard, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
ctx = Context(0)
bgr = synth.frames_torch(0x5EED0002, P, w, h, dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
CONFIGS = {
    "fp4 (default)": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0},
    "fp4, trees in front": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, ctx.OPT_TREE_FORK: 0},
    "int8 4x64 (round 3)": {ctx.OPT_MATCH_FORM: 2, ctx.OPT_MATCH_SHAPE: 2},
    "int8 8x32": {ctx.OPT_MATCH_FORM: 2, ctx.OPT_MATCH_SHAPE: 1},
    "fp4 4x64": {ctx.OPT_MATCH_FORM: 1, ctx.OPT_MATCH_SHAPE: 2},
    "fp4, trees behind matcher": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, ctx.OPT_TREE_FORK: 1},
    "fp4, trees behind sets": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, ctx.OPT_TREE_FORK: 2},
    "fp4, trees behind solve": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, ctx.OPT_TREE_FORK: 3},
    "fp4, trees behind screen": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, ctx.OPT_TREE_FORK: 4},
    "fp4 4x64, trees behind matcher": {ctx.OPT_MATCH_FORM: 1, ctx.OPT_MATCH_SHAPE: 2, ctx.OPT_TREE_FORK: 1},
    "int8 4x64, trees behind matcher": {ctx.OPT_MATCH_FORM: 2, ctx.OPT_MATCH_SHAPE: 2, ctx.OPT_TREE_FORK: 1},
    "fp4, no k-d trees": {ctx.OPT_MATCH_FORM: 0, ctx.OPT_MATCH_SHAPE: 0, "nodes": False},
    "int8 4x64, no k-d trees": {ctx.OPT_MATCH_FORM: 2, ctx.OPT_MATCH_SHAPE: 2, "nodes": False},
}
if os.environ.get("AB_ONLY_DEFAULT"):   # process-level knobs (environment variables) are compared across processes
    CONFIGS = {k: v for k, v in CONFIGS.items() if k in ("fp4 (default)", "fp4, no k-d trees")}
out = None
times = {k: [] for k in CONFIGS}
for rnd in range(rounds + 1):
    for name, opts in CONFIGS.items():
        ctx.set_option(ctx.OPT_TREE_FORK, -1 if name.startswith("fp4 (default)") else 0)
        for o, v in opts.items():
            if o != "nodes":
                ctx.set_option(o, v)
        if out is not None:
            if "nodes_kept" not in out:
                out["nodes_kept"] = out["nodes"]
            out["nodes"] = out["nodes_kept"] if opts.get("nodes", True) else None   # NULL d_nodes: no trees are built
        for _ in range(3):
            out = ctx.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=out)
        ctx.synchronize()
        t0 = time.perf_counter()
        n = 20 if wl != "C5" else 6
        for _ in range(n):
            out = ctx.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=out)
        ctx.synchronize()
        if rnd:   # the first round warms everything up
            times[name].append((time.perf_counter() - t0) / n * 1e3)
for name, t in times.items():
    t = np.array(t)
    print("%s %-22s %.4f ms  (min %.4f, max %.4f, %d blocks)" % (wl, name, t.mean(), t.min(), t.max(), len(t)))

#!/usr/bin/env python3
"""Opt-in MFMA / normal-matrix 8-point solver (VSLAM_OPT_RANSAC_SOLVER 1) against the exact solver on the C5 shape
(1920x1080, 4000 keypoints, 8192 hypotheses): how often does the pair's result change?  Prints one JSON object."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

w, h, K, H = 1920, 1080, 4000, 8192
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = 0x5EED0005
ctx = Context(0)
dev = torch.device("cuda", 0)
bgr = synth.frames_torch(seed, P, w, h, dev)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(seed, 0, P).view(np.int32)).to(dev)
res = {}
for name, opt in (("exact", 0), ("gram", 1)):
    ctx.set_option(ctx.OPT_RANSAC_SOLVER, opt)
    out = ctx.frontend_pairs(bgr, P, K, ca, sa, pat, seeds, H, 10.0)
    ctx.synchronize()
    res[name] = {k: v.cpu().numpy().copy() for k, v in out.items()}
ctx.set_option(ctx.OPT_RANSAC_SOLVER, 0)
jac, same, dcount = [], 0, []
for p in range(P):
    ne, ng = int(res["exact"]["best"][p, 3]), int(res["gram"]["best"][p, 3])
    a = {tuple(r) for r in res["exact"]["matches"][p, :ne]}
    b = {tuple(r) for r in res["gram"]["matches"][p, :ng]}
    jac.append(len(a & b) / max(1, len(a | b)))
    same += a == b
    dcount.append(ng - ne)
print(json.dumps({"workload": f"C5 shape, {P} pairs", "pairs_with_identical_inlier_matches": same,
                  "mean_jaccard_of_inlier_match_sets": float(np.mean(jac)), "min_jaccard": float(np.min(jac)),
                  "mean_inlier_count_exact": float(res["exact"]["best"][:, 3].mean()),
                  "mean_inlier_count_gram": float(res["gram"]["best"][:, 3].mean()),
                  "max_abs_inlier_count_difference": int(np.abs(dcount).max())}))

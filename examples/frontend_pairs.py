#!/usr/bin/env python3
"""Minimal use of the batched front-end from Python: two synthetic frame pairs in, fundamental matrices and inlier
matches out.  (The ctypes wrapper in vslam_amd/capi.py is a thin mirror of include/vslam_amd.h.)

    python examples/frontend_pairs.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402


def main():
    pairs, width, height = 2, 640, 480
    ctx = Context(0)                                            # raises if there is no MI355X / no built library
    bgr = torch.from_numpy(synth.frames_numpy(7, pairs, width, height)).cuda()   # frames [0, P) last, [P, 2P) current
    pattern = torch.from_numpy(synth.brief_pattern()).cuda()    # 256 x 4 int8 rBRIEF test pairs: ORB's learned table (None would mean the same)
    cos_a, sin_a = synth.keypoint_rotation()                    # cv::KeyPoint's default angle, -1 degree
    seeds = torch.from_numpy(shard.pair_seeds(1234, 0, pairs).view(np.int32)).cuda()
    out = ctx.frontend_pairs(bgr, pairs, 500, cos_a, sin_a, pattern, seeds, hyp=512, threshold=10.0)
    ctx.synchronize()
    for p in range(pairs):
        winner, inliers, _, n = out["best"][p].tolist()
        print(f"pair {p}: {int(out['n'][p])} / {int(out['n'][pairs + p])} keypoints, hypothesis {winner} with {inliers} inliers, "
              f"{n} inlier matches")
        print("  F =", np.array2string(out["F"][p].cpu().numpy().reshape(3, 3), precision=6))
        print("  first matches (keypoint index in last, in current):", out["matches"][p, :4].cpu().numpy().tolist())


if __name__ == "__main__":
    main()

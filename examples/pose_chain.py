#!/usr/bin/env python3
"""The per-pair chain of the reference's main loop, resident on the device (src/vslam.cpp:60-88,120-161,186-251): batches of frame
pairs as tickets of a pipeline, each ticket = extract x2 + match + RANSAC + extract_Rt + triangulate + reprojection filter
(vslam_pipeline_submit_pairs_pose); then the map-association block on a batch's own triangulated points
(vslam_associate_map_points).  Only poses, counts and the few numbers printed here leave the device.

    python examples/pose_chain.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import capi, shard, synth  # noqa: E402


def main():
    pairs, width, height, max_corners, hyp = 4, 640, 480, 1000, 512
    dev = torch.device("cuda", 0)
    pipe = capi.Pipeline(0, 2)                                   # two batches in flight
    cos_a, sin_a = synth.keypoint_rotation()
    K = np.array([[525.0, 0, width // 2], [0, 525.0, height // 2], [0, 0, 1]], np.float32)   # src/vslam.cpp:32
    batches = [torch.from_numpy(synth.frames_numpy(20 + i, pairs, width, height)).to(dev) for i in range(3)]
    seeds = [torch.from_numpy(shard.pair_seeds(99, i * pairs, (i + 1) * pairs).view(np.int32)).to(dev) for i in range(3)]
    outs = [capi.Pipeline.alloc_pose_outputs(torch, 2 * pairs, pairs, max_corners, dev) for _ in range(2)]
    tickets = []
    for i in range(3):
        if i >= 2:
            report(pipe, tickets[i - 2], outs[(i - 2) % 2], i - 2, pairs)
        tickets.append(pipe.submit_pairs_pose(batches[i], pairs, max_corners, cos_a, sin_a, None, seeds[i], hyp, 10.0, K, outs[i % 2]))
    for i in (1, 2):
        report(pipe, tickets[i], outs[i % 2], i, pairs)

    # the association block on the last batch: its triangulated points as the map (one observation each: the matched keypoint of
    # the first frame), looked up in the second frame's k-d tree, radius 2, Hamming threshold 64 (src/vslam.cpp:129-161)
    o = outs[0]                                                   # batch 2 ran on slot 0
    t, ctx = pipe.acquire()
    n_map = o["best"][:, 3].contiguous()
    offs = torch.arange(max_corners + 1, dtype=torch.int32, device=dev).repeat(pairs, 1).contiguous()
    idx1 = o["matches"][:, :, 0].long().clamp(0, max_corners - 1)
    obs = torch.gather(o["desc"][:pairs], 1, idx1[:, :, None].expand(pairs, max_corners, 32)).contiguous()
    ids = torch.full((pairs, max_corners), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    claim = ctx.associate(o["points4d"], n_map, o["c2"], width, height, o["nodes"][pairs:].contiguous(), o["xy"][pairs:].contiguous(),
                          o["desc"][pairs:].contiguous(), o["n"][pairs:].contiguous(), offs, obs, ids)
    pipe.commit(t)
    pipe.wait(t)
    for p in range(pairs):
        print(f"association, pair {p}: {int(n_map[p])} map points, {int((claim[p] >= 0).sum())} found their keypoint again")
    pipe.close()


def report(pipe, ticket, out, batch, pairs):
    pipe.wait(ticket)
    for p in range(pairs):
        t = out["t"][p].cpu().numpy()
        print(f"batch {batch} pair {p}: {int(out['best'][p, 3])} inlier matches, {int(out['n_inliers'][p])} pass the reprojection filter, "
              f"t = [{t[0]:+.3f} {t[1]:+.3f} {t[2]:+.3f}]")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Sizing aid (a simulation in float64, not a parity check): how many (hypothesis, match) evaluations of ransac_count_kernel
would remain if a hypothesis that can at best TIE the maximum count were also dropped as soon as an upper bound of its
residual sum -- what it has accumulated + threshold x (matches not yet looked at: they would all have to be inliers for the
tie) -- lies below the winner's sum?  (The accept rule takes the larger sum among equal counts, src/RansacFilter.cpp:59.)
usage: python tools/sumbound_sim.py [pairs] [easy|hard]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

w, h, K, H = 1280, 720, 2000, 4096
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "easy"
thr = 10.0
seed = 0x5EED0002
ctx = Context(0)
dev = torch.device("cuda", 0)
bgr = (synth.frames_torch if kind == "easy" else synth.frames_torch_hard)(seed, P, w, h, dev)
ca, sa = synth.keypoint_rotation()
ex = ctx.extract_features(bgr, K, ca, sa, None)
xy, desc, n = ex["xy"], ex["desc"], ex["n"]
pairs, m = ctx.match_knn2_ratio(desc[:P].contiguous(), n[:P].contiguous(), desc[P:].contiguous(), n[P:].contiguous())
seeds = torch.from_numpy(shard.pair_seeds(seed, 0, P).view(np.int32)).to(dev)
sets = ctx.ransac_sets(seeds, m, H)
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)
out = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m, sets, thr)
ctx.synchronize()
F = out["hypF"].double()
mm = m.cpu().numpy()
tot_now = tot_new = tot_all = 0
for p in range(P):
    M = int(mm[p])
    pr = pairs[p, :M].long()
    a = xy[p][pr[:, 0]].double(); c = xy[P + p][pr[:, 1]].double()
    f = F[p]
    x1, y1, x2, y2 = a[:, 0], a[:, 1], c[:, 0], c[:, 1]
    a0 = f[:, 0:1] * x1 + f[:, 1:2] * y1 + f[:, 2:3]
    a1 = f[:, 3:4] * x1 + f[:, 4:5] * y1 + f[:, 5:6]
    a2 = f[:, 6:7] * x1 + f[:, 7:8] * y1 + f[:, 8:9]
    nn = x2 * a0 + y2 * a1 + a2
    t0 = f[:, 0:1] * x2 + f[:, 3:4] * y2 + f[:, 6:7]
    t1 = f[:, 1:2] * x2 + f[:, 4:5] * y2 + f[:, 7:8]
    e = (nn * nn / (a0 * a0) + a1 * a1 + t0 * t0 + t1 * t1).cpu().numpy()      # [H, M]
    e = np.nan_to_num(e, nan=np.inf)
    I = e <= thr
    counts = I.sum(1)
    cmax = counts.max()
    sums = e.sum(1)
    win_sum = sums[counts == cmax].max()
    pil = [(H * k) // 8 for k in range(8)]
    order = np.argsort(-(~I[pil]).sum(0), kind="stable")            # most-missed first, as ransac_rank does
    eo, Io = e[:, order], I[:, order]
    nsub = (M + 255) // 256
    ends = np.minimum(256 * (np.arange(nsub) + 1), M)
    allowed = M - cmax
    miss_cum = np.stack([(~Io[:, :t]).sum(1) for t in ends], 1)      # [H, nsub]
    sum_cum = np.stack([np.where(np.isfinite(eo[:, :t]), eo[:, :t], 0).sum(1) + np.where(np.isfinite(eo[:, :t]), 0, 1e300).sum(1) for t in ends], 1)
    # today's rule: dropped after the sub-block in which the misses exceed the allowance
    over = miss_cum > allowed
    stop_now = np.where(over.any(1), over.argmax(1), nsub - 1)
    # the floor as the kernels can know it: per eighth of the hypotheses, the one with the best screen score (inliers among
    # the 128 most-missed matches) and, among those, the largest sum over those matches; the floor is the largest total sum
    # among the eight that reach the maximum count
    pot = Io[:, :128].sum(1)
    ssum = np.where(np.isfinite(eo[:, :128]), eo[:, :128], 0).sum(1)
    cands = []
    per = (H + 7) // 8
    for w8 in range(8):
        sl = np.arange(w8 * per, min(H, (w8 + 1) * per))
        best = sl[pot[sl] == pot.max()]
        if len(best):
            cands.append(best[np.argmax(ssum[best])])
    cands = [c for c in cands if counts[c] == cmax]
    floor = max([sums[c] for c in cands]) if cands else -np.inf
    if len(sys.argv) > 3:
        floor = win_sum
    # new rule: also dropped when it can at best tie (misses == allowance) and its sum cannot reach the floor
    tie_lost = (miss_cum == allowed) & (sum_cum + thr * (M - ends)[None, :] < floor * (1 - 1e-6))
    either = over | tie_lost
    stop_new = np.where(either.any(1), either.argmax(1), nsub - 1)
    ev_now, ev_new = ends[stop_now].sum(), ends[stop_new].sum()
    tot_now += ev_now; tot_new += ev_new; tot_all += H * M
    tied = counts == cmax
    print("pair %2d  M %4d  max count %4d  at the max %4d  evaluated today %.3f  with the sum rule %.3f   floor / winner %.4f  tied sums: median / winner %.4f" %
          (p, M, cmax, int(tied.sum()), ev_now / (H * M), ev_new / (H * M), floor / win_sum, np.median(sums[tied]) / win_sum))
print("%s data, %d pairs: evaluated today %.3f of all (hypothesis, match) pairs, with the sum rule %.3f  (-%.0f %%)" %
      (kind, P, tot_now / tot_all, tot_new / tot_all, 100 * (1 - tot_new / tot_now)))

#!/usr/bin/env python3
"""How crowded is the top of the RANSAC score table on the bench workload?  Per pair: matches, best count,
hypotheses tied at the best count, hypotheses within 1 / 2 / 5 / 1% of it.  (Sizing aid for ransac_tiesum_kernel.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

w, h, K, H, P = 1280, 720, 2000, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = 0x5EED0001
ctx = Context(0)
dev = torch.device("cuda", 0)
bgr = synth.frames_torch(seed, P, w, h, dev)
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
ca, sa = synth.keypoint_rotation()
ex = ctx.extract_features(bgr, K, ca, sa, pat)
xy, desc, n = ex["xy"], ex["desc"], ex["n"]
pairs, m = ctx.match_knn2_ratio(desc[:P].contiguous(), n[:P].contiguous(), desc[P:].contiguous(), n[P:].contiguous())
seeds = torch.from_numpy(shard.pair_seeds(seed, 0, P).view(np.int32)).to(dev)
sets = ctx.ransac_sets(seeds, m, H)
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)     # every count, not only the maximal ones
out = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m, sets, 10.0)
ctx.synchronize()
cnt = out["hyp_count"].cpu().numpy()
mm = m.cpu().numpy()
print("pair matches best tied within1 within2 within5 within1pct")
rows = []
for p in range(P):
    c = cnt[p]
    b = c.max()
    rows.append((mm[p], b, (c == b).sum(), (c >= b - 1).sum(), (c >= b - 2).sum(), (c >= b - 5).sum(), (c >= b * 0.99).sum()))
    print(p, *rows[-1])
r = np.array(rows)
print("mean", r.mean(0).round(1), "max tied", r[:, 2].max())

# how many of the tied hypotheses survive the bound-based pruning (hyp_sum is -inf for the pruned ones)
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)
hs = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m, sets, 10.0)["hyp_sum"].cpu().numpy()
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)
kept = [(int(((cnt[p] == cnt[p].max()) & ~np.isneginf(hs[p])).sum())) for p in range(P)]
print("candidates kept per pair:", kept, "mean", float(np.mean(kept)), "max", max(kept))

# --- would early bail-out pay?  Score every hypothesis on the first 512 matches only, then see how many could be
# dropped because even with all remaining matches as inliers they cannot reach the pair's best full count.
m512 = torch.clamp(m, max=512)
out512 = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m512, sets, 10.0)
ctx.synchronize()
c512 = out512["hyp_count"].cpu().numpy()
drop, work = [], []
for p in range(P):
    full, part, M = cnt[p], c512[p], int(mm[p])
    best = full.max()
    can_reach = part + (M - min(M, 512)) >= best
    drop.append(1.0 - can_reach.mean())
    work.append((min(M, 512) + can_reach.mean() * (M - min(M, 512))) / M)
print("hypotheses droppable after 512 matches: mean %.3f (min %.3f max %.3f); evaluations left: mean %.3f of all" %
      (np.mean(drop), np.min(drop), np.max(drop), np.mean(work)))
q = np.array([np.quantile(cnt[p] / mm[p], [0.1, 0.25, 0.5, 0.75]) for p in range(P)])
print("inlier ratio quantiles over hypotheses (10/25/50/75 %), mean over pairs:", q.mean(0).round(3))

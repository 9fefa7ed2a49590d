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
hs = out["hyp_sum"].cpu().numpy()
kept = [(int(((cnt[p] == cnt[p].max()) & ~np.isneginf(hs[p])).sum())) for p in range(P)]
print("candidates kept per pair:", kept, "mean", float(np.mean(kept)), "max", max(kept))

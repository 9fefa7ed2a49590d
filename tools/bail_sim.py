#!/usr/bin/env python3
"""Sizing aid for ransac_count_kernel's bail-out: how many (hypothesis, match) evaluations remain under different
visiting schemes on the bench workload?  Inlier tables come from a float64 torch evaluation of the as-written residual
(not bit-exact; a simulation, not a parity check).

  A  today's scheme: matches in list order, the pair's 256-match sub-blocks dealt over 2 waves, each wave bails on the
     outliers of its own part only, the second visitor skips what the first abandoned
  B  matches ranked by how many of the 8 pilot hypotheses miss them (most-missed first), sub-blocks dealt round-robin
     over the waves, all waves of the workgroup decide together after every round of sub-blocks
  C  same ranking, one wave walks all sub-blocks of a hypothesis
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

w, h, K, H, P = 1280, 720, 2000, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 8
thr = 10.0
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
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)
out = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m, sets, thr)
ctx.synchronize()
F = out["hypF"].double()          # [P, H, 9]
cnt_ref = out["hyp_count"].cpu().numpy()
mm = m.cpu().numpy()


def inlier_table(p):
    M = int(mm[p])
    pr = pairs[p, :M].long()
    a = xy[p][pr[:, 0]].double()
    c = xy[P + p][pr[:, 1]].double()
    f = F[p]                       # [H, 9]
    x1, y1, x2, y2 = a[:, 0], a[:, 1], c[:, 0], c[:, 1]
    a0 = f[:, 0:1] * x1 + f[:, 1:2] * y1 + f[:, 2:3]
    a1 = f[:, 3:4] * x1 + f[:, 4:5] * y1 + f[:, 5:6]
    a2 = f[:, 6:7] * x1 + f[:, 7:8] * y1 + f[:, 8:9]
    nn = x2 * a0 + y2 * a1 + a2
    t0 = f[:, 0:1] * x2 + f[:, 3:4] * y2 + f[:, 6:7]
    t1 = f[:, 1:2] * x2 + f[:, 4:5] * y2 + f[:, 7:8]
    e = nn * nn / (a0 * a0) + a1 * a1 + t0 * t0 + t1 * t1
    return (e <= thr).cpu().numpy()


def rounds_until(miss_cum, allowed):
    """miss_cum [H, R]: cumulative misses after each round; -> index of the first round whose total exceeds `allowed`
    (R - 1 if none: the hypothesis is evaluated in full)."""
    over = miss_cum > allowed
    first = np.where(over.any(1), over.argmax(1), miss_cum.shape[1] - 1)
    return first


res = []
for p in range(P):
    I = inlier_table(p)            # [H, M]
    M = I.shape[1]
    counts = I.sum(1)
    best = counts.max()
    if abs(int(best) - int(cnt_ref[p].max())) > 2:
        print("warning: simulated best", best, "vs kernel", cnt_ref[p].max())
    pil = [(H * k) // 8 for k in range(8)]
    bound = counts[pil].max()
    allowed = M - bound
    miss = ~I
    nsub = (M + 255) // 256
    sizes = np.array([min(256, M - 256 * s) for s in range(nsub)])

    def sub_miss(order):
        mo = miss[:, order]
        return np.stack([mo[:, 256 * s:256 * s + 256].sum(1) for s in range(nsub)], 1)   # [H, nsub]

    # A: list order, two waves, contiguous halves
    sm = sub_miss(np.arange(M))
    nw = 2
    lo = [(nsub * wv) // nw for wv in range(nw + 1)]
    evalA = np.zeros(H)
    dropped_by = np.zeros((H, nw), bool)
    stop = []
    for wv in range(nw):
        part = sm[:, lo[wv]:lo[wv + 1]]
        cum = np.cumsum(part, 1)
        r = rounds_until(cum, allowed)
        sz = np.cumsum(sizes[lo[wv]:lo[wv + 1]])
        stop.append(sz[r])
        dropped_by[:, wv] = (cum > allowed).any(1)
    ring = np.arange(H) % 64
    first = np.where(ring < 32, 0, 1)
    for wv in range(nw):
        is_first = first == wv
        other_dropped = dropped_by[:, 1 - wv]
        evalA += np.where(is_first, stop[wv], np.where(other_dropped, 0, stop[wv]))
    fracA = evalA.sum() / (H * M)

    # ranking by pilot misses (stable, most-missed first)
    pm = miss[pil].sum(0)
    order = np.argsort(-pm, kind="stable")
    smr = sub_miss(order)
    # B: round r = sub-blocks r*nw .. r*nw+nw-1, decision after each round with everybody's misses
    nr = (nsub + nw - 1) // nw
    cumB = np.stack([smr[:, :min(nsub, (r + 1) * nw)].sum(1) for r in range(nr)], 1)
    szB = np.array([sizes[:min(nsub, (r + 1) * nw)].sum() for r in range(nr)])
    fracB = szB[rounds_until(cumB, allowed)].sum() / (H * M)
    # C: one sub-block at a time
    cumC = np.cumsum(smr, 1)
    fracC = np.cumsum(sizes)[rounds_until(cumC, allowed)].sum() / (H * M)
    # C with the true maximum as the bound (what a perfect bound would give)
    fracC_ideal = np.cumsum(sizes)[rounds_until(cumC, M - best)].sum() / (H * M)
    # B without the ranking (list order), to separate the two effects
    cumB0 = np.stack([sm[:, :min(nsub, (r + 1) * nw)].sum(1) for r in range(nr)], 1)
    fracB0 = szB[rounds_until(cumB0, allowed)].sum() / (H * M)
    # a better bound: screen every hypothesis on the first sub-block of the ranked list, count the T most promising
    # ones in full (fewest misses there, first index among equals), take the best of those and the pilots
    extra = []
    for T in (8, 16, 64):
        top = np.argsort(smr[:, 0], kind="stable")[:T]
        b2 = max(bound, counts[top].max())
        fB = szB[rounds_until(cumB, M - b2)].sum() / (H * M)
        fC = np.cumsum(sizes)[rounds_until(cumC, M - b2)].sum() / (H * M)
        extra.append((T, int(best - b2), round(float(fB), 3), round(float(fC), 3)))
    mo = miss[:, order]
    for N in (64, 128):
        score = (~mo[:, :N]).sum(1)           # inliers among the N most-missed matches
        smax = score.max()
        cand = np.nonzero(score == smax)[0][:8]
        if len(cand) < 8:
            cand = np.concatenate([cand, np.nonzero(score == smax - 1)[0][:8 - len(cand)]])
        b3 = max(bound, counts[cand].max())
        fC = np.cumsum(sizes)[rounds_until(cumC, M - b3)].sum() / (H * M)
        extra.append(("screen%d" % N, int(best - b3), int((score == smax).sum()), round(float(fC), 3)))
    fracB_ideal = szB[rounds_until(cumB, M - best)].sum() / (H * M)
    print("   screened bound (T, best - bound, B, C):", extra, " B(ideal) %.3f" % fracB_ideal,
          " hyps at best-1: %.3f best-2: %.3f" % ((counts == best - 1).mean(), (counts == best - 2).mean()))
    tied = (counts == best).mean()
    res.append((M, best, bound, tied, fracA, fracB0, fracB, fracC, fracC_ideal))
    print(p, "M %d best %d pilot-bound %d tied %.3f | A %.3f  B(no rank) %.3f  B %.3f  C %.3f  C(ideal bound) %.3f" % res[-1])
r = np.array(res)
print("mean:", r.mean(0).round(3))

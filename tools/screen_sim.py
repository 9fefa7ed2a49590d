#!/usr/bin/env python3
"""could an APPROXIMATE 8-point solve (the Gram / MFMA one -- DESIGN.md 5, HISTORY.md notes 5.3 -- or any other) serve as
a SCREEN in front of the exact Jacobi replay?  Idea: solve every hypothesis approximately (F~), bound every match's
residual from below over all F within delta of F~, and run the exact solve only for hypotheses whose certified upper bound
of the inlier count still reaches the best verified count.  That is sound only if |F_exact - F~| <= delta is KNOWN for the
hypothesis, F_exact being what OpenCV's float Jacobi produces -- rounding included.  This tool measures, on C3 data:

  d(h)      the actual distance between F_exact (the device's, bit-identical to the oracle's) and the best F~ any
            approximate solver could give: the float64 null vector of the column-scaled system, rank 2 enforced in float64
            -- in the column-scaled metric the rounding analysis of a row-rotation Jacobi lives in (an error in column k is
            relative to column k's size), and as a plain relative Frobenius distance;
  b(h)      the a-priori first-order bound one could certify without running the exact solve: c eps_f kappa_s(h), with
            kappa_s = sigma_1 / sigma_8 of the column-scaled design matrix, eps_f = 2^-24 and c = 3 x 94 rotations;
  for delta = 1e-6 .. 1e-1: the share of hypotheses with d(h) <= delta (the screen would be VALID), with b(h) <= delta (it would
            be CERTIFIED), and the share that are valid / certified AND excluded (count upper bound below the pair's best count).

    python tools/screen_sim.py [pairs] [easy|hard]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, shard, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "hard"
w, h, K, H = 1280, 720, 2000, 4096
thr, seed = 10.0, 0x5EED0001
ctx = Context(0)
dev = torch.device("cuda", 0)
bgr = (synth.frames_torch_hard if kind == "hard" else synth.frames_torch)(seed, P, w, h, dev)
ca, sa = synth.keypoint_rotation()
ex = ctx.extract_features(bgr, K, ca, sa, None)
xy, desc, n = ex["xy"], ex["desc"], ex["n"]
pairs, m = ctx.match_knn2_ratio(desc[:P].contiguous(), n[:P].contiguous(), desc[P:].contiguous(), n[P:].contiguous())
seeds = torch.from_numpy(shard.pair_seeds(seed, 0, P).view(np.int32)).to(dev)
sets = ctx.ransac_sets(seeds, m, H)
ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)
out = ctx.ransac_fundamental(xy[:P].contiguous(), xy[P:].contiguous(), pairs, m, sets, thr)
ctx.synchronize()
Fx = out["hypF"].double()                    # [P, H, 9]  exact (OpenCV float Jacobi replayed)
cnt = out["hyp_count"]                       # [P, H]
mm = m.cpu().numpy()
DELTAS = [1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1]
EPS_F, C_ROT = 2.0 ** -24, 3 * 94
acc = {d: dict(valid=0, cert=0, valid_excl=0, cert_excl=0, any_excl=0) for d in DELTAS}
dist_scaled, dist_plain, bounds, total = [], [], [], 0
for p in range(P):
    M = int(mm[p])
    pr = pairs[p, :M].long()
    a = xy[p][pr[:, 0]]                      # f32 [M, 2]
    c = xy[P + p][pr[:, 1]]
    S = sets[p].long()                       # [H, 8]
    u1, v1, u2, v2 = a[S][..., 0], a[S][..., 1], c[S][..., 0], c[S][..., 1]     # [H, 8] f32
    one = torch.ones_like(u1)
    A = torch.stack([u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, one], -1).double()   # f32 products, as the reference forms them
    D = A.norm(dim=1)                        # [H, 9] column norms
    D = torch.where(D > 0, D, torch.ones_like(D))
    As = A / D[:, None, :]
    U_, Sg, Vh = torch.linalg.svd(As, full_matrices=True)
    g = Vh[:, 8, :]                          # null vector of the scaled system
    kappa = Sg[:, 0] / Sg[:, 7].clamp_min(1e-300)
    f0 = g / D
    f0 = f0 / f0.norm(dim=1, keepdim=True)
    # rank 2 in float64 (the reference's second SVD works on F0's columns; the projection is the same matrix)
    U3, S3, V3 = torch.linalg.svd(f0.reshape(-1, 3, 3))
    S3[:, 2] = 0
    Fa = (U3 * S3[:, None, :]) @ V3
    Fa = Fa.reshape(-1, 9)
    fx = Fx[p]
    # column scale of the metric for F itself: entry k of F multiplies monomial k, whose size is D[k] / sqrt(8)
    Wk = D / np.sqrt(8.0)
    dplus = ((fx - Fa) * Wk).norm(dim=1)
    dminus = ((fx + Fa) * Wk).norm(dim=1)
    sgn = torch.where(dplus <= dminus, 1.0, -1.0)
    ref = (Fa * Wk).norm(dim=1).clamp_min(1e-300)
    ds = torch.minimum(dplus, dminus) / ref
    dp = torch.minimum((fx - Fa).norm(dim=1), (fx + Fa).norm(dim=1)) / Fa.norm(dim=1).clamp_min(1e-300)
    bnd = C_ROT * EPS_F * kappa
    dist_scaled.append(ds.cpu().numpy()); dist_plain.append(dp.cpu().numpy()); bounds.append(bnd.cpu().numpy())
    best = int(cnt[p].max())
    # certified count upper bound of every hypothesis under |W (F - F~)| <= delta |W F~|
    x1, y1, x2, y2 = a[:, 0].double(), a[:, 1].double(), c[:, 0].double(), c[:, 1].double()
    Fs = Fa * sgn[:, None]
    o = torch.ones_like(x1)

    def form(cols, mon):                     # L(F) = sum_k F[cols[k]] * mon[k]: value [H, M] and the norm of mon / W over its columns
        val = sum(Fs[:, ck:ck + 1] * mk[None, :] for ck, mk in zip(cols, mon))
        wn = torch.sqrt(sum((mk[None, :] / Wk[:, ck:ck + 1]) ** 2 for ck, mk in zip(cols, mon)))
        return val, wn
    a0, wa0 = form((0, 1, 2), (x1, y1, o))
    a1, wa1 = form((3, 4, 5), (x1, y1, o))
    t0, wt0 = form((0, 3, 6), (x2, y2, o))
    t1, wt1 = form((1, 4, 7), (x2, y2, o))
    nn, wnn = form(tuple(range(9)), (x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, o))
    scale = ref[:, None]
    for d in DELTAS:
        r = d * scale

        def lo(v, wv):
            return (v.abs() - r * wv).clamp_min(0)
        e_lb = lo(nn, wnn) ** 2 / (a0.abs() + r * wa0).clamp_min(1e-300) ** 2 + lo(a1, wa1) ** 2 + lo(t0, wt0) ** 2 + lo(t1, wt1) ** 2
        ub = (e_lb <= thr).sum(1)
        excl = ub < best
        valid = ds <= d
        cert = bnd <= d
        acc[d]["valid"] += int(valid.sum()); acc[d]["cert"] += int(cert.sum())
        acc[d]["valid_excl"] += int((valid & excl).sum()); acc[d]["cert_excl"] += int((cert & excl).sum())
        acc[d]["any_excl"] += int(excl.sum())
    total += H
ds = np.concatenate(dist_scaled); dp = np.concatenate(dist_plain); bn = np.concatenate(bounds)
q = [0.01, 0.1, 0.5, 0.9, 0.99]
res = {"data": kind, "pairs": P, "hypotheses": total,
       "distance_exact_vs_float64_null_vector": {"scaled_metric_quantiles": dict(zip(map(str, q), np.quantile(ds, q).tolist())),
                                                "plain_relative_quantiles": dict(zip(map(str, q), np.quantile(dp, q).tolist()))},
       "a_priori_bound_quantiles": dict(zip(map(str, q), np.quantile(bn, q).tolist())),
       "actual_distance_within_the_bound": float((ds <= bn).mean()),
       "per_delta": {str(d): {k: v / total for k, v in acc[d].items()} for d in DELTAS}}
print(json.dumps(res, indent=1))
print(f"\n{kind} data, {total} hypotheses: share of hypotheses")
print("  delta    valid(d<=delta)  certified(b<=delta)  excluded if valid  excluded if certified  (excluded ignoring validity)")
for d in DELTAS:
    a_ = res["per_delta"][str(d)]
    print(f"  {d:7.0e}      {a_['valid']:6.3f}            {a_['cert']:6.3f}             {a_['valid_excl']:6.3f}              {a_['cert_excl']:6.3f}                 {a_['any_excl']:6.3f}")

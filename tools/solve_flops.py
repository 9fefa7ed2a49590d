#!/usr/bin/env python3
"""Count what compute_fundamental (src/RansacFilter.cpp:69-103) costs per hypothesis on the bench's data, and how the
Jacobi sweep counts are distributed over the lanes of a wave.  CPU only (the oracle counts its own sweeps / (i, j) visits /
rotations: vso_compute_fundamental_work); writes profiles/<tag>_solve_work.json, which bench.py's roofline reads for the
algorithmic flop count of ransac_solve_kernel (it replaces SURVEY 8(d)'s "about 3000 flop" guess).

Flop model (one FMA = 2, sqrt and division = 1 each; the algorithm as the reference runs it, except the V matrix of the
first SVD, which nothing reads -- the kernel does not form it either):
  A (8 x 9):            4 products per row                                            = 32
  SVD 1 (8 rows of 9):  start-up norms 8 x 9 FMA                                      = 144
                        per (i, j) visit: dot product 9 FMA + convergence test 4      = 22
                        per rotation: parameters 13 + 9 x (rotate 6 + two norm FMAs 4) = 103
                        closing: norms 8 x 19, normalise 8 x 10, the null-space row (9 signs, 2 Gram-Schmidt passes over 8
                        rows at 18 + 27 + 10, norm 19, scale 10)                       = 1150
  SVD 2 (3 x 3, with V): start-up 18; per visit 6 + 4 = 10; per rotation 13 + 3 x 10 + 3 x 6 = 61; closing 35
  F = U diag(D) Vt:     two 3 x 3 products                                            = 90
usage: python tools/solve_flops.py [tag] [pairs]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle_lib import Oracle
    from vslam_amd import synth
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    w, h, K, H = 1280, 720, 2000, 4096            # C3
    o = Oracle()
    fn = o.lib.vso_compute_fundamental_work
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    bgr = synth.frames_numpy(0x5EED0002, n_pairs, w, h)
    work = []
    for p in range(n_pairs):
        a = o.extract_features(bgr[p], K, ca, sa, pat)
        b = o.extract_features(bgr[n_pairs + p], K, ca, sa, pat)
        pairs, rc = o.match_knn2_ratio(a["desc"], b["desc"])
        assert rc == 0 and len(pairs) >= 8
        sets = o.ransac_sets(0x5EED0002 ^ p, len(pairs), H)
        p1 = np.ascontiguousarray(a["xy"][pairs[sets, 0]])      # [H][8][2]
        p2 = np.ascontiguousarray(b["xy"][pairs[sets, 1]])
        out = np.zeros((H, 6), np.int32)
        F = np.zeros(9, np.float32)
        for i in range(H):
            assert fn(p1[i].ctypes.data_as(C.POINTER(C.c_float)), p2[i].ctypes.data_as(C.POINTER(C.c_float)), 8,
                      F.ctypes.data_as(C.POINTER(C.c_float)), out[i].ctypes.data_as(C.POINTER(C.c_int32))) == 0
        work.append(out)
        print(f"pair {p}: {len(pairs)} matches, mean sweeps {out[:, 0].mean():.2f}", file=sys.stderr)
    wk = np.concatenate(work).astype(np.float64)                 # [pairs * H][6]
    s1, v1, r1, s2, v2, r2 = wk.mean(0)
    flops = 32 + 144 + 22 * v1 + 103 * r1 + 1150 + 18 + 10 * v2 + 61 * r2 + 35 + 90
    # as the kernel runs them: one lane per hypothesis, 64 consecutive hypotheses of a pair per wave; a wave sweeps until
    # its last lane has converged (the sweep that finds nothing to rotate included)
    sw = np.concatenate(work)[:, 0].reshape(-1, 64)
    wave_max = sw.max(1)
    hist = np.bincount(np.concatenate(work)[:, 0], minlength=12)
    # wave-sweeps that serve few lanes: sweep t of a wave is useful to the lanes with sweeps >= t
    served = []
    for t in range(1, int(wave_max.max()) + 1):
        alive = wave_max >= t
        served.append(((sw[alive] >= t).sum(1) / 64.0))
    served = np.concatenate(served)
    res = {
        "what": "compute_fundamental per hypothesis on C3-like data (oracle counters), tools/solve_flops.py",
        "pairs": n_pairs, "hypotheses": int(wk.shape[0]),
        "svd_8x9": {"sweeps": s1, "visits": v1, "rotations": r1},
        "svd_3x3": {"sweeps": s2, "visits": v2, "rotations": r2},
        "flop_per_hypothesis": flops,
        "flop_model": "32 + 144 + 22 visits1 + 103 rotations1 + 1150 + 18 + 10 visits2 + 61 rotations2 + 35 + 90 (FMA = 2)",
        "sweeps_histogram_per_lane": {str(i): int(c) for i, c in enumerate(hist) if c},
        "sweeps_per_lane_mean": float(sw.mean()), "sweeps_per_wave_mean": float(wave_max.mean()),
        "sweeps_histogram_per_wave": {str(i): int(c) for i, c in enumerate(np.bincount(wave_max)) if c},
        "lane_idle_fraction": float(1.0 - sw.sum() / (64.0 * wave_max.sum())),
        "wave_sweeps_serving_under_25pct_of_lanes": float((served < 0.25).mean()),
        "wave_sweeps_serving_under_50pct_of_lanes": float((served < 0.50).mean()),
    }
    out = os.path.join(ROOT, "profiles", f"{tag}_solve_work.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

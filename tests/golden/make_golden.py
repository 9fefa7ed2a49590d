#!/usr/bin/env python3
"""Generates tests/golden/frontend_v1.npz from the ORACLE (oracle/liboracle.so) on small seeded
inputs.  The reference cannot run in this container (every hot-path file needs OpenCV) and ships no
vectors of its own besides the k-d tree property test, so these vectors pin the oracle — they are
what the HIP kernels and any later oracle edit are held to, on every machine.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vslam_amd import synth  # noqa: E402


def main():
    o = Oracle()
    g = {}
    # ---- k-d tree: integer points with heavy ties, radius queries in visit order
    rng = np.random.default_rng(1)
    pts = np.rint(np.stack([rng.uniform(0, 60, 300), rng.uniform(0, 40, 300)], 1)).astype(np.float32)
    g["kd_pts"] = pts
    g["kd_nodes"] = o.kdtree_build_frame(pts)
    qs = (pts[rng.integers(0, 300, 40)] + rng.uniform(-2, 2, (40, 2))).astype(np.float32)
    g["kd_queries"] = qs
    hits = np.full((40, 16), -1, np.int32); cnt = np.zeros(40, np.int32)
    for i, q in enumerate(qs):
        h, c = o.kdtree_radius_frame(g["kd_nodes"], pts, q, 2.0, cap=16)
        hits[i, :len(h)] = h; cnt[i] = c
    g["kd_hits"], g["kd_counts"] = hits, cnt
    # ---- matching
    d1, d2, _ = synth.descriptors_pair(2, 64, 72)
    d2[9] = d2[30]
    g["m_d1"], g["m_d2"] = d1, d2
    g["m_knn"] = np.stack(o.match_knn2(d1, d2), 1)
    g["m_pairs"], _ = o.match_knn2_ratio(d1, d2)
    # ---- RANSAC
    g["r_sets_seed"] = np.array([0x5EED0001], np.uint32)
    g["r_sets"] = o.ransac_sets(0x5EED0001, 37, 16)
    p1, p2, _ = synth.two_view_points(3, 120, 640, 480, inlier_frac=0.7)
    pairs = np.stack([np.arange(100), np.arange(100)], 1).astype(np.int32)
    sets = o.ransac_sets(77, 100, 48)
    r = o.find_fundamental(p1, p2, pairs, sets, 10.0)
    g["r_p1"], g["r_p2"], g["r_pairs"], g["r_fsets"] = p1, p2, pairs, sets
    for k in ("hypF", "hyp_count", "hyp_sum", "F", "mask"):
        g["r_" + k] = r[k]
    g["r_best"] = np.array([r["winner"], r["count"], np.float32(r["sum"]).view(np.int32)], np.int32)
    # ---- extraction on one small frame pair + the whole pipeline
    w, h, maxc, H = 160, 128, 150, 64
    bgr = synth.frames_numpy(4, 1, w, h)
    pat = synth.brief_pattern(); ca, sa = synth.keypoint_rotation()
    g["e_bgr"], g["e_pattern"], g["e_rot"] = bgr, pat, np.array([ca, sa], np.float32)
    gray = o.bgr2gray(bgr[0])
    g["e_gray"], g["e_eig"], g["e_blur"] = gray, o.min_eigen(gray), o.gaussian7(gray)
    g["e_corners"] = o.good_features(gray, maxc)
    for f in range(2):
        e = o.extract_features(bgr[f], maxc, ca, sa, pat)
        g[f"e_xy{f}"], g[f"e_desc{f}"], g[f"e_nodes{f}"] = e["xy"], e["desc"], e["nodes"]
        g[f"e_counts{f}"] = np.array([e["n"], e["n_detected"]], np.int32)
    mf = o.match_features(g["e_xy0"], g["e_desc0"], g["e_xy1"], g["e_desc1"], 0x5EED0000, H, 10.0)
    assert mf["rc"] == 0
    g["p_matches"], g["p_F"], g["p_prelim"] = mf["matches"], mf["F"], np.array([mf["prelim"]], np.int32)
    out = os.path.join(HERE, "frontend_v1.npz")
    np.savez_compressed(out, **g)
    # ---- grid ORB/FAST extractor (src/Frame.cpp:16-51) on one small frame
    bgr = synth.frames_numpy(6, 1, 256, 192)[0]
    img, xy, desc, ao = o.extract_features_grid(bgr, 2, 2, pat)
    g2 = dict(g_bgr=bgr, g_pattern=pat, g_outlined=img, g_xy=xy, g_desc=desc, g_angle_octave=ao,
              g_fast20=o.fast9_16(o.bgr2gray(bgr), 20), g_resized=o.resize_linear_exact(o.bgr2gray(bgr), 213, 160))
    out2 = os.path.join(HERE, "frontend_v2_grid.npz")
    np.savez_compressed(out2, **g2)
    print(out2, os.path.getsize(out2), "bytes;", len(xy), "grid keypoints")
    print(out, os.path.getsize(out), "bytes;", {k: v.shape for k, v in g.items() if v.ndim} and len(g), "arrays")


if __name__ == "__main__":
    main()

"""The oracle reproduces the committed golden vectors (tests/golden/frontend_v1.npz), bit for bit,
on this machine: guards against oracle edits and compiler / libm / libstdc++ drift."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_v1.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_kdtree(oracle):
    assert np.array_equal(oracle.kdtree_build_frame(G["kd_pts"]), G["kd_nodes"])
    for i, q in enumerate(G["kd_queries"]):
        h, c = oracle.kdtree_radius_frame(G["kd_nodes"], G["kd_pts"], q, 2.0, cap=16)
        assert c == G["kd_counts"][i] and np.array_equal(h, G["kd_hits"][i, :c])


def test_match(oracle):
    assert np.array_equal(np.stack(oracle.match_knn2(G["m_d1"], G["m_d2"]), 1), G["m_knn"])
    p, rc = oracle.match_knn2_ratio(G["m_d1"], G["m_d2"])
    assert rc == 0 and np.array_equal(p, G["m_pairs"])


def test_ransac(oracle):
    assert np.array_equal(oracle.ransac_sets(int(G["r_sets_seed"][0]), 37, 16), G["r_sets"])
    r = oracle.find_fundamental(G["r_p1"], G["r_p2"], G["r_pairs"], G["r_fsets"], 10.0)
    assert np.array_equal(bits(r["hypF"]), bits(G["r_hypF"]))
    assert np.array_equal(r["hyp_count"], G["r_hyp_count"]) and np.array_equal(bits(r["hyp_sum"]), bits(G["r_hyp_sum"]))
    assert np.array_equal(bits(r["F"]), bits(G["r_F"])) and np.array_equal(r["mask"], G["r_mask"])
    assert [r["winner"], r["count"]] == G["r_best"][:2].tolist()


def test_extract_and_pipeline(oracle):
    bgr, pat = G["e_bgr"], G["e_pattern"]
    ca, sa = map(float, G["e_rot"])
    gray = oracle.bgr2gray(bgr[0])
    assert np.array_equal(gray, G["e_gray"])
    assert np.array_equal(bits(oracle.min_eigen(gray)), bits(G["e_eig"]))
    assert np.array_equal(oracle.gaussian7(gray), G["e_blur"])
    assert np.array_equal(oracle.good_features(gray, 150), G["e_corners"])
    ex = []
    for f in range(2):
        e = oracle.extract_features(bgr[f], 150, ca, sa, pat)
        assert np.array_equal(e["xy"], G[f"e_xy{f}"]) and np.array_equal(e["desc"], G[f"e_desc{f}"])
        assert np.array_equal(e["nodes"], G[f"e_nodes{f}"]) and [e["n"], e["n_detected"]] == G[f"e_counts{f}"].tolist()
        ex.append(e)
    mf = oracle.match_features(ex[0]["xy"], ex[0]["desc"], ex[1]["xy"], ex[1]["desc"], 0x5EED0000, 64, 10.0)
    assert mf["prelim"] == int(G["p_prelim"][0]) and np.array_equal(mf["matches"], G["p_matches"])
    assert np.array_equal(bits(mf["F"]), bits(G["p_F"]))


def test_grid_orb_golden(oracle):
    G2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_v2_grid.npz"))
    img, xy, desc, ao = oracle.extract_features_grid(G2["g_bgr"], 2, 2, G2["g_pattern"])
    assert np.array_equal(img, G2["g_outlined"])
    assert np.array_equal(bits(xy), bits(G2["g_xy"])) and np.array_equal(desc, G2["g_desc"])
    assert np.array_equal(bits(ao), bits(G2["g_angle_octave"]))
    gray = oracle.bgr2gray(G2["g_bgr"])
    assert np.array_equal(oracle.fast9_16(gray, 20), G2["g_fast20"])
    assert np.array_equal(oracle.resize_linear_exact(gray, 213, 160), G2["g_resized"])

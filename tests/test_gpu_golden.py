"""HIP path vs the committed golden vectors (no oracle involved)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_v1.npz"))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_kdtree_golden(ctx):
    n = torch.tensor([300], dtype=torch.int32).cuda()
    nodes = ctx.kdtree_build(t(G["kd_pts"][None]), n)
    assert np.array_equal(nodes[0].cpu().numpy(), G["kd_nodes"])
    hits, cnt = ctx.kdtree_radius(nodes, t(G["kd_pts"][None]), n, t(G["kd_queries"][None]),
                                  torch.tensor([40], dtype=torch.int32).cuda(), 2.0, hit_cap=16)
    assert np.array_equal(cnt[0].cpu().numpy(), G["kd_counts"])
    assert np.array_equal(hits[0].cpu().numpy(), G["kd_hits"])


def test_match_golden(ctx):
    d1 = np.zeros((1, 72, 32), np.uint8); d1[0, :64] = G["m_d1"]
    pairs, m, knn = ctx.match_knn2_ratio(t(d1), torch.tensor([64], dtype=torch.int32).cuda(), t(G["m_d2"][None]),
                                         torch.tensor([72], dtype=torch.int32).cuda(), want_knn=True)
    assert np.array_equal(knn[0, :64].cpu().numpy(), G["m_knn"])
    k = int(m[0])
    assert k == len(G["m_pairs"]) and np.array_equal(pairs[0, :k].cpu().numpy(), G["m_pairs"])


def test_ransac_golden(ctx):
    sets = ctx.ransac_sets(t(G["r_sets_seed"].view(np.int32)), torch.tensor([37], dtype=torch.int32).cuda(), 16)
    assert np.array_equal(sets[0].cpu().numpy(), G["r_sets"])
    pairs = np.zeros((1, 120, 2), np.int32); pairs[0, :100] = G["r_pairs"]
    for all_sums in (False, True):   # default scoring path (sums where the accept rule can consult them), then every sum
        ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, all_sums)
        out = ctx.ransac_fundamental(t(G["r_p1"][None]), t(G["r_p2"][None]), t(pairs), torch.tensor([100], dtype=torch.int32).cuda(),
                                     t(G["r_fsets"][None]), 10.0)
        ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)
        out = {k: v.cpu().numpy() for k, v in out.items()}
        assert np.array_equal(bits(out["hypF"][0]), bits(G["r_hypF"]))
        from test_gpu_ransac import check_counts, check_sums
        check_counts(out["hyp_count"][0], G["r_hyp_count"], "all" if all_sums else "ties", None, G["r_hyp_sum"])
        check_sums(out["hyp_sum"][0], G["r_hyp_count"], G["r_hyp_sum"], "all" if all_sums else "ties", None, out["hyp_count"][0])
        assert np.array_equal(bits(out["F"][0]), bits(G["r_F"])) and np.array_equal(out["mask"][0, :100], G["r_mask"])
        assert out["best"][0, :3].tolist() == G["r_best"].tolist()


def test_extract_and_pipeline_golden(ctx):
    bgr, pat = G["e_bgr"], G["e_pattern"]
    ca, sa = map(float, G["e_rot"])
    gray = ctx.bgr2gray(t(bgr))
    assert np.array_equal(gray[0].cpu().numpy(), G["e_gray"])
    assert np.array_equal(bits(ctx.min_eigen(gray)[0].cpu().numpy()), bits(G["e_eig"]))
    assert np.array_equal(ctx.gaussian7(gray)[0].cpu().numpy(), G["e_blur"])
    xy, n = ctx.good_features(gray, 150)
    assert np.array_equal(xy[0, :int(n[0])].cpu().numpy(), G["e_corners"])
    seeds = np.array([0x5EED0000], np.uint32)
    out = ctx.frontend_pairs(t(bgr), 1, 150, ca, sa, t(pat), t(seeds.view(np.int32)), 64, 10.0)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for f in range(2):
        k = int(G[f"e_counts{f}"][0])
        assert out["n"][f] == k
        assert np.array_equal(out["xy"][f, :k], G[f"e_xy{f}"]) and np.array_equal(out["desc"][f, :k], G[f"e_desc{f}"])
        assert np.array_equal(out["nodes"][f, :k], G[f"e_nodes{f}"])
    k = len(G["p_matches"])
    assert out["best"][0, 3] == k and np.array_equal(out["matches"][0, :k], G["p_matches"])
    assert np.array_equal(bits(out["F"][0]), bits(G["p_F"]))


def test_grid_orb_golden(ctx):
    G2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_v2_grid.npz"))
    dev = t(G2["g_bgr"][None].copy())
    out = ctx.extract_features_grid(dev, 2, 2, t(G2["g_pattern"]), 8192)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    n = len(G2["g_xy"])
    assert out["n"][0] == n
    assert np.array_equal(dev[0].cpu().numpy(), G2["g_outlined"])
    assert np.array_equal(bits(out["xy"][0, :n]), bits(G2["g_xy"])) and np.array_equal(out["desc"][0, :n], G2["g_desc"])
    assert np.array_equal(bits(out["angle_octave"][0, :n]), bits(G2["g_angle_octave"]))

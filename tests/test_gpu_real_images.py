"""The front-end on PHOTOGRAPHS (tests/golden/real_v1.npz: four public-domain photographs, second frame = the first after a
small camera motion, resampled at sub-pixel positions), bit-exact against the stored oracle outputs and, at other
settings, against the oracle run here.  Every other input of the suite is synthetic texture or noise; saturated plateaus,
smooth gradients and JPEG-like near-ties are where the certified margins of the two-tier corner detector and the from-memory
details of the OpenCV routines are most likely to bite."""
import os

import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def real():
    g = np.load(os.path.join(HERE, "golden", "real_v1.npz"))
    frames = [synth.real_pair(g[f"crop{i}"], tuple(g["motions"][i])) for i in range(len(g["names"]))]
    return g, np.stack([a for a, _ in frames] + [b for _, b in frames])      # [last frames | current frames]


def test_frontend_pairs_on_photographs_equals_the_golden_outputs(ctx, real):
    g, bgr = real
    P = bgr.shape[0] // 2
    maxc, hyp, seed = (int(v) for v in g["params"])
    thr = float(g["threshold"][0])
    ca, sa = synth.keypoint_rotation()
    seeds = torch.from_numpy((np.uint32(seed) ^ np.arange(P, dtype=np.uint32)).view(np.int32)).cuda()
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, None, seeds, hyp, thr)
    ctx.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(P):
        for tag, f in (("a", i), ("b", P + i)):
            xy = g[f"xy_{tag}{i}"]
            n = len(xy)
            assert o["n"][f] == n, (i, tag, o["n"][f], n)
            assert np.array_equal(o["xy"][f, :n], xy) and np.array_equal(o["desc"][f, :n], g[f"desc_{tag}{i}"]), (i, tag)
            assert np.array_equal(o["nodes"][f, :n], g[f"nodes_{tag}{i}"]), (i, tag)
        m = g[f"matches{i}"]
        assert o["best"][i, 3] == len(m) and np.array_equal(o["matches"][i, :len(m)], m), i
        assert np.array_equal(o["F"][i].view(np.uint32), g[f"F{i}"].view(np.uint32)), i


@pytest.mark.parametrize("maxc,hyp,quality_note", [(300, 64, "budget below the corner count: the selection truncates"),
                                                   (2000, 128, "budget far above it")])
def test_photographs_at_other_settings_equal_the_oracle(ctx, oracle, real, maxc, hyp, quality_note):
    g, bgr = real
    P = bgr.shape[0] // 2
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    seeds = torch.arange(40, 40 + P, dtype=torch.int32).cuda()
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, None, seeds, hyp, 10.0)
    ctx.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(P):
        fa = oracle.extract_features(bgr[i], maxc, ca, sa, pat)
        fb = oracle.extract_features(bgr[P + i], maxc, ca, sa, pat)
        for f, r in ((i, fa), (P + i, fb)):
            assert o["n"][f] == r["n"], (i, quality_note)
            assert np.array_equal(o["xy"][f, :r["n"]], r["xy"]) and np.array_equal(o["desc"][f, :r["n"]], r["desc"]), i
            assert np.array_equal(o["nodes"][f, :r["n"]], r["nodes"]), i
        r = oracle.match_features(fa["xy"], fa["desc"], fb["xy"], fb["desc"], 40 + i, hyp, 10.0)
        k = len(r["matches"])
        assert o["best"][i, 3] == k and np.array_equal(o["matches"][i, :k], r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(o["F"][i].view(np.uint32), np.asarray(r["F"], np.float32).view(np.uint32)), i


@pytest.mark.parametrize("w,h", [(1241, 376), (1226, 370), (723, 481)])
def test_photographic_windows_of_widths_the_device_pads(ctx, oracle, w, h):
    """Windows of the photographs at frame sizes whose width is no multiple of 4 (the first two are the KITTI odometry sizes): the
    device gives its gray / blurred planes padded rows with a mirrored tail and runs the dword kernels on them; the grid extractor
    does the same with level 0 of its pyramids.  Everything equals the oracle on the same bytes."""
    P, maxc, hyp = 2, 1500, 128
    dev = torch.device("cuda", 0)
    bgr_t = synth.frames_torch_photo(77 + w, P, w, h, dev)
    bgr = bgr_t.cpu().numpy()
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    seeds = torch.arange(7, 7 + P, dtype=torch.int32).cuda()
    out = ctx.frontend_pairs(bgr_t, P, maxc, ca, sa, None, seeds, hyp, 10.0)
    ctx.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(P):
        fa = oracle.extract_features(bgr[i], maxc, ca, sa, pat)
        fb = oracle.extract_features(bgr[P + i], maxc, ca, sa, pat)
        assert fa["n"] > 200, "the window should hold corners"
        for f, r in ((i, fa), (P + i, fb)):
            assert o["n"][f] == r["n"], (w, h, f)
            assert np.array_equal(o["xy"][f, :r["n"]], r["xy"]) and np.array_equal(o["desc"][f, :r["n"]], r["desc"]), (w, h, f)
            assert np.array_equal(o["nodes"][f, :r["n"]], r["nodes"]), (w, h, f)
        r = oracle.match_features(fa["xy"], fa["desc"], fb["xy"], fb["desc"], 7 + i, hyp, 10.0)
        k = len(r["matches"])
        assert o["best"][i, 3] == k and np.array_equal(o["matches"][i, :k], r["matches"]), (w, h, i)
        if r["rc"] == 0:
            assert np.array_equal(o["F"][i].view(np.uint32), np.asarray(r["F"], np.float32).view(np.uint32)), (w, h, i)
    # the grid extractor on the first frame (it outlines the cells into the image it is given: a copy)
    work = bgr_t[:1].clone()
    g = ctx.extract_features_grid(work, 3, 4, torch.from_numpy(pat).cuda(), 8192)
    ctx.synchronize()
    ref_img, xy, desc, ao = oracle.extract_features_grid(bgr[0], 3, 4, pat)
    n = int(g["n"][0])
    assert n == len(xy) and np.array_equal(work[0].cpu().numpy(), ref_img), (w, h)
    assert np.array_equal(g["xy"][0, :n].cpu().numpy().view(np.uint32), xy.view(np.uint32)), (w, h)
    assert np.array_equal(g["desc"][0, :n].cpu().numpy(), desc), (w, h)
    assert np.array_equal(g["angle_octave"][0, :n].cpu().numpy().view(np.uint32), ao.view(np.uint32)), (w, h)


def test_stage_outputs_on_photographs(ctx, oracle, real):
    """cvtColor, the response image, the blur -- whole images compared, not only what survives the selection."""
    _, bgr = real
    t = torch.from_numpy(bgr).cuda()
    gray = ctx.bgr2gray(t)
    eig = ctx.min_eigen(gray)
    blur = ctx.gaussian7(gray)
    ctx.synchronize()
    gray, eig, blur = gray.cpu().numpy(), eig.cpu().numpy(), blur.cpu().numpy()
    for f in range(bgr.shape[0]):
        ref = oracle.bgr2gray(bgr[f])
        assert np.array_equal(gray[f], ref), f
        assert np.array_equal(eig[f].view(np.uint32), oracle.min_eigen(ref).view(np.uint32)), f
        assert np.array_equal(blur[f], oracle.gaussian7(ref)), f


def test_grid_orb_extractor_on_photographs(ctx, oracle, real):
    """extract_features(Frame&, nrows, ncols) (src/Frame.cpp:16-51: outlines drawn into the image, ORB's pyramid, FAST, Harris
    ranking, retainBest, intensity-centroid angles, steered BRIEF) on the photographic frames: everything bit-exact."""
    _, bgr = real
    pat = synth.brief_pattern()
    for nrows, ncols, frames in ((3, 4, bgr[:4]), (1, 2, bgr[4:6])):
        dev = torch.from_numpy(np.ascontiguousarray(frames).copy()).cuda()
        out = ctx.extract_features_grid(dev, nrows, ncols, torch.from_numpy(pat).cuda(), 16384)
        ctx.synchronize()
        out = {k: v.cpu().numpy() for k, v in out.items()}
        outlined = dev.cpu().numpy()
        for f in range(frames.shape[0]):
            ref_img, xy, desc, ao = oracle.extract_features_grid(frames[f], nrows, ncols, pat)
            n = len(xy)
            assert np.array_equal(outlined[f], ref_img), f
            assert n > 50 and out["n"][f] == n, (f, out["n"][f], n)
            assert np.array_equal(out["xy"][f, :n].view(np.uint32), xy.view(np.uint32)), f
            assert np.array_equal(out["angle_octave"][f, :n].view(np.uint32), ao.view(np.uint32)), f
            assert np.array_equal(out["desc"][f, :n], desc), f


def test_pose_chain_on_photographs(ctx, oracle, real):
    """What the reference does with match_features' result (src/vslam.cpp:77-186): extract_Rt, the camera matrix,
    triangulate, the reprojection filter -- fed by the photographs' own F and inlier matches."""
    g, bgr = real
    P = bgr.shape[0] // 2
    maxc, hyp, seed = (int(v) for v in g["params"])
    ca, sa = synth.keypoint_rotation()
    seeds = torch.from_numpy((np.uint32(seed) ^ np.arange(P, dtype=np.uint32)).view(np.int32)).cuda()
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, None, seeds, hyp, float(g["threshold"][0]))
    Kmat = np.array([[525.0, 0, 320], [0, 525.0, 240], [0, 0, 1]], np.float32)
    xy1, xy2 = out["xy"][:P].contiguous(), out["xy"][P:].contiguous()
    R, tv, c2 = ctx.extract_Rt(out["F"], out["best"], Kmat)
    pts = ctx.triangulate(xy1, xy2, out["matches"], out["best"], Kmat, c2)
    ids = np.full((P, maxc), -1, np.int32)
    ids[:, ::5] = 1
    ridx, rn, rerr = ctx.reprojection_filter(pts, xy1, xy2, out["matches"], out["best"], Kmat, c2, torch.from_numpy(ids).cuda(), 4.0)
    ctx.synchronize()
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    R, tv, c2, pts, ridx, rn, rerr = (a.cpu().numpy() for a in (R, tv, c2, pts, ridx, rn, rerr))
    c1 = np.c_[Kmat, np.zeros(3, np.float32)]
    for b in range(P):
        assert o["best"][b, 0] >= 0 and np.array_equal(bits(o["F"][b]), bits(g[f"F{b}"]))
        Rr, tr = oracle.extract_Rt(o["F"][b], Kmat)
        assert np.array_equal(bits(R[b]), bits(Rr.reshape(9))) and np.array_equal(bits(tv[b]), bits(tr)), b
        c2r = oracle.camera_matrix(Kmat, Rr, tr)
        assert np.array_equal(bits(c2[b]), bits(c2r.reshape(12))), b
        k = o["best"][b, 3]
        mm = o["matches"][b, :k]
        p1, p2 = o["xy"][b][mm[:, 0]], o["xy"][P + b][mm[:, 1]]
        ref = oracle.triangulate(p1, p2, c1, c2r)
        assert np.array_equal(bits(pts[b, :k]), bits(ref)), b
        kept, err = oracle.reprojection_filter(ref, p1, p2, c1, c2r, ids[b, :k], 4.0)
        assert rn[b] == len(kept) and np.array_equal(ridx[b, :rn[b]], kept) and rerr[b] == err, b


def test_chained_pose_entry_on_photographs(ctx, oracle, real):
    """vslam_frontend_pairs_pose: extract + match + RANSAC + extract_Rt + triangulate + reprojection filter in one call, the
    matches never leaving the device; every output equals what the separate entry points give and what the oracle computes."""
    g, bgr = real
    P = bgr.shape[0] // 2
    maxc, hyp, seed = (int(v) for v in g["params"])
    ca, sa = synth.keypoint_rotation()
    seeds = torch.from_numpy((np.uint32(seed) ^ np.arange(P, dtype=np.uint32)).view(np.int32)).cuda()
    Kmat = np.array([[525.0, 0, 320], [0, 525.0, 240], [0, 0, 1]], np.float32)
    d_bgr = torch.from_numpy(bgr).cuda()
    out = ctx.frontend_pairs_pose(d_bgr, P, maxc, ca, sa, None, seeds, hyp, float(g["threshold"][0]), Kmat)
    ctx.synchronize()
    sep = ctx.frontend_pairs(d_bgr, P, maxc, ca, sa, None, seeds, hyp, float(g["threshold"][0]))
    ctx.synchronize()
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    s_ = {k: v.cpu().numpy() for k, v in sep.items()}
    for k in ("n", "xy", "desc", "nodes", "best", "matches"):
        assert np.array_equal(o[k], s_[k]), k
    assert np.array_equal(bits(o["F"]), bits(s_["F"]))
    c1 = np.c_[Kmat, np.zeros(3, np.float32)]
    for b in range(P):
        assert o["best"][b, 0] >= 0 and np.array_equal(bits(o["F"][b]), bits(g[f"F{b}"]))
        Rr, tr = oracle.extract_Rt(o["F"][b], Kmat)
        c2r = oracle.camera_matrix(Kmat, Rr, tr)
        assert np.array_equal(bits(o["R"][b]), bits(Rr.reshape(9))) and np.array_equal(bits(o["t"][b]), bits(tr)), b
        assert np.array_equal(bits(o["c2"][b]), bits(c2r.reshape(12))), b
        k = o["best"][b, 3]
        mm = o["matches"][b, :k]
        p1, p2 = o["xy"][b][mm[:, 0]], o["xy"][P + b][mm[:, 1]]
        ref = oracle.triangulate(p1, p2, c1, c2r)
        assert np.array_equal(bits(o["points4d"][b, :k]), bits(ref)), b
        kept, err = oracle.reprojection_filter(ref, p1, p2, c1, c2r, np.full(k, -1, np.int32), 4.0)
        assert o["n_inliers"][b] == len(kept) and np.array_equal(o["inlier_idx"][b, :len(kept)], kept) and o["error"][b] == err, b


def test_triangulate_points_generic_form(ctx, oracle):
    """vslam_triangulate_points: triangulate(p1, p2, c1, c2, points_4d) with any two camera matrices (include/helpers.h:19)."""
    rng = np.random.default_rng(5)
    for n in (1, 7, 130, 1000):
        p1 = rng.uniform(0, 640, (n, 2)).astype(np.float32)
        p2 = (p1 + rng.normal(0, 3, (n, 2))).astype(np.float32)
        c1 = rng.normal(0, 1, (3, 4)).astype(np.float32) * np.array([500, 500, 1], np.float32)[:, None]
        c2 = rng.normal(0, 1, (3, 4)).astype(np.float32) * np.array([500, 500, 1], np.float32)[:, None]
        pts = ctx.triangulate_points(torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda(), c1, c2)
        ctx.synchronize()
        ref = oracle.triangulate(p1, p2, c1, c2)
        assert np.array_equal(pts.cpu().numpy().view(np.uint32), ref.view(np.uint32)), n

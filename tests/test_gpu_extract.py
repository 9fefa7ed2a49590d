"""HIP extraction kernels vs the oracle, stage by stage and end to end (all bit-exact)."""
import os

import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu

SIZES = [(320, 240), (203, 131), (64, 96), (640, 480)]


def frames_for(w, h, seed, n=2):
    return synth.frames_numpy(seed, n, w, h)      # (2n, h, w, 3)


@pytest.mark.parametrize("w,h", SIZES)
def test_gray_eig_blur_bit_exact(ctx, oracle, w, h):
    bgr = frames_for(w, h, 10 + w)
    t = torch.from_numpy(bgr).cuda()
    gray = ctx.bgr2gray(t)
    eig = ctx.min_eigen(gray)
    blur = ctx.gaussian7(gray)
    ctx.synchronize()
    gray, eig, blur = gray.cpu().numpy(), eig.cpu().numpy(), blur.cpu().numpy()
    for f in range(bgr.shape[0]):
        g = oracle.bgr2gray(bgr[f])
        assert np.array_equal(gray[f], g), f
        e = oracle.min_eigen(g)
        bad = np.argwhere(eig[f].view(np.uint32) != e.view(np.uint32))
        assert bad.size == 0, (f, bad[:5], eig[f][tuple(bad[0])], e[tuple(bad[0])])
        assert np.array_equal(blur[f], oracle.gaussian7(g)), f


def test_gray_unaligned_rows(ctx, oracle):
    """width not a multiple of 4 -> the byte path of bgr2gray."""
    bgr = frames_for(131, 77, 5, n=1)
    gray = ctx.bgr2gray(torch.from_numpy(bgr).cuda()).cpu().numpy()
    for f in range(2):
        assert np.array_equal(gray[f], oracle.bgr2gray(bgr[f]))


@pytest.mark.parametrize("w,h,maxc", [(320, 240, 300), (203, 131, 5000), (640, 480, 1000), (64, 96, 50)])
def test_good_features_bit_exact(ctx, oracle, w, h, maxc):
    bgr = frames_for(w, h, 20 + w)
    gray = ctx.bgr2gray(torch.from_numpy(bgr).cuda())
    xy, n = ctx.good_features(gray, maxc)
    ctx.synchronize()
    gray, xy, n = gray.cpu().numpy(), xy.cpu().numpy(), n.cpu().numpy()
    for f in range(bgr.shape[0]):
        ref = oracle.good_features(gray[f], maxc)
        assert n[f] == len(ref), (f, n[f], len(ref))
        assert np.array_equal(xy[f, :n[f]], ref), f


@pytest.mark.parametrize("pct", [1, 20, 0, 100000])
def test_exact_window_size_never_changes_results(ctx, oracle, pct):
    """The two-tier detector evaluates only the best-ranked possible corners exactly (VSLAM_OPT_CORNER_WINDOW_PCT) and
    redoes a frame with all of them when the selection needs more: pct 1 and 20 leave far fewer than max_corners above
    the cut, so every frame takes the redo; 0 / 100000 evaluate everything up front.  Same corners, same order."""
    ctx.set_option(ctx.OPT_CORNER_WINDOW_PCT, pct)
    try:
        for (w, h, maxc, md) in ((640, 480, 1000, 3.0), (320, 240, 300, 7.0), (1280, 96, 2000, 1.0)):
            bgr = frames_for(w, h, 77 + w)
            gray = ctx.bgr2gray(torch.from_numpy(bgr).cuda())
            xy, n = ctx.good_features(gray, maxc, min_distance=md)
            ctx.synchronize()
            gray, xy, n = gray.cpu().numpy(), xy.cpu().numpy(), n.cpu().numpy()
            for f in range(bgr.shape[0]):
                ref = oracle.good_features(gray[f], maxc, min_dist=md)
                assert n[f] == len(ref), (pct, w, h, f, n[f], len(ref))
                assert np.array_equal(xy[f, :n[f]], ref), (pct, w, h, f)
    finally:
        ctx.set_option(ctx.OPT_CORNER_WINDOW_PCT, 135)


def test_good_features_plateaus_and_flat(ctx, oracle):
    """Synthetic eigenvalue plateaus: a checkerboard gives many pixels with EQUAL response, so the
    address tie-break of the sort and the equal-value suppression chains are exercised; a flat
    frame gives max == 0 and no corners."""
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    board = (((yy // 8) + (xx // 8)) % 2 * 200 + 20).astype(np.uint8)
    flat = np.full((h, w), 77, np.uint8)
    stripes = ((xx // 5) % 2 * 180 + 30).astype(np.uint8)
    gray = np.stack([board, flat, stripes])
    xy, n = ctx.good_features(torch.from_numpy(gray).cuda(), 4000)
    xy, n = xy.cpu().numpy(), n.cpu().numpy()
    for f in range(3):
        ref = oracle.good_features(gray[f], 4000)
        assert n[f] == len(ref), f
        assert np.array_equal(xy[f, :n[f]], ref), f
    assert n[1] == 0


def _noise_and_sparse(n_each, w, h, seed):
    """n_each pure-noise frames (every ninth pixel or so is a 3x3 maximum: the longest lists there are) alternating with
    n_each nearly flat frames that carry a dozen rectangles."""
    rng = np.random.default_rng(seed)
    kinds, imgs = [], []
    for i in range(2 * n_each):
        if i % 2 == 0:
            kinds.append("n")
            imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
        else:
            img = np.full((h, w), 90, np.uint8)
            for _ in range(12):
                x, y = rng.integers(8, w - 24), rng.integers(8, h - 24)
                img[y:y + 9, x:x + 11] = rng.integers(150, 250)
            kinds.append("s")
            imgs.append(img)
    return kinds, np.stack(imgs)


@pytest.mark.parametrize("cap", [0, 700, 40, -1])
def test_bounded_corner_lists_never_change_results(ctx, oracle, cap):
    """The detector's per-frame lists are bounded (VSLAM_OPT_CORNER_LIST_CAP; default 16 x max_corners + 4096 entries);
    a frame that overflows its list is redone by the plain exact pipeline on whole-image scratch from a pool.  Noise
    frames overflow the default bound, cap = 700 makes the textured ones overflow as well, 40 nearly everything, -1 sizes
    the lists for the whole image (nothing overflows): the same corners in the same order every time."""
    ctx.set_option(ctx.OPT_CORNER_LIST_CAP, cap)
    try:
        w, h, maxc = 320, 240, 60
        kinds, gray = _noise_and_sparse(2, w, h, 5)
        assert kinds.count("n") == 2
        bgr = frames_for(w, h, 12, n=1)                       # two textured frames
        tex = np.stack([oracle.bgr2gray(b) for b in bgr])
        # the pool of a call of up to 64 frames has 4 sets: no more than four frames may overflow
        gray = {0: np.concatenate([gray, tex]), -1: np.concatenate([gray, tex]), 40: gray,
                700: np.stack([gray[0], tex[0], gray[2], tex[1]])}[cap]
        for md in (3.0, 1.0):
            xy, n = ctx.good_features(torch.from_numpy(gray).cuda(), maxc, min_distance=md)
            ctx.synchronize()
            xy, n = xy.cpu().numpy(), n.cpu().numpy()
            for f in range(gray.shape[0]):
                ref = oracle.good_features(gray[f], maxc, min_dist=md)
                assert n[f] == len(ref) and np.array_equal(xy[f, :n[f]], ref), (cap, md, f, n[f], len(ref))
    finally:
        ctx.set_option(ctx.OPT_CORNER_LIST_CAP, 0)


def test_corner_pool_exhaustion_is_reported(ctx, oracle):
    """More overflowing frames in ONE call than the pool has sets (4 for up to 64 frames): the frames that found no set
    come back without corners and the context reports VSLAM_ERR_CAPACITY -- never a silently wrong answer; the frames
    that did find a set are exact, and the next call is clean."""
    from vslam_amd import VslamError
    w, h, maxc = 320, 240, 60
    bgr = frames_for(w, h, 13, n=4)                           # 8 textured frames
    gray = np.stack([oracle.bgr2gray(b) for b in bgr])
    ctx.set_option(ctx.OPT_CORNER_LIST_CAP, 40)
    try:
        xy, n = ctx.good_features(torch.from_numpy(gray).cuda(), maxc)
        with pytest.raises(VslamError, match="CAPACITY"):
            ctx.synchronize()
        xy, n = xy.cpu().numpy(), n.cpu().numpy()
        refs = [oracle.good_features(g, maxc) for g in gray]
        done = [f for f in range(8) if n[f] > 0]
        assert len(done) == 4 and all(len(refs[f]) > 0 for f in range(8))
        for f in done:
            assert n[f] == len(refs[f]) and np.array_equal(xy[f, :n[f]], refs[f]), f
    finally:
        ctx.set_option(ctx.OPT_CORNER_LIST_CAP, 0)
    xy, n = ctx.good_features(torch.from_numpy(gray).cuda(), maxc)
    ctx.synchronize()
    xy, n = xy.cpu().numpy(), n.cpu().numpy()
    for f in range(8):
        assert n[f] == len(refs[f]) and np.array_equal(xy[f, :n[f]], refs[f]), f


def test_good_features_other_min_distance(ctx, oracle):
    bgr = frames_for(320, 240, 31)
    gray = ctx.bgr2gray(torch.from_numpy(bgr).cuda())
    g = gray.cpu().numpy()
    for md in (0.0, 1.0, 2.5, 7.0):
        xy, n = ctx.good_features(gray, 800, min_distance=md)
        xy, n = xy.cpu().numpy(), n.cpu().numpy()
        for f in range(2):
            ref = oracle.good_features(g[f], 800, min_dist=md)
            assert n[f] == len(ref) and np.array_equal(xy[f, :n[f]], ref), (md, f)


def test_describe_bit_exact_and_border_filter(ctx, oracle):
    w, h = 320, 240
    bgr = frames_for(w, h, 41)
    gray = ctx.bgr2gray(torch.from_numpy(bgr).cuda())
    blur = ctx.gaussian7(gray)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    rng = np.random.default_rng(1)
    K = 500
    pts = np.rint(np.stack([rng.uniform(0, w - 1, (4, K)), rng.uniform(0, h - 1, (4, K))], -1)).astype(np.float32)
    pts[0, :6] = [[31, 31], [30, 31], [31, 30], [w - 32, h - 32], [w - 31, 50], [50, h - 31]]   # border cases
    n = np.array([K, 300, 0, 1], np.int32)
    xy_out, desc, n_out = ctx.orb_describe(blur, torch.from_numpy(pts).cuda(), torch.from_numpy(n).cuda(), ca, sa,
                                           torch.from_numpy(pat).cuda())
    xy_out, desc, n_out, blur = xy_out.cpu().numpy(), desc.cpu().numpy(), n_out.cpu().numpy(), blur.cpu().numpy()
    for f in range(4):
        rd, keep = oracle.orb_describe(blur[f], pts[f, :n[f]], ca, sa, pat)
        assert n_out[f] == len(keep), f
        assert np.array_equal(xy_out[f, :len(keep)], pts[f, :n[f]][keep]), f
        assert np.array_equal(desc[f, :len(keep)], rd), f


@pytest.mark.parametrize("w,h,maxc", [(320, 240, 400), (640, 480, 500)])
def test_extract_features_end_to_end(ctx, oracle, w, h, maxc):
    """extract_features(Frame&), src/Frame.cpp:53-80: points, descriptors, k-d tree, both counts."""
    bgr = frames_for(w, h, 50 + w)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    out = ctx.extract_features(torch.from_numpy(bgr).cuda(), maxc, ca, sa, torch.from_numpy(pat).cuda())
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for f in range(bgr.shape[0]):
        ref = oracle.extract_features(bgr[f], maxc, ca, sa, pat)
        k = ref["n"]
        assert out["n"][f] == k and out["n_detected"][f] == ref["n_detected"], f
        assert k > maxc // 2, "synthetic frame should be corner-rich"
        assert np.array_equal(out["xy"][f, :k], ref["xy"]), f
        assert np.array_equal(out["desc"][f, :k], ref["desc"]), f
        assert np.array_equal(out["nodes"][f, :k], ref["nodes"]), f


def test_null_pattern_means_orbs_learned_table(ctx, oracle):
    """A NULL d_pattern is the default: ORB's learned table (vslam_brief_pattern_31), i.e. what cv::ORB::compute samples
    (src/Frame.cpp:57,68) -- in extract_features, in orb_describe and in the grid extractor."""
    w, h, maxc = 320, 240, 300
    bgr = frames_for(w, h, 77)
    pat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "brief_pattern_31.npy"))
    ca, sa = synth.keypoint_rotation()
    t = torch.from_numpy(bgr).cuda()
    out = ctx.extract_features(t, maxc, ca, sa, None)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for f in range(bgr.shape[0]):
        ref = oracle.extract_features(bgr[f], maxc, ca, sa, pat)
        k = ref["n"]
        assert out["n"][f] == k and np.array_equal(out["desc"][f, :k], ref["desc"]), f
    blur = ctx.gaussian7(ctx.bgr2gray(t))
    xy = torch.from_numpy(out["xy"]).cuda(); n = torch.from_numpy(out["n"]).cuda()
    a = ctx.orb_describe(blur, xy, n, ca, sa, None)
    b = ctx.orb_describe(blur, xy, n, ca, sa, torch.from_numpy(pat).cuda())
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    ga = ctx.extract_features_grid(t.clone(), 2, 2, None, 4096)
    gb = ctx.extract_features_grid(t.clone(), 2, 2, torch.from_numpy(pat).cuda(), 4096)
    ctx.synchronize()
    assert all(torch.equal(ga[k], gb[k]) for k in ga) and int(ga["n"].min()) > 50


def test_frontend_pairs_end_to_end(ctx, oracle):
    """The whole path bench.py times, on C1-like inputs (640x480, 500 keypoints, 512 hypotheses):
    extract both frames, match, RANSAC; inlier matches and F must equal the oracle's."""
    w, h, maxc, Hy, thr, P = 640, 480, 500, 512, 10.0, 2
    bgr = synth.frames_numpy(0x5EED0000, P, w, h)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    seeds = np.array([0x5EED0000 ^ p for p in range(P)], np.uint32)
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, torch.from_numpy(pat).cuda(),
                             torch.from_numpy(seeds.view(np.int32)).cuda(), Hy, thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for p in range(P):
        a = oracle.extract_features(bgr[p], maxc, ca, sa, pat)
        b = oracle.extract_features(bgr[P + p], maxc, ca, sa, pat)
        ref = oracle.match_features(a["xy"], a["desc"], b["xy"], b["desc"], int(seeds[p]), Hy, thr)
        assert ref["rc"] == 0 and ref["prelim"] > 100, "synthetic pair should match"
        k = len(ref["matches"])
        assert out["best"][p, 3] == k and k > 30, p
        assert np.array_equal(out["matches"][p, :k], ref["matches"]), p
        assert np.array_equal(out["F"][p].view(np.uint32), ref["F"].view(np.uint32)), p


def test_good_features_tiny_max_corners_forces_slow_path(ctx, oracle):
    """With a handful of corners wanted the LDS rank window (2x maxCorners) is too small when the
    best-ranked candidates sit on response plateaus and mostly suppress each other: the kernel must
    fall back to suppression over every candidate and still equal the sequential greedy."""
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    board = (((yy // 8) + (xx // 8)) % 2 * 200 + 20).astype(np.uint8)
    blobs = np.full((h, w), 30, np.uint8)
    for cy, cx in [(30, 40), (30, 44), (34, 40), (80, 100), (80, 103), (60, 20)]:
        blobs[cy:cy + 3, cx:cx + 3] = 220
    noisy = synth.frames_numpy(77, 1, w, h)[0, :, :, 1]
    gray = np.stack([board, blobs, noisy])
    g = torch.from_numpy(gray).cuda()
    for maxc in (1, 2, 3, 4, 7, 16):
        xy, n = ctx.good_features(g, maxc)
        xy, n = xy.cpu().numpy(), n.cpu().numpy()
        for f in range(3):
            ref = oracle.good_features(gray[f], maxc)
            assert n[f] == len(ref), (maxc, f, n[f], len(ref))
            assert np.array_equal(xy[f, :n[f]], ref), (maxc, f)


def test_frontend_headline_size_one_pair(ctx, oracle):
    """One frame pair at BASELINE.json's headline shape (1280x720, 2000 keypoints, 4096 hypotheses):
    every output of the whole path equals the oracle's, bit for bit."""
    w, h, maxc, Hy, thr = 1280, 720, 2000, 4096, 10.0
    bgr = synth.frames_numpy(0x5EED0002, 1, w, h)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    seeds = np.array([0x5EED0002], np.uint32)
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), 1, maxc, ca, sa, torch.from_numpy(pat).cuda(),
                             torch.from_numpy(seeds.view(np.int32)).cuda(), Hy, thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    ex = [oracle.extract_features(bgr[f], maxc, ca, sa, pat) for f in range(2)]
    for f in range(2):
        k = ex[f]["n"]
        assert out["n"][f] == k and k > 1500
        assert np.array_equal(out["xy"][f, :k], ex[f]["xy"]) and np.array_equal(out["desc"][f, :k], ex[f]["desc"])
        assert np.array_equal(out["nodes"][f, :k], ex[f]["nodes"])
    ref = oracle.match_features(ex[0]["xy"], ex[0]["desc"], ex[1]["xy"], ex[1]["desc"], int(seeds[0]), Hy, thr)
    k = len(ref["matches"])
    assert ref["rc"] == 0 and out["best"][0, 3] == k and k > 500
    assert np.array_equal(out["matches"][0, :k], ref["matches"])
    assert np.array_equal(out["F"][0].view(np.uint32), ref["F"].view(np.uint32))

"""extract_Rt / camera matrix / triangulate on the device vs the oracle (bit-exact), fed by a real
RANSAC result so the chain F -> (R, t) -> c2 -> 3-D points is the one src/vslam.cpp:77-186 runs."""
import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_pose_chain_bit_exact(ctx, oracle):
    B, K, Hy, thr = 5, 400, 128, 10.0
    w, h, f = 1280, 720, 525.0
    Kmat = np.array([[f, 0, w // 2], [0, f, h // 2], [0, 0, 1]], np.float32)       # src/vslam.cpp:32
    xy1 = np.zeros((B, K, 2), np.float32); xy2 = np.zeros((B, K, 2), np.float32)
    pairs = np.zeros((B, K, 2), np.int32); m = np.zeros(B, np.int32)
    sizes = [400, 120, 9, 300, 5]
    for b, n in enumerate(sizes):
        xy1[b], xy2[b], _ = synth.two_view_points(900 + b, K, w, h, inlier_frac=0.75)
        pairs[b, :n] = np.stack([np.arange(n), np.arange(n)], 1)
        m[b] = n
    sets = np.stack([oracle.ransac_sets(30 + b, max(n, 8), Hy) if n >= 8 else np.zeros((Hy, 8), np.int32) for b, n in enumerate(sizes)])
    t = lambda a: torch.from_numpy(a).cuda()
    out = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
    R, tv, c2 = ctx.extract_Rt(out["F"], out["best"], Kmat)
    pts = ctx.triangulate(t(xy1), t(xy2), out["matches"], out["best"], Kmat, c2)
    ids = np.full((B, K), -1, np.int32)
    ids[:, ::7] = 3                                   # some matches are skipped through the match-indexed id test
    ridx, rn, rerr = ctx.reprojection_filter(pts, t(xy1), t(xy2), out["matches"], out["best"], Kmat, c2, t(ids), 4.0)
    ctx.synchronize()
    ridx, rn, rerr = ridx.cpu().numpy(), rn.cpu().numpy(), rerr.cpu().numpy()
    Fh, best, matches = out["F"].cpu().numpy(), out["best"].cpu().numpy(), out["matches"].cpu().numpy()
    R, tv, c2, pts = R.cpu().numpy(), tv.cpu().numpy(), c2.cpu().numpy(), pts.cpu().numpy()
    c1 = np.c_[Kmat, np.zeros(3, np.float32)]
    for b, n in enumerate(sizes):
        if best[b, 0] < 0:
            assert not R[b].any()                      # nothing accepted: outputs untouched
            continue
        Rr, tr = oracle.extract_Rt(Fh[b], Kmat)
        assert np.array_equal(bits(R[b]), bits(Rr.reshape(9))), b
        assert np.array_equal(bits(tv[b]), bits(tr)), b
        c2r = oracle.camera_matrix(Kmat, Rr, tr)
        assert np.array_equal(bits(c2[b]), bits(c2r.reshape(12))), b
        k = best[b, 3]
        mm = matches[b, :k]
        ref = oracle.triangulate(xy1[b][mm[:, 0]], xy2[b][mm[:, 1]], c1, c2r)
        assert np.array_equal(bits(pts[b, :k]), bits(ref)), b
        kept, err = oracle.reprojection_filter(ref, xy1[b][mm[:, 0]], xy2[b][mm[:, 1]], c1, c2r, ids[b, :k], 4.0)
        assert rn[b] == len(kept) and np.array_equal(ridx[b, :rn[b]], kept), b
        assert rerr[b] == err, b
        # and it is a rotation with unit translation
        assert abs(np.linalg.det(Rr.astype(np.float64)) - 1) < 1e-4 and abs(np.linalg.norm(tr) - 1) < 1e-5


def test_extract_Rt_degenerate_inputs(ctx, oracle):
    """Rank-deficient / zero / huge F matrices drive the SVD's zero-singular-value branch."""
    Kmat = np.array([[525, 0, 640], [0, 525, 360], [0, 0, 1]], np.float32)
    Fs = np.zeros((6, 9), np.float32)
    Fs[1] = np.eye(3, dtype=np.float32).reshape(9)
    Fs[2, 0] = 1.0
    Fs[3] = np.arange(9, dtype=np.float32) * 1e-6
    Fs[4] = np.array([0, -1e-7, 3e-5, 1e-7, 0, -2e-4, -3e-5, 2e-4, 0], np.float32)     # skew-symmetric: pure translation
    Fs[5] = np.random.default_rng(0).normal(size=9).astype(np.float32) * 1e4
    R, tv, c2 = ctx.extract_Rt(torch.from_numpy(Fs).cuda(), None, Kmat)
    R, tv = R.cpu().numpy(), tv.cpu().numpy()
    for b in range(6):
        Rr, tr = oracle.extract_Rt(Fs[b], Kmat)
        assert np.array_equal(bits(R[b]), bits(Rr.reshape(9))), b
        assert np.array_equal(bits(tv[b]), bits(tr)), b

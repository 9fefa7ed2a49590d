"""extract_features(Frame&, nrows, ncols) (src/Frame.cpp:16-51) on the device vs the oracle: outlined
image, keypoint coordinates and order, angles, octaves and descriptors — all bit-exact."""
import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,nrows,ncols", [(640, 480, 5, 5), (640, 480, 2, 3), (333, 250, 1, 1), (1280, 720, 5, 5)])
def test_grid_orb_bit_exact(ctx, oracle, w, h, nrows, ncols):
    P = 1
    bgr = synth.frames_numpy(70 + w + nrows, P, w, h)
    pat = synth.brief_pattern()
    cap = 16384
    dev = torch.from_numpy(bgr.copy()).cuda()
    out = ctx.extract_features_grid(dev, nrows, ncols, torch.from_numpy(pat).cuda(), cap)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    outlined = dev.cpu().numpy()
    for f in range(2 * P):
        ref_img, xy, desc, ao = oracle.extract_features_grid(bgr[f], nrows, ncols, pat)
        assert np.array_equal(outlined[f], ref_img), f            # cv::rectangle side effect, :32
        n = len(xy)
        assert n > 100, "synthetic frame should give ORB keypoints"
        assert out["n"][f] == n, (f, out["n"][f], n)
        assert np.array_equal(out["angle_octave"][f, :n].view(np.uint32), ao.view(np.uint32)), f
        assert np.array_equal(out["xy"][f, :n].view(np.uint32), xy.view(np.uint32)), f
        assert np.array_equal(out["desc"][f, :n], desc), f
        assert np.all(np.diff(ao[:, 1]) >= 0)                     # ORB::compute groups by level


def test_grid_orb_long_lists_use_global_scratch(ctx, oracle):
    """One 640x480 cell of pure noise gives FAST lists far longer than the 4095 entries the LDS
    selection buffers hold, so retainBest's replay runs out of the per-slot global scratch."""
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (1, 480, 640, 3), dtype=np.uint8)
    pat = synth.brief_pattern()
    gray = oracle.bgr2gray(bgr[0])
    assert len(oracle.fast9_16(gray, 20)) > 6000
    dev = torch.from_numpy(bgr.copy()).cuda()
    out = ctx.extract_features_grid(dev, 1, 1, torch.from_numpy(pat).cuda(), 4096)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    ref_img, xy, desc, ao = oracle.extract_features_grid(bgr[0], 1, 1, pat)
    n = len(xy)
    assert out["n"][0] == n and n >= 400
    assert np.array_equal(out["xy"][0, :n].view(np.uint32), xy.view(np.uint32))
    assert np.array_equal(out["angle_octave"][0, :n].view(np.uint32), ao.view(np.uint32))
    assert np.array_equal(out["desc"][0, :n], desc)

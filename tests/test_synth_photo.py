"""synth.frames_torch_photo (bench.py --data photo): deterministic, integer resampling that equals synth.resample_fixed wherever
the window stays inside the photograph, mirrored continuation beyond it."""
import os

import numpy as np
import torch

from vslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_photo_frames_are_deterministic_and_photographic():
    dev = torch.device("cpu")
    a = synth.frames_torch_photo(123, 3, 320, 240, dev)
    b = synth.frames_torch_photo(123, 3, 320, 240, dev)
    c = synth.frames_torch_photo(124, 3, 320, 240, dev)
    assert a.shape == (6, 240, 320, 3) and a.dtype == torch.uint8
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.float().std() > 20          # image content, not a flat field
    # frame B of a pair is frame A after a small motion: strongly correlated, not identical
    fa, fb = a[0].float().mean(2), a[3].float().mean(2)
    r = np.corrcoef(fa.numpy().ravel(), fb.numpy().ravel())[0, 1]
    assert 0.5 < r < 0.9999


def test_integer_resampling_matches_resample_fixed_inside_the_crop():
    g = np.load(os.path.join(ROOT, "tests", "golden", "real_v1.npz"))
    crop = g["crop0"]
    # a window that stays inside the 736 x 576 crop: identity scale, an offset, a small rotation
    th = np.deg2rad(1.0)
    ca, sa = float(np.cos(th)), float(np.sin(th))
    ref = synth.resample_fixed(crop, 200, 300, ca, -sa, sa, ca, 150.25, 120.5)
    # the same through the torch sampler's arithmetic (reimplemented here on the crop with the generator's formula)
    Q = 1 << 16
    xs = torch.arange(300, dtype=torch.int64)[None, :]
    ys = torch.arange(200, dtype=torch.int64)[:, None]
    fx = int(round(ca * Q)) * xs + int(round(-sa * Q)) * ys + int(round(150.25 * Q))
    fy = int(round(sa * Q)) * xs + int(round(ca * Q)) * ys + int(round(120.5 * Q))
    H, W = crop.shape[:2]
    assert int(fx.min()) >= 0 and int(fx.max()) <= (W - 1) * Q and int(fy.min()) >= 0 and int(fy.max()) <= (H - 1) * Q
    x0, y0 = fx >> 16, fy >> 16
    x1, y1 = torch.clamp(x0 + 1, max=W - 1), torch.clamp(y0 + 1, max=H - 1)
    wx, wy = ((fx & (Q - 1)) >> 8)[..., None], ((fy & (Q - 1)) >> 8)[..., None]
    src = torch.from_numpy(crop).to(torch.int64)
    top = src[y0, x0] * (256 - wx) + src[y0, x1] * wx
    bot = src[y1, x0] * (256 - wx) + src[y1, x1] * wx
    out = ((top * (256 - wy) + bot * wy + (1 << 15)) >> 16).to(torch.uint8).numpy()
    assert np.array_equal(out, ref)

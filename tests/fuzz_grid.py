"""Randomised parity of the grid ORB/FAST extractor (extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51) after its
round-6 rebuild: random frame sizes (widths on and off the 4-pixel path, cells from a few levels deep to too small for any
keypoint), grids of 1 x 1 .. 6 x 6 cells, content kinds (synthetic scenes, noise -- the candidate list of the FAST kernel
overflows and its chunk-loop fall-back runs --, blocks on noise, checkerboards, ramps, flat), one to three frames per call,
keypoint capacities below and above what is found.  Outlined image, keypoint count, coordinates and order, angles, octaves and
descriptors must equal the oracle's.

`python tests/fuzz_grid.py SEED SECONDS` runs it for a wall-clock budget; tests/test_gpu_fuzz.py runs a fixed slice."""
import sys
import time

import numpy as np
import torch


def run(ctx, o, seed, cases=None, seconds=None):
    from vslam_amd import synth
    rng = np.random.default_rng(seed)
    pat = synth.brief_pattern()
    dpat = torch.from_numpy(pat).cuda()

    def content(kind, n, h, w):
        if kind == 0:
            return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        if kind == 1:
            return synth.frames_numpy(int(rng.integers(1, 1 << 30)), (n + 1) // 2, w, h)[:n]
        if kind == 2:
            g = rng.integers(100, 140, (n, h, w, 3), dtype=np.uint8)
            for f in range(n):
                for _ in range(max(1, w * h // 300)):
                    x, y = rng.integers(0, w), rng.integers(0, h)
                    g[f, y:y + rng.integers(2, 20), x:x + rng.integers(2, 20)] = rng.integers(0, 256, 3)
            return g
        if kind == 3:
            yy, xx = np.mgrid[0:h, 0:w]
            p = int(rng.integers(3, 12))
            c = ((((yy // p) + (xx // p)) % 2) * int(rng.integers(50, 220)) + 20).astype(np.uint8)
            return np.stack([np.stack([c, c, c], -1)] * n)
        if kind == 4:
            yy, xx = np.mgrid[0:h, 0:w]
            return np.stack([np.stack([((xx * 3 + yy * 5 + f * 7) % 256).astype(np.uint8)] * 3, -1) for f in range(n)])
        return np.full((n, h, w, 3), int(rng.integers(0, 256)), np.uint8)

    t0, done = time.time(), 0
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        w = int(rng.choice([64, 96, 128, 200, 256, 320, 333, 400, 512, 640]))
        if rng.random() < 0.3:
            w += int(rng.integers(1, 4))
        h = int(rng.choice([64, 90, 120, 180, 240, 250, 300, 480]))
        nrows, ncols = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        if w // ncols < 7 or h // nrows < 7:
            continue
        n = int(rng.integers(1, 4))
        kind = int(rng.choice([0, 1, 1, 1, 2, 2, 3, 4, 5]))
        cap = int(rng.choice([64, 1000, 20000]))
        bgr = np.ascontiguousarray(content(kind, n, h, w))
        dev = torch.from_numpy(bgr.copy()).cuda()
        out = ctx.extract_features_grid(dev, nrows, ncols, dpat, cap)
        ctx.synchronize()
        out = {k: v.cpu().numpy() for k, v in out.items()}
        outlined = dev.cpu().numpy()
        for f in range(n):
            ref_img, xy, desc, ao = o.extract_features_grid(bgr[f], nrows, ncols, pat)
            tag = (w, h, nrows, ncols, kind, n, cap, f)
            assert np.array_equal(outlined[f], ref_img), ("outline",) + tag
            k = min(len(xy), cap)
            assert out["n"][f] == k, ("count",) + tag + (int(out["n"][f]), len(xy))
            assert np.array_equal(out["xy"][f, :k].view(np.uint32), xy[:k].view(np.uint32)), ("xy",) + tag
            assert np.array_equal(out["angle_octave"][f, :k].view(np.uint32), ao[:k].view(np.uint32)), ("angle / octave",) + tag
            assert np.array_equal(out["desc"][f, :k], desc[:k]), ("descriptors",) + tag
        done += 1
    return done


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import Oracle
    from vslam_amd import Context
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    print("fuzz ok:", run(Context(0), Oracle(), seed, seconds=secs), "cases")

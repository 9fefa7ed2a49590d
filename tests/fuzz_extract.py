"""Randomised parity of the extraction stages: random frame shapes (widths on and off the vectorised path, heights
around the row-segment seams), content kinds (noise, blocks on noise, checkerboards, ramps, flat), corner budgets
and suppression distances; corner lists, blurred images and descriptors must equal the oracle's.

`python tests/fuzz_extract.py SEED SECONDS` runs it for a wall-clock budget (13 000 cases were run that way while
the streaming kernels were written); tests/test_gpu_fuzz.py runs a fixed 150 cases."""
import sys
import time

import numpy as np
import torch


def run(ctx, o, seed, cases=None, seconds=None):
    from vslam_amd import synth
    rng = np.random.default_rng(seed)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    dpat = torch.from_numpy(pat).cuda()

    def content(kind, n, h, w):
        if kind == 0:
            return rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        if kind == 1:
            g = rng.integers(100, 140, (n, h, w), dtype=np.uint8)
            for f in range(n):
                for _ in range(max(1, w * h // 400)):
                    x, y = rng.integers(0, w), rng.integers(0, h)
                    g[f, y:y + rng.integers(1, 12), x:x + rng.integers(1, 12)] = rng.integers(0, 256)
            return g
        if kind == 2:
            yy, xx = np.mgrid[0:h, 0:w]
            p = int(rng.integers(2, 9))
            return np.stack([((((yy // p) + (xx // p)) % 2) * int(rng.integers(50, 220)) + 20).astype(np.uint8)] * n)
        if kind == 3:
            yy, xx = np.mgrid[0:h, 0:w]
            return np.stack([((xx * 3 + yy * 5 + f * 7) % 256).astype(np.uint8) for f in range(n)])
        return np.full((n, h, w), int(rng.integers(0, 256)), np.uint8)

    t0, done = time.time(), 0
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        w = int(rng.choice([4, 8, 12, 64, 128, 252, 256, 260, 300, 511, 512, 516, 640, 770, 1024, 1028, 1280]))
        if rng.random() < 0.3:
            w += int(rng.integers(1, 4))
        h = int(rng.choice([4, 5, 7, 24, 47, 48, 90, 91, 135, 136, 180, 240, 271, 480]))
        n = int(rng.integers(1, 4))
        kind = int(rng.integers(0, 5))
        maxc = int(rng.choice([1, 3, 50, 300, 1000, 2000, 4000]))
        md = float(rng.choice([0.0, 1.0, 2.0, 3.0, 3.0, 3.0, 4.5, 7.0, 9.0]))
        g = content(kind, n, h, w)
        t = torch.from_numpy(g).cuda()
        xy, cnt = ctx.good_features(t, maxc, min_distance=md)
        blur = ctx.gaussian7(t)
        ctx.synchronize()
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for f in range(n):
            ref = o.good_features(g[f], maxc, min_dist=md)
            assert cnt[f] == len(ref) and np.array_equal(xy[f, :cnt[f]], ref), ("corners", w, h, kind, maxc, md, f, cnt[f], len(ref))
            assert np.array_equal(blur[f].cpu().numpy(), o.gaussian7(g[f])), ("blur", w, h, kind, f)
        if w >= 70 and h >= 70:
            K = 64
            pts = np.rint(np.stack([rng.uniform(0, w - 1, (n, K)), rng.uniform(0, h - 1, (n, K))], -1)).astype(np.float32)
            nn = np.full(n, K, np.int32)
            _, de, no = ctx.orb_describe(blur, torch.from_numpy(pts).cuda(), torch.from_numpy(nn).cuda(), ca, sa, dpat)
            de, no, bl = de.cpu().numpy(), no.cpu().numpy(), blur.cpu().numpy()
            for f in range(n):
                rd, keep = o.orb_describe(bl[f], pts[f], ca, sa, pat)
                assert no[f] == len(keep) and np.array_equal(de[f, :len(keep)], rd), ("descriptors", w, h, f)
        if w >= 4 and h >= 4 and rng.random() < 0.5:
            # the whole of extract_features from 3-byte rows (cvtColor inside the detector when the width is a multiple of 4,
            # a launch of its own otherwise): colours whose gray values are the content above plus a colour cast, rows
            # with random padding
            cast = rng.integers(-40, 41, (n, h, w, 3))
            bgr = np.clip(g[..., None].astype(np.int64) + cast, 0, 255).astype(np.uint8)
            pad = int(rng.choice([0, 0, 4, 8, 1, 2, 3]))
            rows = np.zeros((n, h, 3 * w + pad), dtype=np.uint8)
            rows[:, :, :3 * w] = bgr.reshape(n, h, 3 * w)
            if pad:
                rows[:, :, 3 * w:] = rng.integers(0, 256, (n, h, pad), dtype=np.uint8)
            kc = min(maxc, 2000)
            out = ctx.extract_features(torch.from_numpy(rows).cuda(), kc, ca, sa, dpat, width=w)
            ctx.synchronize()
            out = {k: v.cpu().numpy() for k, v in out.items()}
            for f in range(n):
                r = o.extract_features(bgr[f], kc, ca, sa, pat)
                k = r["n"]
                assert out["n_detected"][f] == r["n_detected"] and out["n"][f] == k, ("bgr counts", w, h, pad, kind, kc, f)
                assert np.array_equal(out["xy"][f, :k], r["xy"]) and np.array_equal(out["desc"][f, :k], r["desc"]), ("bgr", w, h, pad, kind, f)
                assert np.array_equal(out["nodes"][f, :k], r["nodes"]), ("bgr tree", w, h, pad, kind, f)
        done += 1
        if seconds is not None and done % 100 == 0:   # a long run says so as it goes
            print("fuzz_extract: %d cases ok, %.0f s" % (done, time.time() - t0), flush=True)
    return done


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    from oracle_lib import Oracle
    from vslam_amd import Context
    n = run(Context(0), Oracle(), int(sys.argv[1]) if len(sys.argv) > 1 else 1, seconds=float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("fuzz ok:", n, "cases")

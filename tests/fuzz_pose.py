"""Randomised parity of the pose stages (src/helpers.cpp:3-80, src/vslam.cpp:192-251) against the oracle, bit for bit: the 3 x 3 and
4 x 4 Jacobi SVDs meet matrices of every rank and scale here, not only the well-conditioned ones a RANSAC winner gives.
  * extract_Rt on random F: generic, rank 2 (true fundamental matrices of random two-view geometry), rank 1, zero rows, skew-symmetric
    (pure translation), scaled by 1e-8 .. 1e8;
  * triangulate (the declared signature, any two camera matrices) on point pairs consistent with the cameras, perturbed, coincident,
    at the principal point, far outside the image;
  * the reprojection filter on what comes out, with some matches already carrying a map point id.
`python tests/fuzz_pose.py SEED SECONDS` runs it for a wall-clock budget; tests/test_gpu_fuzz.py runs a fixed slice."""
import sys
import time

import numpy as np
import torch

bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)


def _rot(rng, deg):
    a = np.deg2rad(rng.uniform(-deg, deg, 3))
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
            np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))


def run(ctx, o, seed, cases=None, seconds=None):
    rng = np.random.default_rng(seed)
    t0, done = time.time(), 0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        w, h = int(rng.choice([320, 640, 1280])), int(rng.choice([240, 480, 720]))
        f = float(rng.choice([525.0, 300.0, 1000.0]))
        K = np.array([[f, 0, w // 2], [0, f, h // 2], [0, 0, 1]], np.float32)
        Kd = K.astype(np.float64)
        # ---- extract_Rt
        B = 24
        Fs = np.zeros((B, 9), np.float32)
        for b in range(B):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                M = rng.normal(size=(3, 3))
            elif kind == 1:                                   # a true fundamental matrix
                R, tt = _rot(rng, 5), rng.normal(size=3)
                tx = np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
                M = np.linalg.inv(Kd).T @ tx @ R @ np.linalg.inv(Kd)
            elif kind == 2:
                M = np.outer(rng.normal(size=3), rng.normal(size=3))
            elif kind == 3:
                M = rng.normal(size=(3, 3)); M[int(rng.integers(0, 3))] = 0
            elif kind == 4:
                v = rng.normal(size=3); M = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
            else:
                M = np.zeros((3, 3))
            Fs[b] = (M * 10.0 ** rng.uniform(-8, 8)).astype(np.float32).reshape(9)
        R, tv, c2 = ctx.extract_Rt(t(Fs), None, K)
        # ---- triangulate + reprojection filter, two random cameras
        n = int(rng.choice([1, 7, 64, 65, 300]))
        R2, t2 = _rot(rng, 3), rng.normal(size=3) * 0.2
        c1 = np.c_[Kd, np.zeros(3)].astype(np.float32)
        c2m = (Kd @ np.c_[R2, t2]).astype(np.float32)
        X = np.c_[rng.uniform(-2, 2, (n, 2)), rng.uniform(2, 12, n), np.ones(n)]
        p1 = X @ c1.astype(np.float64).T; p1 = p1[:, :2] / p1[:, 2:]
        p2 = X @ c2m.astype(np.float64).T; p2 = p2[:, :2] / p2[:, 2:]
        mode = int(rng.integers(0, 5))
        if mode == 1:
            p2 += rng.normal(0, 3.0, p2.shape)
        elif mode == 2:
            p2 = p1.copy()
        elif mode == 3:
            p1[: n // 2] = [w // 2, h // 2]
        elif mode == 4:
            p1 *= 50
        p1 = np.rint(p1).astype(np.float32) if rng.random() < 0.5 else p1.astype(np.float32)
        p2 = np.rint(p2).astype(np.float32) if rng.random() < 0.5 else p2.astype(np.float32)
        pts = ctx.triangulate_points(t(p1), t(p2), c1, c2m)
        # the chain form of the same points (matches = identity, c1 = [K | 0]) through the reprojection filter
        ids = np.where(rng.random(n) < 0.2, 5, -1).astype(np.int32)
        matches = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)[None]
        best = np.array([[0, n, 0, n]], np.int32)
        pts_chain = ctx.triangulate(t(p1[None]), t(p2[None]), t(matches), t(best), K, t(c2m.reshape(1, 12)))
        ridx, rn, rerr = ctx.reprojection_filter(pts_chain, t(p1[None]), t(p2[None]), t(matches), t(best), K, t(c2m.reshape(1, 12)),
                                                 t(ids[None]), 4.0)
        ctx.synchronize()
        R, tv, c2 = R.cpu().numpy(), tv.cpu().numpy(), c2.cpu().numpy()
        for b in range(B):
            Rr, tr = o.extract_Rt(Fs[b], K)
            assert np.array_equal(bits(R[b]), bits(Rr.reshape(9))) and np.array_equal(bits(tv[b]), bits(tr)), ("extract_Rt", seed, done, b)
            assert np.array_equal(bits(c2[b]), bits(o.camera_matrix(K, Rr, tr).reshape(12))), ("camera matrix", seed, done, b)
        ref = o.triangulate(p1, p2, c1, c2m)
        assert np.array_equal(bits(pts.cpu().numpy()), bits(ref)), ("triangulate", seed, done, n, mode)
        assert np.array_equal(bits(pts_chain.cpu().numpy()[0, :n]), bits(ref)), ("triangulate (chain form)", seed, done, n, mode)
        kept, err = o.reprojection_filter(ref, p1, p2, c1, c2m, ids, 4.0)
        assert int(rn.cpu().numpy()[0]) == len(kept) and np.array_equal(ridx.cpu().numpy()[0, :len(kept)], kept), ("filter", seed, done, n, mode)
        e = float(rerr.cpu().numpy()[0])
        assert e == err or (np.isnan(e) and np.isnan(err)), ("filter error sum", seed, done, e, err)
        done += 1
    return done


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import Oracle
    from vslam_amd import Context
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    print(f"fuzz_pose: seed {seed}: {run(Context(), Oracle(), seed, seconds=seconds)} cases identical to the oracle")

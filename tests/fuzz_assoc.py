"""Randomised parity of map association (src/vslam.cpp:129-161 + orb_distance, src/PointMap.cpp:36-46): the reference walks
the map points one at a time, the device settles them in parallel rounds, and the two must give every map point the same
keypoint.  Cases mix keypoint layouts (random, one-pixel lattices where every map point sees 9-13 keypoints, clusters), map
sizes from 0 to several thousand per item, descriptor regimes (all acceptable / mostly rejected / a mix), 0-3 stored
observations per map point, keypoints already assigned before the call, and items of different sizes in one batch.

`python tests/fuzz_assoc.py SEED SECONDS` runs it for a wall-clock budget; tests/test_gpu_fuzz.py runs a fixed slice."""
import sys
import time

import numpy as np
import torch


def _item(o, rng, w, h):
    layout = int(rng.integers(0, 3))
    if layout == 0:
        n = int(rng.integers(1, 2500))
        kp = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1)
    elif layout == 1:
        sx, sy = int(rng.integers(5, 50)), int(rng.integers(5, 50))
        x0, y0 = int(rng.integers(0, w - sx)), int(rng.integers(0, h - sy))
        gx, gy = np.meshgrid(np.arange(x0, x0 + sx), np.arange(y0, y0 + sy))
        kp = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64)
    else:
        c = np.stack([rng.uniform(10, w - 10, 12), rng.uniform(10, h - 10, 12)], 1)
        kp = c[rng.integers(0, 12, 600)] + rng.normal(0, 4, (600, 2))
        kp = kp[(kp[:, 0] >= 0) & (kp[:, 0] < w - 1) & (kp[:, 1] >= 0) & (kp[:, 1] < h - 1)]
    kp = np.unique(np.rint(kp), axis=0).astype(np.float32)
    rng.shuffle(kp)
    n_kp = len(kp)
    regime = int(rng.integers(0, 3))
    proto = rng.integers(0, 256, 32, dtype=np.uint8)
    if regime == 0:                                                        # everything within a few bits of one descriptor
        desc = np.repeat(proto[None], n_kp, 0)
        desc[:, int(rng.integers(0, 32))] ^= rng.integers(0, 256, n_kp, dtype=np.uint8)
    else:
        desc = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
    n_map = int(rng.choice([0, 1, 5, 63, 64, 65, 255, 256, 257, 700, 1500, 4000]))
    src = rng.integers(0, n_kp, max(n_map, 1))[:n_map]
    px = kp[src].astype(np.float64) + rng.uniform(-1.9, 1.9, (n_map, 2))
    px[rng.random(n_map) < 0.05] += 4000
    z = rng.uniform(1.5, 9, n_map)
    mp = np.stack([(px[:, 0] - w // 2) / 525.0 * z, (px[:, 1] - h // 2) / 525.0 * z, z, np.ones(n_map)], 1).astype(np.float32)
    n_obs = rng.integers(0, 4, n_map) if regime != 0 else np.ones(n_map, np.int64)
    offs = np.zeros(n_map + 1, np.int32); offs[1:] = np.cumsum(n_obs)
    od = rng.integers(0, 256, (max(int(offs[-1]), 1), 32), dtype=np.uint8)
    for i in range(n_map):
        for ob in range(offs[i], offs[i + 1]):
            if regime == 0:
                od[ob] = proto
            elif rng.random() < (0.9 if regime == 1 else 0.4):
                od[ob] = desc[src[i]] ^ np.packbits(rng.random(256) < rng.choice([0.02, 0.15, 0.3]))
    ids = np.full(n_kp, -1, np.int32)
    ids[rng.random(n_kp) < rng.choice([0.0, 0.2, 0.8])] = 3
    return dict(kp=kp, desc=desc, nodes=o.kdtree_build_frame(kp), mp=mp, offs=offs, od=od, ids=ids)


def run(ctx, o, seed, cases=None, seconds=None):
    rng = np.random.default_rng(seed)
    t0, done = time.time(), 0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        w, h = int(rng.choice([160, 320, 640])), int(rng.choice([120, 240, 480]))
        c2 = np.array([[525, 0, w // 2, 0], [0, 525, h // 2, 0], [0, 0, 1, 0]], np.float32)
        radius = float(rng.choice([2.0, 2.0, 1.0, 2.05]))
        items = [_item(o, rng, w, h) for _ in range(int(rng.integers(1, 5)))]
        B = len(items)
        Kp = max(len(s["kp"]) for s in items) + int(rng.integers(0, 40))
        Mp = max(max(len(s["mp"]) for s in items), 1) + int(rng.integers(0, 40))
        Os = max(len(s["od"]) for s in items)
        xy = np.zeros((B, Kp, 2), np.float32); desc = np.zeros((B, Kp, 32), np.uint8); nodes = np.zeros((B, Kp), np.int32)
        n = np.zeros(B, np.int32); mp = np.zeros((B, Mp, 4), np.float32); nm = np.zeros(B, np.int32)
        offs = np.zeros((B, Mp + 1), np.int32); od = np.zeros((B, Os, 32), np.uint8); ids = np.full((B, Kp), -1, np.int32)
        for b, s in enumerate(items):
            k, m = len(s["kp"]), len(s["mp"])
            xy[b, :k], desc[b, :k], nodes[b, :k], n[b] = s["kp"], s["desc"], s["nodes"], k
            mp[b, :m], nm[b], offs[b, :m + 1], od[b, :len(s["od"])] = s["mp"], m, s["offs"], s["od"]
            offs[b, m + 1:] = s["offs"][-1]
            ids[b, :k] = s["ids"]
        d_ids = t(ids)
        claim = ctx.associate(t(mp), t(nm), t(np.repeat(c2.reshape(1, 12), B, 0)), w, h, t(nodes), t(xy), t(desc), t(n), t(offs),
                              t(od), d_ids, radius=radius)
        ctx.synchronize()
        claim, got = claim.cpu().numpy(), d_ids.cpu().numpy()
        for b, s in enumerate(items):
            k, m = len(s["kp"]), len(s["mp"])
            ref_ids, ref_claim = o.associate(s["mp"], c2, w, h, s["nodes"], s["kp"], s["desc"], s["offs"], s["od"], s["ids"],
                                             radius=radius)
            tag = (seed, done, b, w, h, k, m, radius)
            assert np.array_equal(claim[b, :m], ref_claim), ("claims",) + tag
            assert np.array_equal(got[b, :k], ref_ids), ("map_point_ids",) + tag
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
    n = run(Context(), Oracle(), seed, seconds=seconds)
    print(f"fuzz_assoc: seed {seed}: {n} batches identical to the oracle")

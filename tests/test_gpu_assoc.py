"""Map association (src/vslam.cpp:129-161 + orb_distance) on the device vs the oracle: the claimed
keypoint of every map point and the updated map_point_ids, bit-exact, including contended keypoints."""
import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu


def _scenario(oracle, seed, w, h, n_kp, n_map):
    rng = np.random.default_rng(seed)
    kp = np.unique(np.rint(np.stack([rng.uniform(0, w - 1, n_kp), rng.uniform(0, h - 1, n_kp)], 1)), axis=0).astype(np.float32)
    rng.shuffle(kp)
    n_kp = len(kp)
    desc = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
    nodes = oracle.kdtree_build_frame(kp)
    K = np.array([[525, 0, w // 2], [0, 525, h // 2], [0, 0, 1]], np.float64)
    ang = np.deg2rad(1.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.05, -0.02, 0.1])
    c2 = (K @ np.c_[R, t]).astype(np.float32)
    # map points: back-project keypoints (some several times -> contention), some random / out of view
    src = rng.integers(0, n_kp, n_map)
    src[: n_map // 6] = src[n_map // 6: 2 * (n_map // 6)]                 # duplicates claim the same keypoint
    depth = rng.uniform(2, 10, n_map)
    pix = kp[src].astype(np.float64) + rng.uniform(-1.6, 1.6, (n_map, 2))
    pix[rng.random(n_map) < 0.1] += 5000                                  # projects outside the image
    cam = np.linalg.inv(K) @ np.c_[pix, np.ones(n_map)].T * depth
    X = (R.T @ (cam - t[:, None])).T
    mp = np.c_[X, np.ones(n_map)].astype(np.float32)
    n_obs = rng.integers(0, 4, n_map)                                     # 0 observations -> distance u32_max
    offs = np.zeros(n_map + 1, np.int32); offs[1:] = np.cumsum(n_obs)
    od = rng.integers(0, 256, (max(int(offs[-1]), 1), 32), dtype=np.uint8)
    for i in range(n_map):
        for o in range(offs[i], offs[i + 1]):
            if rng.random() < 0.8:                                        # near-copy of the keypoint's descriptor
                flips = rng.random(256) < rng.choice([0.02, 0.1, 0.3])
                od[o] = desc[src[i]] ^ np.packbits(flips)
    ids = np.full(n_kp, -1, np.int32)
    ids[rng.integers(0, n_kp, n_kp // 5)] = rng.integers(0, 50, n_kp // 5)   # already propagated by matching
    return dict(kp=kp, desc=desc, nodes=nodes, c2=c2, mp=mp, offs=offs, od=od, ids=ids)


def test_association_bit_exact(ctx, oracle):
    w, h = 640, 480
    items = [_scenario(oracle, 1, w, h, 1500, 3000), _scenario(oracle, 2, w, h, 400, 90), _scenario(oracle, 3, w, h, 2000, 10)]
    B = len(items)
    Kp = max(len(s["kp"]) for s in items); Mp = max(len(s["mp"]) for s in items); Os = max(len(s["od"]) for s in items)
    xy = np.zeros((B, Kp, 2), np.float32); desc = np.zeros((B, Kp, 32), np.uint8); nodes = np.zeros((B, Kp), np.int32)
    n = np.zeros(B, np.int32); mp = np.zeros((B, Mp, 4), np.float32); nm = np.zeros(B, np.int32)
    offs = np.zeros((B, Mp + 1), np.int32); od = np.zeros((B, Os, 32), np.uint8); ids = np.full((B, Kp), -1, np.int32)
    c2 = np.zeros((B, 12), np.float32)
    for b, s in enumerate(items):
        k, m = len(s["kp"]), len(s["mp"])
        xy[b, :k], desc[b, :k], nodes[b, :k], n[b] = s["kp"], s["desc"], s["nodes"], k
        mp[b, :m], nm[b], offs[b, :m + 1], od[b, :len(s["od"])] = s["mp"], m, s["offs"], s["od"]
        ids[b, :k], c2[b] = s["ids"], s["c2"].reshape(12)
    t = lambda a: torch.from_numpy(a).cuda()
    d_ids = t(ids)
    claim = ctx.associate(t(mp), t(nm), t(c2), w, h, t(nodes), t(xy), t(desc), t(n), t(offs), t(od), d_ids)
    ctx.synchronize()
    claim, got_ids = claim.cpu().numpy(), d_ids.cpu().numpy()
    total = 0
    for b, s in enumerate(items):
        k, m = len(s["kp"]), len(s["mp"])
        ref_ids, ref_claim = oracle.associate(s["mp"], s["c2"], w, h, s["nodes"], s["kp"], s["desc"], s["offs"], s["od"], s["ids"])
        assert np.array_equal(claim[b, :m], ref_claim), b
        assert np.array_equal(got_ids[b, :k], ref_ids), b
        total += int((ref_claim >= 0).sum())
    assert total > 300, "scenario should produce associations"


def test_association_edge_cases(ctx, oracle):
    """No map points; map points without observations (distance u32_max: never accepted); everything out of view; every map
    point projecting onto the same keypoint (the lowest map index claims it, the others find it taken)."""
    w, h = 320, 240
    rng = np.random.default_rng(9)
    kp = np.unique(np.rint(np.stack([rng.uniform(0, w - 1, 300), rng.uniform(0, h - 1, 300)], 1)), axis=0).astype(np.float32)
    rng.shuffle(kp)
    n_kp = len(kp)
    desc = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
    nodes = oracle.kdtree_build_frame(kp)
    c2 = np.array([[525, 0, w // 2, 0], [0, 525, h // 2, 0], [0, 0, 1, 0]], np.float32)       # K [I | 0]
    Kinv = lambda px, z: np.array([(px[0] - w // 2) / 525.0 * z, (px[1] - h // 2) / 525.0 * z, z, 1.0], np.float32)
    same = np.stack([Kinv(kp[5] + np.array([0.25, -0.5], np.float32), 3.0 + 0.1 * i) for i in range(6)])
    cases = {
        "empty map": (np.zeros((0, 4), np.float32), np.zeros(1, np.int32), np.zeros((1, 32), np.uint8)),
        "no observations": (np.stack([Kinv(kp[i], 4.0) for i in range(8)]), np.zeros(9, np.int32), np.zeros((1, 32), np.uint8)),
        "out of view": (np.stack([Kinv(kp[i] + 5000, 4.0) for i in range(8)]), np.arange(9, dtype=np.int32), np.repeat(desc[:8], 1, 0)),
        "behind one keypoint": (same, np.arange(7, dtype=np.int32), np.repeat(desc[5:6], 6, 0)),
    }
    for name, (mp, offs, od) in cases.items():
        ids = np.full(n_kp, -1, np.int32)
        B, Mp, Os = 1, max(len(mp), 1), len(od)
        mpb = np.zeros((B, Mp, 4), np.float32); mpb[0, :len(mp)] = mp
        offb = np.zeros((B, Mp + 1), np.int32); offb[0, :len(offs)] = offs; offb[0, len(offs):] = offs[-1]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d_ids = t(ids[None])
        claim = ctx.associate(t(mpb), t(np.array([len(mp)], np.int32)), t(c2.reshape(1, 12)), w, h, t(nodes[None]), t(kp[None]), t(desc[None]),
                              t(np.array([n_kp], np.int32)), t(offb), t(od[None]), d_ids)
        ctx.synchronize()
        ref_ids, ref_claim = oracle.associate(mp, c2, w, h, nodes, kp, desc, offs, od, ids)
        assert np.array_equal(claim.cpu().numpy()[0, :len(mp)], ref_claim), name
        assert np.array_equal(d_ids.cpu().numpy()[0], ref_ids), name
        if name == "behind one keypoint":
            assert ref_claim[0] == 5 and (ref_claim[1:] == -1).all()
        elif name != "empty map":
            assert (ref_claim == -1).all(), name


@pytest.mark.parametrize("radius,fill", [(2.0, 0.0), (2.05, 0.3), (2.05, 0.7)])
def test_association_long_claim_chains(ctx, oracle, radius, fill):
    """Keypoints on a one-pixel lattice, every descriptor acceptable, several map points per keypoint: each map point lists
    9-13 hits and most of them are contended, so the claims cascade (a map point loses its first hits to lower indices and
    takes a later one, which a higher index then finds taken). The assignment must still be the sequential loop's."""
    w, h = 160, 120
    rng = np.random.default_rng(int(radius * 100) + int(fill * 100))
    gx, gy = np.meshgrid(np.arange(20, 60), np.arange(30, 70))
    kp = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    rng.shuffle(kp)
    n_kp = len(kp)
    proto = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.repeat(proto[None], n_kp, 0)
    desc[:, 0] ^= rng.integers(0, 256, n_kp, dtype=np.uint8)              # <= 8 bits from the prototype
    nodes = oracle.kdtree_build_frame(kp)
    c2 = np.array([[525, 0, w // 2, 0], [0, 525, h // 2, 0], [0, 0, 1, 0]], np.float32)
    items = []
    for n_map in (4000, 700, 64, 1):
        px = np.stack([rng.uniform(18, 62, n_map), rng.uniform(28, 72, n_map)], 1)
        z = rng.uniform(2, 6, n_map)
        mp = np.stack([(px[:, 0] - w // 2) / 525.0 * z, (px[:, 1] - h // 2) / 525.0 * z, z, np.ones(n_map)], 1).astype(np.float32)
        offs = np.arange(n_map + 1, dtype=np.int32)
        od = np.repeat(proto[None], n_map, 0)
        ids = np.full(n_kp, -1, np.int32)
        ids[rng.random(n_kp) < fill] = 7
        items.append((mp, offs, od, ids))
    B = len(items); Mp = max(len(i[0]) for i in items)
    mpb = np.zeros((B, Mp, 4), np.float32); offb = np.zeros((B, Mp + 1), np.int32); odb = np.zeros((B, Mp, 32), np.uint8)
    idb = np.full((B, n_kp), -1, np.int32); nm = np.zeros(B, np.int32)
    for b, (mp, offs, od, ids) in enumerate(items):
        m = len(mp)
        mpb[b, :m], offb[b, :m + 1], odb[b, :m], idb[b], nm[b] = mp, offs, od, ids, m
        offb[b, m + 1:] = offs[-1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rep = lambda a: t(np.repeat(a[None], B, 0))
    d_ids = t(idb)
    claim = ctx.associate(t(mpb), t(nm), rep(c2.reshape(12)), w, h, rep(nodes), rep(kp), rep(desc), t(np.full(B, n_kp, np.int32)),
                          t(offb), t(odb), d_ids, radius=radius)
    ctx.synchronize()
    claim, got = claim.cpu().numpy(), d_ids.cpu().numpy()
    later = 0
    for b, (mp, offs, od, ids) in enumerate(items):
        ref_ids, ref_claim = oracle.associate(mp, c2, w, h, nodes, kp, desc, offs, od, ids, radius=radius)
        assert np.array_equal(claim[b, :len(mp)], ref_claim), b
        assert np.array_equal(got[b], ref_ids), b
        later += int((ref_claim >= 0).sum())
    free = int((items[0][3] < 0).sum())
    assert (claim[0, :4000] >= 0).sum() > 0.9 * free, "the big item should use up nearly every free keypoint"

"""HIP k-d tree vs the oracle: node arrays and radius hit lists, bit-exact and in order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _points(seed, n, w, h, integer=True):
    rng = np.random.default_rng(seed)
    p = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1)
    return (np.rint(p) if integer else p).astype(np.float32)


def test_build_bit_exact_with_ties(ctx, oracle):
    K = 2100
    cases = [_points(1, 2000, 1280, 720), _points(2, 2100, 100, 100), _points(3, 1, 10, 10), _points(4, 2, 10, 10),
             _points(5, 3, 10, 10), _points(6, 777, 1280, 720, integer=False), np.zeros((0, 2), np.float32),
             np.full((300, 2), 5.0, np.float32), _points(7, 1500, 8, 2000)]
    xy = np.zeros((len(cases), K, 2), np.float32); n = np.zeros(len(cases), np.int32)
    for b, c in enumerate(cases):
        xy[b, :len(c)] = c; n[b] = len(c)
    nodes = ctx.kdtree_build(torch.from_numpy(xy).cuda(), torch.from_numpy(n).cuda()).cpu().numpy()
    for b, c in enumerate(cases):
        ref = oracle.kdtree_build_frame(c)
        assert np.array_equal(nodes[b, :len(c)], ref), f"case {b}"


def test_radius_search_order_and_counts(ctx, oracle):
    K, Q = 2000, 300
    cases = [(_points(11, 2000, 1280, 720), 2.0), (_points(12, 1800, 100, 100), 2.0), (_points(13, 900, 300, 300), 7.5),
             (_points(14, 5, 50, 50), 60.0)]
    B = len(cases)
    xy = np.zeros((B, K, 2), np.float32); n = np.zeros(B, np.int32)
    qs = np.zeros((B, Q, 2), np.float32); nq = np.full(B, Q, np.int32)
    for b, (c, r) in enumerate(cases):
        xy[b, :len(c)] = c; n[b] = len(c)
        rng = np.random.default_rng(50 + b)
        base = c[rng.integers(0, len(c), Q)] + rng.uniform(-2.5, 2.5, size=(Q, 2))
        qs[b] = base.astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    nodes = ctx.kdtree_build(t(xy), t(n))
    for r in sorted({c[1] for c in cases}):
        hits, counts = ctx.kdtree_radius(nodes, t(xy), t(n), t(qs), t(nq), r, hit_cap=64)
        hits, counts = hits.cpu().numpy(), counts.cpu().numpy()
        for b, (c, rr) in enumerate(cases):
            if rr != r:
                continue
            ref_nodes = oracle.kdtree_build_frame(c)
            for q in range(Q):
                ref, cnt = oracle.kdtree_radius_frame(ref_nodes, c, qs[b, q], r, cap=64)
                assert counts[b, q] == cnt, (b, q)
                assert np.array_equal(hits[b, q, :min(cnt, 64)], ref), (b, q)


def test_radius_property_full_size(ctx):
    """4000 points, 4000 queries: the hit SET equals brute force (d^2 < r^2 with the |split| <= r
    pruning never losing a point) — the reference test's own acceptance rule (test_kdtree.cpp:119-129)."""
    K = 4000
    c = _points(21, K, 1920, 1080)
    xy = torch.from_numpy(c[None]).cuda(); n = torch.tensor([K], dtype=torch.int32).cuda()
    nodes = ctx.kdtree_build(xy, n)
    assert sorted(nodes[0].cpu().tolist()) == list(range(K))      # a permutation of the points
    rng = np.random.default_rng(22)
    qs = (c + rng.uniform(-3, 3, size=c.shape)).astype(np.float32)
    hits, counts = ctx.kdtree_radius(nodes, xy, n, torch.from_numpy(qs[None]).cuda(), n, 4.0, hit_cap=32)
    hits, counts = hits[0].cpu().numpy(), counts[0].cpu().numpy()
    d2 = ((qs[:, None, 0] - c[None, :, 0]) ** 2 + (qs[:, None, 1] - c[None, :, 1]) ** 2)
    for q in range(0, K, 7):
        want = set(np.nonzero(d2[q] < np.float32(16.0))[0].tolist())
        assert counts[q] == len(want) and set(hits[q, :counts[q]].tolist()) == want

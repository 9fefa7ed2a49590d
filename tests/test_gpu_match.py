"""HIP match kernel vs the oracle (bit-exact: indices, distances, pair lists)."""
import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu


def _pack(items, K):
    B = len(items)
    d = np.zeros((B, K, 32), dtype=np.uint8)
    n = np.zeros(B, dtype=np.int32)
    for b, a in enumerate(items):
        d[b, :len(a)] = a
        n[b] = len(a)
    return torch.from_numpy(d).cuda(), torch.from_numpy(n).cuda()


@pytest.fixture(params=[(0, 0), (1, 1), (2, 1), (1, 2), (2, 2)], ids=["product", "exp-fp4-8x32", "exp-fp4-4x64", "exp-i8-8x32", "exp-i8-4x64"])
def mctx(request):
    """The product library's one matcher (FP4 +-1 products, 8 waves x 32 rows), and -- through the EXPERIMENTS build, the only
    one that carries them -- both workgroup shapes (VSLAM_OPT_MATCH_SHAPE) of both matrix-core forms (VSLAM_OPT_MATCH_FORM:
    FP4, int8 0 / 1 products).  Yields the context to use."""
    if request.param == (0, 0):
        yield request.getfixturevalue("ctx")
        return
    c = request.getfixturevalue("ctx_exp")
    c.set_option(c.OPT_MATCH_SHAPE, request.param[0])
    c.set_option(c.OPT_MATCH_FORM, request.param[1])
    yield c
    c.set_option(c.OPT_MATCH_SHAPE, 0)
    c.set_option(c.OPT_MATCH_FORM, 0)


def test_product_library_carries_one_matcher(ctx):
    """The matcher's variants are settable in the experiments build only."""
    from vslam_amd import VslamError
    for opt, val in ((ctx.OPT_MATCH_SHAPE, 1), (ctx.OPT_MATCH_SHAPE, 2), (ctx.OPT_MATCH_FORM, 2)):
        with pytest.raises(VslamError):
            ctx.set_option(opt, val)
    ctx.set_option(ctx.OPT_MATCH_FORM, 1)   # FP4 is the product's form
    ctx.set_option(ctx.OPT_MATCH_FORM, 0)


def test_knn2_and_ratio_bit_exact_ragged_batch(mctx, oracle):
    ctx = mctx
    K = 700
    sizes = [(500, 500), (700, 650), (1, 2), (513, 257), (256, 512), (0, 10), (10, 1), (10, 0), (3, 2)]
    items = [synth.descriptors_pair(100 + i, a, b) for i, (a, b) in enumerate(sizes)]
    for it in items[:2]:
        if len(it[1]) > 50:
            it[1][7] = it[1][33]      # equal train rows -> distance ties, lower index must win
    d1, n1 = _pack([it[0] for it in items], K)
    d2, n2 = _pack([it[1] for it in items], K)
    pairs, m, knn = ctx.match_knn2_ratio(d1, n1, d2, n2, want_knn=True)
    ctx.synchronize()
    pairs, m, knn = pairs.cpu().numpy(), m.cpu().numpy(), knn.cpu().numpy()
    for b, (a, t, _) in enumerate(items):
        if len(t) >= 2:
            i0, e0, i1, e1 = oracle.match_knn2(a, t)
            g = knn[b, :len(a)]
            assert np.array_equal(g[:, 0], i0) and np.array_equal(g[:, 1], e0), b
            assert np.array_equal(g[:, 2], i1) and np.array_equal(g[:, 3], e1), b
            ref, rc = oracle.match_knn2_ratio(a, t)
            assert rc == 0
            assert m[b] == len(ref), b
            assert np.array_equal(pairs[b, :m[b]], ref), b
        else:
            assert m[b] == 0      # reference reads m[1] of a 1-row result: undefined; we emit nothing


def test_full_size_property_self_match(mctx):
    ctx = mctx
    """At the headline size (K = 2000, B = 8 of the 256) every row's best match against a
    shuffled copy of itself is its own image at distance 0, and a copy is its own 2nd-NN-ratio
    survivor: checks index packing over the whole range without the oracle."""
    B, K = 8, 2000
    g = torch.Generator().manual_seed(1)
    d1 = torch.randint(0, 256, (B, K, 32), dtype=torch.uint8, generator=g)
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])
    d2 = torch.stack([d1[b][perm[b]] for b in range(B)])
    n = torch.full((B,), K, dtype=torch.int32)
    pairs, m, knn = ctx.match_knn2_ratio(d1.cuda(), n.cuda(), d2.cuda(), n.cuda(), want_knn=True)
    ctx.synchronize()
    knn, m, pairs = knn.cpu(), m.cpu(), pairs.cpu()
    inv = torch.argsort(perm, dim=1).to(torch.int32)
    assert torch.equal(knn[:, :, 0], inv) and int(knn[:, :, 1].abs().sum()) == 0
    assert torch.all(m == K)
    assert torch.equal(pairs[:, :, 1], inv)


def test_extreme_distances_and_index_range(mctx, oracle):
    ctx = mctx
    """Distances 0, 1, 255 and 256 (the +-1 products of the FP4 form then sum to +-256, where the key arithmetic changes
    sign), all-zero and all-one descriptors, and train indices up to the largest a key can carry at this stride."""
    rng = np.random.default_rng(77)
    K = 2100
    q = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    q[0] = 0; q[1] = 255; q[2] = 0; q[3] = 255
    t = rng.integers(0, 256, (K, 32), dtype=np.uint8)
    t[5] = ~q[10]                         # distance 256 from query 10
    t[K - 1] = q[11]                      # distance 0 at the last index
    t[K - 2] = q[11]; t[K - 2, 0] ^= 1    # distance 1 right before it
    t[100] = 255; t[101] = 0              # |b| = 256 and 0
    far = np.stack([~q[12]] * 40)         # a query whose two best are both far: every train row at distance >= 250
    far[:, 0] ^= np.arange(40, dtype=np.uint8) % 7
    d1, n1 = _pack([q, q[12:13]], K)
    d2, n2 = _pack([t, far], K)
    pairs, m, knn = ctx.match_knn2_ratio(d1, n1, d2, n2, want_knn=True)
    ctx.synchronize()
    knn, m, pairs = knn.cpu().numpy(), m.cpu().numpy(), pairs.cpu().numpy()
    for b, (a, tr) in enumerate(((q, t), (q[12:13], far))):
        i0, e0, i1, e1 = oracle.match_knn2(a, tr)
        g = knn[b, :len(a)]
        assert np.array_equal(g[:, 0], i0) and np.array_equal(g[:, 1], e0), b
        assert np.array_equal(g[:, 2], i1) and np.array_equal(g[:, 3], e1), b
        ref, _ = oracle.match_knn2_ratio(a, tr)
        assert m[b] == len(ref) and np.array_equal(pairs[b, :m[b]], ref), b
    assert knn[0, 11, 0] == K - 1 and knn[0, 11, 1] == 0 and knn[0, 11, 2] == K - 2 and knn[0, 11, 3] == 1
    assert knn[1, 0, 1] >= 250

"""HIP RANSAC kernels vs the oracle: sets, per-hypothesis F / count / sum, winner, mask, F —
all bit-exact (the float outputs are compared as uint32 bit patterns)."""
import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(params=["ties", "all"])
def sums_mode(ctx, request):
    """Both scoring paths: the default one (exact counts everywhere, exact sums for the maximum-count hypotheses only:
    ransac_count / ties / tiesum kernels) and VSLAM_OPT_RANSAC_ALL_SUMS (ransac_score_kernel: every sum)."""
    ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, request.param == "all")
    yield request.param
    ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)


def _beaten(ref_count, ref_sum):
    """Maximum-count hypotheses whose float residual sum is strictly below the largest one among them: the accept rule
    (count, then the larger sum, src/RansacFilter.cpp:59) can neither pick them nor be tied by them."""
    ref_count = np.asarray(ref_count); rs = np.asarray(ref_sum, dtype=np.float32)
    top = ref_count == ref_count.max()
    finite = rs[top & ~np.isnan(rs)]
    if not finite.size:
        return np.zeros(len(rs), bool)
    return top & (rs < finite.max())


def check_counts(got_count, ref_count, mode, tag=None, ref_sum=None):
    """'all' mode: every count exact.  Default mode: every hypothesis below the pair's maximum count reads -1 -- the counting
    kernel abandons a hypothesis as soon as it can no longer reach a count already verified for the pair -- and a hypothesis
    that reaches the maximum carries it, unless (round 4, ref_sum given) it was abandoned because a certified bound put
    its residual sum below that of a verified maximum-count hypothesis: then it reads -1 as well, which is allowed exactly
    for hypotheses that are beaten on the sum."""
    got_count = np.asarray(got_count); ref_count = np.asarray(ref_count)
    if mode == "all":
        assert np.array_equal(got_count, ref_count), tag
        return
    top = ref_count == ref_count.max()
    assert (got_count[~top] == -1).all(), tag
    carried = got_count == ref_count
    if ref_sum is None:
        assert carried[top].all(), tag
    else:
        assert (carried | ((got_count == -1) & _beaten(ref_count, ref_sum)))[top].all(), tag
        assert carried[top & ~_beaten(ref_count, ref_sum)].all(), tag     # whoever can win or tie is counted in full


def check_sums(got_sum, ref_count, ref_sum, mode, tag=None, got_count=None):
    """'all' mode: every sum bit-exact.  Default mode: sums exist only where the accept rule can consult them — for
    hypotheses whose count is the pair's maximum — and of those only the ones that can still be the largest: each such
    sum is bit-exact or -inf ("pruned: certainly smaller than the winner's"), never -inf for a hypothesis whose float
    sum is >= every other tied one (that would change the winner), and NaN for every hypothesis below the maximum."""
    ref_count = np.asarray(ref_count)
    got_sum = np.asarray(got_sum); ref_sum = np.asarray(ref_sum, dtype=np.float32)
    tied = ref_count == ref_count.max()
    if mode == "all":
        assert np.array_equal(bits(got_sum), bits(ref_sum)), tag
        return
    assert np.isnan(got_sum[~tied]).all(), tag
    exact = bits(got_sum) == bits(ref_sum)
    pruned = np.isneginf(got_sum) & ~exact
    if got_count is not None:     # abandoned on the sum rule (count -1): no sum either, and only if beaten (check_counts)
        gone = tied & (np.asarray(got_count) == -1)
        assert np.isnan(got_sum[gone]).all(), tag
        pruned = pruned | gone
    assert (exact | pruned)[tied].all(), tag
    finite_ref = ref_sum[tied & ~np.isnan(ref_sum)]
    if finite_ref.size:
        top = finite_ref.max()
        assert not (pruned & tied & (ref_sum >= top)).any(), tag     # a possible winner was never pruned
    assert not (pruned & np.isnan(ref_sum)).any(), tag                # NaN sums take part in the accept rule


@pytest.mark.parametrize("H", [64, 333])
def test_sets_bit_exact(ctx, oracle, H):
    ms = [8, 9, 37, 150, 1100, 4000, 5, 0]
    seeds = [1, 0x5EED0002, 12345, 0xFFFFFFFF, 77, 2024, 3, 4]
    s = torch.tensor(np.array(seeds, dtype=np.uint32).view(np.int32)).cuda()
    m = torch.tensor(ms, dtype=torch.int32).cuda()
    got = ctx.ransac_sets(s, m, H).cpu().numpy()
    for b, (n, sd) in enumerate(zip(ms, seeds)):
        if n >= 8:
            assert np.array_equal(got[b], oracle.ransac_sets(sd, n, H)), b
        else:
            assert not got[b].any()


@pytest.mark.parametrize("mi", [0, 1, 5, 7])
def test_sets_with_fewer_than_eight_items(ctx, oracle, mi):
    """RansacFilter(min_items < 8): the reference draws min_items indices into 8-wide sets (src/RansacFilter.cpp:17,22);
    entries min_items .. 7 stay 0 and the generator advances min_items outputs per set.  Then the whole of
    find_fundamental on such sets (every hypothesis also uses match 0, :49-53)."""
    H = 96
    ms = [8, 9, 150, 4000, 5, 3]
    seeds = [11, 0x5EED0002, 12345, 0xFFFFFFFF, 77, 9]
    s = torch.tensor(np.array(seeds, dtype=np.uint32).view(np.int32)).cuda()
    m = torch.tensor(ms, dtype=torch.int32).cuda()
    ctx.set_option(ctx.OPT_RANSAC_MIN_ITEMS, mi)
    ctx.set_option(ctx.OPT_RANSAC_MIN_MATCHES, max(mi, 1))
    try:
        got = ctx.ransac_sets(s, m, H).cpu().numpy()
        for b, (n, sd) in enumerate(zip(ms, seeds)):
            if n >= mi and mi > 0:
                assert np.array_equal(got[b], oracle.ransac_sets(sd, n, H, min_items=mi)), (mi, b)
            else:
                assert not got[b].any()
            assert not got[b][:, mi:].any()
        K, W, Hh = 160, 640, 480
        sizes = [150, 40, 9] + ([6] if mi <= 5 else [])       # 6 matches: legal in the reference when min_items <= 6
        xy1, xy2, pairs, mm = _batch(900 + mi, sizes, K, W, Hh)
        sd2 = np.arange(5, 5 + len(sizes)).astype(np.uint32)
        sets = ctx.ransac_sets(torch.from_numpy(sd2.view(np.int32)).cuda(), torch.from_numpy(mm).cuda(), H)
        r = ctx.ransac_fundamental(torch.from_numpy(xy1).cuda(), torch.from_numpy(xy2).cuda(), torch.from_numpy(pairs).cuda(),
                                   torch.from_numpy(mm).cuda(), sets, 10.0)
        ctx.synchronize()
        r = {k: v.cpu().numpy() for k, v in r.items()}
        sets = sets.cpu().numpy()
        for b in range(len(sizes)):
            n = int(mm[b])
            if mi > 0:
                assert np.array_equal(sets[b], oracle.ransac_sets(int(sd2[b]), n, H, min_items=mi)), (mi, b)
            ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], 10.0)
            assert r["best"][b, 0] == ref["winner"] and r["best"][b, 1] == ref["count"], (mi, b)
            if ref["winner"] >= 0:
                assert np.array_equal(r["F"][b].view(np.uint32), ref["F"].view(np.uint32)), (mi, b)
                assert np.array_equal(r["mask"][b, :n] != 0, ref["mask"][:n] != 0), (mi, b)
    finally:
        ctx.set_option(ctx.OPT_RANSAC_MIN_ITEMS, 8)
        ctx.set_option(ctx.OPT_RANSAC_MIN_MATCHES, 8)


def _zero_shift_rejections(seed, n, H):
    """Raw outputs that Lemire's test rejects if no earlier output was rejected (exact for the first one, which is all
    that is needed to know that a stream has rejections at all)."""
    raw = np.random.RandomState(seed).randint(0, 2 ** 32, size=H * 8, dtype=np.uint64)
    rng = np.uint64(n) - (np.arange(H * 8, dtype=np.uint64) & np.uint64(7))
    return np.nonzero(((raw * rng) & np.uint64(0xFFFFFFFF)) < (np.uint64(2 ** 32) - rng) % rng)[0]


def test_sets_rejection_path(ctx, oracle):
    """Lemire rejections are ~n/2^32 per draw, and one shifts every later draw of its pair by one raw output.  The
    seeds below were picked (by the zero-shift count above) so that four streams contain at least two rejections each,
    one contains exactly one, and the others none: the parallel re-mapping passes of ransac_map_kernel all run."""
    ms = [16001, 16001, 16001, 16001, 15999, 16000, 14000, 11000]
    seeds = [1073, 1176, 1309, 1376, 901, 900, 904, 907]
    counts = [len(_zero_shift_rejections(sd, n, 8192)) for sd, n in zip(seeds, ms)]
    assert min(counts[:4]) >= 2 and counts[4] == 1 and counts[5:] == [0, 0, 0], counts
    s = torch.tensor(np.array(seeds, dtype=np.uint32).view(np.int32)).cuda()
    m = torch.tensor(ms, dtype=torch.int32).cuda()
    got = ctx.ransac_sets(s, m, 8192).cpu().numpy()
    ctx.synchronize()
    for b in range(8):
        assert np.array_equal(got[b], oracle.ransac_sets(seeds[b], ms[b], 8192)), b


def _batch(seed0, sizes, K, W, H):
    B = len(sizes)
    xy1 = np.zeros((B, K, 2), np.float32); xy2 = np.zeros((B, K, 2), np.float32)
    pairs = np.zeros((B, K, 2), np.int32); m = np.zeros(B, np.int32)
    for b, n in enumerate(sizes):
        p1, p2, _ = synth.two_view_points(seed0 + b, K, W, H, inlier_frac=0.65)
        xy1[b], xy2[b] = p1, p2
        rng = np.random.default_rng(seed0 * 7 + b)
        q = np.sort(rng.permutation(K)[:n])
        pairs[b, :n, 0] = q
        pairs[b, :n, 1] = q                       # correspondence i <-> i, as two_view_points builds it
        m[b] = n
    return xy1, xy2, pairs, m


def test_fundamental_bit_exact(ctx, oracle, sums_mode):
    K, Hy, thr = 600, 192, 10.0
    sizes = [300, 8, 9, 600, 150, 5]
    xy1, xy2, pairs, m = _batch(500, sizes, K, 1280, 720)
    seeds = np.arange(40, 40 + len(sizes)).astype(np.uint32)
    sets = np.zeros((len(sizes), Hy, 8), np.int32)
    for b, n in enumerate(sizes):
        if n >= 8:
            sets[b] = oracle.ransac_sets(int(seeds[b]), n, Hy)
    t = lambda a: torch.from_numpy(a).cuda()
    out = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(sizes):
        if n < 8:
            assert out["best"][b, 0] == -1 and out["best"][b, 3] == 0
            continue
        ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
        bad = np.nonzero((bits(out["hypF"][b]) != bits(ref["hypF"])).any(axis=1))[0]
        assert bad.size == 0, f"item {b}: {bad.size} of {Hy} hypothesis F differ, first {bad[:5]}"
        check_counts(out["hyp_count"][b], ref["hyp_count"], sums_mode, b, ref["hyp_sum"])
        check_sums(out["hyp_sum"][b], ref["hyp_count"], ref["hyp_sum"], sums_mode, b, out["hyp_count"][b])
        assert out["best"][b, 0] == ref["winner"] and out["best"][b, 1] == ref["count"], b
        assert out["best"][b, 2] == int(bits(np.float32(ref["sum"])).reshape(-1)[0]), b
        assert np.array_equal(bits(out["F"][b]), bits(ref["F"])), b
        assert np.array_equal(out["mask"][b, :n], ref["mask"]), b
        keep = pairs[b, :n][ref["mask"].astype(bool)]
        assert out["best"][b, 3] == len(keep)
        assert np.array_equal(out["matches"][b, :len(keep)], keep), b


def test_degenerate_geometry_still_bit_exact(ctx, oracle, sums_mode):
    """Collinear / repeated / zero-motion samples drive the Jacobi into its zero-singular-value
    branch (the cv::RNG fill) and the residual into 0/0 and x/0; NaN and Inf must propagate
    identically (NaN <= thr is false, src/RansacFilter.cpp:130)."""
    K, Hy, thr = 64, 128, 10.0
    xy1 = np.zeros((3, K, 2), np.float32); xy2 = np.zeros((3, K, 2), np.float32)
    xs = np.arange(K, dtype=np.float32)
    xy1[0, :, 0] = xs; xy1[0, :, 1] = 2 * xs; xy2[0] = xy1[0]               # all collinear, no motion
    xy1[1, :, 0] = 100; xy1[1, :, 1] = 50; xy2[1, :, 0] = 101; xy2[1, :, 1] = 50   # one repeated point
    rng = np.random.default_rng(3)
    xy1[2] = np.rint(rng.uniform(0, 4, size=(K, 2))); xy2[2] = xy1[2]       # tiny integer grid, identity
    pairs = np.tile(np.stack([np.arange(K), np.arange(K)], 1)[None], (3, 1, 1)).astype(np.int32)
    m = np.full(3, K, np.int32)
    sets = np.stack([oracle.ransac_sets(60 + b, K, Hy) for b in range(3)])
    t = lambda a: torch.from_numpy(a).cuda()
    out = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b in range(3):
        ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b], sets[b], thr)
        assert np.array_equal(bits(out["hypF"][b]), bits(ref["hypF"])), b
        check_counts(out["hyp_count"][b], ref["hyp_count"], sums_mode, b, ref["hyp_sum"])
        check_sums(out["hyp_sum"][b], ref["hyp_count"], ref["hyp_sum"], sums_mode, b, out["hyp_count"][b])
        assert out["best"][b, 0] == ref["winner"], b
        if ref["winner"] >= 0:
            assert np.array_equal(out["mask"][b], ref["mask"]), b


def test_match_features_pipeline_bit_exact(ctx, oracle, sums_mode):
    """match -> sets -> RANSAC -> inlier filter through vslam_match_features (src/Frame.cpp:82-105)."""
    B, K, Hy, thr = 4, 500, 256, 10.0
    n1s, n2s = [500, 420, 300, 12], [480, 500, 310, 40]
    xy1 = np.zeros((B, K, 2), np.float32); xy2 = np.zeros((B, K, 2), np.float32)
    d1 = np.zeros((B, K, 32), np.uint8); d2 = np.zeros((B, K, 32), np.uint8)
    for b in range(B):
        a, c, truth = synth.descriptors_pair(800 + b, n1s[b], n2s[b], match_frac=0.7)
        p1, p2, _ = synth.two_view_points(810 + b, n1s[b], 1280, 720, inlier_frac=0.8)
        q2 = np.rint(np.random.default_rng(820 + b).uniform(0, 700, size=(n2s[b], 2))).astype(np.float32)
        ok = truth >= 0
        q2[truth[ok]] = p2[ok]                    # matched descriptors carry the two-view geometry
        d1[b, :n1s[b]], d2[b, :n2s[b]] = a, c
        xy1[b, :n1s[b]], xy2[b, :n2s[b]] = p1, q2
    seeds = np.array([5, 6, 7, 8], np.uint32)
    t = lambda a: torch.from_numpy(a).cuda()
    out = ctx.match_features(t(xy1), t(d1), torch.tensor(n1s, dtype=torch.int32).cuda(), t(xy2), t(d2),
                             torch.tensor(n2s, dtype=torch.int32).cuda(), t(seeds.view(np.int32)), Hy, thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b in range(B):
        ref = oracle.match_features(xy1[b, :n1s[b]], d1[b, :n1s[b]], xy2[b, :n2s[b]], d2[b, :n2s[b]],
                                    int(seeds[b]), Hy, thr)
        assert out["prelim_m"][b] == ref["prelim"], b
        if ref["rc"] != 0:                         # < 8 preliminary matches: reference is undefined
            assert out["best"][b, 3] == 0
            continue
        k = len(ref["matches"])
        assert out["best"][b, 3] == k, b
        assert np.array_equal(out["matches"][b, :k], ref["matches"]), b
        assert np.array_equal(bits(out["F"][b]), bits(ref["F"])), b


def _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr):
    t = lambda a: torch.from_numpy(a).cuda()
    out = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
    ctx.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _compare(out, ref, b, n, mode):
    check_counts(out["hyp_count"][b], ref["hyp_count"], mode, b, ref["hyp_sum"])
    check_sums(out["hyp_sum"][b], ref["hyp_count"], ref["hyp_sum"], mode, b, out["hyp_count"][b])
    assert out["best"][b, 0] == ref["winner"], b
    if ref["winner"] >= 0:
        assert out["best"][b, 1] == ref["count"], b
        assert out["best"][b, 2] == int(bits(np.float32(ref["sum"])).reshape(-1)[0]), b
        assert np.array_equal(bits(out["F"][b]), bits(ref["F"])), b
        assert np.array_equal(out["mask"][b, :n], ref["mask"]), b


def test_counts_first_path_at_scale(ctx, oracle, sums_mode):
    """The shapes the counting kernel is built around: several 256-match sub-blocks per hypothesis, a partial last
    sub-block, hypothesis counts that are not a multiple of a workgroup's 128, thresholds that put many evaluations next
    to the decision boundary, and more matches than the LDS staging holds (4096: coordinates read from memory)."""
    for K, sizes, Hy, thr in ((2304, [2304, 1500, 1025, 257, 256, 64], 200, 10.0), (2304, [1800, 700], 130, 3.0),
                              (2304, [1200, 2000], 64, 40.0), (16384, [16384, 9000, 4097], 70, 10.0)):
        xy1, xy2, pairs, m = _batch(900 + Hy, sizes, K, 1280, 720)
        sets = np.stack([oracle.ransac_sets(70 + b, n, Hy) for b, n in enumerate(sizes)])
        out = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
        for b, n in enumerate(sizes):
            ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
            _compare(out, ref, b, n, sums_mode)


def test_screen_and_rank_edge_shapes(ctx, oracle, sums_mode):
    """Shapes around the edges of the ranking / screening stages in front of the counting kernel: fewer hypotheses than
    pilots (8) or candidates, hypothesis counts around a workgroup's 128, match counts around the screen's 128 and the
    256-match sub-blocks (the ranked arrays are padded to a multiple of 256)."""
    for Hy, sizes in ((1, [8, 200]), (3, [127, 128, 129]), (7, [255, 256, 257]), (9, [383, 384, 385]),
                      (127, [130, 511]), (129, [64, 512, 513]), (257, [100, 300])):
        K = 520
        xy1, xy2, pairs, m = _batch(4000 + Hy, sizes, K, 1280, 720)
        sets = np.stack([oracle.ransac_sets(11 + b, n, Hy) for b, n in enumerate(sizes)])
        out = _find(ctx, oracle, xy1, xy2, pairs, m, sets, 10.0)
        for b, n in enumerate(sizes):
            ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], 10.0)
            _compare(out, ref, b, n, sums_mode)


def test_many_tied_hypotheses(ctx, oracle, sums_mode):
    """Identical frames on an integer grid: a large share of the hypotheses reaches the same (maximal) count, so the
    winner is decided by the sums."""
    K, Hy, thr = 400, 512, 10.0
    rng = np.random.default_rng(17)
    xy1 = np.zeros((2, K, 2), np.float32); xy2 = np.zeros((2, K, 2), np.float32)
    xy1[0] = np.rint(rng.uniform(0, 640, (K, 2))); xy2[0] = xy1[0]
    xy1[1] = np.rint(rng.uniform(0, 64, (K, 2))); xy2[1] = xy1[1] + np.float32(1.0)
    pairs = np.tile(np.stack([np.arange(K), np.arange(K)], 1)[None], (2, 1, 1)).astype(np.int32)
    m = np.array([K, 300], np.int32)
    sets = np.stack([oracle.ransac_sets(31 + b, int(m[b]), Hy) for b in range(2)])
    out = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
    tied = []
    for b in range(2):
        n = int(m[b])
        ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
        tied.append(int((ref["hyp_count"] == ref["hyp_count"].max()).sum()))
        _compare(out, ref, b, n, sums_mode)
    assert max(tied) >= 192, tied      # the top really is crowded


def test_sum_rule_on_a_plateau_of_tied_hypotheses(ctx, oracle):
    """One dominant motion plus a block that moves differently (the bench data's structure): every hypothesis drawn from
    the dominant motion alone has the same inlier set, so hundreds tie at the maximum count and the accept rule picks
    the largest residual sum among them.  The counting kernel abandons a tied hypothesis once a certified upper bound of its
    sum lies below the sum of a candidate it counted exactly (ransac_cand_kernel's floor): winner, mask and F must be the
    oracle's all the same, the abandoned ones must all be beaten on the sum (check_counts), and the rule must actually
    have fired here."""
    K, Hy, thr = 1500, 512, 10.0
    B = 3
    xy1 = np.zeros((B, K, 2), np.float32); xy2 = np.zeros((B, K, 2), np.float32)
    for b in range(B):   # an exact two-view geometry (no noise, sub-pixel coordinates) for 85 %, uniform outliers for the rest
        xy1[b], xy2[b], _ = synth.two_view_points(4100 + b, K, 1280, 720, inlier_frac=0.85, noise_px=0.0, integer=False)
    pairs = np.tile(np.stack([np.arange(K), np.arange(K)], 1)[None], (B, 1, 1)).astype(np.int32)
    m = np.array([K, K - 100, 1100], np.int32)
    sets = np.stack([oracle.ransac_sets(900 + b, int(m[b]), Hy) for b in range(B)])
    out = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
    fired = 0
    crowded = 0
    for b in range(B):
        n = int(m[b])
        ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
        _compare(out, ref, b, n, "ties")
        tied = ref["hyp_count"] == ref["hyp_count"].max()
        crowded = max(crowded, int(tied.sum()))
        fired += int((tied & (out["hyp_count"][b] == -1)).sum())
    assert crowded >= 40, crowded     # the top really is a plateau
    assert fired > 0                  # and some of it was abandoned on the sum


def test_threshold_and_scale_outside_certified_range(ctx, oracle, sums_mode):
    """Thresholds / coordinates beyond the range the cheap evaluation's bounds are derived for (thr in [2^-20, 2^20],
    |coordinates| <= 2^20): every evaluation must take the exact sequence and still give the reference's counts."""
    K, Hy = 300, 96
    for scale, thr in ((1.0, 1e-7), (1.0, 3e6), (4e6, 10.0), (1e-9, 10.0), (1.0, -1.0), (1.0, float("inf"))):
        xy1, xy2, pairs, m = _batch(1200, [300, 77], K, 1280, 720)
        xy1 = (xy1 * np.float32(scale)).astype(np.float32); xy2 = (xy2 * np.float32(scale)).astype(np.float32)
        sets = np.stack([oracle.ransac_sets(5 + b, int(m[b]), Hy) for b in range(2)])
        out = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
        for b in range(2):
            n = int(m[b])
            ref = oracle.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
            _compare(out, ref, b, n, sums_mode)


def test_tiny_denominators_take_the_exact_path(ctx, oracle, sums_mode):
    """Hypotheses whose first row is about 1e-20 and smaller: the residual's denominator Fx1[0]^2 is a float denormal
    or zero, where v_rcp_f32 is not a reciprocal.  The counting kernel's cheap evaluation certifies nothing there and
    must hand every such evaluation to the exact sequence (the guard compares with 2^-120, "unordered or less than")."""
    K, Hy, thr = 700, 160, 10.0
    sizes = [700, 300]
    xy1, xy2, pairs, m = _batch(7100, sizes, K, 1280, 720)
    sets = np.stack([oracle.ransac_sets(3 + b, n, Hy) for b, n in enumerate(sizes)])
    t = lambda a: torch.from_numpy(a).cuda()
    solved = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
    ctx.synchronize()
    hypF = solved["hypF"].cpu().numpy().copy()
    rng = np.random.default_rng(5)
    for b in range(2):
        for h in range(Hy):
            kind = h % 4
            if h % 16 == 0:    # the whole matrix tiny: n^2 and Fx1[0]^2 both denormal, their quotient an ordinary number
                hypF[b, h] *= np.float32(10.0 ** -rng.uniform(16.5, 19.5))
            elif kind == 1:    # denormal squares
                hypF[b, h, 0:3] *= np.float32(10.0 ** -rng.uniform(18, 21))
            elif kind == 2:    # squares that underflow to zero, and an exactly zero row
                hypF[b, h, 0:3] = 0 if h % 8 == 2 else hypF[b, h, 0:3] * np.float32(1e-30)
            elif kind == 3:    # one tiny coefficient only: tiny for some matches, not for others
                hypF[b, h, 2] = np.float32(-(hypF[b, h, 0] * 640 + hypF[b, h, 1] * 360)) + np.float32(1e-22)
    out = ctx.ransac_evaluate(t(xy1), t(xy2), t(pairs), t(m), t(hypF), thr)
    ctx.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(sizes):
        counts = np.zeros(Hy, np.int32); sums = np.zeros(Hy, np.float32)
        best, best_sum, winner = 0, np.float32(0.0), -1
        with np.errstate(all="ignore"):
            for h in range(Hy):
                _, c, s_ = oracle.residual(xy1[b], xy2[b], pairs[b, :n], hypF[b, h], thr)
                counts[h], sums[h] = c, s_
                if c > best or (c == best and s_ > best_sum):      # src/RansacFilter.cpp:59
                    best, best_sum, winner = c, s_, h
        check_counts(out["hyp_count"][b], counts, sums_mode, b, sums)
        check_sums(out["hyp_sum"][b], counts, sums, sums_mode, b, out["hyp_count"][b])
        assert out["best"][b, 0] == winner, (b, out["best"][b], winner, best)
        if winner >= 0:
            assert out["best"][b, 1] == best, b
            mask, _, _ = oracle.residual(xy1[b], xy2[b], pairs[b, :n], hypF[b, winner], thr)
            assert np.array_equal(out["mask"][b, :n], mask), b


def test_mfma_solver_is_opt_in_and_agrees_within_its_stated_tolerance(ctx, oracle):
    """VSLAM_OPT_RANSAC_SOLVER = 1 (BASELINE.json configs[4]: the 8-point solve as an MFMA contraction) is an
    approximate solver and says so: it is off unless asked for, and what it promises is
      * per hypothesis: F equal to the exact solver's up to sign, relative Frobenius error <= 1e-3 for at least 85 % of
        the hypotheses (median <= 2e-5; measured: 92 % and 3.5e-7).  The remainder are ill-conditioned 8-point samples
        — un-normalised pixel coordinates give the design matrix a condition number around 1e6, so where the two
        smallest singular values are close the reference's own float Jacobi is no more "right" than this solver;
      * per hypothesis: where F agrees to 1e-4 the inlier counts agree to within 2 % of the matches for at least 95 %
        of those hypotheses (a count changes only through matches whose residual sits at the threshold);
      * per pair: the best inlier count found is at least 90 % of the exact path's (which hypothesis wins may differ:
        the accept rule is a tie-break over near-equal counts, so the mask agreement is reported, not promised).
    The default path is untouched by the option (re-checked bit for bit after switching it off)."""
    K, Hy, thr = 1600, 512, 10.0
    sizes = [1500, 800, 300]
    xy1, xy2, pairs, m = _batch(4100, sizes, K, 1920, 1080)
    sets = np.stack([oracle.ransac_sets(90 + b, n, Hy) for b, n in enumerate(sizes)])
    ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)     # every per-hypothesis count, for the comparison below
    try:
        exact = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
        ctx.set_option(ctx.OPT_RANSAC_SOLVER, 1)
        try:
            approx = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
        finally:
            ctx.set_option(ctx.OPT_RANSAC_SOLVER, 0)
        again = _find(ctx, oracle, xy1, xy2, pairs, m, sets, thr)
    finally:
        ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)
    assert np.array_equal(bits(again["hypF"]), bits(exact["hypF"])) and np.array_equal(again["mask"], exact["mask"])
    for b, n in enumerate(sizes):
        Fe, Fa = exact["hypF"][b].astype(np.float64), approx["hypF"][b].astype(np.float64)
        sign = np.sign((Fe * Fa).sum(1, keepdims=True))
        err = np.linalg.norm(Fa * sign - Fe, axis=1) / np.linalg.norm(Fe, axis=1)
        ok = np.isfinite(err)
        frac = float((err[ok] <= 1e-3).mean())
        print(f"pair {b}: median rel. error {np.median(err[ok]):.2e}, within 1e-3: {100 * frac:.1f} %, "
              f"winner inliers exact {exact['best'][b, 1]} approx {approx['best'][b, 1]}, "
              f"mask agreement {100 * (exact['mask'][b, :n] == approx['mask'][b, :n]).mean():.2f} %")
        assert ok.mean() > 0.99 and np.median(err[ok]) <= 2e-5 and frac >= 0.85, (b, np.median(err[ok]), frac)
        close = ok & (err <= 1e-4)
        dc = np.abs(exact["hyp_count"][b][close].astype(np.int64) - approx["hyp_count"][b][close])
        assert (dc <= 0.02 * n).mean() >= 0.95, (b, float((dc <= 0.02 * n).mean()))
        assert int(approx["best"][b, 1]) >= 0.9 * int(exact["best"][b, 1]), b

"""The front-end on PHOTOGRAPHS (tests/golden/real_v1.npz: four public-domain photographs, second frame = the first after a
small camera motion, resampled at sub-pixel positions), bit-exact against the stored oracle outputs and, at other
settings, against the oracle run here.  Every other input of the suite is synthetic texture or noise; saturated plateaus,
smooth gradients and JPEG-like near-ties are where the certified margins of the two-tier corner detector and the from-memory
details of the OpenCV routines are most likely to bite."""
import os

import numpy as np
import pytest
import torch

from vslam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def real():
    g = np.load(os.path.join(HERE, "golden", "real_v1.npz"))
    frames = [synth.real_pair(g[f"crop{i}"], tuple(g["motions"][i])) for i in range(len(g["names"]))]
    return g, np.stack([a for a, _ in frames] + [b for _, b in frames])      # [last frames | current frames]


def test_frontend_pairs_on_photographs_equals_the_golden_outputs(ctx, real):
    g, bgr = real
    P = bgr.shape[0] // 2
    maxc, hyp, seed = (int(v) for v in g["params"])
    thr = float(g["threshold"][0])
    ca, sa = synth.keypoint_rotation()
    seeds = torch.from_numpy((np.uint32(seed) ^ np.arange(P, dtype=np.uint32)).view(np.int32)).cuda()
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, None, seeds, hyp, thr)
    ctx.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(P):
        for tag, f in (("a", i), ("b", P + i)):
            xy = g[f"xy_{tag}{i}"]
            n = len(xy)
            assert o["n"][f] == n, (i, tag, o["n"][f], n)
            assert np.array_equal(o["xy"][f, :n], xy) and np.array_equal(o["desc"][f, :n], g[f"desc_{tag}{i}"]), (i, tag)
            assert np.array_equal(o["nodes"][f, :n], g[f"nodes_{tag}{i}"]), (i, tag)
        m = g[f"matches{i}"]
        assert o["best"][i, 3] == len(m) and np.array_equal(o["matches"][i, :len(m)], m), i
        assert np.array_equal(o["F"][i].view(np.uint32), g[f"F{i}"].view(np.uint32)), i


@pytest.mark.parametrize("maxc,hyp,quality_note", [(300, 64, "budget below the corner count: the selection truncates"),
                                                   (2000, 128, "budget far above it")])
def test_photographs_at_other_settings_equal_the_oracle(ctx, oracle, real, maxc, hyp, quality_note):
    g, bgr = real
    P = bgr.shape[0] // 2
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    seeds = torch.arange(40, 40 + P, dtype=torch.int32).cuda()
    out = ctx.frontend_pairs(torch.from_numpy(bgr).cuda(), P, maxc, ca, sa, None, seeds, hyp, 10.0)
    ctx.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(P):
        fa = oracle.extract_features(bgr[i], maxc, ca, sa, pat)
        fb = oracle.extract_features(bgr[P + i], maxc, ca, sa, pat)
        for f, r in ((i, fa), (P + i, fb)):
            assert o["n"][f] == r["n"], (i, quality_note)
            assert np.array_equal(o["xy"][f, :r["n"]], r["xy"]) and np.array_equal(o["desc"][f, :r["n"]], r["desc"]), i
            assert np.array_equal(o["nodes"][f, :r["n"]], r["nodes"]), i
        r = oracle.match_features(fa["xy"], fa["desc"], fb["xy"], fb["desc"], 40 + i, hyp, 10.0)
        k = len(r["matches"])
        assert o["best"][i, 3] == k and np.array_equal(o["matches"][i, :k], r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(o["F"][i].view(np.uint32), np.asarray(r["F"], np.float32).view(np.uint32)), i


def test_stage_outputs_on_photographs(ctx, oracle, real):
    """cvtColor, the response image, the blur -- whole images compared, not only what survives the selection."""
    _, bgr = real
    t = torch.from_numpy(bgr).cuda()
    gray = ctx.bgr2gray(t)
    eig = ctx.min_eigen(gray)
    blur = ctx.gaussian7(gray)
    ctx.synchronize()
    gray, eig, blur = gray.cpu().numpy(), eig.cpu().numpy(), blur.cpu().numpy()
    for f in range(bgr.shape[0]):
        ref = oracle.bgr2gray(bgr[f])
        assert np.array_equal(gray[f], ref), f
        assert np.array_equal(eig[f].view(np.uint32), oracle.min_eigen(ref).view(np.uint32)), f
        assert np.array_equal(blur[f], oracle.gaussian7(ref)), f

"""tests/golden/real_v1.npz (photographic frame pairs, tests/golden/make_real.py): the frames can be remade from the stored
crops byte for byte on this machine, and the oracle still computes what the file holds for them."""
import os
import zlib

import numpy as np
import pytest

from vslam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def real():
    return np.load(os.path.join(HERE, "golden", "real_v1.npz"))


def pairs_of(g):
    for i in range(len(g["names"])):
        a, b = synth.real_pair(g[f"crop{i}"], tuple(g["motions"][i]))
        yield i, a, b


def test_frames_are_remade_byte_for_byte(real):
    crc = []
    for i, a, b in pairs_of(real):
        assert a.shape == b.shape == (480, 640, 3) and a.dtype == np.uint8
        crc += [zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes())]
        assert not np.array_equal(a, b)
    assert np.array_equal(np.array(crc, np.uint32), real["frame_crc32"])


def test_photographs_are_not_the_synthetic_regime(real):
    """What the fixture is for: saturated plateaus and smooth regions, which the synthetic textures do not have."""
    flat = 0
    for i, a, _ in pairs_of(real):
        g = a.astype(np.int32).sum(2)
        same = (np.abs(np.diff(g, axis=1)) <= 3).mean()
        flat += same > 0.3
        sat = ((a == 0) | (a == 255)).mean()
        assert sat > 0 or same > 0.3, real["names"][i]
    assert flat >= 3


def test_oracle_reproduces_the_stored_outputs(real, oracle):
    maxc, hyp, seed = (int(v) for v in real["params"])
    thr = float(real["threshold"][0])
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    for i, a, b in pairs_of(real):
        fa = oracle.extract_features(a, maxc, ca, sa, pat)
        fb = oracle.extract_features(b, maxc, ca, sa, pat)
        for tag, f in (("a", fa), ("b", fb)):
            assert np.array_equal(f["xy"], real[f"xy_{tag}{i}"]) and np.array_equal(f["desc"], real[f"desc_{tag}{i}"]), (i, tag)
            assert np.array_equal(f["nodes"], real[f"nodes_{tag}{i}"]) and f["n_detected"] == int(real[f"ndet_{tag}{i}"][0]), (i, tag)
        r = oracle.match_features(fa["xy"], fa["desc"], fb["xy"], fb["desc"], seed ^ i, hyp, thr)
        assert np.array_equal(r["matches"], real[f"matches{i}"]) and (r["prelim"], r["rc"]) == tuple(real[f"prelim{i}"]), i
        assert np.array_equal(np.asarray(r["F"], np.float32).view(np.uint32), real[f"F{i}"].view(np.uint32)), i

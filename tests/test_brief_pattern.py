"""The shipped rBRIEF table is ORB's learned one (OpenCV `bit_pattern_31_`, the table cv::ORB::compute samples:
reference src/Frame.cpp:57,68), and every copy of it in the tree is the same 1024 bytes."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "brief_pattern_31.npy")


def test_first_rows_are_opencvs_and_coordinates_fit_the_patch():
    t = np.load(GOLDEN)
    assert t.shape == (256, 4) and t.dtype == np.int8
    # the first entries of OpenCV's table, as published in its ORB source and quoted in VERDICT.md / SURVEY.md
    assert t[:3].tolist() == [[8, -3, 9, 5], [4, 2, 7, -12], [-11, 9, -8, 2]]
    assert t[3].tolist() == [7, -12, 12, -13] and t[-1].tolist() == [-1, -6, 0, -11]
    assert np.abs(t.astype(np.int32)).max() <= 13          # a rotated sample stays inside the 31 px border
    assert not np.any((t[:, 0] == t[:, 2]) & (t[:, 1] == t[:, 3]))   # no test compares a pixel with itself
    assert len({tuple(r) for r in t.tolist()}) == 256      # learned for low correlation: all pairs distinct


def test_python_header_and_library_hold_the_same_table():
    from vslam_amd import build, capi, synth
    t = np.load(GOLDEN)
    assert np.array_equal(synth.brief_pattern(), t)
    text = open(os.path.join(ROOT, "include", "vslam_brief_pattern_31.h")).read()
    body = text[text.index("{") + 1:text.index("}")]
    vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int8)
    assert np.array_equal(vals.reshape(256, 4), t)
    build.build()
    lib = capi.load_library()
    lib.vslam_brief_pattern_31.restype = ctypes.POINTER(ctypes.c_int8)
    p = lib.vslam_brief_pattern_31()
    assert np.array_equal(np.ctypeslib.as_array(p, shape=(1024,)).reshape(256, 4), t)


def test_synthetic_pattern_is_something_else():
    from vslam_amd import synth
    assert not np.array_equal(synth.synthetic_pattern(), synth.brief_pattern())

"""Pins the oracle to REAL OpenCV output — when somebody has produced it.

Everything the reference computes inside OpenCV (SURVEY.md 8 a2, a3, a4, a8, a11-a13 and the 8f pose helpers: cvtColor,
goodFeaturesToTrack, ORB::detect / compute, FAST, resize, BFMatcher::knnMatch, SVDecomp / SVD::compute, the cv::Mat algebra
of the residual and of extract_Rt / triangulate) is restated in oracle/ from
OpenCV's published algorithms, and OpenCV is not in this image, so that restatement is unverified: PARITY UNPINNED.
tools/opencv_dump.cpp runs the reference's own OpenCV calls on the seeded inputs of tests/golden/frontend_v1.npz on a
machine that has OpenCV; its output, committed as tests/golden/opencv_v1.npz, turns this file from a skip into the pin.
Until then the test reports the unpinned state and nothing else.  (The round trip of the container format is
checked either way.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import opencv_case  # noqa: E402

DUMP = os.path.join(ROOT, "tests", "golden", "opencv_v1.npz")
G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_v1.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_case_file_round_trip(tmp_path):
    arrays = opencv_case.case_inputs()
    p = str(tmp_path / "case.bin")
    opencv_case.write_records(p, arrays)
    back = opencv_case.read_records(p)
    assert sorted(back) == sorted(arrays)
    for k, v in arrays.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k


def _dump():
    if not os.path.exists(DUMP):
        pytest.skip("PARITY UNPINNED for the OpenCV-internal stages: tests/golden/opencv_v1.npz has not been produced "
                    "(build tools/opencv_dump.cpp where OpenCV 4 is installed; see its header)")
    return np.load(DUMP)


def test_extraction_matches_opencv(oracle):
    D = _dump()
    bgr = G["e_bgr"]
    for f in range(2):
        gray = oracle.bgr2gray(bgr[f])
        assert np.array_equal(gray, D[f"cv_gray{f}"]), "cvtColor(BGR2GRAY)"
        assert np.array_equal(bits(oracle.min_eigen(gray)), bits(D[f"cv_eig{f}"])), "cornerMinEigenVal"
        assert np.array_equal(oracle.good_features(gray, 150), D[f"cv_corners{f}"]), "goodFeaturesToTrack"
        assert np.array_equal(oracle.gaussian7(gray), D[f"cv_blur{f}"]), "GaussianBlur 7x7 sigma 2"
        # the golden pattern is ORB's learned table since round 4 (tests/golden/brief_pattern_31.npy), so the
        # descriptors are comparable with what OpenCV's own ORB::compute returned
        pat = G["e_pattern"]
        ca, sa = map(float, G["e_rot"])
        e = oracle.extract_features(bgr[f], 150, ca, sa, pat)
        assert np.array_equal(e["xy"], D[f"cv_kept_xy{f}"]), "ORB::compute border filter"
        assert np.array_equal(e["desc"], D[f"cv_desc{f}"]), "rBRIEF descriptors"


def test_photographs_match_opencv(oracle):
    """The same stages on the photographic frames of tests/golden/real_v1.npz, plus knnMatch + ratio test per pair."""
    D = _dump()
    if "cv_p_gray0" not in D.files:
        pytest.skip("PARITY UNPINNED on photographs: the dump predates tools/opencv_dump.cpp's p_ block")
    case = opencv_case.case_inputs()
    bgr, maxc = case["p_bgr"], int(case["p_maxc"][0])
    P = bgr.shape[0] // 2
    pat = G["e_pattern"]
    ca, sa = map(float, G["e_rot"])
    feats = []
    for f in range(2 * P):
        gray = oracle.bgr2gray(bgr[f])
        assert np.array_equal(gray, D[f"cv_p_gray{f}"]), "cvtColor(BGR2GRAY)"
        assert np.array_equal(bits(oracle.min_eigen(gray)), bits(D[f"cv_p_eig{f}"])), "cornerMinEigenVal"
        assert np.array_equal(oracle.good_features(gray, maxc), D[f"cv_p_corners{f}"]), "goodFeaturesToTrack"
        assert np.array_equal(oracle.gaussian7(gray), D[f"cv_p_blur{f}"]), "GaussianBlur 7x7 sigma 2"
        e = oracle.extract_features(bgr[f], maxc, ca, sa, pat)
        assert np.array_equal(e["xy"], D[f"cv_p_kept_xy{f}"]) and np.array_equal(e["desc"], D[f"cv_p_desc{f}"]), "ORB::compute"
        feats.append(e)
    for i in range(P):
        p, rc = oracle.match_knn2_ratio(feats[i]["desc"], feats[P + i]["desc"])
        assert rc == 0 and np.array_equal(p, D[f"cv_p_pairs{i}"]), "knnMatch + ratio test"


def test_matching_matches_opencv(oracle):
    D = _dump()
    assert np.array_equal(np.stack(oracle.match_knn2(G["m_d1"], G["m_d2"]), 1), D["cv_knn"]), "BFMatcher::knnMatch(k=2)"
    p, rc = oracle.match_knn2_ratio(G["m_d1"], G["m_d2"])
    assert rc == 0 and np.array_equal(p, D["cv_pairs"]), "ratio test"


def test_ransac_arithmetic_matches_opencv(oracle):
    D = _dump()
    if int(D["opencv_have_lapack_macro"][0]):
        pytest.skip("this OpenCV build routes SVDecomp through LAPACK: its F differs from the built-in Jacobi by design "
                    "(DESIGN.md section 2); dump with a non-LAPACK build to pin a11")
    r = oracle.find_fundamental(G["r_p1"], G["r_p2"], G["r_pairs"], G["r_fsets"], 10.0)
    assert np.array_equal(bits(r["hypF"]), bits(D["cv_hypF"])), "two SVDecomp + U diag(D) Vt"
    assert np.array_equal(r["hyp_count"], D["cv_hyp_count"]), "inlier counts"
    assert np.array_equal(bits(r["hyp_sum"]), bits(D["cv_hyp_sum"])), "cv::sum(e_sq) (element order inside cv::sum is build dependent)"


def test_grid_extractor_matches_opencv(oracle):
    """a4: extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51 — ORB's pyramid, FAST, Harris ranking, retainBest,
    intensity-centroid angles — and two of its building blocks on their own."""
    D = _dump()
    if "cv_g_xy" not in D.files:
        pytest.skip("PARITY UNPINNED for the grid extractor: the dump predates tools/opencv_dump.cpp's grid block")
    G2 = np.load(os.path.join(ROOT, "tests", "golden", "frontend_v2_grid.npz"))
    gray = oracle.bgr2gray(G2["g_bgr"])
    assert np.array_equal(oracle.fast9_16(gray, 20), D["cv_g_fast20"]), "FAST-9/16 threshold 20 + non-max"
    assert np.array_equal(oracle.resize_linear_exact(gray, 213, 160), D["cv_g_resized"]), "resize INTER_LINEAR_EXACT"
    pat_path = os.environ.get("VSLAM_BRIEF_PATTERN")
    pat = np.fromfile(pat_path, np.int8).reshape(256, 4) if pat_path else G2["g_pattern"]
    img, xy, desc, ao = oracle.extract_features_grid(G2["g_bgr"], 2, 2, pat)
    assert np.array_equal(img, D["cv_g_outlined"].reshape(img.shape)), "cv::rectangle outlines"
    assert np.array_equal(xy, D["cv_g_xy"]), "grid keypoints (positions, order)"
    assert np.array_equal(bits(ao), bits(D["cv_g_angle_octave"])), "ICAngles / octaves"
    if pat_path:
        assert np.array_equal(desc, D["cv_g_desc"]), "steered BRIEF descriptors"


def test_pose_helpers_match_opencv(oracle):
    """8f: extract_Rt and triangulate, src/helpers.cpp:3-80, on the golden pair's F and inlier matches."""
    D = _dump()
    if "cv_t_R" not in D.files:
        pytest.skip("PARITY UNPINNED for the pose helpers: the dump predates tools/opencv_dump.cpp's pose block")
    if int(D["opencv_have_lapack_macro"][0]):
        pytest.skip("this OpenCV build routes SVD through LAPACK (see test_ransac_arithmetic_matches_opencv)")
    case = opencv_case.case_inputs()
    R, t = oracle.extract_Rt(case["t_F"], case["t_K"])
    assert np.array_equal(bits(R), bits(D["cv_t_R"])), "extract_Rt: rotation"
    assert np.array_equal(bits(t), bits(D["cv_t_t"].reshape(-1))), "extract_Rt: translation"
    c2 = oracle.camera_matrix(case["t_K"], R, t)
    assert np.array_equal(bits(c2), bits(D["cv_t_c2"])), "K [R | t]"
    c1 = np.zeros((3, 4), np.float32)
    c1[:, :3] = case["t_K"]
    p4 = oracle.triangulate(case["t_p1"], case["t_p2"], c1, c2)
    assert np.array_equal(bits(p4), bits(D["cv_t_points4d"])), "triangulate: 4x4 DLT SVD per match"

"""The C++ drop-in layer (include/vslam/*.h + libvslam_host.so) used like the reference's consumers
use Frame / KDTree / RansacFilter, checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from vslam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h", [(320, 240), (323, 241)])   # (a width the device pads internally: rows with a mirrored tail)
def test_cpp_surfaces_match_oracle(oracle, tmp_path, w, h):
    from vslam_amd import build
    build.build_host()
    exe = str(tmp_path / "adapter_demo")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "adapter_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_host", "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    maxc, H, seed = 400, 96, 4242
    bgr = synth.frames_numpy(61, 1, w, h)
    pat = synth.brief_pattern()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("5i", w, h, maxc, H, seed))
        f.write(bgr.tobytes())       # no table follows: the adapters' default, ORB's learned table (= pat), is what is checked
    subprocess.run([exe, fin, fout], check=True, timeout=120)
    buf = open(fout, "rb").read()
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a

    ca, sa = synth.keypoint_rotation()        # the adapter derives the same (cos, sin) from angle = -1 deg
    ex = []
    for i in range(2):
        n, nd = take(np.int32, 2)
        ref = oracle.extract_features(bgr[i], maxc, ca, sa, pat)
        assert (n, nd) == (ref["n"], ref["n_detected"])
        assert np.array_equal(take(np.float32, 2 * n).reshape(n, 2), ref["xy"])
        assert np.array_equal(take(np.uint8, 32 * n).reshape(n, 32), ref["desc"])
        assert np.array_equal(take(np.int32, n), ref["nodes"])
        assert int(take(np.int32, 1)[0]) == int(np.floor(np.log2(n)) + 1)
        ex.append(ref)
    mf = oracle.match_features(ex[0]["xy"], ex[0]["desc"], ex[1]["xy"], ex[1]["desc"], seed, H, 10.0)
    k = int(take(np.int32, 1)[0])
    assert k == len(mf["matches"]) and k >= 8
    matches = take(np.int32, 2 * k).reshape(k, 2)
    assert np.array_equal(matches, mf["matches"])
    assert int(take(np.int32, 1)[0]) == 1
    assert np.array_equal(take(np.float32, 9).view(np.uint32), mf["F"].view(np.uint32))

    nq = int(take(np.int32, 1)[0])
    pts = ex[1]["xy"]
    for q in range(nq):
        qp = (pts[q] + np.array([0.75, -1.25], np.float32)).astype(np.float32)
        ref, cnt = oracle.kdtree_radius_frame(ex[1]["nodes"], pts, qp, 2.0)
        c = int(take(np.int32, 1)[0])
        assert c == cnt and np.array_equal(take(np.int32, c), ref)
    assert int(take(np.int32, 1)[0]) == 1          # batch form == single-query form
    assert int(take(np.int32, 1)[0]) == 1          # 3000 single queries (cell table) == host walk == batched device query
    assert int(take(np.int32, 1)[0]) > 1000        # ... and most of them had hits

    import ctypes as C
    tree = np.zeros((len(pts), 2), np.float32)
    assert oracle.lib.vso_kdtree_build_points(pts.ctypes.data_as(C.POINTER(C.c_float)), len(pts), tree.ctypes.data_as(C.POINTER(C.c_float))) == 0
    nn = np.zeros(2, np.float32)
    oracle.lib.vso_kdtree_nearest_points(tree.ctypes.data_as(C.POINTER(C.c_float)), len(pts), C.c_float(100.5), C.c_float(80.25),
                                         C.c_float(np.inf), nn.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(take(np.float32, 2), nn)
    near = np.zeros((len(pts), 2), np.float32)
    cnt = oracle.lib.vso_kdtree_radius_points(tree.ctypes.data_as(C.POINTER(C.c_float)), len(pts), C.c_float(100.5), C.c_float(80.25),
                                              C.c_float(12.0), near.ctypes.data_as(C.POINTER(C.c_float)), len(pts))
    assert int(take(np.int32, 1)[0]) == cnt
    assert np.array_equal(take(np.float32, 2 * cnt).reshape(cnt, 2), near[:cnt])

    s1 = ex[0]["xy"][matches[:8, 0]]; s2 = ex[1]["xy"][matches[:8, 1]]
    F8 = oracle.compute_fundamental(s1, s2)
    assert np.array_equal(take(np.float32, 9).view(np.uint32), F8.view(np.uint32))
    mask, c, s = oracle.residual(ex[0]["xy"], ex[1]["xy"], matches, F8, 10.0)
    assert int(take(np.int32, 1)[0]) == c
    assert take(np.float32, 1).view(np.uint32)[0] == np.float32(s).view(np.uint32)
    assert np.array_equal(take(np.int32, k), mask.astype(np.int32))
    # extract_features(frame, 3, 4): the grid ORB/FAST extractor through the C++ surface
    ref_img, gxy, gdesc, _ = oracle.extract_features_grid(bgr[0], 3, 4, pat)
    gn = int(take(np.int32, 1)[0])
    assert gn == len(gxy) and gn > 50
    assert np.array_equal(take(np.float32, 2 * gn).reshape(gn, 2).view(np.uint32), gxy.view(np.uint32))
    assert np.array_equal(take(np.uint8, 32 * gn).reshape(gn, 32), gdesc)
    assert np.array_equal(take(np.uint8, w * h * 3).reshape(h, w, 3), ref_img)
    # compute_fundamental_residual on 5 matches, then on none (src/RansacFilter.cpp:105-140 has no lower limit)
    mask5, c5, s5 = oracle.residual(ex[0]["xy"], ex[1]["xy"], matches[:5], F8, 10.0)
    assert int(take(np.int32, 1)[0]) == c5
    assert take(np.float32, 1).view(np.uint32)[0] == np.float32(s5).view(np.uint32)
    assert np.array_equal(take(np.int32, 5), mask5.astype(np.int32))
    assert int(take(np.int32, 1)[0]) == 1
    # device tree copies are validated: edited points and a recycled root address both give the fresh answer
    assert int(take(np.int32, 1)[0]) == 1
    take(np.int32, 1)                      # whether malloc really handed the old address back (usually 1; informational)
    # node-pointer overloads == device entry points; ABS / P macros
    assert int(take(np.int32, 1)[0]) == 1
    # RansacFilter(5, 64, 10): sets of 5 drawn indices + three zeros, the reference's hypothesis loop on them
    sets5 = oracle.ransac_sets(0xABCD, k, 64, min_items=5)
    r5 = oracle.find_fundamental(ex[0]["xy"], ex[1]["xy"], matches, sets5, 10.0)
    assert np.array_equal(take(np.float32, 9).view(np.uint32), r5["F"].view(np.uint32))
    assert int(take(np.int32, 1)[0]) == r5["count"]
    assert int(take(np.int32, 1)[0]) == 1          # min_items = 9 overruns the sets in the reference: refused
    assert off == len(buf)


def test_helpers_drop_in_matches_oracle(oracle, tmp_path):
    """include/vslam/helpers.h: extract_Rt and triangulate with the reference's signatures (include/helpers.h:17-19), and the batch
    form of the map-association block (src/vslam.cpp:129-161), driven the way the capture loop drives them."""
    from vslam_amd import build
    build.build_host()
    exe = str(tmp_path / "helpers_demo")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "helpers_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_host", "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    w, h, maxc, H, seed = 320, 240, 400, 96, 4242
    bgr = synth.frames_numpy(61, 1, w, h)
    pat = synth.brief_pattern()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("5i", w, h, maxc, H, seed))
        f.write(bgr.tobytes())
    subprocess.run([exe, fin, fout], check=True, timeout=120)
    buf = open(fout, "rb").read()
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a

    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    ca, sa = synth.keypoint_rotation()
    ex = [oracle.extract_features(bgr[i], maxc, ca, sa, pat) for i in range(2)]
    mf = oracle.match_features(ex[0]["xy"], ex[0]["desc"], ex[1]["xy"], ex[1]["desc"], seed, H, 10.0)
    n = int(take(np.int32, 1)[0])
    matches = take(np.int32, 2 * n).reshape(n, 2)
    assert n == len(mf["matches"]) and np.array_equal(matches, mf["matches"])
    F = take(np.float32, 9)
    assert np.array_equal(bits(F), bits(mf["F"]))
    Kmat = np.array([[525, 0, w // 2], [0, 525, h // 2], [0, 0, 1]], np.float32)
    Rr, tr = oracle.extract_Rt(F, Kmat)
    assert np.array_equal(bits(take(np.float32, 9)), bits(Rr.reshape(9)))
    assert np.array_equal(bits(take(np.float32, 3)), bits(tr))
    c2 = take(np.float32, 12)
    c2r = oracle.camera_matrix(Kmat, Rr, tr)
    assert np.array_equal(bits(c2), bits(c2r.reshape(12)))            # the demo's K * [R | t] is cv::Mat's product
    c1 = np.c_[Kmat, np.zeros(3, np.float32)]
    p1, p2 = ex[0]["xy"][matches[:, 0]], ex[1]["xy"][matches[:, 1]]
    pts = take(np.float32, 4 * n).reshape(n, 4)
    assert np.array_equal(bits(pts), bits(oracle.triangulate(p1, p2, c1, c2r)))
    nk = int(take(np.int32, 1)[0])
    assert nk == ex[1]["n"]
    ids_before, ids_after, claim = take(np.int32, nk), take(np.int32, nk), take(np.int32, n)
    n_obs = int(take(np.int32, 1)[0])
    offs = take(np.int32, n + 1)
    od = take(np.uint8, 32 * n_obs).reshape(n_obs, 32)
    ref_ids, ref_claim = oracle.associate(pts, c2r, w, h, ex[1]["nodes"], ex[1]["xy"], ex[1]["desc"], offs, od, ids_before, radius=6.0)
    assert np.array_equal(claim, ref_claim) and np.array_equal(ids_after, ref_ids)
    assert (ref_claim >= 0).sum() >= 10, "some of the triangulated inliers should find a keypoint again"
    assert off == len(buf)

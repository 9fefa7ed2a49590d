"""ctypes/numpy wrapper of oracle/liboracle.so — the CPU oracle (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load(name="liboracle.so"):
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build_oracle()
    return C.CDLL(path)


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Oracle:
    def __init__(self, libname="liboracle.so"):
        self.lib = load(libname)
        self.lib.vso_hamming256.restype = C.c_uint32

    # ------------------------------------------------------------ kd tree
    def kdtree_build_frame(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        n = xy.shape[0]
        out = np.zeros(max(n, 1), dtype=np.int32)
        assert self.lib.vso_kdtree_build_frame(_p(xy, C.c_float), n, _p(out, C.c_int32)) == 0
        return out[:n]

    def kdtree_radius_frame(self, nodes, xy, q, radius, cap=64):
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        out = np.zeros(cap, dtype=np.int32)
        cnt = self.lib.vso_kdtree_radius_frame(_p(nodes, C.c_int32), _p(xy, C.c_float), xy.shape[0],
                                               C.c_float(q[0]), C.c_float(q[1]), C.c_float(radius),
                                               _p(out, C.c_int32), cap)
        return out[:min(cnt, cap)], cnt

    # ------------------------------------------------------------ matching
    def match_knn2(self, d1, d2):
        d1 = np.ascontiguousarray(d1, dtype=np.uint8); d2 = np.ascontiguousarray(d2, dtype=np.uint8)
        n1, n2 = d1.shape[0], d2.shape[0]
        outs = [np.zeros(max(n1, 1), dtype=np.int32) for _ in range(4)]
        assert self.lib.vso_match_knn2(_p(d1, C.c_uint8), n1, _p(d2, C.c_uint8), n2,
                                       *[_p(o, C.c_int32) for o in outs]) == 0
        return [o[:n1] for o in outs]   # idx0, dist0, idx1, dist1

    def match_knn2_ratio(self, d1, d2):
        d1 = np.ascontiguousarray(d1, dtype=np.uint8); d2 = np.ascontiguousarray(d2, dtype=np.uint8)
        n1, n2 = d1.shape[0], d2.shape[0]
        pairs = np.zeros((max(n1, 1), 2), dtype=np.int32)
        m = C.c_int32(0)
        rc = self.lib.vso_match_knn2_ratio(_p(d1, C.c_uint8), n1, _p(d2, C.c_uint8), n2, _p(pairs, C.c_int32), C.byref(m))
        if rc != 0:
            return pairs[:0], rc
        return pairs[:m.value].copy(), 0

    # ------------------------------------------------------------ ransac
    def ransac_sets(self, seed, n_matches, H, min_items=8):
        out = np.zeros((H, 8), dtype=np.int32)
        rc = self.lib.vso_ransac_sets(C.c_uint32(seed & 0xFFFFFFFF), n_matches, min_items, H, _p(out, C.c_int32))
        assert rc == 0, rc
        return out

    def svd(self, A):
        A = np.ascontiguousarray(A, dtype=np.float32)
        m, n = A.shape
        w = np.zeros(min(m, n), dtype=np.float32)
        u = np.zeros((m, m), dtype=np.float32)
        vt = np.zeros((n, n), dtype=np.float32)
        assert self.lib.vso_svd32f_full(_p(A, C.c_float), m, n, _p(w, C.c_float), _p(u, C.c_float), _p(vt, C.c_float)) == 0
        return w, u, vt

    def compute_fundamental(self, p1_set, p2_set):
        p1_set = np.ascontiguousarray(p1_set, dtype=np.float32); p2_set = np.ascontiguousarray(p2_set, dtype=np.float32)
        F = np.zeros(9, dtype=np.float32)
        assert self.lib.vso_compute_fundamental(_p(p1_set, C.c_float), _p(p2_set, C.c_float), p1_set.shape[0], _p(F, C.c_float)) == 0
        return F

    def residual(self, p1, p2, pairs, F, thr):
        p1 = np.ascontiguousarray(p1, dtype=np.float32); p2 = np.ascontiguousarray(p2, dtype=np.float32)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32); F = np.ascontiguousarray(F, dtype=np.float32)
        m = pairs.shape[0]
        mask = np.zeros(max(m, 1), dtype=np.uint8)
        cnt = C.c_int32(); s = C.c_float()
        assert self.lib.vso_fundamental_residual(_p(p1, C.c_float), _p(p2, C.c_float), _p(pairs, C.c_int32), m,
                                                 _p(F, C.c_float), C.c_float(thr), _p(mask, C.c_uint8),
                                                 C.byref(cnt), C.byref(s)) == 0
        return mask[:m], cnt.value, np.float32(s.value)

    def find_fundamental(self, p1, p2, pairs, sets, thr, want_all=True):
        p1 = np.ascontiguousarray(p1, dtype=np.float32); p2 = np.ascontiguousarray(p2, dtype=np.float32)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32); sets = np.ascontiguousarray(sets, dtype=np.int32)
        m, H = pairs.shape[0], sets.shape[0]
        F = np.zeros(9, dtype=np.float32)
        mask = np.zeros(max(m, 1), dtype=np.uint8)
        bc = C.c_int32(); bs = C.c_float(); bi = C.c_int32()
        allF = np.zeros((H, 9), dtype=np.float32); allc = np.zeros(H, dtype=np.int32); alls = np.zeros(H, dtype=np.float32)
        assert self.lib.vso_find_fundamental(_p(p1, C.c_float), _p(p2, C.c_float), _p(pairs, C.c_int32), m,
                                             _p(sets, C.c_int32), H, C.c_float(thr), _p(F, C.c_float),
                                             _p(mask, C.c_uint8), C.byref(bc), C.byref(bs), C.byref(bi),
                                             _p(allF, C.c_float) if want_all else None,
                                             _p(allc, C.c_int32) if want_all else None,
                                             _p(alls, C.c_float) if want_all else None) == 0
        return dict(F=F, mask=mask[:m], count=bc.value, sum=np.float32(bs.value), winner=bi.value,
                    hypF=allF, hyp_count=allc, hyp_sum=alls)

    def match_features(self, xy1, d1, xy2, d2, seed, H, thr):
        xy1 = np.ascontiguousarray(xy1, dtype=np.float32); xy2 = np.ascontiguousarray(xy2, dtype=np.float32)
        d1 = np.ascontiguousarray(d1, dtype=np.uint8); d2 = np.ascontiguousarray(d2, dtype=np.uint8)
        n1, n2 = d1.shape[0], d2.shape[0]
        out = np.zeros((max(n1, 1), 2), dtype=np.int32)
        n = C.c_int32(); npre = C.c_int32()
        F = np.zeros(9, dtype=np.float32)
        rc = self.lib.vso_match_features(_p(xy1, C.c_float), _p(d1, C.c_uint8), n1, _p(xy2, C.c_float), _p(d2, C.c_uint8),
                                         n2, C.c_uint32(seed & 0xFFFFFFFF), H, C.c_float(thr), _p(out, C.c_int32),
                                         C.byref(n), _p(F, C.c_float), C.byref(npre))
        return dict(rc=rc, matches=out[:n.value].copy(), F=F, prelim=npre.value)

    # ------------------------------------------------------------ extraction
    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, _ = bgr.shape
        g = np.zeros((h, w), dtype=np.uint8)
        assert self.lib.vso_bgr2gray(_p(bgr, C.c_uint8), w, h, 3 * w, _p(g, C.c_uint8)) == 0
        return g

    def min_eigen(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        e = np.zeros((h, w), dtype=np.float32)
        assert self.lib.vso_min_eigen(_p(gray, C.c_uint8), w, h, _p(e, C.c_float)) == 0
        return e

    def good_features(self, gray, max_corners, quality=0.01, min_dist=3.0):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        xy = np.zeros((max_corners, 2), dtype=np.float32)
        n = C.c_int32()
        assert self.lib.vso_good_features(_p(gray, C.c_uint8), w, h, max_corners, C.c_double(quality),
                                          C.c_double(min_dist), _p(xy, C.c_float), C.byref(n)) == 0
        return xy[:n.value].copy()

    def gaussian7(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        o = np.zeros((h, w), dtype=np.uint8)
        assert self.lib.vso_gaussian7(_p(gray, C.c_uint8), w, h, _p(o, C.c_uint8)) == 0
        return o

    def orb_describe(self, blurred, xy, cos_a, sin_a, pattern):
        blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
        xy = np.ascontiguousarray(xy, dtype=np.float32); pattern = np.ascontiguousarray(pattern, dtype=np.int8)
        h, w = blurred.shape
        n = xy.shape[0]
        desc = np.zeros((max(n, 1), 32), dtype=np.uint8)
        keep = np.zeros(max(n, 1), dtype=np.int32)
        k = C.c_int32()
        assert self.lib.vso_orb_describe(_p(blurred, C.c_uint8), w, h, _p(xy, C.c_float), n, C.c_float(cos_a),
                                         C.c_float(sin_a), _p(pattern, C.c_int8), _p(desc, C.c_uint8),
                                         _p(keep, C.c_int32), C.byref(k)) == 0
        return desc[:k.value].copy(), keep[:k.value].copy()

    def extract_features(self, bgr, max_corners, cos_a, sin_a, pattern):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8); pattern = np.ascontiguousarray(pattern, dtype=np.int8)
        h, w, _ = bgr.shape
        xy = np.zeros((max_corners, 2), dtype=np.float32)
        desc = np.zeros((max_corners, 32), dtype=np.uint8)
        kd = np.zeros(max_corners, dtype=np.int32)
        n = C.c_int32(); nd = C.c_int32()
        assert self.lib.vso_extract_features(_p(bgr, C.c_uint8), w, h, 3 * w, max_corners, C.c_float(cos_a),
                                             C.c_float(sin_a), _p(pattern, C.c_int8), _p(xy, C.c_float),
                                             _p(desc, C.c_uint8), _p(kd, C.c_int32), C.byref(n), C.byref(nd)) == 0
        k = n.value
        return dict(xy=xy[:k].copy(), desc=desc[:k].copy(), nodes=kd[:k].copy(), n=k, n_detected=nd.value)

    # ------------------------------------------------------------ grid ORB/FAST extractor (a4)
    def fast9_16(self, gray, threshold, cap=200000):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        out = np.zeros((cap, 3), dtype=np.float32)
        n = C.c_int32()
        assert self.lib.vso_fast9_16(_p(gray, C.c_uint8), w, h, threshold, _p(out, C.c_float), cap, C.byref(n)) == 0
        return out[:n.value].copy()

    def resize_linear_exact(self, src, dw, dh):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        sh, sw = src.shape
        dst = np.zeros((dh, dw), dtype=np.uint8)
        assert self.lib.vso_resize_linear_exact(_p(src, C.c_uint8), sw, sh, _p(dst, C.c_uint8), dw, dh) == 0
        return dst

    def orb_detect(self, gray, nfeatures, fast_threshold, cap=20000):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        out = np.zeros((cap, 6), dtype=np.float32)
        n = C.c_int32()
        assert self.lib.vso_orb_detect(_p(gray, C.c_uint8), w, h, nfeatures, fast_threshold, _p(out, C.c_float), cap, C.byref(n)) == 0
        return out[:n.value].copy()

    def extract_features_grid(self, bgr, nrows, ncols, pattern, cap=100000):
        """Returns (outlined bgr copy, xy, desc, angle_octave)."""
        b = np.ascontiguousarray(bgr, dtype=np.uint8).copy()
        pattern = np.ascontiguousarray(pattern, dtype=np.int8)
        h, w, _ = b.shape
        xy = np.zeros((cap, 2), dtype=np.float32); desc = np.zeros((cap, 32), dtype=np.uint8)
        ao = np.zeros((cap, 2), dtype=np.float32)
        n = C.c_int32()
        assert self.lib.vso_extract_features_grid(_p(b, C.c_uint8), w, h, 3 * w, nrows, ncols, _p(pattern, C.c_int8),
                                                  _p(xy, C.c_float), _p(desc, C.c_uint8), _p(ao, C.c_float), cap, C.byref(n)) == 0
        k = n.value
        return b, xy[:k].copy(), desc[:k].copy(), ao[:k].copy()

    def sincos_deg(self, a):
        s = C.c_float(); c = C.c_float()
        self.lib.vso_sincos_deg(C.c_float(a), C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)

    # ------------------------------------------------------------ pose helpers (8f)
    def extract_Rt(self, F, K):
        F = np.ascontiguousarray(F, dtype=np.float32).reshape(9); K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
        assert self.lib.vso_extract_Rt(_p(F, C.c_float), _p(K, C.c_float), _p(R, C.c_float), _p(t, C.c_float)) == 0
        return R.reshape(3, 3), t

    def camera_matrix(self, K, R, t):
        K = np.ascontiguousarray(K, dtype=np.float32).reshape(9); R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        t = np.ascontiguousarray(t, dtype=np.float32)
        c2 = np.zeros(12, np.float32)
        assert self.lib.vso_camera_matrix(_p(K, C.c_float), _p(R, C.c_float), _p(t, C.c_float), _p(c2, C.c_float)) == 0
        return c2.reshape(3, 4)

    def triangulate(self, p1, p2, c1, c2):
        p1 = np.ascontiguousarray(p1, dtype=np.float32); p2 = np.ascontiguousarray(p2, dtype=np.float32)
        c1 = np.ascontiguousarray(c1, dtype=np.float32).reshape(12); c2 = np.ascontiguousarray(c2, dtype=np.float32).reshape(12)
        n = p1.shape[0]
        out = np.zeros((max(n, 1), 4), np.float32)
        assert self.lib.vso_triangulate(_p(p1, C.c_float), _p(p2, C.c_float), n, _p(c1, C.c_float), _p(c2, C.c_float), _p(out, C.c_float)) == 0
        return out[:n]

    def associate(self, map_points, c2, w, h, nodes, kp_xy, kp_desc, obs_offsets, obs_desc, ids, radius=2.0, thr=64):
        mp = np.ascontiguousarray(map_points, dtype=np.float32); c2 = np.ascontiguousarray(c2, dtype=np.float32).reshape(12)
        nodes = np.ascontiguousarray(nodes, dtype=np.int32); kp_xy = np.ascontiguousarray(kp_xy, dtype=np.float32)
        kp_desc = np.ascontiguousarray(kp_desc, dtype=np.uint8); oo = np.ascontiguousarray(obs_offsets, dtype=np.int32)
        od = np.ascontiguousarray(obs_desc, dtype=np.uint8)
        ids = np.ascontiguousarray(ids, dtype=np.int32).copy()
        claim = np.zeros(max(len(mp), 1), np.int32)
        assert self.lib.vso_associate_map_points(_p(mp, C.c_float), len(mp), _p(c2, C.c_float), w, h, _p(nodes, C.c_int32),
                                                 _p(kp_xy, C.c_float), _p(kp_desc, C.c_uint8), len(kp_xy), _p(oo, C.c_int32),
                                                 _p(od, C.c_uint8), C.c_float(radius), C.c_uint32(thr), _p(ids, C.c_int32),
                                                 _p(claim, C.c_int32)) == 0
        return ids, claim[:len(mp)]

    def reprojection_filter(self, pts4d, p1, p2, c1, c2, ids, thr_sq=4.0):
        pts4d = np.ascontiguousarray(pts4d, dtype=np.float32); p1 = np.ascontiguousarray(p1, dtype=np.float32)
        p2 = np.ascontiguousarray(p2, dtype=np.float32); ids = np.ascontiguousarray(ids, dtype=np.int32)
        c1 = np.ascontiguousarray(c1, dtype=np.float32).reshape(12); c2 = np.ascontiguousarray(c2, dtype=np.float32).reshape(12)
        n = len(pts4d)
        idx = np.zeros(max(n, 1), np.int32); k = C.c_int32(); err = C.c_double()
        assert self.lib.vso_reprojection_filter(_p(pts4d, C.c_float), _p(p1, C.c_float), _p(p2, C.c_float), n, _p(c1, C.c_float),
                                                _p(c2, C.c_float), _p(ids, C.c_int32), C.c_float(thr_sq), _p(idx, C.c_int32),
                                                C.byref(k), C.byref(err)) == 0
        return idx[:k.value].copy(), err.value

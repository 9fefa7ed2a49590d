"""ctypes binding of include/vslam_amd.h for the Python harness (tests, bench.py, smoke()).

Plumbing only: torch provides device memory and streams; every computation goes through the
C ABI in libvslam_amd.so.  There is no fallback path: if the library is missing or no HIP device
is visible, construction raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSLAM_AMD_LIB: another build of the same library (A/B timing of kernel variants: tools/ab_kernels.py)
LIB_PATH = os.environ.get("VSLAM_AMD_LIB") or os.path.join(_HERE, "libvslam_amd.so")
# The experiments build of the same sources (-DVSLAM_EXPERIMENTS): the environment-variable A/B switches and the kernel variants
# that were measured and not chosen live only there (tools/ab_*.py, the variant tests); the product library has neither.
EXP_LIB_PATH = os.path.join(_HERE, "libvslam_amd_exp.so")

OK = 0
ERRORS = {-1: "VSLAM_ERR_INVALID", -2: "VSLAM_ERR_HIP", -3: "VSLAM_ERR_NO_DEVICE",
          -4: "VSLAM_ERR_CAPACITY", -5: "VSLAM_ERR_DEGENERATE", -6: "VSLAM_ERR_COMM"}

# every symbol include/vslam_amd.h declares (tests/test_capi_symbols.py checks the header too)
SYMBOLS = [
    "vslam_ctx_create", "vslam_ctx_make_current", "vslam_ctx_destroy", "vslam_ctx_set_stream", "vslam_ctx_set_option", "vslam_ctx_synchronize", "vslam_ctx_wait",
    "vslam_last_error", "vslam_ctx_workspace_bytes", "vslam_version", "vslam_brief_pattern_31", "vslam_dev_alloc", "vslam_dev_free", "vslam_copy_h2d",
    "vslam_copy_d2h", "vslam_debug_stream_copy", "vslam_debug_valu_calib", "vslam_prof_enable", "vslam_prof_reset", "vslam_prof_count", "vslam_prof_get",
    "vslam_match_knn2_ratio", "vslam_ransac_sets", "vslam_ransac_fundamental", "vslam_ransac_solve",
    "vslam_ransac_evaluate", "vslam_kdtree_build",
    "vslam_kdtree_radius", "vslam_kdtree_nearest", "vslam_kdtree_cell_table", "vslam_extract_features", "vslam_extract_features_grid", "vslam_triangulate_points", "vslam_frontend_pairs_pose", "vslam_pipeline_batches_redone", "vslam_corner_stats", "vslam_bgr2gray", "vslam_min_eigen",
    "vslam_good_features", "vslam_gaussian7", "vslam_orb_describe", "vslam_extract_Rt", "vslam_triangulate", "vslam_associate_map_points", "vslam_reprojection_filter",
    "vslam_match_features",
    "vslam_frontend_pairs", "vslam_frontend_sequence", "vslam_pack_records",
    "vslam_host_alloc", "vslam_host_free", "vslam_upload_async", "vslam_upload_fence", "vslam_upload_wait", "vslam_download_async",
    "vslam_shard_range", "vslam_multi_create", "vslam_multi_destroy", "vslam_multi_size", "vslam_multi_ctx", "vslam_multi_last_error",
    "vslam_multi_frontend_pairs", "vslam_multi_frontend_pairs_resident", "vslam_comm_info", "vslam_gather_records_v", "vslam_comm_unique_id", "vslam_comm_create", "vslam_comm_destroy", "vslam_gather_records",
    "vslam_pipeline_create", "vslam_pipeline_destroy", "vslam_pipeline_size", "vslam_pipeline_ctx", "vslam_pipeline_last_error",
    "vslam_pipeline_set_option", "vslam_pipeline_acquire", "vslam_pipeline_commit", "vslam_pipeline_submit_pairs", "vslam_pipeline_submit_pairs_pose",
    "vslam_pipeline_submit_sequence", "vslam_pipeline_poll", "vslam_pipeline_wait", "vslam_pipeline_drain",
]


class ExtractParams(C.Structure):
    _fields_ = [("max_corners", C.c_int32), ("quality", C.c_double), ("min_distance", C.c_double),
                ("cos_a", C.c_float), ("sin_a", C.c_float), ("d_pattern", C.c_void_p)]


class PoseOutputs(C.Structure):   # vslam_pose_outputs
    _fields_ = [("d_R", C.c_void_p), ("d_t", C.c_void_p), ("d_c2", C.c_void_p), ("d_points4d", C.c_void_p),
                ("d_inlier_idx", C.c_void_p), ("d_n_inliers", C.c_void_p), ("d_error", C.c_void_p)]


class VslamError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise VslamError(
            f"{path} is missing: run `python -m vslam_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.vslam_last_error.restype = C.c_char_p
    lib.vslam_version.restype = C.c_char_p
    return lib


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class Context:
    """One vslam_ctx bound to a torch device/stream."""

    def __init__(self, device=0, use_torch_stream=True, lib=None):
        import torch
        self.torch = torch
        self.lib = lib if lib is not None else load_library()
        self.handle = C.c_void_p()
        rc = self.lib.vslam_ctx_create(C.c_int(device), C.byref(self.handle))
        if rc != OK:
            raise VslamError(f"vslam_ctx_create failed: {ERRORS.get(rc, rc)} (no HIP device? there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.own_stream = not use_torch_stream
        if use_torch_stream:
            s = torch.cuda.current_stream(self.device)
            self._check(self.lib.vslam_ctx_set_stream(self.handle, C.c_void_p(s.cuda_stream)))

    def _ready(self):
        """Tensors torch has just filled (kernels on torch's stream) are handed to a context that runs on a stream of its own,
        which does not wait for torch's: wait here, or a fill can land on top of the results."""
        if getattr(self, "own_stream", True):
            self.torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        if self.handle:
            self.lib.vslam_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            msg = self.lib.vslam_last_error(self.handle)
            raise VslamError(f"{ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def _dev(self, t, dtype, name):
        torch = self.torch
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            raise VslamError(f"{name}: need a contiguous cuda tensor of {dtype}, got {t.dtype} on {t.device}")
        return t

    def synchronize(self):
        self._check(self.lib.vslam_ctx_synchronize(self.handle))

    def corner_stats(self):
        """{frames, px, listed, overflowed, pool}: the corner detector's counters of the last batch (vslam_corner_stats)."""
        a = (C.c_uint64 * 5)()
        self._check(self.lib.vslam_corner_stats(self.handle, a))
        return {"frames": int(a[0]), "px_per_frame": int(a[1]), "listed_pixels": int(a[2]), "frames_overflowed": int(a[3]),
                "pool_sets": int(a[4])}

    def workspace_bytes(self):
        n = C.c_size_t()
        self._check(self.lib.vslam_ctx_workspace_bytes(self.handle, C.byref(n)))
        return n.value

    OPT_RANSAC_ALL_SUMS = 1
    OPT_RANSAC_MIN_MATCHES = 2
    OPT_RANSAC_SOLVER = 3
    OPT_MATCH_SHAPE = 4
    OPT_CORNER_WINDOW_PCT = 5
    OPT_RANSAC_MIN_ITEMS = 6
    OPT_CORNER_LIST_CAP = 7
    OPT_MATCH_FORM = 8
    OPT_TREE_FORK = 9

    def set_option(self, option, value):
        self._check(self.lib.vslam_ctx_set_option(self.handle, C.c_int(option), C.c_int(int(value))))

    # ---------------------------------------------------------------- profiling
    def prof_enable(self, on=True):
        self._check(self.lib.vslam_prof_enable(self.handle, C.c_int(1 if on else 0)))

    def prof_reset(self):
        self._check(self.lib.vslam_prof_reset(self.handle))

    def prof_report(self):
        n = self.lib.vslam_prof_count(self.handle)
        if n < 0:
            self._check(n)
        out = {}
        for i in range(n):
            name = C.create_string_buffer(128)
            ms = C.c_double()
            cnt = C.c_int64()
            self._check(self.lib.vslam_prof_get(self.handle, C.c_int(i), name, C.c_int(128), C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def debug_detect(self, bgr, max_corners, width=None):
        """(experiments build only) cvtColor + goodFeaturesToTrack from the 3-byte image: the corners BEFORE the descriptor stage's
        border filter, in rank order -- (xy (F, max_corners, 2), n (F,))."""
        torch = self.torch
        F, H = bgr.shape[0], bgr.shape[1]
        W = width if width is not None else bgr.shape[2]
        stride = bgr.shape[2] * (bgr.shape[3] if bgr.dim() == 4 else 1)
        xy = torch.zeros((F, max_corners, 2), dtype=torch.float32, device=bgr.device)
        n = torch.zeros((F,), dtype=torch.int32, device=bgr.device)
        self._ready()
        self._check(self.lib.vslam_debug_detect(self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H), C.c_int(stride),
                                                C.c_int(max_corners), C.c_int(max_corners), _ptr(xy), _ptr(n)))
        return xy, n

    def debug_valu_calib(self):
        self._check(self.lib.vslam_debug_valu_calib(self.handle))

    def debug_stream_copy(self, src, dst, bytes_per_lane):
        self._check(self.lib.vslam_debug_stream_copy(self.handle, _ptr(src), _ptr(dst), C.c_size_t(src.numel() * src.element_size()),
                                                     C.c_int(bytes_per_lane)))

    # ---------------------------------------------------------------- stages
    def match_knn2_ratio(self, desc1, n1, desc2, n2, want_knn=False):
        torch = self.torch
        B, K, _ = desc1.shape
        self._dev(desc1, torch.uint8, "desc1"); self._dev(desc2, torch.uint8, "desc2")
        self._dev(n1, torch.int32, "n1"); self._dev(n2, torch.int32, "n2")
        pairs = torch.empty((B, K, 2), dtype=torch.int32, device=desc1.device)
        m = torch.empty((B,), dtype=torch.int32, device=desc1.device)
        knn = torch.empty((B, K, 4), dtype=torch.int32, device=desc1.device) if want_knn else None
        self._check(self.lib.vslam_match_knn2_ratio(self.handle, _ptr(desc1), _ptr(n1), _ptr(desc2), _ptr(n2),
                                                    C.c_int(B), C.c_int(K), _ptr(pairs), _ptr(m), _ptr(knn)))
        return (pairs, m, knn) if want_knn else (pairs, m)

    def ransac_sets(self, seeds, m, hyp):
        torch = self.torch
        B = seeds.shape[0]
        self._dev(seeds, torch.int32, "seeds"); self._dev(m, torch.int32, "m")
        sets = torch.empty((B, hyp, 8), dtype=torch.int32, device=seeds.device)
        draws = torch.empty((B, hyp * 8), dtype=torch.int32, device=seeds.device)
        self._check(self.lib.vslam_ransac_sets(self.handle, _ptr(seeds), _ptr(m), C.c_int(B), C.c_int(hyp),
                                               _ptr(sets), _ptr(draws)))
        return sets

    def ransac_fundamental(self, xy1, xy2, pairs, m, sets, threshold):
        torch = self.torch
        B, K, _ = xy1.shape
        H = sets.shape[1]
        dev = xy1.device
        for t, dt, nm in ((xy1, torch.float32, "xy1"), (xy2, torch.float32, "xy2"), (pairs, torch.int32, "pairs"),
                          (m, torch.int32, "m"), (sets, torch.int32, "sets")):
            self._dev(t, dt, nm)
        out = dict(
            F=torch.zeros((B, 9), dtype=torch.float32, device=dev),
            mask=torch.zeros((B, K), dtype=torch.uint8, device=dev),
            best=torch.zeros((B, 4), dtype=torch.int32, device=dev),
            matches=torch.zeros((B, K, 2), dtype=torch.int32, device=dev),
            hypF=torch.zeros((B, H, 9), dtype=torch.float32, device=dev),
            hyp_count=torch.zeros((B, H), dtype=torch.int32, device=dev),
            hyp_sum=torch.zeros((B, H), dtype=torch.float32, device=dev),
        )
        self._check(self.lib.vslam_ransac_fundamental(
            self.handle, _ptr(xy1), _ptr(xy2), _ptr(pairs), _ptr(m), _ptr(sets), C.c_int(B), C.c_int(K),
            C.c_int(H), C.c_float(threshold), _ptr(out["F"]), _ptr(out["mask"]), _ptr(out["best"]),
            _ptr(out["matches"]), _ptr(out["hypF"]), _ptr(out["hyp_count"]), _ptr(out["hyp_sum"])))
        return out

    def ransac_evaluate(self, xy1, xy2, pairs, m, hypF, threshold):
        """compute_fundamental_residual for GIVEN hypotheses + the accept rule (vslam_ransac_evaluate)."""
        torch = self.torch
        B, K, _ = xy1.shape
        H = hypF.shape[1]
        dev = xy1.device
        for t, dt, nm in ((xy1, torch.float32, "xy1"), (xy2, torch.float32, "xy2"), (pairs, torch.int32, "pairs"),
                          (m, torch.int32, "m"), (hypF, torch.float32, "hypF")):
            self._dev(t, dt, nm)
        out = dict(
            F=torch.zeros((B, 9), dtype=torch.float32, device=dev),
            mask=torch.zeros((B, K), dtype=torch.uint8, device=dev),
            best=torch.zeros((B, 4), dtype=torch.int32, device=dev),
            matches=torch.zeros((B, K, 2), dtype=torch.int32, device=dev),
            hyp_count=torch.zeros((B, H), dtype=torch.int32, device=dev),
            hyp_sum=torch.zeros((B, H), dtype=torch.float32, device=dev),
        )
        self._check(self.lib.vslam_ransac_evaluate(
            self.handle, _ptr(xy1), _ptr(xy2), _ptr(pairs), _ptr(m), _ptr(hypF), C.c_int(B), C.c_int(K), C.c_int(H),
            C.c_float(threshold), _ptr(out["F"]), _ptr(out["mask"]), _ptr(out["best"]), _ptr(out["matches"]),
            _ptr(out["hyp_count"]), _ptr(out["hyp_sum"])))
        return out

    def kdtree_build(self, xy, n):
        torch = self.torch
        B, K, _ = xy.shape
        self._dev(xy, torch.float32, "xy"); self._dev(n, torch.int32, "n")
        nodes = torch.full((B, K), -1, dtype=torch.int32, device=xy.device)
        self._check(self.lib.vslam_kdtree_build(self.handle, _ptr(xy), _ptr(n), C.c_int(B), C.c_int(K), _ptr(nodes)))
        return nodes

    def kdtree_radius(self, nodes, xy, n, queries, nq, radius, hit_cap=8):
        torch = self.torch
        B, K, _ = xy.shape
        Q = queries.shape[1]
        self._dev(nodes, torch.int32, "nodes"); self._dev(xy, torch.float32, "xy"); self._dev(n, torch.int32, "n")
        self._dev(queries, torch.float32, "queries"); self._dev(nq, torch.int32, "nq")
        hits = torch.full((B, Q, hit_cap), -1, dtype=torch.int32, device=xy.device)
        counts = torch.zeros((B, Q), dtype=torch.int32, device=xy.device)
        self._check(self.lib.vslam_kdtree_radius(self.handle, _ptr(nodes), _ptr(xy), _ptr(n), C.c_int(B), C.c_int(K),
                                                 _ptr(queries), _ptr(nq), C.c_int(Q), C.c_float(radius),
                                                 _ptr(hits), _ptr(counts), C.c_int(hit_cap)))
        return hits, counts

    def kdtree_nearest(self, nodes, xy, n, queries, nq, max_distance_sq=float("inf")):
        torch = self.torch
        B, K, _ = xy.shape
        Q = queries.shape[1]
        best = torch.full((B, Q), -2, dtype=torch.int32, device=xy.device)
        self._check(self.lib.vslam_kdtree_nearest(self.handle, _ptr(nodes), _ptr(xy), _ptr(n), C.c_int(B), C.c_int(K),
                                                  _ptr(queries), _ptr(nq), C.c_int(Q), C.c_float(max_distance_sq),
                                                  _ptr(best)))
        return best

    def bgr2gray(self, bgr):
        torch = self.torch
        F, H, W, _ = bgr.shape
        self._dev(bgr, torch.uint8, "bgr")
        gray = torch.empty((F, H, W), dtype=torch.uint8, device=bgr.device)
        self._check(self.lib.vslam_bgr2gray(self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H), C.c_int(3 * W), _ptr(gray)))
        return gray

    def min_eigen(self, gray):
        torch = self.torch
        F, H, W = gray.shape
        self._dev(gray, torch.uint8, "gray")
        eig = torch.empty((F, H, W), dtype=torch.float32, device=gray.device)
        self._check(self.lib.vslam_min_eigen(self.handle, _ptr(gray), C.c_int(F), C.c_int(W), C.c_int(H), _ptr(eig)))
        return eig

    def good_features(self, gray, max_corners, quality=0.01, min_distance=3.0, kp_stride=None):
        torch = self.torch
        F, H, W = gray.shape
        K = kp_stride or max_corners
        self._dev(gray, torch.uint8, "gray")
        xy = torch.zeros((F, K, 2), dtype=torch.float32, device=gray.device)
        n = torch.zeros((F,), dtype=torch.int32, device=gray.device)
        self._check(self.lib.vslam_good_features(self.handle, _ptr(gray), C.c_int(F), C.c_int(W), C.c_int(H),
                                                 C.c_int(max_corners), C.c_double(quality), C.c_double(min_distance),
                                                 C.c_int(K), _ptr(xy), _ptr(n)))
        return xy, n

    def gaussian7(self, gray):
        torch = self.torch
        F, H, W = gray.shape
        self._dev(gray, torch.uint8, "gray")
        out = torch.empty_like(gray)
        self._check(self.lib.vslam_gaussian7(self.handle, _ptr(gray), C.c_int(F), C.c_int(W), C.c_int(H), _ptr(out)))
        return out

    def orb_describe(self, blurred, xy, n, cos_a, sin_a, pattern):
        torch = self.torch
        F, H, W = blurred.shape
        K = xy.shape[1]
        self._dev(blurred, torch.uint8, "blurred"); self._dev(xy, torch.float32, "xy"); self._dev(n, torch.int32, "n")
        self._dev(pattern, torch.int8, "pattern")
        xy_out = torch.zeros_like(xy)
        desc = torch.zeros((F, K, 32), dtype=torch.uint8, device=xy.device)
        n_out = torch.zeros_like(n)
        self._check(self.lib.vslam_orb_describe(self.handle, _ptr(blurred), C.c_int(F), C.c_int(W), C.c_int(H),
                                                _ptr(xy), _ptr(n), C.c_int(K), C.c_float(cos_a), C.c_float(sin_a),
                                                _ptr(pattern), _ptr(xy_out), _ptr(desc), _ptr(n_out)))
        return xy_out, desc, n_out

    def _params(self, max_corners, cos_a, sin_a, pattern, quality=0.01, min_distance=3.0):
        p = ExtractParams()
        p.max_corners = max_corners
        p.quality = quality
        p.min_distance = min_distance
        p.cos_a = cos_a
        p.sin_a = sin_a
        p.d_pattern = pattern.data_ptr() if pattern is not None else None   # NULL: ORB's learned table
        return p

    def extract_features(self, bgr, max_corners, cos_a, sin_a, pattern, kp_stride=None, out=None, width=None):
        """bgr: (F, H, W, 3), or rows with padding as (F, H, row_bytes) together with `width`."""
        torch = self.torch
        if bgr.dim() == 3:
            F, H, row_bytes = bgr.shape
            W = int(width)
            assert row_bytes >= 3 * W
        else:
            F, H, W, _ = bgr.shape
            row_bytes = 3 * W
        K = kp_stride or max_corners
        self._dev(bgr, torch.uint8, "bgr"); self._dev(pattern, torch.int8, "pattern")
        dev = bgr.device
        if out is None:
            out = dict(xy=torch.zeros((F, K, 2), dtype=torch.float32, device=dev),
                       desc=torch.zeros((F, K, 32), dtype=torch.uint8, device=dev),
                       nodes=torch.full((F, K), -1, dtype=torch.int32, device=dev),
                       n=torch.zeros((F,), dtype=torch.int32, device=dev),
                       n_detected=torch.zeros((F,), dtype=torch.int32, device=dev))
            self._ready()
        p = self._params(max_corners, cos_a, sin_a, pattern)
        self._check(self.lib.vslam_extract_features(self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H),
                                                    C.c_int(row_bytes), C.byref(p), C.c_int(K), _ptr(out["xy"]),
                                                    _ptr(out["desc"]), _ptr(out["nodes"]), _ptr(out["n"]),
                                                    _ptr(out["n_detected"])))
        return out

    def extract_features_grid(self, bgr, nrows, ncols, pattern, kp_stride, out=None):
        """extract_features(frame, nrows, ncols): bgr is modified in place (cell outlines)."""
        torch = self.torch
        F, H, W, _ = bgr.shape
        self._dev(bgr, torch.uint8, "bgr"); self._dev(pattern, torch.int8, "pattern")
        dev = bgr.device
        if out is None:
            out = dict(xy=torch.zeros((F, kp_stride, 2), dtype=torch.float32, device=dev),
                       desc=torch.zeros((F, kp_stride, 32), dtype=torch.uint8, device=dev),
                       angle_octave=torch.zeros((F, kp_stride, 2), dtype=torch.float32, device=dev),
                       n=torch.zeros((F,), dtype=torch.int32, device=dev))
            self._ready()
        self._check(self.lib.vslam_extract_features_grid(self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H),
                                                         C.c_int(3 * W), C.c_int(nrows), C.c_int(ncols), _ptr(pattern),
                                                         C.c_int(kp_stride), _ptr(out["xy"]), _ptr(out["desc"]),
                                                         _ptr(out["angle_octave"]), _ptr(out["n"])))
        return out

    def extract_Rt(self, F, best, K):
        import numpy as np
        torch = self.torch
        B = F.shape[0]
        Kh = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        R = torch.zeros((B, 9), dtype=torch.float32, device=F.device)
        t = torch.zeros((B, 3), dtype=torch.float32, device=F.device)
        c2 = torch.zeros((B, 12), dtype=torch.float32, device=F.device)
        self._check(self.lib.vslam_extract_Rt(self.handle, _ptr(F), _ptr(best), C.c_int(B), Kh.ctypes.data_as(C.c_void_p),
                                              _ptr(R), _ptr(t), _ptr(c2)))
        return R, t, c2

    def triangulate(self, xy1, xy2, matches, best, K, c2):
        import numpy as np
        torch = self.torch
        B, Kp, _ = xy1.shape
        Kh = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        pts = torch.zeros((B, Kp, 4), dtype=torch.float32, device=xy1.device)
        self._check(self.lib.vslam_triangulate(self.handle, _ptr(xy1), _ptr(xy2), _ptr(matches), _ptr(best), C.c_int(B),
                                               C.c_int(Kp), Kh.ctypes.data_as(C.c_void_p), _ptr(c2), _ptr(pts)))
        return pts

    def reprojection_filter(self, pts4d, xy1, xy2, matches, best, K, c2, ids, thr_sq=4.0):
        import numpy as np
        torch = self.torch
        B, Kp, _ = xy1.shape
        Kh = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        idx = torch.full((B, Kp), -1, dtype=torch.int32, device=xy1.device)
        n = torch.zeros((B,), dtype=torch.int32, device=xy1.device)
        err = torch.zeros((B,), dtype=torch.float64, device=xy1.device)
        self._check(self.lib.vslam_reprojection_filter(self.handle, _ptr(pts4d), _ptr(xy1), _ptr(xy2), _ptr(matches), _ptr(best),
                                                       C.c_int(B), C.c_int(Kp), Kh.ctypes.data_as(C.c_void_p), _ptr(c2), _ptr(ids),
                                                       C.c_float(thr_sq), _ptr(idx), _ptr(n), _ptr(err)))
        return idx, n, err

    def associate(self, map_points, n_map, c2, w, h, nodes, xy, desc, n, obs_offsets, obs_desc, ids, radius=2.0, thr=64, claim=None):
        torch = self.torch
        B, Mp, _ = map_points.shape
        Kp = xy.shape[1]
        if claim is None:
            claim = torch.full((B, Mp), -3, dtype=torch.int32, device=xy.device)
            self._ready()
        self._check(self.lib.vslam_associate_map_points(
            self.handle, _ptr(map_points), _ptr(n_map), C.c_int(B), C.c_int(Mp), _ptr(c2), C.c_int(w), C.c_int(h), _ptr(nodes),
            _ptr(xy), _ptr(desc), _ptr(n), C.c_int(Kp), _ptr(obs_offsets), _ptr(obs_desc), C.c_int(obs_desc.shape[1]),
            C.c_float(radius), C.c_uint32(thr), _ptr(ids), _ptr(claim)))
        return claim

    def match_features(self, xy1, desc1, n1, xy2, desc2, n2, seeds, hyp, threshold, out=None):
        torch = self.torch
        B, K, _ = xy1.shape
        dev = xy1.device
        if out is None:
            out = dict(matches=torch.zeros((B, K, 2), dtype=torch.int32, device=dev),
                       best=torch.zeros((B, 4), dtype=torch.int32, device=dev),
                       F=torch.zeros((B, 9), dtype=torch.float32, device=dev),
                       prelim_m=torch.zeros((B,), dtype=torch.int32, device=dev))
        self._check(self.lib.vslam_match_features(
            self.handle, _ptr(xy1), _ptr(desc1), _ptr(n1), _ptr(xy2), _ptr(desc2), _ptr(n2), C.c_int(B), C.c_int(K),
            _ptr(seeds), C.c_int(hyp), C.c_float(threshold), _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"]),
            _ptr(out["prelim_m"])))
        return out

    def frontend_pairs(self, bgr, pairs, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, kp_stride=None, out=None):
        torch = self.torch
        F, H, W, _ = bgr.shape
        assert F == 2 * pairs
        K = kp_stride or max_corners
        dev = bgr.device
        if out is None:
            out = dict(xy=torch.zeros((F, K, 2), dtype=torch.float32, device=dev),
                       desc=torch.zeros((F, K, 32), dtype=torch.uint8, device=dev),
                       nodes=torch.full((F, K), -1, dtype=torch.int32, device=dev),
                       n=torch.zeros((F,), dtype=torch.int32, device=dev),
                       matches=torch.zeros((pairs, K, 2), dtype=torch.int32, device=dev),
                       best=torch.zeros((pairs, 4), dtype=torch.int32, device=dev),
                       F=torch.zeros((pairs, 9), dtype=torch.float32, device=dev))
            self._ready()
        p = self._params(max_corners, cos_a, sin_a, pattern)
        self._check(self.lib.vslam_frontend_pairs(
            self.handle, _ptr(bgr), C.c_int(pairs), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(K),
            _ptr(seeds), C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out["nodes"]),
            _ptr(out["n"]), _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"])))
        return out

    def triangulate_points(self, p1, p2, c1, c2):
        """triangulate(p1, p2, c1, c2, points_4d) as the reference declares it: (n, 2) device points, host 3 x 4 matrices."""
        import numpy as np
        torch = self.torch
        n = p1.shape[0]
        pts = torch.zeros((n, 4), dtype=torch.float32, device=p1.device)
        c1h = np.ascontiguousarray(c1, dtype=np.float32).reshape(12)
        c2h = np.ascontiguousarray(c2, dtype=np.float32).reshape(12)
        self._check(self.lib.vslam_triangulate_points(self.handle, _ptr(p1), _ptr(p2), C.c_int(n), c1h.ctypes.data_as(C.c_void_p),
                                                      c2h.ctypes.data_as(C.c_void_p), _ptr(pts)))
        return pts

    def frontend_pairs_pose(self, bgr, pairs, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, K, ids=None, thr_sq=4.0,
                            kp_stride=None, out=None):
        """frontend_pairs + extract_Rt + triangulate + reprojection filter in one call (vslam_frontend_pairs_pose)."""
        import numpy as np
        torch = self.torch
        F, H, W, _ = bgr.shape
        assert F == 2 * pairs
        Kp = kp_stride or max_corners
        dev = bgr.device
        if out is None:
            out = dict(xy=torch.zeros((F, Kp, 2), dtype=torch.float32, device=dev),
                       desc=torch.zeros((F, Kp, 32), dtype=torch.uint8, device=dev),
                       nodes=torch.full((F, Kp), -1, dtype=torch.int32, device=dev),
                       n=torch.zeros((F,), dtype=torch.int32, device=dev),
                       matches=torch.zeros((pairs, Kp, 2), dtype=torch.int32, device=dev),
                       best=torch.zeros((pairs, 4), dtype=torch.int32, device=dev),
                       F=torch.zeros((pairs, 9), dtype=torch.float32, device=dev),
                       R=torch.zeros((pairs, 9), dtype=torch.float32, device=dev),
                       t=torch.zeros((pairs, 3), dtype=torch.float32, device=dev),
                       c2=torch.zeros((pairs, 12), dtype=torch.float32, device=dev),
                       points4d=torch.zeros((pairs, Kp, 4), dtype=torch.float32, device=dev),
                       inlier_idx=torch.zeros((pairs, Kp), dtype=torch.int32, device=dev),
                       n_inliers=torch.zeros((pairs,), dtype=torch.int32, device=dev),
                       error=torch.zeros((pairs,), dtype=torch.float64, device=dev))
            self._ready()
        p = self._params(max_corners, cos_a, sin_a, pattern)
        po = PoseOutputs(*(C.c_void_p(out[k].data_ptr()) for k in ("R", "t", "c2", "points4d", "inlier_idx", "n_inliers", "error")))
        Kh = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        self._check(self.lib.vslam_frontend_pairs_pose(
            self.handle, _ptr(bgr), C.c_int(pairs), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(Kp),
            _ptr(seeds), C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out["nodes"]),
            _ptr(out["n"]), _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"]), Kh.ctypes.data_as(C.c_void_p), _ptr(ids),
            C.c_float(thr_sq), C.byref(po)))
        return out

    def pack_records(self, F, best, matches, out=None):
        """(P,9) f32, (P,4) i32, (P,K,2) i32 -> (P, 13 + K) i32 records (same layout as shard.pack_records)."""
        torch = self.torch
        P, K = matches.shape[0], matches.shape[1]
        if out is None:
            out = torch.empty((P, 13 + K), dtype=torch.int32, device=F.device)
        self._check(self.lib.vslam_pack_records(self.handle, _ptr(F), _ptr(best), _ptr(matches), C.c_int(P), C.c_int(K), _ptr(out)))
        return out

    def frontend_sequence(self, bgr, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, kp_stride=None, out=None):
        """Consecutive frames: every frame extracted once, pair i = (frame i, frame i + 1); seeds has F - 1 entries."""
        torch = self.torch
        F, H, W, _ = bgr.shape
        assert F >= 2 and seeds.shape[0] == F - 1
        K = kp_stride or max_corners
        dev = bgr.device
        if out is None:
            out = dict(xy=torch.zeros((F, K, 2), dtype=torch.float32, device=dev),
                       desc=torch.zeros((F, K, 32), dtype=torch.uint8, device=dev),
                       nodes=torch.full((F, K), -1, dtype=torch.int32, device=dev),
                       n=torch.zeros((F,), dtype=torch.int32, device=dev),
                       matches=torch.zeros((F - 1, K, 2), dtype=torch.int32, device=dev),
                       best=torch.zeros((F - 1, 4), dtype=torch.int32, device=dev),
                       F=torch.zeros((F - 1, 9), dtype=torch.float32, device=dev))
            self._ready()
        p = self._params(max_corners, cos_a, sin_a, pattern)
        self._check(self.lib.vslam_frontend_sequence(
            self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(K),
            _ptr(seeds), C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out["nodes"]),
            _ptr(out["n"]), _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"])))
        return out


class MultiDevice:
    """ctypes stub of vslam_multi_* (include/vslam_amd.h): one process, several device slots, host arrays in and out."""

    def __init__(self, devices, lib=None):
        self.lib = lib or load_library()
        self.lib.vslam_multi_last_error.restype = C.c_char_p
        self.lib.vslam_multi_ctx.restype = C.c_void_p
        arr = (C.c_int * len(devices))(*devices)
        self.handle = C.c_void_p()
        rc = self.lib.vslam_multi_create(arr, C.c_int(len(devices)), C.byref(self.handle))
        if rc != OK:
            raise VslamError(f"{ERRORS.get(rc, rc)}: vslam_multi_create({list(devices)})")

    def size(self):
        return self.lib.vslam_multi_size(self.handle)

    def set_option(self, option, value):
        for i in range(self.size()):
            ctx = C.c_void_p(self.lib.vslam_multi_ctx(self.handle, C.c_int(i)))
            rc = self.lib.vslam_ctx_set_option(ctx, C.c_int(option), C.c_int(int(value)))
            if rc != OK:
                raise VslamError(f"{ERRORS.get(rc, rc)}: set_option")

    def frontend_pairs(self, last, cur, max_corners, cos_a, sin_a, pattern, base_seed, hyp, threshold):
        """last / cur: numpy uint8 (P, H, W, 3); pattern: numpy int8 (256, 4) or None.  Returns (records (P, 13 + K) int32,
        keypoint counts (2 P,) int32)."""
        import numpy as np
        P, H, W, _ = last.shape
        last = np.ascontiguousarray(last, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        p = ExtractParams()
        p.max_corners, p.quality, p.min_distance, p.cos_a, p.sin_a, p.d_pattern = max_corners, 0.01, 3.0, cos_a, sin_a, None
        rec = np.zeros((P, 13 + max_corners), np.int32)
        n = np.zeros(2 * P, np.int32)
        pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8)
        rc = self.lib.vslam_multi_frontend_pairs(
            self.handle, last.ctypes.data_as(C.c_void_p), cur.ctypes.data_as(C.c_void_p), C.c_int(P), C.c_int(W), C.c_int(H),
            C.c_int(3 * W), C.byref(p), pat.ctypes.data_as(C.c_void_p) if pat is not None else C.c_void_p(0), C.c_int(max_corners),
            C.c_uint32(base_seed & 0xFFFFFFFF), C.c_int(hyp), C.c_float(threshold), rec.ctypes.data_as(C.c_void_p),
            n.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise VslamError(f"{ERRORS.get(rc, rc)}: {self.lib.vslam_multi_last_error(self.handle).decode()}")
        return rec, n

    def frontend_pairs_resident(self, slices, pairs, max_corners, cos_a, sin_a, pattern, base_seed, hyp, threshold):
        """slices[r]: torch uint8 (2 * ps, H, W, 3) ALREADY on slot r's device = the slot's `last` frames then its `current` ones
        (None for an empty slice).  Returns (records, keypoint counts) like frontend_pairs."""
        import numpy as np
        shape = next(t.shape for t in slices if t is not None)
        H, W = int(shape[1]), int(shape[2])
        p = ExtractParams()
        p.max_corners, p.quality, p.min_distance, p.cos_a, p.sin_a, p.d_pattern = max_corners, 0.01, 3.0, cos_a, sin_a, None
        rec = np.zeros((pairs, 13 + max_corners), np.int32)
        n = np.zeros(2 * pairs, np.int32)
        pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8)
        ptrs = (C.c_void_p * len(slices))(*[(t.data_ptr() if t is not None else None) for t in slices])
        rc = self.lib.vslam_multi_frontend_pairs_resident(
            self.handle, ptrs, C.c_int(pairs), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p),
            pat.ctypes.data_as(C.c_void_p) if pat is not None else C.c_void_p(0), C.c_int(max_corners),
            C.c_uint32(base_seed & 0xFFFFFFFF), C.c_int(hyp), C.c_float(threshold), rec.ctypes.data_as(C.c_void_p),
            n.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise VslamError(f"{ERRORS.get(rc, rc)}: {self.lib.vslam_multi_last_error(self.handle).decode()}")
        return rec, n

    def close(self):
        if self.handle:
            self.lib.vslam_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """ctypes stub of vslam_pipeline_* (include/vslam_amd.h): k contexts on one device, batches handed to them round-robin.

    `acquire()` returns (ticket, Context view of the slot's vslam_ctx); enqueue the batch on it, then `commit(ticket)`.
    `submit_pairs` is the one-call form.  Output tensors are the caller's: keep one set per batch in flight, alive until the
    batch has been waited for, and complete on torch's side before they are handed over (the contexts run on streams of their
    own that do not wait for torch's: `alloc_outputs` synchronizes, inputs made with torch kernels need the same)."""

    class _Borrowed(Context):
        """A Context object over a vslam_ctx the pipeline owns (never destroyed from here)."""

        def __init__(self, lib, handle, device):   # noqa: no super().__init__: nothing is created
            import torch
            self.torch = torch
            self.lib = lib
            self.handle = C.c_void_p(handle)
            self.device = torch.device("cuda", device)

        def close(self):
            self.handle = C.c_void_p()

    def __init__(self, device=0, n_ctx=3, lib=None):
        self.lib = lib or load_library()
        self.lib.vslam_pipeline_last_error.restype = C.c_char_p
        self.lib.vslam_pipeline_ctx.restype = C.c_void_p
        self.handle = C.c_void_p()
        rc = self.lib.vslam_pipeline_create(C.c_int(device), C.c_int(n_ctx), C.byref(self.handle))
        if rc != OK:
            raise VslamError(f"vslam_pipeline_create({device}, {n_ctx}) failed: {ERRORS.get(rc, rc)}")
        self.device = device
        self.contexts = [self._Borrowed(self.lib, self.lib.vslam_pipeline_ctx(self.handle, C.c_int(i)), device)
                         for i in range(self.size())]

    def size(self):
        return self.lib.vslam_pipeline_size(self.handle)

    def _check(self, rc):
        if rc != OK:
            raise VslamError(f"{ERRORS.get(rc, rc)}: {self.lib.vslam_pipeline_last_error(self.handle).decode()}")

    def set_option(self, option, value):
        self._check(self.lib.vslam_pipeline_set_option(self.handle, C.c_int(option), C.c_int(int(value))))

    def acquire(self):
        ctx = C.c_void_p()
        t = C.c_int64()
        self._check(self.lib.vslam_pipeline_acquire(self.handle, C.byref(ctx), C.byref(t)))
        return t.value, self.contexts[t.value % len(self.contexts)]

    def commit(self, ticket):
        self._check(self.lib.vslam_pipeline_commit(self.handle, C.c_int64(ticket)))

    def submit_pairs(self, bgr, pairs, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, out, records=None, kp_stride=None):
        F, H, W, _ = bgr.shape
        assert F == 2 * pairs
        K = kp_stride or max_corners
        p = self.contexts[0]._params(max_corners, cos_a, sin_a, pattern)
        t = C.c_int64()
        self._check(self.lib.vslam_pipeline_submit_pairs(
            self.handle, _ptr(bgr), C.c_int(pairs), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(K), _ptr(seeds),
            C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out.get("nodes")), _ptr(out["n"]),
            _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"]), _ptr(records), C.byref(t)))
        return t.value

    def submit_pairs_pose(self, bgr, pairs, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, K, out, ids=None, thr_sq=4.0,
                          records=None, kp_stride=None):
        """vslam_pipeline_submit_pairs_pose: `out` = alloc_outputs(...) plus the pose arrays (alloc_pose_outputs)."""
        import numpy as np
        F, H, W, _ = bgr.shape
        assert F == 2 * pairs
        Kp = kp_stride or max_corners
        p = self.contexts[0]._params(max_corners, cos_a, sin_a, pattern)
        po = PoseOutputs(*(C.c_void_p(out[k].data_ptr()) for k in ("R", "t", "c2", "points4d", "inlier_idx", "n_inliers", "error")))
        Kh = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        t = C.c_int64()
        self._check(self.lib.vslam_pipeline_submit_pairs_pose(
            self.handle, _ptr(bgr), C.c_int(pairs), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(Kp), _ptr(seeds),
            C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out.get("nodes")), _ptr(out["n"]),
            _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"]), Kh.ctypes.data_as(C.c_void_p), _ptr(ids), C.c_float(thr_sq),
            C.byref(po), _ptr(records), C.byref(t)))
        return t.value

    @staticmethod
    def alloc_pose_outputs(torch, frames, pairs, K, device):
        """alloc_outputs + the arrays of vslam_pose_outputs, ready for a context's own stream."""
        out = Pipeline._alloc_outputs(torch, frames, pairs, K, device)
        out.update(R=torch.zeros((pairs, 9), dtype=torch.float32, device=device), t=torch.zeros((pairs, 3), dtype=torch.float32, device=device),
                   c2=torch.zeros((pairs, 12), dtype=torch.float32, device=device),
                   points4d=torch.zeros((pairs, K, 4), dtype=torch.float32, device=device),
                   inlier_idx=torch.zeros((pairs, K), dtype=torch.int32, device=device),
                   n_inliers=torch.zeros((pairs,), dtype=torch.int32, device=device),
                   error=torch.zeros((pairs,), dtype=torch.float64, device=device))
        torch.cuda.current_stream(device).synchronize()
        return out

    def submit_sequence(self, bgr, max_corners, cos_a, sin_a, pattern, seeds, hyp, threshold, out, records=None, kp_stride=None):
        F, H, W, _ = bgr.shape
        K = kp_stride or max_corners
        p = self.contexts[0]._params(max_corners, cos_a, sin_a, pattern)
        t = C.c_int64()
        self._check(self.lib.vslam_pipeline_submit_sequence(
            self.handle, _ptr(bgr), C.c_int(F), C.c_int(W), C.c_int(H), C.c_int(3 * W), C.byref(p), C.c_int(K), _ptr(seeds),
            C.c_int(hyp), C.c_float(threshold), _ptr(out["xy"]), _ptr(out["desc"]), _ptr(out.get("nodes")), _ptr(out["n"]),
            _ptr(out["matches"]), _ptr(out["best"]), _ptr(out["F"]), _ptr(records), C.byref(t)))
        return t.value

    @staticmethod
    def alloc_outputs(torch, frames, pairs, K, device):
        """One set of output tensors, READY for a context's own stream: torch fills them with kernels on ITS stream, which the
        pipeline's streams do not wait for -- without the synchronize below a fill can land after the batch has written its
        results (seen: keypoint counts reading 0 with five batches in flight)."""
        out = Pipeline._alloc_outputs(torch, frames, pairs, K, device)
        torch.cuda.current_stream(device).synchronize()
        return out

    @staticmethod
    def _alloc_outputs(torch, frames, pairs, K, device):
        return dict(xy=torch.zeros((frames, K, 2), dtype=torch.float32, device=device),
                    desc=torch.zeros((frames, K, 32), dtype=torch.uint8, device=device),
                    nodes=torch.full((frames, K), -1, dtype=torch.int32, device=device),
                    n=torch.zeros((frames,), dtype=torch.int32, device=device),
                    matches=torch.zeros((pairs, K, 2), dtype=torch.int32, device=device),
                    best=torch.zeros((pairs, 4), dtype=torch.int32, device=device),
                    F=torch.zeros((pairs, 9), dtype=torch.float32, device=device))

    def poll(self, ticket):
        rc = self.lib.vslam_pipeline_poll(self.handle, C.c_int64(ticket))
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def wait(self, ticket):
        self._check(self.lib.vslam_pipeline_wait(self.handle, C.c_int64(ticket)))

    def wait_status(self, ticket):
        """(rc, message) instead of raising."""
        rc = self.lib.vslam_pipeline_wait(self.handle, C.c_int64(ticket))
        return rc, (self.lib.vslam_pipeline_last_error(self.handle).decode() if rc else "")

    def batches_redone(self):
        """Batches submit_pairs / submit_sequence queued that the pipeline queued a second time with whole-image corner lists."""
        self.lib.vslam_pipeline_batches_redone.restype = C.c_int64
        return int(self.lib.vslam_pipeline_batches_redone(self.handle))

    def drain(self):
        self._check(self.lib.vslam_pipeline_drain(self.handle))

    def workspace_bytes(self):
        return sum(c.workspace_bytes() for c in self.contexts)

    def close(self):
        if self.handle:
            for c in self.contexts:
                c.close()
            self.lib.vslam_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id(lib=None):
    """128 bytes from vslam_comm_unique_id (rank 0 makes it, every rank passes the same bytes to Comm)."""
    lib = lib or load_library()
    uid = (C.c_ubyte * 128)()
    rc = lib.vslam_comm_unique_id(uid)
    if rc != OK:
        raise VslamError(f"{ERRORS.get(rc, rc)}: vslam_comm_unique_id (is librccl.so there?)")
    return bytes(uid)


class Comm:
    """ctypes stub of vslam_comm_* / vslam_gather_records*: the library's own RCCL communicator, made on a context's device."""

    def __init__(self, ctx, uid, world, rank):
        self.lib = ctx.lib
        self.world, self.rank = world, rank
        self.handle = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        ctx._check(self.lib.vslam_comm_create(ctx.handle, buf, C.c_int(world), C.c_int(rank), C.byref(self.handle)))

    def info(self):
        w, r = C.c_int(-1), C.c_int(-1)
        rc = self.lib.vslam_comm_info(self.handle, C.byref(w), C.byref(r))
        if rc != OK:
            raise VslamError(f"{ERRORS.get(rc, rc)}: vslam_comm_info")
        return w.value, r.value

    def gather(self, ctx, rec, out):
        """Equal blocks: ncclAllGather of rec (this rank's records) into out (world x rec), on ctx's stream."""
        ctx._check(self.lib.vslam_gather_records(ctx.handle, self.handle, _ptr(rec), C.c_size_t(rec.numel()), _ptr(out)))

    def gather_v(self, ctx, rec, words, out, root=-1):
        arr = (C.c_size_t * len(words))(*words)
        ctx._check(self.lib.vslam_gather_records_v(ctx.handle, self.handle, _ptr(rec), arr, C.c_int(root), _ptr(out)))

    def close(self):
        if self.handle:
            self.lib.vslam_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

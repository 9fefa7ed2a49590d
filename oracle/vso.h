/*
 * vso.h — C ABI of the CPU ORACLE ("vso" = vslam oracle).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a single-threaded CPU restatement of
 * the reference's per-frame front-end (rahulaggarwal965/vslam: src/Frame.cpp,
 * src/RansacFilter.cpp, src/KDTree.cpp, src/PointMap.cpp:36-46) and of the
 * OpenCV 4.x routines those files call.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (vslam_amd/) never
 * includes, links or calls anything in oracle/.
 *
 * PARITY PIN STATUS
 *   - KDTree / frame_kdtree (construct, nearest, radius_search): pinned by replaying
 *     the reference's own randomised differential test (tests/test_kdtree.cpp:47-151,
 *     unseeded glibc rand(), 2 x 1000 trials) against this restatement:
 *     tests/test_oracle_kdtree_replay.py expects 1000/1000 and 1000/1000.
 *   - Everything that goes through OpenCV in the reference (cvtColor, goodFeaturesToTrack,
 *     ORB::compute, BFMatcher::knnMatch, SVDecomp, Mat algebra, cv::sum): PARITY UNPINNED.
 *     OpenCV (version unpinned in the reference: makefile:4,7 `pkg-config opencv4`) is
 *     not in this container and the reference ships no golden vectors for these steps.
 *     The restatement follows OpenCV 4.x's published built-in (non-LAPACK, non-IPP)
 *     algorithms; each function says which.
 *
 * All pointers are host pointers.  Every function returns 0 on success, <0 on bad
 * arguments.  Float code is built with -ffp-contract=off, no fast-math.
 */
#ifndef VSO_ORACLE_H
#define VSO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ k-d tree */
/* frame_kdtree build: src/KDTree.cpp:107-143.  out_idx[k] = pt_index of the k-th
 * node of the malloc'd node array, which the reference fills in PRE-ORDER.        */
int vso_kdtree_build_frame(const float *xy, int n, int32_t *out_idx);
/* KDTree (point-storing twin) build: src/KDTree.cpp:3-35.  out_xy = node points in
 * pre-order (2 floats per node).                                                   */
int vso_kdtree_build_points(const float *xy, int n, float *out_xy);
/* height = floor(log2 N)+1: src/KDTree.cpp:33,119 */
int vso_kdtree_height(int n);
/* radius_search(frame_kdtree): src/KDTree.cpp:145-171.  Returns hit count (may exceed
 * cap; only the first cap hits are written), hits in visit (pre-order) order.      */
int vso_kdtree_radius_frame(const int32_t *pre_idx, const float *xy, int n,
                            float qx, float qy, float radius, int32_t *out_idx, int cap);
/* radius_search(KDTree): src/KDTree.cpp:73-101; out_xy gets 2 floats per hit.      */
int vso_kdtree_radius_points(const float *pre_xy, int n, float qx, float qy, float radius,
                             float *out_xy, int cap);
/* nearest(KDTree): src/KDTree.cpp:37-71.  out_xy[2]; stays {0,0} when nothing beats
 * max_distance_sq (the reference's TODO at :38-39).                                */
int vso_kdtree_nearest_points(const float *pre_xy, int n, float qx, float qy,
                              float max_distance_sq, float *out_xy);

/* ------------------------------------------------------------------ matching */
/* BFMatcher(NORM_HAMMING)->knnMatch(k=2): src/Frame.cpp:83-85.  Per query row the two
 * smallest Hamming distances over all train rows, ties -> lower train index first.  */
int vso_match_knn2(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                   int32_t *idx0, int32_t *dist0, int32_t *idx1, int32_t *dist1);
/* knn-2 + Lowe ratio `m[0].distance < m[1].distance * 0.7` (float x double, as written
 * at src/Frame.cpp:91) -> (queryIdx, trainIdx) in query order.  out_pairs holds 2*n1. */
int vso_match_knn2_ratio(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                         int32_t *out_pairs, int32_t *out_m);
/* orb_distance core (src/PointMap.cpp:36-46): min Hamming of one row against a list */
uint32_t vso_hamming256(const uint8_t *a, const uint8_t *b);

/* -------------------------------------------------------------------- RANSAC */
/* initialize_sets: src/RansacFilter.cpp:6-34 with the seed INJECTED (the reference
 * seeds from std::random_device, :15-16).  out_sets is H x 8 row-major.             */
int vso_ransac_sets(uint32_t seed, int n_matches, int min_items, int H, int32_t *out_sets);
/* cv::SVDecomp(A, w, u, vt, MODIFY_A|FULL_UV) for CV_32F, OpenCV 4.x built-in one-sided
 * Jacobi (modules/core/src/lapack.cpp JacobiSVDImpl_ / _SVDcompute). A is m x n
 * row-major; w has min(m,n); u is m x m; vt is n x n (FULL_UV).                      */
int vso_compute_fundamental_work(const float *p1_set, const float *p2_set, int n_set, float *F, int32_t *work);
int vso_svd32f_full(const float *A, int m, int n, float *w, float *u, float *vt);
/* compute_fundamental: src/RansacFilter.cpp:69-103 (8 x 2 floats each set).         */
int vso_compute_fundamental(const float *p1_set, const float *p2_set, int n_set, float *F);
/* compute_fundamental_residual: src/RansacFilter.cpp:105-140                         */
int vso_fundamental_residual(const float *p1, const float *p2, const int32_t *pairs, int m,
                             const float *F, float threshold, uint8_t *mask,
                             int32_t *count, float *sum);
/* find_fundamental: src/RansacFilter.cpp:36-67 given pre-drawn sets.  best_iter = -1 and
 * F/mask untouched when no hypothesis is ever accepted.  Optional per-hypothesis
 * outputs (all_F H*9, all_count H, all_sum H) may be NULL.                           */
int vso_find_fundamental(const float *p1, const float *p2, const int32_t *pairs, int m,
                         const int32_t *sets, int H, float threshold,
                         float *F, uint8_t *mask, int32_t *best_count, float *best_sum,
                         int32_t *best_iter, float *all_F, int32_t *all_count, float *all_sum);

/* ---------------------------------------------------------------- extraction */
/* cv::cvtColor(BGR2GRAY) 8U: OpenCV 4.x RGB2Gray<uchar>, 15-bit fixed point.        */
int vso_bgr2gray(const uint8_t *bgr, int w, int h, int bgr_stride, uint8_t *gray);
/* cv::cornerMinEigenVal(gray, eig, blockSize=3, ksize=3), BORDER_REFLECT_101.        */
int vso_min_eigen(const uint8_t *gray, int w, int h, float *eig);
/* cv::goodFeaturesToTrack(gray, pts, max_corners, quality, min_dist): src/Frame.cpp:61.
 * out_xy gets 2 floats per corner (integer-valued), out_n the count.                 */
int vso_good_features(const uint8_t *gray, int w, int h, int max_corners, double quality,
                      double min_dist, float *out_xy, int32_t *out_n);
/* GaussianBlur 7x7 sigma 2 on 8U, BORDER_REFLECT_101 (ORB's pre-descriptor blur).     */
int vso_gaussian7(const uint8_t *gray, int w, int h, uint8_t *out);
/* ORB::compute for provided keypoints (src/Frame.cpp:64-68): border filter (edge 31),
 * steered BRIEF with rotation (cos_a, sin_a), WTA_K=2, 32 bytes.  pattern = 256*4 int8
 * (x0,y0,x1,y1).  out_keep[i] = source index of surviving keypoint i.                 */
int vso_orb_describe(const uint8_t *blurred, int w, int h, const float *xy, int n,
                     float cos_a, float sin_a, const int8_t *pattern,
                     uint8_t *out_desc, int32_t *out_keep, int32_t *out_n);
/* extract_features(Frame&): src/Frame.cpp:53-80, end to end on a BGR image.
 * Outputs: points (kept), descriptors, pre-order frame_kdtree, n_kept, n_detected
 * (map_point_ids is sized from n_detected: src/Frame.cpp:73).                         */
int vso_extract_features(const uint8_t *bgr, int w, int h, int bgr_stride, int max_corners,
                         float cos_a, float sin_a, const int8_t *pattern,
                         float *out_xy, uint8_t *out_desc, int32_t *out_kd,
                         int32_t *out_n, int32_t *out_n_detected);

/* ------------------------------------------------- grid ORB/FAST extractor (a4) */
/* pinned sin/cos of a keypoint angle in degrees (OpenCV calls libm on a float here) */
int vso_sincos_deg(float angle_deg, float *s, float *c);
/* cv::FAST(gray, kps, threshold, nonmax=true), FAST-9/16: (x, y, score) triples in raster order */
int vso_fast9_16(const uint8_t *gray, int w, int h, int threshold, float *out_xys, int cap, int32_t *out_n);
/* cv::resize(..., INTER_LINEAR_EXACT) for 8U single channel */
int vso_resize_linear_exact(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
/* ORB::create(nfeatures, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, fast_threshold)->detect(gray):
 * 6 floats per keypoint (x, y, size, angle, response, octave) */
int vso_orb_detect(const uint8_t *gray, int w, int h, int nfeatures, int fast_threshold, float *out_kp,
                   int cap, int32_t *out_n);
/* extract_features(Frame&, nrows, ncols): src/Frame.cpp:16-51.  bgr is modified (:32). */
int vso_extract_features_grid(uint8_t *bgr, int w, int h, int stride, int nrows, int ncols,
                              const int8_t *pattern, float *out_xy, uint8_t *out_desc,
                              float *out_angle_octave, int cap, int32_t *out_n);

/* ------------------------------------------------- pose helpers */
/* extract_Rt(fundamental, K, rotation, translation): src/helpers.cpp:3-35 (3x3 row-major, t[3]) */
int vso_extract_Rt(const float *F, const float *K, float *R_out, float *t_out);
/* c2 = K * [R | t] (src/vslam.cpp:83-85,125), 3x4 row-major */
int vso_camera_matrix(const float *K, const float *R, const float *t, float *c2);
/* triangulate(p1, p2, c1, c2, points_4d): src/helpers.cpp:37-80 */
int vso_triangulate(const float *p1, const float *p2, int n, const float *c1, const float *c2, float *points_4d);

/* reprojection-error filter: src/vslam.cpp:192-251 (bug-for-bug: flat stride-3 de-homogenise, match-indexed
 * map_point_ids test).  p1/p2: the matched coordinates (n x 2).  out_idx: kept match indices. */
int vso_reprojection_filter(const float *points_4d, const float *p1, const float *p2, int n, const float *c1,
                            const float *c2, const int32_t *map_point_ids, float threshold_sq,
                            int32_t *out_idx, int32_t *out_n, double *out_err);
/* map association: src/vslam.cpp:129-161 + orb_distance (src/PointMap.cpp:36-46) */
int vso_associate_map_points(const float *map_points, int n_map, const float *c2, int img_w, int img_h,
                             const int32_t *kd_nodes, const float *kp_xy, const uint8_t *kp_desc, int n_kp,
                             const int32_t *obs_offsets, const uint8_t *obs_desc, float radius,
                             uint32_t dist_threshold, int32_t *map_point_ids, int32_t *out_claim);

/* ------------------------------------------------------------------ pipeline */
/* match_features: src/Frame.cpp:82-105 with injected seed.  out_matches 2*n1 ints.   */
int vso_match_features(const float *xy1, const uint8_t *d1, int n1,
                       const float *xy2, const uint8_t *d2, int n2,
                       uint32_t seed, int H, float threshold,
                       int32_t *out_matches, int32_t *out_n, float *F, int32_t *n_prelim);

#ifdef __cplusplus
}
#endif
#endif

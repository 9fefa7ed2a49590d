/*
 * vslam_amd.h — C ABI of the MI355X (gfx950) front-end: keypoint extraction + rBRIEF,
 * Hamming k=2 matching, RANSAC fundamental matrix, 2-D k-d tree build / radius query.
 *
 * This is the drop-in boundary for the per-frame hot path of rahulaggarwal965/vslam.
 * The reference has no FFI: its consumers include Frame.h / KDTree.h / RansacFilter.h and link
 * the objects (src/vslam.cpp:6-10).  The header-only adapters under include/vslam/ re-present
 * those C++ surfaces and call the entry points below; INTEGRATION.md shows the binding.
 * Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - Plain pointers and sizes only.  `d_` parameters are DEVICE pointers (HBM), `h_` are host.
 *   - Batched layout: item b of a batch lives at base + b * stride elements, with a per-item
 *     count array (device, int32).  "kp_stride" = keypoint slots per frame.
 *   - All device entry points are asynchronous on the context's stream and return a status;
 *     VSLAM_OK == 0.  There is NO CPU fallback: without a HIP device every call fails.
 *   - Descriptors are 32 bytes per keypoint (cv::ORB default), points are (x, y) float pairs
 *     (cv::Point2f), index pairs are (queryIdx, trainIdx) int32.
 */
#ifndef VSLAM_AMD_H
#define VSLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_OK 0
#define VSLAM_ERR_INVALID (-1)   /* bad argument (null pointer, size out of range)           */
#define VSLAM_ERR_HIP (-2)       /* a HIP runtime call failed; see vslam_last_error()         */
#define VSLAM_ERR_NO_DEVICE (-3) /* no gfx950 device visible                                 */
#define VSLAM_ERR_CAPACITY (-4)  /* a size exceeds what the kernels were built for           */
#define VSLAM_ERR_DEGENERATE (-5)/* input the reference leaves undefined (<2 train rows, <8 matches) */
#define VSLAM_ERR_COMM (-6)      /* RCCL is missing or one of its calls failed; see vslam_last_error()  */

#define VSLAM_DESC_BYTES 32
#define VSLAM_SET_SIZE 8         /* RansacFilter draws 8-subsets: src/RansacFilter.cpp:17    */
#define VSLAM_MAX_KP 16384       /* keypoint slots per frame the match key packing supports  */

typedef struct vslam_ctx vslam_ctx;

/* ------------------------------------------------------------------ context */
int vslam_ctx_create(int device, vslam_ctx **out);
int vslam_ctx_destroy(vslam_ctx *ctx);
/* HIP's current device is per host thread: vslam_ctx_create leaves the context's device current on the creating thread, and
 * every call on a context (allocation included) expects it to be.  A thread that did not create the context -- or that has
 * since worked on another device -- calls this first.                                                                      */
int vslam_ctx_make_current(vslam_ctx *ctx);
/* Borrow a caller-owned hipStream_t (e.g. torch's current stream).  Taken literally: NULL is
 * HIP's default stream.  A fresh context runs on a private non-blocking stream.               */
int vslam_ctx_set_stream(vslam_ctx *ctx, void *hip_stream);
/* Waits for the context's stream, then reads AND CLEARS the device-side error word (VSLAM_ERR_CAPACITY if a bounded list
 * overflowed since the last call).  Not for a context that a vslam_pipeline owns: there the word belongs to the ticket in
 * flight (vslam_pipeline_commit files it under the batch), and clearing it out of band would take that batch's status away --
 * use vslam_ctx_wait / vslam_pipeline_wait on those.                                                                       */
int vslam_ctx_synchronize(vslam_ctx *ctx);
/* Waits for the context's stream and nothing else (vslam_ctx_synchronize also fetches the device-side error word). */
int vslam_ctx_wait(vslam_ctx *ctx);
const char *vslam_last_error(vslam_ctx *ctx);
/* Diagnostics of the corner detector's LAST batch on this context (vslam_extract_features / vslam_frontend_* /
 * vslam_good_features); waits for the context's stream.  h_stats[0] = frames, [1] = pixels per frame, [2] = pixels the certified
 * cheap tier listed as possible corners, summed over the frames (the reference's exact arithmetic runs on these only),
 * [3] = frames whose bounded list overflowed and were redone from whole-image scratch, [4] = scratch sets in the pool (more
 * overflowing frames than this in one call: VSLAM_ERR_CAPACITY).  What bench.py reports per data regime.                   */
int vslam_corner_stats(vslam_ctx *ctx, uint64_t h_stats[5]);
/* Device memory the context's grow-only workspaces hold at the moment (bytes; what a batch shape costs beside its own
 * inputs and outputs).                                                                                    */
int vslam_ctx_workspace_bytes(vslam_ctx *ctx, size_t *bytes_out);
const char *vslam_version(void);
/* ORB's learned rBRIEF test pairs (OpenCV `bit_pattern_31_`, the table behind cv::ORB::compute, src/Frame.cpp:57,68):
 * HOST pointer to 256 x (x0, y0, x1, y1) int8, static storage.  Every d_pattern argument below accepts NULL for a
 * device copy of this table that the context keeps; other tables (tests, wider patches) are passed explicitly.       */
const int8_t *vslam_brief_pattern_31(void);
/* Context options.
 *   VSLAM_OPT_RANSAC_ALL_SUMS  0 (default): vslam_ransac_* compute what find_fundamental's accept rule can observe
 *       (src/RansacFilter.cpp:59: a hypothesis matters only if its inlier count is the pair's maximum, and its
 *       residual sum only breaks ties among those).  Every hypothesis that reaches the maximum count is counted in
 *       full and exactly; a hypothesis is abandoned as soon as it can no longer reach a count already verified for
 *       the pair -- or (round 4) can at best TIE such a count while a certified bound puts its residual sum below
 *       that of a hypothesis verified to reach it: among equal counts the rule keeps the larger sum, so it can neither
 *       win nor tie.  d_hyp_count holds the maximum for the hypotheses that reach it -- except for those abandoned on
 *       the sum -- and -1 for all others; d_hyp_sum holds the exact sum where it can still decide the winner, -inf for
 *       maximum-count hypotheses a certified bound places below the winner's, NaN where the count is -1.  Winner,
 *       mask, F and matches are the reference's either way.
 *       1: the count and the residual sum of EVERY hypothesis are computed as the reference does (:105-140) — what
 *       the per-hypothesis parity tests ask for; about 3x the scoring time.
 *   VSLAM_OPT_RANSAC_MIN_MATCHES  8 (default) .. 1: vslam_ransac_evaluate skips items with fewer matches than this
 *       (no model: winner -1).  find_fundamental needs 8 to draw a set (src/RansacFilter.cpp:24), but
 *       compute_fundamental_residual scores a GIVEN F on any number of matches (:105-140): its adapter sets 1.
 *   VSLAM_OPT_RANSAC_SOLVER  0 (default): compute_fundamental as the reference computes it (OpenCV's Jacobi SVD replayed
 *       operation by operation; bit-exact with the oracle).  1: opt-in APPROXIMATE solver for throughput experiments
 *       (BASELINE.json configs[4]): conditioned 9x9 normal matrix on the matrix cores (v_mfma_f32_16x16x4_f32) +
 *       inverse iteration.  NOT bit-exact: F equals the exact solver's up to sign and about 1e-4 (unit-norm F) on
 *       well-conditioned samples; inlier masks may differ.  Never used unless set.
 *   VSLAM_OPT_MATCH_SHAPE  0 (default, and the only value the product library accepts: 8 waves x 32 query rows).
 *       EXPERIMENTS build (libvslam_amd_exp.so, -DVSLAM_EXPERIMENTS) only: 1 the same, 2: 4 waves x 64 query rows.  Same
 *       results bit for bit; kept there so that the choice can be re-measured (tools/ab_match.sh).                 */
#define VSLAM_OPT_RANSAC_ALL_SUMS 1
#define VSLAM_OPT_RANSAC_MIN_MATCHES 2
#define VSLAM_OPT_RANSAC_SOLVER 3
#define VSLAM_OPT_MATCH_SHAPE 4
/*   VSLAM_OPT_CORNER_WINDOW_PCT  135 (default): how many of the detector's possible corners get the exact arithmetic
 *       at once, in percent of max_corners (+ 128).  The selection ranks about 1.3 x max_corners candidates; a frame
 *       whose selection needs more than were evaluated is redone with all of them, so the value changes speed, never
 *       results (0 = always everything; a small value forces the redo: a test knob).                              */
#define VSLAM_OPT_CORNER_WINDOW_PCT 5
/*   VSLAM_OPT_RANSAC_MIN_ITEMS  8 (default) .. 0: RansacFilter::min_items.  initialize_sets draws min_items indices
 *       without replacement into sets that are 8 wide whatever min_items is (src/RansacFilter.cpp:17,22): entries
 *       min_items .. 7 stay 0 and find_fundamental uses all 8 (:49-53).  Values above 8 overrun the set in the
 *       reference and are rejected here.                                                                          */
#define VSLAM_OPT_RANSAC_MIN_ITEMS 6
/*   VSLAM_OPT_CORNER_LIST_CAP  0 (default): the corner detector's per-frame lists hold 16 x max_corners + 4096 entries
 *       (image data lists about 10 x max_corners).  A frame that needs more (response plateaus, pure noise) is redone
 *       from whole-image scratch taken from a pool of max(4, frames / 16) (at most 64) sets: the results never depend on
 *       the bound.  Only if more frames of ONE call overflow than the pool has sets do those frames come back without
 *       corners, and vslam_ctx_synchronize returns VSLAM_ERR_CAPACITY.  n > 0: n entries per frame (a test knob);
 *       -1: every list sized for the whole image, as before round 4 (nothing can overflow; 16 bytes per pixel and frame). */
#define VSLAM_OPT_CORNER_LIST_CAP 7
/*   VSLAM_OPT_MATCH_FORM  0 (default) / 1: the matcher forms Hamming distances as FP4 (+-1) dot products on the matrix
 *       cores (v_mfma_scale_f32_32x32x64_f8f6f4).  EXPERIMENTS build only: 2 = as int8 (0 / 1) dot products
 *       (v_mfma_i32_32x32x32_i8); the product library carries the FP4 kernel alone and answers 2 with VSLAM_ERR_INVALID.
 *       Exact either way, same results bit for bit.                                                                */
#define VSLAM_OPT_MATCH_FORM 8
/*   VSLAM_OPT_TREE_FORK  where vslam_frontend_pairs / _sequence start the k-d build (an output of the path that no later
 *       stage reads) on the auxiliary stream: -1 (default) by size (behind the matcher up to 2048 keypoint slots, in front
 *       of it above), 0 in front of the matcher, 1 behind it, 2 behind the set mapping, 3 behind the 8-point solves,
 *       4 behind the screen, 5 not forked at all (in line on the main stream, at the end of extraction).  The call's last
 *       kernel waits for it.  Same results; a tuning knob.                                                             */
#define VSLAM_OPT_TREE_FORK 9
int vslam_ctx_set_option(vslam_ctx *ctx, int option, int value);

/* device memory + copies for hosts that have no other allocator (the C++ adapters) */
int vslam_dev_alloc(vslam_ctx *ctx, size_t bytes, void **d_out);
int vslam_dev_free(vslam_ctx *ctx, void *d_ptr);
int vslam_copy_h2d(vslam_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int vslam_copy_d2h(vslam_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* Frame ingest (what replaces cv::VideoCapture's per-frame host Mat, src/vslam.cpp:54-60).
 * Page-locked host buffers and uploads that run on the context's copy stream beside the kernels:
 *   vslam_host_alloc / vslam_host_free   page-locked host memory (hipHostMalloc)
 *   vslam_upload_async                   enqueue host -> device on the copy stream (returns at once)
 *   vslam_upload_fence                   later work on the compute stream waits for every upload enqueued so far
 *   vslam_upload_wait                    the calling thread waits for them (before a host buffer is refilled)  */
int vslam_host_alloc(vslam_ctx *ctx, size_t bytes, void **h_out);
int vslam_host_free(vslam_ctx *ctx, void *h_ptr);
int vslam_upload_async(vslam_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int vslam_upload_fence(vslam_ctx *ctx);
int vslam_upload_wait(vslam_ctx *ctx);
/* The way back: device -> host on the context's COMPUTE stream, behind everything queued on it so far (the result copy that
 * closes a batch).  Returns at once; the bytes are there when the stream has been waited for (vslam_ctx_wait /
 * vslam_ctx_synchronize / vslam_pipeline_wait).  Page-locked h_dst keeps the copy asynchronous.                          */
int vslam_download_async(vslam_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* per-kernel timing with HIP events on the context's stream (bench.py's roofline leg) */
int vslam_prof_enable(vslam_ctx *ctx, int on);
int vslam_prof_reset(vslam_ctx *ctx);
int vslam_prof_count(vslam_ctx *ctx);   /* synchronises, folds pending events, returns #kernels */
int vslam_prof_get(vslam_ctx *ctx, int i, char *name, int name_cap, double *total_ms, int64_t *launches);

/* profiling aid: a plain streaming copy of `bytes` (multiple of 16) with 4 or 16 bytes per lane, so
 * rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count per access width */
int vslam_debug_stream_copy(vslam_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int bytes_per_lane);
/* profiling aid: a kernel that keeps every SIMD's vector pipe busy for its whole duration (8 waves per SIMD of independent
 * v_fma_f32), so that rocprofv3's SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE ratio that means "vector pipe 100 % busy" is measured
 * rather than assumed (tools/sq_summary.py) */
int vslam_debug_valu_calib(vslam_ctx *ctx);

/* ----------------------------------------------------------------- matching */
/* Replaces match_features' front half, src/Frame.cpp:83-94:
 *   BFMatcher(NORM_HAMMING)->knnMatch(desc1, desc2, k=2) + `m[0].distance < m[1].distance*0.7`.
 * d_desc1/d_desc2: [batch][kp_stride][32] u8; d_n1/d_n2: [batch] int32.
 * d_pairs: [batch][kp_stride][2] int32 (queryIdx, trainIdx) in query order; d_m: [batch].
 * Optional d_knn (may be NULL): [batch][kp_stride][4] int32 = idx0, dist0, idx1, dist1.
 * Items with fewer than 2 train rows produce m = 0 (the reference reads m[1] regardless).    */
int vslam_match_knn2_ratio(vslam_ctx *ctx, const uint8_t *d_desc1, const int32_t *d_n1,
                           const uint8_t *d_desc2, const int32_t *d_n2, int batch, int kp_stride,
                           int32_t *d_pairs, int32_t *d_m, int32_t *d_knn);

/* ------------------------------------------------------------------- RANSAC */
/* Replaces RansacFilter::initialize_sets, src/RansacFilter.cpp:6-34, with the seed injected
 * (std::mt19937(seed) + libstdc++ uniform_int_distribution, draws without replacement).
 * d_seeds: [batch] u32; d_m: [batch] matches per item; d_sets: [batch][hyp][8] int32.
 * d_draw_scratch: [batch][hyp*8] u32 workspace.  Items with m < 8 get all-zero sets.          */
int vslam_ransac_sets(vslam_ctx *ctx, const uint32_t *d_seeds, const int32_t *d_m, int batch,
                      int hyp, int32_t *d_sets, uint32_t *d_draw_scratch);

/* Replaces RansacFilter::find_fundamental (+ compute_fundamental, compute_fundamental_residual),
 * src/RansacFilter.cpp:36-140, for pre-drawn sets, and match_features' back half
 * (src/Frame.cpp:96-102: keep the winner's inlier matches).
 * d_xy1/d_xy2: [batch][kp_stride][2] f32; d_pairs/d_m as produced by vslam_match_knn2_ratio.
 * Outputs: d_F [batch][9] f32 (row-major 3x3, untouched when no hypothesis is accepted),
 *          d_mask [batch][kp_stride] u8, d_best [batch][4] int32 = winner, count, bits(sum), n_out,
 *          d_matches [batch][kp_stride][2] int32 compacted inlier matches (n_out of them).
 * Workspaces: d_hypF [batch][hyp][9] f32, d_hyp_count [batch][hyp] int32, d_hyp_sum [batch][hyp] f32
 * (also the per-hypothesis outputs the parity tests read: every F always; every count and sum
 * with VSLAM_OPT_RANSAC_ALL_SUMS, otherwise those of the maximum-count hypotheses, see above). */
int vslam_ransac_fundamental(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2,
                             const int32_t *d_pairs, const int32_t *d_m, const int32_t *d_sets,
                             int batch, int kp_stride, int hyp, float threshold, float *d_F,
                             uint8_t *d_mask, int32_t *d_best, int32_t *d_matches, float *d_hypF,
                             int32_t *d_hyp_count, float *d_hyp_sum);

/* The two halves of vslam_ransac_fundamental, exposed because RansacFilter's public surface has
 * them as separate methods:
 *   vslam_ransac_solve     = compute_fundamental for every set (src/RansacFilter.cpp:69-103)
 *   vslam_ransac_evaluate  = compute_fundamental_residual for every given hypothesis (:105-140) +
 *                            the accept rule of find_fundamental (:59) + the winner's mask/matches.
 * With hyp = 1 they are the single-call forms of those methods.                               */
int vslam_ransac_solve(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2, const int32_t *d_pairs,
                       const int32_t *d_m, const int32_t *d_sets, int batch, int kp_stride, int hyp,
                       float *d_hypF);
int vslam_ransac_evaluate(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2,
                          const int32_t *d_pairs, const int32_t *d_m, const float *d_hypF, int batch,
                          int kp_stride, int hyp, float threshold, float *d_F, uint8_t *d_mask,
                          int32_t *d_best, int32_t *d_matches, int32_t *d_hyp_count, float *d_hyp_sum);

/* ------------------------------------------------------------------ k-d tree */
/* Replaces construct_kdtree(frame_kdtree&, points), src/KDTree.cpp:107-143.  The tree is the
 * reference's pre-order node array reduced to its pt_index column: d_nodes [batch][kp_stride].
 * Child positions are implicit (left subtree len/2 nodes, right len - len/2 - 1).  Tie placement
 * reproduces libstdc++'s std::nth_element (introselect) exactly.  A tree is built in one workgroup's
 * LDS (20 bytes per slot): kp_stride <= 8160, VSLAM_ERR_CAPACITY beyond — which is also the limit of
 * vslam_extract_features / vslam_frontend_* when they are asked for the trees (d_nodes != NULL).  */
int vslam_kdtree_build(vslam_ctx *ctx, const float *d_xy, const int32_t *d_n, int batch,
                       int kp_stride, int32_t *d_nodes);
/* Replaces radius_search(frame_kdtree, points, query, radius), src/KDTree.cpp:145-171.
 * d_queries [batch][q_stride][2], d_nq [batch]; hits in the reference's visit (pre-order) order:
 * d_hits [batch][q_stride][hit_cap] (first hit_cap only), d_counts [batch][q_stride] (true count). */
int vslam_kdtree_radius(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy,
                        const int32_t *d_n, int batch, int kp_stride, const float *d_queries,
                        const int32_t *d_nq, int q_stride, float radius, int32_t *d_hits,
                        int32_t *d_counts, int hit_cap);

/* Replaces nearest(KDTree, query, max_distance_sq), src/KDTree.cpp:37-71, on the same pre-order
 * array: d_best_idx [batch][q_stride] = index of the nearest point, or -1 when none is closer than
 * max_distance_sq (the reference then returns a default-constructed {0,0} point).             */
int vslam_kdtree_nearest(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy,
                         const int32_t *d_n, int batch, int kp_stride, const float *d_queries,
                         const int32_t *d_nq, int q_stride, float max_distance_sq,
                         int32_t *d_best_idx);

/* The points of a tree filed by integer pixel cell, for callers that ask ONE radius query at a time
 * (radius_search(frame.kdtree, frame.points, q, 2) once per in-view map point, src/vslam.cpp:146-160): a device round
 * trip per query costs 200 x the reference's pointer chase, a probe of this table on the host does not.
 * d_table [batch][slots][2] uint32 = {cell key ((floor(y) + 32768) << 16 | (floor(x) + 32768)), pre-order position},
 * 0xFFFFFFFF = empty; open addressing from ((key * 2654435761) >> 7) & (slots - 1), linear; every point owns a slot;
 * slots a power of two >= 2 * kp_stride.  d_ok [batch] = 0 when a coordinate does not fit the key (table unusable).
 * Hits of a radius query = the points of the cells floor(q - r) .. floor(q + r) with dx * dx + dy * dy < r * r, in
 * ascending pre-order position: exactly the reference's result and order (src/KDTree.cpp:151-171).              */
int vslam_kdtree_cell_table(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy, const int32_t *d_n, int batch,
                            int kp_stride, int slots, uint32_t *d_table, int32_t *d_ok);

/* ---------------------------------------------------------------- extraction */
typedef struct vslam_extract_params {
    int32_t max_corners;     /* goodFeaturesToTrack maxCorners (3000 in src/Frame.cpp:61)      */
    double quality;          /* 0.01                                                           */
    double min_distance;     /* 3                                                              */
    float cos_a, sin_a;      /* steered-BRIEF rotation; KeyPoint(p,20) has angle -1 deg        */
    const int8_t *d_pattern; /* DEVICE [256][4] int8 (x0,y0,x1,y1) rBRIEF test pairs; NULL = ORB's
                                learned table (vslam_brief_pattern_31), what cv::ORB::compute
                                samples (src/Frame.cpp:57,68)                                  */
} vslam_extract_params;

/* Replaces extract_features(Frame&), src/Frame.cpp:53-80, for a batch of BGR frames:
 * cvtColor -> goodFeaturesToTrack -> ORB::compute (border filter, 7x7 blur, rBRIEF) -> kd-tree.
 * d_bgr: [frames][height][row_stride] u8 (3 bytes per pixel).
 * Outputs per frame: d_xy [frames][kp_stride][2], d_desc [frames][kp_stride][32],
 * d_nodes [frames][kp_stride] (may be NULL), d_n [frames] kept keypoints, d_n_detected [frames]
 * (the pre-filter count that sizes map_point_ids, src/Frame.cpp:73).                          */
int vslam_extract_features(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                           int row_stride, const vslam_extract_params *params, int kp_stride,
                           float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                           int32_t *d_n_detected);

/* Replaces extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51 (the grid ORB/FAST extractor;
 * dead in the reference: its call at src/vslam.cpp:63 is commented out).  Per grid cell: black outline
 * drawn INTO d_bgr (:32), ORB(500, 1.2, 8, 31, 0, 2, HARRIS, 31, fastThreshold 20)->detect, replaced by
 * the fastThreshold-5 detector's result when fewer than 500 keypoints were found (:33-36); then
 * ORB::compute on the whole outlined image (:43).  Like the reference it builds no k-d tree and does
 * not touch map_point_ids.  Outputs per frame: d_xy [frames][kp_stride][2] (ORB::compute's order:
 * grouped by pyramid level), d_desc [frames][kp_stride][32], optional d_angle_octave
 * [frames][kp_stride][2] (degrees, level), d_n [frames].                                          */
int vslam_extract_features_grid(vslam_ctx *ctx, uint8_t *d_bgr, int frames, int width, int height,
                                int row_stride, int nrows, int ncols, const int8_t *d_pattern,
                                int kp_stride, float *d_xy, uint8_t *d_desc, float *d_angle_octave,
                                int32_t *d_n);

/* stage-level entry points (parity tests; each is one step of vslam_extract_features) */
int vslam_bgr2gray(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                   int row_stride, uint8_t *d_gray);
int vslam_min_eigen(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                    float *d_eig);
int vslam_good_features(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                        int max_corners, double quality, double min_distance, int kp_stride,
                        float *d_xy, int32_t *d_n);
int vslam_gaussian7(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                    uint8_t *d_out);
int vslam_orb_describe(vslam_ctx *ctx, const uint8_t *d_blurred, int frames, int width, int height,
                       const float *d_xy_in, const int32_t *d_n_in, int kp_stride, float cos_a,
                       float sin_a, const int8_t *d_pattern, float *d_xy_out, uint8_t *d_desc,
                       int32_t *d_n_out);

/* ------------------------------------------------------- pose (SURVEY.md 8f "next" rows) */
/* Replaces extract_Rt(fundamental, K, rotation, translation), src/helpers.cpp:3-35, for a batch of
 * fundamental matrices, plus the camera matrix c2 = K * [R | t] of src/vslam.cpp:83-85,125.
 * d_F [batch][9]; d_best [batch][4] as written by vslam_ransac_* (items with winner < 0 are skipped;
 * may be NULL); h_K: HOST 3x3 intrinsics (row-major).  d_R [batch][9], d_t [batch][3], d_c2 [batch][12]. */
int vslam_extract_Rt(vslam_ctx *ctx, const float *d_F, const int32_t *d_best, int batch, const float *h_K,
                     float *d_R, float *d_t, float *d_c2);
/* Replaces triangulate(p1, p2, c1, c2, points_4d), src/helpers.cpp:37-80, with c1 = [K | 0]
 * (src/vslam.cpp:123-124): one 4x4 SVD per inlier match (d_matches / d_best[.][3] from RANSAC).
 * d_points4d [batch][kp_stride][4] (x, y, z, 1).                                                    */
int vslam_triangulate(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2, const int32_t *d_matches,
                      const int32_t *d_best, int batch, int kp_stride, const float *h_K,
                      const float *d_c2, float *d_points4d);

/* triangulate(p1, p2, c1, c2, points_4d) exactly as the reference declares it (include/helpers.h:19, src/helpers.cpp:37-80):
 * n point pairs (d_p1, d_p2: [n][2] f32 on the device), any two 3 x 4 camera matrices (HOST, row-major) -> d_points4d [n][4]
 * (x, y, z, 1).  What the C++ drop-in triangulate() of include/vslam/helpers.h calls.                                       */
int vslam_triangulate_points(vslam_ctx *ctx, const float *d_p1, const float *d_p2, int n, const float *h_c1,
                             const float *h_c2, float *d_points4d);

/* Replaces the reprojection-error filter of src/vslam.cpp:192-251, reproducing the reference exactly,
 * including its two indexing quirks (the de-homogenise loop strides the flat N x 3 array by 3 up to N, and
 * map_point_ids is tested at the MATCH index).  d_points4d from vslam_triangulate; d_map_point_ids
 * [batch][kp_stride] of the current frame; threshold_sq = 4 in the reference (src/vslam.cpp:50).
 * d_inlier_idx [batch][kp_stride] kept match indices (ascending), d_n_inliers [batch], d_error [batch] f64. */
int vslam_reprojection_filter(vslam_ctx *ctx, const float *d_points4d, const float *d_xy1, const float *d_xy2,
                              const int32_t *d_matches, const int32_t *d_best, int batch, int kp_stride,
                              const float *h_K, const float *d_c2, const int32_t *d_map_point_ids,
                              float threshold_sq, int32_t *d_inlier_idx, int32_t *d_n_inliers,
                              double *d_error);

/* Replaces the map-association loop of src/vslam.cpp:129-161 and orb_distance (src/PointMap.cpp:36-46):
 * project each map point with c2, radius_search (r = 2 in the reference) in the frame's k-d tree, and
 * give it the first hit that is unassigned and within `dist_threshold` (64) Hamming of the map point's
 * observations.  Sequential semantics preserved: lower map indices claim first.
 * d_map_points [batch][map_stride][4] (x,y,z,1), d_n_map [batch]; d_c2 [batch][12];
 * d_nodes/d_xy/d_desc/d_n: the frame's features as written by vslam_extract_features;
 * observations in CSR form: d_obs_offsets [batch][map_stride+1], d_obs_desc [batch][obs_stride][32];
 * d_map_point_ids [batch][kp_stride] in/out (-1 = free); d_claim [batch][map_stride] = keypoint or -1.
 * At most 16 acceptable hits per map point are kept; more sets a sticky flag that
 * vslam_ctx_synchronize reports as VSLAM_ERR_CAPACITY.                                            */
int vslam_associate_map_points(vslam_ctx *ctx, const float *d_map_points, const int32_t *d_n_map, int batch,
                               int map_stride, const float *d_c2, int img_w, int img_h,
                               const int32_t *d_nodes, const float *d_xy, const uint8_t *d_desc,
                               const int32_t *d_n, int kp_stride, const int32_t *d_obs_offsets,
                               const uint8_t *d_obs_desc, int obs_stride, float radius,
                               uint32_t dist_threshold, int32_t *d_map_point_ids, int32_t *d_claim);

/* ------------------------------------------------------------------ pipeline */
/* match_features(frame1, frame2, rf, matches, F), src/Frame.cpp:82-105, for a batch of pairs
 * whose features are already on the device: match -> sets -> RANSAC -> inlier matches.
 * Workspaces are owned by the context and sized on first use.                                */
int vslam_match_features(vslam_ctx *ctx, const float *d_xy1, const uint8_t *d_desc1,
                         const int32_t *d_n1, const float *d_xy2, const uint8_t *d_desc2,
                         const int32_t *d_n2, int batch, int kp_stride, const uint32_t *d_seeds,
                         int hyp, float threshold, int32_t *d_matches, int32_t *d_best, float *d_F,
                         int32_t *d_prelim_m);

/* The whole front-end for a batch of independent frame pairs: frames [0, pairs) are the "last"
 * frames, frames [pairs, 2*pairs) the "current" ones; pair p = (frame p, frame pairs + p).
 * Extract all 2*pairs frames, match last->current, RANSAC.  This is what bench.py times.
 * d_xy / d_desc / d_nodes / d_n are the per-frame outputs of vslam_extract_features for all
 * 2*pairs frames; d_matches / d_best / d_F are per pair as in vslam_match_features.          */
int vslam_frontend_pairs(vslam_ctx *ctx, const uint8_t *d_bgr, int pairs, int width, int height,
                         int row_stride, const vslam_extract_params *params, int kp_stride,
                         const uint32_t *d_seeds, int hyp, float threshold,
                         float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                         int32_t *d_matches, int32_t *d_best, float *d_F);

/* The capture loop's whole per-pair chain in one call, nothing leaving the device in between (src/vslam.cpp:60-88: extract,
 * match_features, extract_Rt, R_t; :120-125 the camera matrices; :186 triangulate; :192-251 the reprojection filter):
 * vslam_frontend_pairs, then vslam_extract_Rt, vslam_triangulate (c1 = [K | 0]) and vslam_reprojection_filter on its matches.
 * h_K: HOST 3 x 3 intrinsics.  d_map_point_ids [pairs][kp_stride] of the current frames, or NULL for "none assigned" (-1
 * everywhere).  Outputs as the four entry points write them; every pointer of `pose` is required.  Usable on a context of a
 * vslam_pipeline like every other entry point.                                                                              */
typedef struct vslam_pose_outputs {
    float *d_R;             /* [pairs][9]  */
    float *d_t;             /* [pairs][3]  */
    float *d_c2;            /* [pairs][12] */
    float *d_points4d;      /* [pairs][kp_stride][4] */
    int32_t *d_inlier_idx;  /* [pairs][kp_stride] match indices that pass the reprojection filter, ascending */
    int32_t *d_n_inliers;   /* [pairs] */
    double *d_error;        /* [pairs] summed reprojection error of the kept matches */
} vslam_pose_outputs;
int vslam_frontend_pairs_pose(vslam_ctx *ctx, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                              const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                              float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                              int32_t *d_matches, int32_t *d_best, float *d_F, const float *h_K,
                              const int32_t *d_map_point_ids, float reproj_threshold_sq, const vslam_pose_outputs *pose);

/* Fixed-size per-pair result records for the one exchange of the multi-GPU path: per pair
 * 13 + kp_stride int32 words = F (9 words, bit-preserving), d_best's 4 words, then one word per match slot,
 * query index | train index << 16 (keypoint indices are below VSLAM_MAX_KP).  d_records: [pairs][13 + kp_stride]. */
int vslam_pack_records(vslam_ctx *ctx, const float *d_F, const int32_t *d_best, const int32_t *d_matches,
                       int pairs, int kp_stride, int32_t *d_records);

/* The same path for a run of consecutive video frames, the shape of the reference's main loop
 * (src/vslam.cpp:60-77: every new frame is matched against the previous one): extract each of the `frames`
 * frames ONCE, then pair i = (frame i, frame i + 1) for i in [0, frames - 1).
 * Per-frame outputs as in vslam_extract_features ([frames] slots); d_seeds, d_matches, d_best, d_F have
 * frames - 1 slots.  frames >= 2.                                                              */
int vslam_frontend_sequence(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                            int row_stride, const vslam_extract_params *params, int kp_stride,
                            const uint32_t *d_seeds, int hyp, float threshold, float *d_xy,
                            uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n, int32_t *d_matches,
                            int32_t *d_best, float *d_F);

/* ------------------------------------------------------------ batches in flight on one device
 * The reference's capture loop (src/vslam.cpp:53-77) finishes one frame pair before it looks at the next.  On the device
 * a batch runs through stages that fill the chip and stages of one workgroup per frame or pair that leave most of it
 * idle; the entry points are stream-ordered and a context owns its streams and workspaces, so a caller with a queue of
 * batches keeps k of them in flight on k contexts and the idle parts of one are filled by another (2.90 -> 2.61 ms per
 * batch at the headline shape with k = 4; contexts made here are arranged for company: DESIGN.md 6).  A TICKET is one batch:
 *   vslam_pipeline_acquire   next context, round-robin; waits for the batch that used it k tickets ago (at most k batches
 *                            are ever queued) and files that batch's status.  Enqueue the batch on the context returned --
 *                            any entry points of this header, uploads and vslam_gather_records included;
 *   vslam_pipeline_commit    closes the batch (its status = the context's device-side error word at this point of the
 *                            stream: copied and cleared in stream order, so one batch's overflow is reported for that
 *                            ticket only and does not reach the next batch of the same context);
 *   vslam_pipeline_submit_pairs / _sequence   acquire + vslam_frontend_pairs / _sequence (+ vslam_pack_records when
 *                            d_records is not NULL) + commit.  Output buffers are the caller's, one set per batch in flight.
 *                            An error of the wrapped call is returned at once, *ticket_out = -1, the slot stays usable;
 *   vslam_pipeline_wait      blocks until that batch is complete; returns its status (VSLAM_OK, VSLAM_ERR_CAPACITY, ...).
 *                            Statuses of failed batches nobody asked about are kept (the last 256);
 *   vslam_pipeline_poll      1 = complete, 0 = not yet (never blocks);
 *   vslam_pipeline_drain     waits for everything; the first failure not yet collected by vslam_pipeline_wait, else VSLAM_OK.
 * Inputs of a batch must be ordered before it: resident, or uploaded through the acquired context (vslam_upload_async +
 * vslam_upload_fence).  One submitting thread at a time (calls are serialised by a mutex).  n_ctx: 1 .. 16.           */
typedef struct vslam_pipeline vslam_pipeline;
int vslam_pipeline_create(int device, int n_ctx, vslam_pipeline **out);
int vslam_pipeline_destroy(vslam_pipeline *p);
int vslam_pipeline_size(const vslam_pipeline *p);
vslam_ctx *vslam_pipeline_ctx(vslam_pipeline *p, int slot);   /* ticket t runs on slot t % size */
const char *vslam_pipeline_last_error(vslam_pipeline *p);
int vslam_pipeline_set_option(vslam_pipeline *p, int option, int value);   /* vslam_ctx_set_option on every context */
int vslam_pipeline_acquire(vslam_pipeline *p, vslam_ctx **ctx_out, int64_t *ticket_out);
int vslam_pipeline_commit(vslam_pipeline *p, int64_t ticket);
int vslam_pipeline_submit_pairs(vslam_pipeline *p, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                                const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                int32_t *d_matches, int32_t *d_best, float *d_F, int32_t *d_records, int64_t *ticket_out);
/* vslam_frontend_pairs_pose as a ticket (h_K is copied; *pose is copied, the arrays it names are the caller's until the
 * ticket has been waited for).  A batch that exhausts the corner pool is queued once more, pose stages included. */
int vslam_pipeline_submit_pairs_pose(vslam_pipeline *p, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                                     const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                     float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                     int32_t *d_matches, int32_t *d_best, float *d_F, const float *h_K,
                                     const int32_t *d_map_point_ids, float reproj_threshold_sq, const vslam_pose_outputs *pose,
                                     int32_t *d_records, int64_t *ticket_out);
int vslam_pipeline_submit_sequence(vslam_pipeline *p, const uint8_t *d_bgr, int frames, int width, int height, int row_stride,
                                   const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                   float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                   int32_t *d_matches, int32_t *d_best, float *d_F, int32_t *d_records, int64_t *ticket_out);
int vslam_pipeline_poll(vslam_pipeline *p, int64_t ticket);
int vslam_pipeline_wait(vslam_pipeline *p, int64_t ticket);
int vslam_pipeline_drain(vslam_pipeline *p);
/* Batches that vslam_pipeline_submit_pairs / _sequence queued and the pipeline then queued a SECOND time by itself: a batch in
 * which more frames needed the corner detector's whole-image fallback than its pool holds (VSLAM_OPT_CORNER_LIST_CAP; pure
 * noise, response plateaus) is done again, when its status is collected, with every list sized for the whole image; its
 * outputs are then complete and the ticket reports VSLAM_OK.  (A ticket built with acquire / commit cannot be queued again --
 * the pipeline does not know what was enqueued -- and reports VSLAM_ERR_CAPACITY as before.)                                */
int64_t vslam_pipeline_batches_redone(vslam_pipeline *p);

/* ------------------------------------------------------------ several devices
 * Frame pairs are independent (src/RansacFilter.cpp:38: all state is per call), so a batch shards by contiguous slices,
 * one slice per device, with per-pair seeds base ^ GLOBAL pair index -- a pair's result does not depend on the split --
 * and the only exchange is the final gather of fixed-size result records (vslam_pack_records' layout).
 * vslam_shard_range: the slice [lo, hi) of `items` that `rank` of `world` owns (the first items % world ranks get one more). */
int vslam_shard_range(int items, int rank, int world, int *lo, int *hi);

/* (i) One process that owns the devices: one context + one host thread per entry of `devices` (an entry may repeat: several
 * contexts on one device).  vslam_multi_frontend_pairs = vslam_frontend_pairs + vslam_pack_records on every slice at once.
 *   h_bgr_last / h_bgr_cur: HOST, [pairs][height][row_stride] each (page-locked memory from vslam_host_alloc uploads faster);
 *   params->d_pattern must be NULL; h_pattern: HOST [256][4] int8 or NULL for ORB's learned table;
 *   h_records: HOST [pairs][13 + kp_stride] int32, pair order; h_n_keypoints: HOST [2 * pairs] (last frames, then current)
 *   or NULL.  Blocking: returns when every slice is done.  Options are per context: vslam_multi_ctx(m, i).            */
typedef struct vslam_multi vslam_multi;
int vslam_multi_create(const int *devices, int n_devices, vslam_multi **out);
int vslam_multi_destroy(vslam_multi *m);
int vslam_multi_size(const vslam_multi *m);
vslam_ctx *vslam_multi_ctx(vslam_multi *m, int i);
const char *vslam_multi_last_error(vslam_multi *m);
int vslam_multi_frontend_pairs(vslam_multi *m, const uint8_t *h_bgr_last, const uint8_t *h_bgr_cur, int pairs, int width,
                               int height, int row_stride, const vslam_extract_params *params, const int8_t *h_pattern,
                               int kp_stride, uint32_t base_seed, int hyp, float threshold, int32_t *h_records,
                               int32_t *h_n_keypoints);
/* Host frames reach each device in chunks of 64 pairs, chunk k + 1 uploading while chunk k computes; all the same a slot is
 * bound by its host link (57 GB/s page-locked = about 20 k pairs/s at 1280x720 against 90 k for the kernels).  When the
 * frames are ALREADY on the devices (decoded there, produced by an earlier stage, uploaded ahead of time):
 * d_bgr[r] = slot r's slice on slot r's device in vslam_frontend_pairs' layout -- the slice's `last` frames, then its
 * `current` frames, (hi - lo) of each for [lo, hi) = vslam_shard_range(pairs, r, size) -- NULL allowed for an empty slice.
 * Everything else as above: records and counts come back to host memory in pair order.                                 */
int vslam_multi_frontend_pairs_resident(vslam_multi *m, const uint8_t *const *d_bgr, int pairs, int width, int height,
                                        int row_stride, const vslam_extract_params *params, const int8_t *h_pattern,
                                        int kp_stride, uint32_t base_seed, int hyp, float threshold, int32_t *h_records,
                                        int32_t *h_n_keypoints);

/* (ii) One process per device (any launcher): every rank runs its slice on its own context; the records are exchanged once,
 * all-gather over RCCL (xGMI inside a node) on the context's stream.  RCCL is loaded when the first of these is called.
 *   vslam_comm_unique_id: rank 0 makes the 128-byte id and hands it to the other ranks by whatever channel the launcher
 *   has (a file, a socket, an environment variable, MPI); vslam_comm_create: collective over all ranks.
 *   vslam_gather_records: every rank contributes words_per_rank int32 (equal on all ranks: pad the last slice);
 *   d_all receives world x words_per_rank in rank order = pair order.  Asynchronous on the context's stream.          */
#define VSLAM_COMM_ID_BYTES 128
typedef struct vslam_comm vslam_comm;
int vslam_comm_unique_id(void *id_out);
int vslam_comm_create(vslam_ctx *ctx, const void *id, int world, int rank, vslam_comm **out);
int vslam_comm_destroy(vslam_comm *comm);
int vslam_gather_records(vslam_ctx *ctx, vslam_comm *comm, const int32_t *d_records, size_t words_per_rank,
                         int32_t *d_all);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), not what vslam_comm_create was told.  */
int vslam_comm_info(vslam_comm *comm, int *world_out, int *rank_out);
/* Uneven slices and the rooted form.  Rank r contributes h_words[r] int32 words (HOST array of `world` counts, the same on
 * every rank; 0 allowed); d_all receives them in rank order = pair order.  root < 0: every rank receives everything (an
 * all-gather with counts); root >= 0: only that rank does -- each peer sends its block once, straight to the root, over its
 * own xGMI link (d_all may be NULL on the others).  One group of ncclSend / ncclRecv on the context's stream. */
int vslam_gather_records_v(vslam_ctx *ctx, vslam_comm *comm, const int32_t *d_records, const size_t *h_words, int root,
                           int32_t *d_all);

#ifdef __cplusplus
}
#endif
#endif

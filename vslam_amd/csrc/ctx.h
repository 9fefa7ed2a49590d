// Context, error plumbing, workspace arena and per-kernel event timing shared by the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/vslam_amd.h"

struct vslam_prof_pending {
    int slot;
    hipEvent_t start, stop;
};

struct vslam_prof_slot {
    std::string name;
    double total_ms = 0;
    int64_t launches = 0;
};

struct vslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream + fork/join events: independent stages (the k-d tree build is not an input of
    // match/RANSAC) run beside the main stream inside vslam_frontend_pairs
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // uploads (frame ingest) run on their own stream; ev_upload marks the last one enqueued
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_upload = nullptr;
    // raw mt19937 outputs produced ahead of time on the auxiliary stream (vslam_frontend_pairs / _sequence): valid for
    // exactly the (seeds, batch, hyp) recorded here until vslam_match_features consumes them
    hipEvent_t ev_raw = nullptr;
    const uint32_t *raw_seeds = nullptr;
    int raw_batch = 0, raw_hyp = 0;
    // An independent stage (the k-d build: an output of the path, nobody's input) that vslam_frontend_pairs / _sequence hand
    // to the matching stages to be forked onto the auxiliary stream at the point where it disturbs them least:
    // aux_job_at = 1 after the matcher, 2 after the set mapping, 3 after the solves, 4 after the screen (0: at once / none).  vs_aux_job_point(ctx, k)
    // launches it when k is that point; the caller joins on ev_join.
    std::function<int()> aux_job;
    int aux_job_at = 0;
    int tree_fork = -1;   // VSLAM_OPT_TREE_FORK: where the batched front-end forks the k-d build (-1: by size; 0 in front of the matcher)
    int overlap_blur = 2;       // VSLAM_OVERLAP_BLUR: 0 blur on the main stream, otherwise on the auxiliary stream from min_eigen (where the gray image is complete) on
    // One of several contexts with batches in flight on this device (vs_ctx_create): blur on the main stream, the auxiliary
    // stream at the main stream's priority, streams it may never use made on first use.
    bool shared_chip = false;
    bool lazy_streams = false;   // the copy stream is made on first use
    int solve_split = 0;        // VSLAM_RANSAC_SOLVE_SPLIT (read when the context is made): 0 one solve kernel, 4 / 5 sweeps + closing kernel
    bool sets_prefetch = true;  // VSLAM_SETS_PREFETCH: the raw mt19937 outputs generated ahead of time on the auxiliary stream
    bool fork_after_eigen = false;   // transient: good_features records ev_fork once the response kernel is queued
    bool rbrief_table_ready = false; // transient: the rotated rBRIEF table of the coming describe call is already queued
    int img_pitch = 0;               // transient (vslam_extract_features): bytes per row of the call's internal gray / blurred planes when that is
                                     // not the width -- a width that is no multiple of 4 gets rows of a multiple of 16 bytes whose tail holds the
                                     // REFLECT_101 continuation of the row, so that the dword kernels take it; 0: rows are `width` bytes
    int ransac_min_matches = VSLAM_SET_SIZE;   // VSLAM_OPT_RANSAC_MIN_MATCHES
    int ransac_min_items = VSLAM_SET_SIZE;     // VSLAM_OPT_RANSAC_MIN_ITEMS: indices drawn per 8-wide set
    int ransac_solver = 0;                      // VSLAM_OPT_RANSAC_SOLVER: 0 exact Jacobi replay, 1 Gram / MFMA (not bit-exact)
    int match_shape = 0;                        // VSLAM_OPT_MATCH_SHAPE: 0 by size, 1 = 8 waves x 32 rows, 2 = 4 waves x 64 rows
    int match_form = 0;                         // VSLAM_OPT_MATCH_FORM: 0 default (FP4), 1 FP4, 2 int8
    int corner_window_pct = 135;                // VSLAM_OPT_CORNER_WINDOW_PCT
    int corner_list_cap = 0;                    // VSLAM_OPT_CORNER_LIST_CAP: 0 = 16 x max_corners + 4096, -1 = whole image
    bool ransac_all_sums = false;   // VSLAM_OPT_RANSAC_ALL_SUMS: exact residual sum of every hypothesis (ransac_score_kernel)
    // where the corner detector's last batch left its per-frame counters (vslam_corner_stats reads them after a stream wait)
    const uint32_t *stat_counts = nullptr;
    const int32_t *stat_pool_count = nullptr;
    int stat_frames = 0, stat_pool_slots = 0;
    size_t stat_px = 0;
    std::string err;

    // named, grow-only device buffers (never allocated inside a timed loop after warm-up)
    struct Buf {
        void *ptr = nullptr;
        size_t bytes = 0;
    };
    std::map<std::string, Buf> arena;
    std::map<std::string, bool> attr_done;   // hipFuncSetAttribute is per device: remembered per context

    bool prof = false;
    std::vector<vslam_prof_slot> prof_slots;
    std::map<std::string, int> prof_index;
    std::vector<vslam_prof_pending> prof_pending;
    std::vector<hipEvent_t> event_pool;
};

#define VS_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                \
            return VSLAM_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define VS_REQUIRE(ctx, cond, code)                                                         \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            if (ctx) (ctx)->err = std::string("requirement failed: ") + #cond;              \
            return (code);                                                                  \
        }                                                                                   \
    } while (0)

// sticky device-side error word (bit 0: a fixed-size candidate list overflowed); vslam_ctx_synchronize
// reads and clears it and reports VSLAM_ERR_CAPACITY
int vs_device_errflag(vslam_ctx *ctx, int32_t **out);
int vs_ctx_create(int device, bool shared_chip, vslam_ctx **out);   // vslam_ctx_create = (device, false, out)
std::string vs_errflag_message(int32_t flag);

// fork the pending auxiliary job (if it was asked for at `point`) onto the auxiliary stream behind everything queued so far
int vs_aux_job_point(vslam_ctx *ctx, int point);

// grow-only named workspace
int vs_arena_get(vslam_ctx *ctx, const char *name, size_t bytes, void **out);

// event bracket around one kernel launch when profiling is on
struct VsProfScope {
    vslam_ctx *ctx;
    vslam_prof_pending p;
    bool active;
    VsProfScope(vslam_ctx *c, const char *name);
    ~VsProfScope();
};

static inline int vs_div_up(int a, int b) { return (a + b - 1) / b; }
// bytes per row of the internal gray / blurred planes of the running extract call (vslam_ctx::img_pitch)
static inline int vs_pitch(const vslam_ctx *ctx, int w) { return ctx->img_pitch > 0 ? ctx->img_pitch : w; }

// A/B switches (slower kernel variants, stream arrangements, tile shapes measured and not chosen) exist only in the
// EXPERIMENTS build of the library (-DVSLAM_EXPERIMENTS -> libvslam_amd_exp.so, which tools/ab_*.py and the variant tests load
// through VSLAM_AMD_LIB / capi.load_library): the default build reads no environment variable and carries one kernel per stage.
#ifdef VSLAM_EXPERIMENTS
#include <cstdlib>
#define VS_EXPERIMENT_ENV(name) getenv(name)
#else
#define VS_EXPERIMENT_ENV(name) (static_cast<const char *>(nullptr))
#endif

#ifdef __HIPCC__
// Workgroups are dealt round-robin over the 8 XCDs (linear id L lands on XCD L % 8 — observed, used
// for speed only, never for correctness).  Remap a 1-D grid of ceil(items/8)*8*per_item blocks so that
// all `per_item` blocks of one item (frame / pair) share an XCD and therefore its 4 MiB L2.
__device__ __forceinline__ void vs_xcd_item_block(int linear, int per_item, int &item, int &blk) {
    const int xcd = linear & 7, slot = linear >> 3;
    item = ((slot / per_item) << 3) + xcd;
    blk = slot - (slot / per_item) * per_item;
}
#endif
static inline int vs_xcd_grid(int items, int per_item) { return vs_div_up(items, 8) * 8 * per_item; }

// Row segments per column strip for the streaming image kernels: about 90 rows per wave (6 warm-up rows each,
// 7 %), more and shorter segments (down to 24 rows) when the batch alone would leave the 1024 SIMDs with fewer
// than four waves each.
static inline int vs_stream_segments(int h, int frames, int strips) {
    int segs = h >= 135 ? (h + 45) / 90 : 1;
    const long long waves = (long long)frames * strips * segs;
    if (waves < 4096) {
        const long long want = (4096 + (long long)frames * strips - 1) / ((long long)frames * strips);
        const int most = h / 24 > 1 ? h / 24 : 1;
        segs = (int)(want < most ? want : most);
        if (segs < 1) segs = 1;
    }
    return segs;
}

// ---- stage launchers implemented in the .hip files (all async on ctx->stream) ----
int vs_launch_match(vslam_ctx *ctx, const uint8_t *d1, const int32_t *n1, const uint8_t *d2,
                    const int32_t *n2, int batch, int kp_stride, int32_t *pairs, int32_t *m,
                    int32_t *knn);
int vs_launch_ransac_sets(vslam_ctx *ctx, const uint32_t *seeds, const int32_t *m, int batch, int hyp,
                          int32_t *sets, uint32_t *draws);
// the same sets in two steps: raw mt19937 outputs (seed only; can run ahead on another stream), then the mapping
size_t vs_ransac_raw_words(int hyp);   // uint32 words of raw outputs per item
int vs_launch_ransac_mt(vslam_ctx *ctx, const uint32_t *seeds, int batch, int hyp, uint32_t *raw);
int vs_launch_ransac_map(vslam_ctx *ctx, const int32_t *m, int batch, int hyp, const uint32_t *raw, int32_t *sets,
                         uint32_t *draws);
int vs_launch_ransac(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                     const int32_t *m, const int32_t *sets, int batch, int kp_stride, int hyp,
                     float threshold, float *F, uint8_t *mask, int32_t *best, int32_t *matches,
                     float *hypF, int32_t *hyp_count, float *hyp_sum);
int vs_launch_ransac_solve(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                           const int32_t *m, const int32_t *sets, int batch, int kp_stride, int hyp,
                           float *hypF);
int vs_launch_ransac_evaluate(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                              const int32_t *m, const float *hypF, int batch, int kp_stride, int hyp,
                              float threshold, float *F, uint8_t *mask, int32_t *best, int32_t *matches,
                              int32_t *hyp_count, float *hyp_sum);
int vs_launch_kdtree_build(vslam_ctx *ctx, const float *xy, const int32_t *n, int batch, int kp_stride,
                           int32_t *nodes);
int vs_launch_kdtree_radius(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n,
                            int batch, int kp_stride, const float *queries, const int32_t *nq,
                            int q_stride, float radius, int32_t *hits, int32_t *counts, int hit_cap);
int vs_launch_kdtree_nearest(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n,
                             int batch, int kp_stride, const float *queries, const int32_t *nq,
                             int q_stride, float max_distance_sq, int32_t *best_idx);
int vs_launch_kdtree_cell_table(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n, int batch,
                                int kp_stride, int slots, uint32_t *table, int32_t *ok);
int vs_launch_bgr2gray(vslam_ctx *ctx, const uint8_t *bgr, int frames, int w, int h, int stride,
                       uint8_t *gray);
int vs_launch_min_eigen(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, float *eig,
                        uint32_t *frame_max_bits);
// per-frame counters of the corner pipeline: one zero-initialised block (select.hip lays it out)
struct VsCornerCounters {
    uint32_t *counts;    // entries in the detector's list
    uint32_t *fmax;      // ordered exact maximum response
    uint32_t *low;       // two-tier detector: certified lower bound of maximum / c0 (ordered)
    uint32_t *count2;    // exact keys above the cut
    uint32_t *count3;    // exact keys of the rerun (flagged frames, nothing cut)
    uint32_t *need;      // the selection ran out of candidates above the cut: rerun this frame with everything
    uint32_t *cutkey;    // ordered response below which keys2 is incomplete (0: complete)
    uint32_t *hist;      // two-tier detector: the listed upper bounds by magnitude
};
// Whole-image scratch for the few frames the bounded two-tier path cannot finish (see vs_launch_good_features): `slots`
// sets of (keys, response image, per-pixel state).  A frame claims a slot by a ticket on `count`; everything per slot.
struct VsCornerPool {
    int32_t *count = nullptr;      // tickets handed out (may exceed slots: the excess frames raise bit 2 of the error word)
    int32_t *frame = nullptr;      // [slots] the frame filed under each slot
    uint32_t *counts = nullptr;    // [slots] candidates found
    uint32_t *fmax = nullptr;      // [slots] ordered exact maximum response
    unsigned long long *keys = nullptr;   // [slots][key_cap]
    float *eig = nullptr;          // [slots][w * h]
    uint8_t *state = nullptr;      // [slots][w * h]
    size_t key_cap = 0;
    int slots = 0;
};
#ifdef __HIPCC__
__device__ __forceinline__ int vs_pool_used(const VsCornerPool &p) {
    const int used = *p.count;
    return used < p.slots ? used : p.slots;
}
#endif
int vs_launch_pool_candidates(vslam_ctx *ctx, const uint8_t *gray, int w, int h, double quality, const VsCornerPool &pool);
size_t vs_response_hist_words(int frames);
// `gray` still to be formed from a 3-byte image (cvtColor is then the detector's job: fused into its first kernel when
// the layout allows, a launch of its own otherwise)
struct VsBgrSource {
    const uint8_t *data;
    int stride;   // bytes per row
};
int vs_launch_response_candidates(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, double quality,
                                  float *eig, const VsCornerCounters &c, unsigned long long *keys,
                                  unsigned long long *keys2, size_t key_cap, uint32_t n_safe, int *raw_list,
                                  const VsBgrSource *bgr = nullptr);
int vs_launch_corner_exact(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, const VsCornerCounters &c,
                           const unsigned long long *keys, unsigned long long *keys2, size_t key_cap, uint32_t n_safe,
                           int mode);
int vs_launch_good_features(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h,
                            int max_corners, double quality, double min_distance, int kp_stride,
                            float *xy, int32_t *n, const VsBgrSource *bgr = nullptr);
int vs_launch_gaussian7(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, uint8_t *out);
int vs_launch_gray_pad(vslam_ctx *ctx, const uint8_t *src, int frames, int w, int h, uint8_t *dst, int pitch);
int vs_launch_gray_unpad(vslam_ctx *ctx, const uint8_t *src, int frames, int w, int h, int pitch, uint8_t *dst);
int vs_launch_gaussian7_rows(vslam_ctx *ctx, const uint8_t *src, uint8_t *dst, int frames, size_t frame_pitch, int w, int h,
                             int y_begin, int y_end);
int vs_launch_rbrief_rotate(vslam_ctx *ctx, const int8_t *pattern, float ca, float sa);
int vs_launch_orb_describe(vslam_ctx *ctx, const uint8_t *blurred, int frames, int w, int h,
                           const float *xy_in, const int32_t *n_in, int kp_stride, float ca, float sa,
                           const int8_t *pattern, float *xy_out, uint8_t *desc, int32_t *n_out);
int vs_launch_extract_grid(vslam_ctx *ctx, uint8_t *bgr, int frames, int w, int h, int stride, int nrows, int ncols,
                           const int8_t *pattern, int kp_cap, float *xy, uint8_t *desc, float *angle_octave,
                           int32_t *n_out);
int vs_launch_extract_Rt(vslam_ctx *ctx, const float *F, const int32_t *best, int batch, const float *h_K, float *R,
                         float *t, float *c2);
int vs_launch_triangulate(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *matches,
                          const int32_t *best, int batch, int kp_stride, const float *h_K, const float *c2,
                          float *points4d);
int vs_launch_triangulate_points(vslam_ctx *ctx, const float *p1, const float *p2, int n, const float *h_c1, const float *h_c2,
                                 float *points4d);
int vs_launch_associate(vslam_ctx *ctx, const float *map_points, const int32_t *n_map, int batch, int map_stride,
                        const float *c2, int img_w, int img_h, const int32_t *nodes, const float *xy, const uint8_t *desc,
                        const int32_t *n_kp, int kp_stride, const int32_t *obs_offsets, const uint8_t *obs_desc,
                        int obs_stride, float radius, uint32_t dist_threshold, int32_t *map_point_ids, int32_t *claim);
int vs_launch_reproj_filter(vslam_ctx *ctx, const float *points4d, const float *xy1, const float *xy2, const int32_t *matches,
                            const int32_t *best, int batch, int kp_stride, const float *h_K, const float *c2,
                            const int32_t *map_point_ids, float threshold_sq, int32_t *out_idx, int32_t *out_n,
                            double *out_err);

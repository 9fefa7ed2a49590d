// 2-D k-d tree for gfx950: batched build and batched radius query.
//
// Replaces construct_kdtree(frame_kdtree&, points) (/root/reference/src/KDTree.cpp:107-143) and
// radius_search(frame_kdtree, points, query, radius) (:145-171).
//
// Layout: the reference appends nodes to one malloc'd array before recursing, so that array is
// the tree in PRE-ORDER, and because the split is always at l + len/2 the shape depends on n
// only: a subtree of len nodes at position p has its left child at p+1 (len/2 nodes) and its
// right child at p+1+len/2 (len - len/2 - 1 nodes).  The device tree is therefore just the
// pt_index column, int32 [n] — 4 B/node instead of 24 — and child links are arithmetic.
//
// Build: one workgroup per frame.  (x, y, index) triples live in LDS; the recursion is run level
// by level (all subtrees of one depth are independent): a wave per subtree while the subtrees have
// 48 points or more (introselect.h: the partition as two ballot scans and a parallel swap), a lane
// per subtree below — either way libstdc++'s introselect replayed step for step, so tied
// coordinates land exactly where std::nth_element puts them.  The top levels are serial by nature
// (depth 0 is one 2000-element selection); 512 frames per batch keep the other CUs busy, and a
// single frame gets a 16-wave workgroup (see the launcher).
//
// Query: one lane per query, explicit stack in LDS, hits appended in visit order (node, left,
// right), which is the order vslam.cpp:150-158 observes through its `break`.
#include "ctx.h"
#include "introselect.h"

namespace {


struct Triple {
    float x, y;
    int id;
};

template <class I>   // I: how point indices are kept in LDS (int, or uint16_t: they are below 16384)
struct LdsStore {
    using value_type = Triple;
    using key_type = float;
    float *kx, *ky;
    I *id;
    const float *kaxis;   // kx or ky: the coordinate this level splits on
    bool use_y;
    __device__ LdsStore(float *x, float *y, I *i, int axis) : kx(x), ky(y), id(i), kaxis(axis ? y : x), use_y(axis != 0) {}
    __device__ Triple get(int i) const { return Triple{kx[i], ky[i], (int)id[i]}; }
    __device__ void set(int i, const Triple &t) {
        kx[i] = t.x;
        ky[i] = t.y;
        id[i] = (I)t.id;
    }
    __device__ void swap(int i, int j) {
        const Triple a = get(i), b = get(j);
        set(i, b);
        set(j, a);
    }
    __device__ float key(int i) const { return kaxis[i]; }
    __device__ float key_of(const Triple &t) const { return use_y ? t.y : t.x; }
    // P(points[i1], axis) < P(points[i2], axis), src/KDTree.cpp:128
    __device__ bool less(float a, float b) const { return a < b; }
};

// T = threads per workgroup; ranges shorter than MINLEN: one lane per subtree is cheaper than a wave each
template <int T, int MINLEN, class I>
__global__ __launch_bounds__(T) void kdtree_build_kernel(const float *__restrict__ xy,
                                                                     const int32_t *__restrict__ n_arr,
                                                                     int kp_stride,
                                                                     int32_t *__restrict__ nodes) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = n_arr[b];
    if (n <= 0) return;   // root = NULL, src/KDTree.cpp:109-110
    float *kx = reinterpret_cast<float *>(smem);
    float *ky = kx + kp_stride;
    I *id = reinterpret_cast<I *>(ky + kp_stride);
    I *sl = id + kp_stride;              // stopper lists of the wave-cooperative partition
    I *sr = sl + kp_stride + 1;
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)b * kp_stride;
    int32_t *out = nodes + (size_t)b * kp_stride;
    for (int i = tid; i < n; i += T) {
        const float2 p = P[i];
        kx[i] = p.x;
        ky[i] = p.y;
        id[i] = (I)i;   // point_indices = 0..N-1, :113-117
    }
    __syncthreads();

    int height = 0;
    for (int t = n; t > 0; t >>= 1) height++;   // floor(log2 n) + 1, :119
    const int wave = tid >> 6, lane = tid & 63;
    for (int depth = 0; depth < height; depth++) {
        const int tasks = 1 << depth;
        const bool by_wave = (n >> depth) >= MINLEN;   // subtree sizes at this depth are n/2^depth (+-1)
        const int step = by_wave ? T / 64 : T;
        for (int t = by_wave ? wave : tid; t < tasks; t += step) {
            int first = 0, last = n, pos = 0;
            for (int bit = depth - 1; bit >= 0 && last > first; bit--) {
                const int len = last - first, mid = first + len / 2;
                if ((t >> bit) & 1) {
                    pos += 1 + len / 2;
                    first = mid + 1;
                } else {
                    pos += 1;
                    last = mid;
                }
            }
            if (last > first) {
                LdsStore<I> s(kx, ky, id, depth & 1);
                const int mid = first + (last - first) / 2;
                if (by_wave) {
                    vs_sel::wave_nth_element(s, first, mid, last, sl, sr);
                    if (lane == 0) out[pos] = (int32_t)id[mid];
                } else {
                    vs_sel::nth_element(s, first, mid, last);
                    out[pos] = (int32_t)id[mid];
                }
            }
        }
        __syncthreads();
    }
}

constexpr int kQueryThreads = 256;
constexpr int kStackDepth = 18;

__global__ __launch_bounds__(kQueryThreads) void kdtree_radius_kernel(
    const int32_t *__restrict__ nodes, const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
    int kp_stride, const float *__restrict__ queries, const int32_t *__restrict__ nq_arr, int q_stride,
    float radius, int32_t *__restrict__ hits, int32_t *__restrict__ counts, int hit_cap) {
    __shared__ uint32_t stack[kStackDepth * kQueryThreads];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int q = blockIdx.x * kQueryThreads + tid;
    const int n = n_arr[b];
    if (q >= nq_arr[b]) return;
    const float2 qp = reinterpret_cast<const float2 *>(queries)[(size_t)b * q_stride + q];
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)b * kp_stride;
    const int32_t *T = nodes + (size_t)b * kp_stride;
    int32_t *H = hits + ((size_t)b * q_stride + q) * hit_cap;
    const float radius_sq = radius * radius;   // SQ(radius), :146

    int cnt = 0, sp = 0;
    // stack entry: pos (15 bits) | len (15 bits) << 15 | axis << 30
    if (n > 0) stack[(sp++) * kQueryThreads + tid] = 0u | ((uint32_t)n << 15);
    while (sp > 0) {
        const uint32_t e = stack[(--sp) * kQueryThreads + tid];
        const int pos = (int)(e & 0x7FFFu), len = (int)((e >> 15) & 0x7FFFu), axis = (int)(e >> 30);
        const int idx = T[pos];
        const float2 pt = P[idx];
        const float split = (axis == 0 ? qp.x : qp.y) - (axis == 0 ? pt.x : pt.y);   // :156
        const int nl = len / 2, nr = len - nl - 1;
        const uint32_t nax = (uint32_t)(1 - axis) << 30;
        const uint32_t le = (uint32_t)(pos + 1) | ((uint32_t)nl << 15) | nax;
        const uint32_t re = (uint32_t)(pos + 1 + nl) | ((uint32_t)nr << 15) | nax;
        const float abs_split = (split > 0) ? split : -split;   // ABS macro
        if (abs_split <= radius) {                              // inclusive, :158
            const float dx = qp.x - pt.x, dy = qp.y - pt.y;
            const float d2 = dx * dx + dy * dy;
            if (d2 < radius_sq) {                               // strict, :161
                if (cnt < hit_cap) H[cnt] = idx;
                cnt++;
            }
            if (nr > 0) stack[(sp++) * kQueryThreads + tid] = re;   // right is visited after left
            if (nl > 0) stack[(sp++) * kQueryThreads + tid] = le;
        } else if (split < 0) {
            if (nl > 0) stack[(sp++) * kQueryThreads + tid] = le;
        } else {
            if (nr > 0) stack[(sp++) * kQueryThreads + tid] = re;
        }
    }
    counts[(size_t)b * q_stride + q] = cnt;
}

// nearest(KDTree, query, max_distance_sq), src/KDTree.cpp:37-71: descend the near side, test the
// node on the way back (strict <), visit the far side iff split^2 < best.  Entry = pos | len << 15 |
// axis << 30 | stage << 31 (stage 1 = "near side done").  Result: point index or -1 when nothing
// beats max_distance_sq (the reference then returns a default-constructed {0,0}, :38-42).
__global__ __launch_bounds__(kQueryThreads) void kdtree_nearest_kernel(
    const int32_t *__restrict__ nodes, const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
    int kp_stride, const float *__restrict__ queries, const int32_t *__restrict__ nq_arr, int q_stride,
    float max_distance_sq, int32_t *__restrict__ best_idx) {
    __shared__ uint32_t stack[kStackDepth * kQueryThreads];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int q = blockIdx.x * kQueryThreads + tid;
    const int n = n_arr[b];
    if (q >= nq_arr[b]) return;
    const float2 qp = reinterpret_cast<const float2 *>(queries)[(size_t)b * q_stride + q];
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)b * kp_stride;
    const int32_t *T = nodes + (size_t)b * kp_stride;
    float best = max_distance_sq;
    int best_i = -1, sp = 0;
    if (n > 0) stack[(sp++) * kQueryThreads + tid] = 0u | ((uint32_t)n << 15);
    while (sp > 0) {
        const uint32_t e = stack[(--sp) * kQueryThreads + tid];
        const int pos = (int)(e & 0x7FFFu), len = (int)((e >> 15) & 0x7FFFu), axis = (int)((e >> 30) & 1u);
        const bool back = (e >> 31) != 0;
        const int idx = T[pos];
        const float2 pt = P[idx];
        const float split = (axis == 0 ? qp.x : qp.y) - (axis == 0 ? pt.x : pt.y);
        const int nl = len / 2, nr = len - nl - 1;
        const uint32_t nax = (uint32_t)(1 - axis) << 30;
        const uint32_t le = (uint32_t)(pos + 1) | ((uint32_t)nl << 15) | nax;
        const uint32_t re = (uint32_t)(pos + 1 + nl) | ((uint32_t)nr << 15) | nax;
        const bool near_left = split < 0;
        if (!back) {
            stack[(sp++) * kQueryThreads + tid] = e | 0x80000000u;
            if (near_left ? nl > 0 : nr > 0) stack[(sp++) * kQueryThreads + tid] = near_left ? le : re;
        } else {
            const float dx = pt.x - qp.x, dy = pt.y - qp.y;
            const float cur = dx * dx + dy * dy;
            if (cur < best) {
                best = cur;
                best_i = idx;
            }
            if (split * split < best && (near_left ? nr > 0 : nl > 0))
                stack[(sp++) * kQueryThreads + tid] = near_left ? re : le;
        }
    }
    best_idx[(size_t)b * q_stride + q] = best_i;
}

}  // namespace

// The tree's points filed by integer pixel cell (floor x, floor y) in an open-addressing table, one slot per point:
// slot = {cell key, pre-order position}.  A radius query then only has to look at the (2 r + 2)^2 cells around it — a
// handful of host-side probes for the reference's r = 2 (src/vslam.cpp:149) — and sorting the hits by pre-order position
// gives the reference's visit order (src/KDTree.cpp:151-171 pushes a node before its subtrees, left before right, and
// every point with d^2 < r^2 is visited: a subtree is skipped only when it lies beyond r along the split axis).
// ok[b] = 0 if a coordinate falls outside the key range (the caller then does not use the table).
__global__ __launch_bounds__(256) void kdtree_cell_table_kernel(const int32_t *__restrict__ nodes, const float *__restrict__ xy,
                                                                const int32_t *__restrict__ n_arr, int kp_stride, int slots,
                                                                uint32_t *__restrict__ table, int32_t *__restrict__ ok) {
    const int b = blockIdx.y, rank = blockIdx.x * 256 + threadIdx.x;
    if (rank >= n_arr[b]) return;
    const float2 pt = reinterpret_cast<const float2 *>(xy)[(size_t)b * kp_stride + nodes[(size_t)b * kp_stride + rank]];
    const float fx = floorf(pt.x), fy = floorf(pt.y);
    if (!(fx >= -32768.f && fx <= 32766.f && fy >= -32768.f && fy <= 32766.f)) {   // also NaN
        ok[b] = 0;
        return;
    }
    const uint32_t key = ((uint32_t)((int)fy + 32768) << 16) | (uint32_t)((int)fx + 32768);
    uint32_t *T = table + (size_t)b * slots * 2;
    uint32_t slot = (key * 2654435761u) >> 7;
    while (true) {
        slot &= (uint32_t)slots - 1u;
        if (atomicCAS(&T[2 * slot], 0xFFFFFFFFu, key) == 0xFFFFFFFFu) break;   // every point takes a slot of its own
        slot++;
    }
    T[2 * slot + 1] = (uint32_t)rank;
}

int vs_launch_kdtree_cell_table(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n, int batch,
                                int kp_stride, int slots, uint32_t *table, int32_t *ok) {
    VS_REQUIRE(ctx, nodes && xy && n && table && ok, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, slots >= 2 * kp_stride && (slots & (slots - 1)) == 0, VSLAM_ERR_INVALID);   // power of two, load <= 1/2
    VS_HIP(ctx, hipMemsetAsync(table, 0xFF, sizeof(uint32_t) * 2 * (size_t)slots * batch, ctx->stream));
    VS_HIP(ctx, hipMemsetAsync(ok, 0x01, sizeof(int32_t) * (size_t)batch, ctx->stream));   // non-zero = usable
    VsProfScope ps(ctx, "kdtree_cell_table_kernel");
    dim3 grid(vs_div_up(kp_stride, 256), batch);
    kdtree_cell_table_kernel<<<grid, 256, 0, ctx->stream>>>(nodes, xy, n, kp_stride, slots, table, ok);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_kdtree_nearest(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n,
                             int batch, int kp_stride, const float *queries, const int32_t *nq,
                             int q_stride, float max_distance_sq, int32_t *best_idx) {
    VS_REQUIRE(ctx, nodes && xy && n && queries && nq && best_idx, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0 && q_stride > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    VsProfScope ps(ctx, "kdtree_nearest_kernel");
    dim3 grid(vs_div_up(q_stride, kQueryThreads), batch);
    kdtree_nearest_kernel<<<grid, kQueryThreads, 0, ctx->stream>>>(nodes, xy, n, kp_stride, queries, nq, q_stride,
                                                                   max_distance_sq, best_idx);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_kdtree_build(vslam_ctx *ctx, const float *xy, const int32_t *n, int batch, int kp_stride,
                           int32_t *nodes) {
    VS_REQUIRE(ctx, xy && n && nodes, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    // Point indices and the partition's stopper lists as 16-bit values (keypoint slots are below VSLAM_MAX_KP = 16384): 14
    // instead of 20 bytes of LDS per point.  At 4000 keypoints that is 56 KB per tree, so two builds share a CU where
    // 80 KB allowed one (VSLAM_KD_IDX16=0 / 1 forces either form; same trees).
    static const char *const idx_env = VS_EXPERIMENT_ENV("VSLAM_KD_IDX16");
    const bool idx16 = idx_env ? idx_env[0] == '1' : true;
    const size_t lds = idx16 ? (((size_t)kp_stride * 8 + (size_t)kp_stride * 2 * 3 + 4 + 15) & ~(size_t)15) : (size_t)kp_stride * 20 + 8;
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    VS_REQUIRE(ctx, lds <= 160 * 1024 - 512, VSLAM_ERR_CAPACITY);
    // Workgroup size by batch: with few trees in flight (the one-frame-at-a-time drop-in use) sixteen waves take the 8, 16
    // and 32 subtrees of depths 3-5 in one or two rounds instead of up to eight (one 2000-point tree: 0.24 -> 0.17 ms);
    // beside the matcher of a large batch the extra waves cost the step more than the build gains (2.92 -> 3.07 ms at C3).
    static const char *const shape_env = VS_EXPERIMENT_ENV("VSLAM_KD_THREADS");   // 256 / 1024 force one (A/B timing)
    const int threads = shape_env ? atoi(shape_env) : (batch <= 32 ? 1024 : 256);
    VsProfScope ps(ctx, "kdtree_build_kernel");
#define VS_KD_LAUNCH(T, M, I)                                                                                               \
    do {                                                                                                                    \
        if (!ctx->attr_done["kdtree_build" #T "_" #M #I]) {                                                                 \
            VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kdtree_build_kernel<T, M, I>),                   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));                 \
            ctx->attr_done["kdtree_build" #T "_" #M #I] = true;                                                             \
        }                                                                                                                   \
        kdtree_build_kernel<T, M, I><<<batch, T, lds, ctx->stream>>>(xy, n, kp_stride, nodes);                              \
    } while (0)
#ifdef VSLAM_EXPERIMENTS
    if (threads == 1024 && !idx16) VS_KD_LAUNCH(1024, 48, int);
    else if (threads != 1024 && !idx16) VS_KD_LAUNCH(256, 48, int);
    else
#endif
    if (threads == 1024) VS_KD_LAUNCH(1024, 48, uint16_t);
    else VS_KD_LAUNCH(256, 48, uint16_t);
#undef VS_KD_LAUNCH
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_kdtree_radius(vslam_ctx *ctx, const int32_t *nodes, const float *xy, const int32_t *n,
                            int batch, int kp_stride, const float *queries, const int32_t *nq,
                            int q_stride, float radius, int32_t *hits, int32_t *counts, int hit_cap) {
    VS_REQUIRE(ctx, nodes && xy && n && queries && nq && hits && counts, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0 && q_stride > 0 && hit_cap > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    VsProfScope ps(ctx, "kdtree_radius_kernel");
    dim3 grid(vs_div_up(q_stride, kQueryThreads), batch);
    kdtree_radius_kernel<<<grid, kQueryThreads, 0, ctx->stream>>>(nodes, xy, n, kp_stride, queries, nq,
                                                                  q_stride, radius, hits, counts, hit_cap);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

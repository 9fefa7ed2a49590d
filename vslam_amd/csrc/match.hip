// Brute-force Hamming k=2 matching + Lowe ratio for gfx950.
//
// Replaces the front half of match_features, /root/reference/src/Frame.cpp:83-94
// (cv::BFMatcher(NORM_HAMMING)->knnMatch(k=2), then `m[0].distance < m[1].distance * 0.7`).
//
// Mapping: one 256-thread workgroup owns 512 query rows of one frame pair (2 per lane, kept in
// 16 VGPRs); train rows stream through LDS in 8 KiB tiles (coalesced 16-B global loads) and are
// read back as wave-uniform ds_read_b128 broadcasts.  Per (query, train) pair the work is
// 8 x v_xor_b32 + 8 x v_bcnt_u32_b32 (accumulating) + one key pack + a 2-op running min-2 (min, med3),
// i.e. the kernel is integer-VALU bound (about 450 int ops per HBM byte), not HBM bound.
// The two best candidates per query are tracked as packed keys (distance << 16 | trainIdx):
// an unsigned min over keys is exactly "smaller distance, then lower train index", which is
// knnMatch's tie rule.  The ratio test is the integer form 10*d0 < 7*d1, equal to the
// reference's float/double comparison for every 0 <= d0 <= d1 <= 256 (tests/test_match.py
// checks that identity exhaustively).
#include "ctx.h"

namespace {

constexpr int kThreads = 256;
constexpr int kQueriesPerLane = 2;
constexpr int kTile = 256;   // train rows per LDS tile (8 KiB)

// v_bcnt_u32_b32 d, x, acc = popcount(x) + acc: the accumulate form keeps a 256-bit Hamming distance at
// 8 xor + 8 bcnt (hipcc otherwise emits bare popcounts plus an add3 tree).  Plain asm (no memory, no
// waits): a pure register instruction.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// second-smallest tracking: with k1 <= k2, the new k2 is the median of (k1, k2, key)
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t ham256(const uint4 &qa, const uint4 &qb, const uint4 &ta,
                                           const uint4 &tb) {
    uint32_t d = bcnt_acc(qa.x ^ ta.x, 0u);
    d = bcnt_acc(qa.y ^ ta.y, d);
    d = bcnt_acc(qa.z ^ ta.z, d);
    d = bcnt_acc(qa.w ^ ta.w, d);
    d = bcnt_acc(qb.x ^ tb.x, d);
    d = bcnt_acc(qb.y ^ tb.y, d);
    d = bcnt_acc(qb.z ^ tb.z, d);
    d = bcnt_acc(qb.w ^ tb.w, d);
    return d;
}

// NQ = query rows per lane in this workgroup (2, or 1 when the second slot would hold no valid query: the last
// workgroup of a pair then issues half the instructions instead of computing clamped duplicates)
template <int NQ>
__device__ __forceinline__ void match_knn2_body(const uint4 *__restrict__ q4, const uint4 *__restrict__ t4, int nq, int nt,
                                                int qbase, uint4 *tile, int32_t *__restrict__ sel_b, int4 *__restrict__ knn_b) {
    uint4 qa[NQ], qb[NQ];
    uint32_t k1[NQ], k2[NQ];
#pragma unroll
    for (int s = 0; s < NQ; s++) {
        const int q = qbase + s * kThreads + threadIdx.x;
        const int qc = q < nq ? q : nq - 1;   // clamp: lanes past the end recompute a valid row
        qa[s] = q4[2 * qc];
        qb[s] = q4[2 * qc + 1];
        k1[s] = 0xFFFFFFFFu;
        k2[s] = 0xFFFFFFFFu;
    }

    for (int t0 = 0; t0 < nt; t0 += kTile) {
        __syncthreads();
        const int rows = min(kTile, nt - t0);
        for (int i = threadIdx.x; i < rows * 2; i += kThreads) tile[i] = t4[2 * t0 + i];
        __syncthreads();
        int j = 0;
        for (; j + 4 <= rows; j += 4) {   // four train rows per trip: 4 * NQ independent distance chains in flight
            uint4 ta[4], tb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ta[u] = tile[2 * (j + u)];
                tb[u] = tile[2 * (j + u) + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int s = 0; s < NQ; s++) {
                    const uint32_t key = (ham256(qa[s], qb[s], ta[u], tb[u]) << 16) | (uint32_t)(t0 + j + u);
                    k2[s] = med3_u32(k1[s], k2[s], key);
                    k1[s] = min(k1[s], key);
                }
        }
        for (; j < rows; j++) {
            const uint4 ta = tile[2 * j], tb = tile[2 * j + 1];
#pragma unroll
            for (int s = 0; s < NQ; s++) {
                const uint32_t key = (ham256(qa[s], qb[s], ta, tb) << 16) | (uint32_t)(t0 + j);
                k2[s] = med3_u32(k1[s], k2[s], key);
                k1[s] = min(k1[s], key);
            }
        }
    }

#pragma unroll
    for (int s = 0; s < NQ; s++) {
        const int q = qbase + s * kThreads + threadIdx.x;
        if (q >= nq) continue;
        const int d0 = (int)(k1[s] >> 16), i0 = (int)(k1[s] & 0xFFFFu);
        const int d1 = (int)(k2[s] >> 16), i1 = (int)(k2[s] & 0xFFFFu);
        const bool pass = (nt >= 2) && (10 * d0 < 7 * d1);
        sel_b[q] = pass ? i0 : -1;
        if (knn_b) {
            int4 o;
            o.x = nt >= 1 ? i0 : -1;
            o.y = nt >= 1 ? d0 : 0x7FFFFFFF;
            o.z = nt >= 2 ? i1 : -1;
            o.w = nt >= 2 ? d1 : 0x7FFFFFFF;
            knn_b[q] = o;
        }
    }
}

__global__ __launch_bounds__(kThreads) void match_knn2_kernel(
    const uint8_t *__restrict__ desc1, const int32_t *__restrict__ n1,
    const uint8_t *__restrict__ desc2, const int32_t *__restrict__ n2, int kp_stride,
    int32_t *__restrict__ sel, int32_t *__restrict__ knn) {
    const int b = blockIdx.y;
    const int nq = n1[b], nt = n2[b];
    const int qbase = blockIdx.x * (kThreads * kQueriesPerLane);
    if (qbase >= nq) return;   // uniform for the whole workgroup

    __shared__ uint4 tile[kTile * 2];

    const uint4 *q4 = reinterpret_cast<const uint4 *>(desc1 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    const uint4 *t4 = reinterpret_cast<const uint4 *>(desc2 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    int32_t *sel_b = sel + (size_t)b * kp_stride;
    int4 *knn_b = knn ? reinterpret_cast<int4 *>(knn) + (size_t)b * kp_stride : nullptr;
    static_assert(kQueriesPerLane == 2, "the two instantiations below cover 1 and 2 rows per lane");
    if (qbase + kThreads >= nq)
        match_knn2_body<1>(q4, t4, nq, nt, qbase, tile, sel_b, knn_b);
    else
        match_knn2_body<2>(q4, t4, nq, nt, qbase, tile, sel_b, knn_b);
}

// Ordered compaction of the ratio-test survivors into (queryIdx, trainIdx) pairs, query order
// (the reference's i_matches.push_back loop, src/Frame.cpp:89-94).  One workgroup per pair.
__global__ __launch_bounds__(kThreads) void match_compact_kernel(const int32_t *__restrict__ sel,
                                                                 const int32_t *__restrict__ n1,
                                                                 int kp_stride,
                                                                 int32_t *__restrict__ pairs,
                                                                 int32_t *__restrict__ m_out) {
    const int b = blockIdx.x;
    const int nq = n1[b];
    __shared__ int wave_cnt[kThreads / 64];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q0 = 0; q0 < nq; q0 += kThreads) {
        const int q = q0 + threadIdx.x;
        const int t = q < nq ? sel[(size_t)b * kp_stride + q] : -1;
        const bool keep = t >= 0;
        const unsigned long long bal = __ballot(keep);
        const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; w++) off += wave_cnt[w];
        if (keep) {
            int2 o;
            o.x = q;
            o.y = t;
            reinterpret_cast<int2 *>(pairs)[(size_t)b * kp_stride + off + in_wave] = o;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < kThreads / 64; w++) tot += wave_cnt[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) m_out[b] = base_s;
}

}  // namespace

int vs_launch_match(vslam_ctx *ctx, const uint8_t *d1, const int32_t *n1, const uint8_t *d2,
                    const int32_t *n2, int batch, int kp_stride, int32_t *pairs, int32_t *m,
                    int32_t *knn) {
    VS_REQUIRE(ctx, d1 && n1 && d2 && n2 && pairs && m, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    int32_t *sel = nullptr;
    int rc = vs_arena_get(ctx, "match.sel", sizeof(int32_t) * (size_t)batch * kp_stride, (void **)&sel);
    if (rc) return rc;
    {
        VsProfScope ps(ctx, "match_knn2_kernel");
        dim3 grid(vs_div_up(kp_stride, kThreads * kQueriesPerLane), batch);
        match_knn2_kernel<<<grid, kThreads, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn);
    }
    {
        VsProfScope ps(ctx, "match_compact_kernel");
        match_compact_kernel<<<batch, kThreads, 0, ctx->stream>>>(sel, n1, kp_stride, pairs, m);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

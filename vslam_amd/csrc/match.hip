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

#include <cstdlib>

namespace {

constexpr int kThreads = 256;
#ifdef VSLAM_EXPERIMENTS   // the vector-ALU matcher of round 1 (VSLAM_MATCH_POPCOUNT): kept for A/B timing only
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

#endif   // VSLAM_EXPERIMENTS

// ------------------------------------------------------------------------------------------------------------------
// The same knn-2 on the matrix cores (the default).  For 0/1 vectors a, b of 256 bits, Hamming(a, b) = |a| + |b| - 2 a.b,
// and a.b over 256 positions is an int8 dot product: v_mfma_i32_32x32x32_i8 forms 32 x 32 of them per instruction at
// 16 times the rate the vector ALUs can xor and popcount.  Everything stays integer, so the distances — and with the
// same packed-key min-2 the indices and the tie order — are exactly those of the popcount kernel above.
//   workgroup = 256 query rows of one pair (64 per wave: two 32-row operand tiles, expanded once to one byte per bit and
//   kept in registers); the train rows come by in tiles of 32: the workgroup expands a tile's packed descriptors to
//   bytes in LDS (double buffered) together with their popcounts, every wave reads it as the B operand;
//   per tile and wave 16 MFMAs (2 row tiles x 8 k-steps of 32 bits), then 4 VALU ops per result for the running two
//   smallest keys ((|b| - 2 a.b + 512) << 16 | train index; |a| is the same for a row and is added at the end);
//   at the end the 32 lanes that hold different columns of a row merge their two-smallest pairs.
// Operand layout (checked on the device with tools/mfma_probe.hip): A lane l = row l & 31, B lane l = column l & 31,
// both carrying the 16 k-positions 16 (l >> 5) .. + 15 of the step in their 16 bytes (any order works as long as A and
// B agree: they are built by the same bit -> byte spread); D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int kMQ = 256;         // query rows per workgroup
constexpr int kMT = 32;          // train rows per tile
#ifdef VSLAM_EXPERIMENTS   // (the int8 form's)
constexpr int kMStride = 272;    // bytes per expanded train row in LDS: 256 + 16, so 16 consecutive rows cover all banks
constexpr uint32_t kMBias = 512; // keeps |b| - 2 a.b positive
constexpr uint32_t kMPad = 0x4000;   // "|b|" of a train row beyond nt: never among the two smallest of a real row
#endif

__device__ __forceinline__ uint32_t spread4(uint32_t nib) {   // 4 bits -> 4 bytes holding 0 / 1
    return (nib * 0x00204081u) & 0x01010101u;
}
[[maybe_unused]] __device__ __forceinline__ v4i spread16(uint32_t bits) {   // (int8 form: experiments build)      // 16 bits -> 16 bytes
    v4i r;
    r.x = (int)spread4(bits & 0xFu);
    r.y = (int)spread4((bits >> 4) & 0xFu);
    r.z = (int)spread4((bits >> 8) & 0xFu);
    r.w = (int)spread4((bits >> 12) & 0xFu);
    return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

// base - (dot << 17) in one instruction (dot <= 256 and -2^17 both fit the 24-bit operands: the compiler folds the
// 24-bit multiply and the add into v_mad_i32_i24).  Not inline assembly: the operand comes straight out of an MFMA, and
// the wait states a vector instruction needs after one are only inserted for instructions the compiler knows.
[[maybe_unused]] __device__ __forceinline__ uint32_t mad24(int dot, int base, int neg_two_17) {
    return (uint32_t)(__mul24(dot, neg_two_17) + base);
}
// (smallest, second smallest) of this lane's pair and the pair of the lane a DPP pattern points at
template <int CTRL>
__device__ __forceinline__ void merge2(uint32_t &x1, uint32_t &x2) {
    const uint32_t o1 = dpp_u32<CTRL>(x1), o2 = dpp_u32<CTRL>(x2);
    const uint32_t lo = min(x1, o1), hi = max(x1, o1);
    x2 = min(hi, min(x2, o2));
    x1 = lo;
}

#ifdef VSLAM_EXPERIMENTS   // the int8 form (VSLAM_OPT_MATCH_FORM 2): kept for A/B timing only
// RT = 32-row query tiles per wave; a workgroup is kMQ / (32 RT) waves.
template <int RT>
__global__ __launch_bounds__(64 * kMQ / (32 * RT)) void match_knn2_mfma_kernel(
    const uint8_t *__restrict__ desc1, const int32_t *__restrict__ n1, const uint8_t *__restrict__ desc2,
    const int32_t *__restrict__ n2, int kp_stride, int32_t *__restrict__ sel, int32_t *__restrict__ knn) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = n1[b], nt = n2[b];
    const int qbase = blockIdx.x * kMQ;
    if (qbase >= nq) return;   // uniform for the whole workgroup

    __shared__ __align__(16) uint8_t s_b[2][kMT * kMStride];
    __shared__ uint32_t s_pb[2][kMT];
    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(desc1 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(desc2 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    int32_t *sel_b = sel + (size_t)b * kp_stride;
    int4 *knn_b = knn ? reinterpret_cast<int4 *>(knn) + (size_t)b * kp_stride : nullptr;

    const int r = lane & 31, half = lane >> 5;
    // A operands: this lane's row of each of the wave's two 32-row tiles, bits 16 half .. + 15 of every dword
    v4i a[RT][8];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int q = min(qbase + wave * (32 * RT) + rt * 32 + r, nq - 1);   // rows past the end repeat the last one (never written)
        const uint4 lo = reinterpret_cast<const uint4 *>(q32)[2 * q], hi = reinterpret_cast<const uint4 *>(q32)[2 * q + 1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 8; ks++) a[rt][ks] = spread16((w[ks] >> (16 * half)) & 0xFFFFu);
    }
    uint32_t k1[RT][16], k2[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            k1[rt][g] = 0xFFFFFFFFu;
            k2[rt][g] = 0xFFFFFFFFu;
        }

    // Staging of train tiles: thread t owns dword t & 7 of row t >> 3 of every tile (one coalesced KiB per tile).  The
    // dword of tile i + 2 is requested while tile i is being multiplied and tile i + 1 (already in a register) is being
    // expanded into the other LDS buffer, so no wave ever waits for global memory inside the loop.
    // With 8 waves (RT = 1) the two halves of the workgroup share a dword: threads 0..255 expand its low 16 bits (and
    // sum the popcounts), threads 256..511 its high 16 bits.
    constexpr bool kSplit = RT == 1;
    const int st = kSplit ? (tid & 255) : tid, shalf = kSplit ? (tid >> 8) : 0;
    const int srow = st >> 3, sd = st & 7;
    auto fetch = [&](int tile) -> uint32_t {
        const int t = tile * kMT + srow;
        return t < nt ? t32[(size_t)t * 8 + sd] : 0u;
    };
    auto expand = [&](uint32_t wv, int tile, int buf) {
        uint8_t *dst = &s_b[buf][srow * kMStride + sd * 32];
        if (kSplit) {
            *reinterpret_cast<v4i *>(dst + 16 * shalf) = spread16((wv >> (16 * shalf)) & 0xFFFFu);
        } else {
            *reinterpret_cast<v4i *>(dst) = spread16(wv & 0xFFFFu);
            *reinterpret_cast<v4i *>(dst + 16) = spread16(wv >> 16);
        }
        if (!kSplit || shalf == 0) {                // whole waves: threads 0..255
            uint32_t pc = (uint32_t)__popc(wv);     // |b|: sum over the row's 8 dwords = 8 neighbouring lanes
            pc += dpp_u32<0xB1>(pc);                // quad_perm [1,0,3,2]
            pc += dpp_u32<0x4E>(pc);                // quad_perm [2,3,0,1]
            pc += dpp_u32<0x141>(pc);               // row_half_mirror: the other quad of the 8
            if (sd == 0) s_pb[buf][srow] = tile * kMT + srow < nt ? pc : kMPad;
        }
    };

    int neg_two_17 = -(1 << 17);
    asm volatile("" : "+s"(neg_two_17));   // opaque, or the multiply becomes a shift and the add a second instruction
    const int ntiles = (nt + kMT - 1) / kMT;
    if (ntiles > 0) expand(fetch(0), 0, 0);
    uint32_t w_next = ntiles > 1 ? fetch(1) : 0u;
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) expand(w_next, tile + 1, buf ^ 1);   // the other buffer was last read two barriers ago
        if (tile + 2 < ntiles) w_next = fetch(tile + 2);
        v16i acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const uint8_t *src = &s_b[buf][r * kMStride + 16 * half];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const v4i bv = *reinterpret_cast<const v4i *>(src + 32 * ks);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt][ks], bv, acc[rt], 0, 0, 0);
            // The first step's accumulator is the constant 0, so its 16 result registers are new, and with bv dead behind the
            // multiply this compiler lets them start on bv's four (8-wave shape: v_mfma v[0:15], v[16:19], v[0:3], 0).  No
            // wrong result was ever seen with it, but an instruction that overwrites an operand it may still be reading is
            // not something to rely on: bv stays alive across the multiply.
            if (ks == 0) asm volatile("" : : "v"(bv));
        }
        // key = (|b| + bias - 2 a.b) << 16 | train index = base - (a.b << 17): one multiply-add per result
        const int base = (int)(((s_pb[buf][r] + kMBias) << 16) | (uint32_t)(tile * kMT + r));
#pragma unroll
        for (int g = 0; g < 16; g++) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const uint32_t key = mad24(acc[rt][g], base, neg_two_17);
                k2[rt][g] = med3_u32(k1[rt][g], k2[rt][g], key);
                k1[rt][g] = min(k1[rt][g], key);
            }
        }
        __syncthreads();
    }

    // the 32 lanes of a half hold different columns of the same rows: merge their (smallest, second smallest)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            uint32_t x1 = k1[rt][g], x2 = k2[rt][g];
            merge2<0xB1>(x1, x2);    // quad_perm [1,0,3,2]
            merge2<0x4E>(x1, x2);    // quad_perm [2,3,0,1]
            merge2<0x141>(x1, x2);   // row_half_mirror: the other quad of the 8
            merge2<0x140>(x1, x2);   // row_mirror: the other 8 of the 16
            {                        // the other 16 of the 32: across DPP rows, through the LDS crossbar
                const uint32_t o1 = __shfl_xor(x1, 16, 64), o2 = __shfl_xor(x2, 16, 64);
                const uint32_t lo = min(x1, o1), hi = max(x1, o1);
                x2 = min(hi, min(x2, o2));
                x1 = lo;
            }
            k1[rt][g] = x1;
            k2[rt][g] = x2;
        }
    // lane (g, half) of each row tile writes row (g & 3) + 8 (g >> 2) + 4 half
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (r != g) continue;
            const int q = qbase + wave * (32 * RT) + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * half;
            if (q >= nq) continue;
            const uint4 lo = reinterpret_cast<const uint4 *>(q32)[2 * q], hi = reinterpret_cast<const uint4 *>(q32)[2 * q + 1];
            const int pa = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) +
                           __popc(hi.w);
            const int d0 = (int)(k1[rt][g] >> 16) - (int)kMBias + pa, i0 = (int)(k1[rt][g] & 0xFFFFu);
            const int d1 = (int)(k2[rt][g] >> 16) - (int)kMBias + pa, i1 = (int)(k2[rt][g] & 0xFFFFu);
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

#endif   // VSLAM_EXPERIMENTS

// ------------------------------------------------------------------------------------------------------------------
// The same knn-2 with FP4 operands (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1, unit block scales): a descriptor bit becomes
// the nibble +1.0 (bit 0) or -1.0 (bit 1), so a.b over the 256 positions is 256 - 2 Hamming(a, b) -- no |a|, |b| terms --
// in four K = 64 steps at twice the int8 form's rate, with half the operand registers and half the LDS traffic.  Every
// product is +-1 and every partial sum an integer below 2^9: the f32 accumulation is exact, so distances, indices and
// the tie order are those of the other two kernels.  The running two best per (lane, row) are float keys
// dot * 16384 - train index (exact: below 2^23), largest first = smallest distance, then lowest index: one v_fma_f32,
// one v_med3_f32 and one v_max_f32 per result.  Layout (tools/mfma_fp4_probe.hip checks it on the device): A lane l = row
// l & 31, B lane l = column l & 31, each carrying 32 positions of the step chosen by l >> 5; D as for the int8 form.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kFStride = 144;          // bytes per expanded train row in LDS: 128 + 16 (as kMStride: rows spread over the banks)
constexpr float kFScale = 16384.f;     // VSLAM_MAX_KP: the index occupies the low 14 bits of a key
constexpr float kFPad = 33554432.f;    // "index" of a train row beyond nt: below every key of a real row
constexpr float kFNone = -67108864.f;  // no candidate yet
constexpr int kFBias = 0x05000000;     // biased integer form of a key for the cross-lane merge: u = kFBias - (int)key > 0

__device__ __forceinline__ uint32_t spread8_pm1(uint32_t b) {   // 8 bits -> 8 nibbles: 0x2 (+1.0) | bit << 3 (sign)
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 3) | 0x22222222u;
}
__device__ __forceinline__ v4i spread32_pm1(uint32_t w) {       // 32 bits -> 32 nibbles (16 bytes)
    v4i r;
    r.x = (int)spread8_pm1(w & 0xFFu);
    r.y = (int)spread8_pm1((w >> 8) & 0xFFu);
    r.z = (int)spread8_pm1((w >> 16) & 0xFFu);
    r.w = (int)spread8_pm1(w >> 24);
    return r;
}
__device__ __forceinline__ float med3_f32(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// The train descriptors of a pair spread to +-1 nibbles once (128 bytes per row) instead of by every workgroup that
// streams them: seven workgroups of a pair at 2000 keypoints would each redo the same spreading, a quarter of the
// matcher's vector instructions.  Row stride 128 bytes, kp_pad rows per pair (a multiple of the tile height).
__global__ __launch_bounds__(256) void match_spread_kernel(const uint8_t *__restrict__ desc2, const int32_t *__restrict__ n2,
                                                           int kp_stride, int kp_pad, uint8_t *__restrict__ tx) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;   // dword i & 7 of row i >> 3
    const int row = i >> 3;
    if (row >= n2[b]) return;
    const uint32_t w = reinterpret_cast<const uint32_t *>(desc2 + (size_t)b * kp_stride * VSLAM_DESC_BYTES)[i];
    *reinterpret_cast<v4i *>(tx + ((size_t)b * kp_pad + row) * 128 + (size_t)(i & 7) * 16) = spread32_pm1(w);
}

// PRE: the train rows come spread already (tx, match_spread_kernel); otherwise the workgroup spreads them itself.
// CT = train tiles per trip of the main loop (one barrier per trip; 2 only with PRE).
template <int RT, bool PRE, int CT>
__global__ __launch_bounds__(64 * kMQ / (32 * RT)) void match_knn2_fp4_kernel(
    const uint8_t *__restrict__ desc1, const int32_t *__restrict__ n1, const uint8_t *__restrict__ desc2,
    const int32_t *__restrict__ n2, int kp_stride, int32_t *__restrict__ sel, int32_t *__restrict__ knn,
    const uint8_t *__restrict__ tx, int kp_pad, int batch, int per_pair) {
    // 1-D grid, remapped so that the workgroups of a pair share an XCD: they all stream the pair's train rows, and each XCD
    // has its own L2 -- dealt round-robin (blockIdx.y = pair), the seven workgroups of a pair sat on seven XCDs and each
    // fetched the rows from the fabric for itself (round 4: 425 MB of reads per launch for 57 MB of rows).
    int b, qt;
    vs_xcd_item_block(blockIdx.x, per_pair, b, qt);
    if (b >= batch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = n1[b], nt = n2[b];
    const int qbase = qt * kMQ;
    if (qbase >= nq) return;   // uniform for the whole workgroup

    static_assert(CT == 1 || PRE, "two tiles per trip only with pre-spread train rows");
    __shared__ __align__(16) uint8_t s_b[2][CT * kMT * kFStride];
    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(desc1 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(desc2 + (size_t)b * kp_stride * VSLAM_DESC_BYTES);
    int32_t *sel_b = sel + (size_t)b * kp_stride;
    int4 *knn_b = knn ? reinterpret_cast<int4 *>(knn) + (size_t)b * kp_stride : nullptr;

    const int r = lane & 31, half = lane >> 5;
    // A operands: this lane's row of each of the wave's 32-row tiles; step s takes dword 2 s + half of the descriptor
    v8i a[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int q = min(qbase + wave * (32 * RT) + rt * 32 + r, nq - 1);   // rows past the end repeat the last one (never written)
        const uint2 *qrow = reinterpret_cast<const uint2 *>(q32 + (size_t)q * 8);   // (a run-time index into a register
#pragma unroll                                                                      //  array would put the array in LDS)
        for (int s = 0; s < 4; s++) {
            const uint2 w2 = qrow[s];
            const v4i x = spread32_pm1(half ? w2.y : w2.x);
            a[rt][s] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
    }
    float k1[RT][16], k2[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            k1[rt][g] = kFNone;
            k2[rt][g] = kFNone;
        }

    // Staging as in the int8 form: thread t owns dword t & 7 of row t >> 3 of every tile; with 8 waves (RT = 1) the two
    // halves of the workgroup share a dword and expand 16 bits of it each.
    constexpr bool kSplit = RT == 1;
    const int st = kSplit ? (tid & 255) : tid, shalf = kSplit ? (tid >> 8) : 0;
    const int srow = st >> 3, sd = st & 7;
    // what a thread carries from global memory to LDS per tile: its packed dword, or (PRE) its 16 (8) spread bytes; rows beyond
    // nt are never initialised in tx -- any nibble pattern is a finite number, and those columns' keys carry kFPad
    struct Stage {
        uint4 v[CT];
    };
    const uint8_t *txb = PRE ? tx + (size_t)b * kp_pad * 128 + (size_t)srow * 128 + sd * 16 + (kSplit ? 8 * shalf : 0) : nullptr;
    auto fetch = [&](int trip) -> Stage {
        Stage st_;
#pragma unroll
        for (int c = 0; c < CT; c++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (PRE) {
                const uint8_t *src = txb + (size_t)(trip * CT + c) * (kMT * 128);
                if (kSplit) {
                    const uint2 h = *reinterpret_cast<const uint2 *>(src);
                    v.x = h.x;
                    v.y = h.y;
                } else {
                    v = *reinterpret_cast<const uint4 *>(src);
                }
            } else {
                const int t = trip * kMT + srow;
                v.x = t < nt ? t32[(size_t)t * 8 + sd] : 0u;
            }
            st_.v[c] = v;
        }
        return st_;
    };
    auto expand = [&](const Stage &st_, int buf) {
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const uint4 v = st_.v[c];
            uint8_t *dst = &s_b[buf][(c * kMT + srow) * kFStride + sd * 16];
            if (PRE) {
                if (kSplit) *reinterpret_cast<uint2 *>(dst + 8 * shalf) = make_uint2(v.x, v.y);
                else *reinterpret_cast<uint4 *>(dst) = v;
            } else if (kSplit) {
                const uint32_t hw = (v.x >> (16 * shalf)) & 0xFFFFu;
                *reinterpret_cast<uint2 *>(dst + 8 * shalf) = make_uint2(spread8_pm1(hw & 0xFFu), spread8_pm1(hw >> 8));
            } else {
                *reinterpret_cast<v4i *>(dst) = spread32_pm1(v.x);
            }
        }
    };

    // (A software pipeline over the tiles -- three LDS buffers, the multiplies of tile i + 1 issued in front of the bookkeeping
    // of tile i, two accumulator sets -- was measured and is slower: 0.222 against 0.198 ms at C3, 1.72 against 1.53 at C5; it
    // costs a wave per SIMD, and the waves of the other workgroups already fill the matrix pipe while this one keeps books.)
    const int ntrips = (nt + CT * kMT - 1) / (CT * kMT);
    if (ntrips > 0) expand(fetch(0), 0);
    Stage w_next;
    if (ntrips > 1) w_next = fetch(1);
    __syncthreads();
    for (int trip = 0; trip < ntrips; trip++) {
        const int buf = trip & 1;
        if (trip + 1 < ntrips) expand(w_next, buf ^ 1);   // the other buffer was last read two barriers ago
        if (trip + 2 < ntrips) w_next = fetch(trip + 2);
        v16f acc[CT][RT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[c][rt] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint8_t *src = &s_b[buf][(c * kMT + r) * kFStride + 16 * half];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const v4i bq = *reinterpret_cast<const v4i *>(src + 32 * s);
                const v8i bv = v8i{bq.x, bq.y, bq.z, bq.w, 0, 0, 0, 0};
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    acc[c][rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[rt][s], bv, acc[c][rt], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
        }
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const int t = (trip * CT + c) * kMT + r;
            const float nidx = t < nt ? -(float)t : -kFPad;   // key = dot * 16384 - index
#pragma unroll
            for (int g = 0; g < 16; g++) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const float key = __builtin_fmaf(acc[c][rt][g], kFScale, nidx);
                    k2[rt][g] = med3_f32(k1[rt][g], k2[rt][g], key);
                    k1[rt][g] = fmaxf(k1[rt][g], key);
                }
            }
        }
        __syncthreads();
    }

    // merge over the 32 lanes of a half (different columns of the same rows) on the biased integer form (smaller = better)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            uint32_t x1 = (uint32_t)(kFBias - (int)k1[rt][g]), x2 = (uint32_t)(kFBias - (int)k2[rt][g]);
            merge2<0xB1>(x1, x2);    // quad_perm [1,0,3,2]
            merge2<0x4E>(x1, x2);    // quad_perm [2,3,0,1]
            merge2<0x141>(x1, x2);   // row_half_mirror: the other quad of the 8
            merge2<0x140>(x1, x2);   // row_mirror: the other 8 of the 16
            {                        // the other 16 of the 32: across DPP rows, through the LDS crossbar
                const uint32_t o1 = __shfl_xor(x1, 16, 64), o2 = __shfl_xor(x2, 16, 64);
                const uint32_t lo = min(x1, o1), hi = max(x1, o1);
                x2 = min(hi, min(x2, o2));
                x1 = lo;
            }
            if (r != g) continue;
            // lane (g, half) of each row tile writes row (g & 3) + 8 (g >> 2) + 4 half
            const int q = qbase + wave * (32 * RT) + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * half;
            if (q >= nq) continue;
            const int e1 = kFBias - (int)x1, e2 = kFBias - (int)x2;             // dot * 16384 - index
            const int dot1 = (e1 + 16383) >> 14, dot2 = (e2 + 16383) >> 14;     // ceil (arithmetic shift: floor)
            const int i0 = dot1 * 16384 - e1, i1 = dot2 * 16384 - e2;
            const int d0 = (256 - dot1) >> 1, d1 = (256 - dot2) >> 1;
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
#ifndef VSLAM_EXPERIMENTS
    {   // the product's one matcher: FP4 products, 8 waves x 32 rows, train rows spread once per pair, two tiles per trip
        // (tools/ab_match.sh on the experiments build: 0.186 / 1.30 ms at C3 / C5 against 0.191-0.208 / 1.34-1.51 for the other
        // FP4 arrangements and 0.279 / 1.97 for the int8 form)
        VsProfScope ps(ctx, "match_knn2_kernel");
        const int per_pair = vs_div_up(kp_stride, kMQ);
        const int xgrid = vs_xcd_grid(batch, per_pair);
        const int kp_pad = vs_div_up(kp_stride, 2 * kMT) * 2 * kMT;
        uint8_t *tx = nullptr;
        if ((rc = vs_arena_get(ctx, "match.spread", (size_t)batch * kp_pad * 128, (void **)&tx))) return rc;
        match_spread_kernel<<<dim3(vs_div_up(kp_stride * 8, 256), batch), 256, 0, ctx->stream>>>(d2, n2, kp_stride, kp_pad, tx);
        match_knn2_fp4_kernel<1, true, 2><<<xgrid, 512, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, tx, kp_pad, batch, per_pair);
    }
#else
    static const bool popcount_path = VS_EXPERIMENT_ENV("VSLAM_MATCH_POPCOUNT") != nullptr;   // the vector-ALU kernel, for A/B timing
    {
        VsProfScope ps(ctx, "match_knn2_kernel");
        if (popcount_path) {
            dim3 grid(vs_div_up(kp_stride, kThreads * kQueriesPerLane), batch);
            match_knn2_kernel<<<grid, kThreads, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn);
        } else {
            // Two shapes of the same kernel.  8 waves x 32 rows (106 VGPRs, 4 waves / SIMD, results straight out of
            // the MFMA's vector registers: 3 vector ops per result instead of 4) is the faster kernel on its own
            // (0.254 vs 0.268 ms at C3, 1.89 vs 2.02 at C5); 4 waves x 64 rows (196 VGPRs, 2 waves / SIMD) leaves the
            // k-d build, which the front end runs beside the matcher, the wave slots it needs, and the STEP is faster
            // with it while the two take about equally long (C3: 3.70 vs 3.76 ms).  With more keypoints the matcher
            // dominates (quadratic against n log n) and the first shape wins the step too (C5: 18.1 vs 18.3 ms).
            static const char *const shape = VS_EXPERIMENT_ENV("VSLAM_MATCH_SHAPE");   // "8x32" / "4x64" force one (A/B timing)
#ifdef VSLAM_EXPERIMENTS
            const int pick = ctx->match_shape ? ctx->match_shape : (shape ? (shape[0] == '4' ? 2 : 1) : 0);
#else
            constexpr int pick = 0;   // (VSLAM_OPT_MATCH_SHAPE / _FORM are settable in the experiments build only)
            (void)shape;
#endif
            const bool wide = pick ? pick == 2 : kp_stride <= 2048;
            dim3 grid(vs_div_up(kp_stride, kMQ), batch);
            static const char *const form = VS_EXPERIMENT_ENV("VSLAM_MATCH_FORM");   // "i8": the int8 form (A/B timing); default FP4
#ifdef VSLAM_EXPERIMENTS
            const bool fp4 = ctx->match_form ? ctx->match_form == 1 : !(form && form[0] == 'i');
#else
            constexpr bool fp4 = true;
            (void)form;
#endif
            if (fp4) {
                // The FP4 form as measured (tools/ab_match.sh, kernel alone, C3 / C5 in ms; int8 form: 0.279 / 1.97):
                //   8 x 32, pre-spread, 2 tiles per trip   0.186 / 1.30   <- the default
                //   8 x 32, pre-spread, 1 tile per trip    0.191 / 1.34
                //   4 x 64, pre-spread, 1 (2) per trip     0.200 (0.205) / 1.35 (1.38)
                //   8 x 32 / 4 x 64 spreading in place     0.198 / 1.51 and 0.208 / 1.44
                // (8 x 32: 5 waves per SIMD; the pre-spread rows pay more the more workgroups share them)
                const bool wide4 = pick == 2;
                const int per_pair = vs_div_up(kp_stride, kMQ);
                const int xgrid = vs_xcd_grid(batch, per_pair);
                static const char *const nopre = VS_EXPERIMENT_ENV("VSLAM_MATCH_NO_PRESPREAD");   // A/B timing: every workgroup spreads for itself
                static const char *const ct_env = VS_EXPERIMENT_ENV("VSLAM_MATCH_TILES_PER_TRIP");
                const int ct = ct_env ? atoi(ct_env) : 2;
                if (!nopre) {
                    const int kp_pad = vs_div_up(kp_stride, 2 * kMT) * 2 * kMT;
                    uint8_t *tx = nullptr;
                    if ((rc = vs_arena_get(ctx, "match.spread", (size_t)batch * kp_pad * 128, (void **)&tx))) return rc;
                    match_spread_kernel<<<dim3(vs_div_up(kp_stride * 8, 256), batch), 256, 0, ctx->stream>>>(d2, n2, kp_stride, kp_pad, tx);
                    if (wide4 && ct == 2) match_knn2_fp4_kernel<2, true, 2><<<xgrid, 256, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, tx, kp_pad, batch, per_pair);
                    else if (wide4) match_knn2_fp4_kernel<2, true, 1><<<xgrid, 256, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, tx, kp_pad, batch, per_pair);
                    else if (ct == 2) match_knn2_fp4_kernel<1, true, 2><<<xgrid, 512, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, tx, kp_pad, batch, per_pair);
                    else match_knn2_fp4_kernel<1, true, 1><<<xgrid, 512, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, tx, kp_pad, batch, per_pair);
                } else if (wide4) {
                    match_knn2_fp4_kernel<2, false, 1><<<xgrid, 256, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, nullptr, 0, batch, per_pair);
                } else {
                    match_knn2_fp4_kernel<1, false, 1><<<xgrid, 512, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn, nullptr, 0, batch, per_pair);
                }
            } else if (wide) match_knn2_mfma_kernel<2><<<grid, 256, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn);
            else match_knn2_mfma_kernel<1><<<grid, 512, 0, ctx->stream>>>(d1, n1, d2, n2, kp_stride, sel, knn);
        }
    }
#endif   // VSLAM_EXPERIMENTS
    {
        VsProfScope ps(ctx, "match_compact_kernel");
        match_compact_kernel<<<batch, kThreads, 0, ctx->stream>>>(sel, n1, kp_stride, pairs, m);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

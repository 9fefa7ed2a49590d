// cv::GaussianBlur 7x7, sigma 2, 8-bit fixed point, as ORB::compute applies it before sampling (reference:
// src/Frame.cpp:68) for gfx950: gaussian7_stream_kernel (rows of a multiple of 4 bytes: width % 4 == 0, or padded rows with a
// mirrored tail, vslam_ctx::img_pitch) and gaussian7_kernel (any width).
#include "image_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2, 8U fixed point: Q8 taps (18,34,48,56,48,34,18), Q16 accumulate
// ------------------------------------------------------------------------------------------
constexpr int kBT = 256, kBTW = 64, kBTH = 16;

__global__ __launch_bounds__(kBT) void gaussian7_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                        uint8_t *__restrict__ out) {
    __shared__ uint8_t G[kBTH + 6][kBTW + 8];
    __shared__ uint16_t RP[kBTH + 6][kBTW];
    const int f = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * kBTW, y0 = blockIdx.y * kBTH;
    const uint8_t *src = gray + (size_t)f * w * h;
    for (int i = tid; i < (kBTH + 6) * (kBTW + 6); i += kBT) {
        const int r = i / (kBTW + 6), c = i - r * (kBTW + 6);
        G[r][c] = src[(size_t)reflect101(y0 - 3 + r, h) * w + reflect101(x0 - 3 + c, w)];
    }
    __syncthreads();
    for (int i = tid; i < (kBTH + 6) * kBTW; i += kBT) {
        const int r = i / kBTW, c = i - r * kBTW;
        const uint32_t s = 18u * G[r][c] + 34u * G[r][c + 1] + 48u * G[r][c + 2] + 56u * G[r][c + 3] +
                           48u * G[r][c + 4] + 34u * G[r][c + 5] + 18u * G[r][c + 6];
        RP[r][c] = (uint16_t)s;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ly = ty * 4 + k, y = y0 + ly, x = x0 + tx;
        if (x < w && y < h) {
            const uint32_t s = 18u * RP[ly][tx] + 34u * RP[ly + 1][tx] + 48u * RP[ly + 2][tx] + 56u * RP[ly + 3][tx] +
                               48u * RP[ly + 4][tx] + 34u * RP[ly + 5][tx] + 18u * RP[ly + 6][tx];
            out[((size_t)f * h + y) * w + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
    }
}

// Vectorised form for widths that are a multiple of 4 (every config in BASELINE.json): 256x32
// tile, one lane = 4 adjacent pixels x 8 rows.  The gray tile sits in LDS as dwords; the row pass
// is two v_dot4_u32_u8 per pixel on byte windows cut with v_alignbyte, its results stay in a
// rolling 7-row register window, and the column pass runs straight from those registers, so the
// kernel reads each gray byte once from HBM and writes each output byte once as a dword store.
// Streaming form (width % 4 == 0, height >= 4): one wave per 256-pixel column strip and row segment, the last
// seven horizontally filtered rows rolling in registers, gray rows loaded straight from global memory three
// rows ahead (no LDS, no barriers, no tile seams: the tile form filters 14 rows to produce 8).
// Step t of a segment owning rows [ys, ye): filter gray row ys - 3 + t horizontally; from t = 6 on, output
// row ys - 6 + t.  Same Q8 taps, Q16 accumulate and rounding as gaussian7_kernel.
struct BlurState {
    uint32_t pk[7][4];    // horizontally filtered rows (Q8, < 2^16) in pairs: slot k holds row k - 1 | row k << 16
    uint32_t last[4];     // the newest filtered row by itself
    uint32_t raw[7][3];   // prefetched gray dwords x-4, x, x+4 of the next seven rows
};
struct BlurArgs {
    const uint8_t *src;
    uint8_t *dst;
    int w, h, ys, steps, x;
    uint32_t voff_l, voff_c, voff_r;
    bool edge, left_fix, right_fix, own_lane;
};

// v_mad_u32_u24 through the builtin, not inline asm: the result feeds v_dot2_u32_u16 as its addend, and the wait states the
// hardware wants between the two are inserted by the compiler only if it can see what wrote the register (seen: with the
// asm form the third and fourth pixel of a lane, whose multiply-adds come last, read a stale addend).
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
typedef unsigned short blur_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_u16(uint32_t rows, uint32_t taps, uint32_t acc) {   // v_dot2_u32_u16
    return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_u16x2, rows), __builtin_bit_cast(blur_u16x2, taps), acc, false);
}

template <int K>   // K = t % 7: ring slot of both the filtered row and the prefetched gray row
__device__ __forceinline__ void blur_step(BlurState &st, const BlurArgs &a, int t) {
    constexpr uint32_t W0 = 18u | (34u << 8) | (48u << 16) | (56u << 24);   // taps 0..3
    constexpr uint32_t W1 = 48u | (34u << 8) | (18u << 16);                 // taps 4..6
    uint32_t d0 = st.raw[K][0];
    const uint32_t d1 = st.raw[K][1];
    uint32_t d2 = st.raw[K][2];
    {   // The row seven steps ahead goes into the slot just consumed (past the segment's last step the index is clamped:
        // a redundant load costs less than the register copies a conditional one brings).  The lane offsets are made
        // opaque so that their zero-extension is not hoisted into 64-bit register pairs: base in SGPRs + 32-bit lane
        // offset is an addressing mode, a 64-bit vector add is an instruction per load.
        const int tn = t + 7 < a.steps ? t + 7 : a.steps - 1;
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + tn, a.h) * a.w;
        uint32_t ol = a.voff_l, oc = a.voff_c, orr = a.voff_r;
        asm volatile("" : "+v"(ol), "+v"(oc), "+v"(orr));
        st.raw[K][0] = *reinterpret_cast<const uint32_t *>(rowp + ol);
        st.raw[K][1] = *reinterpret_cast<const uint32_t *>(rowp + oc);
        st.raw[K][2] = *reinterpret_cast<const uint32_t *>(rowp + orr);
    }
    if (a.edge) {   // BORDER_REFLECT_101: columns -3..-1 are 3..1, columns w..w+2 are w-2..w-4
        asm volatile("" ::: "memory");   // keep this a branch (uniform per wave): as selects it would cost every strip
        if (a.left_fix) d0 = __builtin_amdgcn_perm(d1, d1, 0x01020300u);
        if (a.right_fix) d2 = __builtin_amdgcn_perm(d1, d1, 0x00000102u);
    }
    uint32_t row[4];
    row[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), W0,
                                    __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), W1, 0u, false), false);
    row[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), W0,
                                    __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), W1, 0u, false), false);
    row[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), W0,
                                    __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), W1, 0u, false), false);
    row[3] = __builtin_amdgcn_udot4(d1, W0, __builtin_amdgcn_udot4(d2, W1, 0u, false), false);
#pragma unroll
    for (int i = 0; i < 4; i++) {   // filed with the row before it: the column pass takes two rows per instruction
        st.pk[K][i] = st.last[i] | (row[i] << 16);
        st.last[i] = row[i];
    }
    if (t < 6) return;
    const int y = a.ys - 6 + t;
    // column pass over the window's rows r0 .. r6 = slots K + 1 .. K + 6, K: three v_dot2_u32_u16 on the pairs (r0, r1),
    // (r2, r3), (r4, r5) and one v_mad_u32_u24 for r6 (every sum is below 2^24), the rounding constant rides in as the
    // first addend, and the result byte (bits 16..23 of the sum) is picked with v_perm_b32
    uint32_t sum[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t acc = mad_u24(row[i], 18u, 1u << 15);
        acc = dot2_u16(st.pk[(K + 2) % 7][i], 18u | (34u << 16), acc);
        acc = dot2_u16(st.pk[(K + 4) % 7][i], 48u | (56u << 16), acc);
        sum[i] = dot2_u16(st.pk[(K + 6) % 7][i], 48u | (34u << 16), acc);
    }
    const uint32_t lo = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u);   // bytes: sum0[2], sum1[2], 0, 0
    const uint32_t hi = __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);   // bytes: 0, 0, sum2[2], sum3[2]
    const uint32_t packed = lo | hi;
    if (a.own_lane) {
        uint8_t *rowd = a.dst + (size_t)y * a.w;   // uniform base + 32-bit lane offset, as for the loads
        uint32_t ox = (uint32_t)a.x;
        asm volatile("" : "+v"(ox));
        *reinterpret_cast<uint32_t *>(rowd + ox) = packed;
    }
}

// Rows [y_begin, y_end) of every image are produced (the whole image for the plain blur; the grid ORB extractor blurs pyramid
// levels that carry their own reflect margin and asks for the rows between the margins only); frame_pitch = bytes from one
// image to the next.
__global__ __launch_bounds__(256) void gaussian7_stream_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                               uint8_t *__restrict__ out, int seg_rows, int frames,
                                                               int strips, int per_frame, size_t frame_pitch, int y_begin,
                                                               int y_end) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    int f, blk;   // a frame's strips and segments share an XCD (see min_eigen_stream_kernel)
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    const int strip = blk % strips, segblk = blk / strips;
    BlurArgs a;
    a.ys = y_begin + (segblk * 4 + wave) * seg_rows;
    if (a.ys >= y_end) return;   // whole wave; no barriers in this kernel
    const int ye = a.ys + seg_rows < y_end ? a.ys + seg_rows : y_end;
    a.steps = ye - a.ys + 6;
    a.w = w;
    a.h = h;
    a.src = gray + (size_t)f * frame_pitch;
    a.dst = out + (size_t)f * frame_pitch;
    const int x0 = strip * 256;
    a.x = x0 + 4 * lane;
    a.own_lane = a.x < w;
    a.edge = x0 == 0 || x0 + 256 + 4 > w;
    a.left_fix = a.x == 0;
    a.right_fix = a.x + 4 == w;
    const int xc = a.x > w - 4 ? w - 4 : a.x;
    a.voff_c = (uint32_t)xc;
    a.voff_l = (uint32_t)(xc - 4 < 0 ? 0 : xc - 4);
    a.voff_r = (uint32_t)(xc + 4 > w - 4 ? w - 4 : xc + 4);
    BlurState st;
#pragma unroll
    for (int i = 0; i < 4; i++) st.last[i] = 0u;
#pragma unroll
    for (int k = 0; k < 7; k++) {   // steps >= 7 always (a segment has at least one row)
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + k, h) * w;
        st.raw[k][0] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_l);
        st.raw[k][1] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
        st.raw[k][2] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_r);
    }
    for (int t0 = 0; t0 < a.steps; t0 += 7) {   // seven steps per trip keep every ring index a compile-time constant
#define VS_BLUR_STEP(J) if (t0 + (J) < a.steps) blur_step<(J)>(st, a, t0 + (J));
        VS_BLUR_STEP(0) VS_BLUR_STEP(1) VS_BLUR_STEP(2) VS_BLUR_STEP(3) VS_BLUR_STEP(4) VS_BLUR_STEP(5) VS_BLUR_STEP(6)
#undef VS_BLUR_STEP
    }
}

}  // namespace

int vs_launch_gaussian7(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, uint8_t *out) {
    VS_REQUIRE(ctx, gray && out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 4 && h >= 4, VSLAM_ERR_INVALID);
    VsProfScope ps(ctx, "gaussian7_kernel");
    if (vs_pitch(ctx, w) != w) {
        // Padded rows (vslam_ctx::img_pitch): at least three mirrored columns follow column w - 1, which is all a 7-tap row
        // filter with BORDER_REFLECT_101 reads past it, so the padded plane filtered as an image of `pitch` columns holds the
        // image's result in its first w columns (its own mirroring happens at column pitch - 1, three or more columns away).
        w = vs_pitch(ctx, w);
        VS_REQUIRE(ctx, w % 4 == 0, VSLAM_ERR_INVALID);
    }
    if (w % 4 == 0 && ((reinterpret_cast<uintptr_t>(gray) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
        const int strips = vs_div_up(w, 256);
        const int segs = vs_stream_segments(h, frames, strips);
        const int seg_rows = vs_div_up(h, segs);
        const int per_frame = strips * vs_div_up(segs, 4);
        gaussian7_stream_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(gray, w, h, out, seg_rows, frames,
                                                                                          strips, per_frame, (size_t)w * h, 0, h);
    } else {
        dim3 grid(vs_div_up(w, kBTW), vs_div_up(h, kBTH), frames);
        gaussian7_kernel<<<grid, kBT, 0, ctx->stream>>>(gray, w, h, out);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}


// The same filter on rows [y_begin, y_end) of images of w x h bytes that lie frame_pitch bytes apart (w % 4 == 0, dword-aligned
// rows): the caller's images carry the border values in their own margins, so no row or column this touches is reflected.
int vs_launch_gaussian7_rows(vslam_ctx *ctx, const uint8_t *src, uint8_t *dst, int frames, size_t frame_pitch, int w, int h,
                             int y_begin, int y_end) {
    VS_REQUIRE(ctx, src && dst && frames > 0 && w >= 8 && w % 4 == 0 && frame_pitch % 4 == 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, y_begin >= 3 && y_end <= h - 3 && y_begin < y_end, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0, VSLAM_ERR_INVALID);
    const int rows = y_end - y_begin;
    const int strips = vs_div_up(w, 256);
    const int segs = vs_stream_segments(rows, frames, strips);
    const int seg_rows = vs_div_up(rows, segs);
    const int per_frame = strips * vs_div_up(segs, 4);
    gaussian7_stream_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(src, w, h, dst, seg_rows, frames, strips,
                                                                                      per_frame, frame_pitch, y_begin, y_end);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

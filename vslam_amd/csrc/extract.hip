// Keypoint extraction + rBRIEF description for gfx950.
//
// Replaces extract_features(Frame&), /root/reference/src/Frame.cpp:53-80:
//   cv::cvtColor(BGR2GRAY) :56            -> bgr2gray_kernel
//   cv::goodFeaturesToTrack(...) :61      -> min_eigen_stream_kernel (Sobel + products + 3x3 box + min eigenvalue +
//                                            3x3 local maxima, per-frame max; one wave per column strip, rolling
//                                            registers), corner_select_kernel (exact threshold, greedy
//                                            min-distance as a fixpoint on the best-ranked window, sorted output).
//                                            Widths that are not a multiple of 4, and vslam_min_eigen, use the tiled
//                                            min_eigen_kernel / min_eigen_v4_kernel + corner_candidates_kernel.
//   cv::ORB::compute(gray, kps, desc) :68 -> gaussian7_stream_kernel (gaussian7_kernel for other widths),
//                                            keypoint_border_kernel, rbrief_rotate_kernel + rbrief_lds_kernel
//                                            (rbrief_kernel for other widths)
// The arithmetic follows the oracle (oracle/vso_extract.cpp) operation for operation; float
// steps are written so that no contraction or reassociation can occur (-ffp-contract=off).
//
// The image kernels read every input byte once and write every output once; what bounds them in practice is
// VALU issue for the corner response (exact f64 box sums, correctly rounded sqrt) and HBM for the rest
// (DESIGN.md section 5).
#include "ctx.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);   // clamp only guards halo cells that are never used
}

// monotone float <-> u32 so an unsigned max / radix order is the float order (no NaNs here)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ------------------------------------------------------------------------------------------
// cvtColor(BGR2GRAY), 8U: (b*3735 + g*19235 + r*9798 + 2^14) >> 15
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t *__restrict__ bgr, int w, int h,
                                                       int stride, uint8_t *__restrict__ gray,
                                                       int aligned) {
    const int f = blockIdx.y;
    const int qpr = (w + 3) >> 2;   // 4-pixel groups per row
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * h) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const uint8_t *src = bgr + ((size_t)f * h + y) * stride + 3 * x;
    uint8_t *dst = gray + ((size_t)f * h + y) * w + x;
    if (aligned && x + 3 < w) {
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        const uint32_t a = s4[0], b = s4[1], c = s4[2];   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const uint32_t g0 = gray_of(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF);
        const uint32_t g1 = gray_of(a >> 24, b & 0xFF, (b >> 8) & 0xFF);
        const uint32_t g2 = gray_of((b >> 16) & 0xFF, b >> 24, c & 0xFF);
        const uint32_t g3 = gray_of((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24);
        *reinterpret_cast<uint32_t *>(dst) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (int i = 0; i < 4 && x + i < w; i++) dst[i] = (uint8_t)gray_of(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
    }
}

// ------------------------------------------------------------------------------------------
// cornerMinEigenVal(gray, eig, 3, 3) + per-frame max
// ------------------------------------------------------------------------------------------
constexpr int kET = 256;          // threads
constexpr int kETW = 64, kETH = 16;   // output tile
constexpr int kGW = kETW + 4, kGH = kETH + 4;   // gray tile (halo 2)
constexpr int kHW = kETW + 2;                   // hx / R / cov width (halo 1)

__global__ __launch_bounds__(kET) void min_eigen_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                        float *__restrict__ eig,
                                                        uint32_t *__restrict__ frame_max) {
    __shared__ uint8_t G[kGH][kGW];
    __shared__ float HX[kGH][kHW], RR[kGH][kHW];
    __shared__ float CXX[kETH + 2][kHW], CXY[kETH + 2][kHW], CYY[kETH + 2][kHW];
    __shared__ uint32_t s_max;
    const int f = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * kETW, y0 = blockIdx.y * kETH;
    const uint8_t *src = gray + (size_t)f * w * h;
    if (tid == 0) s_max = 0;

    // gray tile at raw coordinates [x0-2, x0+TW+2) x [y0-2, y0+TH+2), REFLECT_101 filled
    for (int i = tid; i < kGH * kGW; i += kET) {
        const int r = i / kGW, c = i - r * kGW;
        G[r][c] = src[(size_t)reflect101(y0 - 2 + r, h) * w + reflect101(x0 - 2 + c, w)];
    }
    __syncthreads();

    // row pass of both Sobels on raw rows [y0-2, ..), raw cols [x0-1, x0+TW+1)
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)scale, k0 = 2.0f * k1;
    for (int i = tid; i < kGH * kHW; i += kET) {
        const int r = i / kHW, c = i - r * kHW;   // G column of this pixel is c + 1
        const int gm = G[r][c], g0 = G[r][c + 1], gp = G[r][c + 2];
        HX[r][c] = (float)(gp - gm);
        const float a = (float)g0 * k0;
        const float b = (float)(gm + gp) * k1;
        RR[r][c] = a + b;
    }
    __syncthreads();

    // column pass + products on raw rows [y0-1, y0+TH+1)
    for (int i = tid; i < (kETH + 2) * kHW; i += kET) {
        const int r = i / kHW, c = i - r * kHW;   // HX/RR row of this pixel is r + 1
        const float a = HX[r + 1][c] * k0;
        const float b = (HX[r][c] + HX[r + 2][c]) * k1;
        const float dx = a + b;
        const float dy = RR[r + 2][c] - RR[r][c];
        CXX[r][c] = dx * dx;
        CXY[r][c] = dx * dy;
        CYY[r][c] = dy * dy;
    }
    __syncthreads();

    // 3x3 box in double: r(y) = (c(x-1) + c(x)) + c(x+1), S = (r(y-1) + r(y)) + r(y+1), with the
    // box filter's own REFLECT_101 applied to cov coordinates.  Each lane walks 4 rows of one column.
    const int tx = tid & 63, ty = tid >> 6;
    const int x = x0 + tx;
    uint32_t kmax = 0;   // f2ord() of any float is > 0, so 0 is the identity of the max
    if (x < w) {
        const int cm = reflect101(x - 1, w) - (x0 - 1), c0 = tx + 1, cp = reflect101(x + 1, w) - (x0 - 1);
        double rxx[6], rxy[6], ryy[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int yy = y0 + ty * 4 - 1 + k;
            int lr = reflect101(yy, h) - (y0 - 1);
            lr = lr < 0 ? 0 : (lr > kETH + 1 ? kETH + 1 : lr);   // rows past the image are never output
            rxx[k] = ((double)CXX[lr][cm] + (double)CXX[lr][c0]) + (double)CXX[lr][cp];
            rxy[k] = ((double)CXY[lr][cm] + (double)CXY[lr][c0]) + (double)CXY[lr][cp];
            ryy[k] = ((double)CYY[lr][cm] + (double)CYY[lr][c0]) + (double)CYY[lr][cp];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = y0 + ty * 4 + k;
            if (y < h) {
                const float sxx = (float)((rxx[k] + rxx[k + 1]) + rxx[k + 2]);
                const float sxy = (float)((rxy[k] + rxy[k + 1]) + rxy[k + 2]);
                const float syy = (float)((ryy[k] + ryy[k + 1]) + ryy[k + 2]);
                const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
                const float amc = a - c;
                const float t = amc * amc + b * b;
                const float e = (a + c) - sqrtf(t);
                eig[((size_t)f * h + y) * w + x] = e;
                const uint32_t ke = f2ord(e);
                kmax = ke > kmax ? ke : kmax;
            }
        }
    }
    if (frame_max) {
        uint32_t k = kmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(k, off, 64);
            k = o > k ? o : k;
        }
        if ((tid & 63) == 0) atomicMax(&s_max, k);
        __syncthreads();
        if (tid == 0) atomicMax(&frame_max[f], s_max);
    }
}

// Vectorised form (width % 4 == 0): 256x32 tile, one lane = 4 adjacent pixels x 8 rows walked top
// to bottom with every intermediate (row-pass Sobel terms, products, double row sums) in a rolling
// register window; only the gray tile lives in LDS.
//
// Border rule used here: cornerEigenValsVecs box-filters the product images with REFLECT_101, i.e.
// the product at row -1 is the product at row 1.  Evaluating the derivative stencils at raw row -1
// on the reflect-filled gray tile gives Dx(-1) = Dx(1) and Dy(-1) = -Dy(1) exactly (a - b ==
// -(b - a) in IEEE), so dx*dx and dy*dy are already right and dx*dy only needs its sign flipped; the
// same holds per mirrored column with the roles of Dx and Dy swapped.  Negation commutes with every
// rounding, so flipping the sign of the xy product of mirrored rows/columns is bit-exact.
constexpr int kE4W = 256, kE4H = 32, kE4C = kE4W / 4 + 2;

// v_max3_f32 on values that are never NaN: fmaxf() would first canonicalise every operand (one extra
// v_max_f32 each).  Pure register instruction.
__device__ __forceinline__ float max3_nonan(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Correctly rounded sqrt of four non-negative finite floats.  Fast path: v_sqrt_f32 (1 ulp) plus the
// two-sided fma residual test; valid for 0 and for inputs >= 2^-96 (v_sqrt_f32 flushes denormal inputs, which
// is why the general sequence rescales).  If any lane of the wave holds a smaller non-zero input the whole
// wave takes sqrtf().  Both paths return the IEEE result, so which one runs never shows in the output.
__device__ __forceinline__ float sqrt_rn_fast1(float t) {
    const float r = __builtin_amdgcn_sqrtf(t);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, t), e_up = __builtin_fmaf(-r_up, r, t);
    float o = e_dn <= 0.f ? r_dn : r;   // t == 0: r_dn is NaN, the comparison is false
    o = e_up > 0.f ? r_up : o;
    return o;
}
__device__ __forceinline__ void sqrt_rn4(const float t[4], float out[4]) {
    // bits - 1 < 0x0F800000 - 1  <=>  0 < t < 2^-96  (t >= 0, so the bit pattern orders like the value)
    const uint32_t a = __float_as_uint(t[0]) - 1u, b = __float_as_uint(t[1]) - 1u, c = __float_as_uint(t[2]) - 1u,
                   d = __float_as_uint(t[3]) - 1u;
    const uint32_t lo = min(min(a, b), min(c, d));
    if (__builtin_expect(__any(lo < 0x0F800000u - 1u), 0)) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = sqrtf(t[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = sqrt_rn_fast1(t[i]);
    }
}

// Plain cornerMinEigenVal for vslam_min_eigen (the front-end path uses min_eigen_stream_kernel below, which
// follows this kernel's arithmetic): a 256x32 tile per workgroup, every pixel of it owned.
__global__ __launch_bounds__(256) void min_eigen_v4_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                           float *__restrict__ eig,
                                                           uint32_t *__restrict__ frame_max) {
    __shared__ uint32_t G[kE4H + 4][kE4C];   // bytes x0-4 .. x0+259 of raw rows y0-2 .. y0+33
    __shared__ uint32_t s_max;
    const int f = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * kE4W, y0 = blockIdx.y * kE4H;
    const uint8_t *src = gray + (size_t)f * w * h;
    if (tid == 0) s_max = 0;
    for (int i = tid; i < (kE4H + 4) * kE4C; i += 256) {
        const int r = i / kE4C, c = i - r * kE4C;
        const int xs = x0 - 4 + 4 * c;
        const uint8_t *row = src + (size_t)reflect101(y0 - 2 + r, h) * w;
        uint32_t v;
        if (xs >= 0 && xs + 3 < w) {
            v = *reinterpret_cast<const uint32_t *>(row + xs);
        } else {
            v = (uint32_t)row[reflect101(xs, w)] | ((uint32_t)row[reflect101(xs + 1, w)] << 8) |
                ((uint32_t)row[reflect101(xs + 2, w)] << 16) | ((uint32_t)row[reflect101(xs + 3, w)] << 24);
        }
        G[r][c] = v;
    }
    __syncthreads();

    const int lane = tid & 63, grp = tid >> 6;
    const int x = x0 + 4 * lane;
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)scale, k0 = 2.0f * k1;
    const float ninf = -__builtin_inff();
    // tiles whose 6-wide / 3-tall product windows never leave the image skip the mirror-sign logic
    const bool interior = x0 >= 4 && x0 + kE4W + 4 <= w && y0 >= 2 && y0 + kE4H + 2 <= h;
    float emax = ninf;
    if (x >= 0 && x < w) {
        bool colflip[6];
#pragma unroll
        for (int c = 0; c < 6; c++) colflip[c] = (x - 1 + c < 0) || (x - 1 + c >= w);
        float hx[3][6], rr[3][6];
        // column sums S(y) = (r(y-1) + r(y)) + r(y+1) carried as: prev = r(y), pair = r(y-1) + r(y)
        double prev[12], pair[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int t = grp * 8 + k;   // tile row; raw image row y0 - 2 + t
            const uint32_t d0 = G[t][lane], d1 = G[t][lane + 1], d2 = G[t][lane + 2];
            float g[8];   // gray at columns x-2 .. x+5 = bytes 2..9 of the 12-byte window (the compiler emits v_cvt_f32_ubyteN)
            g[0] = (float)((d0 >> 16) & 0xFFu); g[1] = (float)(d0 >> 24);
            g[2] = (float)((d1 >> 0) & 0xFFu); g[3] = (float)((d1 >> 8) & 0xFFu);
            g[4] = (float)((d1 >> 16) & 0xFFu); g[5] = (float)(d1 >> 24);
            g[6] = (float)((d2 >> 0) & 0xFFu); g[7] = (float)((d2 >> 8) & 0xFFu);
#pragma unroll
            for (int c = 0; c < 6; c++) {
                hx[k % 3][c] = g[c + 2] - g[c];          // small integers: exact in float
                const float a = g[c + 1] * k0;
                const float b = (g[c] + g[c + 2]) * k1;  // the integer sum is exact in float
                rr[k % 3][c] = a + b;
            }
            if (k >= 2) {
                // products on raw row (y0 - 2 + t) - 1
                const int crow = y0 - 3 + t;
                const bool rowflip = crow < 0 || crow >= h;
                float cxx[6], cxy[6], cyy[6];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const float a = hx[(k - 1) % 3][c] * k0;
                    const float b = (hx[(k - 2) % 3][c] + hx[k % 3][c]) * k1;
                    const float dx = a + b;
                    const float dy = rr[k % 3][c] - rr[(k - 2) % 3][c];
                    cxx[c] = dx * dx;
                    const float xy = dx * dy;
                    cxy[c] = (!interior && (rowflip != colflip[c])) ? -xy : xy;
                    cyy[c] = dy * dy;
                }
                double cur[12];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    cur[i] = ((double)cxx[i] + (double)cxx[i + 1]) + (double)cxx[i + 2];
                    cur[4 + i] = ((double)cxy[i] + (double)cxy[i + 1]) + (double)cxy[i + 2];
                    cur[8 + i] = ((double)cyy[i] + (double)cyy[i + 1]) + (double)cyy[i + 2];
                }
                if (k >= 4) {
                    const int lr = grp * 8 + (k - 4);   // row inside the tile
                    const int y = y0 + lr;
                    float e4[4] = {ninf, ninf, ninf, ninf};
                    if (y >= 0 && y < h) {
                        float apc[4], tt[4], rt[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float sxx = (float)(pair[i] + cur[i]);
                            const float sxy = (float)(pair[4 + i] + cur[4 + i]);
                            const float syy = (float)(pair[8 + i] + cur[8 + i]);
                            const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
                            const float amc = a - c;
                            tt[i] = amc * amc + b * b;
                            apc[i] = a + c;
                        }
                        sqrt_rn4(tt, rt);
#pragma unroll
                        for (int i = 0; i < 4; i++) e4[i] = apc[i] - rt[i];
                        emax = max3_nonan(max3_nonan(e4[0], e4[1], e4[2]), e4[3], emax);
                        *reinterpret_cast<float4 *>(eig + ((size_t)f * h + y) * w + x) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    if (k >= 3) pair[i] = prev[i] + cur[i];
                    prev[i] = cur[i];
                }
            }
        }
    }
    const uint32_t kmax = emax == ninf ? 0u : f2ord(emax);   // 0 is the identity of the ordered-key max
    if (frame_max) {
        uint32_t k = kmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(k, off, 64);
            k = o > k ? o : k;
        }
        if ((tid & 63) == 0) atomicMax(&s_max, k);
        __syncthreads();
        if (tid == 0) atomicMax(&frame_max[f], s_max);   // fire and forget
    }
}

// ------------------------------------------------------------------------------------------
// Streaming form of the fused response + 3x3-maxima kernel (width % 4 == 0)
// ------------------------------------------------------------------------------------------
// One wave owns a column strip (64 lanes x 4 pixels) and walks down a segment of rows.  Everything that the
// tile form recomputes at tile seams rolls in registers instead: the two previous rows of horizontal Sobel
// parts, the two previous rows of horizontal product sums, the two previous rows of responses and of their
// horizontal 3-maxima.  The corner threshold needs the frame's maximum, which is not known yet, so candidates
// are prefiltered with a running maximum (always <= the final one, hence a superset) and
// corner_select_kernel applies the exact threshold.  Gray rows come straight from
// global memory (three coalesced dwords per lane and row, issued three rows ahead), responses of the
// neighbouring lanes come through DPP wave shifts, so the kernel uses no LDS except the candidate queue
// and has no barriers.  Arithmetic and its order are those of min_eigen_v4_kernel (see there).
//
// Step t of a segment owning rows [ys, ye):   gray row g = ys - 3 + t   (Sobel parts of row g)
//   t >= 2: products and their horizontal sums on row p = g - 1
//   t >= 4: response row y = p - 1 = ys - 5 + t  (stored when ys <= y < ye)
//   t >= 6: 3x3-maxima test of row y - 1 = ys - 6 + t  -> candidates
// so a segment takes (ye - ys) + 6 steps, 6 of them warm-up (7 % at 90 rows per segment).
constexpr int kSW = 256;     // pixels per strip (64 lanes x 4), all owned
constexpr int kSQ = 512;     // candidate queue entries per wave
// Strips do not overlap, so the 3x3 test of a strip's first and last column lacks the neighbouring strip's
// column.  Such candidates are emitted with a flag in the (otherwise unused) top bits of the pixel offset and
// corner_select_kernel completes their test against the stored responses before anything else looks at them.
constexpr uint32_t kKeyCheckLeft = 0x80000000u, kKeyCheckRight = 0x40000000u;
constexpr uint32_t kOffMask = 0x3FFFFFFFu;   // pixel offset part of a key's low word

struct StreamState {
    float hx[3][6], rr[3][6];      // per gray row: x-derivative parts and smoothed values, columns x-1 .. x+4
    double S[3][12];               // per product row: horizontal 3-sums of xx, xy, yy for the lane's 4 pixels
    float ctr[3][4], hm[3][4];     // per response row: the values and their horizontal 3-maxima
    uint32_t raw[3][3];            // prefetched gray dwords (x-4, x, x+4) of the next three rows
    float emax;
};

template <int B>
__device__ __forceinline__ float cvt_ubyte(uint32_t d) {   // (float) of byte B of d
    float r;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(d));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(d));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(d));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(d));
    return r;
}
__device__ __forceinline__ float dpp_wave_shr1(float v, float fill) {   // lane i <- lane i-1, lane 0 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shl1(float v, float fill) {   // lane i <- lane i+1, lane 63 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

struct StreamArgs {
    const uint8_t *src;          // frame base
    float *eig;                  // frame base
    unsigned long long *queue;   // this wave's LDS queue
    unsigned long long *keys;    // frame base
    uint32_t *count;             // this frame's candidate counter
    size_t key_cap;
    int w, h, ys, ye, x, steps;
    uint32_t voff_l, voff_c, voff_r;
    bool edge, left_fix, right_fix, own_lane;
    float k0, k1, thr_p;
};

__device__ __forceinline__ void stream_flush(const StreamArgs &a, int &qn, int lane) {
    if (qn == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.count, (uint32_t)qn);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < qn; i += 64) {
        const size_t pos = (size_t)base + i;
        if (pos < a.key_cap) a.keys[pos] = a.queue[i];
    }
    qn = 0;
}

template <int K>
__device__ __forceinline__ void stream_step(StreamState &st, const StreamArgs &a, int t, int &qn, int lane) {
    constexpr int K1 = (K + 2) % 3, K2 = (K + 1) % 3;   // slots of the previous row and the one before
    uint32_t d0 = st.raw[K][0];
    const uint32_t d1 = st.raw[K][1];
    uint32_t d2 = st.raw[K][2];
    if (t + 3 < a.steps) {   // prefetch the row three steps ahead into the slot just consumed
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + t + 3, a.h) * a.w;
        st.raw[K][0] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_l);
        st.raw[K][1] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
        st.raw[K][2] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_r);
    }
    if (a.edge) {   // BORDER_REFLECT_101 in x: columns -2, -1 are columns 2, 1; columns w, w+1 are w-2, w-3
        if (a.left_fix) d0 = (d1 & 0x00FF0000u) | ((d1 & 0x0000FF00u) << 16);
        if (a.right_fix) d2 = ((d1 >> 16) & 0xFFu) | (d1 & 0xFF00u);
    }
    // gray at columns x-2 .. x+5.  The conversions are opaque to the compiler on purpose: it would otherwise
    // rewrite float(a) - float(b) as float(a - b) with byte-select integer ops, which issue slower here
    // than one v_cvt_f32_ubyteN per pixel plus plain float subtract / add (tools/valu_rate.hip).
    float g[8];
    g[0] = cvt_ubyte<2>(d0); g[1] = cvt_ubyte<3>(d0);
    g[2] = cvt_ubyte<0>(d1); g[3] = cvt_ubyte<1>(d1); g[4] = cvt_ubyte<2>(d1); g[5] = cvt_ubyte<3>(d1);
    g[6] = cvt_ubyte<0>(d2); g[7] = cvt_ubyte<1>(d2);
#pragma unroll
    for (int c = 0; c < 6; c++) {
        st.hx[K][c] = g[c + 2] - g[c];
        const float p = g[c + 1] * a.k0;
        const float q = (g[c] + g[c + 2]) * a.k1;
        st.rr[K][c] = p + q;
    }
    if (t < 2) return;

    // products on row p = g - 1
    const int prow = a.ys - 4 + t;
    const bool rowflip = prow < 0 || prow >= a.h;
    float cxx[6], cxy[6], cyy[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const float p = st.hx[K1][c] * a.k0;
        const float q = (st.hx[K2][c] + st.hx[K][c]) * a.k1;
        const float dx = p + q;
        const float dy = st.rr[K][c] - st.rr[K2][c];
        cxx[c] = dx * dx;
        cxy[c] = dx * dy;
        cyy[c] = dy * dy;
    }
    if (a.edge) {    // mirrored column: the xy product changes sign (see min_eigen_v4_kernel)
        if (a.left_fix) cxy[0] = -cxy[0];
        if (a.right_fix) cxy[5] = -cxy[5];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        st.S[K][i] = ((double)cxx[i] + (double)cxx[i + 1]) + (double)cxx[i + 2];
        st.S[K][4 + i] = ((double)cxy[i] + (double)cxy[i + 1]) + (double)cxy[i + 2];
        st.S[K][8 + i] = ((double)cyy[i] + (double)cyy[i + 1]) + (double)cyy[i + 2];
    }
    if (rowflip) {   // mirrored row (two per frame): same rule; negating the sums equals summing the negated products
        asm volatile("" ::: "memory");   // keep this a branch: as selects it would cost every row
#pragma unroll
        for (int i = 0; i < 4; i++) st.S[K][4 + i] = -st.S[K][4 + i];
    }
    if (t >= 4) {
        const int y = a.ys - 5 + t;
        float apc[4], tt[4], rt[4], e4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // column sums S(y) = (r(y-1) + r(y)) + r(y+1), rows in slots K2, K1, K
            const float sxx = (float)((st.S[K2][i] + st.S[K1][i]) + st.S[K][i]);
            const float sxy = (float)((st.S[K2][4 + i] + st.S[K1][4 + i]) + st.S[K][4 + i]);
            const float syy = (float)((st.S[K2][8 + i] + st.S[K1][8 + i]) + st.S[K][8 + i]);
            const float ea = sxx * 0.5f, eb = sxy, ec = syy * 0.5f;
            const float amc = ea - ec;
            tt[i] = amc * amc + eb * eb;
            apc[i] = ea + ec;
        }
        sqrt_rn4(tt, rt);
#pragma unroll
        for (int i = 0; i < 4; i++) e4[i] = apc[i] - rt[i];
        if (y >= a.ys && y < a.ye && a.own_lane) {
            st.emax = max3_nonan(max3_nonan(e4[0], e4[1], e4[2]), e4[3], st.emax);
            *reinterpret_cast<float4 *>(a.eig + (size_t)y * a.w + a.x) = make_float4(e4[0], e4[1], e4[2], e4[3]);
        }
        // rows y < 0 or y >= h, and the pixels of lanes outside the image, are never a neighbour of a testable
        // pixel (tests cover rows 1 .. h-2 and columns 1 .. w-2), so their values need no special marking
        const float ninf = -__builtin_inff();   // what lane 0 / lane 63 see beyond the strip: resolved later (kKeyCheck*)
        const float lf = dpp_wave_shr1(e4[3], ninf), rg = dpp_wave_shl1(e4[0], ninf);
#pragma unroll
        for (int i = 0; i < 4; i++) st.ctr[K][i] = e4[i];
        st.hm[K][0] = max3_nonan(lf, e4[0], e4[1]);
        st.hm[K][1] = max3_nonan(e4[0], e4[1], e4[2]);
        st.hm[K][2] = max3_nonan(e4[1], e4[2], e4[3]);
        st.hm[K][3] = max3_nonan(e4[2], e4[3], rg);
        if (t >= 6) {
            const int ty = y - 1;   // ys <= ty < ye by construction
            if (qn > kSQ - 256) stream_flush(a, qn, lane);
            const bool row_ok = ty >= 1 && ty < a.h - 1 && a.own_lane;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = st.ctr[K1][i];
                const float m = max3_nonan(st.hm[K2][i], st.hm[K1][i], st.hm[K][i]);
                const int xx = a.x + i;
                const bool cand = row_ok && xx >= 1 && xx < a.w - 1 && v > a.thr_p && !(m > v);
                const unsigned long long bal = __ballot(cand);
                if (bal) {
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    uint32_t lo = (uint32_t)(ty * a.w + xx);
                    if (i == 0 && lane == 0) lo |= kKeyCheckLeft;     // xx >= 1 here, so a strip lies to the left
                    if (i == 3 && lane == 63) lo |= kKeyCheckRight;   // xx < w - 1 here, so a strip lies to the right
                    if (cand) a.queue[pos] = ((unsigned long long)f2ord(v) << 32) | lo;
                    qn += __popcll(bal);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void min_eigen_stream_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                               float *__restrict__ eig, uint32_t *__restrict__ frame_max,
                                                               double quality, unsigned long long *__restrict__ keys,
                                                               uint32_t *__restrict__ counts, size_t key_cap, int seg_rows,
                                                               int frames, int strips, int per_frame) {
    __shared__ unsigned long long queue[4][kSQ];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // all strips and segments of a frame run on one XCD: the cache lines two neighbouring strips (or segments)
    // both touch are then fetched from HBM once, into that XCD's L2
    int f, blk;
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    const int strip = blk % strips, segblk = blk / strips;
    StreamArgs a;
    a.ys = (segblk * 4 + wave) * seg_rows;
    if (a.ys >= h) return;   // whole wave; the kernel has no barriers
    a.ye = a.ys + seg_rows < h ? a.ys + seg_rows : h;
    a.steps = a.ye - a.ys + 6;
    a.w = w;
    a.h = h;
    a.src = gray + (size_t)f * w * h;
    a.eig = eig + (size_t)f * w * h;
    a.queue = queue[wave];
    a.keys = keys + (size_t)f * key_cap;
    a.count = counts + f;
    a.key_cap = key_cap;
    const int x0 = strip * kSW;
    a.x = x0 + 4 * lane;
    a.own_lane = a.x < w;
    a.edge = x0 == 0 || x0 + kSW + 4 > w;
    a.left_fix = a.x == 0;
    a.right_fix = a.x + 4 == w;
    const int xc = a.x < 0 ? 0 : (a.x > w - 4 ? w - 4 : a.x);
    a.voff_c = (uint32_t)xc;
    a.voff_l = (uint32_t)(xc - 4 < 0 ? 0 : xc - 4);
    a.voff_r = (uint32_t)(xc + 4 > w - 4 ? w - 4 : xc + 4);
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    a.k1 = (float)scale;
    a.k0 = 2.0f * a.k1;
    const float ninf = -__builtin_inff();

    StreamState st;
    st.emax = ninf;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 12; i++) st.S[k][i] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + k, h) * w;
        st.raw[k][0] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_l);
        st.raw[k][1] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
        st.raw[k][2] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_r);
    }
    // candidate prefilter: the best maximum known so far (other waves publish theirs as they go); always <= the
    // frame's final maximum, so the candidates are a superset and corner_select_kernel applies the exact threshold
    uint32_t run_max = frame_max[f];
    a.thr_p = (float)((double)ord2f(run_max) * quality);
    if (run_max == 0u) a.thr_p = ninf;   // nothing published yet
    int qn = 0;
    for (int t0 = 0; t0 < a.steps; t0 += 3) {
        stream_step<0>(st, a, t0, qn, lane);
        if (t0 + 1 < a.steps) stream_step<1>(st, a, t0 + 1, qn, lane);
        if (t0 + 2 < a.steps) stream_step<2>(st, a, t0 + 2, qn, lane);
        if ((t0 % 12) == 9) {   // every 12 rows: tighten the prefilter with this wave's own maximum and publish it
            uint32_t k = st.emax == ninf ? 0u : f2ord(st.emax);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(k, off, 64);
                k = o > k ? o : k;
            }
            if (k > run_max) {   // fire and forget: later waves start from it (re-reading it here would stall the wave)
                run_max = k;
                if (lane == 0) atomicMax(&frame_max[f], k);
            }
            if (run_max != 0u) a.thr_p = (float)((double)ord2f(run_max) * quality);
        }
    }
    {
        uint32_t k = st.emax == ninf ? 0u : f2ord(st.emax);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(k, off, 64);
            k = o > k ? o : k;
        }
        if (lane == 0 && k != 0u) atomicMax(&frame_max[f], k);
    }
    stream_flush(a, qn, lane);
}

// ------------------------------------------------------------------------------------------
// threshold + 3x3 local maximum -> candidate keys (response << 32 | pixel offset)
// ------------------------------------------------------------------------------------------
constexpr int kCT = 256, kCTW = 64, kCTH = 16;

__global__ __launch_bounds__(kCT) void corner_candidates_kernel(
    const float *__restrict__ eig, int w, int h, const uint32_t *__restrict__ frame_max, double quality,
    unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts,
    size_t key_cap) {
    __shared__ float E[kCTH + 2][kCTW + 2];
    __shared__ uint32_t s_cnt, s_base;
    const int f = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * kCTW, y0 = blockIdx.y * kCTH;
    const float *src = eig + (size_t)f * w * h;
    const float mx = ord2f(frame_max[f]);
    const float thr = (float)((double)mx * quality);   // threshold(eig, maxVal*qualityLevel, THRESH_TOZERO)
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < (kCTH + 2) * (kCTW + 2); i += kCT) {
        const int r = i / (kCTW + 2), c = i - r * (kCTW + 2);
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        // outside the image the dilate sees nothing: -inf never wins a max
        float v = -__builtin_inff();
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            v = src[(size_t)yy * w + xx];
            v = v > thr ? v : 0.f;   // THRESH_TOZERO
        }
        E[r][c] = v;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    unsigned long long mykeys[4];
    int nk = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ly = ty * 4 + k, y = y0 + ly, x = x0 + tx;
        if (x < w && y < h) {
            const float v = E[ly + 1][tx + 1];
            bool cand = false;
            if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1 && v != 0.f) {   // interior only, val != 0
                float m = v;
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const float nb = E[ly + dy][tx + dx];
                        m = nb > m ? nb : m;
                    }
                cand = (m == v);   // val == dilate(val) on the thresholded image
            }
            if (cand) mykeys[nk++] = ((unsigned long long)f2ord(v) << 32) | (uint32_t)(y * w + x);
        }
    }
    uint32_t my_off = 0;
    if (nk) my_off = atomicAdd(&s_cnt, (uint32_t)nk);
    __syncthreads();
    if (tid == 0 && s_cnt) s_base = atomicAdd(&counts[f], s_cnt);
    __syncthreads();
    for (int i = 0; i < nk; i++) {
        const size_t pos = (size_t)s_base + my_off + i;
        if (pos < key_cap) keys[(size_t)f * key_cap + pos] = mykeys[i];
    }
}

// ------------------------------------------------------------------------------------------
// greedy min-distance suppression + first-N by rank, one workgroup per frame
// ------------------------------------------------------------------------------------------
// goodFeaturesToTrack sorts candidates by (value desc, address desc) and accepts a candidate iff
// no already-accepted corner lies within minDistance (featureselect.cpp; the cell grid there is
// only an index).  Acceptance of c depends on higher-ranked candidates within that radius only,
// so the sequential scan is the least fixpoint of
//     c accepted  <=> every higher-ranked neighbour is rejected
//     c rejected  <=> some higher-ranked neighbour is accepted
// which is reached by rounds in which every undecided candidate inspects its neighbourhood;
// each round settles at least the highest-ranked undecided candidate.  The first maxCorners
// accepted in rank order are then the reference's output, in its order.
constexpr int kST = 1024;

struct SelectShared {
    uint32_t wave_cnt[kST / 64];
    uint32_t hist[256];
    uint32_t flag, fill, need, d_star, above;
    unsigned long long prefix;
};

// One fixpoint visit of candidate `off`: 2 = accepted, 3 = rejected, 1 = still blocked by an
// undecided higher-ranked neighbour.  Rank = (response desc, address desc).
__device__ __forceinline__ int nms_visit(const float *__restrict__ E, const uint8_t *S, int w, int h, uint32_t off,
                                         int R, float min_dist_sq) {
    const int y = off / w, x = off - y * w;
    const float val = E[off];
    bool blocked = false;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t noff = (uint32_t)(yy * w + xx);
            const uint8_t sn = S[noff];
            if (sn == 0 || sn == 3) continue;
            const float vn = E[noff];
            const bool higher = (vn > val) || (vn == val && noff > off);
            if (!higher) continue;
            if (sn == 2) return 3;
            blocked = true;
        }
    }
    return blocked ? 1 : 2;
}

// Key T such that exactly `want` of the n distinct keys are >= T (0 when want >= n): MSB radix
// select, 8 bits per pass, stopping as soon as a whole bucket is wanted.
__device__ unsigned long long radix_select_nth(const unsigned long long *K, uint32_t n, uint32_t want,
                                               SelectShared &sh) {
    if (want >= n) return 0ull;
    const int tid = threadIdx.x;
    unsigned long long prefix = 0;
    uint32_t need = want;
    int known_bits = 0;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        __syncthreads();
        for (int i = tid; i < 256; i += kST) sh.hist[i] = 0;
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kST) {
            const unsigned long long key = K[i];
            if (known_bits == 0 || (key >> (64 - known_bits)) == (prefix >> (64 - known_bits)))
                atomicAdd(&sh.hist[(uint32_t)(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid < 256) {   // bucket d is the one where the running count (from the top) crosses `need`
            uint32_t above = 0;
            for (int d = tid + 1; d < 256; d++) above += sh.hist[d];
            const uint32_t mine = sh.hist[tid];
            if (above < need && need <= above + mine) {
                sh.prefix = prefix | ((unsigned long long)tid << shift);
                sh.need = need - above;
                sh.flag = (need - above == mine) ? 1u : 0u;   // whole bucket wanted: done
            }
        }
        __syncthreads();
        prefix = sh.prefix;
        need = sh.need;
        known_bits += 8;
        if (sh.flag) break;
    }
    __syncthreads();
    return prefix;
}

__device__ void bitonic_sort_desc_lds(unsigned long long *buf, int cap) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= cap; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < cap; i += kST) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = buf[i], b = buf[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        buf[i] = b;
                        buf[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
}

// The same network for cap = EPT * kST keys with EPT consecutive keys per thread held in registers:
// strides below EPT are compare-exchanges inside a thread, strides below 64 * EPT are wave shuffles, and
// only the strides that cross waves (10 of the 78 stages at cap = 4096) go through LDS and barriers.
template <int EPT>
__device__ void bitonic_sort_desc_regs(unsigned long long *buf) {
    const int tid = threadIdx.x;
    constexpr int cap = EPT * kST;
    unsigned long long v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) v[e] = buf[EPT * tid + e];
    for (int k = 2; k <= cap; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64 * EPT; j >>= 1) {   // partner in another wave
            const int tj = j / EPT;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; e++) buf[EPT * tid + e] = v[e];
            __syncthreads();
            const bool lower = (tid & tj) == 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const unsigned long long p = buf[EPT * (tid ^ tj) + e];
                const bool desc = ((EPT * tid + e) & k) == 0;
                const bool keep_max = desc == lower;
                v[e] = keep_max ? (v[e] > p ? v[e] : p) : (v[e] < p ? v[e] : p);
            }
        }
        for (; j >= EPT; j >>= 1) {   // partner in another lane of this wave
            const int tj = j / EPT;
            const bool lower = (tid & tj) == 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const unsigned long long p = __shfl_xor(v[e], tj, 64);
                const bool desc = ((EPT * tid + e) & k) == 0;
                const bool keep_max = desc == lower;
                v[e] = keep_max ? (v[e] > p ? v[e] : p) : (v[e] < p ? v[e] : p);
            }
        }
#pragma unroll
        for (int jj = EPT / 2; jj > 0; jj >>= 1) {   // partner in this thread
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    if ((e & jj) == 0) {
                        const bool desc = ((EPT * tid + e) & k) == 0;
                        const unsigned long long a = v[e], b = v[e | jj];
                        const bool swap = desc ? (a < b) : (a > b);
                        v[e] = swap ? b : a;
                        v[e | jj] = swap ? a : b;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; e++) buf[EPT * tid + e] = v[e];
    __syncthreads();
}

// buf[0 .. cap) sorted descending, cap a power of two; every thread of the workgroup calls it.
// EPT = cap / kST is a template parameter of the selection kernel (0: cap < kST) so that each instantiation only
// carries the registers of the network it uses.
template <int EPT>
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *buf, int cap) {
    __syncthreads();
    if (EPT == 0) bitonic_sort_desc_lds(buf, cap);
    else bitonic_sort_desc_regs<(EPT > 0 ? EPT : 1)>(buf);
}

// Gather the keys >= T into LDS (unordered) and sort them descending; returns how many.
template <int EPT>
__device__ uint32_t gather_sorted(const unsigned long long *K, uint32_t n, unsigned long long T,
                                  unsigned long long *sortbuf, int sort_cap, SelectShared &sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) sh.fill = 0;
    for (int i = tid; i < sort_cap; i += kST) sortbuf[i] = 0ull;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kST) {
        const unsigned long long key = K[i];
        if (key >= T) {
            const uint32_t p = atomicAdd(&sh.fill, 1u);
            if (p < (uint32_t)sort_cap) sortbuf[p] = key;
        }
    }
    __syncthreads();
    bitonic_sort_desc<EPT>(sortbuf, sort_cap);
    return sh.fill < (uint32_t)sort_cap ? sh.fill : (uint32_t)sort_cap;
}

// ---- rank window in two passes over the candidate list ---------------------------------------
// Every kept key has thr < response <= max, both known, so the ordered responses share the top
// L = clz(ord(thr) ^ ord(max)) bits; the next log2(bins) bits form a monotone digit.  Pass 1 histograms the
// digits of the kept keys (the histogram borrows the sort buffer), a suffix scan finds the digit d* in
// whose bin the N-th best key lies, pass 2 gathers every kept key with digit >= d* into the sort buffer,
// which is then sorted: its first N entries are the N best-ranked candidates.  Returns false when the
// gathered set would not fit the buffer (heavy ties); the caller then uses the generic radix select.
// n_kept receives the number of keys above the threshold, N_io is clamped to it.
template <int EPT>
__device__ bool rank_window_2pass(unsigned long long *K, uint32_t n, unsigned long long tkey, uint32_t t32,
                                  uint32_t m32, uint32_t &N_io, unsigned long long *sortbuf, int sort_cap,
                                  SelectShared &sh, uint32_t &n_kept, const float *__restrict__ E, int w) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *hist = reinterpret_cast<uint32_t *>(sortbuf);
    const int bins = 2 * sort_cap < 4096 ? 2 * sort_cap : 4096;
    const int db = 31 - __clz(bins);
    const int L = (t32 ^ m32) ? __clz(t32 ^ m32) : 32;
    const int shift = 32 - L - db > 0 ? 32 - L - db : 0;
    const uint32_t dmask = (uint32_t)bins - 1u;
    __syncthreads();
    for (int i = tid; i < bins; i += kST) hist[i] = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * kST) {   // four independent loads in flight per thread
        unsigned long long key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * kST + tid;
            key[u] = i < n ? K[i] : 0ull;   // 0 never passes the threshold test
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t lo = (uint32_t)key[u];
            if (lo & (kKeyCheckLeft | kKeyCheckRight)) {
                // candidate on the first / last column of a detector strip: finish its 3x3 test with the column the
                // strip could not see, then store the key without the flag (or 0: never a candidate)
                const uint32_t off = lo & kOffMask;
                const float v = ord2f((uint32_t)(key[u] >> 32));
                bool ok = true;
                if (lo & kKeyCheckLeft) {
                    const float *e = E + (size_t)off - w - 1;
                    ok = ok && !(e[0] > v) && !(e[w] > v) && !(e[2 * (size_t)w] > v);
                }
                if (lo & kKeyCheckRight) {
                    const float *e = E + (size_t)off - w + 1;
                    ok = ok && !(e[0] > v) && !(e[w] > v) && !(e[2 * (size_t)w] > v);
                }
                key[u] = ok ? ((key[u] & 0xFFFFFFFF00000000ull) | off) : 0ull;
                K[i0 + u * kST + tid] = key[u];
            }
            if (key[u] > tkey) atomicAdd(&hist[((uint32_t)(key[u] >> 32) >> shift) & dmask], 1u);
        }
    }
    __syncthreads();
    // suffix scan: thread t owns `per` consecutive bins; above = keys in bins owned by higher threads
    const int per = bins > kST ? bins / kST : 1;
    const int lo = tid * per;
    uint32_t own = 0;
    if (lo < bins)
        for (int b = 0; b < per; b++) own += hist[lo + b];
    uint32_t incl = own;   // becomes the sum over lanes >= lane of this wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += o;
    }
    if (lane == 0) sh.wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t higher = 0, total = 0;
    for (int wv = 0; wv < kST / 64; wv++) {
        const uint32_t c = sh.wave_cnt[wv];
        if (wv > wave) higher += c;
        total += c;
    }
    n_kept = total;
    uint32_t N = N_io;
    if (N > total) N = total;
    if (N > (uint32_t)sort_cap) N = (uint32_t)sort_cap;
    N_io = N;
    if (N == 0) {
        __syncthreads();
        return true;
    }
    uint32_t run = higher + incl - own;
    if (lo < bins)
        for (int b = per - 1; b >= 0; b--) {
            const uint32_t mine = hist[lo + b];
            if (run < N && N <= run + mine) {   // exactly one bin satisfies this
                sh.d_star = (uint32_t)(lo + b);
                sh.above = run;
                sh.need = mine;
            }
            run += mine;
        }
    __syncthreads();
    const uint32_t d_star = sh.d_star;
    const uint32_t gathered = sh.above + sh.need;
    __syncthreads();   // every thread is done with the histogram (and sh.*) before the buffer is reused
    if (gathered > (uint32_t)sort_cap) return false;
    if (tid == 0) sh.fill = 0;
    for (int i = tid; i < sort_cap; i += kST) sortbuf[i] = 0ull;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * kST) {
        unsigned long long key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * kST + tid;
            key[u] = i < n ? K[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (key[u] > tkey && (((uint32_t)(key[u] >> 32) >> shift) & dmask) >= d_star)
                sortbuf[atomicAdd(&sh.fill, 1u)] = key[u];
    }
    __syncthreads();
    bitonic_sort_desc<EPT>(sortbuf, sort_cap);
    return true;
}

// ---- suppression on the rank window, entirely in LDS ---------------------------------------------
// After the sort the responses are no longer needed (rank == index), so the window is compacted to
// offs[i] = pixel offset | status << 30 (status 0 undecided, 1 accepted, 2 rejected) in the first half of
// the buffer, and the second half becomes an open-addressing table of 2 * sort_cap 16-bit slots
// (pixel offset -> rank, verified against offs[]; 0xFFFF = empty; load factor <= 1/2).

__device__ __forceinline__ uint32_t slot_hash(uint32_t q, int hshift) { return (q * 2654435761u) >> hshift; }

__device__ __forceinline__ void slot_insert(uint32_t *slot32, uint32_t smask, int hshift, uint32_t q, uint32_t rank) {
    uint32_t hs = slot_hash(q, hshift);
    while (true) {
        const int sh16 = (hs & 1u) * 16;
        const uint32_t old = slot32[hs >> 1];
        if (((old >> sh16) & 0xFFFFu) == 0xFFFFu) {
            const uint32_t upd = (old & ~(0xFFFFu << sh16)) | (rank << sh16);
            if (atomicCAS(&slot32[hs >> 1], old, upd) == old) return;
            continue;   // the word changed under us (its other half, or this slot): look again
        }
        hs = (hs + 1u) & smask;
    }
}

// The better-ranked window entries (rank < i) within the distance of entry i: their count, and the ranks
// of the first four in list[] (padded with 0xFFFF).  Every candidate that can decide entry i's fate is in
// the table, so this list is all a later visit needs.
__device__ __forceinline__ int nms_collect(const uint32_t *offs, const uint32_t *slot32, uint32_t smask, int hshift,
                                           int w, int h, uint32_t i, int R, float min_dist_sq, uint32_t list[4]) {
    const uint32_t off = offs[i] & kOffMask;
    const int y = off / w, x = off - y * w;
    int cnt = 0;
    list[0] = list[1] = list[2] = list[3] = 0xFFFFu;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t q = (uint32_t)(yy * w + xx);
            uint32_t hs = slot_hash(q, hshift);
            while (true) {
                const uint32_t r = (slot32[hs >> 1] >> ((hs & 1u) * 16)) & 0xFFFFu;
                if (r == 0xFFFFu) break;
                if ((offs[r] & kOffMask) == q) {
                    if (r < i) {
                        if (cnt == 0) list[0] = r;
                        else if (cnt == 1) list[1] = r;
                        else if (cnt == 2) list[2] = r;
                        else if (cnt == 3) list[3] = r;
                        cnt++;
                    }
                    break;
                }
                hs = (hs + 1u) & smask;
            }
        }
    }
    return cnt;
}

// One visit of window entry i through the table: 1 accepted, 2 rejected, 0 still blocked by an undecided
// better-ranked neighbour.  Statuses are read as they are at this moment (other waves publish theirs
// without a barrier); they only ever go from undecided to decided, so a stale read costs a later visit.
__device__ __forceinline__ int nms_visit_lds(const volatile uint32_t *offs, const uint32_t *slot32, uint32_t smask,
                                             int hshift, int w, int h, uint32_t i, int R, float min_dist_sq) {
    const uint32_t off = offs[i] & kOffMask;
    const int y = off / w, x = off - y * w;
    bool blocked = false;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t q = (uint32_t)(yy * w + xx);
            uint32_t hs = slot_hash(q, hshift);
            while (true) {
                const uint32_t r = (slot32[hs >> 1] >> ((hs & 1u) * 16)) & 0xFFFFu;
                if (r == 0xFFFFu) break;
                const uint32_t e = offs[r];
                if ((e & kOffMask) == q) {
                    if (r < i) {
                        const uint32_t st = e >> 30;
                        if (st == 1u) return 2;
                        if (st == 0u) blocked = true;
                    }
                    break;
                }
                hs = (hs + 1u) & smask;
            }
        }
    }
    return blocked ? 0 : 1;
}

// One workgroup per frame.
//  fast path: only the first maxCorners ACCEPTED corners in rank order are wanted, and a
//    candidate's fate depends on higher-ranked candidates only, so suppression is run on the N
//    best-ranked candidates (N a little above maxCorners), sorted in LDS; if they yield fewer than
//    maxCorners survivors N is doubled (decisions already made stay valid).
//  slow path (N would exceed the LDS sort buffer): suppression over every candidate, then select.
template <int EPT>
__global__ __launch_bounds__(kST) __attribute__((amdgpu_waves_per_eu(8, 8))) void corner_select_kernel(
    const float *__restrict__ eig, int w, int h, uint8_t *__restrict__ state,
    unsigned long long *__restrict__ keys, const uint32_t *__restrict__ counts, size_t key_cap,
    int max_corners, float min_dist, float min_dist_sq, int sort_cap, float *__restrict__ out_xy,
    int32_t *__restrict__ out_n, int kp_stride, int32_t *__restrict__ overflow,
    const uint32_t *__restrict__ frame_max, double quality, int use_lists) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned long long *sortbuf = reinterpret_cast<unsigned long long *>(smem_raw);
    __shared__ SelectShared sh;

    const int f = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float *E = eig + (size_t)f * w * h;
    uint8_t *S = state + (size_t)f * w * h;
    unsigned long long *K = keys + (size_t)f * key_cap;
    float2 *O = reinterpret_cast<float2 *>(out_xy) + (size_t)f * kp_stride;
    uint32_t n = counts[f];
    if (n > key_cap) {
        if (tid == 0) atomicAdd(overflow, 1);
        n = (uint32_t)key_cap;
    }
    const int R = min_dist >= 1.f ? (int)ceilf(min_dist) : 0;
    const uint32_t want_max = (uint32_t)max_corners;

    // exact threshold: the fused detector prefilters with a running maximum, so only keys whose response is
    // > (float)(max * quality) count (every key of an exactly-thresholded list passes)
    float thr = (float)((double)ord2f(frame_max[f]) * quality);
    if (thr == 0.f) thr = 0.f;   // -0 -> +0 so the ordered-key compare equals the float compare
    const uint32_t t32 = f2ord(thr), m32 = frame_max[f];
    unsigned long long tkey = ((unsigned long long)t32 << 32) | 0xFFFFFFFFull;
    bool compacted = false;
    // drop the keys at or below the threshold from K (generic paths only; the two-pass window filters on the fly)
    auto compact_keys = [&]() {
        uint32_t kept = 0;
        for (uint32_t base = 0; base < n; base += kST) {
            const uint32_t i = base + tid;
            unsigned long long key = 0;
            bool keep = false;
            if (i < n) {
                key = K[i];
                keep = key > tkey;
            }
            const unsigned long long bal = __ballot(keep);
            __syncthreads();
            if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int wv = 0; wv < kST / 64; wv++) {
                const uint32_t c = sh.wave_cnt[wv];
                if (wv < wave) pre += c;
                tot += c;
            }
            if (keep) K[kept + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = key;
            kept += tot;
        }
        n = kept;
        tkey = 0ull;   // every remaining key passes (no key is 0: its response would be a NaN pattern)
        compacted = true;
        __syncthreads();
    };

    // ---------------------------------------------------------------- fast path
    uint32_t *offs = reinterpret_cast<uint32_t *>(sortbuf);
    uint32_t *slot32 = offs + sort_cap;
    const uint32_t smask = 2u * (uint32_t)sort_cap - 1u;
    const int hshift = 31 - (31 - __clz(sort_cap));   // 32 - log2(2 * sort_cap)
    bool done = false;
    {
        uint32_t N = want_max + want_max / 4 + 64;
        if (R == 0) N = want_max;
        while (true) {
            uint32_t got = N, n_kept = 0;
            if (!rank_window_2pass<EPT>(K, n, tkey, t32, m32, got, sortbuf, sort_cap, sh, n_kept, E, w)) {
                if (!compacted) compact_keys();
                n_kept = n;
                got = N < n ? N : n;
                if (got > (uint32_t)sort_cap) got = (uint32_t)sort_cap;
                const unsigned long long T = radix_select_nth(K, n, got, sh);
                got = gather_sorted<EPT>(K, n, T, sortbuf, sort_cap, sh);
            }
            if (R > 0) {
                // window -> offs[] (entries are read, then a barrier, then written: offs[i] overlays sortbuf[i / 2])
                for (uint32_t c0 = 0; c0 < got; c0 += kST) {
                    const uint32_t i = c0 + tid;
                    const uint32_t v = i < got ? (uint32_t)sortbuf[i] : 0u;
                    __syncthreads();
                    if (i < got) offs[i] = v;
                }
                __syncthreads();
                for (int i = tid; i < sort_cap; i += kST) slot32[i] = 0xFFFFFFFFu;
                __syncthreads();
                uint32_t pend = 0;   // bit k: window entry tid + k * kST is undecided (sort_cap <= 16 * kST)
                {
                    int k = 0;
                    for (uint32_t i = tid; i < got; i += kST, k++) {
                        slot_insert(slot32, smask, hshift, offs[i], i);
                        pend |= 1u << k;
                    }
                }
                __syncthreads();
                // first visit: who can decide my fate?  Nobody -> accepted; up to four -> remember their
                // ranks (when the launch provides the list region); more -> table visits every time.
                uint2 *nbr = use_lists ? reinterpret_cast<uint2 *>(smem_raw + sizeof(unsigned long long) * (size_t)sort_cap) : nullptr;
                uint32_t full = 0;
                {
                    int k = 0;
                    for (uint32_t i = tid; i < got; i += kST, k++) {
                        uint32_t list[4];
                        const int cnt = nms_collect(offs, slot32, smask, hshift, w, h, i, R, min_dist_sq, list);
                        if (cnt == 0) {
                            offs[i] |= 1u << 30;
                            pend &= ~(1u << k);
                        } else if (nbr && cnt <= 4) {
                            nbr[i] = make_uint2(list[0] | (list[1] << 16), list[2] | (list[3] << 16));
                        } else {
                            full |= 1u << k;
                        }
                    }
                }
                __syncthreads();
                // Suppression fixpoint.  An entry's fate needs its better-ranked neighbours decided first, and chains
                // of such dependencies run along image edges, so the number of rounds is the longest chain: each
                // wave therefore polls on its own, without workgroup barriers (statuses live in LDS, every wave of the
                // workgroup is resident, and the rank order makes the dependency graph acyclic).
                {
                    volatile uint32_t *voffs = offs;
                    int spins = 0;
                    while (__any(pend != 0)) {
                        int k = 0;
                        for (uint32_t i = tid; i < got; i += kST, k++) {
                            if (!((pend >> k) & 1u)) continue;
                            int d;
                            if ((full >> k) & 1u) {
                                d = nms_visit_lds(voffs, slot32, smask, hshift, w, h, i, R, min_dist_sq);
                            } else {
                                const uint2 l = nbr[i];
                                const uint32_t r[4] = {l.x & 0xFFFFu, l.x >> 16, l.y & 0xFFFFu, l.y >> 16};
                                bool blocked = false, rejected = false;
#pragma unroll
                                for (int t = 0; t < 4; t++) {
                                    if (r[t] == 0xFFFFu) continue;
                                    const uint32_t st = voffs[r[t]] >> 30;
                                    rejected |= st == 1u;
                                    blocked |= st == 0u;
                                }
                                d = rejected ? 2 : (blocked ? 0 : 1);
                            }
                            if (d) {
                                voffs[i] = voffs[i] | ((uint32_t)d << 30);
                                pend &= ~(1u << k);
                            }
                        }
                        if (++spins > (1 << 22)) {   // cannot happen (acyclic); never hang the device on a defect
                            if (lane == 0) atomicAdd(overflow, 1);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
            }
            // survivors in rank order: ordered scan over the window
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t i0 = 0; i0 < got; i0 += kST) {
                const uint32_t i = i0 + tid;
                uint32_t off = 0;
                bool acc = false;
                if (i < got) {
                    if (R == 0) {
                        off = (uint32_t)sortbuf[i];
                        acc = true;
                    } else {
                        const uint32_t e = offs[i];
                        off = e & kOffMask;
                        acc = (e >> 30) == 1u;
                    }
                }
                const unsigned long long bal = __ballot(acc);
                __syncthreads();
                if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t pre = 0, tot = 0;
                for (int wv = 0; wv < kST / 64; wv++) {
                    const uint32_t c = sh.wave_cnt[wv];
                    if (wv < wave) pre += c;
                    tot += c;
                }
                const uint32_t pos = base + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                // written speculatively: if this attempt falls short the next one rewrites the
                // same prefix with the same values (rank order does not change)
                if (acc && pos < want_max) {
                    const int y = off / w, x = off - y * w;
                    O[pos] = make_float2((float)x, (float)y);
                }
                base += tot;
            }
            if (base >= want_max || got == n_kept) {
                if (tid == 0) out_n[f] = (int32_t)(base < want_max ? base : want_max);
                done = true;
                break;
            }
            if (got == (uint32_t)sort_cap) break;   // cannot widen in LDS: slow path
            N *= 2;
        }
    }
    if (done) return;

    // ---------------------------------------------------------------- slow path (rare)
    // suppression over every candidate through a per-pixel state map in global memory
    // (0 none, 1 undecided, 2 accepted, 3 rejected), which this path initialises itself
    if (!compacted) compact_keys();
    if (R > 0) {
        for (uint32_t i = tid; i < (uint32_t)(w * h); i += kST) S[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kST) S[(uint32_t)K[i]] = 1;
    }
    if (R > 0) {
        while (true) {
            __syncthreads();
            if (tid == 0) sh.flag = 0;
            __syncthreads();
            bool pending = false;
            for (uint32_t i = tid; i < n; i += kST) {
                const uint32_t off = (uint32_t)K[i];
                if (S[off] != 1) continue;
                const int d = nms_visit(E, S, w, h, off, R, min_dist_sq);
                if (d == 1) pending = true;
                else S[off] = (uint8_t)d;
            }
            if (pending) sh.flag = 1;
            __syncthreads();
            if (!sh.flag) break;
        }
    }
    __syncthreads();
    // compact accepted keys to the front of K (entries are read before any write of the same round)
    uint32_t n_acc = 0;
    for (uint32_t base = 0; base < n; base += kST) {
        const uint32_t i = base + tid;
        unsigned long long key = 0;
        bool acc = false;
        if (i < n) {
            key = K[i];
            acc = (R == 0) || S[(uint32_t)key] == 2;
        }
        const unsigned long long bal = __ballot(acc);
        __syncthreads();
        if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int wv = 0; wv < kST / 64; wv++) {
            const uint32_t c = sh.wave_cnt[wv];
            if (wv < wave) pre += c;
            tot += c;
        }
        if (acc) K[n_acc + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = key;
        n_acc += tot;
    }
    __syncthreads();
    uint32_t want = n_acc < want_max ? n_acc : want_max;
    if (want > (uint32_t)sort_cap) want = (uint32_t)sort_cap;
    const unsigned long long T = radix_select_nth(K, n_acc, want, sh);
    gather_sorted<EPT>(K, n_acc, T, sortbuf, sort_cap, sh);
    for (uint32_t i = tid; i < want; i += kST) {
        const uint32_t off = (uint32_t)sortbuf[i];
        const int y = off / w, x = off - y * w;
        O[i] = make_float2((float)x, (float)y);
    }
    if (tid == 0) out_n[f] = (int32_t)want;
}

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
    uint32_t rp[7][4];    // horizontally filtered rows (Q8, < 2^16)
    uint32_t raw[7][3];   // prefetched gray dwords x-4, x, x+4 of the next seven rows
};
struct BlurArgs {
    const uint8_t *src;
    uint8_t *dst;
    int w, h, ys, steps, x;
    uint32_t voff_l, voff_c, voff_r;
    bool edge, left_fix, right_fix, own_lane;
};

template <int K>   // K = t % 7: ring slot of both the filtered row and the prefetched gray row
__device__ __forceinline__ void blur_step(BlurState &st, const BlurArgs &a, int t) {
    constexpr uint32_t W0 = 18u | (34u << 8) | (48u << 16) | (56u << 24);   // taps 0..3
    constexpr uint32_t W1 = 48u | (34u << 8) | (18u << 16);                 // taps 4..6
    uint32_t d0 = st.raw[K][0];
    const uint32_t d1 = st.raw[K][1];
    uint32_t d2 = st.raw[K][2];
    if (t + 7 < a.steps) {   // the row seven steps ahead goes into the slot just consumed
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + t + 7, a.h) * a.w;
        st.raw[K][0] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_l);
        st.raw[K][1] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
        st.raw[K][2] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_r);
    }
    if (a.edge) {   // BORDER_REFLECT_101: columns -3..-1 are 3..1, columns w..w+2 are w-2..w-4
        if (a.left_fix) d0 = __builtin_amdgcn_perm(d1, d1, 0x01020300u);
        if (a.right_fix) d2 = __builtin_amdgcn_perm(d1, d1, 0x00000102u);
    }
    st.rp[K][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), W0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), W1, 0u, false), false);
    st.rp[K][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), W0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), W1, 0u, false), false);
    st.rp[K][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), W0,
                                         __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), W1, 0u, false), false);
    st.rp[K][3] = __builtin_amdgcn_udot4(d1, W0, __builtin_amdgcn_udot4(d2, W1, 0u, false), false);
    if (t < 6) return;
    const int y = a.ys - 6 + t;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t sum = 18u * (st.rp[(K + 1) % 7][i] + st.rp[K][i]) + 34u * (st.rp[(K + 2) % 7][i] + st.rp[(K + 6) % 7][i]) +
                             48u * (st.rp[(K + 3) % 7][i] + st.rp[(K + 5) % 7][i]) + 56u * st.rp[(K + 4) % 7][i];
        packed |= ((sum + (1u << 15)) >> 16) << (8 * i);
    }
    if (a.own_lane) *reinterpret_cast<uint32_t *>(a.dst + (size_t)y * a.w + a.x) = packed;
}

__global__ __launch_bounds__(256) void gaussian7_stream_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                               uint8_t *__restrict__ out, int seg_rows, int frames,
                                                               int strips, int per_frame) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    int f, blk;   // a frame's strips and segments share an XCD (see min_eigen_stream_kernel)
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    const int strip = blk % strips, segblk = blk / strips;
    BlurArgs a;
    a.ys = (segblk * 4 + wave) * seg_rows;
    if (a.ys >= h) return;   // whole wave; no barriers in this kernel
    const int ye = a.ys + seg_rows < h ? a.ys + seg_rows : h;
    a.steps = ye - a.ys + 6;
    a.w = w;
    a.h = h;
    a.src = gray + (size_t)f * w * h;
    a.dst = out + (size_t)f * w * h;
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

// ------------------------------------------------------------------------------------------
// ORB::compute for provided keypoints: border filter (ordered) + steered BRIEF
// ------------------------------------------------------------------------------------------
constexpr int kKT = 256;

// KeyPointsFilter::runByImageBorder(kps, size, 31): keep pt inside [31, w-31) x [31, h-31), order kept
__global__ __launch_bounds__(kKT) void keypoint_border_kernel(const float *__restrict__ xy_in,
                                                              const int32_t *__restrict__ n_in, int kp_stride,
                                                              int w, int h, float *__restrict__ xy_out,
                                                              int32_t *__restrict__ n_out) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_cnt[kKT / 64];
    __shared__ int s_base;
    const float2 *I = reinterpret_cast<const float2 *>(xy_in) + (size_t)f * kp_stride;
    float2 *O = reinterpret_cast<float2 *>(xy_out) + (size_t)f * kp_stride;
    const int n = n_in[f];
    const int border = 31;
    const bool any = !(h <= border * 2 || w <= border * 2);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kKT) {
        const int i = i0 + tid;
        float2 p = make_float2(0.f, 0.f);
        bool keep = false;
        if (i < n) {
            p = I[i];
            keep = any && p.x >= (float)border && p.x < (float)(w - border) && p.y >= (float)border &&
                   p.y < (float)(h - border);
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_cnt[wave] = (int)__popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int wv = 0; wv < wave; wv++) off += s_cnt[wv];
        if (keep) O[off + (int)__popcll(bal & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int wv = 0; wv < kKT / 64; wv++) t += s_cnt[wv];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) n_out[f] = s_base;
}

// One lane per (keypoint, descriptor byte): 8 tests = 16 gathers from the blurred image.
// The 512 rotated sample offsets are the same for every keypoint (one global angle), so each
// workgroup rotates the pattern once into LDS:  x = px*a - py*b, y = px*b + py*a, cvRound.
__global__ __launch_bounds__(kKT) void rbrief_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                     const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
                                                     int kp_stride, float ca, float sa,
                                                     const int8_t *__restrict__ pattern,
                                                     uint8_t *__restrict__ desc, int frames, int per_frame) {
    __shared__ int s_off[512];
    const int tid = threadIdx.x;
    int f, bx;
    vs_xcd_item_block(blockIdx.x, per_frame, f, bx);   // a frame's patches are gathered through one XCD's L2
    if (f >= frames) return;
    const int n = n_arr[f];
    const int kp0 = bx * (kKT / 32);
    if (kp0 >= n) return;
    for (int i = tid; i < 512; i += kKT) {
        const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
        const float a1 = px * ca, a2 = py * sa, b1 = px * sa, b2 = py * ca;
        const float rx = a1 - a2, ry = b1 + b2;
        const int ix = (int)rintf(rx), iy = (int)rintf(ry);
        s_off[i] = iy * w + ix;
    }
    __syncthreads();
    const int kp = kp0 + (tid >> 5), byte = tid & 31;
    if (kp >= n) return;
    const float2 p = reinterpret_cast<const float2 *>(xy)[(size_t)f * kp_stride + kp];
    const int cx = (int)rintf(p.x), cy = (int)rintf(p.y);
    const uint8_t *center = blurred + (size_t)f * w * h + (size_t)cy * w + cx;
    uint32_t val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int t0 = center[s_off[(byte * 8 + bit) * 2]];
        const int t1 = center[s_off[(byte * 8 + bit) * 2 + 1]];
        val |= (uint32_t)(t0 < t1) << bit;
    }
    desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + byte] = (uint8_t)val;
}

// The same descriptor with the keypoint's patch staged through LDS (width % 4 == 0).
// The direct form is a chain of dependent round trips per workgroup of 8 keypoints (pattern, keypoint, 16
// scattered byte loads per lane, each 64 separate addresses for the texture addresser).  Here
//  * the pattern is rotated once per launch (rbrief_rotate_kernel) instead of once per workgroup,
//  * a workgroup walks through 64 keypoints, 8 at a time; the 32 lanes of a keypoint copy its patch rows
//    (2R+1 rows of up to 36 bytes, R = the largest rotated offset; dword loads, one or two cache lines per row) into
//    LDS and take the 16 samples per lane from there,
//  * the patch loads of the next 8 keypoints are in flight while the current 8 are sampled (two LDS buffers).
// Sample offsets are kept transposed ([sample][byte]) so the 32 lanes of a keypoint read 32 consecutive words.
constexpr int kRPitch = 36;     // bytes per staged patch row (9 dwords: 2 * 16 + 1 columns plus alignment slack)
constexpr int kRMax = 16;       // largest |offset| staged: covers the 31x31 pattern at the angles cv::KeyPoint's default
                                // (-1 degree) and small rotations give; wider rotated patterns are sampled directly.
                                // Keeping the patch small is what lets 7 workgroups share a CU's LDS.
constexpr int kRRows = 2 * kRMax + 1;
constexpr int kRGroups = 8;     // groups of 8 keypoints per workgroup
constexpr int kRLoads = (kRRows * (kRPitch / 4) + 31) / 32;   // patch dwords per lane, worst case

// table[i] = (iy << 16) | (ix & 0xFFFF) for sample i (test i / 2, side i & 1), table[512] = max(|ix|, |iy|)
__global__ __launch_bounds__(512) void rbrief_rotate_kernel(const int8_t *__restrict__ pattern, float ca, float sa,
                                                            int32_t *__restrict__ table) {
    __shared__ int s_R;
    const int i = threadIdx.x;
    if (i == 0) s_R = 0;
    __syncthreads();
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float a1 = px * ca, a2 = py * sa, b1 = px * sa, b2 = py * ca;
    const float rx = a1 - a2, ry = b1 + b2;
    const int ix = (int)rintf(rx), iy = (int)rintf(ry);
    table[i] = (int32_t)(((uint32_t)iy << 16) | ((uint32_t)ix & 0xFFFFu));
    atomicMax(&s_R, max(abs(ix), abs(iy)));
    __syncthreads();
    if (i == 0) table[512] = s_R;
}

__global__ __launch_bounds__(kKT) void rbrief_lds_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                         const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
                                                         int kp_stride, const int32_t *__restrict__ table,
                                                         uint8_t *__restrict__ desc, int frames, int per_frame) {
    __shared__ int s_off[16][32];   // [2 * bit + side][byte]: offset relative to the centre, in the staged patch (or the image)
    __shared__ uint32_t s_patch[2][kKT / 32][kRRows * (kRPitch / 4)];
    const int tid = threadIdx.x;
    int f, bx;
    vs_xcd_item_block(blockIdx.x, per_frame, f, bx);
    if (f >= frames) return;
    const int n = n_arr[f];
    const int kp0 = bx * (kRGroups * (kKT / 32));
    if (kp0 >= n) return;   // whole workgroup
    const int R = table[512];
    const bool staged = R <= kRMax;
    for (int i = tid; i < 512; i += kKT) {
        const int32_t e = table[i];
        const int ix = (int)(int16_t)(e & 0xFFFF), iy = e >> 16;
        const int t = i >> 1, byte = t >> 3, bit = t & 7;
        s_off[2 * bit + (i & 1)][byte] = staged ? iy * kRPitch + ix : iy * w + ix;
    }
    const int slot = tid >> 5, l32 = tid & 31;
    const uint8_t *img = blurred + (size_t)f * w * h;
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)f * kp_stride;
    auto keypoint = [&](int g) -> float2 {   // this slot's keypoint of group g (its 32 lanes read the same one)
        const int kp = kp0 + g * (kKT / 32) + slot;
        return (g < kRGroups && kp < n) ? P[kp] : make_float2(-1.f, -1.f);   // x < 0: no keypoint
    };
    // dwords per staged row: columns (cx - R) & ~3 .. cx + R, at most 2R + 4 bytes; rows keep the fixed pitch
    const int nd = staged ? (2 * R + 3) / 4 + 1 : 1;
    const int total = (2 * R + 1) * nd;
    const int r0 = l32 / nd, c0 = l32 - r0 * nd, dq = 32 / nd, dr = 32 - dq * nd;   // dword l32 + 32k = row r, column c
    uint32_t pre[kRLoads];
    auto prefetch = [&](int cx, int cy, bool live) {
        if (!staged || !live) return;
        const uint8_t *src = img + (size_t)(cy - R) * w + ((cx - R) & ~3);
        int r = r0, c = c0;
#pragma unroll
        for (int k = 0; k < kRLoads; k++) {
            if (l32 + 32 * k < total) pre[k] = *reinterpret_cast<const uint32_t *>(src + (size_t)r * w + 4 * c);
            r += dq;
            c += dr;
            if (c >= nd) {
                c -= nd;
                r++;
            }
        }
    };
    // keypoints are read two groups ahead and patches one group ahead, so no load waits on another inside the loop
    float2 p0 = keypoint(0), p1 = keypoint(1);
    int cx = (int)rintf(p0.x), cy = (int)rintf(p0.y);
    bool live = p0.x >= 0.f;
    prefetch(cx, cy, live);
#pragma unroll 1   // one copy of the body: the fully unrolled form needs twice the registers and halves the occupancy
    for (int g = 0; g < kRGroups; g++) {
        if (staged && live) {
            int r = r0, c = c0;
#pragma unroll
            for (int k = 0; k < kRLoads; k++) {
                if (l32 + 32 * k < total) s_patch[g & 1][slot][r * (kRPitch / 4) + c] = pre[k];
                r += dq;
                c += dr;
                if (c >= nd) {
                    c -= nd;
                    r++;
                }
            }
        }
        __syncthreads();   // also orders the first s_off reads after their writes
        const float2 p2 = keypoint(g + 2);
        const int cx_n = (int)rintf(p1.x), cy_n = (int)rintf(p1.y);
        const bool live_n = p1.x >= 0.f;
        prefetch(cx_n, cy_n, live_n);
        if (live) {
            const int kp = kp0 + g * (kKT / 32) + slot;
            const uint8_t *pb = staged ? reinterpret_cast<const uint8_t *>(s_patch[g & 1][slot]) + R * kRPitch + (cx - ((cx - R) & ~3))
                                       : img + (size_t)cy * w + cx;
            uint32_t val = 0;
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const int t0 = pb[s_off[2 * bit][l32]];
                const int t1 = pb[s_off[2 * bit + 1][l32]];
                val |= (uint32_t)(t0 < t1) << bit;
            }
            desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + l32] = (uint8_t)val;
        }
        cx = cx_n;
        cy = cy_n;
        live = live_n;
        p1 = p2;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int vs_launch_bgr2gray(vslam_ctx *ctx, const uint8_t *bgr, int frames, int w, int h, int stride,
                       uint8_t *gray) {
    VS_REQUIRE(ctx, bgr && gray, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w > 0 && h > 0 && stride >= 3 * w, VSLAM_ERR_INVALID);
    const int aligned = (stride % 4 == 0) && (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(bgr) & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(gray) & 3) == 0) && (((size_t)h * stride) % 4 == 0);
    VsProfScope ps(ctx, "bgr2gray_kernel");
    dim3 grid(vs_div_up(((w + 3) / 4) * h, 256), frames);
    bgr2gray_kernel<<<grid, 256, 0, ctx->stream>>>(bgr, w, h, stride, gray, aligned);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_min_eigen(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, float *eig,
                        uint32_t *frame_max_bits) {
    VS_REQUIRE(ctx, gray && eig, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 3 && h >= 3, VSLAM_ERR_INVALID);
    if (frame_max_bits) VS_HIP(ctx, hipMemsetAsync(frame_max_bits, 0, sizeof(uint32_t) * (size_t)frames, ctx->stream));
    VsProfScope ps(ctx, "min_eigen_kernel");
    if (w % 4 == 0 && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0) && ((reinterpret_cast<uintptr_t>(eig) & 15) == 0)) {
        dim3 grid(vs_div_up(w, kE4W), vs_div_up(h, kE4H), frames);
        min_eigen_v4_kernel<<<grid, 256, 0, ctx->stream>>>(gray, w, h, eig, frame_max_bits);
    } else {
        dim3 grid(vs_div_up(w, kETW), vs_div_up(h, kETH), frames);
        min_eigen_kernel<<<grid, kET, 0, ctx->stream>>>(gray, w, h, eig, frame_max_bits);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_good_features(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h,
                            int max_corners, double quality, double min_distance, int kp_stride,
                            float *xy, int32_t *n) {
    VS_REQUIRE(ctx, gray && xy && n, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 3 && h >= 3, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, max_corners > 0 && max_corners <= kp_stride, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, min_distance < 64.0, VSLAM_ERR_CAPACITY);
    VS_REQUIRE(ctx, (size_t)w * h < (1u << 30), VSLAM_ERR_CAPACITY);   // pixel offsets carry 2 status bits in the selection
    const size_t px = (size_t)w * h;
    float *eig = nullptr;
    uint32_t *fmax = nullptr, *counts = nullptr;
    uint8_t *state = nullptr;
    unsigned long long *keys = nullptr;
    int32_t *overflow = nullptr;
    int rc;
    if ((rc = vs_arena_get(ctx, "gf.eig", sizeof(float) * px * frames, (void **)&eig))) return rc;
    if ((rc = vs_arena_get(ctx, "gf.fmax", sizeof(uint32_t) * (size_t)frames, (void **)&fmax))) return rc;
    if ((rc = vs_arena_get(ctx, "gf.counts", sizeof(uint32_t) * (size_t)frames + sizeof(int32_t), (void **)&counts))) return rc;
    if ((rc = vs_arena_get(ctx, "gf.state", px * frames, (void **)&state))) return rc;
    // every interior pixel can be a candidate (a plateau equals its own dilation), so the key list
    // is sized for the whole image: exactness over memory
    const size_t key_cap = px;
    if ((rc = vs_arena_get(ctx, "gf.keys", sizeof(unsigned long long) * key_cap * frames, (void **)&keys))) return rc;
    overflow = reinterpret_cast<int32_t *>(counts + frames);

    const bool fused = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
    VS_HIP(ctx, hipMemsetAsync(counts, 0, sizeof(uint32_t) * (size_t)frames + sizeof(int32_t), ctx->stream));
    if (fused) {
        VS_HIP(ctx, hipMemsetAsync(fmax, 0, sizeof(uint32_t) * (size_t)frames, ctx->stream));
        {
            VsProfScope ps(ctx, "min_eigen_kernel");
            const int segs = h >= 135 ? (h + 45) / 90 : 1;   // about 90 rows per wave
            const int seg_rows = vs_div_up(h, segs);
            const int strips = vs_div_up(w, kSW), per_frame = strips * vs_div_up(segs, 4);
            min_eigen_stream_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(
                gray, w, h, eig, fmax, quality, keys, counts, key_cap, seg_rows, frames, strips, per_frame);
        }
        // A frame whose maximum response is negative has no corners (its threshold max * quality lies above every
        // response, THRESH_TOZERO clears the image and zeros are not corners); corner_select_kernel's exact
        // threshold drops every key of such a frame, so it needs no special handling here.
    } else {
        if ((rc = vs_launch_min_eigen(ctx, gray, frames, w, h, eig, fmax))) return rc;
        VsProfScope ps(ctx, "corner_candidates_kernel");
        dim3 grid(vs_div_up(w, kCTW), vs_div_up(h, kCTH), frames);
        corner_candidates_kernel<<<grid, kCT, 0, ctx->stream>>>(eig, w, h, fmax, quality, keys, counts, key_cap);
    }
    if (ctx->fork_after_eigen) {   // the caller runs an independent stage on the auxiliary stream beside the selection
        VS_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        VS_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    }
    {
        int sort_cap = 2;
        while (sort_cap < 2 * max_corners && sort_cap < 16384) sort_cap <<= 1;
        while (sort_cap < max_corners) sort_cap <<= 1;   // at least max_corners slots
        size_t lds = sizeof(unsigned long long) * (size_t)sort_cap;
        VS_REQUIRE(ctx, lds <= 128 * 1024, VSLAM_ERR_CAPACITY);
        // second region of the same size: per window entry the ranks of up to four candidates that can suppress it
        const int use_lists = 2 * lds <= 128 * 1024;
        if (use_lists) lds *= 2;
        const float md = (float)min_distance;
        const float md2 = (float)(min_distance * min_distance);   // `minDistance *= minDistance` in double, compared as float
        VsProfScope ps(ctx, "corner_select_kernel");
#define VS_SELECT_LAUNCH(EPT)                                                                                              \
    do {                                                                                                                   \
        if (!ctx->attr_done["corner_select" #EPT]) {                                                                       \
            VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(corner_select_kernel<EPT>),                     \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));                     \
            ctx->attr_done["corner_select" #EPT] = true;                                                                   \
        }                                                                                                                  \
        corner_select_kernel<EPT><<<frames, kST, lds, ctx->stream>>>(eig, w, h, state, keys, counts, key_cap, max_corners, \
                                                                     md, md2, sort_cap, xy, n, kp_stride, overflow, fmax,  \
                                                                     quality, use_lists);                                  \
    } while (0)
        switch (sort_cap / kST) {
            case 1: VS_SELECT_LAUNCH(1); break;
            case 2: VS_SELECT_LAUNCH(2); break;
            case 4: VS_SELECT_LAUNCH(4); break;
            case 8: VS_SELECT_LAUNCH(8); break;
            case 16: VS_SELECT_LAUNCH(16); break;
            default: VS_SELECT_LAUNCH(0); break;   // sort_cap < kST
        }
#undef VS_SELECT_LAUNCH
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_gaussian7(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, uint8_t *out) {
    VS_REQUIRE(ctx, gray && out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 4 && h >= 4, VSLAM_ERR_INVALID);
    VsProfScope ps(ctx, "gaussian7_kernel");
    if (w % 4 == 0 && ((reinterpret_cast<uintptr_t>(gray) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
        const int segs = h >= 135 ? (h + 45) / 90 : 1;   // about 90 rows per wave
        const int seg_rows = vs_div_up(h, segs);
        const int strips = vs_div_up(w, 256), per_frame = strips * vs_div_up(segs, 4);
        gaussian7_stream_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(gray, w, h, out, seg_rows, frames,
                                                                                          strips, per_frame);
    } else {
        dim3 grid(vs_div_up(w, kBTW), vs_div_up(h, kBTH), frames);
        gaussian7_kernel<<<grid, kBT, 0, ctx->stream>>>(gray, w, h, out);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_orb_describe(vslam_ctx *ctx, const uint8_t *blurred, int frames, int w, int h,
                           const float *xy_in, const int32_t *n_in, int kp_stride, float ca, float sa,
                           const int8_t *pattern, float *xy_out, uint8_t *desc, int32_t *n_out) {
    VS_REQUIRE(ctx, blurred && xy_in && n_in && pattern && xy_out && desc && n_out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    {
        VsProfScope ps(ctx, "keypoint_border_kernel");
        keypoint_border_kernel<<<frames, kKT, 0, ctx->stream>>>(xy_in, n_in, kp_stride, w, h, xy_out, n_out);
    }
    {
        VsProfScope ps(ctx, "rbrief_kernel");
        const int per_frame = vs_div_up(kp_stride, kKT / 32);
        if (w % 4 == 0 && (reinterpret_cast<uintptr_t>(blurred) & 3) == 0) {
            int32_t *table = nullptr;
            int rc = vs_arena_get(ctx, "rbrief.table", sizeof(int32_t) * 513, (void **)&table);
            if (rc) return rc;
            rbrief_rotate_kernel<<<1, 512, 0, ctx->stream>>>(pattern, ca, sa, table);
            const int per_frame_lds = vs_div_up(kp_stride, kRGroups * (kKT / 32));
            rbrief_lds_kernel<<<vs_xcd_grid(frames, per_frame_lds), kKT, 0, ctx->stream>>>(
                blurred, w, h, xy_out, n_out, kp_stride, table, desc, frames, per_frame_lds);
        } else {
            rbrief_kernel<<<vs_xcd_grid(frames, per_frame), kKT, 0, ctx->stream>>>(blurred, w, h, xy_out, n_out, kp_stride, ca,
                                                                                    sa, pattern, desc, frames, per_frame);
        }
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// Shi-Tomasi corner response (cv::cornerMinEigenVal inside cv::goodFeaturesToTrack, reference: src/Frame.cpp:61)
// and the 3x3-local-maximum candidates for gfx950.
//   min_eigen_tiered_kernel   the front-end path, tier 1: a certified bound of the response for every pixel + the list of pixels
//                             that can matter, one streaming pass (BGR form: cvtColor inside; gray form: rows of a multiple of 4
//                             bytes, i.e. width % 4 == 0 or padded rows with a mirrored tail, vslam_ctx::img_pitch)
//   corner_exact_kernel       tier 2: the oracle's arithmetic for the listed pixels
//   min_eigen_v4_kernel       response only, tiled (vslam_min_eigen, the pool's rerun; width % 4 == 0)
//   min_eigen_kernel          response only, any width and row pitch
//   corner_candidates_kernel  threshold + 3x3 maxima over a stored response map (any width)
// The arithmetic follows the oracle (oracle/vso_extract.cpp) operation for operation; float steps are written so
// that no contraction or reassociation can occur (-ffp-contract=off).  What bounds the response kernels is VALU
// issue (exact f64 box sums, correctly rounded sqrt), not bandwidth.
#include "image_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// cornerMinEigenVal(gray, eig, 3, 3) + per-frame max
// ------------------------------------------------------------------------------------------
constexpr int kET = 256;          // threads
constexpr int kETW = 64, kETH = 16;   // output tile
constexpr int kGW = kETW + 4, kGH = kETH + 4;   // gray tile (halo 2)
constexpr int kHW = kETW + 2;                   // hx / R / cov width (halo 1)

// `pitch` = bytes per gray row (eig rows are w floats).  With pool.count set, blockIdx.z is a slot of the corner pipeline's
// fallback pool, as in min_eigen_v4_kernel below.
__global__ __launch_bounds__(kET) void min_eigen_kernel(const uint8_t *__restrict__ gray, int w, int h, int pitch,
                                                        float *__restrict__ eig,
                                                        uint32_t *__restrict__ frame_max, const VsCornerPool pool) {
    __shared__ uint8_t G[kGH][kGW];
    __shared__ float HX[kGH][kHW], RR[kGH][kHW];
    __shared__ float CXX[kETH + 2][kHW], CXY[kETH + 2][kHW], CYY[kETH + 2][kHW];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x;
    int f = blockIdx.z, fin = f;   // f: where the outputs go; fin: the frame read
    if (pool.count) {
        if (f >= vs_pool_used(pool)) return;
        fin = pool.frame[f];
    }
    const int x0 = blockIdx.x * kETW, y0 = blockIdx.y * kETH;
    const uint8_t *src = gray + (size_t)fin * pitch * h;
    if (tid == 0) s_max = 0;

    // gray tile at raw coordinates [x0-2, x0+TW+2) x [y0-2, y0+TH+2), REFLECT_101 filled
    for (int i = tid; i < kGH * kGW; i += kET) {
        const int r = i / kGW, c = i - r * kGW;
        G[r][c] = src[(size_t)reflect101(y0 - 2 + r, h) * pitch + reflect101(x0 - 2 + c, w)];
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

// Plain cornerMinEigenVal for vslam_min_eigen (the front-end path uses min_eigen_stream_kernel below, which
// follows this kernel's arithmetic): a 256x32 tile per workgroup, every pixel of it owned.
// With pool.count set, blockIdx.z is a slot of the corner pipeline's fallback pool (select.hip): input = the frame filed
// under the slot, outputs (eig, frame_max) by slot; slots nobody claimed return at once.
__global__ __launch_bounds__(256) void min_eigen_v4_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                           float *__restrict__ eig,
                                                           uint32_t *__restrict__ frame_max, const VsCornerPool pool) {
    __shared__ uint32_t G[kE4H + 4][kE4C];   // bytes x0-4 .. x0+259 of raw rows y0-2 .. y0+33
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x;
    int f = blockIdx.z, fin = f;   // f: where the outputs go; fin: the frame read
    if (pool.count) {
        if (f >= vs_pool_used(pool)) return;
        fin = pool.frame[f];
    }
    const int x0 = blockIdx.x * kE4W, y0 = blockIdx.y * kE4H;
    const uint8_t *src = gray + (size_t)fin * w * h;
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
// Two-tier streaming form (the front-end path): certified cheap responses for every pixel, the oracle's exact
// arithmetic only where a value can be observed
// ------------------------------------------------------------------------------------------
// What later stages can observe of cornerMinEigenVal is (a) the frame maximum, (b) which interior pixels are above
// max * quality and not below any 3x3 neighbour, and (c) the responses of exactly those pixels (they rank the
// corners).  About 2 % of the pixels.  min_eigen_stream_kernel nevertheless pays the exact sequence — f64 box sums,
// correctly rounded sqrt — for every pixel, and is bound by those instructions.  Here every pixel first gets
//     U~ = (A + C) - sqrt((A - C)^2 + 4 B^2),   A = sum dxi^2, B = sum dxi dyi, C = sum dyi^2 over the 3x3 window,
// where dxi, dyi are the INTEGER Sobel responses of the 8-bit image.  All of it up to A, B, C is integer arithmetic on
// values below 2^24, carried in f32 registers (v_add / v_sub / v_mul_f32 are the cheapest vector instructions of this
// chip and exact on such values), so A, B, C are the exact real quantities that the reference's float pipeline
// approximates; only tr = A + C (< 2^25), the square root and the last subtraction round.
//
// Certified distance to the reference value e (float, scaled by c0 = 0.5 / (4*3*255)^2):  |e / c0 - U~| <= m(tr),
//     m(tr) = 0.016 sqrt(tr) + 32 u tr + 1e-3,     u = 2^-24.
// Derivation (s = 1 / 3060; "int units" = multiples of s resp. s^2):
//   * Dx = fl(fl(hx0 k0) + fl((hxm + hxp) k1)), k1 = fl(s), k0 = 2 k1: each product carries a relative error u on a
//     term of up to 510 s, the sum may cancel, so |Dx / s - dxi| <= u (2 |dxi| + 1020) <= 3060 u.  Dy = fl(R2 - R0)
//     with R = fl(fl(g0 k0) + fl((gm + gp) k1)) <= 1020 s carrying three roundings each:
//     |Dy / s - dyi| <= 3.01 u * 2040 + 1.01 u |dyi| <= 7200 u =: ed = 4.3e-4 (absolute, whatever the derivative is —
//     this is the "rounding residue" of a derivative that is zero in exact arithmetic).
//   * products: |c / s^2 - di dj| <= ed (|di| + |dj|) + ed^2 + u (|di| + ed)(|dj| + ed); nine of them, summed exactly
//     enough in f64 (2^-53), rounded once to float: with Cauchy-Schwarz (sum |di| <= 3 sqrt(A)),
//     |Sxx/s^2 - A| <= 6 ed sqrt(A) + 2.1 u A + 1e-5, likewise C, and |Sxy/s^2 - B| <= 3 ed (sqrt A + sqrt C) + 2.1 u sqrt(AC) + u|B| + 1e-5.
//   * U is 2-Lipschitz in each of A, C, B; calcMinEigenVal's own float steps add <= 5 u tr; this kernel's tr, square
//     root (v_sqrt_f32, 1 ulp) and subtraction add <= 5 u tr.  Together
//     |e / c0 - U~| <= 25.5 ed sqrt(tr) + 19.3 u tr + 6e-5 = 0.011 sqrt(tr) + 19.3 u tr + 6e-5  <  m(tr).
//   tests/test_min_eigen_bound.py measures the distance on synthetic and adversarial images (noise at 0/255, bright
//   low-contrast noise, checkerboards, ramps): it stays below 4 % of m.
//
// A pixel p is POSSIBLE if U~p + 2 mw >= every neighbour's U~ and >= the running threshold bound, where mw is m() of
// the largest tr in the lane's 6 x 3 neighbourhood (so it bounds the margin of p and of every neighbour, plus the
// roundings of these comparisons).  Possible pixels (a superset of the candidates and of the frame maximum) queue up
// per wave in LDS and are evaluated 64 at a time with the oracle's exact sequence — that value is what is stored in the
// key.  If U~p - 2 mw >= every neighbour's U~ the pixel is certainly a 3x3 maximum; otherwise (0.7 % of the possible
// ones on image data: near-ties, and the candidates on a strip's first / last column, whose outer neighbours the wave
// does not have) the needed neighbours are evaluated exactly as well, eight candidates x eight neighbours per round,
// and compared as the reference compares them.  The first / last row and column are not candidate positions but count
// for the maximum: they queue when U~ + 2 mw reaches the best certified lower bound of the maximum.
// Keys, the frame maximum and their consumers (corner_select_kernel) are unchanged; no key carries a strip flag.
constexpr int kSW = 256;     // pixels per strip (64 lanes x 4), all owned
constexpr int kTierMaxSteps = 144;   // rows a wave visits: at most 134 + 6 (vs_stream_segments)
constexpr int kTQ1 = 576;    // possible pixels staged per wave for its next global append (at most 63 left over + 2 rows of 256)
constexpr float kTierSqrt = 0.032f, kTierLin = 80.f / 16777216.f, kTierAbs = 0.004f;   // 2 m(tr) + comparison roundings
constexpr float kTierC0Up = 5.3398290e-8f;   // c0 = 0.5 / 3060^2 = 5.33982656e-8, rounded up by 2^-21 and more
// histogram of the listed upper bounds: 8 bins per octave (exponent + 3 mantissa bits) from 2^-6 to 2^26 (U~ < 2^25.2);
// smaller values share bin 0, whose lower edge is therefore "everything"
constexpr int kTierBins = 256, kTierBin0 = (127 - 6) << 3;
__device__ __forceinline__ uint32_t tier_bin(uint32_t ordered_hi) {
    const int b = (int)((ordered_hi >> 20) & 0x7FFu) - kTierBin0;
    return (uint32_t)(b < 0 ? 0 : (b > kTierBins - 1 ? kTierBins - 1 : b));
}
__device__ __forceinline__ uint32_t tier_bin_edge(uint32_t bin) {   // smallest ordered value of a bin
    return bin == 0u ? 0u : (0x80000000u | ((bin + (uint32_t)kTierBin0) << 20));
}

// cvtColor(BGR2GRAY) of one pixel (bytes B, G, R, x of px; gray.hip has the plain form): twice the weighted sum, so that
// the gray value (sum >> 15) is byte 2 of the result — where v_cvt_f32_ubyte2 picks it up — and with the weights split
// into high and low bytes so that two v_dot4_u32_u8 form it: 2 * (3735, 19235, 9798) = 256 * (29, 150, 76) + (46, 70, 140),
// 2 * 2^14 = 256 * 128.  At most 255 * 65536 + 32768 < 2^24: byte 3 stays zero.
__device__ __forceinline__ uint32_t gray_x2_16(uint32_t px) {
    constexpr uint32_t kHi = 29u | (150u << 8) | (76u << 16), kLo = 46u | (70u << 8) | (140u << 16);
    return __builtin_amdgcn_udot4(px, kLo, __builtin_amdgcn_udot4(px, kHi, 128u, false) << 8, false);
}

// Every ring has two slots (index t & 1): a value of row t - 2 is read, at the latest, while row t's is formed.
struct TierState {
    float hx[2][6], q[2][6], rs[2][6];   // per gray row: x-derivative parts, their two-row sums, smoothed values (columns x-1 .. x+4)
    float T[12], X[2][12];               // horizontal 3-sums of xx, xy, yy (see StreamState), integers in f32
    float ctr[2][4], hm[2][4];           // per response row: U~ and its horizontal 3-maxima
    float lf[2], rg[2], trm[2];          // U~ of the lane to the left / right, largest tr of the row's six columns
    uint32_t raw[2][3];                  // gray dwords (x-4, x, x+4) of the next two rows; BGR input: the lane's 12 bytes
    uint32_t halo;                       // BGR input: the next row's pixels beside the strip
    float lanelow;                       // certified lower bound of the best response of this lane's pixels (U units)
};

struct TierArgs {
    const uint8_t *src;
    unsigned long long *queue;       // this wave's LDS staging queue
    unsigned long long *list;        // this frame's list of possible pixels
    uint32_t *whist;                 // this wave's histogram of the listed upper bounds (LDS)
    uint32_t *count;
    size_t cap;
    int w, h, ys, ye, x, steps;
    int bstride;                     // bytes per row of src (the gray form's rows may be longer than w: vslam_ctx::img_pitch)
    uint8_t *gout;                   // BGR input: this frame's gray image (written for the rows [ys, ye))
    const uint32_t *halo;            // BGR input, LDS: per row of the wave's segment the gray pixels left / right of the strip
    uint32_t voff_l, voff_c, voff_r;
    bool edge, left_fix, right_fix, own_lane;
    unsigned long long own;          // owned lanes
    bool flip2, flip3, flip4;        // padded gray rows (w % 4 != 0): this lane's column 2 / 3 / 4 of six is the image's column w
    unsigned long long cand_ok[4];   // lanes whose pixel i is an owned candidate position (1 <= x < w - 1)
    unsigned long long bord[4];      // lanes whose pixel i is an owned first / last column
    float thrU;                      // lower bound of the final threshold, U units (-inf: none yet)
    float lowU;                      // certified lower bound of the frame maximum, U units
    float qf;                        // quality level, a hair low
    uint32_t *low_max;               // this frame's shared lower bound (ordered float)
    uint32_t published;
    int qn;
};

// Tighten the wave's bounds with the best certified response its lanes have seen, and share it with the frame's other waves.
__device__ __forceinline__ void tier_tighten(TierArgs &a, float lanelow, int lane) {
    const float ninf = -__builtin_inff();
    float low = a.own_lane ? lanelow : ninf;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) low = fmaxf(low, __shfl_xor(low, off, 64));
    low = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(low)));   // the wave's bounds live in scalar registers
    if (low > a.lowU) {
        a.lowU = low;
        a.thrU = low > 0.f ? low * a.qf : ninf;
        const uint32_t k = f2ord(low);
        if (lane == 0 && k > a.published) atomicMax(a.low_max, k);   // fire and forget
        a.published = k > a.published ? k : a.published;
    }
}

__device__ __forceinline__ void tier_flush(TierArgs &a, int lane) {
    if (a.qn == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.count, (uint32_t)a.qn);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < a.qn; i += 64) {
        const size_t pos = (size_t)base + i;
        const unsigned long long ent = a.queue[i];
        if (pos < a.cap) a.list[pos] = ent;
        atomicAdd(a.whist + tier_bin((uint32_t)(ent >> 32)), 1u);   // this wave's histogram, in LDS
    }
    a.qn = 0;
}

template <int P, bool BGR>   // P = t % 2; BGR: src is the 3-byte image and the gray rows are formed (and written) here
__device__ __forceinline__ void tier_step(TierState &st, TierArgs &a, int t, int lane) {
    constexpr int Q = P ^ 1;
    uint32_t d0 = st.raw[P][0];
    uint32_t d1 = st.raw[P][1];
    uint32_t d2 = st.raw[P][2];
    float g[8];
    if constexpr (BGR) {
        // the lane's 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const uint32_t u0 = gray_x2_16(d0), u1 = gray_x2_16(__builtin_amdgcn_alignbyte(d1, d0, 3)),
                       u2 = gray_x2_16(__builtin_amdgcn_alignbyte(d2, d1, 2)), u3 = gray_x2_16(d2 >> 8);
        {   // Prefetch two rows ahead into the slot just consumed — "consumed" made a fact for the compiler by letting the
            // addresses depend on the values formed from the old contents: with the loads in front of their last use
            // (where its scheduler likes them) old and new contents are alive together, the new ones get registers of their
            // own, and the copy into the slot's at the end of the iteration waits for every load in flight.
            const int tn = t + 2 < a.steps ? t + 2 : a.steps - 1;
            const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + tn, a.h) * a.bstride;
            uint32_t oc = a.voff_c;
            asm volatile("" : "+v"(oc) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            const uint32_t *vp = reinterpret_cast<const uint32_t *>(rowp + oc);
            st.raw[P][0] = vp[0];
            st.raw[P][1] = vp[1];
            st.raw[P][2] = vp[2];
        }
        g[2] = cvt_ubyte<2>(u0); g[3] = cvt_ubyte<2>(u1); g[4] = cvt_ubyte<2>(u2); g[5] = cvt_ubyte<2>(u3);
        const uint32_t p01 = __builtin_amdgcn_perm(u1, u0, 0x0c0c0602u);
        d1 = __builtin_amdgcn_perm(u3, __builtin_amdgcn_perm(u2, p01, 0x0c060100u), 0x06020100u);   // the lane's four gray bytes
        const int gy = a.ys - 3 + t;
        if (gy >= a.ys && gy < a.ye && a.own_lane) {
            // Written as an instruction of its own, not as a C++ store: loads and stores share one counter on this chip
            // and the compiler, which has to assume that they complete out of order relative to each other, would wait for
            // EVERYTHING in flight (vmcnt(0): the next row's prefetch and this store's acknowledgement) in front of every row.
            // Unseen by it, the store only makes its waits for the loads longer than they need to be, never too short.
            // The scheme leans on gfx950 behaviour (one vmcnt for loads and stores, loads returning in order among
            // themselves) and on nothing in this kernel reading gout back: any other target gets the plain store, and
            // VSLAM_NO_GRAY_FUSION=1 (bgr2gray in front of the gray form) stays the A/B reference in the tests.
            uint8_t *rowp = a.gout + (size_t)gy * a.w;
#if defined(__gfx950__)
            asm volatile("global_store_dword %0, %1, %2" : : "v"(a.x), "v"(d1), "s"(rowp) : "memory");
#else
            *reinterpret_cast<uint32_t *>(rowp + a.x) = d1;
#endif
        }
        // the neighbours' bytes; beyond the strip: the two pixels left of it (bytes 2, 3 of the dword a lane 0 wants) and the
        // two right of it (bytes 0, 1), converted when the wave started
        // (one dword per row: bytes 2, 3 = the left pair, bytes 0, 1 = the right pair — each consumer looks at its own half;
        // read one row ahead, or every row would wait out the LDS latency in front of the two moves below)
        const uint32_t gl = st.halo, gr = st.halo;
        st.halo = a.halo[t + 1];
        d0 = (uint32_t)__builtin_amdgcn_update_dpp((int)gl, (int)d1, 0x138, 0xf, 0xf, false);   // lane i <- lane i - 1
        d2 = (uint32_t)__builtin_amdgcn_update_dpp((int)gr, (int)d1, 0x130, 0xf, 0xf, false);   // lane i <- lane i + 1
    }
    if (a.edge) {
        if (a.left_fix) d0 = (d1 & 0x00FF0000u) | ((d1 & 0x0000FF00u) << 16);
        if (a.right_fix) d2 = ((d1 >> 16) & 0xFFu) | (d1 & 0xFF00u);
    }
    g[0] = cvt_ubyte<2>(d0); g[1] = cvt_ubyte<3>(d0);
    if constexpr (!BGR) {
        g[2] = cvt_ubyte<0>(d1); g[3] = cvt_ubyte<1>(d1); g[4] = cvt_ubyte<2>(d1); g[5] = cvt_ubyte<3>(d1);
    }
    g[6] = cvt_ubyte<0>(d2); g[7] = cvt_ubyte<1>(d2);
    if constexpr (!BGR) {   // prefetch two rows ahead into the slot just consumed (see the other form above)
        const int tn = t + 2 < a.steps ? t + 2 : a.steps - 1;
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + tn, a.h) * a.bstride;
        uint32_t ol = a.voff_l, oc = a.voff_c, orr = a.voff_r;
        asm volatile("" : "+v"(ol), "+v"(oc), "+v"(orr) : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]));
        st.raw[P][0] = *reinterpret_cast<const uint32_t *>(rowp + ol);
        st.raw[P][1] = *reinterpret_cast<const uint32_t *>(rowp + oc);
        st.raw[P][2] = *reinterpret_cast<const uint32_t *>(rowp + orr);
    }
    float dy[6];
    {
        float pr[7];
#pragma unroll
        for (int j = 0; j < 7; j++) pr[j] = g[j] + g[j + 1];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            st.hx[P][c] = g[c + 2] - g[c];
            st.q[P][c] = st.hx[Q][c] + st.hx[P][c];          // hx of this row and the one before
            const float rs = pr[c] + pr[c + 1];              // g(c-1) + 2 g(c) + g(c+1)
            dy[c] = rs - st.rs[P][c];                        // the slot still holds row g - 2
            st.rs[P][c] = rs;
        }
    }
    if (t < 2) return;

    // integer Sobel responses on row p = g - 1 and their products
    const int prow = a.ys - 4 + t;
    const bool rowflip = prow < 0 || prow >= a.h;
    float dx[6];
#pragma unroll
    for (int c = 0; c < 6; c++) dx[c] = st.q[Q][c] + st.q[P][c];   // hx(p-1) + 2 hx(p) + hx(p+1)
    float dx0 = dx[0], dx5 = dx[5];   // for the xy products only: a mirrored column changes that product's sign
    float dx2 = dx[2], dx3 = dx[3], dx4 = dx[4];
    if (a.edge) {
        if (a.left_fix) dx0 = -dx0;
        if (a.right_fix) dx5 = -dx5;
        // Padded rows: the bytes behind the last column are the row's mirror image, so column w of the derivatives is the
        // mirrored column w - 2 (the one the box filter's own REFLECT_101 asks for at column w - 1) up to the sign of dx.
        if constexpr (!BGR) {   // (the BGR form takes packed rows only: nothing of this in its code)
            if (a.flip2) dx2 = -dx2;
            if (a.flip3) dx3 = -dx3;
            if (a.flip4) dx4 = -dx4;
        }
    }
    float (&r)[12] = st.X[P];
    const float (&rp)[12] = st.X[Q];
    {   // r(i) = c(i) + c(i+1) + c(i+2), i = 0..3, over the six columns: 3 multiplies, 5 fused and 4 plain adds per channel
        const float a1 = dx[1] * dx[1], a2 = dx[2] * dx[2], a3 = dx[3] * dx[3];
        r[0] = __builtin_fmaf(dx[0], dx[0], a1) + a2;
        r[1] = (a1 + a2) + a3;
        r[2] = __builtin_fmaf(dx[4], dx[4], a2 + a3);
        r[3] = __builtin_fmaf(dx[5], dx[5], __builtin_fmaf(dx[4], dx[4], a3));
        const float b1 = dx[1] * dy[1], b2 = dx2 * dy[2], b3 = dx3 * dy[3];
        r[4] = __builtin_fmaf(dx0, dy[0], b1) + b2;
        r[5] = (b1 + b2) + b3;
        r[6] = __builtin_fmaf(dx4, dy[4], b2 + b3);
        r[7] = __builtin_fmaf(dx5, dy[5], __builtin_fmaf(dx4, dy[4], b3));
        const float c1 = dy[1] * dy[1], c2 = dy[2] * dy[2], c3 = dy[3] * dy[3];
        r[8] = __builtin_fmaf(dy[0], dy[0], c1) + c2;
        r[9] = (c1 + c2) + c3;
        r[10] = __builtin_fmaf(dy[4], dy[4], c2 + c3);
        r[11] = __builtin_fmaf(dy[5], dy[5], __builtin_fmaf(dy[4], dy[4], c3));
    }
    if (rowflip) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; i++) r[4 + i] = -r[4 + i];
    }
    if (t >= 4) {
        float u4[4], tr4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float A = st.T[i] + r[i], B = st.T[4 + i] + r[4 + i], C = st.T[8 + i] + r[8 + i];
            const float tr = A + C, d = A - C, b2 = B + B;
            const float tt = __builtin_fmaf(b2, b2, d * d);
            tr4[i] = tr;
            u4[i] = tr - __builtin_amdgcn_sqrtf(tt);
        }
        const float ninf = -__builtin_inff();
        const float lf = dpp_wave_shr1(u4[3], ninf), rg = dpp_wave_shl1(u4[0], ninf);
        float hmn[4];
        hmn[0] = max3_nonan(lf, u4[0], u4[1]);
        hmn[1] = max3_nonan(u4[0], u4[1], u4[2]);
        hmn[2] = max3_nonan(u4[1], u4[2], u4[3]);
        hmn[3] = max3_nonan(u4[2], u4[3], rg);
        const float trn = max3_nonan(max3_nonan(tr4[0], tr4[1], tr4[2]), dpp_wave_shr1(tr4[3], 0.f),
                                     max3_nonan(tr4[3], dpp_wave_shl1(tr4[0], 0.f), 0.f));
        if (t >= 6) {
            const int ty = a.ys - 6 + t;   // ys <= ty < ye by construction
            const float tm = max3_nonan(st.trm[P], st.trm[Q], trn);
            const float m2 = __builtin_fmaf(kTierSqrt, __builtin_amdgcn_sqrtf(tm), __builtin_fmaf(kTierLin, tm, kTierAbs));
            const float (&v)[4] = st.ctr[Q];
            if (!BGR && a.edge) {   // a lane across the last column (padded rows): its pixels behind the image certify nothing
                float vin[4];
#pragma unroll
                for (int i = 0; i < 4; i++) vin[i] = (((a.cand_ok[i] | a.bord[i]) >> lane) & 1ull) ? v[i] : -__builtin_inff();
                st.lanelow = max3_nonan(st.lanelow, max3_nonan(vin[0], vin[1], vin[2]) - m2, vin[3] - m2);
            } else {
                st.lanelow = max3_nonan(st.lanelow, max3_nonan(v[0], v[1], v[2]) - m2, v[3] - m2);
            }
            const bool brow = ty == 0 || ty == a.h - 1;   // uniform
            if (brow) tier_tighten(a, st.lanelow, lane);   // this row's pixels only count for the maximum: know it first
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float left = i == 0 ? st.lf[Q] : v[i - 1], right = i == 3 ? st.rg[Q] : v[i + 1];
                const float m8 = max3_nonan(max3_nonan(st.hm[P][i], hmn[i], left), right, a.thrU);
                const float vp = v[i] + m2;
                unsigned long long bal = 0ull, bmax = 0ull;
                if (!brow) {
                    bal = __builtin_amdgcn_fcmpf(vp, m8, 3) & a.cand_ok[i];   // 3 = ordered >=
                    // (first / last column of the image: pixel 0 / 3 of a lane when w % 4 == 0; with padded rows -- the gray form only --
                    // the last column can be any of a lane's four)
                    if (a.edge && (!BGR || i == 0 || i == 3)) bmax = __builtin_amdgcn_fcmpf(vp, a.lowU, 3) & a.bord[i];
                } else {
                    bmax = __builtin_amdgcn_fcmpf(vp, a.lowU, 3) & (BGR ? a.own : (a.cand_ok[i] | a.bord[i]));   // the lane's pixels inside the image
                }
                const unsigned long long both = bal | bmax;
                if (both) {
                    const int pos = a.qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(both >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)both, 0u));
                    if ((both >> lane) & 1ull) {
                        // entry: upper bound (vp > 0, so | sign bit = its ordered form) << 32 | offset | "not certainly a
                        // 3x3 maximum".  What else the exact tier has to know follows from the position: first / last row
                        // or column = counts for the maximum only; first / last column of a strip = the neighbours beyond
                        // the strip were -inf here.
                        uint32_t lo = (uint32_t)(ty * a.w + a.x + i);
                        if (!((v[i] - m2) - m8 >= 0.f)) lo |= kKeyUncertain;
                        a.queue[pos] = ((unsigned long long)(__float_as_uint(vp) | 0x80000000u) << 32) | lo;
                    }
                    a.qn += __popcll(both);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            st.ctr[P][i] = u4[i];
            st.hm[P][i] = hmn[i];
        }
        st.lf[P] = lf;
        st.rg[P] = rg;
        st.trm[P] = trn;
    }
#pragma unroll
    for (int e = 0; e < 12; e++) st.T[e] = rp[e] + r[e];
}

// Tier 1: U~ for every pixel; the possible pixels go to list[f][0 .. counts[f]) as raw entries (image_common.h; the
// upper bound in units of c0); low_max[f] collects the certified lower bound of the frame maximum (ordered float, same
// units) that the waves share, hist[f][] counts the entries by upper bound.
// BGR: `src` is the 3-byte image (bstride bytes per row; rows that do not start on a dword cost nothing measurable: the lane's
// 12 bytes are one unaligned global_load_dwordx3) and the kernel is cvtColor as well:
// it forms the gray rows it needs from the lane's own 12 bytes, hands the neighbour pixels across lanes, and writes the
// rows it owns to gray_out for the stages that follow (blur, exact tier) — bgr2gray's 0.30 ms of pure HBM time are paid
// here as ≈ 25 vector instructions per row on top of 225, while the kernel's loads wait behind its arithmetic anyway.
template <bool BGR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void min_eigen_tiered_kernel(
    const uint8_t *__restrict__ src, int bstride, uint8_t *__restrict__ gray_out, int w, int h,
    uint32_t *__restrict__ low_max, uint32_t *__restrict__ hist,
    double quality, unsigned long long *__restrict__ list, uint32_t *__restrict__ counts, size_t cap, int seg_rows,
    int frames, int strips, int per_frame) {
    __shared__ unsigned long long queue[4][kTQ1];
    __shared__ uint32_t whist[4][kTierBins];
    __shared__ uint32_t halo[BGR ? 4 : 1][BGR ? kTierMaxSteps + 1 : 1];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    int f, blk;
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    const int strip = blk % strips, segblk = blk / strips;
    TierArgs a;
    a.ys = (segblk * 4 + wave) * seg_rows;
    if (a.ys >= h) return;   // whole wave; the kernel has no barriers
    a.ye = a.ys + seg_rows < h ? a.ys + seg_rows : h;
    a.steps = a.ye - a.ys + 6;
    a.w = w;
    a.h = h;
    a.src = src + (size_t)f * h * (size_t)bstride;
    a.bstride = bstride;
    a.gout = BGR ? gray_out + (size_t)f * w * h : nullptr;
    a.queue = queue[wave];
    a.list = list + (size_t)f * cap;
    a.whist = whist[wave];
    for (int i = lane; i < kTierBins; i += 64) a.whist[i] = 0u;
    a.count = counts + f;
    a.cap = cap;
    const int x0 = strip * kSW;
    a.x = x0 + 4 * lane;
    a.own_lane = a.x < w;
    a.own = __ballot(a.own_lane);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a.cand_ok[i] = __ballot(a.own_lane && a.x + i >= 1 && a.x + i < w - 1);
        a.bord[i] = __ballot(a.own_lane && (a.x + i == 0 || a.x + i == w - 1));
    }
    a.edge = x0 == 0 || x0 + kSW + 4 > w;
    a.left_fix = a.x == 0;
    a.right_fix = a.x + 4 == w;
    a.flip2 = w - a.x == 1;   // (never with w % 4 == 0)
    a.flip3 = w - a.x == 2;
    a.flip4 = w - a.x == 3;
    const int lw = BGR ? w : bstride;   // what a row holds: the gray form's rows may carry a mirrored tail
    const int xc = a.x < 0 ? 0 : (a.x > lw - 4 ? lw - 4 : a.x);
    if constexpr (BGR) {   // the lane's 12 bytes
        a.voff_c = (uint32_t)(3 * xc);
    } else {
        a.voff_c = (uint32_t)xc;
        a.voff_l = (uint32_t)(xc - 4 < 0 ? 0 : xc - 4);
        a.voff_r = (uint32_t)(xc + 4 > lw - 4 ? lw - 4 : xc + 4);
    }
    if constexpr (BGR) {
        // The two gray pixels on either side of the strip, for every row this wave will visit: 4 x steps <= 576 pixels, up to
        // nine per lane, all loads in flight together while the registers are still free; the rows then fetch their pair of
        // dwords from LDS.  (On the scalar unit, row by row, the same cost 0.11 ms: 50 scalar instructions per row are a
        // quarter of a row's issue time, and a wave that is at them is not feeding the vector pipe.)
        uint8_t *hb = reinterpret_cast<uint8_t *>(halo[wave]);
        uint32_t v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int idx = lane + 64 * k;
            const int r = (idx >> 2) < a.steps ? (idx >> 2) : a.steps - 1, j = idx & 3;
            int px = j < 2 ? x0 - 2 + j : x0 + kSW - 2 + j;
            px = px < 0 ? 0 : (px > w - 2 ? w - 2 : px);   // out of the image: not used (the mirrored columns come from the lane's own)
            __builtin_memcpy(&v[k], a.src + (size_t)reflect101(a.ys - 3 + r, h) * bstride + 3 * px, 4);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int idx = lane + 64 * k;
            if (idx < 4 * a.steps) hb[(idx >> 2) * 4 + ((idx + 2) & 3)] = (uint8_t)(gray_x2_16(v[k]) >> 16);   // j = 0, 1 -> bytes 2, 3; j = 2, 3 -> bytes 0, 1
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        a.halo = halo[wave];
    }
    const float ninf = -__builtin_inff();
    const float qf = __int_as_float(__builtin_amdgcn_readfirstlane(
        __float_as_int((float)quality * (1.f - 1.f / 1048576.f))));   // threshold bound: quality, a hair low
    a.qn = 0;

    TierState st;
    st.lanelow = ninf;
    st.halo = 0u;
    if constexpr (BGR) st.halo = a.halo[0];
#pragma unroll
    for (int i = 0; i < 12; i++) st.T[i] = st.X[0][i] = st.X[1][i] = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) st.hx[0][c] = st.hx[1][c] = st.q[0][c] = st.q[1][c] = st.rs[0][c] = st.rs[1][c] = 0.f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        st.trm[k] = 0.f;
        if constexpr (BGR) {
            const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + k, h) * bstride;
            const uint32_t *vp = reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
            st.raw[k][0] = vp[0]; st.raw[k][1] = vp[1]; st.raw[k][2] = vp[2];
        } else {
            const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + k, h) * bstride;
            st.raw[k][0] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_l);
            st.raw[k][1] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_c);
            st.raw[k][2] = *reinterpret_cast<const uint32_t *>(rowp + a.voff_r);
        }
    }
    // what other waves have certified so far (0 = nothing yet)
    a.qf = qf;
    a.low_max = low_max + f;
    a.published = low_max[f];
    a.lowU = a.published == 0u ? ninf : ord2f(a.published);
    a.thrU = a.lowU > 0.f ? a.lowU * qf : ninf;
    if constexpr (BGR) {
        // Enter the loop with nothing in flight.  The compiler's wait in front of a row's first use has to hold on every
        // path to it, and on the path from here the two rows above would be the youngest loads outstanding: it would
        // wait for everything in every iteration (seen: vmcnt(0) at the loop head, 0.90 instead of 0.7 ms).
        asm volatile("" : : "v"(st.raw[0][0]), "v"(st.raw[0][1]), "v"(st.raw[0][2]), "v"(st.raw[1][0]), "v"(st.raw[1][1]), "v"(st.raw[1][2]));
    }
    // Two rows per iteration, unconditionally (an odd last row follows the loop): with the second one under a condition,
    // the path around it reaches the loop head with the first row's loads as the youngest in flight, and the wait in front
    // of their use — one wait for all paths — could not leave the second row's loads outstanding.
    int t0 = 0;
    for (; t0 + 1 < a.steps; t0 += 2) {
        tier_step<0, BGR>(st, a, t0, lane);
        tier_step<1, BGR>(st, a, t0 + 1, lane);
        if (a.qn >= kTQ1 - 512) tier_flush(a, lane);
        // after the first two tested rows, then every 12 rows: tighten the bounds with what this wave has seen
        if ((t0 % 12) == 10 || t0 == 6) tier_tighten(a, st.lanelow, lane);
    }
    if (t0 < a.steps) tier_step<0, BGR>(st, a, t0, lane);
    tier_tighten(a, st.lanelow, lane);
    tier_flush(a, lane);
    for (int i = lane; i < kTierBins; i += 64) {   // this wave's share of the frame's histogram
        const uint32_t c = a.whist[i];
        if (c) atomicAdd(hist + (size_t)f * kTierBins + i, c);
    }
}

// Tier 2: the oracle's exact arithmetic for the listed pixels that can matter.  The selection needs the best-ranked
// candidates only (about 1.3 x maxCorners, twice that when the suppression rejects many), so the entries are cut at the
// upper bound above which the list holds `n_safe` entries (from the detector's histogram; a listed bound is never below
// the exact response, so whatever is cut lies below that edge exactly as well); entries that may hold the frame maximum
// are kept in any case.  Kept entries are evaluated 64 at a time per wave: their own value first — certain 3x3 maxima
// become keys — then, eight pixels x eight neighbours per round, the neighbours of those that still have to be
// compared (featureselect.cpp: val == dilate(val)).  Output: keys2[f][0 .. count2[f]) exact keys, unordered;
// frame_max[f] exact; cutkey[f] = the edge in response units (0: nothing was cut).  mode 1 is the rerun for the frames
// the selection flagged in need[] because it ran out of candidates above the edge: everything is evaluated.
constexpr int kXQ1 = 320, kXQ2 = 72, kXKQ = 256;   // kept entries (63 left over + 4 x 64 per scan step), neighbour work, finished keys
struct ExactLds {
    uint32_t q1[kXQ1];
    uint32_t q2pos[kXQ2], q2e[kXQ2], q2mask[kXQ2];
    unsigned long long keys[kXKQ];
};
struct ExactWave {
    const uint8_t *src;
    ExactLds *lds;
    unsigned long long *out;
    uint32_t *out_count;
    size_t cap;
    int w, h, pitch;   // pitch: bytes per row of src
    float k0, k1;
    int q1n, q2n, kqn;
    float emax;   // per lane
};

__device__ __forceinline__ void exact_flush_keys(ExactWave &x, int lane) {
    if (x.kqn == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(x.out_count, (uint32_t)x.kqn);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < x.kqn; i += 64) {
        const size_t pos = (size_t)base + i;
        if (pos < x.cap) x.out[pos] = x.lds->keys[i];
    }
    x.kqn = 0;
}

__device__ __forceinline__ void exact_emit(ExactWave &x, int lane, bool emit, uint32_t pos, float e) {
    const unsigned long long bal = __ballot(emit);
    if (!bal) return;
    if (x.kqn > kXKQ - 64) exact_flush_keys(x, lane);
    const int at = x.kqn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (emit) x.lds->keys[at] = ((unsigned long long)f2ord(e) << 32) | pos;
    x.kqn += __popcll(bal);
}

// rounds while a queue holds a full one (all: until both are empty)
__device__ __forceinline__ void exact_rounds(ExactWave &x, int lane, bool all) {
    const float ninf = -__builtin_inff();
    while (true) {
        int mode, take;
        if (x.q2n >= 8 || (all && x.q1n == 0 && x.q2n > 0)) {
            mode = 2;
            take = x.q2n < 8 ? x.q2n : 8;
        } else if (x.q1n >= 64 || (all && x.q1n > 0)) {
            mode = 1;
            take = x.q1n < 64 ? x.q1n : 64;
        } else {
            break;
        }
        bool have;
        uint32_t pos = 0, aux = 0, mask = 0;
        int px, py;
        const int k = lane & 7;
        if (mode == 1) {
            have = lane < take;
            uint32_t lo = 0;
            if (have) lo = x.lds->q1[x.q1n - take + lane];
            pos = lo & kOffMask;
            x.q1n -= take;
            py = (int)(pos / (uint32_t)x.w);
            px = (int)pos - py * x.w;
            if (have) {
                // the detector's strips are kSW wide: it did not see the neighbours beyond a strip's first / last column
                if ((px & (kSW - 1)) == 0) mask |= 0x29u;          // neighbours 0, 3, 5
                if ((px & (kSW - 1)) == kSW - 1) mask |= 0x94u;    // neighbours 2, 4, 7
                if (lo & kKeyUncertain) mask = 0xFFu;
                if (px == 0 || px == x.w - 1 || py == 0 || py == x.h - 1) mask = 0x100u;   // not a candidate position
            }
        } else {
            const int ent = x.q2n - take + (lane >> 3);
            have = (lane >> 3) < take;
            if (have) {
                pos = x.lds->q2pos[ent];
                aux = x.lds->q2e[ent];
                mask = x.lds->q2mask[ent];
            }
            x.q2n -= take;
            have = have && ((mask >> k) & 1u);
            py = (int)(pos / (uint32_t)x.w);
            px = (int)pos - py * x.w;
            // neighbour k: 0 1 2 / 3 . 4 / 5 6 7
            px += k < 3 ? k - 1 : (k == 3 ? -1 : (k == 4 ? 1 : k - 6));
            py += k < 3 ? -1 : (k < 5 ? 0 : 1);
        }
        const bool inner = px >= 2 && px < x.w - 2 && py >= 2 && py < x.h - 2;
        float e = ninf;
        if (have && inner) e = min_eigen_exact_interior(x.src, x.pitch, px, py, x.k0, x.k1);
        if (__builtin_expect(__any(have && !inner), 0)) {   // rare: pixels whose window leaves the image
            if (have && !inner) e = min_eigen_exact_border(x.src, x.pitch, x.w, x.h, px, py, x.k0, x.k1);
        }
        x.emax = e > x.emax ? e : x.emax;   // e = -inf where there was nothing to evaluate
        if (mode == 1) {
            exact_emit(x, lane, have && mask == 0u, pos, e);
            const bool more = have && mask != 0u && mask != 0x100u;
            const unsigned long long bal = __ballot(more);
            if (bal) {
                const int p2 = x.q2n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (more) {
                    x.lds->q2pos[p2] = pos;
                    x.lds->q2e[p2] = __float_as_uint(e);
                    x.lds->q2mask[p2] = mask;
                }
                x.q2n += __popcll(bal);
            }
        } else {
            const float ep = __uint_as_float(aux);
            const unsigned long long gr = __ballot(have && e > ep);
            const bool rejected = ((gr >> (lane & ~7)) & 0xFFull) != 0ull;
            exact_emit(x, lane, (lane >> 3) < take && k == 0 && !rejected, pos, ep);
        }
    }
    if (all) exact_flush_keys(x, lane);
}

__global__ __launch_bounds__(256) void corner_exact_kernel(const uint8_t *__restrict__ gray, int w, int h, int pitch,
                                                           const unsigned long long *__restrict__ list,
                                                           const uint32_t *__restrict__ counts, size_t cap,
                                                           const uint32_t *__restrict__ hist, const uint32_t *__restrict__ low_max,
                                                           uint32_t n_safe, unsigned long long *__restrict__ keys2,
                                                           uint32_t *__restrict__ count2, uint32_t *__restrict__ frame_max,
                                                           uint32_t *__restrict__ cutkey, const uint32_t *__restrict__ need,
                                                           int mode, int frames, int per_frame) {
    __shared__ ExactLds lds[4];
    __shared__ uint32_t s_wave[4], s_cut;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int lane = tid & 63;
    int f, blk;
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    if (mode == 1 && need[f] == 0u) return;
    uint32_t n = counts[f];
    if ((size_t)n > cap) n = (uint32_t)cap;   // the selection kernel reports the overflow

    // the cut: lower edge of the histogram bin at which the count from the top reaches n_safe
    uint32_t cut = 0u;
    if (mode == 0) {
        const uint32_t *H = hist + (size_t)f * kTierBins;
        constexpr int per = kTierBins / 256;
        static_assert(per >= 1, "one thread per bin at least");
        uint32_t own = 0;
#pragma unroll
        for (int b = 0; b < per; b++) own += H[tid * per + b];
        uint32_t incl = own;   // becomes the sum over lanes >= lane of this wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_down(incl, off, 64);
            if (lane + off < 64) incl += o;
        }
        if (lane == 0) s_wave[wave] = incl;
        if (tid == 0) s_cut = 0u;
        __syncthreads();
        uint32_t higher = 0;
        for (int wv = wave + 1; wv < 4; wv++) higher += s_wave[wv];
        uint32_t run = higher + incl - own;   // entries in bins above this thread's
        for (int b = per - 1; b >= 0; b--) {
            const uint32_t mine = H[tid * per + b];
            if (run < n_safe && n_safe <= run + mine) s_cut = tier_bin_edge((uint32_t)(tid * per + b));   // one bin at most
            run += mine;
        }
        __syncthreads();
        cut = s_cut;   // 0: fewer than n_safe entries, keep all
        const uint32_t low = low_max[f];   // whatever may hold the maximum stays in
        if (low < cut) cut = low;
        if (blk == 0 && tid == 0) {
            // the same edge in response units, rounded up: everything that was cut has an exact response below it
            cutkey[f] = cut > 0x80000000u ? f2ord(ord2f(cut) * kTierC0Up) : 0u;
        }
    }

    ExactWave x;
    x.src = gray + (size_t)f * pitch * h;
    x.pitch = pitch;
    x.lds = &lds[wave];
    x.out = keys2 + (size_t)f * cap;
    x.out_count = count2 + f;
    x.cap = cap;
    x.w = w;
    x.h = h;
    x.k1 = (float)(1.0 / ((double)(1 << 2) * 3 * 255.0));
    x.k0 = 2.0f * x.k1;
    x.q1n = x.q2n = x.kqn = 0;
    x.emax = -__builtin_inff();
    const unsigned long long *L = list + (size_t)f * cap;
    // scan this wave's share of the list four batches at a time (independent loads), keep what reaches the cut
    const uint32_t stride = (uint32_t)per_frame * 256u;
    for (uint32_t base = (uint32_t)(blk * 4 + wave) * 64u; base < n; base += 4u * stride) {
        unsigned long long ent[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = base + (uint32_t)u * stride + lane;
            ent[u] = idx < n ? L[idx] : 0ull;   // 0 never reaches a cut (which is 0 only when everything is kept: then idx < n decides)
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = base + (uint32_t)u * stride + lane;
            const bool keep = idx < n && (uint32_t)(ent[u] >> 32) >= cut;
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int at = x.q1n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (keep) x.lds->q1[at] = (uint32_t)ent[u];
                x.q1n += __popcll(bal);
            }
        }
        if (x.q1n >= 64) exact_rounds(x, lane, false);
    }
    exact_rounds(x, lane, true);
    float em = x.emax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) em = fmaxf(em, __shfl_xor(em, off, 64));
    if (lane == 0 && em != -__builtin_inff()) atomicMax(&frame_max[f], f2ord(em));
}

// ------------------------------------------------------------------------------------------
// threshold + 3x3 local maximum -> candidate keys (response << 32 | pixel offset)
// ------------------------------------------------------------------------------------------
constexpr int kCT = 256, kCTW = 64, kCTH = 16;

__global__ __launch_bounds__(kCT) void corner_candidates_kernel(
    const float *__restrict__ eig, int w, int h, const uint32_t *__restrict__ frame_max, double quality,
    unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts,
    size_t key_cap, const VsCornerPool pool) {
    __shared__ float E[kCTH + 2][kCTW + 2];
    __shared__ uint32_t s_cnt, s_base;
    const int f = blockIdx.z, tid = threadIdx.x;   // a frame, or a slot of the fallback pool (everything here is by slot then)
    if (pool.count && f >= vs_pool_used(pool)) return;
    const int x0 = blockIdx.x * kCTW;
    const float *src = eig + (size_t)f * w * h;
    const float mx = ord2f(frame_max[f]);
    const float thr = (float)((double)mx * quality);   // threshold(eig, maxVal*qualityLevel, THRESH_TOZERO)
    for (int y0 = blockIdx.y * kCTH; y0 < h; y0 += gridDim.y * kCTH) {   // the grid may cover only some tile rows
    __syncthreads();
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
}

}  // namespace

int vs_launch_min_eigen(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, float *eig,
                        uint32_t *frame_max_bits) {
    VS_REQUIRE(ctx, gray && eig, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 3 && h >= 3, VSLAM_ERR_INVALID);
    if (frame_max_bits) VS_HIP(ctx, hipMemsetAsync(frame_max_bits, 0, sizeof(uint32_t) * (size_t)frames, ctx->stream));
    VsProfScope ps(ctx, "min_eigen_kernel");
    if (w % 4 == 0 && vs_pitch(ctx, w) == w && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0) && ((reinterpret_cast<uintptr_t>(eig) & 15) == 0)) {
        dim3 grid(vs_div_up(w, kE4W), vs_div_up(h, kE4H), frames);
        min_eigen_v4_kernel<<<grid, 256, 0, ctx->stream>>>(gray, w, h, eig, frame_max_bits, VsCornerPool{});
    } else {
        dim3 grid(vs_div_up(w, kETW), vs_div_up(h, kETH), frames);
        min_eigen_kernel<<<grid, kET, 0, ctx->stream>>>(gray, w, h, vs_pitch(ctx, w), eig, frame_max_bits, VsCornerPool{});
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}


// Candidates for vs_launch_good_features.  Everything in `c` must be zero on entry.
//   width % 4 == 0 (two-tier detector): keys = the raw list, keys2[f][0 .. count2[f]) = exact keys of the entries above
//   the cut, fmax exact, cutkey set; returns raw_list = 1.
//   otherwise (tiled detector): eig filled, keys[f][0 .. counts[f]) = exact keys; raw_list = 0.
size_t vs_response_hist_words(int frames) { return (size_t)frames * kTierBins; }

int vs_launch_response_candidates(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, double quality,
                                  float *eig, const VsCornerCounters &c, unsigned long long *keys,
                                  unsigned long long *keys2, size_t key_cap, uint32_t n_safe, int *raw_list,
                                  const VsBgrSource *bgr) {
    int rc;
    const int gp = vs_pitch(ctx, w);   // bytes per gray row: w, or longer with a mirrored tail (vslam_ctx::img_pitch)
    const bool fused = (gp % 4 == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
    *raw_list = fused ? 1 : 0;
    // the gray image has not been formed yet (bgr != nullptr): the two-tier detector does it on the way for packed gray
    // rows (width % 4 == 0, any row stride of the source), a cvtColor launch in front writes padded rows otherwise
    static const char *const nofuse = VS_EXPERIMENT_ENV("VSLAM_NO_GRAY_FUSION");
    const bool from_bgr = bgr && fused && gp == w && !nofuse &&
                          vs_div_up(h, vs_stream_segments(h, frames, vs_div_up(w, kSW))) + 6 <= kTierMaxSteps;
    if (bgr && !from_bgr)
        if ((rc = vs_launch_bgr2gray(ctx, bgr->data, frames, w, h, bgr->stride, const_cast<uint8_t *>(gray)))) return rc;
    if (fused) {
        if (from_bgr) {
            VsProfScope ps(ctx, "min_eigen_kernel");
            const int strips = vs_div_up(w, kSW);
            const int segs = vs_stream_segments(h, frames, strips);
            const int seg_rows = vs_div_up(h, segs);
            const int per_frame = strips * vs_div_up(segs, 4);
            min_eigen_tiered_kernel<true><<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(
                bgr->data, bgr->stride, const_cast<uint8_t *>(gray), w, h, c.low, c.hist, quality, keys, c.counts, key_cap,
                seg_rows, frames, strips, per_frame);
        } else {
            VsProfScope ps(ctx, "min_eigen_kernel");
            const int strips = vs_div_up(w, kSW);
            const int segs = vs_stream_segments(h, frames, strips);
            const int seg_rows = vs_div_up(h, segs);
            const int per_frame = strips * vs_div_up(segs, 4);
            min_eigen_tiered_kernel<false><<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(
                gray, gp, nullptr, w, h, c.low, c.hist, quality, keys, c.counts, key_cap, seg_rows, frames, strips, per_frame);
        }
        // A frame whose maximum response is not positive has no corners (its threshold max * quality lies at or above
        // every response, THRESH_TOZERO clears the image and zeros are not corners); the selection's exact threshold
        // drops every key of such a frame, so it needs no special handling here.
        if ((rc = vs_launch_corner_exact(ctx, gray, frames, w, h, c, keys, keys2, key_cap, n_safe, 0))) return rc;
    } else {
        if ((rc = vs_launch_min_eigen(ctx, gray, frames, w, h, eig, c.fmax))) return rc;
        VsProfScope ps(ctx, "corner_candidates_kernel");
        dim3 grid(vs_div_up(w, kCTW), vs_div_up(h, kCTH), frames);
        corner_candidates_kernel<<<grid, kCT, 0, ctx->stream>>>(eig, w, h, c.fmax, quality, keys, c.counts, key_cap, VsCornerPool{});
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// The fallback of the two-tier path (select.hip): the frames filed in the pool -- their bounded lists overflowed, or their
// selection needs the per-pixel maps -- are redone with the plain pipeline on whole-image scratch: exact response of every
// pixel and the frame maximum (by slot), then every candidate's exact key.  Always queued; returns at once when the pool
// is empty (two launches of a few thousand idle workgroups).
int vs_launch_pool_candidates(vslam_ctx *ctx, const uint8_t *gray, int w, int h, double quality, const VsCornerPool &pool) {
    VsProfScope ps(ctx, "corner_rerun_kernels");
    if (vs_pitch(ctx, w) != w) {   // padded gray rows: the any-width kernel reads them
        dim3 grid(vs_div_up(w, kETW), vs_div_up(h, kETH), pool.slots);
        min_eigen_kernel<<<grid, kET, 0, ctx->stream>>>(gray, w, h, vs_pitch(ctx, w), pool.eig, pool.fmax, pool);
    } else {
        dim3 grid(vs_div_up(w, kE4W), vs_div_up(h, kE4H), pool.slots);
        min_eigen_v4_kernel<<<grid, 256, 0, ctx->stream>>>(gray, w, h, pool.eig, pool.fmax, pool);
    }
    const int rows = vs_div_up(h, kCTH);
    dim3 grid2(vs_div_up(w, kCTW), rows < 4 ? rows : 4, pool.slots);
    corner_candidates_kernel<<<grid2, kCT, 0, ctx->stream>>>(pool.eig, w, h, pool.fmax, quality, pool.keys, pool.counts, pool.key_cap, pool);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// mode 0: the entries above the cut -> keys2 / count2; mode 1: every entry of the frames flagged in c.need -> keys2 / count3
int vs_launch_corner_exact(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, const VsCornerCounters &c,
                           const unsigned long long *keys, unsigned long long *keys2, size_t key_cap, uint32_t n_safe,
                           int mode) {
    VsProfScope ps(ctx, mode == 0 ? "corner_exact_kernel" : "corner_rerun_kernels");
    static const char *const pf_env = VS_EXPERIMENT_ENV("VSLAM_CORNER_EXACT_WGS");
    const int per_frame = pf_env ? atoi(pf_env) : 4;
    corner_exact_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(
        gray, w, h, vs_pitch(ctx, w), keys, c.counts, key_cap, c.hist, c.low, n_safe, keys2, mode == 0 ? c.count2 : c.count3, c.fmax, c.cutkey,
        c.need, mode, frames, per_frame);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

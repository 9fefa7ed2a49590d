// Shi-Tomasi corner response (cv::cornerMinEigenVal inside cv::goodFeaturesToTrack, reference: src/Frame.cpp:61)
// and the 3x3-local-maximum candidates for gfx950.
//   min_eigen_stream_kernel   the front-end path: response + candidates in one streaming pass (width % 4 == 0)
//   min_eigen_v4_kernel       response only, tiled (vslam_min_eigen, width % 4 == 0)
//   min_eigen_kernel          response only, any width
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
// and has no barriers.  It writes no response image: candidate keys, the frame maximum and each strip's two
// edge columns of responses are everything later stages read.  Arithmetic and its order are those of min_eigen_v4_kernel (see there).
//
// Step t of a segment owning rows [ys, ye):   gray row g = ys - 3 + t   (Sobel parts of row g)
//   t >= 2: products and their horizontal sums on row p = g - 1
//   t >= 4: response row y = p - 1 = ys - 5 + t  (stored when ys <= y < ye)
//   t >= 6: 3x3-maxima test of row y - 1 = ys - 6 + t  -> candidates
// so a segment takes (ye - ys) + 6 steps, 6 of them warm-up (7 % at 90 rows per segment).
constexpr int kSW = 256;     // pixels per strip (64 lanes x 4), all owned
constexpr int kSQ = 512;     // candidate queue entries per wave
struct StreamState {
    float hx[3][6], rr[3][6];      // per gray row: x-derivative parts and smoothed values, columns x-1 .. x+4
    double T[12], X[2][12];        // horizontal 3-sums of xx, xy, yy for the lane's 4 pixels: X[t & 1] = those of the
                                   // product row of step t, X[~t & 1] of the row before, T = the sum of the two rows
                                   // before this step's.  The column sum S(y) = (r(y-1) + r(y)) + r(y+1) is T + r(y+1)
                                   // — the same two additions in the same order as three stored rows would give —
                                   // and neither a third row of state nor a register copy per value is needed: the
                                   // new row is computed straight into the slot of the row that just left the window
                                   // (the difference between 2 and 3 waves per SIMD)
    float ctr[3][4], hm[3][4];     // per response row: the values and their horizontal 3-maxima
    uint32_t raw[3][3];            // prefetched gray dwords (x-4, x, x+4) of the next three rows
    float emax;
};

struct StreamArgs {
    const uint8_t *src;          // frame base
    float *edge_l, *edge_r;      // this strip's first / last column of responses, indexed by row
    unsigned long long *queue;   // this wave's LDS queue
    unsigned long long *keys;    // frame base
    uint32_t *count;             // this frame's candidate counter
    size_t key_cap;
    int w, h, ys, ye, x, steps;
    uint32_t voff_l, voff_c, voff_r;
    bool edge, left_fix, right_fix, own_lane;
    unsigned long long col_ok[4];   // lanes whose pixel i is an owned, testable column (1 <= x < w - 1)
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

template <int K, int P>   // K = t % 3 (three-row rings), P = t % 2 (X)
__device__ __forceinline__ void stream_step(StreamState &st, const StreamArgs &a, int t, int &qn, int lane) {
    constexpr int K1 = (K + 2) % 3, K2 = (K + 1) % 3;   // slots of the previous row and the one before
    uint32_t d0 = st.raw[K][0];
    const uint32_t d1 = st.raw[K][1];
    uint32_t d2 = st.raw[K][2];
    {   // Prefetch the row three steps ahead into the slot just consumed (past the segment's last step the index is
        // clamped: a redundant load costs less than the register copies a conditional one brings).  The lane offsets
        // are made opaque so that their zero-extension is not hoisted into 64-bit register pairs: base in SGPRs +
        // 32-bit lane offset is an addressing mode, a 64-bit vector add is an instruction per load.
        const int tn = t + 3 < a.steps ? t + 3 : a.steps - 1;
        const uint8_t *rowp = a.src + (size_t)reflect101(a.ys - 3 + tn, a.h) * a.w;
        uint32_t ol = a.voff_l, oc = a.voff_c, orr = a.voff_r;
        asm volatile("" : "+v"(ol), "+v"(oc), "+v"(orr));
        st.raw[K][0] = *reinterpret_cast<const uint32_t *>(rowp + ol);
        st.raw[K][1] = *reinterpret_cast<const uint32_t *>(rowp + oc);
        st.raw[K][2] = *reinterpret_cast<const uint32_t *>(rowp + orr);
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
    double (&r)[12] = st.X[P];
    const double (&rp)[12] = st.X[P ^ 1];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r[i] = ((double)cxx[i] + (double)cxx[i + 1]) + (double)cxx[i + 2];
        r[4 + i] = ((double)cxy[i] + (double)cxy[i + 1]) + (double)cxy[i + 2];
        r[8 + i] = ((double)cyy[i] + (double)cyy[i + 1]) + (double)cyy[i + 2];
    }
    if (rowflip) {   // mirrored row (two per frame): same rule; negating the sums equals summing the negated products
        asm volatile("" ::: "memory");   // keep this a branch: as selects it would cost every row
#pragma unroll
        for (int i = 0; i < 4; i++) r[4 + i] = -r[4 + i];
    }
    if (t >= 4) {
        const int y = a.ys - 5 + t;
        float apc[4], tt[4], rt[4], e4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // column sums S(y) = (r(y-1) + r(y)) + r(y+1)
            const float sxx = (float)(st.T[i] + r[i]);
            const float sxy = (float)(st.T[4 + i] + r[4 + i]);
            const float syy = (float)(st.T[8 + i] + r[8 + i]);
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
            // the response image itself is not kept: all that is read back later are the strip's edge columns
            // (corner_select_kernel completes the 3x3 test of the neighbouring strips' edge candidates with them)
            if (lane == 0) a.edge_l[y] = e4[0];
            if (lane == 63) a.edge_r[y] = e4[3];
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
            const bool row_ok = ty >= 1 && ty < a.h - 1;   // uniform; the lane's part of the test is in a.col_ok
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = st.ctr[K1][i];
                const float m = max3_nonan(st.hm[K2][i], st.hm[K1][i], st.hm[K][i]);
                const int xx = a.x + i;
                // candidate: v > thr && !(m > v), in the columns and rows that can be tested — formed from compare masks
                // (as a bool the compiler materialises it per lane and compares it again to get the ballot)
                const unsigned long long bal = row_ok ? (__builtin_amdgcn_fcmpf(v, a.thr_p, 2) & ~__builtin_amdgcn_fcmpf(m, v, 2) & a.col_ok[i]) : 0ull;
                const bool cand = (bal >> lane) & 1ull;
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
    // the two-row sum moves on only now: the block above read the old one
#pragma unroll
    for (int e = 0; e < 12; e++) st.T[e] = rp[e] + r[e];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void min_eigen_stream_kernel(const uint8_t *__restrict__ gray, int w, int h,
                                                               float *__restrict__ edge, uint32_t *__restrict__ frame_max,
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
    a.edge_l = edge + (((size_t)f * strips + strip) * 2 + 0) * h;   // [frames][strips][2][h]
    a.edge_r = a.edge_l + h;
    a.queue = queue[wave];
    a.keys = keys + (size_t)f * key_cap;
    a.count = counts + f;
    a.key_cap = key_cap;
    const int x0 = strip * kSW;
    a.x = x0 + 4 * lane;
    a.own_lane = a.x < w;
#pragma unroll
    for (int i = 0; i < 4; i++) a.col_ok[i] = __ballot(a.own_lane && a.x + i >= 1 && a.x + i < w - 1);
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
    for (int i = 0; i < 12; i++) st.T[i] = st.X[0][i] = st.X[1][i] = 0.0;
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
    for (int t0 = 0; t0 < a.steps; t0 += 6) {   // six steps per trip: every ring index (t % 3, t % 2) is a compile-time constant
        stream_step<0, 0>(st, a, t0, qn, lane);
        if (t0 + 1 < a.steps) stream_step<1, 1>(st, a, t0 + 1, qn, lane);
        if (t0 + 2 < a.steps) stream_step<2, 0>(st, a, t0 + 2, qn, lane);
        if (t0 + 3 < a.steps) stream_step<0, 1>(st, a, t0 + 3, qn, lane);
        if (t0 + 4 < a.steps) stream_step<1, 0>(st, a, t0 + 4, qn, lane);
        if (t0 + 5 < a.steps) stream_step<2, 1>(st, a, t0 + 5, qn, lane);
        if ((t0 % 12) == 6) {   // every 12 rows: tighten the prefilter with this wave's own maximum and publish it
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

}  // namespace

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


// Responses + candidate keys for vs_launch_good_features: keys[f][0 .. counts[f]) = (ordered response << 32 | pixel
// offset [| kKeyCheck*]), frame_max[f] = ordered maximum response.  counts and fmax must be zero on entry.  The streaming
// form fills edge[f][strip][2][h] (vs_response_strips(w) strips) and leaves eig untouched; the tiled form fills eig.
int vs_response_strips(int w) { return vs_div_up(w, kSW); }

int vs_launch_response_candidates(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h, double quality,
                                  float *eig, float *edge, uint32_t *fmax, unsigned long long *keys, uint32_t *counts,
                                  size_t key_cap) {
    int rc;
    const bool fused = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
    if (fused) {   // fmax is zero on entry, like counts (the caller clears both with one memset)
        VsProfScope ps(ctx, "min_eigen_kernel");
        const int strips = vs_div_up(w, kSW);
        const int segs = vs_stream_segments(h, frames, strips);
        const int seg_rows = vs_div_up(h, segs);
        const int per_frame = strips * vs_div_up(segs, 4);
        min_eigen_stream_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, ctx->stream>>>(
            gray, w, h, edge, fmax, quality, keys, counts, key_cap, seg_rows, frames, strips, per_frame);
        // A frame whose maximum response is negative has no corners (its threshold max * quality lies above every
        // response, THRESH_TOZERO clears the image and zeros are not corners); corner_select_kernel's exact
        // threshold drops every key of such a frame, so it needs no special handling here.
    } else {
        if ((rc = vs_launch_min_eigen(ctx, gray, frames, w, h, eig, fmax))) return rc;
        VsProfScope ps(ctx, "corner_candidates_kernel");
        dim3 grid(vs_div_up(w, kCTW), vs_div_up(h, kCTH), frames);
        corner_candidates_kernel<<<grid, kCT, 0, ctx->stream>>>(eig, w, h, fmax, quality, keys, counts, key_cap);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// Grid ORB/FAST extractor for gfx950.
//
// Replaces extract_features(Frame&, nrows, ncols), /root/reference/src/Frame.cpp:16-51 (the
// extractor whose only call is commented out at src/vslam.cpp:63; it is the "ORB/FAST" of the
// north star).  Per cell: black 1-px outline drawn into the image (:32), ORB(500, 1.2, 8, 31, 0, 2,
// HARRIS_SCORE, 31, fastThreshold 20)->detect, replaced by the fastThreshold-5 result when fewer
// than 500 were found (:33-36); keypoints shifted to image coordinates (:37-40); then
// ORB::compute over the whole (outlined) image (:43).
//
// The OpenCV internals follow the oracle (oracle/vso_orb.cpp) result for result.  What the arrangement
// below rests on (round 6; the first-correct form of rounds 1-5 took 239 us per 1280x720 frame):
//   * ORB::detect never reads the 32-px reflect frame of a cell's pyramid: FAST keeps corners >= 31 px inside
//     a level (runByImageBorder), its circle reaches 3 px, Harris 4, the intensity centroid 15.  Cell pyramids
//     are therefore tight images; level 0 is a window of the gray frame.  Levels whose inner region is empty
//     (a side <= 62) produce nothing, nothing depends on them, and they are not built.
//   * FAST-9/16: M = max(dark arc score, bright arc score) per pixel; a pixel is a corner at threshold t iff
//     M > t, its OpenCV score is M - 1 for every t, and the 3x3 non-max test on scores is "M greater than its
//     eight neighbours' M" for every t >= 5 -- so one pass over the inner region (+ a 1-px ring) yields both
//     detectors' raster-ordered lists.  One workgroup per (cell, level) walks row strips: source tile and M
//     tile in LDS, ballot masks per 64-pixel chunk, one scan per strip, ordered writes.
//   * src/Frame.cpp:34-36 keeps the threshold-20 result only if it holds >= 500 keypoints.  With c1 = the
//     count after the first retainBest (known from a histogram of FAST scores, no replay needed) the final
//     count of a level lies in [min(c1, quota), c1]: a cell whose upper bounds sum below 500 takes the
//     threshold-5 detector without running the other, one whose lower bounds reach 500 the reverse; only
//     cells in between run both.
//   * KeyPointsFilter::retainBest = std::nth_element + std::partition: replayed (introselect.h) so the
//     surviving ORDER is libstdc++'s -- it decides descriptor row order.  One wave per list, several lists
//     per workgroup, LDS sized by tier of list length; std::partition's swap pairing is the same
//     two-stopper-list scheme as the Hoare step and runs wave-wide.
//   * Harris only where the second retainBest has something to decide; intensity centroid: half a wave
//     per keypoint, a lane per patch column (rows are contiguous bytes), integer sums reduced by shuffles.
//   * ORB::compute: frame pyramid levels >= 1 carry a 3-row / 4-column reflect margin so the 7x7 blur needs
//     no border logic and every row is dword-aligned; a descriptor sample outside a level's image reads the
//     UNBLURRED reflect value (OpenCV blurs the level in place inside its bordered buffer).
#include "ctx.h"
#include "introselect.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kMaxLevels = 8;
constexpr int kEdge = 31;         // edgeThreshold = patchSize
constexpr int kNFeatures = 500;   // src/Frame.cpp:22-23
constexpr int kFastMin = 5;       // lowest FAST threshold any detector here uses
constexpr int kThr[2] = {20, 5};  // list t = 0: fastThreshold 20, t = 1: the fallback's 5

struct Geom {
    int w, h, cw, ch, ncols, nrows, cells;
    int gp;    // bytes per row of the gray frame and its blurred twin: w, or (w % 4 != 0) a multiple of 16 >= w + 3 whose tail mirrors the row
    int nlv;   // levels that can hold keypoints: both sides of the cell's level > 62
    float scale[kMaxLevels];
    // cell pyramid: level 0 = window of the gray frame; levels 1..nlv-1 tight, row stride a multiple of 4
    int clw[kMaxLevels], clh[kMaxLevels], cstride[kMaxLevels], coff[kMaxLevels], cunit;
    // frame pyramid: level 0 = the gray frame; levels 1..nlv-1 with a margin of 3 rows and 4 (left) / >= 4 (right) columns
    int flw[kMaxLevels], flh[kMaxLevels], fstride[kMaxLevels], foff[kMaxLevels], fframe;
    // inner regions (corners at least 31 px inside the level) and list slots per (unit, threshold)
    int rw[kMaxLevels], rh[kMaxLevels], cap[kMaxLevels], loff[kMaxLevels + 1];
    int per_level[kMaxLevels];   // computeKeyPoints' nfeaturesPerLevel
    // INTER_LINEAR_EXACT tables (ofs | c1 << 16 per destination index): frame x / y, cell x / y, levels 1..nlv-1
    int tXF[kMaxLevels], tYF[kMaxLevels], tXC[kMaxLevels], tYC[kMaxLevels], ttotal;
};

int make_geom(int w, int h, int nrows, int ncols, Geom &G) {
    G = Geom{};
    G.w = w;
    G.h = h;
    G.gp = (w % 4 == 0 || w < 64) ? w : (w + 3 + 15) & ~15;
    G.ncols = ncols;
    G.nrows = nrows;
    G.cells = nrows * ncols;
    G.cw = w / ncols;   // src/Frame.cpp:20
    G.ch = h / nrows;
    G.nlv = 0;
    bool open = true;
    for (int l = 0; l < kMaxLevels; l++) {   // orb.cpp: Size(cvRound(w / scale), cvRound(h / scale)) with float scale
        const float sc = (float)std::pow(1.2, (double)l);
        G.scale[l] = sc;
        const float inv = 1.0f / sc;
        G.clw[l] = (int)std::lrint(G.cw * inv);
        G.clh[l] = (int)std::lrint(G.ch * inv);
        G.flw[l] = (int)std::lrint(w * inv);
        G.flh[l] = (int)std::lrint(h * inv);
        G.rw[l] = G.clw[l] - 2 * kEdge;
        G.rh[l] = G.clh[l] - 2 * kEdge;
        open = open && G.rw[l] > 0 && G.rh[l] > 0;
        if (open) G.nlv = l + 1;
    }
    int co = 0, fo = 0, lo = 0, to = 0;
    G.loff[0] = 0;
    for (int l = 0; l < G.nlv; l++) {
        G.cstride[l] = l == 0 ? G.gp : (G.clw[l] + 3) & ~3;
        G.coff[l] = co;
        if (l > 0) co += G.cstride[l] * G.clh[l];
        G.fstride[l] = l == 0 ? G.gp : ((G.flw[l] + 3) & ~3) + 8;
        G.foff[l] = fo;
        if (l > 0) fo += G.fstride[l] * (G.flh[l] + 6);
        // NMS survivors are isolated: at most ceil(rw / 2) * ceil(rh / 2) of them
        G.cap[l] = ((G.rw[l] + 1) / 2) * ((G.rh[l] + 1) / 2);
        lo += G.cap[l];
        G.loff[l + 1] = lo;
        if (l > 0) {
            G.tXF[l] = to; to += G.flw[l];
            G.tYF[l] = to; to += G.flh[l];
            G.tXC[l] = to; to += G.clw[l];
            G.tYC[l] = to; to += G.clh[l];
        }
    }
    G.cunit = (co + 15) & ~15;
    G.fframe = (fo + 15) & ~15;
    G.ttotal = to;
    {   // computeKeyPoints' nfeaturesPerLevel (all 8 levels: the quotas do not depend on what a level can hold)
        const float factor = (float)(1.0 / 1.2);
        float ndesired = kNFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kMaxLevels));
        int sum = 0;
        for (int l = 0; l < kMaxLevels - 1; l++) {
            G.per_level[l] = (int)std::lrint(ndesired);
            sum += G.per_level[l];
            ndesired *= factor;
        }
        G.per_level[kMaxLevels - 1] = std::max(kNFeatures - sum, 0);
    }
    return 0;
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

// level l of unit u's cell pyramid (unit = frame * cells + cell, cells counted columns outer, rows inner: src/Frame.cpp:27-28)
__device__ __forceinline__ const uint8_t *cell_level(const Geom &G, const uint8_t *gray, const uint8_t *cpyr, int u, int l,
                                                     int &stride) {
    stride = G.cstride[l];
    if (l == 0) {
        const int f = u / G.cells, c = u - f * G.cells;
        const int ci = c / G.nrows, cj = c - ci * G.nrows;
        return gray + (size_t)f * G.gp * G.h + (size_t)(cj * G.ch) * G.gp + ci * G.cw;
    }
    return cpyr + (size_t)u * G.cunit + G.coff[l];
}

// ------------------------------------------------------------------------------------------
// cv::rectangle outlines into the caller's BGR image + gray of the outlined image
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_outline_gray_kernel(uint8_t *__restrict__ bgr, int w, int h, int stride,
                                                                int cw, int ch, int ncols, int nrows,
                                                                uint8_t *__restrict__ gray) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    uint8_t *p = bgr + ((size_t)f * h + y) * stride + 3 * x;
    bool outline = false;
    if (x < cw * ncols && y < ch * nrows) {
        const int cx = x % cw, cy = y % ch;
        outline = cx == 0 || cx == cw - 1 || cy == 0 || cy == ch - 1;
    }
    if (outline) {
        p[0] = p[1] = p[2] = 0;
        gray[((size_t)f * h + y) * w + x] = 0;
    } else {
        gray[((size_t)f * h + y) * w + x] = (uint8_t)gray_of(p[0], p[1], p[2]);
    }
}

// the same for dword-aligned rows and widths that are a multiple of 4: a lane takes 4 pixels = three dwords of BGR
__global__ __launch_bounds__(256) void grid_outline_gray4_kernel(uint8_t *__restrict__ bgr, int w, int h, int stride,
                                                                 int cw, int ch, int ncols, int nrows,
                                                                 uint8_t *__restrict__ gray) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int groups = w >> 2;
    if (i >= groups * h) return;
    const int y = i / groups, x = (i - y * groups) * 4;
    uint8_t *row = bgr + ((size_t)f * h + y) * stride;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(row + 3 * x);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    uint32_t g[4];
    g[0] = gray_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u);
    g[1] = gray_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u);
    g[2] = gray_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u);
    g[3] = gray_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24);
    if (y < ch * nrows) {
        const int cy = y % ch;
        const bool row_line = cy == 0 || cy == ch - 1;
        int cx = x % cw;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (x + k < cw * ncols && (row_line || cx == 0 || cx == cw - 1)) {
                g[k] = 0;
                uint8_t *q = row + 3 * (x + k);
                q[0] = q[1] = q[2] = 0;
            }
            cx = cx + 1 == cw ? 0 : cx + 1;
        }
    }
    *reinterpret_cast<uint32_t *>(gray + ((size_t)f * h + y) * w + x) = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
}

// the same into gray rows of `gp` > w bytes (gp % 4 == 0) whose tail holds the row's BORDER_REFLECT_101 continuation -- of the
// OUTLINED image, which is what the blur that reads past the last column is applied to (Geom::gp).  A lane = 4 gray bytes.
__global__ __launch_bounds__(256) void grid_outline_gray_padded_kernel(uint8_t *__restrict__ bgr, int w, int h, int stride,
                                                                       int cw, int ch, int ncols, int nrows,
                                                                       uint8_t *__restrict__ gray, int gp) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int groups = gp >> 2;
    if (i >= groups * h) return;
    const int y = i / groups, x = (i - y * groups) * 4;
    uint8_t *row = bgr + ((size_t)f * h + y) * stride;
    const bool in_rows = y < ch * nrows;
    const int cy = y % ch;
    const bool row_line = cy == 0 || cy == ch - 1;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sx = reflect101(x + k, w);   // the column whose value lands here
        const uint8_t *q = row + 3 * sx;
        uint32_t g = gray_of(q[0], q[1], q[2]);
        if (in_rows && sx < cw * ncols) {
            const int cx = sx % cw;
            if (row_line || cx == 0 || cx == cw - 1) {
                g = 0;
                if (x + k < w) row[3 * sx] = row[3 * sx + 1] = row[3 * sx + 2] = 0;   // the image's own pixel, once
            }
        }
        out |= g << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(gray + ((size_t)f * h + y) * gp + x) = out;
}

// ------------------------------------------------------------------------------------------
// pyramids: INTER_LINEAR_EXACT (Q8 coefficients on both axes, Q16 accumulate, round half up)
// ------------------------------------------------------------------------------------------
// coefficient for destination index d (resize.cpp interpolationLinear): the table entry is ofs | c1 << 16, c0 = 256 - c1
__device__ __forceinline__ uint32_t linear_coeff(int d, int dst_n, int src_n) {
    const double inv_scale = (double)dst_n / (double)src_n;
    const double scale = 1.0 / inv_scale;
    const double fval = scale * ((double)d + 0.5) - 0.5;
    int ival = (int)floor(fval), c1;
    if (ival >= 0 && src_n > 1) {
        if (ival < src_n - 1) {
            c1 = (int)rint((fval - (double)ival) * 256.0);
        } else {
            ival = src_n - 2;
            c1 = 256;
        }
    } else {
        ival = 0;
        c1 = 0;
    }
    return (uint32_t)ival | ((uint32_t)c1 << 16);
}

__global__ __launch_bounds__(256) void resize_tables_kernel(Geom G, uint32_t *__restrict__ tab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G.ttotal) return;
    for (int l = 1; l < G.nlv; l++) {
        if (i < G.tYF[l]) { tab[i] = linear_coeff(i - G.tXF[l], G.flw[l], G.flw[l - 1]); return; }
        if (i < G.tXC[l]) { tab[i] = linear_coeff(i - G.tYF[l], G.flh[l], G.flh[l - 1]); return; }
        if (i < G.tYC[l]) { tab[i] = linear_coeff(i - G.tXC[l], G.clw[l], G.clw[l - 1]); return; }
        if (i < G.tYC[l] + G.clh[l]) { tab[i] = linear_coeff(i - G.tYC[l], G.clh[l], G.clh[l - 1]); return; }
    }
}

// the index set {reflect101(t, n) : a <= t <= b} is the interval [lo, hi]
__device__ __forceinline__ void reflect_range(int a, int b, int n, int &lo, int &hi) {
    const int ra = reflect101(a, n), rb = reflect101(b, n);
    lo = (a <= 0 && b >= 0) ? 0 : min(ra, rb);
    hi = (a <= n - 1 && b >= n - 1) ? n - 1 : max(ra, rb);
}

// Level l of every frame's pyramid (kFrame: with its reflect margin) or of every cell's pyramid from level l - 1.  A workgroup
// makes a 64 x 64 tile, a lane 4 pixels x 4 rows: the x-side of the interpolation (tap offsets, coefficients) is formed once per
// lane and serves its four rows.  The source pixels a tile touches form a box of about 83 x 83; it is fetched as (unaligned)
// dwords into LDS and the four taps per pixel are LDS byte reads: per-lane byte gathers from global memory cost a wave 16
// address cycles each.  The kernel is held by its vector instructions (0.76 of the pipes at 43 per pixel in the 2-row form).
constexpr int kRTW = 64, kRTH = 64, kRPR = 4, kRBW = 112, kRBH = 88;   // tile; rows per lane; box capacity (row pitch kRBW bytes)

template <bool kFrame>
__device__ __forceinline__ void pyr_resize_tile(const Geom &G, int l, int block, int tiles_x, int tiles_per_img, uint32_t *box,
                                                const uint8_t *__restrict__ gray, uint8_t *__restrict__ fpyr,
                                                uint8_t *__restrict__ cpyr, const uint32_t *__restrict__ tab) {
    const int tid = threadIdx.x, rg = tid >> 4, g = tid & 15;
    const int img = block / tiles_per_img, tile = block - img * tiles_per_img;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    constexpr int mx = kFrame ? 4 : 0, my = kFrame ? 3 : 0;
    int lw, lh, os, sw, sh, ss, tX, tY, out_h;
    const uint8_t *S;
    uint8_t *D;
    if (kFrame) {
        lw = G.flw[l]; lh = G.flh[l]; os = G.fstride[l];
        sw = G.flw[l - 1]; sh = G.flh[l - 1]; ss = G.fstride[l - 1];
        tX = G.tXF[l]; tY = G.tYF[l];
        out_h = lh + 6;
        S = l == 1 ? gray + (size_t)img * G.gp * G.h : fpyr + (size_t)img * G.fframe + G.foff[l - 1] + 3 * ss + 4;
        D = fpyr + (size_t)img * G.fframe + G.foff[l];
    } else {
        lw = G.clw[l]; lh = G.clh[l]; os = G.cstride[l];
        sw = G.clw[l - 1]; sh = G.clh[l - 1];
        tX = G.tXC[l]; tY = G.tYC[l];
        out_h = lh;
        S = cell_level(G, gray, cpyr, img, l - 1, ss);
        D = cpyr + (size_t)img * G.cunit + G.coff[l];
    }
    const int out_w4 = os >> 2;
    // this lane: dword column gx of the stored rows gy0 .. gy0 + 3; image coordinates differ by the margin
    const int gx = tx * (kRTW / 4) + g, gy0 = ty * kRTH + kRPR * rg;
    const bool mine = gx < out_w4 && gy0 < out_h;
    // per-lane coefficients first: these loads and the box loads below are then in flight together
    uint32_t yt[kRPR], xt[4];
#pragma unroll
    for (int j = 0; j < kRPR; j++) yt[j] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) xt[k] = 0;
    if (mine) {
#pragma unroll
        for (int j = 0; j < kRPR; j++) {
            const int gy = min(gy0 + j, out_h - 1);
            yt[j] = tab[tY + (kFrame ? reflect101(gy - my, lh) : gy)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) xt[k] = tab[tX + (kFrame ? reflect101(4 * gx + k - mx, lw) : min(4 * gx + k, lw - 1))];
    }
    // source box of the whole tile (uniform): image x range of the tile's columns, y range of its rows
    int xlo, xhi, ylo, yhi;
    {
        const int xa = tx * kRTW - mx, xb = min(tx * kRTW + kRTW - 1, os - 1) - mx;
        const int ya = ty * kRTH - my, yb = min(ty * kRTH + kRTH - 1, out_h - 1) - my;
        if (kFrame) {
            reflect_range(xa, xb, lw, xlo, xhi);
            reflect_range(ya, yb, lh, ylo, yhi);
        } else {
            xlo = xa; xhi = min(xb, lw - 1);
            ylo = ya; yhi = yb;
        }
    }
    // The box without a table read in front of its loads: the tap offset of destination index d is
    // floor((d + 0.5) * src / dst - 0.5), which lies in [floor(d * src / dst), floor(d * src / dst) + 1] for src >= dst
    // (levels shrink); the margins absorb the second tap and the rounding of the float arithmetic (its error, below 0.01
    // at these sizes, moves a floor by at most one).
    const float fxs = (float)sw / (float)lw, fys = (float)sh / (float)lh;
    const int bx0 = max((int)((float)xlo * fxs) - 2, 0), bx1 = min((int)((float)xhi * fxs) + 4, sw - 1);
    const int by0 = max((int)((float)ylo * fys) - 2, 0), by1 = min((int)((float)yhi * fys) + 4, sh - 1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    const bool staged = bw <= kRBW - 4 && bh <= kRBH;   // (uniform) otherwise the taps come straight from global memory
    if (staged) {
        const int nd = (bw + 3) >> 2;   // <= 27 dwords per row
        const int c = tid & 31;
        if (c < nd)
            for (int r = tid >> 5; r < bh; r += 8) {
                uint32_t v;
                __builtin_memcpy(&v, S + (size_t)(by0 + r) * ss + bx0 + 4 * c, 4);
                box[r * (kRBW / 4) + c] = v;
            }
    }
    __syncthreads();
    if (!mine) return;
    // x side, once for the lane's rows: tap offsets relative to the row base, second-tap offsets, coefficients
    int xa_[4], xb_[4];
    uint32_t x0c[4], x1c[4];
    const int xbase = staged ? bx0 : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xo = (int)(xt[k] & 0xffffu);
        x1c[k] = xt[k] >> 16;
        x0c[k] = 256u - x1c[k];
        xa_[k] = xo - xbase;
        xb_[k] = min(xo + 1, sw - 1) - xbase;
    }
    const uint8_t *bb = reinterpret_cast<const uint8_t *>(box);
#pragma unroll
    for (int j = 0; j < kRPR; j++) {
        if (gy0 + j >= out_h) break;
        const int yo = (int)(yt[j] & 0xffffu), yb = min(yo + 1, sh - 1);
        const uint32_t y1c = yt[j] >> 16, y0c = 256u - y1c;
        uint32_t out = 0;
        if (staged) {   // two code paths, not one pointer: LDS reads stay LDS reads (and no LDS offset ever goes negative)
            const uint8_t *r0 = bb + (yo - by0) * kRBW, *r1 = bb + (yb - by0) * kRBW;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t h0 = (uint32_t)r0[xa_[k]] * x0c[k] + (uint32_t)r0[xb_[k]] * x1c[k];
                const uint32_t h1 = (uint32_t)r1[xa_[k]] * x0c[k] + (uint32_t)r1[xb_[k]] * x1c[k];
                out |= ((h0 * y0c + h1 * y1c + (1u << 15)) >> 16) << (8 * k);
            }
        } else {
            const uint8_t *r0 = S + (size_t)yo * ss, *r1 = S + (size_t)yb * ss;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t h0 = (uint32_t)r0[xa_[k]] * x0c[k] + (uint32_t)r0[xb_[k]] * x1c[k];
                const uint32_t h1 = (uint32_t)r1[xa_[k]] * x0c[k] + (uint32_t)r1[xb_[k]] * x1c[k];
                out |= ((h0 * y0c + h1 * y1c + (1u << 15)) >> 16) << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(D + (size_t)(gy0 + j) * os + 4 * gx) = out;
    }
}

// one launch per level: blocks [0, frame_blocks) make the frames' level, the rest the cells' (the two are independent)
__global__ __launch_bounds__(256) void pyr_resize_kernel(Geom G, int l, int frame_blocks, int txF, int nbF, int txC, int nbC,
                                                         const uint8_t *__restrict__ gray, uint8_t *__restrict__ fpyr,
                                                         uint8_t *__restrict__ cpyr, const uint32_t *__restrict__ tab) {
    __shared__ uint32_t box[kRBH * kRBW / 4];
    if ((int)blockIdx.x < frame_blocks) pyr_resize_tile<true>(G, l, blockIdx.x, txF, nbF, box, gray, fpyr, cpyr, tab);
    else pyr_resize_tile<false>(G, l, blockIdx.x - frame_blocks, txC, nbC, box, gray, fpyr, cpyr, tab);
}

// ------------------------------------------------------------------------------------------
// FAST-9/16 (fast.cpp FAST_t<16> + cornerScore<16>), 3x3 non-max suppression, both detectors' raster-ordered lists
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// M of the pixel at c (row stride s): max over the 16 arcs of 9 of min(v - p) (dark) and of min(p - v) (bright);
// values <= kFastMin are reported as 0 (no detector here can tell them apart)
__device__ __forceinline__ int fast_arc_score(const uint8_t *c, int s) {
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * s];      d[1] = v - c[3 * s + 1];  d[2] = v - c[2 * s + 2];   d[3] = v - c[s + 3];
    d[4] = v - c[3];          d[5] = v - c[-s + 3];     d[6] = v - c[-2 * s + 2];  d[7] = v - c[-3 * s + 1];
    d[8] = v - c[-3 * s];     d[9] = v - c[-3 * s - 1]; d[10] = v - c[-2 * s - 2]; d[11] = v - c[-s - 3];
    d[12] = v - c[-3];        d[13] = v - c[s - 3];     d[14] = v - c[2 * s - 2];  d[15] = v - c[3 * s - 1];
    // any arc of 9 holds at least two of the four compass pixels: cheap necessary test
    const int dark = (d[0] > kFastMin) + (d[4] > kFastMin) + (d[8] > kFastMin) + (d[12] > kFastMin);
    const int bright = (d[0] < -kFastMin) + (d[4] < -kFastMin) + (d[8] < -kFastMin) + (d[12] < -kFastMin);
    int best = 0;
    if (dark >= 2 || bright >= 2) {
        int a3[16], b3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
            b3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        }
        int dk = -256, br = 256;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            dk = max(dk, min3i(a3[k], a3[(k + 3) & 15], a3[(k + 6) & 15]));
            br = min(br, max3i(b3[k], b3[(k + 3) & 15], b3[(k + 6) & 15]));
        }
        best = max(dk, -br);
    }
    return best > kFastMin ? min(best, 255) : 0;
}

// Necessary condition for M > kFastMin from the eight even circle pixels alone: an arc of 9 holds at least four consecutive
// even positions, so a corner has four consecutive even pixels all darker than v - 5 or all brighter than v + 5.
__device__ __forceinline__ bool fast_may_be_corner(const uint8_t *c, int s) {
    const int v = c[0];
    int e[8];
    e[0] = v - c[3 * s];  e[1] = v - c[2 * s + 2];  e[2] = v - c[3];  e[3] = v - c[-2 * s + 2];
    e[4] = v - c[-3 * s]; e[5] = v - c[-2 * s - 2]; e[6] = v - c[-3]; e[7] = v - c[2 * s - 2];
    int lo2[8], hi2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo2[k] = min(e[k], e[(k + 1) & 7]);
        hi2[k] = max(e[k], e[(k + 1) & 7]);
    }
    int dk = -256, br = 256;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        dk = max(dk, min(lo2[k], lo2[(k + 2) & 7]));
        br = min(br, max(hi2[k], hi2[(k + 2) & 7]));
    }
    return dk > kFastMin || br < -kFastMin;
}

// exact i / d for i * d < 2^32 with m = floor(2^32 / d) + 1 (which is 2^32 for d = 1 and arrives here as 0)
__device__ __forceinline__ int div_magic(int i, uint32_t m) { return m ? (int)__umulhi((uint32_t)i, m) : i; }

struct FastLds {
    int R, SW, MW, CPR;   // strip rows, tile row strides, 64-pixel chunks per row
    uint8_t *S, *M;
    unsigned long long *mask5, *mask20;
    int *cnt5, *cnt20, *hist, *tot;
    uint16_t *cand;       // M-tile pixels that pass the cheap corner test, [ccap]
    int ccap;
};

// rows a strip may have with `bytes` of LDS (host and device agree through this one function)
__host__ __device__ inline int fast_strip_rows(int rw, int bytes) {
    const int SW = (rw + 8 + 3) & ~3, MW = (rw + 2 + 3) & ~3, CPR = (rw + 63) >> 6;
    const int fixed = 8 * SW + 2 * MW + 2 * MW + 256 * 4 + 64, per_row = SW + MW + MW + CPR * 24;   // S, M, candidates (u16 x half the M tile)
    int R = (bytes - fixed) / per_row;
    const int by_index = 65535 / MW - 2;   // candidates are 16-bit offsets into the M tile
    R = R > by_index ? by_index : R;
    return R > 32 ? 32 : R;
}

// One workgroup per (unit, level); blocks are level-major so the large levels start first.
// lists [unit][t][slot] hold (raster index in the inner region) << 8 | M; cnt0 [unit][t][8] their lengths;
// c1_20 [unit][8] = length of the threshold-20 list after the first retainBest (from the score histogram).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void fast_collect_kernel(Geom G, int units, int lds_bytes, const uint8_t *__restrict__ gray,
                                                           const uint8_t *__restrict__ cpyr, uint32_t *__restrict__ lists,
                                                           int32_t *__restrict__ cnt0, int32_t *__restrict__ c1_20) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int l = blockIdx.x / units, u = blockIdx.x - l * units, tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (in a scalar register: the chunk loops are per wave)
    const int rw = G.rw[l], rh = G.rh[l], cap = G.cap[l];
    int stride;
    const uint8_t *src = cell_level(G, gray, cpyr, u, l, stride);
    FastLds L;
    L.R = fast_strip_rows(rw, lds_bytes);
    L.SW = (rw + 8 + 3) & ~3;
    L.MW = (rw + 2 + 3) & ~3;
    L.CPR = (rw + 63) >> 6;
    {
        unsigned char *p = smem;
        L.hist = reinterpret_cast<int *>(p);                       p += 256 * 4;
        L.tot = reinterpret_cast<int *>(p);                        p += 64;
        const int nc = L.R * L.CPR;
        L.mask5 = reinterpret_cast<unsigned long long *>(p);       p += (size_t)nc * 8;
        L.mask20 = reinterpret_cast<unsigned long long *>(p);      p += (size_t)nc * 8;
        L.cnt5 = reinterpret_cast<int *>(p);                       p += (size_t)nc * 4;
        L.cnt20 = reinterpret_cast<int *>(p);                      p += (size_t)nc * 4;
        L.S = p;                                                   p += (size_t)(L.R + 8) * L.SW;
        L.M = p;                                                   p += (size_t)(L.R + 2) * L.MW;
        L.cand = reinterpret_cast<uint16_t *>(p);
        L.ccap = ((L.R + 2) * L.MW) >> 1;
    }
    L.hist[tid] = 0;
    if (tid < 3) L.tot[tid] = 0;   // running list lengths (5, 20), candidate count of the strip
    const uint32_t mS = 0xffffffffu / (uint32_t)(L.SW >> 2) + 1u, mM = 0xffffffffu / (uint32_t)(rw + 2) + 1u, mMW = 0xffffffffu / (uint32_t)L.MW + 1u;
    const uint32_t mC = 0xffffffffu / (uint32_t)L.CPR + 1u;
    uint32_t *list20 = lists + ((size_t)(u * 2 + 0)) * G.loff[G.nlv] + G.loff[l];
    uint32_t *list5 = lists + ((size_t)(u * 2 + 1)) * G.loff[G.nlv] + G.loff[l];
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (int y0 = 0; y0 < rh; y0 += L.R) {   // strip = inner rows [y0, y0 + rows)
        const int rows = min(L.R, rh - y0);
        for (int id = tid; id < rows * L.CPR; id += 256) {   // the strip's chunk masks start empty (candidates or them in)
            L.mask5[id] = 0ull;
            L.mask20[id] = 0ull;
        }
        // 1. source tile: level rows 31 + y0 - 4 .. (rows + 8 of them), columns 27 .. lw - 28 (rw + 8), as (unaligned) dwords,
        //    eight loads in flight per lane (a dword may reach up to 3 bytes past the tile's last column: still inside the level's row)
        {
            const uint8_t *s0 = src + (size_t)(kEdge + y0 - 4) * stride + (kEdge - 4);
            const int dpr = L.SW >> 2, n = (rows + 8) * dpr;
            uint32_t *S4 = reinterpret_cast<uint32_t *>(L.S);
            for (int i0 = tid; i0 < n; i0 += 256 * 8) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + 256 * k;
                    if (i < n) {
                        const int r = div_magic(i, mS), c = i - r * dpr;
                        __builtin_memcpy(&v[k], s0 + (size_t)r * stride + 4 * c, 4);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + 256 * k;
                    if (i < n) S4[i] = v[k];
                }
            }
        }
        __syncthreads();
        // 2. M for inner rows y0 - 1 .. y0 + rows and inner columns -1 .. rw.  First a cheap necessary test on every pixel (a few
        //    per cent pass); the passing pixels are compacted so that the arc scores are computed by full waves.  A lane walks
        //    the tile 256 pixels at a time; its (row, column) and the two tile offsets advance without a division.
        {
            const int W2 = rw + 2, n = (rows + 2) * W2;
            const int q256 = 256 / W2, r256 = 256 - q256 * W2;
            const int stepS = q256 * L.SW + r256, stepM = q256 * L.MW + r256, wrapS = L.SW - W2, wrapM = L.MW - W2;
            int c = tid - div_magic(tid, mM) * W2;
            int aM = div_magic(tid, mM) * L.MW + c, aS = (div_magic(tid, mM) + 3) * L.SW + c + 3;
            for (int i0 = 0; i0 < n; i0 += 256) {
                bool cnd = false;
                if (i0 + tid < n) {
                    cnd = fast_may_be_corner(L.S + aS, L.SW);
                    L.M[aM] = 0;
                }
                const unsigned long long bal = __ballot(cnd);
                if (bal) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&L.tot[2], (int)__popcll(bal));
                    base = __shfl(base, 0, 64);
                    if (cnd) {
                        const int pos = base + (int)__popcll(bal & lt);
                        if (pos < L.ccap) L.cand[pos] = (uint16_t)aM;
                        else L.M[aM] = (uint8_t)fast_arc_score(L.S + aS, L.SW);   // list full: in place
                    }
                }
                c += r256;
                aS += stepS;
                aM += stepM;
                if (c >= W2) {
                    c -= W2;
                    aS += wrapS;
                    aM += wrapM;
                }
            }
            __syncthreads();
            const int nc2 = min(L.tot[2], L.ccap);
            for (int j = tid; j < nc2; j += 256) {
                const int am = L.cand[j];
                const int r = div_magic(am, mMW), cc = am - r * L.MW;
                L.M[am] = (uint8_t)fast_arc_score(L.S + (r + 3) * L.SW + cc + 3, L.SW);
            }
        }
        __syncthreads();
        // 3.-5. non-max suppression, per-chunk counts, ordered writes.  Every pixel with M > 5 is on the candidate list (unless
        //    the list overflowed), so all three are driven by the list -- a few per cent of the pixels -- instead of walking every
        //    64-pixel chunk: a surviving candidate sets its bit in its chunk's mask (LDS atomic or), the counts are the masks'
        //    popcounts, and a survivor's slot is its chunk's prefix plus the rank of its bit: raster order whatever order the
        //    candidates were listed in.
        const int nc = rows * L.CPR;
        const bool listed = L.tot[2] <= L.ccap;   // (uniform; read after the barrier that closed the list)
        if (listed) {
            const int nc2 = L.tot[2];
            for (int j = tid; j < nc2; j += 256) {
                const int am = L.cand[j];
                const int r = div_magic(am, mMW), cc = am - r * L.MW;
                if (r < 1 || r > rows || cc < 1 || cc > rw) continue;   // the ring only serves as neighbours
                const uint8_t *m = L.M + am;
                const int v = m[0];
                if (v <= kFastMin) continue;
                const int nb = max(max(max3i(m[-L.MW - 1], m[-L.MW], m[-L.MW + 1]), max3i(m[L.MW - 1], m[L.MW], m[L.MW + 1])),
                                   max((int)m[-1], (int)m[1]));
                if (v > nb) {
                    const int id = (r - 1) * L.CPR + ((cc - 1) >> 6);
                    const unsigned long long bit = 1ull << ((cc - 1) & 63);
                    atomicOr(&L.mask5[id], bit);
                    if (v > kThr[0]) {
                        atomicOr(&L.mask20[id], bit);
                        atomicAdd(&L.hist[v], 1);
                    }
                }
            }
        } else {
            for (int id = wave; id < nc; id += 4) {
                const int r = div_magic(id, mC), c = id - r * L.CPR;
                const int x = c * 64 + lane;
                const uint8_t *m = L.M + (r + 1) * L.MW + x + 1;
                const int v = x < rw ? (int)m[0] : 0;
                bool keep = false;
                if (v > kFastMin) {
                    const int nb = max(max(max3i(m[-L.MW - 1], m[-L.MW], m[-L.MW + 1]), max3i(m[L.MW - 1], m[L.MW], m[L.MW + 1])),
                                       max((int)m[-1], (int)m[1]));
                    keep = v > nb;
                }
                const bool keep20 = keep && v > kThr[0];
                const unsigned long long b5 = __ballot(keep), b20 = __ballot(keep20);
                if (keep20) atomicAdd(&L.hist[v], 1);
                if (lane == 0) {
                    L.mask5[id] = b5;
                    L.mask20[id] = b20;
                }
            }
        }
        __syncthreads();
        // exclusive scan of the chunk counts (wave 0: threshold 5, wave 1: threshold 20), running totals carried in LDS
        if (wave < 2) {
            int *cnt = wave == 0 ? L.cnt5 : L.cnt20;
            const unsigned long long *msk = wave == 0 ? L.mask5 : L.mask20;
            int base = L.tot[wave];
            for (int j0 = 0; j0 < nc; j0 += 64) {
                const int j = j0 + lane;
                const int v = j < nc ? (int)__popcll(msk[j]) : 0;
                int inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += t;
                }
                if (j < nc) cnt[j] = base + inc - v;
                base += __shfl(inc, 63, 64);
            }
            if (lane == 0) L.tot[wave] = base;
        }
        __syncthreads();
        // ordered writes
        if (listed) {
            const int nc2 = L.tot[2];
            for (int j = tid; j < nc2; j += 256) {
                const int am = L.cand[j];
                const int r = div_magic(am, mMW), cc = am - r * L.MW;
                if (r < 1 || r > rows || cc < 1 || cc > rw) continue;
                const int id = (r - 1) * L.CPR + ((cc - 1) >> 6), bitpos = (cc - 1) & 63;
                const unsigned long long b5 = L.mask5[id];
                if ((b5 >> bitpos) & 1ull) {
                    const unsigned long long below = (1ull << bitpos) - 1ull;
                    const uint32_t e = ((uint32_t)((y0 + r - 1) * rw + (cc - 1)) << 8) | (uint32_t)L.M[am];
                    const int p5 = L.cnt5[id] + (int)__popcll(b5 & below);
                    if (p5 < cap) list5[p5] = e;
                    const unsigned long long b20 = L.mask20[id];
                    if ((b20 >> bitpos) & 1ull) {
                        const int p20 = L.cnt20[id] + (int)__popcll(b20 & below);
                        if (p20 < cap) list20[p20] = e;
                    }
                }
            }
        } else {
            for (int id = wave; id < nc; id += 4) {
                const int r = div_magic(id, mC), c = id - r * L.CPR;
                const int x = c * 64 + lane;
                const unsigned long long b5 = L.mask5[id], b20 = L.mask20[id];
                if ((b5 >> lane) & 1ull) {
                    const uint32_t e = ((uint32_t)((y0 + r) * rw + x) << 8) | (uint32_t)L.M[(r + 1) * L.MW + x + 1];
                    const int p5 = L.cnt5[id] + (int)__popcll(b5 & lt);
                    if (p5 < cap) list5[p5] = e;
                    if ((b20 >> lane) & 1ull) {
                        const int p20 = L.cnt20[id] + (int)__popcll(b20 & lt);
                        if (p20 < cap) list20[p20] = e;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) L.tot[2] = 0;   // the next strip's candidate count
    }
    __syncthreads();
    if (wave == 0) {
        const int n5 = min(L.tot[0], cap), n20 = min(L.tot[1], cap);
        // length after retainBest(list20, 2 * quota): all of it if it is no longer than that, otherwise the entries
        // whose score is >= the (2 * quota)-th largest one
        const int np = 2 * G.per_level[l];
        const int4 hv = reinterpret_cast<const int4 *>(L.hist)[lane];   // scores 4 * lane .. 4 * lane + 3
        const int s = hv.x + hv.y + hv.z + hv.w;
        int suf = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += t;
        }
        int c1 = n20;
        if (n20 > np && np > 0) {
            const unsigned long long bal = __ballot(suf >= np);   // lane 0 always (suf = n20 > np)
            const int Lm = 63 - (int)__builtin_clzll(bal);
            int acc = suf - s;   // entries with scores above this lane's four
            int got = 0;
            if (lane == Lm) {
                acc += hv.w; if (acc >= np) got = acc;
                if (!got) { acc += hv.z; if (acc >= np) got = acc; }
                if (!got) { acc += hv.y; if (acc >= np) got = acc; }
                if (!got) { acc += hv.x; got = acc; }
            }
            c1 = __shfl(got, Lm, 64);
        } else if (np == 0) {
            c1 = 0;
        }
        if (lane == 0) {
            cnt0[(size_t)(u * 2 + 0) * kMaxLevels + l] = n20;
            cnt0[(size_t)(u * 2 + 1) * kMaxLevels + l] = n5;
            c1_20[(size_t)u * kMaxLevels + l] = c1;
        }
    }
}

// which detectors a cell has to run: bit 0 = threshold 20, bit 1 = threshold 5 (see the head of this file)
__global__ __launch_bounds__(256) void grid_decide_kernel(Geom G, int units, const int32_t *__restrict__ c1_20,
                                                          uint8_t *__restrict__ flags) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    int lb = 0, ub = 0;
    for (int l = 0; l < G.nlv; l++) {
        const int c = c1_20[(size_t)u * kMaxLevels + l];
        lb += min(c, G.per_level[l]);
        ub += c;
    }
    flags[u] = (uint8_t)((ub >= kNFeatures ? 1 : 0) | (lb < kNFeatures ? 2 : 0));
}

// ------------------------------------------------------------------------------------------
// KeyPointsFilter::retainBest(list, n_points) (keypoint.cpp): nth_element by response descending, then
// std::partition of the tail on response >= boundary.  One wave per list.
// ------------------------------------------------------------------------------------------
struct RespStore {   // second retainBest: Harris response + the list entry it belongs to
    using value_type = int2;   // (response bits, payload)
    using key_type = float;
    static constexpr int kBytes = 8;
    float *key_;
    uint32_t *pay_;
    __device__ void bind(unsigned char *base, int entries) {
        key_ = reinterpret_cast<float *>(base);
        pay_ = reinterpret_cast<uint32_t *>(base + (size_t)entries * 4);
    }
    __device__ void load(int i, uint32_t e, float r) {
        key_[i] = r;
        pay_[i] = e;
    }
    __device__ uint32_t entry(int i) const { return pay_[i]; }
    __device__ int2 get(int i) const { return make_int2(__float_as_int(key_[i]), (int)pay_[i]); }
    __device__ void set(int i, const int2 &v) {
        key_[i] = __int_as_float(v.x);
        pay_[i] = (uint32_t)v.y;
    }
    __device__ void swap(int i, int j) {
        const int2 a = get(i), b = get(j);
        set(i, b);
        set(j, a);
    }
    __device__ float key(int i) const { return key_[i]; }
    __device__ float key_of(const int2 &v) const { return __int_as_float(v.x); }
    __device__ bool less(float a, float b) const { return a > b; }   // KeypointResponseGreater
};

struct ScoreStore {   // first retainBest: the FAST score is the low byte of the entry itself
    using value_type = uint32_t;
    using key_type = int;
    static constexpr int kBytes = 4;
    uint32_t *e_;
    __device__ void bind(unsigned char *base, int) { e_ = reinterpret_cast<uint32_t *>(base); }
    __device__ void load(int i, uint32_t e, float) { e_[i] = e; }
    __device__ uint32_t entry(int i) const { return e_[i]; }
    __device__ uint32_t get(int i) const { return e_[i]; }
    __device__ void set(int i, const uint32_t &v) { e_[i] = v; }
    __device__ void swap(int i, int j) {
        const uint32_t a = e_[i], b = e_[j];
        e_[i] = b;
        e_[j] = a;
    }
    __device__ int key(int i) const { return (int)(e_[i] & 255u); }
    __device__ int key_of(const uint32_t &v) const { return (int)(v & 255u); }
    __device__ bool less(int a, int b) const { return a > b; }
};

// std::partition(first = a, last = b, key >= amb), bidirectional form: with L_k the k-th position from the left that fails
// the predicate and R_k the k-th from the right that meets it, it swaps (L_k, R_k) while L_k < R_k and returns
// a + (number that meet it).  Same stopper-list scheme as vs_sel::wave_partition.
template <class S, class I>
__device__ int wave_std_partition(S &s, int a, int b, typename S::key_type amb, I *sl, I *sr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nL = 0, nR = 0;
    for (int p0 = a; p0 < b; p0 += 64) {
        const int p = p0 + lane;
        const bool st = p < b && !(s.key(p) >= amb);
        const unsigned long long bal = __ballot(st);
        if (st) sl[a + nL + (int)__popcll(bal & lt)] = (I)p;
        nL += (int)__popcll(bal);
    }
    for (int p0 = b - 1; p0 >= a; p0 -= 64) {
        const int p = p0 - lane;
        const bool st = p >= a && s.key(p) >= amb;
        const unsigned long long bal = __ballot(st);
        if (st) sr[a + nR + (int)__popcll(bal & lt)] = (I)p;
        nR += (int)__popcll(bal);
    }
    vs_sel::wave_sync_lds();
    const int m = nL < nR ? nL : nR;
    for (int k0 = 0; k0 < m; k0 += 64) {
        const int k = k0 + lane;
        const bool sw = k < m && (int)sl[a + k] < (int)sr[a + k];
        if (sw) s.swap((int)sl[a + k], (int)sr[a + k]);
        if (__ballot(sw) != ~0ull) break;
    }
    vs_sel::wave_sync_lds();
    return a + nR;
}

// S = ScoreStore: keys = FAST score, n_points = 2 * quota, cnt_in = cnt0, cnt_out = cnt1;  S = RespStore: keys = Harris
// response, n_points = quota, cnt_in = cnt1, cnt_out = cnt2.  Survivors are written back to the front of the list in libstdc++'s
// order.  Tiers by list length n: kLo < n + 1 <= kEntries entries of LDS per wave (kGlobal: per-list global scratch, any
// length); lists that retainBest leaves alone (n <= n_points) are settled by the tier with kLo == 0.
template <class S, int kEntries, int kLo, int kWaves, bool kGlobal>
__global__ __launch_bounds__(64 * kWaves) void retain_best_kernel(Geom G, int n_lists, uint32_t *__restrict__ ent,
                                                                  const float *__restrict__ resp,
                                                                  const int32_t *__restrict__ cnt_in, int32_t *__restrict__ cnt_out,
                                                                  const uint8_t *__restrict__ flags, unsigned char *__restrict__ g_store,
                                                                  int *__restrict__ g_sl, int *__restrict__ g_sr) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool kScore = S::kBytes == 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = blockIdx.x * kWaves + wave;
    if (li >= n_lists) return;
    const int l = li % G.nlv, ut = li / G.nlv, u = ut >> 1, t = ut & 1;
    if (!(flags[u] & (1 << t))) return;
    const size_t ci = (size_t)ut * kMaxLevels + l;
    const int n = cnt_in[ci];
    const int n_points = (kScore ? 2 : 1) * G.per_level[l];
    if (n <= n_points) {
        if (kLo == 0 && lane == 0) cnt_out[ci] = n;
        return;
    }
    if (n_points == 0) {
        if (kLo == 0 && lane == 0) cnt_out[ci] = 0;
        return;
    }
    if (n + 1 <= kLo || (!kGlobal && n + 1 > kEntries)) return;
    const size_t slot = (size_t)ut * G.loff[G.nlv] + G.loff[l];
    S s;
    int m;
    if constexpr (kGlobal) {
        s.bind(g_store + slot * 8, G.cap[l]);   // 8 bytes per slot entry hold either store (cap[l] entries of this list's slot)
        int *sl = g_sl + slot, *sr = g_sr + slot + ci;   // one spare element per list
        for (int i = lane; i < n; i += 64) s.load(i, ent[slot + i], kScore ? 0.f : resp[slot + i]);
        __threadfence_block();
        vs_sel::wave_sync_lds();
        vs_sel::wave_nth_element(s, 0, n_points - 1, n, sl, sr);
        __threadfence_block();
        m = wave_std_partition(s, n_points, n, s.key(n_points - 1), sl, sr);
        __threadfence_block();
    } else {
        unsigned char *base = smem + (size_t)wave * kEntries * (S::kBytes + 4);
        s.bind(base, kEntries);
        uint16_t *sl = reinterpret_cast<uint16_t *>(base + (size_t)kEntries * S::kBytes);
        uint16_t *sr = sl + kEntries;
        for (int i = lane; i < n; i += 64) s.load(i, ent[slot + i], kScore ? 0.f : resp[slot + i]);
        vs_sel::wave_sync_lds();
        vs_sel::wave_nth_element(s, 0, n_points - 1, n, sl, sr);
        m = wave_std_partition(s, n_points, n, s.key(n_points - 1), sl, sr);
    }
    for (int i = lane; i < m; i += 64) ent[slot + i] = s.entry(i);
    if (lane == 0) cnt_out[ci] = m;
}

// HarrisResponses(pyramid, keypoints, 7, 0.04) for the lists the second retainBest will cut (n > quota).  A lane per
// keypoint; its 9x9 window is fetched as 27 (unaligned) dwords before any arithmetic so that the loads are in flight together.
__global__ __launch_bounds__(64) void harris_kernel(Geom G, const uint8_t *__restrict__ gray, const uint8_t *__restrict__ cpyr,
                                                    const uint32_t *__restrict__ ent, float *__restrict__ resp,
                                                    const int32_t *__restrict__ cnt1, const uint8_t *__restrict__ flags) {
    const int li = blockIdx.x;
    const int l = li % G.nlv, ut = li / G.nlv, u = ut >> 1, t = ut & 1;
    if (!(flags[u] & (1 << t))) return;
    const int n = cnt1[(size_t)ut * kMaxLevels + l];
    if (n <= G.per_level[l]) return;
    const size_t slot = (size_t)ut * G.loff[G.nlv] + G.loff[l];
    int step;
    const uint8_t *img = cell_level(G, gray, cpyr, u, l, step);
    const int rw = G.rw[l];
    for (int i = threadIdx.x; i < n; i += 64) {
        const int idx = (int)(ent[slot + i] >> 8);
        const int py = idx / rw, px = idx - py * rw;
        const uint8_t *p0 = img + (size_t)(py + kEdge - 4) * step + px + kEdge - 4;   // window rows / columns -4 .. 4
        uint32_t wv[9][3];
#pragma unroll
        for (int r = 0; r < 9; r++) __builtin_memcpy(wv[r], p0 + (size_t)r * step, 12);
        int a = 0, b = 0, c = 0;
#pragma unroll
        for (int yy = 1; yy <= 7; yy++)
#pragma unroll
            for (int xx = 1; xx <= 7; xx++) {
#define VS_W(R, C) ((int)((wv[R][(C) >> 2] >> (8 * ((C) & 3))) & 255u))
                const int Ix = (VS_W(yy, xx + 1) - VS_W(yy, xx - 1)) * 2 + (VS_W(yy - 1, xx + 1) - VS_W(yy - 1, xx - 1)) +
                               (VS_W(yy + 1, xx + 1) - VS_W(yy + 1, xx - 1));
                const int Iy = (VS_W(yy + 1, xx) - VS_W(yy - 1, xx)) * 2 + (VS_W(yy + 1, xx - 1) - VS_W(yy - 1, xx - 1)) +
                               (VS_W(yy + 1, xx + 1) - VS_W(yy - 1, xx + 1));
#undef VS_W
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        const float scale = 1.f / ((1 << 2) * 7 * 255.f);
        const float scale_sq_sq = scale * scale * scale * scale;
        const float fa = (float)a, fb = (float)b, fc = (float)c;
        const float t1 = fa * fb, t2 = fc * fc, sum = fa + fb;
        const float t3 = 0.04f * sum;
        const float t4 = t3 * sum;
        resp[slot + i] = ((t1 - t2) - t4) * scale_sq_sq;
    }
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {   // cv::fastAtan2
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct UmaxTable {
    int v[17];
};

// ------------------------------------------------------------------------------------------
// frame assembly: choose the detector per cell (:34-36), "pt *= scale", shift (:37-40), runByImageBorder(31) on the
// frame and ORB::compute's regrouping by level (stable: cell order inside a level).  One workgroup per frame; the
// lists of a frame in output order = (level, cell).  Per keypoint: (x, y, raster index in the cell level's inner region,
// cell << 4 | level); the angle is the descriptor kernel's job.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_assemble_kernel(Geom G, const uint32_t *__restrict__ ent,
                                                            const int32_t *__restrict__ cnt2, const uint8_t *__restrict__ flags,
                                                            int kp_cap, float4 *__restrict__ out_kp, int32_t *__restrict__ out_n) {
    extern __shared__ __align__(16) unsigned char smem[];
    int *kept = reinterpret_cast<int *>(smem);              // [nlv * cells] kept keypoints, then their exclusive prefix
    uint8_t *sel = reinterpret_cast<uint8_t *>(kept + G.nlv * G.cells);   // [cells] detector chosen
    __shared__ int s_total;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = G.nlv * G.cells;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c = tid; c < G.cells; c += 256) {
        const int u = f * G.cells + c;
        const int fl = flags[u];
        int t = 1;
        if (!(fl & 2)) {
            t = 0;                                  // the threshold-20 detector is certain to reach 500
        } else if (fl & 1) {
            int tot20 = 0;
            for (int l = 0; l < G.nlv; l++) tot20 += cnt2[(size_t)(u * 2) * kMaxLevels + l];
            t = tot20 >= kNFeatures ? 0 : 1;        // `if (temp.size() < nfeatures)` -> the fallback's result replaces it
        }
        sel[c] = (uint8_t)t;
    }
    __syncthreads();
    const float wlim = (float)(G.w - kEdge), hlim = (float)(G.h - kEdge);
    const bool none = G.h <= 2 * kEdge || G.w <= 2 * kEdge;
    for (int pass = 0; pass < 2; pass++) {
        for (int q = wave; q < nl; q += 4) {
            const int lvl = q / G.cells, c = q - lvl * G.cells;
            const int ut = (f * G.cells + c) * 2 + sel[c];
            const int n = cnt2[(size_t)ut * kMaxLevels + lvl];
            const size_t slot = (size_t)ut * G.loff[G.nlv] + G.loff[lvl];
            const int ci = c / G.nrows, cj = c - ci * G.nrows;
            const float sx = (float)(ci * G.cw), sy = (float)(cj * G.ch), sc = G.scale[lvl];
            const int rw = G.rw[lvl];
            int base = pass ? kept[q] : 0, total = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                bool keep = false;
                float4 k4 = make_float4(0, 0, 0, 0);
                if (i < n) {
                    const int idx = (int)(ent[slot + i] >> 8);
                    const int py = idx / rw, px = idx - py * rw;
                    const float lx = (float)(px + kEdge) * sc, ly = (float)(py + kEdge) * sc;
                    // (x, y) in the frame, where the keypoint sits in its cell's level (the descriptor kernel finds the
                    // intensity centroid there), the level
                    k4 = make_float4(sx + lx, sy + ly, __int_as_float(idx), __int_as_float((c << 4) | lvl));
                    keep = !none && k4.x >= (float)kEdge && k4.x < wlim && k4.y >= (float)kEdge && k4.y < hlim;
                }
                const unsigned long long bal = __ballot(keep);
                if (pass && keep) {
                    const int o = base + total + (int)__popcll(bal & lt);
                    if (o < kp_cap) out_kp[(size_t)f * kp_cap + o] = k4;
                }
                total += (int)__popcll(bal);
            }
            if (!pass && lane == 0) kept[q] = total;
        }
        __syncthreads();
        if (!pass) {
            if (wave == 0) {
                int base = 0;
                for (int j0 = 0; j0 < nl; j0 += 64) {
                    const int jj = j0 + lane;
                    const int v = jj < nl ? kept[jj] : 0;
                    int inc = v;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int tt = __shfl_up(inc, o, 64);
                        if (lane >= o) inc += tt;
                    }
                    if (jj < nl) kept[jj] = base + inc - v;
                    base += __shfl(inc, 63, 64);
                }
                if (lane == 0) s_total = base;
            }
            __syncthreads();
        }
    }
    if (tid == 0) out_n[f] = min(s_total, kp_cap);
}

// ------------------------------------------------------------------------------------------
// ORB::compute: GaussianBlur 7x7 sigma 2 per level (blur.hip's streaming kernel), intensity centroid, steered BRIEF
// ------------------------------------------------------------------------------------------
// pinned sin/cos of an angle in degrees (mirrors vso::sincos_deg_pinned operation for operation)
__device__ __forceinline__ void sincos_deg_pinned(float angle_deg, float &s_out, float &c_out) {
    const float ar = angle_deg * (float)(3.14159265358979323846 / 180.f);
    const double x = (double)ar;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const double kf = rint(x * two_over_pi);
    const int k = (int)kf;
    const double r = (x - kf * pio2_hi) - kf * pio2_lo;
    const double r2 = r * r;
    const double sp = r * (1.0 + r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800))))));
    const double cp = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600))))));
    double s, c;
    switch (k & 3) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    s_out = (float)s;
    c_out = (float)c;
}

// the descriptor kernel's LDS patch: sampling discs of up to this radius (ORB's table: 20), row pitch, bytes per half wave
constexpr int kPatchR = 24, kPatchPitch = 56, kPatchBytes = (2 * kPatchR + 1) * kPatchPitch;

// largest distance of a pattern point from the patch centre, rounded up, + 1: no rotated and rounded sample lies farther out
__global__ __launch_bounds__(512) void pattern_radius_kernel(const int8_t *__restrict__ pattern, int32_t *__restrict__ radius) {
    __shared__ int s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const int x = pattern[2 * threadIdx.x], y = pattern[2 * threadIdx.x + 1];
    atomicMax(&s_max, x * x + y * y);
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = 0;
        while (r * r < s_max) r++;
        *radius = r + 1;
    }
}

// ICAngles + computeOrbDescriptors: half a wave per keypoint.
// Angle: lane j of the half holds column j - 15 of the 31 x 31 patch around the keypoint in its CELL's pyramid level (rows are
// contiguous bytes across the lanes); the two integer moments are summed over the half by shuffles.
// Descriptor: a lane per byte on the FRAME's pyramid level.  A sample inside the level's image reads the blurred level;
// outside it the unblurred reflect value (what OpenCV's in-place blur leaves in the level's border).  A keypoint whose
// whole sampling disc (pattern radius) lies inside the level -- nearly all -- takes a path without per-sample tests.
// `images` = [gray frames | cell pyramids], `blurred` = [blurred frames | blurred frame pyramids]: one base each and 32-bit
// offsets (gbytes = size of the first part).
__global__ __launch_bounds__(256) void orb_desc_kernel(Geom G, UmaxTable U, uint32_t gbytes, const uint8_t *__restrict__ images,
                                                       const uint8_t *__restrict__ blurred, const uint8_t *__restrict__ fpyr,
                                                       const float4 *__restrict__ kps, const int32_t *__restrict__ n_arr,
                                                       int kp_cap, const int8_t *__restrict__ pattern,
                                                       const int32_t *__restrict__ pat_radius, uint8_t *__restrict__ desc,
                                                       float *__restrict__ out_xy, float *__restrict__ out_angle_octave, int frames,
                                                       int per_frame) {
    // all workgroups of a frame on one XCD: its gray image, blurred image and pyramids (a few MB) then stay in that XCD's L2
    // while the frame's keypoints gather from them (dealt round-robin, every XCD would stream every frame: 3.4 KB of HBM
    // traffic per keypoint measured)
    __shared__ __align__(16) uint8_t s_patch[8 * kPatchBytes];
    int f, blk;
    vs_xcd_item_block(blockIdx.x, per_frame, f, blk);
    if (f >= frames) return;
    const int tid = threadIdx.x;
    const int kp = blk * 8 + (tid >> 5), byte = tid & 31;
    // the lane's eight pattern entries ((x1, y1, x2, y2) as int8 x 4 per bit), the frame's count and the keypoint record are
    // independent loads: all in flight before the first use (the record of a slot past the count is never used)
    const int4 pq0 = reinterpret_cast<const int4 *>(pattern)[byte * 2], pq1 = reinterpret_cast<const int4 *>(pattern)[byte * 2 + 1];
    const int pat8[8] = {pq0.x, pq0.y, pq0.z, pq0.w, pq1.x, pq1.y, pq1.z, pq1.w};
    const int n_f = n_arr[f];
    const int R = *pat_radius;
    const float4 k4 = kps[(size_t)f * kp_cap + (kp < kp_cap ? kp : 0)];
    if (kp >= n_f) return;   // whole halves leave; the shuffles below stay inside a half
    const int idx = __float_as_int(k4.z), cl = __float_as_int(k4.w);
    const int l = cl & 15, c = cl >> 4;
    // per-level constants, selected without indexing the argument struct by a per-lane value
    float lscale = G.scale[0];
    int lw = G.flw[0], lh = G.flh[0], fs = G.fstride[0], fo = 0, rw = G.rw[0], cs = G.cstride[0], co = 0;
#pragma unroll
    for (int q = 1; q < kMaxLevels; q++)
        if (l == q) {
            lscale = G.scale[q];
            lw = G.flw[q];
            lh = G.flh[q];
            fs = G.fstride[q];
            fo = G.foff[q];
            rw = G.rw[q];
            cs = G.cstride[q];
            co = G.coff[q];
        }
    float angle;
    {
        uint32_t cbase;   // offset of the cell level's pixel (0, 0) in `images`
        if (l == 0) {
            const int ci = c / G.nrows, cj = c - ci * G.nrows;
            cbase = (uint32_t)f * (uint32_t)(G.gp * G.h) + (uint32_t)(cj * G.ch) * (uint32_t)G.gp + (uint32_t)(ci * G.cw);
        } else {
            cbase = gbytes + (uint32_t)(f * G.cells + c) * (uint32_t)G.cunit + (uint32_t)co;
        }
        const int py = idx / rw, px = idx - py * rw;
        const int uo = byte - 15, auo = uo < 0 ? -uo : uo;
        const uint32_t center = cbase + (uint32_t)(py + kEdge) * (uint32_t)cs + (uint32_t)(px + kEdge + uo);
        // every lane fetches its column of all 31 rows (lane 31's column 16 is still >= 15 px inside the level), unconditionally
        // so that the loads are in flight together; the circular mask is applied to the values
        int val[31];
#pragma unroll
        for (int v = -15; v <= 15; v++) val[v + 15] = images[center + (uint32_t)(v * cs)];
        int m_10 = 0, m_01 = 0;
#pragma unroll
        for (int v = -15; v <= 15; v++) {
            const int x = auo <= U.v[v < 0 ? -v : v] ? val[v + 15] : 0;
            m_10 += uo * x;
            m_01 += v * x;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            m_10 += __shfl_xor(m_10, o, 64);
            m_01 += __shfl_xor(m_01, o, 64);
        }
        angle = fast_atan2_deg((float)m_01, (float)m_10);
    }
    const float scale = 1.f / lscale;
    float a, b;
    sincos_deg_pinned(angle, b, a);
    const int cx = (int)rintf(k4.x * scale), cy = (int)rintf(k4.y * scale);
    // offset of the frame level's pixel (0, 0): in `blurred`, and (levels >= 1) of the unblurred level in fpyr
    const uint32_t lbase = l == 0 ? (uint32_t)f * (uint32_t)(G.gp * G.h) : gbytes + (uint32_t)f * (uint32_t)G.fframe + (uint32_t)(fo + 3 * fs + 4);
    uint32_t val = 0;
    const bool inside = cx - R >= 0 && cx + R < lw && cy - R >= 0 && cy + R < lh;
    if (inside && R <= kPatchR) {
        // The sampling disc as a (2R + 1)^2 patch in LDS, one per half wave: the half fetches it as (unaligned) dwords row by row --
        // a dozen loads per lane, consecutive lanes on consecutive addresses -- and the 16 samples of a lane are LDS byte reads;
        // the same 16 as per-lane byte gathers from global memory cost the wave 16 address cycles each.
        const int half = (tid >> 5) & 7;
        uint8_t *patch = s_patch + half * kPatchBytes;
        const int side = 2 * R + 1, dpr = (side + 3) >> 2;          // dwords per patch row
        const uint32_t org = lbase + (uint32_t)(cy - R) * (uint32_t)fs + (uint32_t)(cx - R);
        const uint32_t mD = 0xffffffffu / (uint32_t)dpr + 1u;
        const int nd = side * dpr;
        for (int i = byte; i < nd; i += 32) {
            const int r = (int)__umulhi((uint32_t)i, mD), cdw = i - r * dpr;
            uint32_t v;
            __builtin_memcpy(&v, blurred + org + (uint32_t)(r * fs + 4 * cdw), 4);
            reinterpret_cast<uint32_t *>(patch)[r * (kPatchPitch / 4) + cdw] = v;
        }
        vs_sel::wave_sync_lds();
        const uint8_t *pc = patch + R * kPatchPitch + R;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const int pp = pat8[bit];
            int t[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float fx = (float)(int8_t)(pp >> (16 * e)), fy = (float)(int8_t)(pp >> (16 * e + 8));
                const float a1 = fx * a, a2 = fy * b, b1 = fx * b, b2 = fy * a;
                const float rx = a1 - a2, ry = b1 + b2;
                t[e] = pc[(int)rintf(ry) * kPatchPitch + (int)rintf(rx)];
            }
            val |= (uint32_t)(t[0] < t[1]) << bit;
        }
    } else if (inside) {
        const uint32_t cbl = lbase + (uint32_t)cy * (uint32_t)fs + (uint32_t)cx;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const int pp = pat8[bit];
            int t[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float fx = (float)(int8_t)(pp >> (16 * e)), fy = (float)(int8_t)(pp >> (16 * e + 8));
                const float a1 = fx * a, a2 = fy * b, b1 = fx * b, b2 = fy * a;
                const float rx = a1 - a2, ry = b1 + b2;
                t[e] = blurred[cbl + (uint32_t)((int)rintf(ry) * fs + (int)rintf(rx))];
            }
            val |= (uint32_t)(t[0] < t[1]) << bit;
        }
    } else {
        const uint8_t *img = l == 0 ? images + lbase : fpyr + (size_t)f * G.fframe + fo + 3 * fs + 4;
        const uint8_t *blr = blurred + lbase;
#pragma unroll 1
        for (int bit = 0; bit < 8; bit++) {
            const int pp = pat8[bit];
            int t[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float fx = (float)(int8_t)(pp >> (16 * e)), fy = (float)(int8_t)(pp >> (16 * e + 8));
                const float a1 = fx * a, a2 = fy * b, b1 = fx * b, b2 = fy * a;
                const float rx = a1 - a2, ry = b1 + b2;
                const int x = cx + (int)rintf(rx), y = cy + (int)rintf(ry);
                if ((unsigned)x < (unsigned)lw && (unsigned)y < (unsigned)lh) t[e] = blr[(size_t)y * fs + x];
                else t[e] = img[(size_t)reflect101(y, lh) * fs + reflect101(x, lw)];
            }
            val |= (uint32_t)(t[0] < t[1]) << bit;
        }
    }
    desc[((size_t)f * kp_cap + kp) * VSLAM_DESC_BYTES + byte] = (uint8_t)val;
    if (byte == 0) {
        out_xy[((size_t)f * kp_cap + kp) * 2] = k4.x;
        out_xy[((size_t)f * kp_cap + kp) * 2 + 1] = k4.y;
        if (out_angle_octave) {
            out_angle_octave[((size_t)f * kp_cap + kp) * 2] = angle;
            out_angle_octave[((size_t)f * kp_cap + kp) * 2 + 1] = (float)l;
        }
    }
}

__global__ __launch_bounds__(256) void grid_zero_counts_kernel(int32_t *__restrict__ n_out, int frames) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < frames) n_out[i] = 0;
}

void umax_table(UmaxTable &U) {   // orb.cpp computeKeyPoints, halfPatchSize = 15
    const int half = 15;
    for (int v = 0; v < 17; v++) U.v[v] = 0;
    const int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) U.v[v] = (int)std::lrint(std::sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (U.v[v0] == U.v[v0 + 1]) ++v0;
        U.v[v] = v0;
        ++v0;
    }
}

// the retainBest tiers: lists of up to 511 keypoints four to a workgroup, up to 2047 and up to 8191 one to a workgroup (a
// one-wave workgroup; LDS decides how many a CU holds), longer ones (one cell of several megapixels) out of global scratch
constexpr int kTier1 = 512, kTier2 = 2048, kTier3 = 8192;

template <class S>
int launch_retain(vslam_ctx *ctx, const Geom &G, int n_lists, int max_len, uint32_t *ent, const float *resp, const int32_t *cnt_in,
                  int32_t *cnt_out, const uint8_t *flags, unsigned char *g_store, int *g_sl, int *g_sr) {
    hipStream_t st = ctx->stream;
    constexpr int per = S::kBytes + 4;
    retain_best_kernel<S, kTier1, 0, 4, false><<<vs_div_up(n_lists, 4), 256, (size_t)kTier1 * per * 4, st>>>(
        G, n_lists, ent, resp, cnt_in, cnt_out, flags, nullptr, nullptr, nullptr);
    if (max_len + 1 > kTier1)
        retain_best_kernel<S, kTier2, kTier1, 1, false><<<n_lists, 64, (size_t)kTier2 * per, st>>>(G, n_lists, ent, resp, cnt_in, cnt_out,
                                                                                               flags, nullptr, nullptr, nullptr);
    if (max_len + 1 > kTier2) {
        auto k = retain_best_kernel<S, kTier3, kTier2, 1, false>;
        const char *key = S::kBytes == 4 ? "orb_grid.retain3s" : "orb_grid.retain3r";
        if (kTier3 * per > 64 * 1024 && !ctx->attr_done[key]) {
            VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kTier3 * per));
            ctx->attr_done[key] = true;
        }
        k<<<n_lists, 64, (size_t)kTier3 * per, st>>>(G, n_lists, ent, resp, cnt_in, cnt_out, flags, nullptr, nullptr, nullptr);
    }
    if (max_len + 1 > kTier3)
        retain_best_kernel<S, 0, kTier3, 1, true><<<n_lists, 64, 0, st>>>(G, n_lists, ent, resp, cnt_in, cnt_out, flags, g_store, g_sl, g_sr);
    return VSLAM_OK;
}

}  // namespace

// extract_features(Frame&, nrows, ncols) for a batch of frames
int vs_launch_extract_grid(vslam_ctx *ctx, uint8_t *bgr, int frames, int w, int h, int stride, int nrows, int ncols,
                           const int8_t *pattern, int kp_cap, float *xy, uint8_t *desc, float *angle_octave,
                           int32_t *n_out) {
    VS_REQUIRE(ctx, bgr && pattern && xy && desc && n_out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w > 0 && h > 0 && stride >= 3 * w && nrows > 0 && ncols > 0 && kp_cap > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, w / ncols >= 7 && h / nrows >= 7, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, w <= 32768 && h <= 32768 && nrows * ncols <= 2048, VSLAM_ERR_CAPACITY);
    if (reinterpret_cast<uintptr_t>(pattern) & 15) {   // the descriptor kernel reads the table as 16-byte groups
        int8_t *aligned = nullptr;
        if (int prc = vs_arena_get(ctx, "grid.pattern", 1024, (void **)&aligned)) return prc;
        VS_HIP(ctx, hipMemcpyAsync(aligned, pattern, 1024, hipMemcpyDeviceToDevice, ctx->stream));
        pattern = aligned;
    }
    Geom G;
    make_geom(w, h, nrows, ncols, G);
    const int cells = G.cells, units = frames * cells;
    hipStream_t st = ctx->stream;

    // [gray frames | cell pyramids] and [blurred frames | blurred frame pyramids] are one allocation each: the descriptor kernel
    // addresses them with one base and 32-bit offsets
    const size_t gbytes = ((size_t)frames * G.gp * h + 16 + 255) & ~(size_t)255;   // + slack: rows are read as dwords
    const size_t images_bytes = gbytes + (size_t)G.cunit * units + 16, blurred_bytes = gbytes + (size_t)G.fframe * frames + 16;
    VS_REQUIRE(ctx, images_bytes < (1ull << 32) && blurred_bytes < (1ull << 32), VSLAM_ERR_CAPACITY);
    uint8_t *images = nullptr, *blurred = nullptr;
    int rc;
    if ((rc = vs_arena_get(ctx, "grid.images", images_bytes, (void **)&images))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.blurred", blurred_bytes, (void **)&blurred))) return rc;
    uint8_t *gray = images, *cpyr = images + gbytes, *blur0 = blurred, *fblur = blurred + gbytes;
    {   // :32 outlines into the caller's image + gray of the result (ORB converts BGR ROIs to gray)
        VsProfScope ps(ctx, "grid_outline_gray_kernel");
        if (G.gp != w)
            grid_outline_gray_padded_kernel<<<dim3(vs_div_up((G.gp / 4) * h, 256), frames), 256, 0, st>>>(bgr, w, h, stride, G.cw, G.ch,
                                                                                                          ncols, nrows, gray, G.gp);
        else if (w % 4 == 0 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(bgr) & 3) == 0)
            grid_outline_gray4_kernel<<<dim3(vs_div_up((w / 4) * h, 256), frames), 256, 0, st>>>(bgr, w, h, stride, G.cw, G.ch, ncols,
                                                                                                 nrows, gray);
        else
            grid_outline_gray_kernel<<<dim3(vs_div_up(w * h, 256), frames), 256, 0, st>>>(bgr, w, h, stride, G.cw, G.ch, ncols, nrows,
                                                                                          gray);
    }
    if (G.nlv == 0) {   // no level of a cell's pyramid is larger than the 31-px border on every side: no keypoints at all
        grid_zero_counts_kernel<<<vs_div_up(frames, 256), 256, 0, st>>>(n_out, frames);
        VS_HIP(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    // one (cell, level) inner region is indexed in 24 bits and its strip tiles have to fit LDS
    for (int l = 0; l < G.nlv; l++) VS_REQUIRE(ctx, (long long)G.rw[l] * G.rh[l] < (1ll << 24), VSLAM_ERR_CAPACITY);
    int fast_lds = 0;
    {
        const int rw0 = G.rw[0];
        const int SW = (rw0 + 8 + 3) & ~3, MW = (rw0 + 2 + 3) & ~3, CPR = (rw0 + 63) >> 6;
        const int fixed = 8 * SW + 2 * MW + 2 * MW + 256 * 4 + 64, per_row = SW + MW + MW + CPR * 24;   // as fast_strip_rows
        int R = std::min(24, G.rh[0]);
        while (R > 4 && fixed + per_row * R > 40 * 1024) R--;
        fast_lds = fixed + per_row * R;
        VS_REQUIRE(ctx, fast_lds <= 150 * 1024, VSLAM_ERR_CAPACITY);
        VS_REQUIRE(ctx, fast_strip_rows(rw0, fast_lds) >= std::min(4, G.rh[0]), VSLAM_ERR_CAPACITY);
    }

    const size_t LT = (size_t)G.loff[G.nlv], slots = (size_t)units * 2 * LT;
    uint8_t *fpyr = nullptr, *flags = nullptr;
    int32_t *pat_r = nullptr;
    uint32_t *tab = nullptr, *ent = nullptr;
    float *resp = nullptr;
    float4 *fkp = nullptr;
    int32_t *cnt = nullptr;
    if ((rc = vs_arena_get(ctx, "grid.pat_r", 16, (void **)&pat_r))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.fpyr", (size_t)G.fframe * frames + 16, (void **)&fpyr))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.tab", sizeof(uint32_t) * (size_t)(G.ttotal + 1), (void **)&tab))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.ent", sizeof(uint32_t) * slots, (void **)&ent))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.resp", sizeof(float) * slots, (void **)&resp))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.fkp", sizeof(float4) * (size_t)frames * kp_cap, (void **)&fkp))) return rc;
    // counters: cnt0 / cnt1 / cnt2 [units][2][8], c1_20 [units][8]; flags [units]
    const size_t cwords = (size_t)units * 2 * kMaxLevels;
    if ((rc = vs_arena_get(ctx, "grid.cnt", sizeof(int32_t) * (3 * cwords + (size_t)units * kMaxLevels), (void **)&cnt))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.flags", (size_t)units, (void **)&flags))) return rc;
    int32_t *cnt0 = cnt, *cnt1 = cnt + cwords, *cnt2 = cnt + 2 * cwords, *c1_20 = cnt + 3 * cwords;
    unsigned char *g_store = nullptr;
    int *g_sl = nullptr, *g_sr = nullptr;
    if (G.cap[0] + 1 > kTier3) {   // per-list global scratch for lists no LDS tier holds
        if ((rc = vs_arena_get(ctx, "grid.sel_store", 8 * slots, (void **)&g_store))) return rc;
        if ((rc = vs_arena_get(ctx, "grid.sel_sl", sizeof(int) * slots, (void **)&g_sl))) return rc;
        if ((rc = vs_arena_get(ctx, "grid.sel_sr", sizeof(int) * (slots + cwords + 1), (void **)&g_sr))) return rc;
    }
    UmaxTable U;
    umax_table(U);

    if (G.nlv > 1) {   // both pyramids, level by level (each level is a resize of the one before)
        VsProfScope ps(ctx, "grid_pyramid_kernels");
        resize_tables_kernel<<<vs_div_up(G.ttotal, 256), 256, 0, st>>>(G, tab);
        for (int l = 1; l < G.nlv; l++) {
            const int txF = vs_div_up(G.fstride[l], kRTW), txC = vs_div_up(G.cstride[l], kRTW);
            const int nbF = txF * vs_div_up(G.flh[l] + 6, kRTH), nbC = txC * vs_div_up(G.clh[l], kRTH);
            pyr_resize_kernel<<<frames * nbF + units * nbC, 256, 0, st>>>(G, l, frames * nbF, txF, nbF, txC, nbC, gray, fpyr, cpyr, tab);
        }
    }
    {
        VsProfScope ps(ctx, "fast_collect_kernel");
        if (fast_lds > 64 * 1024 && !ctx->attr_done["orb_grid.fast"]) {
            VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(fast_collect_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            ctx->attr_done["orb_grid.fast"] = true;
        }
        fast_collect_kernel<<<units * G.nlv, 256, fast_lds, st>>>(G, units, fast_lds, gray, cpyr, ent, cnt0, c1_20);
        grid_decide_kernel<<<vs_div_up(units, 256), 256, 0, st>>>(G, units, c1_20, flags);
    }
    const int n_lists = units * 2 * G.nlv;
    {
        VsProfScope ps(ctx, "orb_select_kernels");
        if ((rc = launch_retain<ScoreStore>(ctx, G, n_lists, G.cap[0], ent, resp, cnt0, cnt1, flags, g_store, g_sl, g_sr))) return rc;
        harris_kernel<<<n_lists, 64, 0, st>>>(G, gray, cpyr, ent, resp, cnt1, flags);
        if ((rc = launch_retain<RespStore>(ctx, G, n_lists, G.cap[0], ent, resp, cnt1, cnt2, flags, g_store, g_sl, g_sr))) return rc;
    }
    {
        VsProfScope ps(ctx, "grid_assemble_kernel");
        const size_t lds = sizeof(int) * (size_t)G.nlv * cells + (size_t)cells + 16;
        grid_assemble_kernel<<<frames, 256, lds, st>>>(G, ent, cnt2, flags, kp_cap, fkp, n_out);
    }
    {   // ORB::compute (:43): per-level blur of the outlined frame's pyramid, steered BRIEF
        VsProfScope ps(ctx, "orb_compute_kernels");
        if (w >= 4 && h >= 4) {
            ctx->img_pitch = G.gp != w ? G.gp : 0;   // padded rows: the streaming blur takes them as they are (blur.hip)
            rc = vs_launch_gaussian7(ctx, gray, frames, w, h, blur0);
            ctx->img_pitch = 0;
            if (rc) return rc;
        }
        // levels >= 1 as images of fstride x (lh + 6) bytes whose margins hold the reflect values: rows 3 .. lh + 2 through the
        // streaming blur (what it writes into the margin columns of fblur is never read)
        for (int l = 1; l < G.nlv; l++)
            if ((rc = vs_launch_gaussian7_rows(ctx, fpyr + G.foff[l], fblur + G.foff[l], frames, (size_t)G.fframe, G.fstride[l],
                                               G.flh[l] + 6, 3, G.flh[l] + 3)))
                return rc;
        pattern_radius_kernel<<<1, 512, 0, st>>>(pattern, pat_r);
        const int per_frame = vs_div_up(kp_cap, 8);
        orb_desc_kernel<<<vs_xcd_grid(frames, per_frame), 256, 0, st>>>(G, U, (uint32_t)gbytes, images, blurred, fpyr, fkp, n_out, kp_cap,
                                                                        pattern, pat_r, desc, xy, angle_octave, frames, per_frame);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

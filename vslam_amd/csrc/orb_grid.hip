// Grid ORB/FAST extractor for gfx950.
//
// Replaces extract_features(Frame&, nrows, ncols), /root/reference/src/Frame.cpp:16-51 (the
// extractor whose only call is commented out at src/vslam.cpp:63; it is the "ORB/FAST" of the
// north star).  Per cell: black 1-px outline drawn into the image (:32), ORB(500, 1.2, 8, 31, 0, 2,
// HARRIS_SCORE, 31, fastThreshold 20)->detect, replaced by the fastThreshold-5 result when fewer
// than 500 were found (:33-36); keypoints shifted to image coordinates (:37-40); then
// ORB::compute over the whole (outlined) image (:43).
//
// The OpenCV internals follow the oracle (oracle/vso_orb.cpp) step for step:
//   pyramid of every cell (8 levels, 32-px REFLECT_101 frame, INTER_LINEAR_EXACT resize in Q8/Q16)
//   FAST-9/16: one pass stores M = max(dark arc score, bright arc score) per pixel; a pixel is a
//     corner at threshold t iff M > t and its OpenCV score is M - 1 for every t, so both detectors
//     (t = 20 and t = 5) share the map; 3x3 non-max suppression and the raster-order list per t
//   KeyPointsFilter::retainBest = std::nth_element + std::partition: replayed with introselect.h so
//     the surviving ORDER is libstdc++'s (it decides descriptor row order)
//   Harris response (7x7, k = 0.04), second retainBest, intensity-centroid angle with fastAtan2's
//     polynomial, steered BRIEF at the keypoint's pyramid level with pinned sin/cos.
// This path is built for coverage and exactness, not yet tuned: the selection replays are serial
// per (cell, level) by nature.
#include "ctx.h"
#include "introselect.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kMaxLevels = 8;

struct PyrLayout {
    int nlevels, border, bufw, bufh;
    int lx[kMaxLevels], ly[kMaxLevels], lw[kMaxLevels], lh[kMaxLevels];
    float scale[kMaxLevels];
    int roi_prefix[kMaxLevels + 1];   // running sum of lw*lh
    int ext_prefix[kMaxLevels + 1];   // running sum of (lw+2b)*(lh+2b)
};

// orb.cpp pyramid layout (host arithmetic only)
PyrLayout make_layout(int w, int h, int nlevels, double scaleFactor) {
    PyrLayout P{};
    const int patchSize = 31, edgeThreshold = 31, HARRIS_BLOCK_SIZE = 9;
    const int halfPatchSize = patchSize / 2;
    const int descPatchSize = (int)std::ceil(halfPatchSize * std::sqrt(2.0));
    P.border = std::max(edgeThreshold, std::max(descPatchSize, HARRIS_BLOCK_SIZE / 2)) + 1;
    P.nlevels = nlevels;
    P.bufw = ((w + P.border * 2) + 15) & ~15;
    int level_dy = h + P.border * 2, ox = 0, oy = 0;
    P.roi_prefix[0] = P.ext_prefix[0] = 0;
    for (int l = 0; l < nlevels; l++) {
        const float sc = (float)std::pow(scaleFactor, (double)l);
        P.scale[l] = sc;
        const float inv = 1.0f / sc;
        const int sw = (int)std::lrint(w * inv), sh = (int)std::lrint(h * inv);
        const int ww = sw + P.border * 2, wh = sh + P.border * 2;
        if (ox + ww > P.bufw) {
            ox = 0;
            oy += level_dy;
            level_dy = wh;
        }
        P.lx[l] = ox + P.border;
        P.ly[l] = oy + P.border;
        P.lw[l] = sw;
        P.lh[l] = sh;
        ox += ww;
        P.roi_prefix[l + 1] = P.roi_prefix[l] + sw * sh;
        P.ext_prefix[l + 1] = P.ext_prefix[l] + ww * wh;
    }
    P.bufh = oy + level_dy;
    return P;
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
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

// level 0 of a pyramid (ROI + reflect frame) from a gray image region: unit u of a frame is the
// cell (i = column, j = row) with origin (i*cw, j*ch); cells == 1 means "whole frame"
__global__ __launch_bounds__(256) void pyr_level0_kernel(const uint8_t *__restrict__ gray, int w, int h, int cw, int ch,
                                                         int ncols, int nrows, PyrLayout L, uint8_t *__restrict__ pyr) {
    const int u = blockIdx.y;
    const int cells = ncols * nrows;
    const int f = u / cells, c = u - f * cells;
    const int ci = c / nrows, cj = c - ci * nrows;   // columns outer, rows inner (src/Frame.cpp:27-28)
    const int sx = ci * cw, sy = cj * ch;
    const int ew = L.lw[0] + 2 * L.border, eh = L.lh[0] + 2 * L.border;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ew * eh) return;
    const int by = i / ew, bx = i - by * ew;
    const int x = reflect101(bx - L.border, L.lw[0]), y = reflect101(by - L.border, L.lh[0]);
    pyr[(size_t)u * L.bufw * L.bufh + (size_t)(L.ly[0] - L.border + by) * L.bufw + (L.lx[0] - L.border + bx)] =
        gray[((size_t)f * h + sy + y) * w + sx + x];
}

// INTER_LINEAR_EXACT coefficient for destination index d (Q8), resize.cpp interpolationLinear
__device__ __forceinline__ void linear_coeff(int d, int dst_n, int src_n, int &ofs, int &c0, int &c1) {
    const double inv_scale = (double)dst_n / (double)src_n;
    const double scale = 1.0 / inv_scale;
    const double fval = scale * ((double)d + 0.5) - 0.5;
    int ival = (int)floor(fval);
    if (ival >= 0 && src_n > 1) {
        if (ival < src_n - 1) {
            c1 = (int)rint((fval - (double)ival) * 256.0);
            c0 = 256 - c1;
        } else {
            ival = src_n - 2;
            c0 = 0;
            c1 = 256;
        }
    } else {
        ival = 0;
        c0 = 256;
        c1 = 0;
    }
    ofs = ival;
}

// level l (ROI + reflect frame) from the ROI of level l-1
__global__ __launch_bounds__(256) void pyr_resize_kernel(PyrLayout L, int l, uint8_t *__restrict__ pyr) {
    const int u = blockIdx.y;
    uint8_t *base = pyr + (size_t)u * L.bufw * L.bufh;
    const int dw = L.lw[l], dh = L.lh[l], sw = L.lw[l - 1], sh = L.lh[l - 1];
    const int ew = dw + 2 * L.border, eh = dh + 2 * L.border;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ew * eh) return;
    const int by = i / ew, bx = i - by * ew;
    const int x = reflect101(bx - L.border, dw), y = reflect101(by - L.border, dh);
    int xo, x0, x1, yo, y0, y1;
    linear_coeff(x, dw, sw, xo, x0, x1);
    linear_coeff(y, dh, sh, yo, y0, y1);
    const uint8_t *src = base + (size_t)L.ly[l - 1] * L.bufw + L.lx[l - 1];
    const uint8_t *r0 = src + (size_t)yo * L.bufw, *r1 = src + (size_t)min(yo + 1, sh - 1) * L.bufw;
    const int xb = min(xo + 1, sw - 1);
    const uint32_t h0 = (uint32_t)r0[xo] * x0 + (uint32_t)r0[xb] * x1;
    const uint32_t h1 = (uint32_t)r1[xo] * x0 + (uint32_t)r1[xb] * x1;
    const uint32_t v = h0 * y0 + h1 * y1;
    base[(size_t)(L.ly[l] - L.border + by) * L.bufw + (L.lx[l] - L.border + bx)] = (uint8_t)((v + (1u << 15)) >> 16);
}

// ------------------------------------------------------------------------------------------
// FAST-9/16 arc score map M (fast.cpp FAST_t<16> + cornerScore<16>)
// ------------------------------------------------------------------------------------------
constexpr int kFastMinThreshold = 5;   // lowest threshold any detector here uses

__global__ __launch_bounds__(256) void fast_score_kernel(PyrLayout L, const uint8_t *__restrict__ pyr,
                                                         uint8_t *__restrict__ M) {
    const int u = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.roi_prefix[L.nlevels]) return;
    int l = 0;
    while (i >= L.roi_prefix[l + 1]) l++;
    const int r = i - L.roi_prefix[l];
    const int lw = L.lw[l], lh = L.lh[l];
    const int y = r / lw, x = r - y * lw;
    const size_t pos = (size_t)u * L.bufw * L.bufh + (size_t)(L.ly[l] + y) * L.bufw + L.lx[l] + x;
    uint8_t out = 0;
    if (x >= 3 && x < lw - 3 && y >= 3 && y < lh - 3) {
        const uint8_t *p = pyr + pos;
        const int s = L.bufw;
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * s];      d[1] = v - p[3 * s + 1];  d[2] = v - p[2 * s + 2];  d[3] = v - p[s + 3];
        d[4] = v - p[3];          d[5] = v - p[-s + 3];     d[6] = v - p[-2 * s + 2]; d[7] = v - p[-3 * s + 1];
        d[8] = v - p[-3 * s];     d[9] = v - p[-3 * s - 1]; d[10] = v - p[-2 * s - 2]; d[11] = v - p[-s - 3];
        d[12] = v - p[-3];        d[13] = v - p[s - 3];     d[14] = v - p[2 * s - 2]; d[15] = v - p[3 * s - 1];
        // any 9-arc contains at least two of the four compass pixels: cheap necessary test
        int dark = 0, bright = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            dark += d[k] > kFastMinThreshold;
            bright += d[k] < -kFastMinThreshold;
        }
        if (dark >= 2 || bright >= 2) {
            int best = 0;
#pragma unroll
            for (int st = 0; st < 16; st++) {
                int mn = d[st], mx = d[st];
#pragma unroll
                for (int j = 1; j < 9; j++) {
                    const int e = d[(st + j) & 15];
                    mn = min(mn, e);
                    mx = max(mx, e);
                }
                best = max(best, max(mn, -mx));   // dark arcs: min d; bright arcs: min(-d) = -max d
            }
            out = (uint8_t)min(max(best, 0), 255);
        }
    }
    M[pos] = out;
}

// ------------------------------------------------------------------------------------------
// per (unit, level) keypoint slots
// ------------------------------------------------------------------------------------------
struct SlotLayout {
    int cap[kMaxLevels];      // capacity of each level's slot
    int off[kMaxLevels + 1];  // prefix of caps
    int per_level[kMaxLevels];   // nfeaturesPerLevel
};

// FAST keypoints of one (unit, level) at threshold t: 3x3 NMS on scores s = (M > t) ? M - 1 : 0,
// runByImageBorder(edgeThreshold 31), raster order (ordered ballot compaction).
__global__ __launch_bounds__(256) void fast_collect_kernel(PyrLayout L, SlotLayout SL, const uint8_t *__restrict__ M, int thr,
                                                           short2 *__restrict__ kp_xy, float *__restrict__ kp_resp,
                                                           int32_t *__restrict__ kp_cnt) {
    const int u = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int lw = L.lw[l], lh = L.lh[l];
    const int edge = 31;
    const size_t slot = (size_t)u * SL.off[L.nlevels] + SL.off[l];
    int32_t *cnt = kp_cnt + (size_t)u * kMaxLevels + l;
    if (lh <= edge * 2 || lw <= edge * 2) {   // runByImageBorder clears everything
        if (tid == 0) *cnt = 0;
        return;
    }
    const int rw = lw - 2 * edge, rh = lh - 2 * edge;
    const uint8_t *Mb = M + (size_t)u * L.bufw * L.bufh + (size_t)L.ly[l] * L.bufw + L.lx[l];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < rw * rh; i0 += 256) {
        const int i = i0 + tid;
        bool keep = false;
        int x = 0, y = 0, sc = 0;
        if (i < rw * rh) {
            y = edge + i / rw;
            x = edge + (i - (i / rw) * rw);
            const uint8_t *c = Mb + (size_t)y * L.bufw + x;
            const int m = c[0];
            if (m > thr) {
                sc = m - 1;
                keep = true;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        const int mn = c[dy * L.bufw + dx];
                        const int sn = mn > thr ? mn - 1 : 0;
                        keep = keep && (sc > sn);
                    }
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = (int)__popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int wv = 0; wv < wave; wv++) off += s_wave[wv];
        off += (int)__popcll(bal & ((1ull << lane) - 1ull));
        if (keep && off < SL.cap[l]) {
            kp_xy[slot + off] = make_short2((short)x, (short)y);
            kp_resp[slot + off] = (float)sc;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) *cnt = min(s_base, SL.cap[l]);
}

// KeyPointsFilter::retainBest(list, n_points) in place (keypoint.cpp): nth_element by response
// descending, then std::partition of the tail on response >= boundary.  One lane replays it.
struct RespStore {
    using value_type = int2;   // (response bits, payload index)
    using key_type = float;
    float *key_;
    int *pay_;
    __device__ int2 get(int i) const { return make_int2(__float_as_int(key_[i]), pay_[i]); }
    __device__ void set(int i, const int2 &v) {
        key_[i] = __int_as_float(v.x);
        pay_[i] = v.y;
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

// xy_in -> xy_out (permuted when a selection happens, copied otherwise); responses permuted in place.
// Work arrays live in LDS when the list fits (the usual case) and in a per-slot global scratch otherwise.
__global__ __launch_bounds__(64) void retain_best_kernel(PyrLayout L, SlotLayout SL, int mult, const short2 *__restrict__ xy_in,
                                                         short2 *__restrict__ xy_out, float *__restrict__ kp_resp,
                                                         int32_t *__restrict__ kp_cnt, int lds_entries,
                                                         float *__restrict__ g_key, int *__restrict__ g_pay,
                                                         int *__restrict__ g_sl, int *__restrict__ g_sr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int u = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
    int32_t *cnt = kp_cnt + (size_t)u * kMaxLevels + l;
    const int n = *cnt;
    const int n_points = mult * SL.per_level[l];
    const size_t slot = (size_t)u * SL.off[L.nlevels] + SL.off[l];
    if (!(n_points >= 0 && n > n_points)) {
        for (int i = tid; i < n; i += 64) xy_out[slot + i] = xy_in[slot + i];
        return;
    }
    if (n_points == 0) {
        if (tid == 0) *cnt = 0;
        return;
    }
    float *key;
    int *pay, *sl, *sr;
    if (n + 1 <= lds_entries) {
        key = reinterpret_cast<float *>(smem);
        pay = reinterpret_cast<int *>(key + lds_entries);
        sl = pay + lds_entries;
        sr = sl + lds_entries;
    } else {
        key = g_key + slot;
        pay = g_pay + slot;
        sl = g_sl + slot;
        sr = g_sr + slot + ((size_t)u * kMaxLevels + l);   // one spare element per slot
    }
    for (int i = tid; i < n; i += 64) {
        key[i] = kp_resp[slot + i];
        pay[i] = i;
    }
    __syncthreads();
    __shared__ int s_new_n;
    RespStore s{key, pay};
    vs_sel::wave_nth_element(s, 0, n_points - 1, n, sl, sr);   // the block is one wave
    __syncthreads();
    if (tid == 0) {
        const float ambiguous = key[n_points - 1];
        // std::partition (bidirectional), pred: response >= ambiguous
        int first = n_points, last = n;
        while (true) {
            while (true) {
                if (first == last) goto done;
                if (key[first] >= ambiguous) ++first;
                else break;
            }
            --last;
            while (true) {
                if (first == last) goto done;
                if (!(key[last] >= ambiguous)) --last;
                else break;
            }
            s.swap(first, last);
            ++first;
        }
    done:
        s_new_n = first;
    }
    __syncthreads();
    const int m = s_new_n;
    for (int i = tid; i < m; i += 64) {
        kp_resp[slot + i] = key[i];
        xy_out[slot + i] = xy_in[slot + pay[i]];
    }
    if (tid == 0) *cnt = m;
}

// HarrisResponses(pyramid, keypoints, 7, 0.04) for every keypoint of every slot
__global__ __launch_bounds__(256) void harris_kernel(PyrLayout L, SlotLayout SL, const uint8_t *__restrict__ pyr,
                                                     const short2 *__restrict__ kp_xy, float *__restrict__ kp_resp,
                                                     const int32_t *__restrict__ kp_cnt) {
    const int u = blockIdx.z, l = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kp_cnt[(size_t)u * kMaxLevels + l]) return;
    const size_t slot = (size_t)u * SL.off[L.nlevels] + SL.off[l];
    const short2 p = kp_xy[slot + i];
    const int step = L.bufw, blockSize = 7, r = blockSize / 2;
    const uint8_t *ptr0 = pyr + (size_t)u * L.bufw * L.bufh + (size_t)(p.y - r + L.ly[l]) * step + p.x - r + L.lx[l];
    int a = 0, b = 0, c = 0;
    for (int yy = 0; yy < blockSize; yy++)
        for (int xx = 0; xx < blockSize; xx++) {
            const uint8_t *ptr = ptr0 + yy * step + xx;
            const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
            const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float t1 = fa * fb, t2 = fc * fc, sum = fa + fb;
    const float t3 = 0.04f * sum;
    const float t4 = t3 * sum;
    kp_resp[slot + i] = ((t1 - t2) - t4) * scale_sq_sq;
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

// ICAngles + "pt *= scale": final per-slot keypoints (level coordinates -> cell coordinates)
__global__ __launch_bounds__(256) void ic_angle_kernel(PyrLayout L, SlotLayout SL, UmaxTable U, const uint8_t *__restrict__ pyr,
                                                       const short2 *__restrict__ kp_xy, const int32_t *__restrict__ kp_cnt,
                                                       float4 *__restrict__ kp_final) {
    const int u = blockIdx.z, l = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kp_cnt[(size_t)u * kMaxLevels + l]) return;
    const size_t slot = (size_t)u * SL.off[L.nlevels] + SL.off[l];
    const short2 p = kp_xy[slot + i];
    const int step = L.bufw, half_k = 15;
    const uint8_t *center = pyr + (size_t)u * L.bufw * L.bufh + (size_t)(p.y + L.ly[l]) * step + p.x + L.lx[l];
    int m_01 = 0, m_10 = 0;
    for (int uu = -half_k; uu <= half_k; ++uu) m_10 += uu * center[uu];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0;
        const int d = U.v[v];
        for (int uu = -d; uu <= d; ++uu) {
            const int val_plus = center[uu + v * step], val_minus = center[uu - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += uu * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    const float angle = fast_atan2_deg((float)m_01, (float)m_10);
    const float sc = L.scale[l];
    kp_final[slot + i] = make_float4((float)p.x * sc, (float)p.y * sc, angle, (float)l);
}

// ------------------------------------------------------------------------------------------
// frame assembly: choose detector per cell (:34-36), shift (:37-40), runByImageBorder(31) on the
// frame and ORB::compute's regrouping by level (stable).  One workgroup per frame.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_assemble_kernel(PyrLayout L, SlotLayout SL, int cells, int nrows, int cw, int ch,
                                                            int w, int h, int nfeatures, const float4 *__restrict__ fin20,
                                                            const int32_t *__restrict__ cnt20, const float4 *__restrict__ fin5,
                                                            const int32_t *__restrict__ cnt5, int kp_cap,
                                                            float4 *__restrict__ out_kp, int32_t *__restrict__ out_n) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    if (tid == 0) s_base = 0;
    __syncthreads();
    float4 *O = out_kp + (size_t)f * kp_cap;
    for (int lvl = 0; lvl < L.nlevels; lvl++) {   // output is grouped by level, cell order inside
        for (int c = 0; c < cells; c++) {
            const int u = f * cells + c;
            int tot20 = 0;
            for (int k = 0; k < L.nlevels; k++) tot20 += cnt20[(size_t)u * kMaxLevels + k];
            const bool use20 = tot20 >= nfeatures;   // `if (temp.size() < nfeatures)` -> fallback replaces
            const float4 *src = (use20 ? fin20 : fin5) + (size_t)u * SL.off[L.nlevels] + SL.off[lvl];
            const int n = (use20 ? cnt20 : cnt5)[(size_t)u * kMaxLevels + lvl];
            const int ci = c / nrows, cj = c - ci * nrows;
            const float sx = (float)(ci * cw), sy = (float)(cj * ch);
            for (int i0 = 0; i0 < n; i0 += 256) {
                const int i = i0 + tid;
                bool keep = false;
                float4 k4 = make_float4(0, 0, 0, 0);
                if (i < n) {
                    k4 = src[i];
                    k4.x = sx + k4.x;
                    k4.y = sy + k4.y;
                    keep = k4.x >= 31.f && k4.x < (float)(w - 31) && k4.y >= 31.f && k4.y < (float)(h - 31) &&
                           !(h <= 62 || w <= 62);
                }
                const unsigned long long bal = __ballot(keep);
                if (lane == 0) s_wave[wave] = (int)__popcll(bal);
                __syncthreads();
                int off = s_base;
                for (int wv = 0; wv < wave; wv++) off += s_wave[wv];
                off += (int)__popcll(bal & ((1ull << lane) - 1ull));
                if (keep && off < kp_cap) O[off] = k4;
                __syncthreads();
                if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
                __syncthreads();
            }
        }
    }
    if (tid == 0) out_n[f] = min(s_base, kp_cap);
}

// GaussianBlur 7x7 sigma 2 of every level's ROI, reading the (reflect-framed) source pyramid
__global__ __launch_bounds__(256) void pyr_blur_kernel(PyrLayout L, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst) {
    const int u = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.roi_prefix[L.nlevels]) return;
    int l = 0;
    while (i >= L.roi_prefix[l + 1]) l++;
    const int r = i - L.roi_prefix[l];
    const int y = r / L.lw[l], x = r - y * L.lw[l];
    const size_t pos = (size_t)u * L.bufw * L.bufh + (size_t)(L.ly[l] + y) * L.bufw + L.lx[l] + x;
    const uint8_t *c = src + pos;
    const int s = L.bufw;
    const int kq[7] = {18, 34, 48, 56, 48, 34, 18};
    uint32_t acc = 0;
#pragma unroll
    for (int dy = -3; dy <= 3; dy++) {
        uint32_t row = 0;
#pragma unroll
        for (int dx = -3; dx <= 3; dx++) row += (uint32_t)kq[dx + 3] * c[dy * s + dx];
        acc += (uint32_t)kq[dy + 3] * row;
    }
    dst[pos] = (uint8_t)((acc + (1u << 15)) >> 16);
}

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

// computeOrbDescriptors: lane per (keypoint, byte) on the blurred frame pyramid
__global__ __launch_bounds__(256) void orb_desc_kernel(PyrLayout L, const uint8_t *__restrict__ blurred,
                                                       const float4 *__restrict__ kps, const int32_t *__restrict__ n_arr,
                                                       int kp_cap, const int8_t *__restrict__ pattern,
                                                       uint8_t *__restrict__ desc, float *__restrict__ out_xy,
                                                       float *__restrict__ out_angle_octave) {
    const int f = blockIdx.y, tid = threadIdx.x;
    const int kp = blockIdx.x * 8 + (tid >> 5), byte = tid & 31;
    if (kp >= n_arr[f]) return;
    const float4 k4 = kps[(size_t)f * kp_cap + kp];
    const int l = (int)k4.w;
    const float scale = 1.f / L.scale[l];
    float a, b;
    sincos_deg_pinned(k4.z, b, a);
    const int cx = (int)rintf(k4.x * scale), cy = (int)rintf(k4.y * scale);
    const uint8_t *center = blurred + (size_t)f * L.bufw * L.bufh + (size_t)(cy + L.ly[l]) * L.bufw + cx + L.lx[l];
    uint32_t val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int8_t *pp = pattern + (size_t)(byte * 8 + bit) * 4;
        int t[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float fx = (float)pp[2 * e], fy = (float)pp[2 * e + 1];
            const float a1 = fx * a, a2 = fy * b, b1 = fx * b, b2 = fy * a;
            const float rx = a1 - a2, ry = b1 + b2;
            t[e] = center[(int)rintf(ry) * L.bufw + (int)rintf(rx)];
        }
        val |= (uint32_t)(t[0] < t[1]) << bit;
    }
    desc[((size_t)f * kp_cap + kp) * VSLAM_DESC_BYTES + byte] = (uint8_t)val;
    if (byte == 0) {
        out_xy[((size_t)f * kp_cap + kp) * 2] = k4.x;
        out_xy[((size_t)f * kp_cap + kp) * 2 + 1] = k4.y;
        if (out_angle_octave) {
            out_angle_octave[((size_t)f * kp_cap + kp) * 2] = k4.z;
            out_angle_octave[((size_t)f * kp_cap + kp) * 2 + 1] = k4.w;
        }
    }
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

}  // namespace

// extract_features(Frame&, nrows, ncols) for a batch of frames
int vs_launch_extract_grid(vslam_ctx *ctx, uint8_t *bgr, int frames, int w, int h, int stride, int nrows, int ncols,
                           const int8_t *pattern, int kp_cap, float *xy, uint8_t *desc, float *angle_octave,
                           int32_t *n_out) {
    VS_REQUIRE(ctx, bgr && pattern && xy && desc && n_out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w > 0 && h > 0 && stride >= 3 * w && nrows > 0 && ncols > 0 && kp_cap > 0, VSLAM_ERR_INVALID);
    const int nfeatures = 500, nlevels = 8;
    const double scaleFactor = 1.2;
    const int cw = w / ncols, ch = h / nrows;   // src/Frame.cpp:20
    VS_REQUIRE(ctx, cw >= 7 && ch >= 7, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, cw < 32000 && ch < 32000, VSLAM_ERR_CAPACITY);
    const int cells = nrows * ncols, units = frames * cells;
    const PyrLayout LC = make_layout(cw, ch, nlevels, scaleFactor);
    const PyrLayout LF = make_layout(w, h, nlevels, scaleFactor);

    SlotLayout SL{};
    {   // computeKeyPoints' nfeaturesPerLevel
        const float factor = (float)(1.0 / scaleFactor);
        float ndesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            SL.per_level[l] = (int)std::lrint(ndesired);
            sum += SL.per_level[l];
            ndesired *= factor;
        }
        SL.per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
        SL.off[0] = 0;
        for (int l = 0; l < nlevels; l++) {
            const int rw = LC.lw[l] - 62, rh = LC.lh[l] - 62;   // NMS survivors are isolated: <= ceil(rw/2)*ceil(rh/2)
            SL.cap[l] = (rw > 0 && rh > 0) ? ((rw + 1) / 2) * ((rh + 1) / 2) : 1;
            SL.off[l + 1] = SL.off[l] + SL.cap[l];
        }
    }
    UmaxTable U;
    umax_table(U);

    const size_t cell_pyr = (size_t)LC.bufw * LC.bufh, frame_pyr = (size_t)LF.bufw * LF.bufh;
    const size_t slots = (size_t)units * SL.off[nlevels];
    uint8_t *gray = nullptr, *pyrc = nullptr, *mmap = nullptr, *pyrf = nullptr, *pyrb = nullptr;
    short2 *kxy[2] = {nullptr, nullptr};
    float *kresp[2] = {nullptr, nullptr};
    float4 *kfin[2] = {nullptr, nullptr}, *fkp = nullptr;
    int32_t *kcnt[2] = {nullptr, nullptr};
    int rc;
    if ((rc = vs_arena_get(ctx, "grid.gray", (size_t)frames * w * h, (void **)&gray))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.pyrc", cell_pyr * units, (void **)&pyrc))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.mmap", cell_pyr * units, (void **)&mmap))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.pyrf", frame_pyr * frames, (void **)&pyrf))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.pyrb", frame_pyr * frames, (void **)&pyrb))) return rc;
    for (int t = 0; t < 2; t++) {
        const std::string s = t ? "5" : "20";
        if ((rc = vs_arena_get(ctx, ("grid.kxy" + s).c_str(), sizeof(short2) * slots, (void **)&kxy[t]))) return rc;
        if ((rc = vs_arena_get(ctx, ("grid.kresp" + s).c_str(), sizeof(float) * slots, (void **)&kresp[t]))) return rc;
        if ((rc = vs_arena_get(ctx, ("grid.kfin" + s).c_str(), sizeof(float4) * slots, (void **)&kfin[t]))) return rc;
        if ((rc = vs_arena_get(ctx, ("grid.kcnt" + s).c_str(), sizeof(int32_t) * (size_t)units * kMaxLevels, (void **)&kcnt[t]))) return rc;
    }
    if ((rc = vs_arena_get(ctx, "grid.fkp", sizeof(float4) * (size_t)frames * kp_cap, (void **)&fkp))) return rc;
    hipStream_t st = ctx->stream;

    {   // :32 outlines into the caller's image + gray of the result (ORB converts BGR ROIs to gray)
        VsProfScope ps(ctx, "grid_outline_gray_kernel");
        grid_outline_gray_kernel<<<dim3(vs_div_up(w * h, 256), frames), 256, 0, st>>>(bgr, w, h, stride, cw, ch, ncols, nrows, gray);
    }
    {   // cell pyramids
        VsProfScope ps(ctx, "grid_pyramid_kernels");
        const int e0 = (LC.lw[0] + 2 * LC.border) * (LC.lh[0] + 2 * LC.border);
        pyr_level0_kernel<<<dim3(vs_div_up(e0, 256), units), 256, 0, st>>>(gray, w, h, cw, ch, ncols, nrows, LC, pyrc);
        for (int l = 1; l < nlevels; l++) {
            const int e = (LC.lw[l] + 2 * LC.border) * (LC.lh[l] + 2 * LC.border);
            pyr_resize_kernel<<<dim3(vs_div_up(e, 256), units), 256, 0, st>>>(LC, l, pyrc);
        }
    }
    {
        VsProfScope ps(ctx, "fast_score_kernel");
        fast_score_kernel<<<dim3(vs_div_up(LC.roi_prefix[nlevels], 256), units), 256, 0, st>>>(LC, pyrc, mmap);
    }
    // selection work arrays: LDS for lists of up to 4095 keypoints, a per-slot global scratch beyond that
    const int lds_entries = std::min(SL.cap[0] + 1, 4096);
    const size_t sel_lds = (size_t)lds_entries * 16;
    short2 *kxy_tmp = nullptr;
    float *g_key = nullptr;
    int *g_pay = nullptr, *g_sl = nullptr, *g_sr = nullptr;
    if ((rc = vs_arena_get(ctx, "grid.kxy_tmp", sizeof(short2) * slots, (void **)&kxy_tmp))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.sel_key", sizeof(float) * slots, (void **)&g_key))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.sel_pay", sizeof(int) * slots, (void **)&g_pay))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.sel_sl", sizeof(int) * slots, (void **)&g_sl))) return rc;
    if ((rc = vs_arena_get(ctx, "grid.sel_sr", sizeof(int) * (slots + (size_t)units * kMaxLevels + 1), (void **)&g_sr))) return rc;
    const int thr[2] = {20, 5};   // src/Frame.cpp:22-23
    for (int t = 0; t < 2; t++) {
        VsProfScope ps(ctx, t ? "orb_detect_t5_kernels" : "orb_detect_t20_kernels");
        fast_collect_kernel<<<dim3(nlevels, units), 256, 0, st>>>(LC, SL, mmap, thr[t], kxy[t], kresp[t], kcnt[t]);
        retain_best_kernel<<<dim3(nlevels, units), 64, sel_lds, st>>>(LC, SL, 2, kxy[t], kxy_tmp, kresp[t], kcnt[t], lds_entries,
                                                                      g_key, g_pay, g_sl, g_sr);
        const int maxk = vs_div_up(SL.cap[0], 256);
        harris_kernel<<<dim3(maxk, nlevels, units), 256, 0, st>>>(LC, SL, pyrc, kxy_tmp, kresp[t], kcnt[t]);
        retain_best_kernel<<<dim3(nlevels, units), 64, sel_lds, st>>>(LC, SL, 1, kxy_tmp, kxy[t], kresp[t], kcnt[t], lds_entries,
                                                                      g_key, g_pay, g_sl, g_sr);
        ic_angle_kernel<<<dim3(maxk, nlevels, units), 256, 0, st>>>(LC, SL, U, pyrc, kxy[t], kcnt[t], kfin[t]);
    }
    {
        VsProfScope ps(ctx, "grid_assemble_kernel");
        grid_assemble_kernel<<<frames, 256, 0, st>>>(LC, SL, cells, nrows, cw, ch, w, h, nfeatures, kfin[0], kcnt[0], kfin[1],
                                                     kcnt[1], kp_cap, fkp, n_out);
    }
    {   // ORB::compute (:43): frame pyramid of the outlined image, per-level blur, steered BRIEF
        VsProfScope ps(ctx, "orb_compute_kernels");
        const int e0 = (LF.lw[0] + 2 * LF.border) * (LF.lh[0] + 2 * LF.border);
        pyr_level0_kernel<<<dim3(vs_div_up(e0, 256), frames), 256, 0, st>>>(gray, w, h, w, h, 1, 1, LF, pyrf);
        for (int l = 1; l < nlevels; l++) {
            const int e = (LF.lw[l] + 2 * LF.border) * (LF.lh[l] + 2 * LF.border);
            pyr_resize_kernel<<<dim3(vs_div_up(e, 256), frames), 256, 0, st>>>(LF, l, pyrf);
        }
        VS_HIP(ctx, hipMemcpyAsync(pyrb, pyrf, frame_pyr * frames, hipMemcpyDeviceToDevice, st));
        pyr_blur_kernel<<<dim3(vs_div_up(LF.roi_prefix[nlevels], 256), frames), 256, 0, st>>>(LF, pyrf, pyrb);
        orb_desc_kernel<<<dim3(vs_div_up(kp_cap, 8), frames), 256, 0, st>>>(LF, pyrb, fkp, n_out, kp_cap, pattern, desc, xy,
                                                                            angle_octave);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

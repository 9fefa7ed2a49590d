// ORACLE (test infrastructure only) — extract_features(Frame&, nrows, ncols) restated.
//
// Follows /root/reference/src/Frame.cpp:16-51 (the grid ORB/FAST extractor; its only call site is
// commented out at src/vslam.cpp:63, but it is the "ORB/FAST" the north star names):
//   per cell (columns outer, rows inner): cv::rectangle(image, cell, black) :32;
//   ORB(500, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, fastThreshold 20)->detect(cell) :33;
//   if fewer than 500: the fastThreshold-5 detector's result REPLACES it :34-36;
//   keypoints shifted by the cell origin :37-40; then ORB::compute(whole image, all keypoints) :43.
// Everything below the loop is OpenCV-internal and restated from OpenCV 4.x's published code
// [OpenCV, from memory] — PARITY UNPINNED:
//   features2d/src/orb.cpp      pyramid layout, computeKeyPoints, HarrisResponses, ICAngles,
//                               computeOrbDescriptors, regrouping of unsorted keypoints by level
//   features2d/src/fast.cpp     FAST_t<16> with cornerScore<16> and 3x3 non-max suppression
//   features2d/src/keypoint.cpp KeyPointsFilter::runByImageBorder / retainBest (std::nth_element +
//                               std::partition, so tie order is libstdc++'s)
//   imgproc/src/resize.cpp      INTER_LINEAR_EXACT for 8U: Q8 coefficients on both axes, Q16
//                               accumulate, round half up
//   core mathfuncs              fastAtan2's degree-7 polynomial
// Pins shared with the HIP kernels: cos/sin of the keypoint angle come from vso::sincos_deg_pinned
// (OpenCV calls libm's cos/sin on a float; libm's last bit is unspecified), see DESIGN.md.
#include "vso.h"
#include "vso_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace vso {

// ---- pinned sin/cos of an angle given in degrees as float (result rounded to float).
// Reduction to [-45, 45] degrees is exact in float arithmetic on multiples of 90; the kernels carry
// the same code.  Polynomials are Taylor series in double, far below float resolution.
void sincos_deg_pinned(float angle_deg, float *s_out, float *c_out) {
    // OpenCV: angle *= (float)(CV_PI/180.f); a = (float)cos(angle); b = (float)sin(angle)
    const float ar = angle_deg * (float)(3.14159265358979323846 / 180.f);
    double x = (double)ar;
    // quadrant reduction in double: k = nearest integer to x / (pi/2)
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const double kf = std::nearbyint(x * two_over_pi);
    const int k = (int)kf;
    double r = (x - kf * pio2_hi) - kf * pio2_lo;
    const double r2 = r * r;
    double sp = r * (1.0 + r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800))))));
    double cp = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600))))));
    double s, c;
    switch (k & 3) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

}  // namespace vso

namespace {

struct KeyPt {
    float x, y, size, angle, response;
    int octave;
};

static inline int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}
static inline int cv_round(double v) { return (int)std::lrint(v); }

// ---------------------------------------------------------------- FAST-9/16 (fast.cpp)
static const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: largest threshold for which the pixel is still a 9-contiguous corner, minus 1 rule as in OpenCV
static int corner_score16(const uint8_t *ptr, const int *pixel, int threshold) {
    const int N = 25;
    int d[N];
    const int v = ptr[0];
    for (int k = 0; k < N; k++) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// FAST_t<16>(img, keypoints, threshold, nonmax = true): keypoints in raster order, response = score
void fast9_16(const uint8_t *img, int w, int h, int step, int threshold, std::vector<KeyPt> &out) {
    out.clear();
    if (w < 7 || h < 7) return;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> score((size_t)w * h, 0);
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const uint8_t *ptr = img + (size_t)i * step + j;
            const int v = ptr[0];
            bool corner = false;
            for (int pass = 0; pass < 2 && !corner; pass++) {
                int count = 0;
                for (int k = 0; k < 25; k++) {
                    const int x = ptr[pixel[k]];
                    const bool hit = pass == 0 ? (x < v - threshold) : (x > v + threshold);
                    if (hit) {
                        if (++count > 8) {
                            corner = true;
                            break;
                        }
                    } else {
                        count = 0;
                    }
                }
            }
            if (corner) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    // 3x3 non-max suppression on the score rows (non-corners score 0), raster order
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const int s = score[(size_t)i * w + j];
            if (!s && true) {
                // a detected corner always has score >= threshold; score 0 can only be a corner when
                // threshold == 0, and then it can never be strictly greater than its neighbours
                continue;
            }
            const uint8_t *p = &score[(size_t)(i - 1) * w + j], *c = &score[(size_t)i * w + j], *n = &score[(size_t)(i + 1) * w + j];
            if (s > c[-1] && s > c[1] && s > p[-1] && s > p[0] && s > p[1] && s > n[-1] && s > n[0] && s > n[1])
                out.push_back({(float)j, (float)i, 7.f, -1.f, (float)s, 0});
        }
}

// ---------------------------------------------------------------- resize INTER_LINEAR_EXACT, 8U
void linear_coeffs(int dst_n, int src_n, std::vector<int> &ofs, std::vector<int> &c0, std::vector<int> &c1) {
    ofs.resize(dst_n);
    c0.resize(dst_n);
    c1.resize(dst_n);
    const double inv_scale = (double)dst_n / src_n;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst_n; d++) {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        if (ival >= 0 && src_n > 1) {
            if (ival < src_n - 1) {
                const int a = (int)std::lrint((fval - (double)ival) * 256.0);   // Q8, round to nearest
                c1[d] = a;
                c0[d] = 256 - a;
            } else {
                ival = src_n - 2;
                c0[d] = 0;
                c1[d] = 256;
            }
        } else {
            ival = 0;
            c0[d] = 256;
            c1[d] = 0;
        }
        ofs[d] = ival;
    }
}

void resize_linear_exact(const uint8_t *src, int sw, int sh, int sstep, uint8_t *dst, int dw, int dh, int dstep) {
    std::vector<int> xo, x0, x1, yo, y0, y1;
    linear_coeffs(dw, sw, xo, x0, x1);
    linear_coeffs(dh, sh, yo, y0, y1);
    for (int y = 0; y < dh; y++) {
        const uint8_t *r0 = src + (size_t)yo[y] * sstep;
        const uint8_t *r1 = src + (size_t)std::min(yo[y] + 1, sh - 1) * sstep;
        for (int x = 0; x < dw; x++) {
            const int xa = xo[x], xb = std::min(xa + 1, sw - 1);
            const uint32_t h0 = (uint32_t)r0[xa] * x0[x] + (uint32_t)r0[xb] * x1[x];   // Q8.8
            const uint32_t h1 = (uint32_t)r1[xa] * x0[x] + (uint32_t)r1[xb] * x1[x];
            const uint32_t v = h0 * y0[y] + h1 * y1[y];                                // Q8.16
            dst[(size_t)y * dstep + x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
}

// ---------------------------------------------------------------- pyramid (orb.cpp layout)
struct Pyramid {
    int nlevels = 0, border = 0, bufw = 0, bufh = 0;
    std::vector<int> lx, ly, lw, lh;   // layerInfo
    std::vector<float> scale;
    std::vector<uint8_t> buf;
    const uint8_t *level_ptr(int l) const { return &buf[(size_t)ly[l] * bufw + lx[l]]; }
    uint8_t *level_ptr(int l) { return &buf[(size_t)ly[l] * bufw + lx[l]]; }
};

void gaussian7_inplace_roi(uint8_t *img, int w, int h, int step);   // below

void build_pyramid(const uint8_t *gray, int w, int h, int step, int nlevels, double scaleFactor, Pyramid &P) {
    const int patchSize = 31, edgeThreshold = 31, HARRIS_BLOCK_SIZE = 9;
    const int halfPatchSize = patchSize / 2;
    const int descPatchSize = (int)std::ceil(halfPatchSize * std::sqrt(2.0));
    P.border = std::max(edgeThreshold, std::max(descPatchSize, HARRIS_BLOCK_SIZE / 2)) + 1;
    const int border = P.border;
    P.nlevels = nlevels;
    P.lx.resize(nlevels); P.ly.resize(nlevels); P.lw.resize(nlevels); P.lh.resize(nlevels); P.scale.resize(nlevels);
    P.bufw = ((w + border * 2) + 15) & ~15;
    int level_dy = h + border * 2, ox = 0, oy = 0;
    for (int l = 0; l < nlevels; l++) {
        const float sc = (float)std::pow(scaleFactor, (double)l);
        P.scale[l] = sc;
        const float inv = 1.0f / sc;
        const int sw = cv_round(w * inv), sh = cv_round(h * inv);
        const int ww = sw + border * 2, wh = sh + border * 2;
        if (ox + ww > P.bufw) {
            ox = 0;
            oy += level_dy;
            level_dy = wh;
        }
        P.lx[l] = ox + border; P.ly[l] = oy + border; P.lw[l] = sw; P.lh[l] = sh;
        ox += ww;
    }
    P.bufh = oy + level_dy;
    P.buf.assign((size_t)P.bufw * P.bufh, 0);
    for (int l = 0; l < nlevels; l++) {
        uint8_t *cur = P.level_ptr(l);
        if (l == 0) {
            for (int y = 0; y < h; y++) std::memcpy(cur + (size_t)y * P.bufw, gray + (size_t)y * step, w);
        } else {
            resize_linear_exact(P.level_ptr(l - 1), P.lw[l - 1], P.lh[l - 1], P.bufw, cur, P.lw[l], P.lh[l], P.bufw);
        }
        // copyMakeBorder(..., BORDER_REFLECT_101): fill the 32-px frame around the level
        const int lw = P.lw[l], lh = P.lh[l];
        for (int y = -border; y < lh + border; y++)
            for (int x = -border; x < lw + border; x++) {
                if (y >= 0 && y < lh && x >= 0 && x < lw) continue;
                cur[(ptrdiff_t)y * P.bufw + x] = cur[(ptrdiff_t)reflect101(y, lh) * P.bufw + reflect101(x, lw)];
            }
    }
}

// ---------------------------------------------------------------- KeyPointsFilter
void run_by_image_border(std::vector<KeyPt> &k, int w, int h, int border) {
    if (border <= 0) return;
    if (h <= border * 2 || w <= border * 2) {
        k.clear();
        return;
    }
    std::vector<KeyPt> out;
    for (const KeyPt &p : k)
        if (p.x >= (float)border && p.x < (float)(w - border) && p.y >= (float)border && p.y < (float)(h - border)) out.push_back(p);
    k.swap(out);
}

void retain_best(std::vector<KeyPt> &k, int n_points) {
    if (n_points >= 0 && k.size() > (size_t)n_points) {
        if (n_points == 0) {
            k.clear();
            return;
        }
        std::nth_element(k.begin(), k.begin() + n_points - 1, k.end(),
                         [](const KeyPt &a, const KeyPt &b) { return a.response > b.response; });
        const float ambiguous = k[n_points - 1].response;
        auto new_end = std::partition(k.begin() + n_points, k.end(), [ambiguous](const KeyPt &p) { return p.response >= ambiguous; });
        k.resize(new_end - k.begin());
    }
}

// ---------------------------------------------------------------- Harris, IC angle, fastAtan2
void harris_responses(const Pyramid &P, std::vector<KeyPt> &pts, int blockSize, float harris_k) {
    const int step = P.bufw, r = blockSize / 2;
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    for (KeyPt &kp : pts) {
        const int x0 = cv_round(kp.x), y0 = cv_round(kp.y), z = kp.octave;
        const uint8_t *ptr0 = P.buf.data() + (size_t)(y0 - r + P.ly[z]) * step + x0 - r + P.lx[z];
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
            for (int j = 0; j < blockSize; j++) {
                const uint8_t *ptr = ptr0 + i * step + j;
                const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
                const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        kp.response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
}

float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = std::abs(x), ay = std::abs(y);
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

void umax_table(int half, std::vector<int> &umax) {
    umax.assign(half + 2, 0);
    const int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

void ic_angles(const Pyramid &P, std::vector<KeyPt> &pts, const std::vector<int> &umax, int half_k) {
    const int step = P.bufw;
    for (KeyPt &kp : pts) {
        const int z = kp.octave;
        const uint8_t *center = P.buf.data() + (size_t)(cv_round(kp.y) + P.ly[z]) * step + cv_round(kp.x) + P.lx[z];
        int m_01 = 0, m_10 = 0;
        for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
        for (int v = 1; v <= half_k; ++v) {
            int v_sum = 0;
            const int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                const int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        kp.angle = fast_atan2((float)m_01, (float)m_10);
    }
}

// ---------------------------------------------------------------- ORB::detect (computeKeyPoints)
void orb_detect(const uint8_t *gray, int w, int h, int step, int nfeatures, double scaleFactor, int nlevels,
                int edgeThreshold, int patchSize, int fastThreshold, std::vector<KeyPt> &all) {
    all.clear();
    Pyramid P;
    build_pyramid(gray, w, h, step, nlevels, scaleFactor, P);
    std::vector<int> per_level(nlevels);
    const float factor = (float)(1.0 / scaleFactor);
    float ndesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        per_level[l] = cv_round(ndesired);
        sum += per_level[l];
        ndesired *= factor;
    }
    per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    const int half = patchSize / 2;
    std::vector<int> umax;
    umax_table(half, umax);

    std::vector<KeyPt> kps;
    std::vector<int> counters(nlevels);
    for (int l = 0; l < nlevels; l++) {
        fast9_16(P.level_ptr(l), P.lw[l], P.lh[l], P.bufw, fastThreshold, kps);
        run_by_image_border(kps, P.lw[l], P.lh[l], edgeThreshold);
        retain_best(kps, 2 * per_level[l]);   // HARRIS_SCORE keeps twice as many first
        counters[l] = (int)kps.size();
        const float sf = P.scale[l];
        for (KeyPt &k : kps) {
            k.octave = l;
            k.size = patchSize * sf;
        }
        all.insert(all.end(), kps.begin(), kps.end());
    }
    if (all.empty()) return;
    harris_responses(P, all, 7, 0.04f);
    std::vector<KeyPt> culled;
    int offset = 0;
    for (int l = 0; l < nlevels; l++) {
        kps.assign(all.begin() + offset, all.begin() + offset + counters[l]);
        offset += counters[l];
        retain_best(kps, per_level[l]);
        culled.insert(culled.end(), kps.begin(), kps.end());
    }
    all.swap(culled);
    ic_angles(P, all, umax, half);
    for (KeyPt &k : all) {
        const float sc = P.scale[k.octave];
        k.x *= sc;
        k.y *= sc;
    }
}

// GaussianBlur 7x7 sigma 2 (Q8 taps 18,34,48,56,48,34,18) on a level inside the bordered pyramid:
// the filter reads the reflect-filled frame around the level (borders are >= 32 px wide)
void gaussian7_inplace_roi(uint8_t *img, int w, int h, int step) {
    static const int kq[7] = {18, 34, 48, 56, 48, 34, 18};
    std::vector<uint16_t> rowp((size_t)w * (h + 6));
    for (int y = -3; y < h + 3; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kq[k + 3] * (int)img[(ptrdiff_t)y * step + x + k];
            rowp[(size_t)(y + 3) * w + x] = (uint16_t)s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < 7; k++) s += (uint32_t)kq[k] * rowp[(size_t)(y + k) * w + x];
            img[(size_t)y * step + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
}

// ORB::compute for keypoints carrying octave and angle (unsorted by level -> regrouped by level)
void orb_compute(const uint8_t *gray, int w, int h, int step, std::vector<KeyPt> &kps, double scaleFactor,
                 const int8_t *pattern, std::vector<uint8_t> &desc) {
    run_by_image_border(kps, w, h, 31);
    desc.clear();
    if (kps.empty()) return;
    int nlevels = 0;
    bool sorted = true;
    for (size_t i = 0; i < kps.size(); i++) {
        nlevels = std::max(nlevels, std::max(kps[i].octave, 0));
        if (i > 0 && kps[i].octave < kps[i - 1].octave) sorted = false;
    }
    nlevels++;
    if (!sorted) {   // orb.cpp regroups by level, keeping the order inside each level
        std::vector<KeyPt> re;
        for (int l = 0; l < nlevels; l++)
            for (const KeyPt &k : kps)
                if (k.octave == l) re.push_back(k);
        kps.swap(re);
    }
    Pyramid P;
    build_pyramid(gray, w, h, step, nlevels, scaleFactor, P);
    for (int l = 0; l < nlevels; l++) gaussian7_inplace_roi(P.level_ptr(l), P.lw[l], P.lh[l], P.bufw);
    desc.assign(kps.size() * 32, 0);
    for (size_t j = 0; j < kps.size(); j++) {
        const KeyPt &kp = kps[j];
        const float scale = 1.f / P.scale[kp.octave];
        float a, b;
        vso::sincos_deg_pinned(kp.angle, &b, &a);
        const uint8_t *center = P.buf.data() + (size_t)(cv_round(kp.y * scale) + P.ly[kp.octave]) * P.bufw +
                                cv_round(kp.x * scale) + P.lx[kp.octave];
        for (int byte = 0; byte < 32; byte++) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const int8_t *pp = pattern + (size_t)(byte * 8 + bit) * 4;
                int t[2];
                for (int e = 0; e < 2; e++) {
                    const float fx = (float)pp[2 * e], fy = (float)pp[2 * e + 1];
                    const float rx = fx * a - fy * b, ry = fx * b + fy * a;
                    t[e] = center[(ptrdiff_t)std::lrintf(ry) * P.bufw + std::lrintf(rx)];
                }
                val |= (t[0] < t[1]) << bit;
            }
            desc[j * 32 + byte] = (uint8_t)val;
        }
    }
}

void bgr_to_gray(const uint8_t *bgr, int w, int h, int stride, std::vector<uint8_t> &g) {
    g.resize((size_t)w * h);
    vso_bgr2gray(bgr, w, h, stride, g.data());
}

}  // namespace

extern "C" {

int vso_sincos_deg(float angle_deg, float *s, float *c) {
    vso::sincos_deg_pinned(angle_deg, s, c);
    return 0;
}

int vso_fast9_16(const uint8_t *gray, int w, int h, int threshold, float *out_xys, int cap, int32_t *out_n) {
    std::vector<KeyPt> k;
    fast9_16(gray, w, h, w, threshold, k);
    *out_n = (int32_t)k.size();
    for (int i = 0; i < (int)k.size() && i < cap; i++) {
        out_xys[3 * i] = k[i].x;
        out_xys[3 * i + 1] = k[i].y;
        out_xys[3 * i + 2] = k[i].response;
    }
    return 0;
}

int vso_resize_linear_exact(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
    resize_linear_exact(src, sw, sh, sw, dst, dw, dh, dw);
    return 0;
}

// ORB(nfeatures, 1.2, 8, 31, 0, 2, HARRIS, 31, fastThreshold)->detect(gray): 6 floats per keypoint
// (x, y, size, angle, response, octave)
int vso_orb_detect(const uint8_t *gray, int w, int h, int nfeatures, int fast_threshold, float *out_kp, int cap,
                   int32_t *out_n) {
    std::vector<KeyPt> k;
    orb_detect(gray, w, h, w, nfeatures, 1.2, 8, 31, 31, fast_threshold, k);
    *out_n = (int32_t)k.size();
    for (int i = 0; i < (int)k.size() && i < cap; i++) {
        float *o = out_kp + 6 * i;
        o[0] = k[i].x; o[1] = k[i].y; o[2] = k[i].size; o[3] = k[i].angle; o[4] = k[i].response; o[5] = (float)k[i].octave;
    }
    return 0;
}

// extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51.  bgr is MODIFIED (cell borders drawn, :32).
// Outputs: points (2 floats, order = ORB::compute's level-grouped order), descriptors, count, and the
// per-keypoint (angle, octave) for inspection.
int vso_extract_features_grid(uint8_t *bgr, int w, int h, int stride, int nrows, int ncols, const int8_t *pattern,
                              float *out_xy, uint8_t *out_desc, float *out_angle_octave, int cap, int32_t *out_n) {
    const int nfeatures = 500;
    const int cw = w / ncols, ch = h / nrows;                     // :20
    std::vector<KeyPt> keypoints;
    std::vector<uint8_t> cell_gray;
    for (int i = 0; i < ncols; i++)
        for (int j = 0; j < nrows; j++) {
            const int sx = i * cw, sy = j * ch;
            // cv::rectangle(image, Rect(sx,sy,cw,ch), Scalar(0,0,0)): 1-px outline, corners tl and br-1
            for (int x = sx; x < sx + cw; x++)
                for (int c = 0; c < 3; c++) {
                    bgr[(size_t)sy * stride + 3 * x + c] = 0;
                    bgr[(size_t)(sy + ch - 1) * stride + 3 * x + c] = 0;
                }
            for (int y = sy; y < sy + ch; y++)
                for (int c = 0; c < 3; c++) {
                    bgr[(size_t)y * stride + 3 * sx + c] = 0;
                    bgr[(size_t)y * stride + 3 * (sx + cw - 1) + c] = 0;
                }
            // ORB::detect on the ROI: cvtColor(BGR2GRAY) of the cell
            bgr_to_gray(bgr + (size_t)sy * stride + 3 * sx, cw, ch, stride, cell_gray);
            std::vector<KeyPt> temp;
            orb_detect(cell_gray.data(), cw, ch, cw, nfeatures, 1.2, 8, 31, 31, 20, temp);      // :33
            if ((int)temp.size() < nfeatures)                                                   // :34
                orb_detect(cell_gray.data(), cw, ch, cw, nfeatures, 1.2, 8, 31, 31, 5, temp);   // :35
            for (KeyPt k : temp) {                                                              // :37-40
                k.x = sx + k.x;
                k.y = sy + k.y;
                keypoints.push_back(k);
            }
        }
    std::vector<uint8_t> gray, desc;
    bgr_to_gray(bgr, w, h, stride, gray);
    orb_compute(gray.data(), w, h, w, keypoints, 1.2, pattern, desc);                           // :43
    *out_n = (int32_t)keypoints.size();
    for (int i = 0; i < (int)keypoints.size() && i < cap; i++) {                                // :47-49
        out_xy[2 * i] = keypoints[i].x;
        out_xy[2 * i + 1] = keypoints[i].y;
        if (out_angle_octave) {
            out_angle_octave[2 * i] = keypoints[i].angle;
            out_angle_octave[2 * i + 1] = (float)keypoints[i].octave;
        }
        std::memcpy(out_desc + (size_t)i * 32, &desc[(size_t)i * 32], 32);
    }
    return 0;
}

}  // extern "C"

// ORACLE (test infrastructure only) — extract_features(Frame&) restated.
//
// Follows /root/reference/src/Frame.cpp:53-80:
//   cvtColor(BGR2GRAY) :56 -> goodFeaturesToTrack(gray, pts, 3000, 0.01, 3) :61 ->
//   KeyPoint(p, 20) :64-67 -> ORB::create()->compute(gray, kps, desc) :57,68 ->
//   points from surviving keypoints :69-72 -> construct_kdtree :76.
// Every step here is OpenCV-internal; OpenCV is absent, so the steps restate OpenCV 4.x's
// published built-in algorithms [OpenCV, from memory] with each implementation-defined float
// order written out.  PARITY UNPINNED.
//   cvtColor 8U           color_rgb.simd.hpp RGB2Gray<uchar>: (b*3735 + g*19235 + r*9798 + 2^14) >> 15
//   cornerMinEigenVal     corner.cpp cornerEigenValsVecs: Sobel 3x3 to CV_32F, scale 1/(4*3*255)
//                         folded into the smoothing taps; dx*dx, dx*dy, dy*dy; 3x3 unnormalised
//                         box whose sums run in double (exact here, see box3_exact below);
//                         (a+c) - sqrt((a-c)^2 + b^2) with a = 0.5*Sxx, b = Sxy, c = 0.5*Syy.
//   goodFeaturesToTrack   featureselect.cpp: max, THRESH_TOZERO at (float)(max*quality),
//                         3x3 dilate equality, interior pixels only, sort by (value desc,
//                         address desc), greedy min-distance with a cell grid, first maxCorners.
//   GaussianBlur 7x7 s=2  smooth.simd.hpp fixed-point path for 8U: Q8 taps by error diffusion
//                         (18,34,48,56,48,34,18), Q16 accumulate, round half up.
//   ORB::compute          orb.cpp: runByImageBorder(edgeThreshold 31), one level, steered BRIEF
//                         x = px*a - py*b, y = px*b + py*a, cvRound, t0 < t1, WTA_K = 2.
//                         The 256-pair learned pattern is an INPUT table here; the tests pass
//                         ORB's own (tests/golden/brief_pattern_31.npy) unless they say otherwise.
#include "vso.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

static inline int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

// Sobel with the scale folded into the smoothing taps (deriv.cpp: "if dx == 0 kx *= scale else
// ky *= scale"), float kernels, 8U source.
//  Dx = column-smooth( row-diff ):   hx = g(x+1) - g(x-1) (exact);  Dx = hx(y)*k0 + (hx(y-1)+hx(y+1))*k1
//  Dy = column-diff( row-smooth ):   R  = g(x)*k0 + (g(x-1)+g(x+1))*k1;  Dy = R(y+1) - R(y-1)
//  with k1 = (float)scale, k0 = 2*k1, each product rounded to float before the add.
void sobel_pair(const uint8_t *g, int w, int h, std::vector<float> &Dx, std::vector<float> &Dy) {
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);   // (1 << (ksize-1)) * block * 255
    const float k1 = (float)scale, k0 = 2.0f * k1;
    std::vector<float> hx((size_t)w * h), R((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = g + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            hx[(size_t)y * w + x] = (float)((int)row[xp] - (int)row[xm]);
            const float a = (float)row[x] * k0;
            const float b = (float)((int)row[xm] + (int)row[xp]) * k1;
            R[(size_t)y * w + x] = a + b;
        }
    }
    Dx.resize((size_t)w * h);
    Dy.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; x++) {
            const float a = hx[(size_t)y * w + x] * k0;
            const float b = (hx[(size_t)ym * w + x] + hx[(size_t)yp * w + x]) * k1;
            Dx[(size_t)y * w + x] = a + b;
            Dy[(size_t)y * w + x] = R[(size_t)yp * w + x] - R[(size_t)ym * w + x];
        }
    }
}

// boxFilter(cov, cov, CV_32F, 3x3, normalize=false, REFLECT_101): OpenCV's RowSum<float,double>
// then ColumnSum<double,float> keep running (sliding) sums in double and round to float once.
// Non-zero products here are floats in about [2^-24, 2^-3) (|Dx|,|Dy| are 0 or >= ~scale), so
// they are multiples of 2^-47 and the double sums are exact; only the rare "numerically zero"
// derivative (two roundings of the same real differing by an ulp) breaks that, far below float
// resolution of the sum.  The oracle pins the order as the non-sliding form of the same
// structure:  r(y) = (c(x-1,y) + c(x,y)) + c(x+1,y);  S = (r(y-1) + r(y)) + r(y+1), in double.
void box3(const std::vector<float> &c, int w, int h, std::vector<float> &out) {
    std::vector<double> r((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const float *row = &c[(size_t)y * w];
        for (int x = 0; x < w; x++)
            r[(size_t)y * w + x] = ((double)row[reflect101(x - 1, w)] + (double)row[x]) +
                                   (double)row[reflect101(x + 1, w)];
    }
    out.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const double *r0 = &r[(size_t)reflect101(y - 1, h) * w], *r1 = &r[(size_t)y * w],
                     *r2 = &r[(size_t)reflect101(y + 1, h) * w];
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = (float)((r0[x] + r1[x]) + r2[x]);
    }
}

void min_eigen(const uint8_t *gray, int w, int h, float *eig) {
    std::vector<float> Dx, Dy;
    sobel_pair(gray, w, h, Dx, Dy);
    std::vector<float> cxx((size_t)w * h), cxy((size_t)w * h), cyy((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        cxx[i] = Dx[i] * Dx[i];
        cxy[i] = Dx[i] * Dy[i];
        cyy[i] = Dy[i] * Dy[i];
    }
    std::vector<float> sxx, sxy, syy;
    box3(cxx, w, h, sxx);
    box3(cxy, w, h, sxy);
    box3(cyy, w, h, syy);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const float a = sxx[i] * 0.5f;               // calcMinEigenVal
        const float b = sxy[i];
        const float c = syy[i] * 0.5f;
        const float amc = a - c;
        const float t = amc * amc + b * b;           // two rounded products, one add
        eig[i] = (a + c) - std::sqrt(t);
    }
}

struct CornerOrder {
    // featureselect.cpp greaterThanPtr: value descending, then address descending
    const float *base;
    bool operator()(int a, int b) const {
        const float va = base[a], vb = base[b];
        return (va > vb) ? true : (va < vb) ? false : (a > b);
    }
};

int good_features(const uint8_t *gray, int w, int h, int maxCorners, double qualityLevel,
                  double minDistance, float *out_xy) {
    std::vector<float> eig((size_t)w * h), tmp((size_t)w * h);
    min_eigen(gray, w, h, eig.data());
    double maxVal = 0;
    {
        float mx = eig[0];
        for (size_t i = 1; i < (size_t)w * h; i++) mx = eig[i] > mx ? eig[i] : mx;   // minMaxLoc
        maxVal = mx;
    }
    const float thr = (float)(maxVal * qualityLevel);
    for (size_t i = 0; i < (size_t)w * h; i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;   // THRESH_TOZERO
    for (int y = 0; y < h; y++)                                                        // dilate 3x3
        for (int x = 0; x < w; x++) {
            float m = eig[(size_t)y * w + x];
            for (int dy = -1; dy <= 1; dy++) {
                const int yy = y + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= w) continue;
                    const float v = eig[(size_t)yy * w + xx];
                    m = v > m ? v : m;
                }
            }
            tmp[(size_t)y * w + x] = m;
        }
    std::vector<int> cand;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            const float val = eig[(size_t)y * w + x];
            if (val != 0 && val == tmp[(size_t)y * w + x]) cand.push_back(y * w + x);
        }
    std::sort(cand.begin(), cand.end(), CornerOrder{eig.data()});

    int ncorners = 0;
    const size_t total = cand.size();
    if (minDistance >= 1) {
        const int cell_size = (int)std::lrint(minDistance);          // cvRound
        const int grid_width = (w + cell_size - 1) / cell_size;
        const int grid_height = (h + cell_size - 1) / cell_size;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)grid_width * grid_height);
        const float minDist2 = (float)(minDistance * minDistance);
        for (size_t i = 0; i < total; i++) {
            const int y = cand[i] / w, x = cand[i] - y * w;
            bool good = true;
            const int x_cell = x / cell_size, y_cell = y / cell_size;
            const int x1 = std::max(0, x_cell - 1), y1 = std::max(0, y_cell - 1);
            const int x2 = std::min(grid_width - 1, x_cell + 1), y2 = std::min(grid_height - 1, y_cell + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (const auto &p : grid[(size_t)yy * grid_width + xx]) {
                        const float dx = x - p.first, dy = y - p.second;
                        if (dx * dx + dy * dy < minDist2) {
                            good = false;
                            break;
                        }
                    }
            if (good) {
                grid[(size_t)y_cell * grid_width + x_cell].push_back({(float)x, (float)y});
                out_xy[2 * ncorners] = (float)x;
                out_xy[2 * ncorners + 1] = (float)y;
                ++ncorners;
                if (maxCorners > 0 && ncorners == maxCorners) break;
            }
        }
    } else {
        for (size_t i = 0; i < total; i++) {
            const int y = cand[i] / w, x = cand[i] - y * w;
            out_xy[2 * ncorners] = (float)x;
            out_xy[2 * ncorners + 1] = (float)y;
            ++ncorners;
            if (maxCorners > 0 && ncorners == maxCorners) break;
        }
    }
    return ncorners;
}

void gaussian7(const uint8_t *g, int w, int h, uint8_t *out) {
    static const int kq[7] = {18, 34, 48, 56, 48, 34, 18};   // Q8, sums to 256
    std::vector<uint16_t> rowp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kq[k + 3] * (int)g[(size_t)y * w + reflect101(x + k, w)];
            rowp[(size_t)y * w + x] = (uint16_t)s;            // <= 255*256
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; k++) s += (uint32_t)kq[k + 3] * rowp[(size_t)reflect101(y + k, h) * w + x];
            out[(size_t)y * w + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
}

int orb_describe(const uint8_t *blurred, int w, int h, const float *xy, int n, float ca, float sa,
                 const int8_t *pattern, uint8_t *desc, int32_t *keep) {
    const int border = 31;                                    // edgeThreshold of ORB::create()
    int kept = 0;
    if (h <= border * 2 || w <= border * 2) return 0;         // runByImageBorder clears everything
    for (int i = 0; i < n; i++) {
        const float px = xy[2 * i], py = xy[2 * i + 1];
        if (!(px >= (float)border && px < (float)(w - border) && py >= (float)border &&
              py < (float)(h - border)))
            continue;
        const int cx = (int)std::lrintf(px), cy = (int)std::lrintf(py);   // cvRound(pt*scale), scale 1
        uint8_t *d = desc + (size_t)kept * 32;
        for (int byte = 0; byte < 32; byte++) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const int8_t *pp = pattern + (size_t)(byte * 8 + bit) * 4;
                int t[2];
                for (int e = 0; e < 2; e++) {
                    const float fx = (float)pp[2 * e], fy = (float)pp[2 * e + 1];
                    const float rx = fx * ca - fy * sa;       // GET_VALUE: x = p.x*a - p.y*b
                    const float ry = fx * sa + fy * ca;       //            y = p.x*b + p.y*a
                    const int ix = (int)std::lrintf(rx), iy = (int)std::lrintf(ry);
                    t[e] = blurred[(size_t)(cy + iy) * w + (cx + ix)];
                }
                val |= (t[0] < t[1]) << bit;
            }
            d[byte] = (uint8_t)val;
        }
        keep[kept++] = i;
    }
    return kept;
}

}  // namespace

extern "C" {

int vso_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray) {
    if (!bgr || !gray || w <= 0 || h <= 0 || stride < 3 * w) return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = bgr + (size_t)y * stride;
        for (int x = 0; x < w; x++)
            gray[(size_t)y * w + x] =
                (uint8_t)((s[3 * x] * 3735 + s[3 * x + 1] * 19235 + s[3 * x + 2] * 9798 + (1 << 14)) >> 15);
    }
    return 0;
}

int vso_min_eigen(const uint8_t *gray, int w, int h, float *eig) {
    if (!gray || !eig || w < 3 || h < 3) return -1;
    min_eigen(gray, w, h, eig);
    return 0;
}

int vso_good_features(const uint8_t *gray, int w, int h, int max_corners, double quality,
                      double min_dist, float *out_xy, int32_t *out_n) {
    if (!gray || !out_xy || !out_n || w < 3 || h < 3 || max_corners <= 0) return -1;
    *out_n = good_features(gray, w, h, max_corners, quality, min_dist, out_xy);
    return 0;
}

int vso_gaussian7(const uint8_t *gray, int w, int h, uint8_t *out) {
    if (!gray || !out || w < 4 || h < 4) return -1;
    gaussian7(gray, w, h, out);
    return 0;
}

int vso_orb_describe(const uint8_t *blurred, int w, int h, const float *xy, int n, float cos_a,
                     float sin_a, const int8_t *pattern, uint8_t *out_desc, int32_t *out_keep,
                     int32_t *out_n) {
    if (!blurred || !pattern || !out_desc || !out_keep || !out_n || n < 0) return -1;
    *out_n = orb_describe(blurred, w, h, xy, n, cos_a, sin_a, pattern, out_desc, out_keep);
    return 0;
}

int vso_extract_features(const uint8_t *bgr, int w, int h, int stride, int max_corners, float cos_a,
                         float sin_a, const int8_t *pattern, float *out_xy, uint8_t *out_desc,
                         int32_t *out_kd, int32_t *out_n, int32_t *out_n_detected) {
    std::vector<uint8_t> gray((size_t)w * h), blur((size_t)w * h);
    int rc = vso_bgr2gray(bgr, w, h, stride, gray.data());                          // Frame.cpp:56
    if (rc) return rc;
    std::vector<float> pts((size_t)2 * max_corners);
    int32_t nd = 0;
    rc = vso_good_features(gray.data(), w, h, max_corners, 0.01, 3.0, pts.data(), &nd);   // :61
    if (rc) return rc;
    gaussian7(gray.data(), w, h, blur.data());
    std::vector<int32_t> keep(nd > 0 ? nd : 1);
    int32_t nk = 0;
    vso_orb_describe(blur.data(), w, h, pts.data(), nd, cos_a, sin_a, pattern, out_desc, keep.data(), &nk);   // :68
    for (int i = 0; i < nk; i++) {                                                  // :69-72
        out_xy[2 * i] = pts[2 * keep[i]];
        out_xy[2 * i + 1] = pts[2 * keep[i] + 1];
    }
    *out_n = nk;
    if (out_n_detected) *out_n_detected = nd;                                       // :73
    if (out_kd) vso_kdtree_build_frame(out_xy, nk, out_kd);                         // :76
    return 0;
}

}  // extern "C"

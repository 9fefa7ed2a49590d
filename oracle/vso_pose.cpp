// ORACLE (test infrastructure only) — pose helpers downstream of the hot path.
//
// Follows /root/reference/src/helpers.cpp:
//   extract_Rt   :3-35   E = K^T F K, SVD, t = U.col(2) / |.|, R from U W V^T / U W^T V^T, sign fixes
//   triangulate  :37-80  per match: 4x4 DLT system, SVD, X = V_t.row(3) / V_t(3,3)
// cv::Mat algebra restated from OpenCV 4.x's built-in paths [OpenCV, from memory] — PARITY UNPINNED:
//   * A*B with flags == 0 and inner length 3 or 4 where it equals a result dimension: the float
//     "small matrix" code, a0*b0 + a1*b1 + ... left to right;
//   * products carrying a transpose flag (K.t()*F, U*W.t()) go through GEMMSingleMul<float,double>:
//     double products and running sum, one rounding to float;
//   * s*row - row (MatOp_AddEx) is addWeighted in float: fl(fl(a*s) - b);
//   * cv::norm(NORM_L2) accumulates squares in double; `m /= s` is convertTo(alpha = 1./s) with the
//     factor cast to float; cv::determinant of a 3x3 CV_32F evaluates in double.
//   * cv::SVD::compute on CV_32F = the Jacobi of vso_svd.cpp.
#include "vso.h"
#include "vso_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

// float small-matrix product C(r x c) = A(r x n) * B(n x c), left-to-right float accumulation
void mul_small(const float *A, const float *B, float *C, int r, int n, int c) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            float t = A[i * n + 0] * B[0 * c + j];
            for (int k = 1; k < n; k++) t = t + A[i * n + k] * B[k * c + j];
            C[i * c + j] = t;
        }
}
// C = At * B (GEMM_1_T) with double accumulation
void mul_At_B_d(const float *A, const float *B, float *C, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += (double)A[k * n + i] * (double)B[k * n + j];
            C[i * n + j] = (float)s;
        }
}
// C = A * Bt (GEMM_2_T) with double accumulation
void mul_A_Bt_d(const float *A, const float *B, float *C, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += (double)A[i * n + k] * (double)B[j * n + k];
            C[i * n + j] = (float)s;
        }
}
double det3(const float *m) {
    return m[0] * ((double)m[4] * m[8] - (double)m[5] * m[7]) - m[1] * ((double)m[3] * m[8] - (double)m[5] * m[6]) +
           m[2] * ((double)m[3] * m[7] - (double)m[4] * m[6]);
}

}  // namespace

extern "C" {

// extract_Rt(fundamental, K, rotation, translation): src/helpers.cpp:3-35
int vso_extract_Rt(const float *F, const float *K, float *R_out, float *t_out) {
    float KtF[9], E[9];
    mul_At_B_d(K, F, KtF, 3);                // K.t() * fundamental
    mul_small(KtF, K, E, 3, 3, 3);           // ... * K
    float D[3], U[9], Vt[9];
    vso::svd32f_full(E, 3, 3, D, U, Vt);     // cv::SVD::compute(E, D, U, V_t), :7
    float t[3] = {U[2], U[5], U[8]};         // U.col(2), :9
    const double nrm = std::sqrt((double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2]);
    const float inv = (float)(1. / nrm);     // translation /= cv::norm(translation), :11
    for (float &v : t) v = v * inv;
    const float W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};   // :13-16
    float UW[9], R1[9], UWt[9], R2[9];
    mul_small(U, W, UW, 3, 3, 3);
    mul_small(UW, Vt, R1, 3, 3, 3);          // R_1 = U * W * V_t, :18
    if (det3(R1) < 0)
        for (float &v : R1) v = -v;
    mul_A_Bt_d(U, W, UWt, 3);                // U * W.t()
    mul_small(UWt, Vt, R2, 3, 3, 3);         // R_2, :23
    if (det3(R2) < 0)
        for (float &v : R2) v = -v;
    const float tr = R1[0] + R1[4] + R1[8];  // :29
    std::memcpy(R_out, tr < 0 ? R2 : R1, sizeof(R1));
    if (t[2] < 0)                            // :31-33
        for (float &v : t) v = v * -1.f;
    std::memcpy(t_out, t, sizeof(t));
    return 0;
}

// c2 = K * R_t.rowRange(0,3) with R_t = [R | t] (src/vslam.cpp:83-85,125): 3x4, float small-matrix path
int vso_camera_matrix(const float *K, const float *R, const float *t, float *c2) {
    float Rt[12];
    for (int r = 0; r < 3; r++) {
        Rt[r * 4 + 0] = R[r * 3 + 0]; Rt[r * 4 + 1] = R[r * 3 + 1]; Rt[r * 4 + 2] = R[r * 3 + 2]; Rt[r * 4 + 3] = t[r];
    }
    mul_small(K, Rt, c2, 3, 3, 4);
    return 0;
}

// triangulate(p1, p2, c1, c2, points_4d): src/helpers.cpp:37-80.  p1/p2: n x 2, c1/c2: 3 x 4, out: n x 4
int vso_triangulate(const float *p1, const float *p2, int n, const float *c1, const float *c2, float *points_4d) {
    for (int i = 0; i < n; i++) {
        float A[16];
        for (int c = 0; c < 4; c++) {
            A[0 * 4 + c] = p1[2 * i] * c1[2 * 4 + c] - c1[0 * 4 + c];       // :49-52
            A[1 * 4 + c] = p1[2 * i + 1] * c1[2 * 4 + c] - c1[1 * 4 + c];
            A[2 * 4 + c] = p2[2 * i] * c2[2 * 4 + c] - c2[0 * 4 + c];
            A[3 * 4 + c] = p2[2 * i + 1] * c2[2 * 4 + c] - c2[1 * 4 + c];
        }
        float D[4], U[16], Vt[16];
        vso::svd32f_full(A, 4, 4, D, U, Vt);                                // :59
        const float *v = &Vt[3 * 4];
        points_4d[4 * i + 0] = v[0] / v[3];                                 // :72-75
        points_4d[4 * i + 1] = v[1] / v[3];
        points_4d[4 * i + 2] = v[2] / v[3];
        points_4d[4 * i + 3] = 1;
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Map association: src/vslam.cpp:129-161 + orb_distance, src/PointMap.cpp:36-46.
//   projected = pm.points(N x 4) * c2.t()  -> GEMM_2_T: double products, four partial sums
//   (k, k+1, k+2, k+3 of each 4-block), (s0+s1+s2+s3) rounded to float [OpenCV, from memory];
//   x /= h, y /= h; in-view test; radius_search(frame.kdtree, frame.points, q, 2); the first hit that is
//   still unassigned and whose orb_distance (min Hamming over the map point's observations) is < 64
//   gets the map point.  The loop is sequential: an earlier map point's claim hides the keypoint from
//   later ones.
// obs_offsets[N+1] / obs_desc[total][32]: descriptors of each map point's observations (CSR).
// map_point_ids (n_kp, in/out).  out_claim[i] = keypoint index claimed by map point i or -1.
extern "C" int vso_associate_map_points(const float *map_points /*N x 4*/, int n_map, const float *c2 /*3x4*/,
                                        int img_w, int img_h, const int32_t *kd_nodes, const float *kp_xy,
                                        const uint8_t *kp_desc, int n_kp, const int32_t *obs_offsets,
                                        const uint8_t *obs_desc, float radius, uint32_t dist_threshold,
                                        int32_t *map_point_ids, int32_t *out_claim) {
    for (int i = 0; i < n_map; i++) {
        out_claim[i] = -1;
        const float *P = map_points + (size_t)i * 4;
        float pr[3];
        for (int r = 0; r < 3; r++) {
            const double s0 = (double)P[0] * (double)c2[r * 4 + 0], s1 = (double)P[1] * (double)c2[r * 4 + 1];
            const double s2 = (double)P[2] * (double)c2[r * 4 + 2], s3 = (double)P[3] * (double)c2[r * 4 + 3];
            pr[r] = (float)(((s0 + s1) + s2) + s3);
        }
        const float hh = pr[2];
        const float x = pr[0] / hh, y = pr[1] / hh;                       // src/vslam.cpp:136-140
        if (!(x >= 0 && x < img_w && y >= 0 && y < img_h)) continue;      // :141-143
        int32_t hits[256];
        int cnt = vso_kdtree_radius_frame(kd_nodes, kp_xy, n_kp, x, y, radius, hits, 256);   // :149
        if (cnt > 256) cnt = 256;
        for (int h = 0; h < cnt; h++) {
            const int idx = hits[h];
            if (map_point_ids[idx] >= 0) continue;                         // :151
            uint32_t mn = 0xFFFFFFFFu;                                     // orb_distance, PointMap.cpp:37-45
            for (int o = obs_offsets[i]; o < obs_offsets[i + 1]; o++) {
                const uint32_t cur = vso_hamming256(kp_desc + (size_t)idx * 32, obs_desc + (size_t)o * 32);
                if (cur < mn) mn = cur;
            }
            if (mn < dist_threshold) {                                     // :153
                map_point_ids[idx] = i;
                out_claim[i] = idx;
                break;                                                     // :157
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Reprojection-error filter, src/vslam.cpp:192-251, restated bug for bug:
//   reproj = points_4d (N x 4) * c.t()  (GEMM_2_T: double products, (s0+s1+s2+s3), one rounding);
//   the de-homogenise loop walks the FLAT N x 3 array with `i += 3` while `i < rows`, so only the first
//   ceil(N/3) points are divided by their third coordinate (:201-211);
//   d = reproj(:, 0:2) - initial_points; re = d.row(i).dot(d.row(i)) (double accumulate, stored as f32);
//   a match is skipped when frame.map_point_ids[i] > 0 — indexed by the MATCH index i (:240);
//   kept when re1 <= thresholdSq and re2 <= thresholdSq; reproj_error += re1 + re2 in double.
extern "C" int vso_reprojection_filter(const float *points_4d, const float *p1, const float *p2, int n, const float *c1,
                                       const float *c2, const int32_t *map_point_ids, float threshold_sq,
                                       int32_t *out_idx, int32_t *out_n, double *out_err) {
    std::vector<float> r1((size_t)n * 3), r2((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        const float *P = points_4d + (size_t)i * 4;
        for (int r = 0; r < 3; r++) {
            r1[(size_t)i * 3 + r] = (float)((((double)P[0] * c1[r * 4 + 0] + (double)P[1] * c1[r * 4 + 1]) + (double)P[2] * c1[r * 4 + 2]) +
                                            (double)P[3] * c1[r * 4 + 3]);
            r2[(size_t)i * 3 + r] = (float)((((double)P[0] * c2[r * 4 + 0] + (double)P[1] * c2[r * 4 + 1]) + (double)P[2] * c2[r * 4 + 2]) +
                                            (double)P[3] * c2[r * 4 + 3]);
        }
    }
    for (int i = 0; i < n; i += 3) {          // flat index, stride 3, bound = rows (:201)
        float &h1 = r1[i + 2];
        r1[i] /= h1;
        r1[i + 1] /= h1;
        h1 = 1;
        float &h2 = r2[i + 2];
        r2[i] /= h2;
        r2[i + 1] /= h2;
        h2 = 1;
    }
    int k = 0;
    double err = 0;
    for (int i = 0; i < n; i++) {
        if (map_point_ids[i] > 0) continue;                                     // :240
        const float d1x = r1[(size_t)i * 3] - p1[2 * i], d1y = r1[(size_t)i * 3 + 1] - p1[2 * i + 1];
        const float re1 = (float)((double)d1x * d1x + (double)d1y * d1y);
        if (re1 > threshold_sq) continue;
        const float d2x = r2[(size_t)i * 3] - p2[2 * i], d2y = r2[(size_t)i * 3 + 1] - p2[2 * i + 1];
        const float re2 = (float)((double)d2x * d2x + (double)d2y * d2y);
        if (re2 > threshold_sq) continue;
        out_idx[k++] = i;
        err += re1 + re2;                                                       // float sum promoted, :249
    }
    *out_n = k;
    *out_err = err;
    return 0;
}

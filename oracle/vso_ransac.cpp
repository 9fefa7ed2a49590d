// ORACLE (test infrastructure only) — RansacFilter restated.
//
// Follows /root/reference/src/RansacFilter.cpp line by line:
//   initialize_sets               :6-34   (seed injected instead of std::random_device, :15-16)
//   find_fundamental              :36-67
//   compute_fundamental           :69-103
//   compute_fundamental_residual  :105-140
// cv::Mat algebra is restated from OpenCV 4.x's built-in paths [OpenCV, from memory]:
//   * A*B for CV_32F with flags==0 and inner length 3 (modules/core/src/matmul.simd.hpp, the
//     "small matrix" switch): plain float  a0*b0 + a1*b1 + a2*b2, left to right.
//   * F.t()*x2 carries GEMM_1_T, so it takes GEMMSingleMul<float,double>: products and the
//     running sum in double, one rounding to float at the end.
//   * Mat::mul, operator/ and operator+ on CV_32F are element-wise IEEE float operations;
//     cv::reduce(REDUCE_SUM, dim 0) on CV_32F adds rows in order (row0 + row1) + row2.
//   * cv::sum on CV_32F accumulates in double; the element order inside OpenCV is SIMD-build
//     dependent, the oracle pins it to index order.
// Parity: UNPINNED (see vso.h).
#include "vso.h"
#include "vso_internal.h"

#include <cstring>
#include <random>
#include <vector>

namespace {

struct P2 { float x, y; };

// src/RansacFilter.cpp:69-103
void compute_fundamental(const P2 *p1_set, const P2 *p2_set, int N, float *F_out, int32_t *work = nullptr) {
    std::vector<float> A((size_t)N * 9);
    for (int i = 0; i < N; i++) {
        const float u1 = p1_set[i].x, v1 = p1_set[i].y;
        const float u2 = p2_set[i].x, v2 = p2_set[i].y;
        float *r = &A[(size_t)i * 9];
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2;      // dst' * F * src = 0, :81-89
        r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2;
        r[6] = u1;      r[7] = v1;      r[8] = 1;
    }
    std::vector<float> D(9), U((size_t)N * N), Vt(81);
    vso::svd32f_full(A.data(), N, 9, D.data(), U.data(), Vt.data());   // :94
    if (work) { work[0] = vso::g_last_sweeps; work[1] = vso::g_last_visits; work[2] = vso::g_last_rotations; }
    float F0[9];
    std::memcpy(F0, &Vt[8 * 9], sizeof(F0));                            // V_t.row(8).reshape(0,3), :95

    float D3[3], U3[9], Vt3[9];
    vso::svd32f_full(F0, 3, 3, D3, U3, Vt3);                            // :98
    if (work) { work[3] = vso::g_last_sweeps; work[4] = vso::g_last_visits; work[5] = vso::g_last_rotations; }
    D3[2] = 0;                                                          // :99

    // temp_F = U * diag(D) * V_t (:101): two 3x3 float products through the small-matrix path
    const float Dg[9] = {D3[0], 0, 0, 0, D3[1], 0, 0, 0, D3[2]};
    float UD[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            UD[i * 3 + j] = U3[i * 3 + 0] * Dg[0 * 3 + j] + U3[i * 3 + 1] * Dg[1 * 3 + j] +
                            U3[i * 3 + 2] * Dg[2 * 3 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            F_out[i * 3 + j] = UD[i * 3 + 0] * Vt3[0 * 3 + j] + UD[i * 3 + 1] * Vt3[1 * 3 + j] +
                               UD[i * 3 + 2] * Vt3[2 * 3 + j];
}

// src/RansacFilter.cpp:105-140
void residual(const P2 *p1, const P2 *p2, const int32_t *pairs, int N, const float *F,
              float threshold, uint8_t *mask, int32_t *count, float *sum) {
    int nInliers = 0;
    double total = 0;                                    // cv::sum(e_sq)[0], :138
    for (int i = 0; i < N; i++) {
        const float x1 = p1[pairs[2 * i]].x, y1 = p1[pairs[2 * i]].y;          // :111-113
        const float x2 = p2[pairs[2 * i + 1]].x, y2 = p2[pairs[2 * i + 1]].y;  // :114-116
        // F_x1 = F * x1 (:119): float, left to right; the homogeneous 1 multiplies exactly
        const float Fx1_0 = F[0] * x1 + F[1] * y1 + F[2] * 1.0f;
        const float Fx1_1 = F[3] * x1 + F[4] * y1 + F[5] * 1.0f;
        const float Fx1_2 = F[6] * x1 + F[7] * y1 + F[8] * 1.0f;
        // F_t_x2 = F.t() * x2 (:120): double accumulation, one rounding
        const float Ftx2_0 = (float)(((double)F[0] * (double)x2 + (double)F[3] * (double)y2) + (double)F[6] * 1.0);
        const float Ftx2_1 = (float)(((double)F[1] * (double)x2 + (double)F[4] * (double)y2) + (double)F[7] * 1.0);
        // x2.mul(F_x1) reduced over rows (:122-123)
        const float n = (x2 * Fx1_0 + y2 * Fx1_1) + 1.0f * Fx1_2;
        // e_sq (:126) exactly as the operators bind: n*n / a*a + b*b + c*c + d*d
        const float q = (n * n) / (Fx1_0 * Fx1_0);
        const float e = ((q + Fx1_1 * Fx1_1) + Ftx2_0 * Ftx2_0) + Ftx2_1 * Ftx2_1;
        if (e <= threshold) {                            // :130, NaN <= thr is false
            mask[i] = 1;
            nInliers++;
        } else {
            mask[i] = 0;
        }
        total += (double)e;
    }
    *count = nInliers;
    *sum = (float)total;
}

}  // namespace

extern "C" {

int vso_ransac_sets(uint32_t seed, int n_matches, int min_items, int H, int32_t *out_sets) {
    if (n_matches < min_items || min_items < 0 || min_items > 8 || H < 0 || !out_sets) return -1;
    std::vector<int> all_indices, available_indices;
    all_indices.reserve(n_matches);
    for (int i = 0; i < n_matches; i++) all_indices.push_back(i);
    std::mt19937 gen(seed);                                         // :16, seed injected
    std::memset(out_sets, 0, sizeof(int32_t) * (size_t)H * 8);      // vector<int>(8, 0), :17
    for (int i = 0; i < H; i++) {
        available_indices = all_indices;                            // :20
        for (int j = 0; j < min_items; j++) {
            std::uniform_int_distribution<> distr(0, (int)available_indices.size() - 1);   // :24
            int r = distr(gen);
            out_sets[(size_t)i * 8 + j] = available_indices[r];
            available_indices[r] = available_indices.back();        // :30
            available_indices.pop_back();
        }
    }
    return 0;
}

int vso_compute_fundamental(const float *p1_set, const float *p2_set, int n_set, float *F) {
    if (!p1_set || !p2_set || n_set <= 0 || n_set >= 9 || !F) return -1;
    compute_fundamental(reinterpret_cast<const P2 *>(p1_set), reinterpret_cast<const P2 *>(p2_set),
                        n_set, F);
    return 0;
}

// compute_fundamental + what its two Jacobi SVDs did: work[6] = (sweeps, (i, j) visits, rotations) of the 8x9 system,
// then of the 3x3 one.  For tools/solve_flops.py (operation counts of the solver; sweeps per hypothesis).
int vso_compute_fundamental_work(const float *p1_set, const float *p2_set, int n_set, float *F, int32_t *work) {
    if (!p1_set || !p2_set || n_set <= 0 || n_set >= 9 || !F || !work) return -1;
    compute_fundamental(reinterpret_cast<const P2 *>(p1_set), reinterpret_cast<const P2 *>(p2_set), n_set, F, work);
    return 0;
}

int vso_fundamental_residual(const float *p1, const float *p2, const int32_t *pairs, int m,
                             const float *F, float threshold, uint8_t *mask, int32_t *count,
                             float *sum) {
    if (m < 0 || !F || !count || !sum) return -1;
    std::vector<uint8_t> tmp;
    if (!mask) {
        tmp.resize(m > 0 ? m : 1);
        mask = tmp.data();
    }
    residual(reinterpret_cast<const P2 *>(p1), reinterpret_cast<const P2 *>(p2), pairs, m, F,
             threshold, mask, count, sum);
    return 0;
}

int vso_find_fundamental(const float *p1, const float *p2, const int32_t *pairs, int m,
                         const int32_t *sets, int H, float threshold, float *F, uint8_t *mask,
                         int32_t *best_count, float *best_sum, int32_t *best_iter, float *all_F,
                         int32_t *all_count, float *all_sum) {
    if (m < 0 || H < 0) return -1;
    const P2 *P1 = reinterpret_cast<const P2 *>(p1), *Q2 = reinterpret_cast<const P2 *>(p2);
    P2 p1_set[8], p2_set[8];
    float temp_F[9];
    float best_score = 0;                               // :44
    int best_nInliers = 0;                              // :45
    int winner = -1;
    std::vector<uint8_t> cur(m > 0 ? m : 1);
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < 8; j++) {                   // ransac_sets[i].size() == 8, :50
            const int idx = sets[(size_t)i * 8 + j];
            p1_set[j] = P1[pairs[2 * idx]];
            p2_set[j] = Q2[pairs[2 * idx + 1]];
        }
        compute_fundamental(p1_set, p2_set, 8, temp_F);
        int32_t cnt;
        float sum;
        residual(P1, Q2, pairs, m, temp_F, threshold, cur.data(), &cnt, &sum);
        if (all_F) std::memcpy(all_F + (size_t)i * 9, temp_F, sizeof(temp_F));
        if (all_count) all_count[i] = cnt;
        if (all_sum) all_sum[i] = sum;
        if (cnt > best_nInliers || (cnt == best_nInliers && sum > best_score)) {   // :59
            best_nInliers = cnt;
            best_score = sum;
            if (F) std::memcpy(F, temp_F, sizeof(temp_F));
            if (mask) std::memcpy(mask, cur.data(), (size_t)m);
            winner = i;
        }
    }
    if (best_count) *best_count = best_nInliers;
    if (best_sum) *best_sum = best_score;
    if (best_iter) *best_iter = winner;
    return 0;
}

int vso_match_features(const float *xy1, const uint8_t *d1, int n1, const float *xy2,
                       const uint8_t *d2, int n2, uint32_t seed, int H, float threshold,
                       int32_t *out_matches, int32_t *out_n, float *F, int32_t *n_prelim) {
    // src/Frame.cpp:82-105
    std::vector<int32_t> pairs((size_t)2 * (n1 > 0 ? n1 : 1));
    int32_t m = 0;
    int rc = vso_match_knn2_ratio(d1, n1, d2, n2, pairs.data(), &m);
    if (rc) return rc;
    if (n_prelim) *n_prelim = m;
    *out_n = 0;
    if (m < 8) return -2;                               // reference: UB (range (0,-1)), :24
    std::vector<int32_t> sets((size_t)H * 8);
    rc = vso_ransac_sets(seed, m, 8, H, sets.data());
    if (rc) return rc;
    std::vector<uint8_t> mask(m);
    int32_t bc, bi;
    float bs;
    vso_find_fundamental(xy1, xy2, pairs.data(), m, sets.data(), H, threshold, F, mask.data(), &bc,
                         &bs, &bi, nullptr, nullptr, nullptr);
    int k = 0;
    if (bi >= 0)
        for (int i = 0; i < m; i++)
            if (mask[i]) {                              // Frame.cpp:98-102
                out_matches[2 * k] = pairs[2 * i];
                out_matches[2 * k + 1] = pairs[2 * i + 1];
                k++;
            }
    *out_n = k;
    return 0;
}

}  // extern "C"

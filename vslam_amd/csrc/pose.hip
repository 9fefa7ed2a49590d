// Pose helpers downstream of the RANSAC winner for gfx950.
//
// Replaces /root/reference/src/helpers.cpp:
//   extract_Rt   :3-35   -> pose_from_F_kernel   (one lane per frame pair)
//   triangulate  :37-80  -> triangulate_kernel   (one lane per inlier match: a 4x4 Jacobi SVD each)
// plus c2 = K * [R|t] (src/vslam.cpp:83-85,125).  Arithmetic follows oracle/vso_pose.cpp operation for
// operation: OpenCV's float small-matrix products, double-accumulated transposed products, and the
// same one-sided Jacobi as the RANSAC solver (here fully in registers: the matrices are 3x3 / 4x4).
#include "ctx.h"

#include <cfloat>

namespace {

__device__ __forceinline__ uint32_t cvrng_next(uint64_t &state) {
    state = (uint64_t)(uint32_t)state * 4164903690ull + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

// cv::SVDecomp / cv::SVD::compute on a square CV_32F matrix (JacobiSVDImpl_<float>, m = n = N).
// A row-major in; outputs w (descending), U (N x N, row-major), Vt (N x N).
template <int N>
__device__ void svd_square(const float *A, float *w, float *U, float *Vt) {
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    float At[N][N], V[N][N];
    double W[N];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int k = 0; k < N; k++) At[i][k] = A[k * N + i];   // temp_a = transpose(src)
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd = __builtin_fma((double)At[i][k], (double)At[i][k], sd);
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) V[i][k] = (i == k) ? 1.f : 0.f;
    }
    constexpr int max_iter = N > 30 ? N : 30;
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < N; k++) p = __builtin_fma((double)At[i][k], (double)At[j][k], p);
                if (!(fabs(p) <= (double)eps * sqrt(a * b))) {
                    p *= 2;
                    const double beta = a - b;
                    const double gamma = sqrt(p * p + beta * beta);   // pinned hypot
                    float c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = (float)sqrt(delta / gamma);
                        c = (float)(p / (gamma * (double)s * 2));
                    } else {
                        c = (float)sqrt((gamma + beta) / (gamma * 2));
                        s = (float)(p / (gamma * (double)c * 2));
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const float t0 = c * At[i][k] + s * At[j][k];
                        const float t1 = (-s) * At[i][k] + c * At[j][k];
                        At[i][k] = t0;
                        At[j][k] = t1;
                        a = __builtin_fma((double)t0, (double)t0, a);
                        b = __builtin_fma((double)t1, (double)t1, b);
                    }
                    W[i] = a;
                    W[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const float t0 = c * V[i][k] + s * V[j][k];
                        const float t1 = (-s) * V[i][k] + c * V[j][k];
                        V[i][k] = t0;
                        V[j][k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd = __builtin_fma((double)At[i][k], (double)At[i][k], sd);
        W[i] = sqrt(sd);
    }
#pragma unroll
    for (int i = 0; i < N - 1; i++) {   // selection sort, descending; rows travel with W
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) {
                j = k;
                wj = W[k];
            }
#pragma unroll
        for (int jj = i + 1; jj < N; jj++)
            if (jj == j) {
                const double tw = W[i];
                W[i] = W[jj];
                W[jj] = tw;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const float x = At[i][k];
                    At[i][k] = At[jj][k];
                    At[jj][k] = x;
                    const float y = V[i][k];
                    V[i][k] = V[jj][k];
                    V[jj][k] = y;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < N; i++) w[i] = (float)W[i];
    uint64_t rng = 0x12345678ull;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const float val0 = (float)(1. / N);
#pragma unroll
            for (int k = 0; k < N; k++) At[i][k] = (cvrng_next(rng) & 256) != 0 ? val0 : -val0;
            for (int it = 0; it < 2; it++) {
#pragma unroll
                for (int j = 0; j < N; j++) {
                    if (j < i) {
                        sd = 0;
#pragma unroll
                        for (int k = 0; k < N; k++) sd += (double)(At[i][k] * At[j][k]);
                        float asum = 0;
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            const float t = (float)((double)At[i][k] - sd * (double)At[j][k]);
                            At[i][k] = t;
                            asum += fabsf(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                        for (int k = 0; k < N; k++) At[i][k] = At[i][k] * asum;
                    }
                }
            }
            sd = 0;
#pragma unroll
            for (int k = 0; k < N; k++) sd = __builtin_fma((double)At[i][k], (double)At[i][k], sd);
            sd = sqrt(sd);
        }
        const float s = (float)(sd > minval ? 1 / sd : 0.);
#pragma unroll
        for (int k = 0; k < N; k++) At[i][k] = At[i][k] * s;
    }
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
        for (int c = 0; c < N; c++) {
            U[r * N + c] = At[c][r];   // u = transpose(temp_u)
            Vt[r * N + c] = V[r][c];
        }
}

// OpenCV float small-matrix product, left-to-right accumulation
template <int R, int NN, int C>
__device__ __forceinline__ void mul_small(const float *A, const float *B, float *Cm) {
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
        for (int j = 0; j < C; j++) {
            float t = A[i * NN + 0] * B[0 * C + j];
#pragma unroll
            for (int k = 1; k < NN; k++) t = t + A[i * NN + k] * B[k * C + j];
            Cm[i * C + j] = t;
        }
}
__device__ __forceinline__ double det3(const float *m) {
    return (double)m[0] * ((double)m[4] * (double)m[8] - (double)m[5] * (double)m[7]) -
           (double)m[1] * ((double)m[3] * (double)m[8] - (double)m[5] * (double)m[6]) +
           (double)m[2] * ((double)m[3] * (double)m[7] - (double)m[4] * (double)m[6]);
}

struct Mat3 {
    float v[9];
};

__global__ __launch_bounds__(64) void pose_from_F_kernel(const float *__restrict__ Fm, const int32_t *__restrict__ best,
                                                         int batch, Mat3 Kc, float *__restrict__ R_out,
                                                         float *__restrict__ t_out, float *__restrict__ c2_out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    if (best && best[(size_t)b * 4] < 0) return;   // no accepted model: `fundamental` is empty in the reference
    float F[9], K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F[i] = Fm[(size_t)b * 9 + i];
        K[i] = Kc.v[i];
    }
    float KtF[9], E[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {   // K.t() * F: double products (exact) and running sum
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma((double)K[k * 3 + i], (double)F[k * 3 + j], s);
            KtF[i * 3 + j] = (float)s;
        }
    mul_small<3, 3, 3>(KtF, K, E);
    float D[3], U[9], Vt[9];
    svd_square<3>(E, D, U, Vt);
    float t[3] = {U[2], U[5], U[8]};
    const double nrm = sqrt(__builtin_fma((double)t[2], (double)t[2],
                                          __builtin_fma((double)t[1], (double)t[1], (double)t[0] * (double)t[0])));
    const float inv = (float)(1. / nrm);
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = t[i] * inv;
    const float W[9] = {0.f, -1.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    float UW[9], R1[9], UWt[9], R2[9];
    mul_small<3, 3, 3>(U, W, UW);
    mul_small<3, 3, 3>(UW, Vt, R1);
    if (det3(R1) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) R1[i] = -R1[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {   // U * W.t()
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma((double)U[i * 3 + k], (double)W[j * 3 + k], s);
            UWt[i * 3 + j] = (float)s;
        }
    mul_small<3, 3, 3>(UWt, Vt, R2);
    if (det3(R2) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) R2[i] = -R2[i];
    }
    const float tr = R1[0] + R1[4] + R1[8];
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = tr < 0 ? R2[i] : R1[i];
    if (t[2] < 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) t[i] = t[i] * -1.f;
    }
    float Rt[12], c2[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        Rt[r * 4 + 0] = R[r * 3 + 0];
        Rt[r * 4 + 1] = R[r * 3 + 1];
        Rt[r * 4 + 2] = R[r * 3 + 2];
        Rt[r * 4 + 3] = t[r];
    }
    mul_small<3, 3, 4>(K, Rt, c2);
#pragma unroll
    for (int i = 0; i < 9; i++) R_out[(size_t)b * 9 + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t_out[(size_t)b * 3 + i] = t[i];
#pragma unroll
    for (int i = 0; i < 12; i++) c2_out[(size_t)b * 12 + i] = c2[i];
}

// one point of triangulate() (src/helpers.cpp:48-76): the 4 x 4 system from the two image points and camera matrices, its
// SVD, the last row of V^T de-homogenised
__device__ __forceinline__ float4 triangulate_one(float2 p1, float2 p2, const float *c1, const float *c2) {
    float A[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {   // s*row - row in float: fl(fl(a*s) - b)
        const float a0 = p1.x * c1[8 + c], a1 = p1.y * c1[8 + c], a2 = p2.x * c2[8 + c], a3 = p2.y * c2[8 + c];
        A[0 + c] = a0 - c1[0 + c];
        A[4 + c] = a1 - c1[4 + c];
        A[8 + c] = a2 - c2[0 + c];
        A[12 + c] = a3 - c2[4 + c];
    }
    float D[4], U[16], Vt[16];
    svd_square<4>(A, D, U, Vt);
    float4 o;
    o.x = Vt[12] / Vt[15];
    o.y = Vt[13] / Vt[15];
    o.z = Vt[14] / Vt[15];
    o.w = 1.f;
    return o;
}

__global__ __launch_bounds__(128) void triangulate_kernel(const float *__restrict__ xy1, const float *__restrict__ xy2,
                                                          const int32_t *__restrict__ matches,
                                                          const int32_t *__restrict__ best, int kp_stride, Mat3 Kc,
                                                          const float *__restrict__ c2_all, float *__restrict__ points4d) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 128 + threadIdx.x;
    const int n = best[(size_t)b * 4 + 3];
    if (best[(size_t)b * 4] < 0 || i >= n) return;
    const int2 m = reinterpret_cast<const int2 *>(matches)[(size_t)b * kp_stride + i];
    const float2 p1 = reinterpret_cast<const float2 *>(xy1)[(size_t)b * kp_stride + m.x];
    const float2 p2 = reinterpret_cast<const float2 *>(xy2)[(size_t)b * kp_stride + m.y];
    float c1[12], c2[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {   // c1 = [K | 0], src/vslam.cpp:123-124
        c1[r * 4 + 0] = Kc.v[r * 3 + 0];
        c1[r * 4 + 1] = Kc.v[r * 3 + 1];
        c1[r * 4 + 2] = Kc.v[r * 3 + 2];
        c1[r * 4 + 3] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) c2[k] = c2_all[(size_t)b * 12 + k];
    reinterpret_cast<float4 *>(points4d)[(size_t)b * kp_stride + i] = triangulate_one(p1, p2, c1, c2);
}

// triangulate(p1, p2, c1, c2, points_4d) as the reference declares it (include/helpers.h:19): n point pairs, any two 3 x 4
// camera matrices
struct Mat34 {
    float v[12];
};
__global__ __launch_bounds__(128) void triangulate_points_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int n,
                                                                 Mat34 C1, Mat34 C2, float *__restrict__ points4d) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    float c1[12], c2[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        c1[k] = C1.v[k];
        c2[k] = C2.v[k];
    }
    reinterpret_cast<float4 *>(points4d)[i] =
        triangulate_one(reinterpret_cast<const float2 *>(p1)[i], reinterpret_cast<const float2 *>(p2)[i], c1, c2);
}

// Reprojection-error filter of src/vslam.cpp:192-251, bug for bug (see oracle/vso_pose.cpp): one workgroup per pair.
// rp1 / rp2: [batch][kp_stride][3] scratch for the reprojected homogeneous points.
__global__ __launch_bounds__(256) void reproj_filter_kernel(const float *__restrict__ points4d, const float *__restrict__ xy1,
                                                            const float *__restrict__ xy2, const int32_t *__restrict__ matches,
                                                            const int32_t *__restrict__ best, int kp_stride, Mat3 Kc,
                                                            const float *__restrict__ c2_all,
                                                            const int32_t *__restrict__ map_point_ids, float threshold_sq,
                                                            float *__restrict__ rp1, float *__restrict__ rp2,
                                                            int32_t *__restrict__ out_idx, int32_t *__restrict__ out_n,
                                                            double *__restrict__ out_err, float *__restrict__ kept_err) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int n = best[(size_t)b * 4] < 0 ? 0 : best[(size_t)b * 4 + 3];
    float *R1 = rp1 + (size_t)b * kp_stride * 3, *R2 = rp2 + (size_t)b * kp_stride * 3;
    float c1[12], c2[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        c1[r * 4 + 0] = Kc.v[r * 3 + 0];
        c1[r * 4 + 1] = Kc.v[r * 3 + 1];
        c1[r * 4 + 2] = Kc.v[r * 3 + 2];
        c1[r * 4 + 3] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) c2[k] = c2_all[(size_t)b * 12 + k];
    for (int i = tid; i < n; i += 256) {   // points_4d * c.t(): exact double products, (s0+s1+s2+s3), one rounding
        const float4 P = reinterpret_cast<const float4 *>(points4d)[(size_t)b * kp_stride + i];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            R1[(size_t)i * 3 + r] = (float)((((double)P.x * (double)c1[r * 4] + (double)P.y * (double)c1[r * 4 + 1]) +
                                             (double)P.z * (double)c1[r * 4 + 2]) + (double)P.w * (double)c1[r * 4 + 3]);
            R2[(size_t)i * 3 + r] = (float)((((double)P.x * (double)c2[r * 4] + (double)P.y * (double)c2[r * 4 + 1]) +
                                             (double)P.z * (double)c2[r * 4 + 2]) + (double)P.w * (double)c2[r * 4 + 3]);
        }
    }
    __syncthreads();
    for (int i = 3 * tid; i < n; i += 3 * 256) {   // flat stride-3 walk bounded by rows, :201-211
        const float h1 = R1[i + 2];
        R1[i] = R1[i] / h1;
        R1[i + 1] = R1[i + 1] / h1;
        R1[i + 2] = 1.f;
        const float h2 = R2[i + 2];
        R2[i] = R2[i] / h2;
        R2[i + 1] = R2[i + 1] / h2;
        R2[i + 2] = 1.f;
    }
    __syncthreads();
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int32_t *ids = map_point_ids + (size_t)b * kp_stride;
    const int2 *M = reinterpret_cast<const int2 *>(matches) + (size_t)b * kp_stride;
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    int32_t *O = out_idx + (size_t)b * kp_stride;
    float *KE = kept_err + (size_t)b * kp_stride;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        bool keep = false;
        float e = 0.f;
        if (i < n && !(ids[i] > 0)) {   // indexed by the match index, :240
            const int2 m = M[i];
            const float2 a = P1[m.x], c = P2[m.y];
            const float d1x = R1[(size_t)i * 3] - a.x, d1y = R1[(size_t)i * 3 + 1] - a.y;
            const float re1 = (float)__builtin_fma((double)d1y, (double)d1y, (double)d1x * (double)d1x);
            if (!(re1 > threshold_sq)) {
                const float d2x = R2[(size_t)i * 3] - c.x, d2y = R2[(size_t)i * 3 + 1] - c.y;
                const float re2 = (float)__builtin_fma((double)d2y, (double)d2y, (double)d2x * (double)d2x);
                if (!(re2 > threshold_sq)) {
                    keep = true;
                    e = re1 + re2;
                }
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = (int)__popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int wv = 0; wv < wave; wv++) off += s_wave[wv];
        off += (int)__popcll(bal & ((1ull << lane) - 1ull));
        if (keep) {
            O[off] = i;
            KE[off] = e;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) {
        double err = 0;   // reproj_error += re1 + re2, in match order
        for (int k = 0; k < s_base; k++) err += (double)KE[k];
        out_n[b] = s_base;
        out_err[b] = err;
    }
}

}  // namespace

int vs_launch_reproj_filter(vslam_ctx *ctx, const float *points4d, const float *xy1, const float *xy2, const int32_t *matches,
                            const int32_t *best, int batch, int kp_stride, const float *h_K, const float *c2,
                            const int32_t *map_point_ids, float threshold_sq, int32_t *out_idx, int32_t *out_n,
                            double *out_err) {
    VS_REQUIRE(ctx, points4d && xy1 && xy2 && matches && best && h_K && c2 && map_point_ids && out_idx && out_n && out_err,
               VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    Mat3 K;
    for (int i = 0; i < 9; i++) K.v[i] = h_K[i];
    float *rp1 = nullptr, *rp2 = nullptr, *ke = nullptr;
    int rc;
    const size_t bk = (size_t)batch * kp_stride;
    if ((rc = vs_arena_get(ctx, "pose.rp1", sizeof(float) * 3 * bk + 16, (void **)&rp1))) return rc;
    if ((rc = vs_arena_get(ctx, "pose.rp2", sizeof(float) * 3 * bk + 16, (void **)&rp2))) return rc;
    if ((rc = vs_arena_get(ctx, "pose.kept_err", sizeof(float) * bk, (void **)&ke))) return rc;
    VsProfScope ps(ctx, "reproj_filter_kernel");
    reproj_filter_kernel<<<batch, 256, 0, ctx->stream>>>(points4d, xy1, xy2, matches, best, kp_stride, K, c2, map_point_ids,
                                                         threshold_sq, rp1, rp2, out_idx, out_n, out_err, ke);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_extract_Rt(vslam_ctx *ctx, const float *F, const int32_t *best, int batch, const float *h_K, float *R,
                         float *t, float *c2) {
    VS_REQUIRE(ctx, F && h_K && R && t && c2, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0, VSLAM_ERR_INVALID);
    Mat3 K;
    for (int i = 0; i < 9; i++) K.v[i] = h_K[i];
    VsProfScope ps(ctx, "pose_from_F_kernel");
    pose_from_F_kernel<<<vs_div_up(batch, 64), 64, 0, ctx->stream>>>(F, best, batch, K, R, t, c2);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_triangulate(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *matches,
                          const int32_t *best, int batch, int kp_stride, const float *h_K, const float *c2,
                          float *points4d) {
    VS_REQUIRE(ctx, xy1 && xy2 && matches && best && h_K && c2 && points4d, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    Mat3 K;
    for (int i = 0; i < 9; i++) K.v[i] = h_K[i];
    VsProfScope ps(ctx, "triangulate_kernel");
    triangulate_kernel<<<dim3(vs_div_up(kp_stride, 128), batch), 128, 0, ctx->stream>>>(xy1, xy2, matches, best, kp_stride, K,
                                                                                        c2, points4d);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_triangulate_points(vslam_ctx *ctx, const float *p1, const float *p2, int n, const float *h_c1, const float *h_c2,
                                 float *points4d) {
    VS_REQUIRE(ctx, p1 && p2 && h_c1 && h_c2 && points4d, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, n >= 0, VSLAM_ERR_INVALID);
    if (n == 0) return VSLAM_OK;
    Mat34 C1, C2;
    for (int i = 0; i < 12; i++) {
        C1.v[i] = h_c1[i];
        C2.v[i] = h_c2[i];
    }
    VsProfScope ps(ctx, "triangulate_kernel");
    triangulate_points_kernel<<<vs_div_up(n, 128), 128, 0, ctx->stream>>>(p1, p2, n, C1, C2, points4d);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// ORACLE (test infrastructure only) — cv::SVDecomp for CV_32F, restated.
//
// The reference calls cv::SVDecomp(A, D, U, V_t, MODIFY_A | FULL_UV) twice per hypothesis
// (/root/reference/src/RansacFilter.cpp:94,98).  OpenCV is a third-party dependency that is
// absent from this container (version unpinned: makefile:4,7), so this file restates the
// PUBLISHED built-in algorithm of OpenCV 4.x, modules/core/src/lapack.cpp:
//   _SVDcompute      — transposes so the working matrix has rows = min(m,n) vectors of
//                      length max(m,n); FULL_UV asks for max(m,n) rows of "U".
//   JacobiSVDImpl_   — one-sided (Hestenes) Jacobi on the rows of At: squared row norms and
//                      row dot products accumulate in double, the rotation (c, s) and the
//                      rotated rows are float; at most max(m,30) sweeps; singular values
//                      sorted descending by selection sort with row swaps; rows beyond the
//                      rank (and the extra FULL_UV rows) are filled from cv::RNG(0x12345678)
//                      sign patterns, orthogonalised against the previous rows twice, and
//                      L2-normalised.
// [OpenCV, from memory] — PARITY UNPINNED.  Two deliberate pins of implementation-defined
// behaviour, shared with the HIP kernels so CPU and GPU agree bit for bit:
//   * std::hypot(p, beta) is replaced by vso_hypot() = sqrt(p*p + beta*beta) in double
//     (libm's hypot is not specified to the last bit; build with -DVSO_LIBM_HYPOT to use it
//     and see tests/test_oracle_svd.py::test_hypot_pin_is_benign).
//   * a build where OpenCV routes SVDecomp to LAPACK sgesdd is a different algorithm and is
//     out of reach here.
#include "vso.h"
#include "vso_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace vso {

// what the last jacobi_svd32f call on this thread did (tools/solve_flops.py counts the solver's work with it)
thread_local int g_last_sweeps = 0, g_last_visits = 0, g_last_rotations = 0;

double pinned_hypot(double a, double b) {
#ifdef VSO_LIBM_HYPOT
    return std::hypot(a, b);
#else
    return std::sqrt(a * a + b * b);
#endif
}

// cv::RNG: multiply-with-carry, modules/core/include/opencv2/core/operations.hpp
struct MwcRng {
    uint64_t state;
    explicit MwcRng(uint64_t s) : state(s) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
};

// JacobiSVDImpl_<float>.  At: n rows (stride astep floats) of length m, holding the vectors
// to orthogonalise; n1 rows of At are normalised/filled on exit (n1 >= n for FULL_UV).
// Vt: n x n (stride vstep) or NULL.
void jacobi_svd32f(float *At, size_t astep, float *_W, float *Vt, size_t vstep, int m, int n,
                   int n1) {
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    std::vector<double> Wbuf(n);
    double *W = Wbuf.data();
    int i, j, k, iter;
    const int max_iter = std::max(m, 30);
    float c, s;
    double sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }

    g_last_sweeps = g_last_visits = g_last_rotations = 0;
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        g_last_sweeps++;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                float *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                g_last_visits++;
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                g_last_rotations++;

                p *= 2;
                double beta = a - b, gamma = pinned_hypot((double)p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }

                a = b = 0;
                for (k = 0; k < m; k++) {
                    float t0 = c * Ai[k] + s * Aj[k];
                    float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += (double)t0 * t0;
                    b += (double)t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;

                if (Vt) {
                    float *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        float t0 = c * Vi[k] + s * Vj[k];
                        float t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
        }
        W[i] = std::sqrt(sd);
    }

    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            if (Vt) {
                for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
                for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
            }
        }
    }

    for (i = 0; i < n; i++) _W[i] = (float)W[i];
    if (!Vt) return;

    MwcRng rng(0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            // zero singular value: random +-1/m vector, projected off the previous rows
            const float val0 = (float)(1. / m);
            for (k = 0; k < m; k++) {
                float val = (rng.next() & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    float asum = 0;
                    for (k = 0; k < m; k++) {
                        float t = (float)(At[i * astep + k] - sd * At[j * astep + k]);
                        At[i * astep + k] = t;
                        asum += std::abs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                float t = At[i * astep + k];
                sd += (double)t * t;
            }
            sd = std::sqrt(sd);
        }
        s = (float)(sd > minval ? 1 / sd : 0.);
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

// _SVDcompute with flags = MODIFY_A | FULL_UV on a CV_32F m x n matrix.
void svd32f_full(const float *A, int m0, int n0, float *w, float *u, float *vt) {
    int m = m0, n = n0;
    bool at = false;
    if (m < n) {
        std::swap(m, n);
        at = true;
    }
    const int urows = m;                      // FULL_UV
    // temp_a: n x m (rows = vectors to orthogonalise), stored inside temp_u (urows x m)
    std::vector<float> ubuf((size_t)urows * m, 0.f);   // `temp_u = Scalar::all(0)` when urows > n
    std::vector<float> vbuf((size_t)n * n, 0.f);
    std::vector<float> wbuf(n);
    if (!at) {
        for (int r = 0; r < m0; r++)
            for (int c = 0; c < n0; c++) ubuf[(size_t)c * m + r] = A[(size_t)r * n0 + c];   // transpose(src, temp_a)
    } else {
        for (int r = 0; r < m0; r++)
            for (int c = 0; c < n0; c++) ubuf[(size_t)r * m + c] = A[(size_t)r * n0 + c];   // src.copyTo(temp_a)
    }
    jacobi_svd32f(ubuf.data(), (size_t)m, wbuf.data(), vbuf.data(), (size_t)n, m, n, urows);
    std::memcpy(w, wbuf.data(), sizeof(float) * n);
    if (!at) {
        // u = temp_u^T (m x m), vt = temp_v (n x n)
        if (u)
            for (int r = 0; r < m; r++)
                for (int c = 0; c < m; c++) u[(size_t)r * m + c] = ubuf[(size_t)c * m + r];
        if (vt) std::memcpy(vt, vbuf.data(), sizeof(float) * n * n);
    } else {
        // u = temp_v^T (m0 x m0 == n x n), vt = temp_u (n0 x n0 == m x m)
        if (u)
            for (int r = 0; r < n; r++)
                for (int c = 0; c < n; c++) u[(size_t)r * n + c] = vbuf[(size_t)c * n + r];
        if (vt) std::memcpy(vt, ubuf.data(), sizeof(float) * m * m);
    }
}

}  // namespace vso

extern "C" int vso_svd32f_full(const float *A, int m, int n, float *w, float *u, float *vt) {
    if (!A || m <= 0 || n <= 0 || !w) return -1;
    vso::svd32f_full(A, m, n, w, u, vt);
    return 0;
}

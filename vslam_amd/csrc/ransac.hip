// RANSAC fundamental-matrix loop for gfx950.
//
// Replaces RansacFilter (/root/reference/src/RansacFilter.cpp):
//   initialize_sets              :6-34    -> ransac_mt_kernel + ransac_map_kernel (mt19937, then Lemire + draws, on device)
//   compute_fundamental          :69-103  -> ransac_solve_kernel  (one lane per hypothesis)
//   compute_fundamental_residual :105-140 -> ransac_score_kernel  (one lane per hypothesis,
//                                            matches broadcast from LDS)
//   find_fundamental             :36-67   -> ransac_select_kernel (argmax with the reference's
//                                            sequential accept rule, winner's mask, and the
//                                            inlier filter of src/Frame.cpp:96-102)
//
// Numerics: the reference's two cv::SVDecomp calls are OpenCV's one-sided Jacobi with double
// accumulators and float rotations; the kernels execute the same operations in the same order
// (no FMA contraction: the file is built with -ffp-contract=off; explicit fma() is used only
// where the product of two floats is exact in double, which makes fma == mul+add bit for bit).
// std::hypot is pinned to sqrt(p*p + beta*beta) on both sides.
//
// Why one lane per hypothesis: the per-hypothesis residual sum is a sequential double
// accumulation whose value breaks inlier-count ties (RansacFilter.cpp:59,138).  A lane that
// walks the matches in index order reproduces that sum exactly with no cross-lane reduction;
// the match coordinates are the same for all 64 lanes, so LDS serves them as broadcasts.
#include "ctx.h"

#include <cfloat>

namespace {

// ------------------------------------------------------------------------------------------
// initialize_sets
// ------------------------------------------------------------------------------------------
constexpr int kSetThreads = 256;
constexpr int kMtN = 624, kMtM = 397;

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7FFFFFFFu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= (y >> 18);
    return y;
}

// The draws come from two kernels.  The raw mt19937 outputs depend on the seed alone (ransac_mt_kernel: one workgroup per
// frame pair, the 624-word state in LDS, a block of 624 outputs per three dependency phases i < 227 | 227 <= i < 454 |
// i >= 454), so the batched front-end produces them on the auxiliary stream while the frames are still being extracted;
// only the mapping to draws, which needs the match count, sits between the matcher and the solver (ransac_map_kernel).
// A pair consumes hyp * 8 raw outputs plus one per Lemire rejection (p ~ n / 2^32 each, i.e. a handful per batch);
// kMtSpare extra outputs are generated, and a pair that would need more raises bit 1 of the context's device error word
// (vslam_ctx_synchronize reports VSLAM_ERR_CAPACITY) instead of reading past them.
constexpr int kMtSpare = kMtN;
static inline int vs_mt_blocks(int hyp) { return (hyp * VSLAM_SET_SIZE + kMtSpare + kMtN - 1) / kMtN; }

__global__ __launch_bounds__(kSetThreads) void ransac_mt_kernel(const uint32_t *__restrict__ seeds, int nblk,
                                                                uint32_t *__restrict__ raw) {
    const int b = blockIdx.x, tid = threadIdx.x;
    uint32_t *R = raw + (size_t)b * nblk * kMtN;
    __shared__ uint32_t mt[kMtN];
    if (tid == 0) {   // std::mt19937(seed) seeding recurrence
        uint32_t x = seeds[b];
        mt[0] = x;
        for (int i = 1; i < kMtN; i++) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            mt[i] = x;
        }
    }
    __syncthreads();
    for (int blk = 0; blk < nblk; blk++) {
        uint32_t v = 0;
        if (tid < kMtN - kMtM) v = mt_twist(mt[tid], mt[tid + 1], mt[tid + kMtM]);
        __syncthreads();
        if (tid < kMtN - kMtM) mt[tid] = v;
        __syncthreads();
        const int i1 = tid + (kMtN - kMtM);   // 227 .. 453
        if (tid < kMtN - kMtM) v = mt_twist(mt[i1], mt[i1 + 1], mt[i1 - (kMtN - kMtM)]);
        __syncthreads();
        if (tid < kMtN - kMtM) mt[i1] = v;
        __syncthreads();
        const int i2 = tid + 2 * (kMtN - kMtM);   // 454 .. 623
        if (i2 < kMtN) v = mt_twist(mt[i2], mt[i2 == kMtN - 1 ? 0 : i2 + 1], mt[i2 - (kMtN - kMtM)]);
        __syncthreads();
        if (i2 < kMtN) mt[i2] = v;
        __syncthreads();
        for (int i = tid; i < kMtN; i += kSetThreads) R[(size_t)blk * kMtN + i] = mt_temper(mt[i]);
    }
}

// Mapping of the raw outputs to draws: libstdc++'s uniform_int_distribution<int>(0, size-1) for a 32-bit URBG is Lemire's
// multiply-shift with rejection (bits/uniform_int_dist.h _S_nd); a rejected output is consumed and the same draw retries
// with the next one, so a rejection shifts every later draw by one raw output.  With all outputs in memory every output
// is mapped in parallel under the current count of rejections; the earliest rejected output (if any: p ~ n / 2^32 per
// draw) finalises everything before it, bumps the count, and the pass repeats from there.  Usually one pass, a second
// one for about one pair in a hundred.  Then draws -> indices without replacement: available[r] = available.back();
// pop_back() (RansacFilter.cpp:26-31) tracked as a <= 8-entry sparse overlay on the identity array.
constexpr int kMapThreads = 1024;
// mi = RansacFilter::min_items: the reference draws min_items indices into sets that are 8 wide whatever min_items is
// (src/RansacFilter.cpp:17,22): entries mi .. 7 stay 0, and a hypothesis consumes mi raw outputs (+ rejections).
template <bool EIGHT>   // min_items == 8, the usual case: d & 7, no per-entry test
__global__ __launch_bounds__(kMapThreads) void ransac_map_kernel(const int32_t *__restrict__ m_arr, int hyp, int nblk, int mi_arg,
                                                                 const uint32_t *__restrict__ raw, int32_t *__restrict__ sets,
                                                                 uint32_t *__restrict__ draws, int32_t *__restrict__ errflag) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int mi = EIGHT ? VSLAM_SET_SIZE : mi_arg;
    const int n = m_arr[b];
    int32_t *S = sets + (size_t)b * hyp * VSLAM_SET_SIZE;
    uint32_t *D = draws + (size_t)b * hyp * VSLAM_SET_SIZE;
    const uint32_t *R = raw + (size_t)b * nblk * kMtN;
    const int total = hyp * mi, avail = nblk * kMtN;
    if (n < mi || n < 1 || mi == 0) {   // n < mi is UB in the reference (distribution over (0,-1)); defined here as zeros
        for (int i = tid; i < hyp * VSLAM_SET_SIZE; i += kMapThreads) S[i] = 0;
        return;
    }
    __shared__ int s_first;
    int lo = 0, rej = 0;
    while (true) {
        if (tid == 0) s_first = 0x7FFFFFFF;
        __syncthreads();
        if (total + rej > avail) {   // more rejections than spare outputs: practically unreachable; reported, never read past
            if (tid == 0) atomicOr(errflag, 2);
            for (int d = lo - rej + tid; d < total; d += kMapThreads) D[d] = 0;
            break;
        }
        for (int t = lo + tid; t < total + rej; t += kMapThreads) {
            const int d = t - rej;
            const uint32_t range = (uint32_t)(n - (EIGHT ? (d & 7) : d % mi));
            const uint64_t prod = (uint64_t)R[t] * (uint64_t)range;
            const uint32_t low = (uint32_t)prod;
            if (low < range && low < (0u - range) % range) atomicMin(&s_first, t);   // rejected: consumed, yields no draw
            else D[d] = (uint32_t)(prod >> 32);   // final if t lies before the first rejection, rewritten otherwise
        }
        __syncthreads();
        const int first = s_first;
        if (first == 0x7FFFFFFF) break;
        rej += 1;
        lo = first + 1;
        __syncthreads();
    }
    __syncthreads();
    __threadfence_block();
    for (int h = tid; h < hyp; h += kMapThreads) {
        int pos[VSLAM_SET_SIZE], val[VSLAM_SET_SIZE];
        int cnt = 0, size = n;
#pragma unroll
        for (int j = 0; j < VSLAM_SET_SIZE; j++) {
            if (!EIGHT && j >= mi) {
                S[(size_t)h * VSLAM_SET_SIZE + j] = 0;
                continue;
            }
            const int r = (int)D[(size_t)h * mi + j];
            int v = r, lv = size - 1, slot = -1;
#pragma unroll
            for (int k = 0; k < VSLAM_SET_SIZE; k++) {
                if (k < cnt && pos[k] == r) {
                    v = val[k];
                    slot = k;
                }
                if (k < cnt && pos[k] == size - 1) lv = val[k];
            }
            S[(size_t)h * VSLAM_SET_SIZE + j] = v;
            if (slot >= 0) {
#pragma unroll
                for (int k = 0; k < VSLAM_SET_SIZE; k++)
                    if (k == slot) val[k] = lv;
            } else {
#pragma unroll
                for (int k = 0; k < VSLAM_SET_SIZE; k++)
                    if (k == cnt) {
                        pos[k] = r;
                        val[k] = lv;
                    }
                cnt++;
            }
            size--;
        }
    }
}

// ------------------------------------------------------------------------------------------
// compute_fundamental: OpenCV JacobiSVDImpl_<float> on per-lane matrices held in LDS
// ------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 64;

__device__ __forceinline__ uint32_t cvrng_next(uint64_t &state) {   // cv::RNG (MWC)
    state = (uint64_t)(uint32_t)state * 4164903690ull + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

// Element (r,k) of this lane's matrix; lanes are interleaved so every ds access is conflict-free.
#define VS_A(r, k) sA[((r) * M + (k)) * kSolveThreads + tid]
#define VS_V(r, k) sV[((r) * N + (k)) * kSolveThreads + tid]

// JacobiSVDImpl_(At, astep, W, Vt, vstep, m = M, n = N, n1 = N1, FLT_MIN, FLT_EPSILON*2).
// Rows 0..N-1 of A are orthogonalised; rows up to N1-1 are normalised / generated.
// wout receives (float)W[i].
// OpenCV keeps the squared row norms W[] between rotations; every W[i] it reads is the k-ordered double
// sum of squares of the CURRENT row i (set so at start-up and after each rotation of that row), so the
// norms are recomputed from the rows where needed instead of being stored: same bits, and the LDS
// footprint per lane drops by 64 B, which is what bounds this kernel's occupancy.
// fabs(p) <= eps * sqrt(a * b) with eps = 2 * FLT_EPSILON = 2^-22, decided without the square root where that is
// safe.  With y = |p| * 2^22 (exact) and x = fl(a * b) the test is y <= RN(sqrt(x)); rounding is monotone, so it
// equals y * y <= x except when x lies within an ulp or so of y * y.  h = fl(y * y) decides every case in which h
// and x differ by more than 2^-50 relative; if any lane of the wave is closer than that (or x is 0) the wave
// evaluates the original expression.
__device__ __forceinline__ bool jacobi_converged(double p, double a, double b) {
    const double y = fabs(p) * 4194304.0;
    const double x = a * b;
    const double h = y * y;
    const bool near = !(fabs(h - x) > x * 0x1p-50);   // also true for NaN / zero
    if (__any(near)) return fabs(p) <= (double)(FLT_EPSILON * 2) * sqrt(x);
    return h < x;
}

// sqrt(x) and x / y for operands in a comfortable exponent range: the instruction sequences hipcc emits for the
// IEEE-correct f64 sqrt and division (v_rsq_f64 / v_rcp_f64 + the Goldschmidt / Newton corrections) without
// the parts that only act on extreme exponents, zeros, infinities and NaNs (v_ldexp rescaling and the class
// select for sqrt; v_div_scale, the scale fix-up of v_div_fmas and v_div_fixup for division) — outside those
// cases these leave the value untouched, so the results are the same bits for 37 resp. 16 fewer issue cycles.
__device__ __forceinline__ double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double div_inrange(double x, double y) {
    double r = __builtin_amdgcn_rcp(y);
    double f = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, f, r);
    f = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, f, r);
    const double q = x * r;
    const double e = __builtin_fma(-y, q, x);
    return __builtin_fma(e, r, q);
}

// Everything after the sweeps: W, the descending sort (rows travel with W), normalisation / regeneration of the rows,
// the row beyond the rank.  The matrix is reached through pA / pV with element stride SA, so the same code runs on a
// lane's LDS columns (SA = kSolveThreads) and on a private copy (SA = 1).
#define FA(r, k) pA[((r) * M + (k)) * SA]
#define FV(r, k) pV[((r) * N + (k)) * SA]
template <int M, int N, int N1, bool HASV, int SA>
__device__ __forceinline__ void jacobi_finish(float *pA, float *pV, float *wout, float *extra_row) {
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    double W[N];   // singular values: registers, every index below is compile-time
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
            const float t = FA(i, k);
            sd = __builtin_fma((double)t, (double)t, sd);
        }
        W[i] = sqrt(sd);
    }

#pragma unroll
    for (int i = 0; i < N - 1; i++) {   // selection sort, descending, rows travel with W
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) {
                j = k;
                wj = W[k];
            }
        if (i != j) {
#pragma unroll
            for (int jj = i + 1; jj < N; jj++)
                if (jj == j) W[jj] = W[i];
            W[i] = wj;
#pragma unroll
            for (int k = 0; k < M; k++) {
                const float x = FA(i, k), y = FA(j, k);
                FA(i, k) = y;
                FA(j, k) = x;
            }
            if (HASV) {
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const float x = FV(i, k), y = FV(j, k);
                    FV(i, k) = y;
                    FV(j, k) = x;
                }
            }
        }
    }

#pragma unroll
    for (int i = 0; i < N; i++) wout[i] = (float)W[i];

    uint64_t rng = 0x12345678ull;
    for (int i = 0; i < N; i++) {
        double sd = 0;   // W[i] again: the sorted row's norm
#pragma unroll
        for (int k = 0; k < M; k++) {
            const float t = FA(i, k);
            sd = __builtin_fma((double)t, (double)t, sd);
        }
        sd = sqrt(sd);
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const float val0 = (float)(1. / M);
#pragma unroll
            for (int k = 0; k < M; k++) FA(i, k) = (cvrng_next(rng) & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; iter++) {
                for (int j = 0; j < i; j++) {
                    float vi[M], vj[M];
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        vi[k] = FA(i, k);
                        vj[k] = FA(j, k);
                        sd += (double)(vi[k] * vj[k]);   // float product, double running sum
                    }
                    float asum = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        const float t = (float)((double)vi[k] - sd * (double)vj[k]);
                        vi[k] = t;
                        asum += fabsf(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < M; k++) FA(i, k) = vi[k] * asum;
                }
            }
            sd = 0;
#pragma unroll
            for (int k = 0; k < M; k++) {
                const float t = FA(i, k);
                sd = __builtin_fma((double)t, (double)t, sd);
            }
            sd = sqrt(sd);
        }
        const float s = (float)(sd > minval ? 1 / sd : 0.);
#pragma unroll
        for (int k = 0; k < M; k++) FA(i, k) = FA(i, k) * s;
    }
    if (N1 > N) {
        // the row beyond the rank (FULL_UV): same procedure with i = N, W = 0; it lives in registers,
        // which keeps the per-lane LDS footprint at N rows
        float v[M];
        double sd = 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const float val0 = (float)(1. / M);
#pragma unroll
            for (int k = 0; k < M; k++) v[k] = (cvrng_next(rng) & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; iter++) {
                for (int j = 0; j < N; j++) {
                    float vj[M];
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        vj[k] = FA(j, k);
                        sd += (double)(v[k] * vj[k]);
                    }
                    float asum = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        const float t = (float)((double)v[k] - sd * (double)vj[k]);
                        v[k] = t;
                        asum += fabsf(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < M; k++) v[k] = v[k] * asum;
                }
            }
            sd = 0;
#pragma unroll
            for (int k = 0; k < M; k++) sd = __builtin_fma((double)v[k], (double)v[k], sd);
            sd = sqrt(sd);
        }
        const float s = (float)(sd > minval ? 1 / sd : 0.);
#pragma unroll
        for (int k = 0; k < M; k++) extra_row[k] = v[k] * s;
    }
}
#undef FA
#undef FV

template <int M, int N, int N1, bool HASV>
__device__ void jacobi_svd_lanes(float *sA, float *sV, int tid, float *wout, float *extra_row) {
    static_assert(N1 == N || N1 == N + 1, "FULL_UV asks for at most one row beyond the rank here");
    constexpr int max_iter = M > 30 ? M : 30;

    if (HASV) {
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int k = 0; k < N; k++) VS_V(i, k) = (i == k) ? 1.f : 0.f;
        }
    }

    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < N - 1; i++)
            for (int j = i + 1; j < N; j++) {
                float ai[M], aj[M];
                double a = 0, p = 0, b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    ai[k] = VS_A(i, k);
                    aj[k] = VS_A(j, k);
                    const double di = (double)ai[k], dj = (double)aj[k];
                    p = __builtin_fma(di, dj, p);
                    a = __builtin_fma(di, di, a);   // W[i]
                    b = __builtin_fma(dj, dj, b);   // W[j]
                }
                if (jacobi_converged(p, a, b)) continue;

                p *= 2;
                const double beta = a - b;
                const double g2 = p * p + beta * beta;
                float c, s;
                // With g2 and p in this range every operand and quotient below stays within 2^+-500 (gamma <= 2^200,
                // the two ratios under the square roots lie in [1/2, 1], |p / (gamma * s * 2)| >= 2^-500), where the
                // short sequences equal the full ones; otherwise the whole wave takes sqrt() and '/'.
                const bool safe = g2 > 0x1p-400 && g2 < 0x1p400 && fabs(p) > 0x1p-300;
                if (!__any(!safe)) {
                    const double gamma = sqrt_inrange(g2);   // pinned hypot
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = (float)sqrt_inrange(div_inrange(delta, gamma));
                        c = (float)div_inrange(p, gamma * (double)s * 2);
                    } else {
                        c = (float)sqrt_inrange(div_inrange(gamma + beta, gamma * 2));
                        s = (float)div_inrange(p, gamma * (double)c * 2);
                    }
                } else {
                    const double gamma = sqrt(g2);   // pinned hypot
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = (float)sqrt(delta / gamma);
                        c = (float)(p / (gamma * (double)s * 2));
                    } else {
                        c = (float)sqrt((gamma + beta) / (gamma * 2));
                        s = (float)(p / (gamma * (double)c * 2));
                    }
                }
#pragma unroll
                for (int k = 0; k < M; k++) {
                    const float t0 = c * ai[k] + s * aj[k];
                    const float t1 = (-s) * ai[k] + c * aj[k];
                    VS_A(i, k) = t0;
                    VS_A(j, k) = t1;
                }
                changed = true;
                if (HASV) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const float vi = VS_V(i, k), vj = VS_V(j, k);
                        const float t0 = c * vi + s * vj;
                        const float t1 = (-s) * vi + c * vj;
                        VS_V(i, k) = t0;
                        VS_V(j, k) = t1;
                    }
                }
            }
        if (!changed) break;
    }

    jacobi_finish<M, N, N1, HASV, kSolveThreads>(sA + tid, HASV ? sV + tid : nullptr, wout, extra_row);
}

// ------------------------------------------------------------------------------------------
// The 8 x 9 instance as ransac_solve_kernel runs it: the lane's whole matrix in registers during the sweeps.
// ------------------------------------------------------------------------------------------
// 72 floats per lane in LDS held the kernel at 8 waves per CU (2 per SIMD); it is bound by dependent f64 chains on the
// vector pipe, where more waves pay (2 -> 3 per SIMD: 0.96 -> 0.88 ms with three of the rows in registers).  With all
// eight rows in registers (72 VGPRs; every (i, j) step written out, the row indices compile-time) the sweeps touch no
// LDS at all, 128 VGPRs suffice and 4 waves fit.  Only the part after the sweeps wants rows by run-time index: rows
// 0..3 go to LDS for it (36 floats per lane).  Every operation on every value is the one
// jacobi_svd_lanes<9, 8, 9, false> (the plain restatement above) would perform, in the same order.
constexpr int kSolveLdsRows = 4;
constexpr int kSolveSplitFloats = kSolveLdsRows * 9;

// c, s of a rotation for operands outside the range the short sqrt / division sequences cover (see jacobi_svd_lanes).
// Deliberately a real call: it is reached by a wave-wide vote that practically never passes, and 28 inlined copies of the
// IEEE sqrt and division expansions would double the size of the sweep loop.
__device__ __attribute__((noinline)) void jacobi_cs_full(double p, double beta, double g2, float *c_out, float *s_out) {
    float c, s;
    const double gamma = sqrt(g2);   // pinned hypot
    if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        s = (float)sqrt(delta / gamma);
        c = (float)(p / (gamma * (double)s * 2));
    } else {
        c = (float)sqrt((gamma + beta) / (gamma * 2));
        s = (float)(p / (gamma * (double)c * 2));
    }
    *c_out = c;
    *s_out = s;
}

// one (i, j) step on two rows held in registers.  Returns whether this lane rotated (the rows are then the rotated ones).
__device__ __forceinline__ bool jacobi_pair_9(float (&ai)[9], float (&aj)[9]) {
    double a = 0, p = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const double di = (double)ai[k], dj = (double)aj[k];
        p = __builtin_fma(di, dj, p);
        a = __builtin_fma(di, di, a);   // W[i]
        b = __builtin_fma(dj, dj, b);   // W[j]
    }
    if (jacobi_converged(p, a, b)) return false;
    p *= 2;
    const double beta = a - b;
    const double g2 = p * p + beta * beta;
    float c, s;
    // see jacobi_svd_lanes for the range argument
    const bool safe = g2 > 0x1p-400 && g2 < 0x1p400 && fabs(p) > 0x1p-300;
    if (!__any(!safe)) {
        // One instruction stream for both signs of beta (lanes of a wave differ in it, so as two branches both would run):
        // the first of (c, s) is a square root of a quotient, the second a quotient by it; which is which, and the operands
        // of the first quotient, are selected.  The operations on every lane are the ones its branch would perform.
        const double gamma = sqrt_inrange(g2);   // pinned hypot
        const bool neg = beta < 0;
        const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
        const double den = neg ? gamma : gamma * 2;
        const float first = (float)sqrt_inrange(div_inrange(num, den));
        const float second = (float)div_inrange(p, gamma * (double)first * 2);
        s = neg ? first : second;
        c = neg ? second : first;
    } else {
        jacobi_cs_full(p, beta, g2, &c, &s);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float t0 = c * ai[k] + s * aj[k];
        const float t1 = (-s) * ai[k] + c * aj[k];
        ai[k] = t0;
        aj[k] = t1;
    }
    return true;
}

// One sweep over all pairs in OpenCV's order, all 28 steps written out so that the row indices are compile-time.
// (Round 3 tried the re-rolled form the 60 % "waiting for an instruction" of this kernel seemed to ask for — the seven
// steps of one row written out, the rows themselves shifted one position after each row's turn, 72 register moves, loop
// 13 KB instead of 40 KB: bit-identical, and 0.92 ms instead of 0.75.  The waits are the dependent f64 chains of a
// rotation's parameters — two square roots and two divisions in sequence, about a hundred dependent instructions with
// four waves per SIMD to hide them — not instruction fetch.)
__device__ __forceinline__ bool jacobi_sweep_8x9_regs(float (&R)[8][9]) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 7; i++)
#pragma unroll
        for (int j = i + 1; j < 8; j++) changed |= jacobi_pair_9(R[i], R[j]);
    return changed;
}

// After the sweeps: V_t.row(8) of SVDecomp(A 8x9, FULL_UV) = the row beyond the rank, orthogonalised against the sorted,
// normalised rows (jacobi_finish<9, 8, 9, false, .> with only extra_row kept).  If a row's norm does not exceed FLT_MIN
// OpenCV regenerates that row from its RNG stream first (degenerate samples): any lane in that case sends the wave
// through jacobi_finish itself on a private copy.  Otherwise a row's normalisation factor is a function of that row alone
// and the sort only fixes the ORDER in which the rows are visited, so nothing has to move: rowid[ii] = the row at sorted
// position ii.  Rows below kSolveLdsRows are parked in LDS (pA = this lane's column, element (r, k) at
// pA[(9 r + k) * 64]) so that a run-time row id is an address; the others are picked with selects.
__device__ __forceinline__ void jacobi_null_row_8x9(float *pA, float (&R)[8][9], float *f0) {
    constexpr int M = 9, N = 8, L = kSolveLdsRows;
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    double W[N];
    bool tiny = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd = __builtin_fma((double)R[i][k], (double)R[i][k], sd);
        W[i] = sqrt(sd);
        tiny = tiny || W[i] <= minval;
    }
    if (__any(tiny)) {
        float rows[N * M], w8[N];
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int k = 0; k < M; k++) rows[i * M + k] = R[i][k];
        jacobi_finish<9, 8, 9, false, 1>(rows, nullptr, w8, f0);
        return;
    }
    // normalise (row *= (float)(1 / W), W = the row's norm), park the first rows
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float s = (float)(1 / W[i]);
#pragma unroll
        for (int k = 0; k < M; k++) {
            R[i][k] = R[i][k] * s;
            if (i < L) pA[(i * 9 + k) * kSolveThreads] = R[i][k];
        }
    }
    // the descending selection sort, on (W, row id) pairs
    int rowid[N];
#pragma unroll
    for (int i = 0; i < N; i++) rowid[i] = i;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) {
                j = k;
                wj = W[k];
            }
        if (i != j) {
            const int ri = rowid[i];
            int rj = ri;
#pragma unroll
            for (int jj = i + 1; jj < N; jj++)
                if (jj == j) {
                    W[jj] = W[i];
                    rj = rowid[jj];
                    rowid[jj] = ri;
                }
            W[i] = wj;
            rowid[i] = rj;
        }
    }
    // the row beyond the rank: jacobi_finish's N1 > N block, rows visited in sorted order
    uint64_t rng = 0x12345678ull;
    float v[M];
    double sd = 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
        const float val0 = (float)(1. / M);
#pragma unroll
        for (int k = 0; k < M; k++) v[k] = (cvrng_next(rng) & 256) != 0 ? val0 : -val0;
        for (int iter = 0; iter < 2; iter++) {
#pragma unroll
            for (int jj = 0; jj < N; jj++) {
                const int r = rowid[jj];
                const int rl = r < L ? r : 0;
                float vj[M];
                sd = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    float t = pA[(rl * 9 + k) * kSolveThreads];
#pragma unroll
                    for (int q = L; q < N; q++) t = r == q ? R[q][k] : t;
                    vj[k] = t;
                    sd += (double)(v[k] * vj[k]);   // float product, double running sum
                }
                float asum = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    const float t = (float)((double)v[k] - sd * (double)vj[k]);
                    v[k] = t;
                    asum += fabsf(t);
                }
                asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                for (int k = 0; k < M; k++) v[k] = v[k] * asum;
            }
        }
        sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd = __builtin_fma((double)v[k], (double)v[k], sd);
        sd = sqrt(sd);
    }
    const float s = (float)(sd > minval ? 1 / sd : 0.);
#pragma unroll
    for (int k = 0; k < M; k++) f0[k] = v[k] * s;
}

// The part of compute_fundamental behind the first SVD (src/RansacFilter.cpp:98-101): the 3 x 3 SVD of F0 (working rows =
// its columns), D[2] = 0, F = U diag(D) Vt.  sA / sV: this wave's 9 + 9 float columns in LDS; `tid` = the lane.
__device__ __forceinline__ void fundamental_rank2(const float (&f0)[9], float *sA, float *sV, int tid, float (&F)[9]) {
    {
        constexpr int M = 3;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) VS_A(i, k) = f0[3 * k + i];
    }
    float d3[3];
    jacobi_svd_lanes<3, 3, 3, true>(sA, sV, tid, d3, nullptr);
    d3[2] = 0.f;   // :99

    float U[9], Vt[9];
    {
        constexpr int M = 3, N = 3;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                U[r * 3 + c] = VS_A(c, r);   // u = transpose(temp_u)
                Vt[r * 3 + c] = VS_V(r, c);
            }
    }
    // temp_F = U * diag(D) * V_t (:101) through OpenCV's 3x3 float fast path (a0*b0 + a1*b1 + a2*b2)
    const float Dg[9] = {d3[0], 0.f, 0.f, 0.f, d3[1], 0.f, 0.f, 0.f, d3[2]};
    float UD[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            UD[i * 3 + j] = U[i * 3 + 0] * Dg[0 * 3 + j] + U[i * 3 + 1] * Dg[1 * 3 + j] + U[i * 3 + 2] * Dg[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            F[i * 3 + j] = UD[i * 3 + 0] * Vt[0 * 3 + j] + UD[i * 3 + 1] * Vt[1 * 3 + j] + UD[i * 3 + 2] * Vt[2 * 3 + j];
}

// One lane per hypothesis, one wave per workgroup.  grid = (ceil(hyp / 64), batch).
// SPLIT (round 5, VSLAM_RANSAC_SOLVE_SPLIT): the kernel stops behind the first SVD and leaves F0 = V_t.row(8) in hypF;
// ransac_close_kernel turns it into F in place.  WPE: waves per SIMD the register budget is cut for.
template <bool SPLIT, int WPE>
__global__ __launch_bounds__(kSolveThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ransac_solve_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, const int32_t *__restrict__ sets, int kp_stride, int hyp,
    float *__restrict__ hypF) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int h = blockIdx.x * kSolveThreads + tid;
    if (m_arr[b] < min_m) return;   // uniform per workgroup (8 unless RansacFilter::min_items is smaller)

    __shared__ float sA[kSolveSplitFloats * kSolveThreads];

    const bool live = h < hyp;
    const int hc = live ? h : hyp - 1;   // idle lanes redo the last hypothesis; no divergence in barriers
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;
    const int32_t *S = sets + ((size_t)b * hyp + hc) * VSLAM_SET_SIZE;

    float R[8][9];
#pragma unroll
    for (int r = 0; r < 8; r++) {   // design matrix, RansacFilter.cpp:75-90
        const int2 pr = PR[S[r]];
        const float2 a = P1[pr.x], c = P2[pr.y];
        const float u1 = a.x, v1 = a.y, u2 = c.x, v2 = c.y;
        R[r][0] = u2 * u1;
        R[r][1] = u2 * v1;
        R[r][2] = u2;
        R[r][3] = v2 * u1;
        R[r][4] = v2 * v1;
        R[r][5] = v2;
        R[r][6] = u1;
        R[r][7] = v1;
        R[r][8] = 1.f;
    }

    float f0[9];
    {   // SVDecomp(A 8x9), :94; f0 = V_t.row(8), :95
        constexpr int max_iter = 30;
        for (int iter = 0; iter < max_iter; iter++)
            if (!jacobi_sweep_8x9_regs(R)) break;
        jacobi_null_row_8x9(sA + tid, R, f0);
    }

    float F[9];
    if (SPLIT) {
#pragma unroll
        for (int k = 0; k < 9; k++) F[k] = f0[k];
    } else {
        fundamental_rank2(f0, sA, sA + 9 * kSolveThreads, tid, F);   // second SVD on the 3x3 (:98) and F = U diag Vt (:101)
    }
    if (live) {
        float *o = hypF + ((size_t)b * hyp + h) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = F[k];
    }
}

#ifdef VSLAM_EXPERIMENTS
// The closing part of the split form (experiments build; measured slower than the one-kernel solve): four waves per
// workgroup, a lane per hypothesis, F0 in, F out, in place.
constexpr int kCloseThreads = 256;
__global__ __launch_bounds__(kCloseThreads) void ransac_close_kernel(const int32_t *__restrict__ m_arr, int min_m, int hyp,
                                                                     float *__restrict__ hypF) {
    const int b = blockIdx.y;
    if (m_arr[b] < min_m) return;
    __shared__ float s[kCloseThreads / 64][18 * kSolveThreads];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.x * kCloseThreads + threadIdx.x;
    const bool live = h < hyp;
    float *o = hypF + ((size_t)b * hyp + (live ? h : 0)) * 9;
    float f0[9], F[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f0[k] = live ? o[k] : (k % 4 == 0 ? 1.f : 0.f);   // lanes past the end work on the identity: no read of a slot a live lane rewrites
    fundamental_rank2(f0, s[wave], s[wave] + 9 * kSolveThreads, lane, F);
    if (live) {
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = F[k];
    }
}
#endif   // VSLAM_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// compute_fundamental, opt-in approximate form: the 8-point system as a dense contraction on the matrix cores
// (BASELINE.json configs[4]; VSLAM_OPT_RANSAC_SOLVER = 1).  NOT bit-exact with the reference and never the default:
// the parity bar pins the default solver to OpenCV's sequential Jacobi sweeps (ransac_solve_kernel above).
//   1. Hartley-style conditioning with ONE similarity per frame of the pair (centroid and mean distance of the
//      pair's matched points; ransac_condition_kernel), so the 9x9 normal matrix is formed from O(1) numbers;
//   2. G = A^T A (9 x 9, A the 8 x 9 design matrix) on v_mfma_f32_16x16x4_f32: the A operand and the B operand of the
//      instruction are the same register (lane l holds design[row 4 step + (l >> 4)][column l & 15]), two
//      instructions per hypothesis, results through LDS to the lane that owns the hypothesis;
//   3. the null vector of A = the eigenvector of G's smallest eigenvalue, by inverse iteration on G + mu I
//      (Cholesky + 4 solves in f64, one lane per hypothesis);
//   4. back to pixel coordinates (F0 = T2^T F^ T1), unit Frobenius norm like V_t.row(8), then the SAME rank-2 step as
//      the exact kernel (jacobi_svd_lanes<3,3,3,true>, RansacFilter.cpp:98-101).
// F agrees with the exact solver up to sign and rounding (tests/test_gpu_ransac.py states the tolerance and measures the
// inlier-mask agreement); on degenerate samples (rank < 8) the two pick different vectors of the null space.
typedef float v4f __attribute__((ext_vector_type(4)));

// the pair's two similarities: cond[b] = (cx1, cy1, s1, cx2, cy2, s2); one workgroup per pair
__global__ __launch_bounds__(256) void ransac_condition_kernel(const float *__restrict__ xy1, const float *__restrict__ xy2,
                                                               const int32_t *__restrict__ pairs, const int32_t *__restrict__ m_arr,
                                                               int kp_stride, float *__restrict__ cond) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int m = min(m_arr[b], kp_stride);
    if (m < VSLAM_SET_SIZE) return;
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;
    __shared__ double red[4][4];
    __shared__ float cen[4];
    auto block_sum4 = [&](double v0, double v1, double v2, double v3, double out[4]) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v0 += __shfl_xor(v0, off, 64); v1 += __shfl_xor(v1, off, 64);
            v2 += __shfl_xor(v2, off, 64); v3 += __shfl_xor(v3, off, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) {
            red[tid >> 6][0] = v0; red[tid >> 6][1] = v1; red[tid >> 6][2] = v2; red[tid >> 6][3] = v3;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    };
    double s[4] = {0, 0, 0, 0}, tot[4];
    for (int i = tid; i < m; i += 256) {
        const int2 pr = PR[i];
        const float2 a = P1[pr.x], c = P2[pr.y];
        s[0] += a.x; s[1] += a.y; s[2] += c.x; s[3] += c.y;
    }
    block_sum4(s[0], s[1], s[2], s[3], tot);
    if (tid < 4) cen[tid] = (float)(tot[tid] / m);
    __syncthreads();
    const float cx1 = cen[0], cy1 = cen[1], cx2 = cen[2], cy2 = cen[3];
    double d1 = 0, d2 = 0;
    for (int i = tid; i < m; i += 256) {
        const int2 pr = PR[i];
        const float2 a = P1[pr.x], c = P2[pr.y];
        d1 += sqrtf((a.x - cx1) * (a.x - cx1) + (a.y - cy1) * (a.y - cy1));
        d2 += sqrtf((c.x - cx2) * (c.x - cx2) + (c.y - cy2) * (c.y - cy2));
    }
    block_sum4(d1, d2, 0, 0, tot);
    if (tid == 0) {
        float *o = cond + (size_t)b * 6;
        o[0] = cx1; o[1] = cy1; o[2] = tot[0] > 0 ? (float)(1.4142135623730951 * m / tot[0]) : 1.f;
        o[3] = cx2; o[4] = cy2; o[5] = tot[1] > 0 ? (float)(1.4142135623730951 * m / tot[1]) : 1.f;
    }
}

__global__ __launch_bounds__(kSolveThreads) void ransac_solve_gram_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, const int32_t *__restrict__ sets, const float *__restrict__ cond, int kp_stride, int hyp,
    float *__restrict__ hypF) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int h = blockIdx.x * kSolveThreads + tid;
    const int m = min(m_arr[b], kp_stride);
    if (m < VSLAM_SET_SIZE) return;   // uniform per workgroup

    __shared__ float4 s_pts[kSolveThreads * 8];          // conditioned (u1, v1, u2, v2) of the 8 points of each hypothesis
    __shared__ float s_G[kSolveThreads * 45];            // lower triangle of G per hypothesis; the 3x3 solver's scratch afterwards
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;
    const float cx1 = cond[b * 6 + 0], cy1 = cond[b * 6 + 1], sc1 = cond[b * 6 + 2];
    const float cx2 = cond[b * 6 + 3], cy2 = cond[b * 6 + 4], sc2 = cond[b * 6 + 5];

    // ---- conditioned sample points of this lane's hypothesis
    const bool live = h < hyp;
    const int hc = live ? h : hyp - 1;
    const int32_t *S = sets + ((size_t)b * hyp + hc) * VSLAM_SET_SIZE;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int2 pr = PR[S[r]];
        const float2 a = P1[pr.x], c = P2[pr.y];
        s_pts[tid * 8 + r] = make_float4((a.x - cx1) * sc1, (a.y - cy1) * sc1, (c.x - cx2) * sc2, (c.y - cy2) * sc2);
    }
    __syncthreads();

    // ---- 2. G = A^T A on the matrix cores, one hypothesis per pair of instructions
    const int col = tid & 15, grp = tid >> 4;
    for (int hh = 0; hh < kSolveThreads; hh++) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int step = 0; step < 2; step++) {
            const float4 q = s_pts[hh * 8 + 4 * step + grp];   // (u1, v1, u2, v2) of design row 4 step + grp
            // column `col` of the row [u2u1, u2v1, u2, v2u1, v2v1, v2, u1, v1, 1] (RansacFilter.cpp:81-89); 0 beyond 8
            const float left = col < 3 ? q.z : (col < 6 ? q.w : (col < 9 ? 1.f : 0.f));
            const int c3 = col - 3 * (col / 3);
            const float right = c3 == 0 ? q.x : (c3 == 1 ? q.y : 1.f);
            const float e = left * right;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e, e, acc, 0, 0, 0);   // D[i][j] += sum_k design[k][i] design[k][j]
        }
        // D: column = lane & 15, row = 4 (lane >> 4) + register; G is symmetric, the lower triangle is kept
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int row = 4 * grp + v;
            if (row < 9 && col <= row) s_G[hh * 45 + row * (row + 1) / 2 + col] = acc[v];
        }
    }
    __syncthreads();

    // ---- 3. smallest eigenvector of G by inverse iteration on G + mu I (f64, this lane's hypothesis)
    double L[45];   // lower triangle, row-major: L[i(i+1)/2 + j]
    double tr = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) tr += (double)s_G[tid * 45 + i * (i + 1) / 2 + i];
    const double mu = tr * 1e-7 + 1e-30;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double v = (double)s_G[tid * 45 + i * (i + 1) / 2 + j] + (i == j ? mu : 0.0);
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
            L[i * (i + 1) / 2 + j] = i == j ? sqrt(v) : v / L[j * (j + 1) / 2 + j];
        }
    double x[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] = (i & 1) ? -1.0 / 3.0 : 1.0 / 3.0;
#pragma unroll 1
    for (int it = 0; it < 4; it++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {   // L y = x
            double v = x[i];
#pragma unroll
            for (int k = 0; k < i; k++) v -= L[i * (i + 1) / 2 + k] * x[k];
            x[i] = v / L[i * (i + 1) / 2 + i];
        }
#pragma unroll
        for (int i = 8; i >= 0; i--) {   // L^T z = y
            double v = x[i];
#pragma unroll
            for (int k = i + 1; k < 9; k++) v -= L[k * (k + 1) / 2 + i] * x[k];
            x[i] = v / L[i * (i + 1) / 2 + i];
        }
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nn += x[i] * x[i];
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] *= inv;
    }

    // ---- 4. F0 = T2^T F^ T1 with T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]], then unit norm
    double f0[9];
    {
        const double s1 = sc1, s2 = sc2, tx1 = -(double)sc1 * cx1, ty1 = -(double)sc1 * cy1, tx2 = -(double)sc2 * cx2, ty2 = -(double)sc2 * cy2;
        double Mx[9];   // F^ T1
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Mx[r * 3 + 0] = x[r * 3 + 0] * s1;
            Mx[r * 3 + 1] = x[r * 3 + 1] * s1;
            Mx[r * 3 + 2] = x[r * 3 + 0] * tx1 + x[r * 3 + 1] * ty1 + x[r * 3 + 2];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {   // T2^T (F^ T1)
            f0[0 * 3 + c] = s2 * Mx[0 * 3 + c];
            f0[1 * 3 + c] = s2 * Mx[1 * 3 + c];
            f0[2 * 3 + c] = tx2 * Mx[0 * 3 + c] + ty2 * Mx[1 * 3 + c] + Mx[2 * 3 + c];
        }
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nn += f0[i] * f0[i];
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int i = 0; i < 9; i++) f0[i] *= inv;
    }
    __syncthreads();   // every lane is done with its G: the buffer becomes the 3x3 solver's scratch

    // the rank-2 step of the exact kernel (RansacFilter.cpp:98-101)
    float *sA = s_G;
    float *sV = s_G + 9 * kSolveThreads;
    {
        constexpr int M = 3;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) VS_A(i, k) = (float)f0[3 * k + i];
    }
    float d3[3];
    jacobi_svd_lanes<3, 3, 3, true>(sA, sV, tid, d3, nullptr);
    d3[2] = 0.f;
    float U[9], Vt[9];
    {
        constexpr int M = 3, N = 3;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                U[r * 3 + c] = VS_A(c, r);
                Vt[r * 3 + c] = VS_V(r, c);
            }
    }
    const float Dg[9] = {d3[0], 0.f, 0.f, 0.f, d3[1], 0.f, 0.f, 0.f, d3[2]};
    float UD[9], F[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            UD[i * 3 + j] = U[i * 3 + 0] * Dg[0 * 3 + j] + U[i * 3 + 1] * Dg[1 * 3 + j] + U[i * 3 + 2] * Dg[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            F[i * 3 + j] = UD[i * 3 + 0] * Vt[0 * 3 + j] + UD[i * 3 + 1] * Vt[1 * 3 + j] + UD[i * 3 + 2] * Vt[2 * 3 + j];
    if (live) {
        float *o = hypF + ((size_t)b * hyp + h) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = F[k];
    }
}
#undef VS_A
#undef VS_V

// ------------------------------------------------------------------------------------------
// compute_fundamental_residual
// ------------------------------------------------------------------------------------------
// e for one correspondence under F, exactly as RansacFilter.cpp:119-126 evaluates it through
// OpenCV: F*x1 in float (left to right), F.t()*x2 in double with one rounding, the row reduce
// as (r0 + r1) + r2, and n*n / a*a + b*b + c*c + d*d with C++ precedence.
struct ResidualF {
    float f[9];
    double ft[6];   // F[0],F[3],F[6], F[1],F[4],F[7] as doubles
};
__device__ __forceinline__ void residual_prepare(ResidualF &R) {
    R.ft[0] = (double)R.f[0];
    R.ft[1] = (double)R.f[3];
    R.ft[2] = (double)R.f[6];
    R.ft[3] = (double)R.f[1];
    R.ft[4] = (double)R.f[4];
    R.ft[5] = (double)R.f[7];
}
__device__ __forceinline__ float residual_e(const ResidualF &R, const float4 c, const double dx2, const double dy2) {
    const float x1 = c.x, y1 = c.y, x2 = c.z, y2 = c.w;
    const float a0 = R.f[0] * x1 + R.f[1] * y1 + R.f[2];
    const float a1 = R.f[3] * x1 + R.f[4] * y1 + R.f[5];
    const float a2 = R.f[6] * x1 + R.f[7] * y1 + R.f[8];
    // products of two floats are exact in double: fma(a,b,c) == a*b + c rounded once
    const float t0 = (float)(__builtin_fma(R.ft[1], dy2, R.ft[0] * dx2) + R.ft[2]);
    const float t1 = (float)(__builtin_fma(R.ft[4], dy2, R.ft[3] * dx2) + R.ft[5]);
    const float n = (x2 * a0 + y2 * a1) + a2;
    const float q = (n * n) / (a0 * a0);
    return ((q + a1 * a1) + t0 * t0) + t1 * t1;
}

// Two correspondences at once: the float multiplies and adds become v_pk_mul_f32 / v_pk_add_f32 (IEEE, same
// roundings as the scalar forms), which halves the float instruction count of the hot loop; the double
// part (F.t()*x2) and the division stay per element.  Lane .x is the lower correspondence index.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f residual_e2(const ResidualF &R, const v2f x1, const v2f y1, const v2f x2, const v2f y2,
                                           const double2 da, const double2 db) {
    const v2f a0 = (R.f[0] * x1 + R.f[1] * y1) + R.f[2];
    const v2f a1 = (R.f[3] * x1 + R.f[4] * y1) + R.f[5];
    const v2f a2 = (R.f[6] * x1 + R.f[7] * y1) + R.f[8];
    v2f t0, t1;
    t0.x = (float)(__builtin_fma(R.ft[1], da.y, R.ft[0] * da.x) + R.ft[2]);
    t0.y = (float)(__builtin_fma(R.ft[1], db.y, R.ft[0] * db.x) + R.ft[2]);
    t1.x = (float)(__builtin_fma(R.ft[4], da.y, R.ft[3] * da.x) + R.ft[5]);
    t1.y = (float)(__builtin_fma(R.ft[4], db.y, R.ft[3] * db.x) + R.ft[5]);
    const v2f n = (x2 * a0 + y2 * a1) + a2;
    const v2f nn = n * n, dd = a0 * a0;
    v2f q;
    q.x = nn.x / dd.x;
    q.y = nn.y / dd.y;
    return ((q + a1 * a1) + t0 * t0) + t1 * t1;
}

constexpr int kScoreThreads = 256;
constexpr int kScoreTile = 1024;   // correspondences per LDS tile (16 B floats + 16 B doubles each: 32 KiB)

// One lane per hypothesis; correspondences gathered once per workgroup into LDS and read back as wave-uniform
// broadcasts, two at a time: corr[2j] = (x1a, x1b, y1a, y1b), corr[2j+1] = (x2a, x2b, y2a, y2b) for the
// correspondences a = 2j, b = 2j+1 of the tile; corrd[i] = ((double)x2, (double)y2) of correspondence i
// (converted once per match, not once per hypothesis).  grid = (ceil(hyp/256), batch).
__global__ __launch_bounds__(kScoreThreads) void ransac_score_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, float threshold,
    const float *__restrict__ hypF, int32_t *__restrict__ hyp_count, float *__restrict__ hyp_sum) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int h = blockIdx.x * kScoreThreads + tid;
    const int m = m_arr[b];
    if (m < min_m) return;

    __shared__ __align__(16) float corr[kScoreTile * 4];
    __shared__ double2 corrd[kScoreTile];
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;

    ResidualF R;
    const int hc = h < hyp ? h : hyp - 1;
    const float *src = hypF + ((size_t)b * hyp + hc) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) R.f[k] = src[k];
    residual_prepare(R);

    int count = 0;
    double total = 0;
    for (int base = 0; base < m; base += kScoreTile) {
        const int rows = min(kScoreTile, m - base);
        __syncthreads();
        for (int i = tid; i < ((rows + 1) & ~1); i += kScoreThreads) {
            const int2 pr = PR[base + min(i, rows - 1)];   // an odd tail is padded with its last correspondence
            const float2 a = P1[pr.x], c = P2[pr.y];
            float *d = corr + (i >> 1) * 8 + (i & 1);
            d[0] = a.x;
            d[2] = a.y;
            d[4] = c.x;
            d[6] = c.y;
            corrd[i] = make_double2((double)c.x, (double)c.y);
        }
        __syncthreads();
        const int full = rows >> 1;
#pragma unroll 2
        for (int j = 0; j < full; j++) {
            const float4 p = *reinterpret_cast<const float4 *>(corr + j * 8);
            const float4 r = *reinterpret_cast<const float4 *>(corr + j * 8 + 4);
            v2f x1, y1, x2, y2;
            x1.x = p.x; x1.y = p.y; y1.x = p.z; y1.y = p.w;
            x2.x = r.x; x2.y = r.y; y2.x = r.z; y2.y = r.w;
            const v2f e = residual_e2(R, x1, y1, x2, y2, corrd[2 * j], corrd[2 * j + 1]);
            count += (e.x <= threshold) ? 1 : 0;   // NaN <= thr is false, :130
            total += (double)e.x;                  // cv::sum in index order, :138
            count += (e.y <= threshold) ? 1 : 0;
            total += (double)e.y;
        }
        if (rows & 1) {
            const int i = rows - 1;
            const float *d = corr + (i >> 1) * 8;
            const double2 d2 = corrd[i];
            const float e = residual_e(R, make_float4(d[0], d[2], d[4], d[6]), d2.x, d2.y);
            count += (e <= threshold) ? 1 : 0;
            total += (double)e;
        }
    }
    if (h < hyp) {
        hyp_count[(size_t)b * hyp + h] = count;
        hyp_sum[(size_t)b * hyp + h] = (float)total;
    }
}

// ------------------------------------------------------------------------------------------
// compute_fundamental_residual, counts first (the default scoring path)
// ------------------------------------------------------------------------------------------
// find_fundamental consults a hypothesis' residual SUM only to break ties among hypotheses whose inlier COUNT
// equals the running maximum (RansacFilter.cpp:59); every other sum is computed by the reference and thrown away,
// and a COUNT below the pair's maximum never reaches the accept rule either.  So the scoring is split:
//   ransac_rank_kernel    exact counts of 8 pilot hypotheses (a first lower bound on the maximum) and, from their
//                         inlier masks, the matches ordered by how many pilots miss them (most-missed first), laid
//                         out as four coordinate arrays;
//   ransac_screen_kernel  every hypothesis on the 128 most-missed matches: how many of those it can have as inliers;
//   ransac_cand_kernel    exact full counts of the (up to 8) hypotheses that do best there -> the bound is now the
//                         pair's maximum count on almost every pair;
//   ransac_count_kernel   exact inlier count of every hypothesis that can reach the maximum (those that do are always
//                         counted in full), from a cheap evaluation of e with a certified error band; the few
//                         evaluations that land inside the band are re-done with the exact sequence (residual_e)
//                         -> these counts are the reference's counts, bit for bit;
//   ransac_ties_kernel    C* = max count per pair; of the hypotheses that reach it, those whose sum can still be
//                         the largest after rounding (the count kernel's cheap sums bound every exact sum);
//   ransac_tiesum_kernel  the exact, index-ordered double sum (cv::sum, :138) for those hypotheses only;
//   ransac_select_kernel  unchanged: it only ever reads the sums of hypotheses whose count is C*
//                         (pruned ones hold -inf).
// ransac_score_kernel above (all counts and all sums, exact) stays behind VSLAM_OPT_RANSAC_ALL_SUMS for callers that
// want every per-hypothesis sum (the per-hypothesis parity tests, RansacFilter::compute_fundamental_residual).
//
// The cheap evaluation.  With u = 2^-24 and everything finite, the reference computes (floats, round to nearest)
//   a_k = (f_3k x1 + f_3k+1 y1) + f_3k+2,  n = (x2 a0 + y2 a1) + a2,  nn = n n,  dd = a0 a0,  q = nn / dd,
//   t_0 = float(double(f0 x2 + f3 y2) + f6),  t_1 likewise,  e = ((q + a1 a1) + t0 t0) + t1 t1.
// The count kernel computes a_k, n, nn, dd with the SAME operations (they are cheap and any other order would need
// a cancellation-dependent bound), and replaces the rest by
//   q~ = nn * v_rcp_f32(dd)                        relative error <= 3.1 u  (rcp: 1 ulp)
//   t~_k = fma(f_3+k, y2, fma(f_k, x2, f_6+k))     |t~_k - t_k| <= 3.1 u S_k,  S_k = |f_k x2| + |f_3+k y2| + |f_6+k|
//   g = fma(t~1, t~1, fma(t~0, t~0, fma(a1, a1, q~)))
// All terms of e are squares, so every rounding is a relative error on a non-negative sum; with
// beta = 4.04 u (S_0 + S_1) (S bounded per hypothesis with the pair's largest |x2|, |y2|) and s = sqrt(thr):
//   g < lo = (s (1 - 2^-20) - 1.01 beta)^2                  =>  e <  thr   (e <= (sqrt(g) + 1.005 beta)^2 (1 + 12 u))
//   g > hi = (s (1 + 2^-20) + 2.5 beta)^2 (1 + 2^-20)       =>  e >  thr   (e >= ((sqrt(g)(1 - 1.7u) - beta)^2 - 2 beta^2)(1 - 9u))
// Anything else — g inside [lo, hi], NaN, a denormal / zero dd (where v_rcp_f32 is not a 1-ulp reciprocal), or a
// hypothesis / pair outside the range the bounds were derived for (|f| <= 2^10, coordinates <= 2^20, thr in
// [2^-20, 2^20], so nothing overflows before the true value does) — is decided by the exact sequence.  On image
// data the band is about 0.3 % of thr wide (beta is a few 1e-3: f32 cancellation in t~), i.e. about one evaluation
// in a thousand, so uncertain evaluations are not handled in place (a wave would leave the fast path for 6 % of its
// evaluations) but queued per wave in LDS as (hypothesis, match) words and evaluated 64 at a time with full lanes.
//
// Bail-out: a hypothesis matters to the accept rule only if its count is the pair's maximum, so work on a hypothesis
// stops once the matches looked at contain more certain outliers than ANY maximum-count hypothesis can have:
// potential inliers seen + all matches not seen < a count some hypothesis of the pair verifiably reaches (`bound`,
// never above the true maximum).  Such hypotheses report -1 (ransac_ties_kernel then writes -1 for every hypothesis
// below the maximum, so the array does not depend on timing).  How soon that happens depends on two things the three
// small kernels in front are there for.  (1) The bound: on clean data a third to two thirds of the hypotheses share one
// count (every true correspondence an inlier) and a handful reach one more; with a bound one short of the maximum that
// whole plateau has to be counted in full, with the maximum itself none of it (tools/bail_sim.py: 54 % -> 29 % of all
// evaluations).  (2) The order: the matches every decent hypothesis misses are looked at first, so the allowance of
// outliers is used up at once and the first real difference decides.  Counts do not depend on the order of the
// matches, the cheap sums only within their certified bound, and ransac_tiesum_kernel sums in list order.
//
// Mapping of the count kernel: a workgroup = 128 hypotheses of a pair; those the screen has not already ruled out are
// handed to its 8 waves one at a time; a wave walks the ranked matches 256 at a time (4 per lane, coordinates staged in
// LDS as four arrays so that a lane's two neighbouring matches are the halves of a packed operand); the hypothesis'
// record (F twice, lo, hi) is a broadcast read from LDS; v_cmp writes lane masks to SGPRs, counting is s_bcnt1 on the
// scalar unit (north_star: ballot / popcount).
constexpr int kCntHyps = 128;
constexpr int kCntWaves = 8;
constexpr int kCntQueue = 320;          // words per wave: the 256 evaluations of one sub-block + 63 carried over
constexpr int kCntLdsMatches = 4096;    // ranked coordinates are staged in LDS up to this many matches (64 KiB); read from memory beyond
constexpr int kScreenMatches = 128;
constexpr int kScreenHyps = 128;        // per workgroup: 4 waves x 32
constexpr int kRankThreads = 512;
constexpr int kPilotHyps = kRankThreads / 64;
constexpr int kCandMax = 8;
constexpr float kCntTinyDD = 0x1p-120f;
// __builtin_amdgcn_fcmpf takes LLVM's FCmpInst::Predicate numbering: 2 = ogt, 4 = olt, 9 = ueq, 12 = ult.  The
// tiny-denominator guards below need ult (true for dd < 2^-120 and for NaN).  Builds from c89bef8 (round 2,
// "counts first") up to cce43df (round 3, where the fix rode along in the bench-parity commit) passed 9 there, so the guard fired only on NaN or dd == 2^-120 exactly, and zero / denormal
// a0^2 were certified through v_rcp_f32 instead of taking the exact path (test_tiny_denominators_take_the_exact_path).
constexpr int kFcmpOGT = 2, kFcmpOLT = 4, kFcmpULT = 12;
// A hypothesis as the counting loop reads it from LDS: F, lo, hi (+ 1 pad): kCntRec floats per hypothesis.  (Storing every
// element of F twice spares the loop nine v_movs per hypothesis — the (f, f) operands of the packed instructions — but
// costs 4 KiB per workgroup, which is the difference between two and three workgroups per CU.)
constexpr int kCntRec = 12;
// a wave's queue: volatile (lanes read what other lanes wrote) and typed as LDS so that the accesses are ds_ instructions
typedef __attribute__((address_space(3))) volatile uint32_t cnt_queue_t;
static_assert(VSLAM_MAX_KP <= 65536, "queue words keep the match index in 16 bits");

__host__ __device__ constexpr int cnt_pad(int n) { return (n + 255) & ~255; }

struct CntBand {
    float lo, hi;
    double beta;
    bool ok;
};
// the certified band of one hypothesis (see above); C1 / C2 = the pair's largest |coordinate| in frame 1 / 2
__device__ __forceinline__ CntBand cnt_band(const float *f, float C1, float C2, float threshold) {
    CntBand B;
    bool ok = threshold >= 0x1p-20f && threshold <= 0x1p20f && C1 <= 0x1p20f && C2 <= 0x1p20f;
#pragma unroll
    for (int k = 0; k < 9; k++) ok = ok && fabsf(f[k]) <= 1024.f;   // false for NaN
    const double S = ((double)fabsf(f[0]) + (double)fabsf(f[3]) + (double)fabsf(f[1]) + (double)fabsf(f[4])) * (double)C2 +
                     (double)fabsf(f[6]) + (double)fabsf(f[7]);
    const double beta = 4.04 * 0x1p-24 * S + 1e-30;
    const double sq = sqrt((double)threshold);
    const double lo_r = sq * (1.0 - 0x1p-20) - 1.01 * beta;
    const double hi_r = sq * (1.0 + 0x1p-20) + 2.5 * beta;
    B.lo = lo_r > 0 ? (float)(lo_r * lo_r * (1.0 - 0x1p-22)) : -1.f;
    B.hi = (float)(hi_r * hi_r * (1.0 + 0x1p-20) * (1.0 + 0x1p-22));
    if (!ok) {
        B.lo = -1.f;       // g >= 0 or NaN: never below lo
        B.hi = INFINITY;   // never above hi: every evaluation takes the exact sequence
    }
    B.beta = beta;
    B.ok = ok;
    return B;
}
__device__ __forceinline__ void cnt_store_record(float *d, const float *f, const CntBand &B) {
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = f[k];
    d[9] = B.lo;
    d[10] = B.hi;
    d[11] = B.ok ? (float)(B.beta * (1.0 + 0x1p-20)) : -1.f;   // beta rounded up (the sum rule's error term); < 0: nothing is certified
}

struct CntRec {
    float f[9];   // the packed instructions take (f, f) operands: a splat of one register is an operand modifier (op_sel_hi)
    float lo, hi;
    float beta;   // >= the band's beta; < 0 when the cheap values of this hypothesis are not certified
};
__device__ __forceinline__ void cnt_load_record(CntRec &R, const float *s_rec, int hh) {
    const float4 *r4 = reinterpret_cast<const float4 *>(s_rec + hh * kCntRec);
    const float4 v0 = r4[0], v1 = r4[1], v2 = r4[2];
    const float f[9] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x};
#pragma unroll
    for (int k = 0; k < 9; k++) R.f[k] = f[k];
    R.lo = v2.y;
    R.hi = v2.z;
    R.beta = v2.w;
}

// the cheap value g for two matches per lane; dd returned for the caller's denormal check
__device__ __forceinline__ v2f cnt_splat(float f) {
    v2f r;
    r.x = f;
    r.y = f;
    return r;
}
__device__ __forceinline__ v2f cnt_cheap(const CntRec &R, const v2f X1, const v2f Y1, const v2f X2, const v2f Y2, v2f &dd) {
    const v2f f0 = cnt_splat(R.f[0]), f1 = cnt_splat(R.f[1]), f2 = cnt_splat(R.f[2]), f3 = cnt_splat(R.f[3]), f4 = cnt_splat(R.f[4]),
              f5 = cnt_splat(R.f[5]), f6 = cnt_splat(R.f[6]), f7 = cnt_splat(R.f[7]), f8 = cnt_splat(R.f[8]);
    const v2f a0 = (f0 * X1 + f1 * Y1) + f2;
    const v2f a1 = (f3 * X1 + f4 * Y1) + f5;
    const v2f a2 = (f6 * X1 + f7 * Y1) + f8;
    const v2f n = (X2 * a0 + Y2 * a1) + a2;
    const v2f nn = n * n;
    dd = a0 * a0;
    v2f r;
    r.x = __builtin_amdgcn_rcpf(dd.x);
    r.y = __builtin_amdgcn_rcpf(dd.y);
    v2f g = __builtin_elementwise_fma(a1, a1, nn * r);
    const v2f t0 = __builtin_elementwise_fma(f3, Y2, __builtin_elementwise_fma(f0, X2, f6));
    const v2f t1 = __builtin_elementwise_fma(f4, Y2, __builtin_elementwise_fma(f1, X2, f7));
    g = __builtin_elementwise_fma(t0, t0, g);
    g = __builtin_elementwise_fma(t1, t1, g);
    return g;
}

// exact evaluation of up to 64 queued (hypothesis, ranked match) words, one per lane
__device__ __forceinline__ void cnt_drain(const cnt_queue_t *q, int from, int count, int lane, const float *s_rec,
                                          int *s_cnt, const float *cx1, const float *cy1, const float *cx2, const float *cy2,
                                          float threshold) {
    if (lane < count) {
        const uint32_t en = q[from + lane];
        const int hh = (int)(en >> 16), i = (int)(en & 0xFFFFu);
        ResidualF R;
#pragma unroll
        for (int k = 0; k < 9; k++) R.f[k] = s_rec[hh * kCntRec + k];
        residual_prepare(R);
        const float x2 = cx2[i], y2 = cy2[i];
        const float e = residual_e(R, make_float4(cx1[i], cy1[i], x2, y2), (double)x2, (double)y2);
        if (e <= threshold) atomicAdd(&s_cnt[hh], 1);
    }
}

// append the lanes of mask U (evaluation `idx` of hypothesis hh, idx per lane) to the wave's queue
__device__ __forceinline__ void cnt_push(cnt_queue_t *q, int &qn, unsigned long long U, int hh, int idx, int lane) {
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(U >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)U, 0u));
    if ((U >> lane) & 1ull) q[qn + rank] = ((uint32_t)hh << 16) | (uint32_t)idx;
    qn += __popcll(U);
}

// two matches per lane (ranked positions ix, ix + 1) under one hypothesis: count of the certain inliers and masks of
// the undecided lanes
template <bool PARTIAL>
__device__ __forceinline__ void cnt_eval_pair(const CntRec &R, const v2f X1, const v2f Y1, const v2f X2, const v2f Y2, int ix,
                                              int m, int &cnt, unsigned long long &ua, unsigned long long &ub,
                                              float &ddmin, v2f &acc) {
    v2f dd;
    v2f g = cnt_cheap(R, X1, Y1, X2, Y2, dd);
    ddmin = fminf(ddmin, fminf(dd.x, dd.y));
    // v_cmp straight into a lane mask (llvm::CmpInst predicates: 4 = ordered <, 2 = ordered >)
    unsigned long long ia = __builtin_amdgcn_fcmpf(g.x, R.lo, kFcmpOLT), oa = __builtin_amdgcn_fcmpf(g.x, R.hi, kFcmpOGT);
    unsigned long long ib = __builtin_amdgcn_fcmpf(g.y, R.lo, kFcmpOLT), ob = __builtin_amdgcn_fcmpf(g.y, R.hi, kFcmpOGT);
    if (PARTIAL) {
        const bool in_a = ix < m, in_b = ix + 1 < m;
        const unsigned long long va = __ballot(in_a), vb = __ballot(in_b);
        ia &= va;
        ib &= vb;
        ua = va & ~(ia | oa);
        ub = vb & ~(ib | ob);
        g.x = in_a ? g.x : 0.f;
        g.y = in_b ? g.y : 0.f;
    } else {
        ua = ~(ia | oa);
        ub = ~(ib | ob);
    }
    cnt += __popcll(ia) + __popcll(ib);
    acc += g;   // running sum of the cheap values: ransac_ties_kernel prunes the tie list with it
}

// sum over the 64 lanes, in a fixed order, left in lane 63 (row butterflies, then row_bcast 15 / 31)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    const int z = 0;
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(z, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

// one 256-match sub-block (the lane's ranked matches ix .. ix + 3) of hypothesis hh: certified count into cnt, the rest queued
template <bool PARTIAL>
__device__ __forceinline__ void cnt_sub_block(const CntRec &R, const float4 X1, const float4 Y1, const float4 X2, const float4 Y2,
                                              int ix, int hh, int lane, int m, int &cnt, cnt_queue_t *q, int &qn, v2f &acc,
                                              int *s_unk, int &pot, bool &unk) {
    unsigned long long u0, u1, u2, u3;
    int c = 0;
    float ddmin = INFINITY;
    v2f x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b;
    x1a.x = X1.x; x1a.y = X1.y; x1b.x = X1.z; x1b.y = X1.w;
    y1a.x = Y1.x; y1a.y = Y1.y; y1b.x = Y1.z; y1b.y = Y1.w;
    x2a.x = X2.x; x2a.y = X2.y; x2b.x = X2.z; x2b.y = X2.w;
    y2a.x = Y2.x; y2a.y = Y2.y; y2b.x = Y2.z; y2b.y = Y2.w;
    cnt_eval_pair<PARTIAL>(R, x1a, y1a, x2a, y2a, ix, m, c, u0, u1, ddmin, acc);
    cnt_eval_pair<PARTIAL>(R, x1b, y1b, x2b, y2b, ix + 2, m, c, u2, u3, ddmin, acc);
    if (__builtin_amdgcn_fcmpf(ddmin, kCntTinyDD, kFcmpULT) != 0ull) {   // unordered or <: some dd is zero / denormal (or NaN)
        // nothing of this sub-block is certified: all of it is queued, and the hypothesis' cheap sum means nothing
        if (lane == 0) s_unk[hh] = 1;
        unk = true;   // wave-uniform
        c = 0;
        u0 = __builtin_amdgcn_sicmp(ix, m, 40);   // 40 = signed <
        u1 = __builtin_amdgcn_sicmp(ix + 1, m, 40);
        u2 = __builtin_amdgcn_sicmp(ix + 2, m, 40);
        u3 = __builtin_amdgcn_sicmp(ix + 3, m, 40);
    }
    cnt += c;
    pot += c;   // inliers this sub-block can still turn out to have: the certain ones plus the undecided ones
    if ((u0 | u1 | u2 | u3) != 0ull) {
        pot += __popcll(u0) + __popcll(u1) + __popcll(u2) + __popcll(u3);
        if (u0) cnt_push(q, qn, u0, hh, ix, lane);
        if (u1) cnt_push(q, qn, u1, hh, ix + 1, lane);
        if (u2) cnt_push(q, qn, u2, hh, ix + 2, lane);
        if (u3) cnt_push(q, qn, u3, hh, ix + 3, lane);
    }
}

// exact inlier count of one hypothesis over the ranked coordinate arrays (one wave, lanes over the matches)
// (sum_out: the sum of every e as well, lanes' partial sums in double -- not the reference's sequential order: a value within
// m 2^-53 of it, for a bound)
__device__ __forceinline__ int cnt_exact_ranked(const float *F9, const float *r, int kp_pad, int m, float threshold, int lane,
                                                double *sum_out = nullptr) {
    ResidualF R;
#pragma unroll
    for (int j = 0; j < 9; j++) R.f[j] = F9[j];
    residual_prepare(R);
    int count = 0;
    double esum = 0;
    for (int i0 = 0; i0 < m; i0 += 256) {   // the arrays are padded to a multiple of 256
        float4 c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane;
            c[u] = make_float4(r[i], r[kp_pad + i], r[2 * kp_pad + i], r[3 * kp_pad + i]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float e = residual_e(R, c[u], (double)c[u].z, (double)c[u].w);
            count += __popcll(__ballot(i0 + u * 64 + lane < m && e <= threshold));
            if (sum_out && i0 + u * 64 + lane < m) esum += (double)e;
        }
    }
    if (sum_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
        *sum_out = esum;
    }
    return count;
}

// One workgroup per pair.  Wave w counts pilot hypothesis w exactly (lanes over the matches) and leaves its inlier mask in
// LDS; cbound[pair] = the best of those counts (stored, not accumulated: nothing has to clear it): a first lower bound on the pair's maximum count (any count of any
// hypothesis is a valid bound; a better one only prunes more).  Then the matches are ordered by the number of pilots that
// miss them, most-missed first, original order among equals (a counting sort over 9 keys), and written as four arrays
// rk[pair][0..3][kp_pad] = x1, y1, x2, y2, padded with a real match up to the next multiple of 256.
// cmax[pair] = the largest |coordinate| per frame (the error bands need it).
__global__ __launch_bounds__(kRankThreads) void ransac_rank_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int kp_pad, int hyp, float threshold,
    const float *__restrict__ hypF, int32_t *__restrict__ cbound, float *__restrict__ rk, float *__restrict__ cmax,
    unsigned long long *__restrict__ sfloor) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = min(m_arr[b], kp_stride);
    if (m < min_m) return;
    __shared__ double s_psum[kPilotHyps];   // the pilots' residual sums (lanes' partial sums: for the sum floor, a bound)
    __shared__ unsigned long long s_bits[kPilotHyps][VSLAM_MAX_KP / 64];
    __shared__ int s_hist[kPilotHyps][kPilotHyps + 1], s_off[kPilotHyps][kPilotHyps + 1], s_pcount[kPilotHyps];
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;
    const int npil = min(kPilotHyps, hyp);
    const int G = (m + 63) >> 6;

    if (wave < npil) {
        const int h = (int)(((long long)hyp * wave) / npil);
        ResidualF R;
        const float *src = hypF + ((size_t)b * hyp + h) * 9;
#pragma unroll
        for (int j = 0; j < 9; j++) R.f[j] = src[j];
        residual_prepare(R);
        int count = 0;
        double esum = 0;
        float c1 = 0.f, c2 = 0.f;
        for (int g0 = 0; g0 < G; g0 += 4) {   // four matches per lane and round: their gathers are in flight together
            int2 pr[4];
            float2 a[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pr[u] = PR[min((g0 + u) * 64 + lane, m - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = P1[pr[u].x];
                c[u] = P2[pr[u].y];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float e = residual_e(R, make_float4(a[u].x, a[u].y, c[u].x, c[u].y), (double)c[u].x, (double)c[u].y);
                const unsigned long long in = __ballot((g0 + u) * 64 + lane < m && e <= threshold);
                count += __popcll(in);
                if ((g0 + u) * 64 + lane < m) esum += (double)e;
                if (lane == 0 && g0 + u < G) s_bits[wave][g0 + u] = in;
                c1 = fmaxf(c1, fmaxf(fabsf(a[u].x), fabsf(a[u].y)));
                c2 = fmaxf(c2, fmaxf(fabsf(c[u].x), fabsf(c[u].y)));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
        if (lane == 0) {
            s_pcount[wave] = count;
            s_psum[wave] = esum;
        }
        if (wave == 0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                c1 = fmaxf(c1, __shfl_xor(c1, off, 64));
                c2 = fmaxf(c2, __shfl_xor(c2, off, 64));
            }
            if (lane == 0) {
                cmax[2 * b] = c1;
                cmax[2 * b + 1] = c2;
            }
        }
    }
    __syncthreads();

    // this wave's share of the 64-match groups; key of a match = number of pilots that miss it
    const int gpw = (G + kPilotHyps - 1) / kPilotHyps;
    const int g_lo = min(G, wave * gpw), g_hi = min(G, g_lo + gpw);
    int hist[kPilotHyps + 1];
#pragma unroll
    for (int v = 0; v <= kPilotHyps; v++) hist[v] = 0;
    for (int g = g_lo; g < g_hi; g++) {
        const bool valid = g * 64 + lane < m;
        int key = 0;
        for (int w = 0; w < npil; w++) key += (int)((~s_bits[w][g] >> lane) & 1ull);
#pragma unroll
        for (int v = 0; v <= kPilotHyps; v++) hist[v] += __popcll(__ballot(valid && key == v));
    }
    if (lane == 0) {
#pragma unroll
        for (int v = 0; v <= kPilotHyps; v++) s_hist[wave][v] = hist[v];
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;   // the first word written to cbound[pair] in a call: a plain store, nothing to clear beforehand
        for (int w = 0; w < npil; w++) best = max(best, s_pcount[w]);
        cbound[b] = best;
        if (sfloor) {   // the first sum floor of the pair (see ransac_cand_kernel): the best-counting pilots' largest sum
            double fl = -INFINITY;
            bool any_nan = false;
            for (int w = 0; w < npil; w++)
                if (s_pcount[w] == best) {
                    any_nan = any_nan || !(s_psum[w] == s_psum[w]);
                    if (s_psum[w] > fl) fl = s_psum[w];
                }
            const bool valid = best > 0 && !any_nan && fl > 0 && fl < 0x1p120;
            sfloor[b] = valid ? ((unsigned long long)(uint32_t)best << 32) | (unsigned long long)__float_as_uint((float)(fl * (1.0 - 0x1p-20))) : 0ull;
        }
        int run = 0;
        for (int v = kPilotHyps; v >= 0; v--)
            for (int w = 0; w < kPilotHyps; w++) {
                s_off[w][v] = run;
                run += s_hist[w][v];
            }
    }
    __syncthreads();
    int off[kPilotHyps + 1];
#pragma unroll
    for (int v = 0; v <= kPilotHyps; v++) off[v] = s_off[wave][v];
    float *r = rk + (size_t)b * 4 * kp_pad;
    for (int g = g_lo; g < g_hi; g++) {
        const int i = g * 64 + lane;
        const bool valid = i < m;
        int key = 0;
        for (int w = 0; w < npil; w++) key += (int)((~s_bits[w][g] >> lane) & 1ull);
        int pos = 0;
#pragma unroll
        for (int v = 0; v <= kPilotHyps; v++) {
            const unsigned long long bal = __ballot(valid && key == v);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (key == v) pos = off[v] + rank;
            off[v] += __popcll(bal);
        }
        if (valid) {
            const int2 pr = PR[i];
            const float2 a = P1[pr.x], c = P2[pr.y];
            r[pos] = a.x;
            r[kp_pad + pos] = a.y;
            r[2 * kp_pad + pos] = c.x;
            r[3 * kp_pad + pos] = c.y;
        }
    }
    const int mpad = cnt_pad(m);
    if (m + tid < mpad) {   // at most 255 slots: lanes beyond m in the last sub-block read a real match (and are masked)
        const int2 pr = PR[m - 1];
        const float2 a = P1[pr.x], c = P2[pr.y];
        r[m + tid] = a.x;
        r[kp_pad + m + tid] = a.y;
        r[2 * kp_pad + m + tid] = c.x;
        r[3 * kp_pad + m + tid] = c.y;
    }
}

// pot0[pair][h] = how many of the kScreenMatches most-missed matches hypothesis h can have as inliers (everything the
// cheap evaluation does not certify as an outlier).  Workgroup = 128 hypotheses, a wave walks 32 of them with two of the
// ranked matches per lane.  grid = (ceil(hyp / 128), batch).
__global__ __launch_bounds__(256) void ransac_screen_kernel(
    const float *__restrict__ rk, int kp_pad, const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp,
    float threshold, const float *__restrict__ hypF, const float *__restrict__ cmax, int32_t *__restrict__ pot0) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hbase = blockIdx.x * kScreenHyps;
    const int m = min(m_arr[b], kp_stride);
    if (m < min_m) return;
    __shared__ __align__(16) float s_rec[kScreenHyps * kCntRec];
    if (tid < kScreenHyps) {
        const float *src = hypF + ((size_t)b * hyp + min(hbase + tid, hyp - 1)) * 9;
        float f[9];
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = src[k];
        const CntBand B = cnt_band(f, cmax[2 * b], cmax[2 * b + 1], threshold);
        cnt_store_record(s_rec + tid * kCntRec, f, B);
    }
    const int n0 = min(m, kScreenMatches);
    const float *r = rk + (size_t)b * 4 * kp_pad;
    const float2 x1 = *reinterpret_cast<const float2 *>(r + 2 * lane);
    const float2 y1 = *reinterpret_cast<const float2 *>(r + kp_pad + 2 * lane);
    const float2 x2 = *reinterpret_cast<const float2 *>(r + 2 * kp_pad + 2 * lane);
    const float2 y2 = *reinterpret_cast<const float2 *>(r + 3 * kp_pad + 2 * lane);
    v2f X1, Y1, X2, Y2;
    X1.x = x1.x; X1.y = x1.y; Y1.x = y1.x; Y1.y = y1.y;
    X2.x = x2.x; X2.y = x2.y; Y2.x = y2.x; Y2.y = y2.y;
    const unsigned long long va = __ballot(2 * lane < n0), vb = __ballot(2 * lane + 1 < n0);
    __syncthreads();
    int mine = 0;
    constexpr int kU = 4;   // hypotheses in flight: the record reads and the dependent chain of one hide behind the others
    for (int j = 0; j < 32; j += kU) {
        CntRec R[kU];
        v2f dd[kU], g[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) cnt_load_record(R[u], s_rec, wave * 32 + j + u);
#pragma unroll
        for (int u = 0; u < kU; u++) g[u] = cnt_cheap(R[u], X1, Y1, X2, Y2, dd[u]);
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const unsigned long long oa = __builtin_amdgcn_fcmpf(g[u].x, R[u].hi, kFcmpOGT), ob = __builtin_amdgcn_fcmpf(g[u].y, R[u].hi, kFcmpOGT);
            int p = __popcll(va & ~oa) + __popcll(vb & ~ob);
            // a zero / denormal (or NaN) dd: v_rcp_f32 is not a 1-ulp reciprocal there, nothing is certified
            if (__builtin_amdgcn_fcmpf(fminf(dd[u].x, dd[u].y), kCntTinyDD, kFcmpULT) != 0ull) p = n0;   // unordered or <
            mine = lane == j + u ? p : mine;
        }
    }
    const int h = hbase + wave * 32 + lane;
    if (lane < 32 && h < hyp) pot0[(size_t)b * hyp + h] = mine;
}

// The hypotheses that do best on the screen (largest pot0, then one less, first indices, at most kCandMax) are counted in
// full, exactly: cbound[pair] = max(cbound[pair], those counts).  One workgroup of 8 waves per pair.
__global__ __launch_bounds__(64 * kCandMax) void ransac_cand_kernel(
    const float *__restrict__ rk, int kp_pad, const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp,
    float threshold, const float *__restrict__ hypF, const int32_t *__restrict__ pot0, int by_sum,
    int32_t *__restrict__ cbound, unsigned long long *__restrict__ sfloor) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = min(m_arr[b], kp_stride);
    if (m < min_m) return;
    __shared__ int s_max[kCandMax], s_l0[kCandMax][kCandMax], s_l1[kCandMax][kCandMax], s_n0[kCandMax], s_n1[kCandMax];
    __shared__ int s_cand[kCandMax], s_nc;
    const int32_t *P = pot0 + (size_t)b * hyp;
    int mx = -1;
    for (int i = tid; i < hyp; i += 64 * kCandMax) mx = max(mx, P[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = s_max[0];
    for (int w = 1; w < kCandMax; w++) mx = max(mx, s_max[w]);
    // wave w scans its contiguous share in index order
    const int per = (hyp + kCandMax - 1) / kCandMax;
    const int lo = min(hyp, wave * per), hi = min(hyp, lo + per);
    int n0 = 0, n1 = 0;
    // (by_sum: the wave also notes the first 32 best-scoring hypotheses of its share -- a random sample: a hypothesis' index
    // says nothing about it -- and afterwards moves the one with the largest residual sum over the screen's matches to the
    // front.  Those matches are mostly the pair's outliers, whose residuals dominate a residual sum, so that candidate's sum
    // is likely to be near the largest among the hypotheses that tie the maximum count: what makes the sum floor below bite.)
    __shared__ int s_samp[kCandMax][32];
    int nsamp = 0;
    for (int i0 = lo; i0 < hi && ((by_sum && nsamp < 32) || n0 < kCandMax || n1 < kCandMax); i0 += 64) {
        const int i = i0 + lane;
        const int v = i < hi ? P[i] : -2;
        unsigned long long b0 = __ballot(v == mx), b1 = __ballot(v == mx - 1);
        if (by_sum) {
            const int rs = nsamp + __popcll(b0 & ((1ull << lane) - 1ull));
            if (v == mx && rs < 32) s_samp[wave][rs] = i;
            nsamp = min(32, nsamp + (int)__popcll(b0));
        }
        const int r0 = n0 + __popcll(b0 & ((1ull << lane) - 1ull)), r1 = n1 + __popcll(b1 & ((1ull << lane) - 1ull));
        if (v == mx && r0 < kCandMax) s_l0[wave][r0] = i;
        if (v == mx - 1 && r1 < kCandMax) s_l1[wave][r1] = i;
        n0 = min(kCandMax, n0 + __popcll(b0));
        n1 = min(kCandMax, n1 + __popcll(b1));
    }
    if (by_sum && nsamp > 1) {   // the sample's best by sum over the screen's matches to the front of the wave's list
        // (a ranking aid only: nothing is certified with these values; NaN / inf simply rank oddly)
        const int n128 = min(m, kScreenMatches);
        const float *r = rk + (size_t)b * 4 * kp_pad;
        const float2 x1 = *reinterpret_cast<const float2 *>(r + 2 * lane), y1 = *reinterpret_cast<const float2 *>(r + kp_pad + 2 * lane);
        const float2 x2 = *reinterpret_cast<const float2 *>(r + 2 * kp_pad + 2 * lane), y2 = *reinterpret_cast<const float2 *>(r + 3 * kp_pad + 2 * lane);
        v2f X1, Y1, X2, Y2;
        X1.x = x1.x; X1.y = x1.y; Y1.x = y1.x; Y1.y = y1.y;
        X2.x = x2.x; X2.y = x2.y; Y2.x = y2.x; Y2.y = y2.y;
        float fl[9];   // lane l < nsamp: the F of sample l
        {
            const int hs = s_samp[wave][lane < nsamp ? lane : 0];
            const float *src = hypF + ((size_t)b * hyp + hs) * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) fl[k] = src[k];
        }
        float best_s = -INFINITY;
        int best_i = -1;
        for (int j = 0; j < nsamp; j++) {
            CntRec R;
#pragma unroll
            for (int k = 0; k < 9; k++) R.f[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fl[k]), j));
            v2f dd;
            const v2f g = cnt_cheap(R, X1, Y1, X2, Y2, dd);
            const float part = wave_sum_to_lane63((2 * lane < n128 ? g.x : 0.f) + (2 * lane + 1 < n128 ? g.y : 0.f));
            const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), 63));
            if (tot > best_s) {   // false for NaN
                best_s = tot;
                best_i = s_samp[wave][j];
            }
        }
        if (lane == 0 && best_i >= 0 && n0 > 0) {
            const int old = s_l0[wave][0];
            for (int k = 1; k < n0; k++)
                if (s_l0[wave][k] == best_i) s_l0[wave][k] = old;
            s_l0[wave][0] = best_i;
        }
    }
    if (lane == 0) {
        s_n0[wave] = n0;
        s_n1[wave] = n1;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < kCandMax && n < kCandMax; w++)   // every wave's first entry first (with ssum: its best by sum)
            if (s_n0[w] > 0) s_cand[n++] = s_l0[w][0];
        for (int w = 0; w < kCandMax && n < kCandMax; w++)
            for (int k = 1; k < s_n0[w] && n < kCandMax; k++) s_cand[n++] = s_l0[w][k];
        for (int w = 0; w < kCandMax && n < kCandMax; w++)
            for (int k = 0; k < s_n1[w] && n < kCandMax; k++) s_cand[n++] = s_l1[w][k];
        s_nc = n;
    }
    __syncthreads();
    // The sum floor (the counting kernel's second way of dropping a hypothesis): among hypotheses of equal count the accept
    // rule keeps the one with the LARGER residual sum (src/RansacFilter.cpp:59), so a candidate with exact count c and
    // residual sum s rules out every hypothesis that can at best tie c with a sum certainly below s.
    // sfloor[pair] = c << 32 | bits of a lower bound of s (a positive float: its bits order as the number does), so that
    // a 64-bit maximum is "the higher count, then the larger sum": ransac_rank_kernel stores the best pilot's, this kernel
    // keeps the larger of that and its own candidates'.  0 = no floor.
    __shared__ int s_ccount[kCandMax];
    __shared__ double s_csum[kCandMax];
    if (wave < s_nc) {
        const int h = s_cand[wave];
        double esum = 0;
        const int count = cnt_exact_ranked(hypF + ((size_t)b * hyp + h) * 9, rk + (size_t)b * 4 * kp_pad, kp_pad, m, threshold, lane, &esum);
        if (lane == 0) {
            atomicMax(&cbound[b], count);
            s_ccount[wave] = count;
            s_csum[wave] = esum;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int cb = -1;
        for (int k = 0; k < s_nc; k++) cb = max(cb, s_ccount[k]);
        double fl = -INFINITY;
        bool any_nan = false;
        for (int k = 0; k < s_nc; k++)
            if (s_ccount[k] == cb) {
                any_nan = any_nan || !(s_csum[k] == s_csum[k]);
                if (s_csum[k] > fl) fl = s_csum[k];
            }
        // a NaN sum among the best candidates takes part in the accept rule in ways a floor cannot express: no floor from them
        const bool valid = cb > 0 && !any_nan && fl > 0 && fl < 0x1p120;
        const float flf = (float)(fl * (1.0 - 0x1p-20));   // float rounding stays inside the 2^-20
        const unsigned long long mine = valid ? ((unsigned long long)(uint32_t)cb << 32) | (unsigned long long)__float_as_uint(flf) : 0ull;
        const unsigned long long pilots = sfloor[b];   // ransac_rank_kernel's: the higher count, then the larger sum
        sfloor[b] = mine > pilots ? mine : pilots;
    }
}

// grid = (ceil(hyp / 128), batch), block = 512; dynamic LDS = (LDS ? 16 B x cnt_pad(kp_stride) : 0) + 8 queues.
// LDS = false (more than kCntLdsMatches slots per pair): the ranked coordinates are read from memory instead.
template <bool LDS>
__global__ __launch_bounds__(64 * kCntWaves) __attribute__((amdgpu_waves_per_eu(6, 6))) void ransac_count_kernel(
    const float *__restrict__ rk, int kp_pad, const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, float threshold,
    const float *__restrict__ hypF, const int32_t *__restrict__ pot0, const float *__restrict__ cmax,
    int32_t *__restrict__ hyp_count, float *__restrict__ hyp_sum, float *__restrict__ approx, int32_t *__restrict__ cbound,
    const unsigned long long *__restrict__ sfloor) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and the compiler may know it
    const int hbase = blockIdx.x * kCntHyps;
    const int nh = min(kCntHyps, hyp - hbase);
    const int m = min(m_arr[b], kp_stride);
    if (m < min_m) return;   // uniform per workgroup

    __shared__ __align__(16) float s_rec[kCntHyps * kCntRec];
    __shared__ int s_cnt[kCntHyps];
    __shared__ float s_part[kCntHyps];   // sum of the cheap values
    __shared__ int s_unk[kCntHyps];      // hypothesis whose cheap sum is not certified
    __shared__ int s_state[kCntHyps];    // 0 = ruled out by the screen, 1 = counted in full, 2 = abandoned on the way
    __shared__ int s_list[kCntHyps];     // survivors of the screen: [64 w, 64 w + s_nlist[w]) found by wave w
    constexpr int kListWaves = kCntHyps / 64;
    static_assert(kCntHyps % 64 == 0 && kListWaves <= kCntWaves, "whole waves find the survivors");
    __shared__ int s_nlist[kListWaves], s_next, s_bound;
    extern __shared__ __align__(16) uint32_t s_dyn[];

    const int n0 = min(m, kScreenMatches);
    const int bound0 = cbound[b];   // a count some hypothesis of this pair is known to reach (never above the true maximum)
    if (tid < kCntHyps) {           // whole waves
        const bool alive = tid < nh && pot0[(size_t)b * hyp + hbase + tid] + (m - n0) >= bound0;
        s_cnt[tid] = 0;
        s_unk[tid] = 0;
        s_part[tid] = 0.f;
        s_state[tid] = alive ? 1 : 0;
        const unsigned long long bal = __ballot(alive);
        if (alive) s_list[wave * 64 + __popcll(bal & ((1ull << lane) - 1ull))] = tid;
        if (lane == 0) s_nlist[wave] = __popcll(bal);
    }
    if (tid == 0) {
        s_next = 0;
        s_bound = bound0;
    }
    __syncthreads();
    int n_alive = 0;
#pragma unroll
    for (int w = 0; w < kListWaves; w++) n_alive += s_nlist[w];
    const size_t out = (size_t)b * hyp + hbase + tid;
    if (n_alive == 0) {   // the usual case on a pair whose maximum only a few hypotheses reach
        if (tid < nh) {
            hyp_count[out] = -1;
            hyp_sum[out] = __int_as_float(0x7FC00000);
            reinterpret_cast<float2 *>(approx)[out] = make_float2(0.f, INFINITY);
        }
        return;
    }

    double my_beta = 0;     // thread t < 128 keeps hypothesis t's beta for the bound on its cheap sum
    bool my_ok = false;
    if (tid < kCntHyps && s_state[tid]) {   // the survivors' records: F, lo, hi
        const float *src = hypF + ((size_t)b * hyp + hbase + tid) * 9;
        float f[9];
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = src[k];
        const CntBand B = cnt_band(f, cmax[2 * b], cmax[2 * b + 1], threshold);
        my_beta = B.beta;
        my_ok = B.ok;
        cnt_store_record(s_rec + tid * kCntRec, f, B);
    }
    const int mpad = cnt_pad(m);
    const float *rg = rk + (size_t)b * 4 * kp_pad;
    float *s_co = reinterpret_cast<float *>(s_dyn);
    if (LDS) {
        for (int k = tid * 4; k < mpad; k += 64 * kCntWaves * 4) {
#pragma unroll
            for (int a = 0; a < 4; a++)
                *reinterpret_cast<float4 *>(s_co + a * mpad + k) = *reinterpret_cast<const float4 *>(rg + (size_t)a * kp_pad + k);
        }
    }
    __syncthreads();
    const int cstride = LDS ? mpad : kp_pad;
    const float *cx1 = LDS ? s_co : rg;
    const float *cy1 = cx1 + cstride, *cx2 = cx1 + 2 * cstride, *cy2 = cx1 + 3 * cstride;

    {
        cnt_queue_t *q = (cnt_queue_t *)(s_dyn + (LDS ? 4 * mpad : 0) + wave * kCntQueue);
        int qn = 0;
        int bound = bound0;
        // the sum floor of ransac_cand_kernel: (the count it belongs to, a lower bound of that hypothesis' residual sum); -1: none
        const unsigned long long fkey = sfloor ? sfloor[b] : 0ull;
        const int floor_count = fkey != 0ull ? (int)(fkey >> 32) : -1;
        const float floor_sum = __uint_as_float((uint32_t)fkey);
        const int nsub = mpad >> 8;
        const bool part = (m & 255) != 0;
        // the survivors are handed out one at a time: what a hypothesis costs (256 evaluations or all of them) is not known beforehand
        int k = 0;
        if (lane == 0) k = atomicAdd(&s_next, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        while (k < n_alive) {
            int knext = 0;
            if (lane == 0) knext = atomicAdd(&s_next, 1);   // consumed at the bottom: the round trip hides behind the work
            int hh;
            {
                int kk = k, w = 0;
                while (w + 1 < kListWaves && kk >= s_nlist[w]) kk -= s_nlist[w++];
                hh = s_list[w * 64 + kk];
            }
            bound = max(bound, *(__attribute__((address_space(3))) const volatile int *)&s_bound);
            CntRec R;
            cnt_load_record(R, s_rec, hh);
            int cnt = 0, pot = 0, seen = 0;
            bool dropped = false, unk = false;
            v2f acc;
            acc.x = 0.f;
            acc.y = 0.f;
            for (int s = 0; s < nsub; s++) {
                // the lane's four matches of this sub-block: straight from LDS (six waves per SIMD hide the round trip;
                // fetching a sub-block ahead cost 16 registers and a copy per value)
                const int ix = s * 256 + 4 * lane;
                const float4 X1 = *reinterpret_cast<const float4 *>(cx1 + ix), Y1 = *reinterpret_cast<const float4 *>(cy1 + ix);
                const float4 X2 = *reinterpret_cast<const float4 *>(cx2 + ix), Y2 = *reinterpret_cast<const float4 *>(cy2 + ix);
                if (s == nsub - 1 && part) {
                    cnt_sub_block<true>(R, X1, Y1, X2, Y2, ix, hh, lane, m, cnt, q, qn, acc, s_unk, pot, unk);
                    seen += m & 255;
                } else {
                    cnt_sub_block<false>(R, X1, Y1, X2, Y2, ix, hh, lane, m, cnt, q, qn, acc, s_unk, pot, unk);
                    seen += 256;
                }
                while (qn >= 64) {   // a sub-block adds at most 256 words to the 63 left over: kCntQueue holds them
                    cnt_drain(q, qn - 64, 64, lane, s_rec, s_cnt, cx1, cy1, cx2, cy2, threshold);
                    qn -= 64;
                }
                // Bail-out.  Even if every match not looked at yet were an inlier, the hypothesis would stay below a
                // count some hypothesis of this pair is already known to reach: it cannot be a maximum-count hypothesis,
                // which is all the accept rule looks at.  `bound` never exceeds the true maximum, so every hypothesis
                // that reaches the maximum is counted in full.
                if (pot + (m - seen) < bound) {
                    dropped = true;
                    break;
                }
                // The sum rule.  The hypothesis can at best TIE the count the floor belongs to (every match not looked at
                // yet would have to be an inlier, i.e. add at most `threshold` each to its residual sum), and among equal
                // counts the accept rule keeps the larger sum: if an upper bound of its sum -- the cheap values so far,
                // their certified error (the bound ransac_ties uses, over `seen` evaluations), threshold x the rest --
                // lies below the floor by more than float rounding can close, it is neither the winner nor a tie.
                // On data with one dominant motion a third to a half of the hypotheses share the maximum count and used to
                // be counted in full for that; their sums are far from the best one's (outlier residuals dominate them).
                if (seen < m && bound == floor_count && pot + (m - seen) == bound && !unk && R.beta >= 0.f) {
                    // in float, every rounding covered by the 1e-4 factors (the terms are all positive)
                    const float S = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(acc.x + acc.y)), 63));
                    const float ns = (float)seen;
                    const float err = (2.02f * R.beta * sqrtf(ns * S) + ns * R.beta * R.beta + 0x1p-18f * S) * 1.0001f;
                    const float upper = ((S + err) + threshold * (float)(m - seen)) * 1.0001f;
                    if (upper < floor_sum * 0.9999f) {   // false for NaN
                        dropped = true;
                        break;
                    }
                }
            }
            if (dropped) {
                if (lane == 0) s_state[hh] = 2;
            } else {
                const float part_sum = wave_sum_to_lane63(acc.x + acc.y);
                if (lane == 63) s_part[hh] = part_sum;
                if (lane == 0) {
                    if (cnt) atomicAdd(&s_cnt[hh], cnt);
                    if (cnt > bound) {   // the certain inliers alone are a count this hypothesis verifiably reaches
                        atomicMax(&s_bound, cnt);
                        atomicMax(&cbound[b], cnt);   // fire and forget: workgroups of this pair that start later begin with it
                    }
                }
                bound = max(bound, cnt);
            }
            k = __builtin_amdgcn_readfirstlane(knext);
        }
        if (qn > 0) cnt_drain(q, 0, qn, lane, s_rec, s_cnt, cx1, cy1, cx2, cy2, threshold);
    }
    __syncthreads();
    if (tid < nh) {
        const int st = s_state[tid];
        hyp_count[out] = st == 1 ? s_cnt[tid] : -1;     // ransac_ties keeps the maximal ones
        hyp_sum[out] = __int_as_float(0x7FC00000);      // defined by ransac_ties / tiesum where it matters
        // The cheap values' sum S~ and a bound on |S~ - (exact double sum of the e)|: per evaluation
        // |g - e| <= 12 u max(g, e) + beta (2 sqrt(g) + beta)  (the derivation above), summed with Cauchy-Schwarz
        // (sum sqrt(g_i) <= sqrt(M sum g_i)), plus 2^-20 S~ for the float additions that formed S~.
        const float S = s_part[tid];
        float err = INFINITY;
        if (st == 1 && my_ok && !s_unk[tid]) {
            const double Sd = (double)S;
            err = (float)((2.02 * my_beta * sqrt((double)m * Sd) + (double)m * my_beta * my_beta + 0x1p-18 * Sd) * (1.0 + 0x1p-20));
        }
        reinterpret_cast<float2 *>(approx)[out] = make_float2(S, err);
    }
}

// C* = the pair's largest inlier count, and which of the hypotheses that reach it need their exact residual sum.
// The accept rule keeps, among the hypotheses with count C*, the one with the largest float sum (first index among equal
// sums; the NaN cases are spelled out at ransac_select_kernel).  The counting kernel left a cheap sum S~ and a bound err
// with |S~ - exact sum| <= err for every hypothesis, so a tied hypothesis whose S~ + err lies below the best S~ - err (by
// more than float rounding can close: factor 1 - 2^-20) cannot win and cannot tie after rounding: its hyp_sum becomes
// -inf (ordered below every real sum, never NaN).  The others — normally a handful — go on the list for
// ransac_tiesum_kernel.  Hypotheses with an uncertified cheap sum carry err = inf and always stay.
// tie_n[2b] = list length, tie_n[2b+1] = C*.  One workgroup per pair.
constexpr int kTieThreads = 256;
__device__ __forceinline__ void ransac_ties_body(const int b, const int32_t *__restrict__ m_arr, int min_m, int hyp,
                                                 int32_t *__restrict__ hyp_count,
                                                 const float *__restrict__ approx, float *__restrict__ hyp_sum,
                                                 int32_t *__restrict__ tie_idx, int32_t *__restrict__ tie_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_w[kTieThreads / 64];
    __shared__ float s_f[kTieThreads / 64];
    __shared__ int s_base;
    if (m_arr[b] < min_m) {
        if (tid == 0) {
            tie_n[2 * b] = 0;
            tie_n[2 * b + 1] = 0;
        }
        return;
    }
    int32_t *C = hyp_count + (size_t)b * hyp;
    const float2 *A = reinterpret_cast<const float2 *>(approx) + (size_t)b * hyp;
    float *Sm = hyp_sum + (size_t)b * hyp;
    int32_t *TI = tie_idx + (size_t)b * hyp;
    int mx = 0;
    for (int i = tid; i < hyp; i += kTieThreads) mx = max(mx, C[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_w[wave] = mx;
    if (tid == 0) s_base = 0;
    __syncthreads();
    mx = s_w[0];
    for (int w = 1; w < kTieThreads / 64; w++) mx = max(mx, s_w[w]);
    // best certified lower bound among the tied hypotheses
    float L = -INFINITY;
    for (int i = tid; i < hyp; i += kTieThreads)
        if (C[i] == mx) {
            const float2 a = A[i];
            const float lowb = a.x - a.y;
            if (lowb > L) L = lowb;   // false for NaN
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) L = fmaxf(L, __shfl_xor(L, off, 64));
    if (lane == 0) s_f[wave] = L;
    __syncthreads();
    L = s_f[0];
    for (int w = 1; w < kTieThreads / 64; w++) L = fmaxf(L, s_f[w]);
    const float cut = L > 0.f ? L * (1.f - 0x1p-20f) : -INFINITY;   // sums are >= 0: nothing is pruned against a bound <= 0
    __syncthreads();
    for (int i0 = 0; i0 < hyp; i0 += kTieThreads) {
        const int i = i0 + tid;
        bool keep = false;
        // which hypotheses below the maximum were counted in full depends on when the counting kernel learned its
        // bounds: the array is made canonical (and says so) — the maximum for those that reach it, -1 for the rest
        if (i < hyp && C[i] != mx) C[i] = -1;
        if (i < hyp && C[i] == mx) {
            const float2 a = A[i];
            keep = !(a.x + a.y < cut);   // NaN or inf bounds stay
            if (!keep) Sm[i] = -INFINITY;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_w[w];
        if (keep) TI[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kTieThreads / 64; w++) tot += s_w[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        tie_n[2 * b] = s_base;
        tie_n[2 * b + 1] = mx;
    }
}
__global__ __launch_bounds__(kTieThreads) void ransac_ties_kernel(const int32_t *__restrict__ m_arr, int min_m, int hyp,
                                                                  int32_t *__restrict__ hyp_count,
                                                                  const float *__restrict__ approx, float *__restrict__ hyp_sum,
                                                                  int32_t *__restrict__ tie_idx, int32_t *__restrict__ tie_n) {
    ransac_ties_body(blockIdx.x, m_arr, min_m, hyp, hyp_count, approx, hyp_sum, tie_idx, tie_n);
}

// The exact residual sum (sequential double accumulation in match order, then one rounding to float: the value
// RansacFilter.cpp:138 returns) for the hypotheses in the tie list.  grid = (kTieGrid, batch), one wave each, looping over
// the list (normally one entry: a larger grid of workgroups that find nothing to do costs more than the work itself).
//   few ties : a wave per tied hypothesis; the lanes evaluate all matches (64 at a time), park every e in LDS, and the
//              sum then walks them in order (every lane computes the same total from broadcast reads);
//   many ties: a lane per tied hypothesis walking all matches (the shape of ransac_score_kernel), 64 per wave.
constexpr int kTieLaneMode = 192;
constexpr int kTieGrid = 4;
// WAVE_ONLY: the caller is one wave of a larger workgroup (ransac_finish_kernel): `slot` of `nslots` is that wave, and the
// LDS hand-overs below are inside the wave, where program order is enough (no workgroup barrier: the waves run different
// numbers of trips).
template <bool WAVE_ONLY>
__device__ __forceinline__ void ransac_tiesum_body(
    const int b, const int slot, const int nslots, const int lane, float *s_e, float4 *sc,
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, const float *__restrict__ hypF,
    const int32_t *__restrict__ tie_idx, const int32_t *__restrict__ tie_n, float *__restrict__ hyp_sum) {
    auto sync = [&]() {
        if (WAVE_ONLY) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    const int m = min(m_arr[b], kp_stride);
    if (m < min_m) return;
    const int T = tie_n[2 * b];
    const int32_t *TI = tie_idx + (size_t)b * hyp;
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;

    if (T >= kTieLaneMode) {
        for (int blk = slot; blk * 64 < T; blk += nslots) {
            const int k = blk * 64 + lane;
            const int h = TI[min(k, T - 1)];
            ResidualF R;
            const float *src = hypF + ((size_t)b * hyp + h) * 9;
#pragma unroll
            for (int j = 0; j < 9; j++) R.f[j] = src[j];
            residual_prepare(R);
            double total = 0;
            for (int i0 = 0; i0 < m; i0 += 64) {   // 64 matches staged by the wave, then walked in order as broadcasts
                const int2 pr = PR[min(i0 + lane, m - 1)];
                const float2 a = P1[pr.x], c = P2[pr.y];
                sync();
                sc[lane] = make_float4(a.x, a.y, c.x, c.y);
                sync();
                const int cnt = min(64, m - i0);
                for (int t = 0; t < cnt; t++) {
                    const float4 v = sc[t];
                    total += (double)residual_e(R, v, (double)v.z, (double)v.w);
                }
            }
            if (k < T) hyp_sum[(size_t)b * hyp + h] = (float)total;
        }
        return;
    }
    // s_e: kp_stride floats: every e of the hypothesis, then summed in order
    for (int k = slot; k < T; k += nslots) {
        const int h = TI[k];
        ResidualF R;
        const float *src = hypF + ((size_t)b * hyp + h) * 9;
#pragma unroll
        for (int j = 0; j < 9; j++) R.f[j] = src[j];
        residual_prepare(R);
        sync();   // one wave: the previous hypothesis' reads are done
        // every e first, four matches per lane and round so that their gathers are in flight together ...
        for (int i0 = 0; i0 < m; i0 += 256) {
            int2 pr[4];
            float2 a[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pr[u] = PR[min(i0 + u * 64 + lane, m - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = P1[pr[u].x];
                c[u] = P2[pr[u].y];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 64 + lane;
                const float e = residual_e(R, make_float4(a[u].x, a[u].y, c[u].x, c[u].y), (double)c[u].x, (double)c[u].y);
                if (i < m) s_e[i] = e;
            }
        }
        sync();
        // ... then the sum, in match order (every lane walks the same broadcast reads and computes the same total);
        // sixteen values are fetched ahead of the dependent additions
        double total = 0;
        int t = 0;
        for (; t + 16 <= m; t += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(s_e + t + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                total += (double)v[u].x;
                total += (double)v[u].y;
                total += (double)v[u].z;
                total += (double)v[u].w;
            }
        }
        for (; t < m; t++) total += (double)s_e[t];
        if (lane == 0) hyp_sum[(size_t)b * hyp + h] = (float)total;
    }
}

__global__ __launch_bounds__(64) void ransac_tiesum_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, const float *__restrict__ hypF,
    const int32_t *__restrict__ tie_idx, const int32_t *__restrict__ tie_n, float *__restrict__ hyp_sum) {
    extern __shared__ __align__(16) float s_e_dyn[];
    __shared__ float4 sc[64];
    ransac_tiesum_body<false>(blockIdx.y, blockIdx.x, gridDim.x, threadIdx.x, s_e_dyn, sc, xy1, xy2, pairs, m_arr, min_m, kp_stride,
                              hyp, hypF, tie_idx, tie_n, hyp_sum);
}

// ------------------------------------------------------------------------------------------
// find_fundamental's accept rule + winner mask + inlier filter
// ------------------------------------------------------------------------------------------
constexpr int kSelThreads = 256;

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
// monotone map float -> u32 (for non-NaN inputs), +0 and -0 collapse
__device__ __forceinline__ uint32_t float_order(float f) {
    if (f == 0.f) f = 0.f;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The reference scans hypotheses in order and accepts when count > best || (count == best &&
// sum > best_sum), starting from (0, 0.0f) (RansacFilter.cpp:44-45,59).  Closed form used here:
//   C* = max count, i0 = first index with count C*.
//   C* > 0 and sum[i0] is NaN      -> winner i0 (a NaN best_sum is never beaten at equal count)
//   otherwise                      -> first index among {count == C*, sum not NaN} with maximal sum
//                                     (for C* == 0 only if that sum > 0.0f, else nothing accepted)
__device__ __forceinline__ void ransac_select_body(
    const int b, const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, float threshold,
    const float *__restrict__ hypF, const int32_t *__restrict__ hyp_count,
    const float *__restrict__ hyp_sum, float *__restrict__ F_out, uint8_t *__restrict__ mask,
    int32_t *__restrict__ best, int32_t *__restrict__ matches) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = m_arr[b];
    __shared__ unsigned long long s_key[kSelThreads / 64];
    __shared__ int s_wave_cnt[kSelThreads / 64];
    __shared__ int s_base;
    uint8_t *MK = mask + (size_t)b * kp_stride;
    int32_t *BO = best + (size_t)b * 4;

    int winner = -1, win_count = 0;
    float win_sum = 0.f;
    if (m >= min_m) {
        const int32_t *C = hyp_count + (size_t)b * hyp;
        const float *Sm = hyp_sum + (size_t)b * hyp;
        unsigned long long k = 0;
        for (int i = tid; i < hyp; i += kSelThreads)   // -1 = "below the maximum" (the counting path's canonical form): never the key
            k = max(k, ((unsigned long long)(uint32_t)max(C[i], 0) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i));
        k = wave_max_u64(k);
        if (lane == 0) s_key[wave] = k;
        __syncthreads();
        k = s_key[0];
        for (int w = 1; w < kSelThreads / 64; w++) k = max(k, s_key[w]);
        __syncthreads();
        const int cstar = (int)(k >> 32);
        const int i0 = (int)(0xFFFFFFFFu - (uint32_t)k);
        const float s0 = Sm[i0];
        if (cstar > 0 && s0 != s0) {
            winner = i0;
            win_sum = s0;
        } else {
            unsigned long long k2 = 0;   // 0 == "no candidate" (float_order never returns 0 for finite/inf)
            for (int i = tid; i < hyp; i += kSelThreads) {
                const float s = Sm[i];
                if (C[i] == cstar && s == s)
                    k2 = max(k2, ((unsigned long long)float_order(s) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i));
            }
            k2 = wave_max_u64(k2);
            if (lane == 0) s_key[wave] = k2;
            __syncthreads();
            k2 = s_key[0];
            for (int w = 1; w < kSelThreads / 64; w++) k2 = max(k2, s_key[w]);
            __syncthreads();
            if (k2 != 0) {
                const int iw = (int)(0xFFFFFFFFu - (uint32_t)k2);
                const float sw = Sm[iw];
                if (cstar > 0 || sw > 0.0f) {
                    winner = iw;
                    win_sum = sw;
                }
            }
        }
        win_count = winner >= 0 ? cstar : 0;
    }

    if (tid == 0) s_base = 0;
    __syncthreads();

    ResidualF R;
    if (winner >= 0) {
        const float *src = hypF + ((size_t)b * hyp + winner) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) R.f[k] = src[k];
        residual_prepare(R);
        if (tid < 9) F_out[(size_t)b * 9 + tid] = R.f[tid];   // temp_F.copyTo(fundamental), :63
    }
    const float2 *P1 = reinterpret_cast<const float2 *>(xy1) + (size_t)b * kp_stride;
    const float2 *P2 = reinterpret_cast<const float2 *>(xy2) + (size_t)b * kp_stride;
    const int2 *PR = reinterpret_cast<const int2 *>(pairs) + (size_t)b * kp_stride;
    int2 *MO = reinterpret_cast<int2 *>(matches) + (size_t)b * kp_stride;

    // winner's mask (inliers.swap, :64) and the ordered inlier filter of Frame.cpp:98-102
    for (int i0 = 0; i0 < m; i0 += kSelThreads) {
        const int i = i0 + tid;
        bool in = false;
        int2 pr = make_int2(0, 0);
        if (i < m) {
            pr = PR[i];
            if (winner >= 0) {
                const float2 a = P1[pr.x], c = P2[pr.y];
                in = residual_e(R, make_float4(a.x, a.y, c.x, c.y), (double)c.x, (double)c.y) <= threshold;
            }
            MK[i] = in ? 1 : 0;
        }
        const unsigned long long bal = __ballot(in);
        const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave_cnt[w];
        if (in) MO[off + in_wave] = pr;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kSelThreads / 64; w++) tot += s_wave_cnt[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        BO[0] = winner;
        BO[1] = win_count;
        BO[2] = __float_as_int(win_sum);
        BO[3] = s_base;
    }
}
__global__ __launch_bounds__(kSelThreads) void ransac_select_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, float threshold,
    const float *__restrict__ hypF, const int32_t *__restrict__ hyp_count,
    const float *__restrict__ hyp_sum, float *__restrict__ F_out, uint8_t *__restrict__ mask,
    int32_t *__restrict__ best, int32_t *__restrict__ matches) {
    ransac_select_body(blockIdx.x, xy1, xy2, pairs, m_arr, min_m, kp_stride, hyp, threshold, hypF, hyp_count, hyp_sum, F_out, mask,
                       best, matches);
}

// The three closing stages of the counting path in one launch, one workgroup per pair: which maximum-count hypotheses
// need their exact sum (ransac_ties_kernel), those sums (ransac_tiesum_kernel: the workgroup's four waves are its four
// slots), the accept rule and the winner's mask (ransac_select_kernel).  Each stage reads what the one before left in
// memory; they are separated by workgroup barriers, which order a workgroup's own global writes.  Saves two dependent
// launches per step (each mostly dispatch and drain at one workgroup per pair).
static_assert(kTieThreads == kSelThreads && kTieThreads == 64 * kTieGrid, "one shape for the fused closing kernel");
__global__ __launch_bounds__(kSelThreads) void ransac_finish_kernel(
    const float *__restrict__ xy1, const float *__restrict__ xy2, const int32_t *__restrict__ pairs,
    const int32_t *__restrict__ m_arr, int min_m, int kp_stride, int hyp, float threshold,
    const float *__restrict__ hypF, int32_t *__restrict__ hyp_count, const float *__restrict__ approx,
    float *__restrict__ hyp_sum, int32_t *__restrict__ tie_idx, int32_t *__restrict__ tie_n, float *__restrict__ F_out,
    uint8_t *__restrict__ mask, int32_t *__restrict__ best, int32_t *__restrict__ matches) {
    extern __shared__ __align__(16) float s_e_all[];   // kTieGrid x kp_stride floats
    __shared__ float4 sc_all[kTieGrid][64];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ransac_ties_body(b, m_arr, min_m, hyp, hyp_count, approx, hyp_sum, tie_idx, tie_n);
    __syncthreads();
    ransac_tiesum_body<true>(b, wave, kTieGrid, lane, s_e_all + (size_t)wave * kp_stride, sc_all[wave], xy1, xy2, pairs, m_arr, min_m,
                             kp_stride, hyp, hypF, tie_idx, tie_n, hyp_sum);
    __syncthreads();
    ransac_select_body(b, xy1, xy2, pairs, m_arr, min_m, kp_stride, hyp, threshold, hypF, hyp_count, hyp_sum, F_out, mask, best,
                       matches);
}

}  // namespace

size_t vs_ransac_raw_words(int hyp) { return (size_t)vs_mt_blocks(hyp) * kMtN; }

int vs_launch_ransac_mt(vslam_ctx *ctx, const uint32_t *seeds, int batch, int hyp, uint32_t *raw) {
    VS_REQUIRE(ctx, seeds && raw, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && hyp > 0, VSLAM_ERR_INVALID);
    VsProfScope ps(ctx, "ransac_mt_kernel");
    ransac_mt_kernel<<<batch, kSetThreads, 0, ctx->stream>>>(seeds, vs_mt_blocks(hyp), raw);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_ransac_map(vslam_ctx *ctx, const int32_t *m, int batch, int hyp, const uint32_t *raw, int32_t *sets,
                         uint32_t *draws) {
    VS_REQUIRE(ctx, m && raw && sets && draws, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && hyp > 0, VSLAM_ERR_INVALID);
    int32_t *flag = nullptr;
    int rc = vs_device_errflag(ctx, &flag);
    if (rc) return rc;
    VsProfScope ps(ctx, "ransac_sets_kernel");
    if (ctx->ransac_min_items == VSLAM_SET_SIZE)
        ransac_map_kernel<true><<<batch, kMapThreads, 0, ctx->stream>>>(m, hyp, vs_mt_blocks(hyp), VSLAM_SET_SIZE, raw, sets, draws, flag);
    else
        ransac_map_kernel<false><<<batch, kMapThreads, 0, ctx->stream>>>(m, hyp, vs_mt_blocks(hyp), ctx->ransac_min_items, raw, sets,
                                                                         draws, flag);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_ransac_sets(vslam_ctx *ctx, const uint32_t *seeds, const int32_t *m, int batch, int hyp,
                          int32_t *sets, uint32_t *draws) {
    VS_REQUIRE(ctx, seeds && m && sets && draws, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && hyp > 0, VSLAM_ERR_INVALID);
    uint32_t *raw = nullptr;
    int rc = vs_arena_get(ctx, "mf.raw", sizeof(uint32_t) * vs_ransac_raw_words(hyp) * (size_t)batch, (void **)&raw);
    if (rc) return rc;
    ctx->raw_seeds = nullptr;   // the buffer is being rewritten: whatever was produced ahead of time is gone
    if ((rc = vs_launch_ransac_mt(ctx, seeds, batch, hyp, raw))) return rc;
    return vs_launch_ransac_map(ctx, m, batch, hyp, raw, sets, draws);
}

int vs_launch_ransac_solve(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                           const int32_t *m, const int32_t *sets, int batch, int kp_stride, int hyp,
                           float *hypF) {
    VS_REQUIRE(ctx, xy1 && xy2 && pairs && m && sets && hypF, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0 && hyp > 0, VSLAM_ERR_INVALID);
    dim3 grid(vs_div_up(hyp, kSolveThreads), batch);
    if (ctx->ransac_solver == 1) {   // opt-in, not bit-exact (VSLAM_OPT_RANSAC_SOLVER)
        float *cond = nullptr;
        int rc = vs_arena_get(ctx, "ransac.cond", sizeof(float) * 6 * (size_t)batch, (void **)&cond);
        if (rc) return rc;
        VsProfScope ps(ctx, "ransac_solve_gram_kernel");
        ransac_condition_kernel<<<batch, 256, 0, ctx->stream>>>(xy1, xy2, pairs, m, kp_stride, cond);
        ransac_solve_gram_kernel<<<grid, kSolveThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, sets, cond, kp_stride, hyp, hypF);
        VS_HIP(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    // VSLAM_RANSAC_SOLVE_SPLIT: 0 one kernel (rounds 2-4); 4 / 5: sweeps + null-space row at 4 / 5 waves per SIMD, then
    // ransac_close_kernel (the 3 x 3 SVD and U diag Vt, a launch of its own at full occupancy).  Same bits either way.
    VsProfScope ps(ctx, "ransac_solve_kernel");
#ifndef VSLAM_EXPERIMENTS
    ransac_solve_kernel<false, 4><<<grid, kSolveThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, ctx->ransac_min_matches, sets, kp_stride, hyp, hypF);
#else
    const int split = ctx->solve_split;
    if (split == 0) {
        ransac_solve_kernel<false, 4><<<grid, kSolveThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, ctx->ransac_min_matches, sets, kp_stride, hyp, hypF);
    } else {
        if (split == 5)
            ransac_solve_kernel<true, 5><<<grid, kSolveThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, ctx->ransac_min_matches, sets, kp_stride, hyp, hypF);
        else
            ransac_solve_kernel<true, 4><<<grid, kSolveThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, ctx->ransac_min_matches, sets, kp_stride, hyp, hypF);
        ransac_close_kernel<<<dim3(vs_div_up(hyp, kCloseThreads), batch), kCloseThreads, 0, ctx->stream>>>(m, ctx->ransac_min_matches, hyp, hypF);
    }
#endif
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_ransac_evaluate(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                              const int32_t *m, const float *hypF, int batch, int kp_stride, int hyp,
                              float threshold, float *F, uint8_t *mask, int32_t *best, int32_t *matches,
                              int32_t *hyp_count, float *hyp_sum) {
    VS_REQUIRE(ctx, xy1 && xy2 && pairs && m && hypF && F && mask && best && matches, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, hyp_count && hyp_sum, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0 && hyp > 0, VSLAM_ERR_INVALID);
    const int min_m = ctx->ransac_min_matches;   // items with fewer matches get no model (8 unless the caller brought its own F)
    if (ctx->ransac_all_sums) {
        VsProfScope ps(ctx, "ransac_score_kernel");
        dim3 grid(vs_div_up(hyp, kScoreThreads), batch);
        ransac_score_kernel<<<grid, kScoreThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, min_m, kp_stride, hyp,
                                                                     threshold, hypF, hyp_count, hyp_sum);
    } else {
        VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
        int32_t *tie_idx = nullptr, *tie_n = nullptr;
        float *approx = nullptr;
        int rc;
        if ((rc = vs_arena_get(ctx, "ransac.approx", sizeof(float) * 2 * (size_t)batch * hyp, (void **)&approx))) return rc;
        if ((rc = vs_arena_get(ctx, "ransac.tie_idx", sizeof(int32_t) * (size_t)batch * hyp, (void **)&tie_idx))) return rc;
        if ((rc = vs_arena_get(ctx, "ransac.tie_n", sizeof(int32_t) * 2 * (size_t)batch, (void **)&tie_n))) return rc;
        int32_t *cbound = nullptr, *pot0 = nullptr;
        float *rk = nullptr, *cmax = nullptr;
        const int kp_pad = cnt_pad(kp_stride);
        if ((rc = vs_arena_get(ctx, "ransac.cbound", sizeof(int32_t) * (size_t)batch, (void **)&cbound))) return rc;
        unsigned long long *sfloor = nullptr;
        if ((rc = vs_arena_get(ctx, "ransac.sfloor", sizeof(unsigned long long) * (size_t)batch, (void **)&sfloor))) return rc;
        static const bool no_sum_rule = VS_EXPERIMENT_ENV("VSLAM_RANSAC_NO_SUM_RULE") != nullptr;   // A/B timing: bail out on counts only
        const unsigned long long *use_floor = no_sum_rule ? nullptr : sfloor;
        if ((rc = vs_arena_get(ctx, "ransac.pot0", sizeof(int32_t) * (size_t)batch * hyp, (void **)&pot0))) return rc;
        if ((rc = vs_arena_get(ctx, "ransac.rk", sizeof(float) * 4 * (size_t)kp_pad * batch, (void **)&rk))) return rc;
        if ((rc = vs_arena_get(ctx, "ransac.cmax", sizeof(float) * 2 * (size_t)batch, (void **)&cmax))) return rc;
        {
            VsProfScope ps(ctx, "ransac_rank_kernel");
            ransac_rank_kernel<<<batch, kRankThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, min_m, kp_stride, kp_pad, hyp, threshold, hypF,
                                                                        cbound, rk, cmax, sfloor);
        }
        {
            VsProfScope ps(ctx, "ransac_screen_kernel");
            dim3 grid(vs_div_up(hyp, kScreenHyps), batch);
            ransac_screen_kernel<<<grid, 256, 0, ctx->stream>>>(rk, kp_pad, m, min_m, kp_stride, hyp, threshold, hypF, cmax, pot0);
        }
        {
            VsProfScope ps(ctx, "ransac_cand_kernel");
            ransac_cand_kernel<<<batch, 64 * kCandMax, 0, ctx->stream>>>(rk, kp_pad, m, min_m, kp_stride, hyp, threshold, hypF, pot0,
                                                                         no_sum_rule ? 0 : 1, cbound, sfloor);
        }
        if (int arc = vs_aux_job_point(ctx, 4)) return arc;
        {
            VsProfScope ps(ctx, "ransac_count_kernel");
            const bool lds = kp_pad <= kCntLdsMatches;
            const size_t dyn = (lds ? sizeof(float) * 4 * (size_t)kp_pad : 0) + sizeof(uint32_t) * kCntQueue * kCntWaves;
            if (dyn > 32 * 1024 && !ctx->attr_done["ransac_count"]) {   // beyond the default static + dynamic LDS limit
                VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ransac_count_kernel<true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)(sizeof(float) * 4 * kCntLdsMatches + sizeof(uint32_t) * kCntQueue * kCntWaves)));
                ctx->attr_done["ransac_count"] = true;
            }
            dim3 grid(vs_div_up(hyp, kCntHyps), batch);
            if (lds)
                ransac_count_kernel<true><<<grid, 64 * kCntWaves, dyn, ctx->stream>>>(rk, kp_pad, m, min_m, kp_stride, hyp, threshold, hypF, pot0,
                                                                                      cmax, hyp_count, hyp_sum, approx, cbound, use_floor);
            else
                ransac_count_kernel<false><<<grid, 64 * kCntWaves, dyn, ctx->stream>>>(rk, kp_pad, m, min_m, kp_stride, hyp, threshold, hypF, pot0,
                                                                                       cmax, hyp_count, hyp_sum, approx, cbound, use_floor);
        }
        // the closing stages: one launch when the four waves' sum buffers fit LDS (kp_stride <= 4096), else three
        const size_t fin_lds = sizeof(float) * (size_t)kTieGrid * kp_stride;
        static const bool split_finish = VS_EXPERIMENT_ENV("VSLAM_RANSAC_SPLIT_FINISH") != nullptr;   // A/B timing
        if (fin_lds <= 64 * 1024 && !split_finish) {
            VsProfScope ps(ctx, "ransac_finish_kernel");
            if (fin_lds > 32 * 1024 && !ctx->attr_done["ransac_finish"]) {
                VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ransac_finish_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                ctx->attr_done["ransac_finish"] = true;
            }
            ransac_finish_kernel<<<batch, kSelThreads, fin_lds, ctx->stream>>>(xy1, xy2, pairs, m, min_m, kp_stride, hyp, threshold, hypF,
                                                                              hyp_count, approx, hyp_sum, tie_idx, tie_n, F, mask, best,
                                                                              matches);
            VS_HIP(ctx, hipGetLastError());
            return VSLAM_OK;
        }
        {
            VsProfScope ps(ctx, "ransac_ties_kernel");
            ransac_ties_kernel<<<batch, kTieThreads, 0, ctx->stream>>>(m, min_m, hyp, hyp_count, approx, hyp_sum, tie_idx, tie_n);
        }
        {
            VsProfScope ps(ctx, "ransac_tiesum_kernel");
            dim3 grid(min(kTieGrid, vs_div_up(hyp, 64)), batch);
            if (sizeof(float) * (size_t)kp_stride > 40 * 1024 && !ctx->attr_done["ransac_tiesum"]) {
                VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ransac_tiesum_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * VSLAM_MAX_KP)));
                ctx->attr_done["ransac_tiesum"] = true;
            }
            ransac_tiesum_kernel<<<grid, 64, sizeof(float) * (size_t)kp_stride, ctx->stream>>>(xy1, xy2, pairs, m, min_m, kp_stride, hyp, hypF, tie_idx, tie_n,
                                                              hyp_sum);
        }
    }
    {
        VsProfScope ps(ctx, "ransac_select_kernel");
        ransac_select_kernel<<<batch, kSelThreads, 0, ctx->stream>>>(xy1, xy2, pairs, m, min_m, kp_stride, hyp,
                                                                     threshold, hypF, hyp_count, hyp_sum,
                                                                     F, mask, best, matches);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_ransac(vslam_ctx *ctx, const float *xy1, const float *xy2, const int32_t *pairs,
                     const int32_t *m, const int32_t *sets, int batch, int kp_stride, int hyp,
                     float threshold, float *F, uint8_t *mask, int32_t *best, int32_t *matches,
                     float *hypF, int32_t *hyp_count, float *hyp_sum) {
    int rc = vs_launch_ransac_solve(ctx, xy1, xy2, pairs, m, sets, batch, kp_stride, hyp, hypF);
    if (rc) return rc;
    if ((rc = vs_aux_job_point(ctx, 3))) return rc;
    return vs_launch_ransac_evaluate(ctx, xy1, xy2, pairs, m, hypF, batch, kp_stride, hyp, threshold, F, mask,
                                     best, matches, hyp_count, hyp_sum);
}

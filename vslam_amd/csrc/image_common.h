// Device helpers shared by the image kernels (gray.hip, response.hip, select.hip, blur.hip, brief.hip).
#pragma once
#include "ctx.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);   // clamp only guards halo cells that are never used
}

// monotone float <-> u32 so an unsigned max / radix order is the float order (no NaNs here)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}


// v_max3_f32 on values that are never NaN: fmaxf() would first canonicalise every operand (one extra
// v_max_f32 each).  Pure register instruction.
__device__ __forceinline__ float max3_nonan(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Correctly rounded sqrt of four non-negative finite floats.  Fast path: v_sqrt_f32 (1 ulp) plus the
// two-sided fma residual test; valid for 0 and for inputs >= 2^-96 (v_sqrt_f32 flushes denormal inputs, which
// is why the general sequence rescales).  If any lane of the wave holds a smaller non-zero input the whole
// wave takes sqrtf().  Both paths return the IEEE result, so which one runs never shows in the output.
__device__ __forceinline__ float sqrt_rn_fast1(float t) {
    const float r = __builtin_amdgcn_sqrtf(t);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, t), e_up = __builtin_fmaf(-r_up, r, t);
    float o = e_dn <= 0.f ? r_dn : r;   // t == 0: r_dn is NaN, the comparison is false
    o = e_up > 0.f ? r_up : o;
    return o;
}
__device__ __forceinline__ void sqrt_rn4(const float t[4], float out[4]) {
    // bits - 1 < 0x0F800000 - 1  <=>  0 < t < 2^-96  (t >= 0, so the bit pattern orders like the value)
    const uint32_t a = __float_as_uint(t[0]) - 1u, b = __float_as_uint(t[1]) - 1u, c = __float_as_uint(t[2]) - 1u,
                   d = __float_as_uint(t[3]) - 1u;
    const uint32_t lo = min(min(a, b), min(c, d));
    if (__builtin_expect(__any(lo < 0x0F800000u - 1u), 0)) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = sqrtf(t[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = sqrt_rn_fast1(t[i]);
    }
}


// Inline asm hides from the compiler's hazard recogniser what kind of instruction reads `d`: after a v_dot4 (response.hip
// feeds these the doubled gray values) a non-DOT vector instruction must not read the result for three wait states, and
// the compiler only guarantees one in front of an asm statement (seen in blur.hip, where an asm v_mad_u32_u24 right behind
// a v_dot4 read stale registers).  The plain C form `(float)((d >> 8 B) & 255)` selects the same instruction and is safe by
// construction, but costs min_eigen 4-6 % through the schedule it leads to; the asm form stays, and
// tests/test_isa_hazards.py (tools/isa_hazards.py) checks the distance in the generated code on every build.
template <int B>
__device__ __forceinline__ float cvt_ubyte(uint32_t d) {   // (float) of byte B of d
    float r;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(d));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(d));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(d));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(d));
    return r;
}
__device__ __forceinline__ float dpp_wave_shr1(float v, float fill) {   // lane i <- lane i-1, lane 0 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shl1(float v, float fill) {   // lane i <- lane i+1, lane 63 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}


// ---- candidate lists of the corner pipeline (response.hip -> select.hip) -----------------------------------------
// 64-bit entries, ordered float in the high word so that an unsigned compare is the float compare.
//   exact key   = ordered(response) << 32 | pixel offset
//   raw entry   = ordered(certified UPPER bound of response / c0) << 32 | kKeyUncertain? | pixel offset
//                 (the two-tier detector's list; kKeyUncertain: the cheap values cannot show that the pixel is a 3x3
//                 maximum, so it has to be compared with its exact neighbours)
constexpr uint32_t kKeyUncertain = 0x20000000u;
constexpr uint32_t kOffMask = 0x0FFFFFFFu;   // pixel offset part of an entry's low word

// ---- cornerMinEigenVal at single pixels, the oracle's exact sequence (oracle/vso_extract.cpp:44-114) -------------
__device__ __forceinline__ float sqrt_rn1(float t) {   // correctly rounded, per lane
    const uint32_t b = __float_as_uint(t) - 1u;
    if (__builtin_expect(b < 0x0F800000u - 1u, 0)) return sqrtf(t);   // 0 < t < 2^-96
    return sqrt_rn_fast1(t);
}

// calcMinEigenVal in float from the three box sums
__device__ __forceinline__ float min_eigen_from_sums(double Sxx, double Sxy, double Syy) {
    const float sxx = (float)Sxx, sxy = (float)Sxy, syy = (float)Syy;
    const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
    const float amc = a - c;
    const float t = amc * amc + b * b;
    return (a + c) - sqrt_rn1(t);
}

// A pixel whose 5x5 gray window lies inside the image (2 <= x < w - 2, 2 <= y < h - 2): float Sobel with the scale
// folded into the smoothing taps, products, horizontal sums then vertical in double.
// (`pitch` = bytes per row of src; `w` = the image's width)
__device__ __forceinline__ float min_eigen_exact_interior(const uint8_t *src, int pitch, int x, int y, float k0, float k1) {
    float g[5][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint8_t *p = src + (size_t)(y - 2 + r) * pitch + (x - 2);
        uint32_t d;
        __builtin_memcpy(&d, p, 4);
        g[r][0] = cvt_ubyte<0>(d); g[r][1] = cvt_ubyte<1>(d); g[r][2] = cvt_ubyte<2>(d); g[r][3] = cvt_ubyte<3>(d);
        g[r][4] = (float)p[4];
    }
    float hx[5][3], rr[5][3];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            hx[r][c] = g[r][c + 2] - g[r][c];
            const float a = g[r][c + 1] * k0;
            const float b = (g[r][c] + g[r][c + 2]) * k1;
            rr[r][c] = a + b;
        }
    double rxx[3], rxy[3], ryy[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float cxx[3], cxy[3], cyy[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float a = hx[r + 1][c] * k0;
            const float b = (hx[r][c] + hx[r + 2][c]) * k1;
            const float dx = a + b;
            const float dy = rr[r + 2][c] - rr[r][c];
            cxx[c] = dx * dx;
            cxy[c] = dx * dy;
            cyy[c] = dy * dy;
        }
        rxx[r] = ((double)cxx[0] + (double)cxx[1]) + (double)cxx[2];
        rxy[r] = ((double)cxy[0] + (double)cxy[1]) + (double)cxy[2];
        ryy[r] = ((double)cyy[0] + (double)cyy[1]) + (double)cyy[2];
    }
    return min_eigen_from_sums((rxx[0] + rxx[1]) + rxx[2], (rxy[0] + rxy[1]) + rxy[2], (ryy[0] + ryy[1]) + ryy[2]);
}

// Any pixel: the box filter's BORDER_REFLECT_101 on product coordinates, the Sobel's on gray coordinates.  Rolled
// loops on purpose (rare path, small code); the sums accumulate in the oracle's order.
__device__ __forceinline__ float min_eigen_exact_border(const uint8_t *src, int pitch, int w, int h, int x, int y, float k0, float k1) {
    double Sxx = 0, Sxy = 0, Syy = 0;
#pragma nounroll
    for (int r = 0; r < 3; r++) {
        double rxx = 0, rxy = 0, ryy = 0;
#pragma nounroll
        for (int c = 0; c < 3; c++) {
            const int px = reflect101(x - 1 + c, w), py = reflect101(y - 1 + r, h);
            const int xm = reflect101(px - 1, w), xp = reflect101(px + 1, w);
            float hx[3], rr[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint8_t *row = src + (size_t)reflect101(py - 1 + k, h) * pitch;
                const float gm = (float)row[xm], g0 = (float)row[px], gp = (float)row[xp];
                hx[k] = gp - gm;
                const float a = g0 * k0;
                const float b = (gm + gp) * k1;
                rr[k] = a + b;
            }
            const float a = hx[1] * k0;
            const float b = (hx[0] + hx[2]) * k1;
            const float dx = a + b;
            const float dy = rr[2] - rr[0];
            const float cxx = dx * dx, cxy = dx * dy, cyy = dy * dy;
            // ((c0 + c1) + c2): the first addition is 0 + c0, which is exact
            rxx = rxx + (double)cxx;
            rxy = rxy + (double)cxy;
            ryy = ryy + (double)cyy;
        }
        Sxx = Sxx + rxx;
        Sxy = Sxy + rxy;
        Syy = Syy + ryy;
    }
    return min_eigen_from_sums(Sxx, Sxy, Syy);
}

__device__ __forceinline__ float min_eigen_exact(const uint8_t *src, int pitch, int w, int h, int x, int y, float k0, float k1) {
    if (x >= 2 && x < w - 2 && y >= 2 && y < h - 2) return min_eigen_exact_interior(src, pitch, x, y, k0, k1);
    return min_eigen_exact_border(src, pitch, w, h, x, y, k0, k1);
}

}  // namespace

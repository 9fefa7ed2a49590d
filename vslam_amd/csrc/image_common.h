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


// Strips do not overlap, so the 3x3 test of a strip's first and last column lacks the neighbouring strip's
// column.  Such candidates are emitted with a flag in the (otherwise unused) top bits of the pixel offset and
// corner_select_kernel completes their test against the stored responses before anything else looks at them.
constexpr uint32_t kKeyCheckLeft = 0x80000000u, kKeyCheckRight = 0x40000000u;
constexpr uint32_t kOffMask = 0x3FFFFFFFu;   // pixel offset part of a key's low word


}  // namespace

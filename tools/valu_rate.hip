// Issue cost of the VALU instructions the front-end kernels are bound by, measured on the device:
// every SIMD runs `kWavesPerSimd` waves, each executing kIters x 16 independent copies of one
// instruction; cycles per wave-instruction = elapsed * clock / (waves per SIMD * instructions per wave).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIters = 2000;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// f32 / packed-f32 / f64 register files the asm bodies work on
#define KERNEL(NAME, BODY)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                          \
        float a[16];                                                                               \
        double d[16];                                                                              \
        typedef float v2f __attribute__((ext_vector_type(2)));                                     \
        v2f p[16];                                                                                 \
        typedef float v4f __attribute__((ext_vector_type(4)));                                     \
        v4f q4[4] = {};                                                                            \
        __shared__ float lds_pad[1024];                                                            \
        lds_pad[threadIdx.x] = seed;                                                               \
        __syncthreads();                                                                           \
        for (int i = 0; i < 16; i++) {                                                             \
            a[i] = seed + (float)i + (float)threadIdx.x;                                           \
            d[i] = (double)a[i];                                                                   \
            p[i] = (v2f){a[i], a[i] + 0.5f};                                                        \
        }                                                                                          \
        const float c = seed * 0.999f;                                                             \
        const double cd = (double)c;                                                               \
        const v2f cp = (v2f){c, c};                                                                 \
        uint32_t u = __float_as_uint(seed);                                                        \
        (void)cd; (void)cp; (void)u;                                                               \
        for (int it = 0; it < kIters; it++) { REP16(BODY) }                                        \
        float acc = 0;                                                                             \
        for (int i = 0; i < 16; i++) acc += a[i] + (float)d[i] + p[i].x + p[i].y + q4[i & 3].x + lds_pad[i];  \
        if (acc == 12345.678f) out[0] = acc;                                                       \
    }

#define B_MUL_F32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B_PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define B_PK_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define B_PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
#define B_ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define B_MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define B_FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
#define B_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define B_CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define B_RCP_F32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_SQRT_F32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define B_RSQ_F64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
#define B_RCP_F64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define B_SQRT_F64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
#define B_DIV_SCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(c) : "vcc");
#define B_DIV_FMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c) : "vcc");
#define B_DIV_FIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(u));
#define B_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B_MIN_U32(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
#define B_CMP_F32(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
#define B_CVT_UBYTE(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u));
#define B_SUB_SDWA(i) asm volatile("v_sub_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3" : "=v"(a[i]) : "v"(u));
#define B_CVT_F32_I32(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
#define B_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(u));
#define B_MAD_U32_U24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(u));
#define B_MUL_LO_U32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_CMP_U64(i) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(d[i]), "v"(cd) : "vcc");
#define B_BPERMUTE(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(u));
#define B_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));

#define B_CMP_CND_VCC(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
#define B_CND_SGPR(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c) : "s20", "s21");
#define B_CMP_CND_SGPR(i) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c) : "s20", "s21");
#define B_CMP_SGPR(i) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(c) : "s20", "s21");
#define B_MAX_F32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_SUB_F32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_AND_B32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_OR_B32(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
#define B_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
#define B_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(u));
#define B_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : : "vcc");
#define B_MUL_F32_E64(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_FMAC_F32(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B_MUL_F32_2(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(c));
#define B_MAC_U64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(u) : "vcc");
#define B_ALIGNBYTE(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(u));
#define B_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(u));
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
#define B_LSHL_ADD_U64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(cd));
#define B_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define B_DS_READ(i) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(u & 0xFFCu));
#define B_DS_READ128(i) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q4[i & 3]) : "v"(u & 0xFF0u));

#define LIST(X)                                                                                    \
    X(mul_f32, B_MUL_F32) X(add_f32, B_ADD_F32) X(fma_f32, B_FMA_F32) X(pk_mul_f32, B_PK_MUL)      \
    X(pk_add_f32, B_PK_ADD) X(pk_fma_f32, B_PK_FMA) X(add_f64, B_ADD_F64) X(mul_f64, B_MUL_F64)    \
    X(fma_f64, B_FMA_F64) X(cvt_f64_f32, B_CVT_F64_F32) X(cvt_f32_f64, B_CVT_F32_F64)              \
    X(rcp_f32, B_RCP_F32) X(sqrt_f32, B_SQRT_F32) X(rsq_f64, B_RSQ_F64) X(rcp_f64, B_RCP_F64)      \
    X(sqrt_f64, B_SQRT_F64) X(div_scale_f32, B_DIV_SCALE) X(div_fmas_f32, B_DIV_FMAS)              \
    X(div_fixup_f32, B_DIV_FIXUP) X(xor_b32, B_XOR) X(bcnt_u32_b32, B_BCNT) X(max3_f32, B_MAX3)    \
    X(min_u32, B_MIN_U32) X(cndmask_b32, B_CNDMASK) X(cmp_gt_f32, B_CMP_F32)                        \
    X(cvt_f32_ubyte1, B_CVT_UBYTE) X(sub_u32_sdwa, B_SUB_SDWA) X(cvt_f32_i32, B_CVT_F32_I32)       \
    X(dot4_u32_u8, B_DOT4) X(mad_u32_u24, B_MAD_U32_U24) X(mul_lo_u32, B_MUL_LO_U32)               \
    X(cmp_gt_u64, B_CMP_U64) X(ds_bpermute_b32, B_BPERMUTE) X(mov_b32_dpp, B_MOV_DPP)              \
    X(cmp_cnd_vcc_pair, B_CMP_CND_VCC) X(cndmask_sgpr, B_CND_SGPR) X(cmp_cnd_sgpr_pair, B_CMP_CND_SGPR) \
    X(cmp_e64_sgpr, B_CMP_SGPR) X(max_f32, B_MAX_F32) X(sub_f32, B_SUB_F32) X(med3_f32, B_MED3)     \
    X(add_u32, B_ADD_U32) X(and_b32, B_AND_B32) X(or_b32, B_OR_B32) X(lshlrev_b32, B_LSHL)          \
    X(mov_b32, B_MOV) X(add3_u32, B_ADD3) X(addc_co_u32, B_ADDC) X(mul_f32_e64, B_MUL_F32_E64)      \
    X(fmac_f32, B_FMAC_F32) X(mul_f32_3reg, B_MUL_F32_2) X(mad_u64_u32, B_MAC_U64)                  \
    X(alignbyte_b32, B_ALIGNBYTE) X(perm_b32, B_PERM) X(bfe_u32, B_BFE) X(lshl_add_u64, B_LSHL_ADD_U64) \
    X(readlane_b32, B_READLANE) X(ds_read_b32_wait, B_DS_READ) X(ds_read_b128_wait, B_DS_READ128)

#define DEF(NAME, BODY) KERNEL(k_##NAME, BODY)
LIST(DEF)

int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const double clk_hz = prop.clockRate * 1e3;
    const int cus = prop.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("device %s  CUs %d  clock %.0f MHz  waves/SIMD %d\n", prop.name, cus, clk_hz / 1e6, waves_per_simd);
    printf("%-18s %10s %14s\n", "instruction", "ms", "cycles/wave-inst");
    // one 256-thread block = 4 waves = one per SIMD of a CU; waves_per_simd blocks per CU
    const int blocks = cus * waves_per_simd;
#define RUN(NAME, BODY)                                                                            \
    {                                                                                              \
        k_##NAME<<<blocks, 256>>>(out, 1.25f);                                                     \
        (void)hipDeviceSynchronize();                                                                  \
        (void)hipEventRecord(e0);                                                                      \
        k_##NAME<<<blocks, 256>>>(out, 1.25f);                                                     \
        (void)hipEventRecord(e1);                                                                      \
        (void)hipEventSynchronize(e1);                                                                 \
        float ms = 0;                                                                              \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                        \
        const double cyc = ms * 1e-3 * clk_hz / ((double)waves_per_simd * kIters * 16);            \
        printf("%-18s %10.3f %14.2f\n", #NAME, ms, cyc);                                           \
    }
    LIST(RUN)
    return 0;
}

// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands on gfx950, as the matcher would use it: 256-bit
// descriptors spread to +-1 nibbles (bit 0 -> +1.0 = 0x2, bit 1 -> -1.0 = 0xA), four K = 64 steps, unit scales (E8M0 127).
// Checks (i) exactness: D[row][col] = 256 - 2 Hamming(a_row, b_col) for random bits, (ii) the result layout
// col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), and (iii) times a dependent-free stream of them.
// hipcc --offload-arch=gfx950 -O2 -o mfma_fp4_probe tools/mfma_fp4_probe.hip && ./mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t spread8(uint32_t b) {   // 8 bits -> 8 nibbles 0x2 | bit << 3
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 3) | 0x22222222u;
}
__device__ __forceinline__ v8i operand(const uint32_t *desc, int step, int half) {   // bits [64 step + 32 half, + 32)
    const uint32_t w = desc[2 * step + half];
    v8i r = {(int)spread8(w & 0xFF), (int)spread8((w >> 8) & 0xFF), (int)spread8((w >> 16) & 0xFF), (int)spread8(w >> 24), 0, 0, 0, 0};
    return r;
}

__global__ void probe(const uint32_t *a, const uint32_t *b, float *out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f c = {0};
    for (int s = 0; s < 4; s++)
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(operand(a + 8 * r, s, h), operand(b + 8 * r, s, h), c, 4, 4, 0, 0x7F7F7F7F, 0,
                                                            0x7F7F7F7F);
    for (int g = 0; g < 16; g++) out[lane * 16 + g] = c[g];
}

__global__ void rate(const uint32_t *a, const uint32_t *b, float *out, int iters) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    v8i x[4], y[4];
    for (int s = 0; s < 4; s++) {
        x[s] = operand(a + 8 * r, s, h);
        y[s] = operand(b + 8 * r, s, h);
    }
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x[0], y[0], c0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x[1], y[1], c1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x[2], y[2], c2, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x[3], y[3], c3, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    std::vector<uint32_t> a(32 * 8), b(32 * 8);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto &v : a) v = rnd();
    for (auto &v : b) v = rnd();
    uint32_t *da, *db;
    float *dout;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 4 * 256 * 1024);
    hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int reg = 0; reg < 16; reg++) {
            const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            int ham = 0;
            for (int k = 0; k < 8; k++) ham += __builtin_popcount(a[row * 8 + k] ^ b[col * 8 + k]);
            const float want = (float)(256 - 2 * ham);
            if (out[lane * 16 + reg] != want) {
                if (bad < 8) printf("lane %d reg %d: got %g want %g\n", lane, reg, out[lane * 16 + reg], want);
                bad++;
            }
        }
    printf("mismatches: %d of 1024\n", bad);
    // rate: 1024 workgroups x 4 waves, each 4 independent accumulators
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    rate<<<1024, 256>>>(da, db, dout, 16);
    hipEventRecord(e0);
    rate<<<1024, 256>>>(da, db, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 1024.0 * 4 * iters * 4;                  // instructions
    printf("%.3f ms for %.0f MFMAs: %.1f cycles per instruction and SIMD at 2.4 GHz (4096 waves on 1024 SIMDs), %.2f Pop/s\n", ms, n,
           ms * 1e-3 * 2.4e9 / (n / 1024.0), n * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e15);
    return bad != 0;
}

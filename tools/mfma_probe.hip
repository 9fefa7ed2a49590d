// Probe of v_mfma_i32_32x32x32_i8 on gfx950: which (lane, register) of the result holds D[row][col] when the A operand
// of lane l carries row (l & 31) and the B operand carries column (l & 31).  Prints the mismatches against
//   col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
// hipcc --offload-arch=gfx950 -O2 -o mfma_probe tools/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int *arow, const int *bcol, int *out) {
    const int lane = threadIdx.x;
    // A: row r = lane & 31 gets arow[r] in every byte; B: column c = lane & 31 gets bcol[c] in every byte
    const int av = arow[lane & 31] * 0x01010101, bv = bcol[lane & 31] * 0x01010101;
    v4i a = {av, av, av, av}, b = {bv, bv, bv, bv};
    v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = c[r];
}

int main() {
    std::vector<int> arow(32), bcol(32), out(64 * 16);
    for (int i = 0; i < 32; i++) {
        arow[i] = i % 7 + 1;        // asymmetric, small: products fit
        bcol[i] = i % 5 + 1;
    }
    int *da, *db, *dout;
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, arow.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(db, bcol.data(), 128, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout);
    hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int reg = 0; reg < 16; reg++) {
            const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const int want = 32 * arow[row] * bcol[col];
            if (out[lane * 16 + reg] != want) {
                if (bad < 8) printf("lane %d reg %d: got %d want %d\n", lane, reg, out[lane * 16 + reg], want);
                bad++;
            }
        }
    printf("mismatches: %d of 1024\n", bad);
    return bad != 0;
}

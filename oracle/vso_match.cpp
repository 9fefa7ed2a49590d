// ORACLE (test infrastructure only) — brute-force Hamming k=2 matching + Lowe ratio.
//
// Follows /root/reference/src/Frame.cpp:83-94 and the OpenCV 4.x routine it calls:
//   cv::BFMatcher(NORM_HAMMING)->knnMatch(query, train, out, 2)
// [OpenCV, from memory: modules/core/src/batch_distance.cpp] keeps, per query row, a K-long
// list sorted by distance; a new train row j enters iff d < worst (strict) and is shifted
// past entries with dist > d (strict), so equal distances keep the LOWER train index first.
// DMatch::distance is the integer Hamming distance stored as float.
// Parity: UNPINNED (no OpenCV here, no golden vectors in the reference).
#include "vso.h"

#include <climits>
#include <cstring>

namespace {
static inline uint32_t hamming32(const uint8_t *a, const uint8_t *b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return (uint32_t)(__builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
                      __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]));
}
}  // namespace

extern "C" {

uint32_t vso_hamming256(const uint8_t *a, const uint8_t *b) { return hamming32(a, b); }

int vso_match_knn2(const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx0,
                   int32_t *dist0, int32_t *idx1, int32_t *dist1) {
    if (n1 < 0 || n2 < 0) return -1;
    for (int q = 0; q < n1; q++) {
        int bd[2] = {INT_MAX, INT_MAX};
        int bi[2] = {-1, -1};
        const uint8_t *qa = d1 + (size_t)q * 32;
        for (int j = 0; j < n2; j++) {
            const int d = (int)hamming32(qa, d2 + (size_t)j * 32);
            if (d < bd[1]) {
                int k = 0;                       // K == 2: shift loop runs over slot 0 only
                if (bd[0] > d) {
                    bd[1] = bd[0];
                    bi[1] = bi[0];
                    k = -1;
                }
                bd[k + 1] = d;
                bi[k + 1] = j;
            }
        }
        idx0[q] = bi[0];
        dist0[q] = bd[0];
        idx1[q] = bi[1];
        dist1[q] = bd[1];
    }
    return 0;
}

int vso_match_knn2_ratio(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                         int32_t *out_pairs, int32_t *out_m) {
    if (n1 < 0 || n2 < 2) return -1;             // m[1] is read unconditionally, Frame.cpp:91
    int m = 0;
    for (int q = 0; q < n1; q++) {
        int32_t i0, e0, i1, e1;
        vso_match_knn2(d1 + (size_t)q * 32, 1, d2, n2, &i0, &e0, &i1, &e1);
        const float f0 = (float)e0, f1 = (float)e1;   // DMatch::distance is float
        if (f0 < f1 * 0.7) {                          // float < (float * double), Frame.cpp:91
            out_pairs[2 * m] = q;                     // queryIdx
            out_pairs[2 * m + 1] = i0;                // trainIdx
            m++;
        }
    }
    *out_m = m;
    return 0;
}

}  // extern "C"

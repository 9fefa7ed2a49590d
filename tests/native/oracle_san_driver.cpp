// The whole oracle front-end on small synthetic frames, built with -fsanitize=address,undefined (tests/test_sanitizers.py):
// every stage the parity tests lean on runs once over inputs with the awkward shapes -- odd sizes, keypoints at the border,
// fewer than 8 matches, empty inputs -- so that an out-of-bounds index or an overflow in the checker itself shows up here
// rather than as a parity "difference" on the GPU box.  Prints a checksum of everything it computed.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../oracle/vso.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}
static uint64_t sum = 1469598103934665603ull;
static void mix(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) sum = (sum ^ b[i]) * 1099511628211ull;
}

// blocky texture + noise; frame 1 = frame 0 shifted
static void make_pair(int w, int h, std::vector<uint8_t> &a, std::vector<uint8_t> &b, int dx, int dy) {
    std::vector<uint8_t> base((size_t)(w + 64) * (h + 64));
    for (int y = 0; y < h + 64; y++)
        for (int x = 0; x < w + 64; x++) base[(size_t)y * (w + 64) + x] = (uint8_t)(((x / 7) * 37 + (y / 5) * 91 + ((x / 7) ^ (y / 5)) * 53) & 255);
    a.assign((size_t)w * h * 3, 0);
    b.assign((size_t)w * h * 3, 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                a[((size_t)y * w + x) * 3 + c] = (uint8_t)((base[(size_t)(y + 32) * (w + 64) + x + 32] + (rnd() & 7) + 11 * c) & 255);
                b[((size_t)y * w + x) * 3 + c] = (uint8_t)((base[(size_t)(y + 32 + dy) * (w + 64) + x + 32 + dx] + (rnd() & 7) + 11 * c) & 255);
            }
}

int main() {
    int8_t pattern[1024];
    for (int i = 0; i < 1024; i++) pattern[i] = (int8_t)((int)(rnd() % 27) - 13);
    float sa, ca;
    if (vso_sincos_deg(-1.0f, &sa, &ca)) return 1;
    const int shapes[][2] = {{161, 123}, {96, 80}, {64, 70}, {320, 200}};
    for (auto &sh : shapes) {
        const int w = sh[0], h = sh[1], maxc = 400;
        std::vector<uint8_t> fa, fb;
        make_pair(w, h, fa, fb, 3, 2);
        std::vector<float> xy1(2 * maxc), xy2(2 * maxc);
        std::vector<uint8_t> d1(32 * maxc), d2(32 * maxc);
        std::vector<int32_t> kd1(maxc), kd2(maxc);
        int32_t n1 = 0, n2 = 0, nd1 = 0, nd2 = 0;
        if (vso_extract_features(fa.data(), w, h, 3 * w, maxc, ca, sa, pattern, xy1.data(), d1.data(), kd1.data(), &n1, &nd1)) return 2;
        if (vso_extract_features(fb.data(), w, h, 3 * w, maxc, ca, sa, pattern, xy2.data(), d2.data(), kd2.data(), &n2, &nd2)) return 2;
        mix(xy1.data(), 8 * (size_t)n1);
        mix(d2.data(), 32 * (size_t)n2);
        mix(kd1.data(), 4 * (size_t)n1);
        // k-d queries, including far outside the image and with radius 0
        for (int q = 0; q < 50 && n1 > 0; q++) {
            int32_t hits[64];
            const float qx = (float)((int)(rnd() % (unsigned)(w + 40)) - 20), qy = (float)((int)(rnd() % (unsigned)(h + 40)) - 20);
            const int c = vso_kdtree_radius_frame(kd1.data(), xy1.data(), n1, qx, qy, (float)(q % 5), hits, 64);
            mix(&c, 4);
        }
        // match + RANSAC, full and with too few matches
        for (int H : {1, 64}) {
            std::vector<int32_t> matches(2 * (size_t)(n1 > 0 ? n1 : 1));
            int32_t nm = 0, prelim = 0;
            float F[9] = {0};
            const int rc = vso_match_features(xy1.data(), d1.data(), n1, xy2.data(), d2.data(), n2, 77u + (uint32_t)H, H, 10.f, matches.data(), &nm,
                                              F, &prelim);
            mix(&rc, 4);
            mix(&nm, 4);
            mix(matches.data(), 8 * (size_t)nm);
            if (rc == 0 && nm >= 8) {
                const float K[9] = {500, 0, w / 2.f, 0, 500, h / 2.f, 0, 0, 1};
                float R[9], t[3], c2[12];
                const float c1[12] = {500, 0, w / 2.f, 0, 0, 500, h / 2.f, 0, 0, 0, 1, 0};
                vso_extract_Rt(F, K, R, t);
                vso_camera_matrix(K, R, t, c2);
                std::vector<float> p1(2 * (size_t)nm), p2(2 * (size_t)nm), pts(4 * (size_t)nm);
                for (int i = 0; i < nm; i++) {
                    p1[2 * i] = xy1[2 * matches[2 * i]], p1[2 * i + 1] = xy1[2 * matches[2 * i] + 1];
                    p2[2 * i] = xy2[2 * matches[2 * i + 1]], p2[2 * i + 1] = xy2[2 * matches[2 * i + 1] + 1];
                }
                vso_triangulate(p1.data(), p2.data(), nm, c1, c2, pts.data());
                std::vector<int32_t> ids((size_t)(n2 > nm ? n2 : nm), -1), keep((size_t)nm);
                int32_t nk = 0;
                double err = 0;
                vso_reprojection_filter(pts.data(), p1.data(), p2.data(), nm, c1, c2, ids.data(), 4.f, keep.data(), &nk, &err);
                mix(&nk, 4);
            }
        }
        // a handful of matches only (< 8: find_fundamental cannot draw a set), and none at all
        {
            std::vector<int32_t> matches(16);
            int32_t nm = 0, prelim = 0;
            float F[9] = {0};
            const int few = n1 < 5 ? n1 : 5;
            int rc = vso_match_features(xy1.data(), d1.data(), few, xy2.data(), d2.data(), n2 < 6 ? n2 : 6, 3u, 16, 10.f, matches.data(), &nm, F, &prelim);
            mix(&rc, 4);
            rc = vso_match_features(xy1.data(), d1.data(), 0, xy2.data(), d2.data(), 0, 3u, 16, 10.f, matches.data(), &nm, F, &prelim);
            mix(&rc, 4);
        }
        // the grid extractor (draws into its input)
        {
            const int cap = 20000;
            std::vector<float> gxy(2 * (size_t)cap), gao(2 * (size_t)cap);
            std::vector<uint8_t> gd(32 * (size_t)cap);
            int32_t gn = 0;
            std::vector<uint8_t> img = fa;
            const int rc = vso_extract_features_grid(img.data(), w, h, 3 * w, 2, 3, pattern, gxy.data(), gd.data(), gao.data(), cap, &gn);
            mix(&rc, 4);
            mix(&gn, 4);
            mix(gd.data(), 32 * (size_t)(gn > 0 ? gn : 0));
        }
    }
    // sets: every n around the Lemire rejection boundaries, min_items below 8
    for (int n : {8, 9, 17, 100, 1000, 65537})
        for (int mi : {8, 3, 0}) {
            std::vector<int32_t> sets(8 * 33);
            vso_ransac_sets(99u + (uint32_t)n, n, mi, 33, sets.data());
            mix(sets.data(), sets.size() * 4);
        }
    std::printf("%016llx\n", (unsigned long long)sum);
    return 0;
}

// A queue of batches through vslam::Pipeline (include/vslam/Pipeline.h): the shape the reference's capture loop
// (src/vslam.cpp:53-77: one frame pair at a time, start to finish) takes on a device -- up to three batches in flight, each
// collected in order.  Frames here are synthetic blocks so that the example is self-contained.
//
//   g++ -std=c++17 -O2 examples/batches_in_flight.cpp -Iinclude -Lvslam_amd -lvslam_amd -Wl,-rpath,$PWD/vslam_amd -o batches_in_flight
//   ./batches_in_flight [batches] [pairs per batch]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

#include "vslam/Pipeline.h"

static void make_frames(std::vector<uint8_t> &last, std::vector<uint8_t> &cur, int pairs, int w, int h, unsigned seed) {
    const size_t fb = (size_t)w * h * 3;
    last.resize(fb * pairs);
    cur.resize(fb * pairs);
    for (int p = 0; p < pairs; p++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                auto tex = [&](int xx, int yy) { return (uint8_t)((((xx / 9) * 37 + (yy / 7) * 91 + ((xx / 9) ^ (yy / 7)) * 53 + (int)seed + 17 * p) & 255)); };
                for (int c = 0; c < 3; c++) {
                    last[(p * fb) + ((size_t)y * w + x) * 3 + c] = (uint8_t)(tex(x + 40, y + 40) + 5 * c);
                    cur[(p * fb) + ((size_t)y * w + x) * 3 + c] = (uint8_t)(tex(x + 44, y + 42) + 5 * c);   // the scene moved by (4, 2)
                }
            }
}

int main(int argc, char **argv) {
    const int batches = argc > 1 ? std::atoi(argv[1]) : 6, pairs = argc > 2 ? std::atoi(argv[2]) : 4;
    const int w = 640, h = 480;
    try {
        vslam::Pipeline pipe(/*device*/ 0, /*in flight*/ 3);
        std::vector<std::vector<uint8_t>> last((size_t)batches), cur((size_t)batches);
        for (int b = 0; b < batches; b++) make_frames(last[(size_t)b], cur[(size_t)b], pairs, w, h, 1000u * (unsigned)b);
        std::deque<int64_t> q;
        size_t records = 0, matches = 0;
        auto take = [&](int64_t t) {
            for (const vslam::PairRecord &r : pipe.collect(t)) {
                records++;
                matches += r.matches.size();
                if (r.first_frame % (uint64_t)pairs == 0)
                    std::printf("pair %llu: hypothesis %d, %d inliers, F[2][2] = %g\n", (unsigned long long)r.first_frame, r.winner, r.inliers, r.F[8]);
            }
        };
        const auto t0 = std::chrono::steady_clock::now();
        for (int b = 0; b < batches; b++) {
            q.push_back(pipe.submit_pairs(last[(size_t)b].data(), cur[(size_t)b].data(), pairs, w, h, 3 * w, /*max_corners*/ 1000,
                                          /*hypotheses*/ 256, /*threshold*/ 10.f, /*seed*/ 42u, /*first_pair*/ (uint64_t)b * pairs));
            if ((int)q.size() == pipe.size()) {   // at most pipe.size() batches queued: collect the oldest
                take(q.front());
                q.pop_front();
            }
        }
        while (!q.empty()) {
            take(q.front());
            q.pop_front();
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%zu records, %zu inlier matches, %.1f ms for %d batches of %d pairs (uploads included)\n", records, matches, s * 1e3,
                    batches, pairs);
        return records == (size_t)batches * pairs ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}

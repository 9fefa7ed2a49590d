// Record files through the C++ classes (include/vslam/Ingest.h): `write <path>` produces a known file,
// `read <path>` dumps one line per record so the Python side can compare.  No device needed.
#include <cstdio>
#include <cstring>
#include <string>

#include "vslam/Ingest.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    const std::string mode = argv[1], path = argv[2];
    try {
        if (mode == "write") {
            vslam::RecordHeader h;
            h.width = 640;
            h.height = 480;
            h.max_corners = 500;
            h.hypotheses = 512;
            h.threshold = 10.f;
            h.seed = 0xC0FFEEu;
            vslam::RecordWriter w(path, h);
            for (int i = 0; i < 3; i++) {
                vslam::PairRecord r;
                r.first_frame = 1000u + (unsigned)i;
                r.winner = i - 1;
                r.inliers = 10 * i;
                r.score = 0.5f * (float)i;
                for (int k = 0; k < 9; k++) r.F[k] = (float)(i * 9 + k) * 0.125f;
                for (int k = 0; k < 10 * i; k++) r.matches.push_back({k, 2 * k + i});
                w.append(r);
            }
            w.close();
            return 0;
        }
        if (mode == "read") {
            vslam::RecordReader rd(path);
            const vslam::RecordHeader &h = rd.header();
            std::printf("H %u %u %u %u %u %a %u\n", h.version, h.width, h.height, h.max_corners, h.hypotheses, (double)h.threshold, h.seed);
            vslam::PairRecord r;
            while (rd.next(r)) {
                unsigned sb;
                std::memcpy(&sb, &r.score, 4);
                std::printf("R %llu %d %d %08x %zu", (unsigned long long)r.first_frame, r.winner, r.inliers, sb, r.matches.size());
                for (int k = 0; k < 9; k++) {
                    unsigned fb;
                    std::memcpy(&fb, &r.F[k], 4);
                    std::printf(" %08x", fb);
                }
                long long acc = 0;
                for (auto &m : r.matches) acc = acc * 31 + m.first * 7 + m.second;
                std::printf(" %lld\n", acc);
            }
            return 0;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 2;
}

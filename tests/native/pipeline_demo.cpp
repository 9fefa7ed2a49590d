// vslam::Pipeline (include/vslam/Pipeline.h) the way a C++ consumer would use it: a queue of batches of uneven size, up to
// `in_flight` of them on the device at once, each collected in submission order; records written out for the Python test
// to compare with the oracle.
//   in : int32 w, h, max_corners, hyp, seed, pairs, in_flight ; then `pairs` last frames, `pairs` current frames
//   out: per pair: int32 winner, inliers, n ; float F[9] ; n x (int32, int32)
#include <cstdio>
#include <deque>
#include <vector>

#include "vslam/Pipeline.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    int hdr[7];
    if (!fi || fread(hdr, 4, 7, fi) != 7) return 3;
    const int w = hdr[0], h = hdr[1], maxc = hdr[2], hyp = hdr[3], pairs = hdr[5], in_flight = hdr[6];
    const unsigned seed = (unsigned)hdr[4];
    const size_t fb = (size_t)w * h * 3;
    std::vector<unsigned char> last(fb * pairs), cur(fb * pairs);
    if (fread(last.data(), 1, last.size(), fi) != last.size() || fread(cur.data(), 1, cur.size(), fi) != cur.size()) return 3;
    fclose(fi);
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 5;
    auto write = [&](const std::vector<vslam::PairRecord> &rec, int first) {
        for (size_t i = 0; i < rec.size(); i++) {
            const vslam::PairRecord &r = rec[i];
            if (r.first_frame != (uint64_t)first + i) return false;
            const int head[3] = {r.winner, r.inliers, (int)r.matches.size()};
            fwrite(head, 4, 3, fo);
            fwrite(r.F, 4, 9, fo);
            for (auto &m : r.matches) {
                const int pr[2] = {m.first, m.second};
                fwrite(pr, 4, 2, fo);
            }
        }
        return true;
    };
    vslam::Pipeline pipe(0, in_flight);
    if (pipe.size() != in_flight) return 4;
    struct Job {
        int64_t ticket;
        int first;
    };
    std::deque<Job> q;
    // batches of 1, 2, 3, 1, 2, 3, ... pairs until the frames run out: the queue length is not a multiple of in_flight
    for (int first = 0, k = 0; first < pairs; k++) {
        const int n = std::min(1 + k % 3, pairs - first);
        q.push_back({pipe.submit_pairs(last.data() + fb * first, cur.data() + fb * first, n, w, h, 3 * w, maxc, hyp, 10.f, seed,
                                       (uint64_t)first),
                     first});
        first += n;
        if ((int)q.size() == pipe.size()) {
            if (!write(pipe.collect(q.front().ticket), q.front().first)) return 6;
            q.pop_front();
        }
    }
    while (!q.empty()) {
        if (!write(pipe.collect(q.front().ticket), q.front().first)) return 6;
        q.pop_front();
    }
    // a ticket cannot be collected twice
    bool threw = false;
    try {
        pipe.collect(0);
    } catch (const std::exception &) {
        threw = true;
    }
    fclose(fo);
    return threw ? 0 : 7;
}

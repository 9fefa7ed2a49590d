// vslam::DevicePool (include/vslam/MultiDevice.h) the way a C++ consumer would use it: a batch of frame pairs from a file,
// sharded over the listed device slots, records written out for the Python test to compare with the oracle.
//   in : int32 w, h, max_corners, hyp, seed, pairs, n_slots, slots[n_slots] ; then `pairs` last frames, `pairs` current frames
//   out: per pair: int32 winner, inliers, n ; float F[9] ; n x (int32, int32)
#include <cstdio>
#include <vector>

#include "vslam/MultiDevice.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    int hdr[7];
    if (!fi || fread(hdr, 4, 7, fi) != 7) return 3;
    const int w = hdr[0], h = hdr[1], maxc = hdr[2], hyp = hdr[3], pairs = hdr[5], n_slots = hdr[6];
    const unsigned seed = (unsigned)hdr[4];
    std::vector<int> slots((size_t)n_slots);
    if (fread(slots.data(), 4, (size_t)n_slots, fi) != (size_t)n_slots) return 3;
    const size_t fb = (size_t)w * h * 3;
    std::vector<unsigned char> last(fb * pairs), cur(fb * pairs);
    if (fread(last.data(), 1, last.size(), fi) != last.size() || fread(cur.data(), 1, cur.size(), fi) != cur.size()) return 3;
    fclose(fi);
    vslam::DevicePool pool(slots);
    if (pool.size() != n_slots) return 4;
    std::vector<vslam::PairRecord> rec = pool.frontend_pairs(last.data(), cur.data(), pairs, w, h, 3 * w, maxc, hyp, 10.f, seed);
    // a second batch on the same pool (buffers are reused; results must not change)
    std::vector<vslam::PairRecord> again = pool.frontend_pairs(last.data(), cur.data(), pairs, w, h, 3 * w, maxc, hyp, 10.f, seed);
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 5;
    for (int i = 0; i < pairs; i++) {
        const vslam::PairRecord &r = rec[(size_t)i], &a = again[(size_t)i];
        const int same = r.winner == a.winner && r.inliers == a.inliers && r.matches == a.matches && !memcmp(r.F, a.F, 36);
        if (!same) return 6;
        const int head[3] = {r.winner, r.inliers, (int)r.matches.size()};
        fwrite(head, 4, 3, fo);
        fwrite(r.F, 4, 9, fo);
        for (auto &m : r.matches) {
            const int pr[2] = {m.first, m.second};
            fwrite(pr, 4, 2, fo);
        }
    }
    fclose(fo);
    return 0;
}

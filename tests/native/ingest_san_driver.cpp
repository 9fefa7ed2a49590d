// vslam::run_sequence / run_sequence_devices (vslam_amd/host/ingest.cpp, compiled into this binary) against
// tests/native/capi_stub.cpp, under -fsanitize=thread or address,undefined: the reader pool, the double-buffer hand-over and
// the slot threads with real concurrency and no GPU.  Writes a small raw video, runs the loop with several batch sizes,
// reader counts and slot counts, and holds every record to what the stub's checksum of that frame pair must give.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../vslam_amd/host/host_internal.h"
#include "vslam/Ingest.h"

extern "C" uint32_t vslam_stub_pair_value(const uint8_t *a, const uint8_t *b, size_t frame_bytes, uint32_t seed);
extern "C" void vslam_stub_set_capacity_period(int period);

// the four helpers ingest.cpp takes from adapters.cpp (which drags in the whole C ABI)
namespace vslam {
namespace detail {
vslam_ctx *context() {
    static vslam_ctx *c = [] {
        vslam_ctx *x = nullptr;
        if (vslam_ctx_create(0, &x) != VSLAM_OK) throw std::runtime_error("stub context");
        return x;
    }();
    return c;
}
void check(int rc, const char *what) {
    if (rc != VSLAM_OK) throw std::runtime_error(what);
}
const std::vector<s8> &brief_pattern() {
    static const std::vector<s8> p(1024, 1);
    return p;
}
void fill_extract_params(vslam_extract_params &p, int max_corners, const int8_t *d_pattern) {
    p.max_corners = max_corners;
    p.quality = 0.01;
    p.min_distance = 3;
    p.cos_a = 1.f;
    p.sin_a = 0.f;
    p.d_pattern = d_pattern;
}
}  // namespace detail
}  // namespace vslam

int main(int argc, char **argv) {
    if (argc != 2) return 2;
    const std::string dir = argv[1];
    const int w = 24, h = 10, K = 12, frames = 53;
    const size_t fb = (size_t)w * h * 3;
    std::vector<uint8_t> video(fb * frames);
    uint32_t s = 1;
    for (auto &b : video) {
        s = s * 1664525u + 1013904223u;
        b = (uint8_t)(s >> 24);
    }
    const std::string vpath = dir + "/v.bgr";
    FILE *f = fopen(vpath.c_str(), "wb");
    if (!f || fwrite(video.data(), 1, video.size(), f) != video.size()) return 3;
    fclose(f);
    int checked = 0;
    auto verify = [&](const std::string &rpath, uint32_t seed, uint64_t n_frames) {
        vslam::RecordReader rd(rpath);
        vslam::PairRecord r;
        uint64_t i = 0;
        while (rd.next(r)) {
            const uint32_t v = vslam_stub_pair_value(video.data() + fb * i, video.data() + fb * (i + 1), fb, seed ^ (uint32_t)i);
            const int n = (int)(v % (uint32_t)(K + 1));
            if (r.first_frame != i || r.winner != (int32_t)(v % 5u) - 1 || r.inliers != n || (int)r.matches.size() != n)
                throw std::runtime_error("record " + std::to_string(i) + " of " + rpath + " is not that pair's");
            for (int j = 0; j < n; j++)
                if (r.matches[(size_t)j].first != (int32_t)((v + (uint32_t)j) % (uint32_t)K)) throw std::runtime_error("matches differ");
            if (r.winner >= 0 && r.F[0] != (float)(v & 1023u)) throw std::runtime_error("F differs");
            i++;
            checked++;
        }
        if (i != n_frames - 1) throw std::runtime_error("record count " + std::to_string(i) + " in " + rpath);
    };
    try {
        int run = 0;
        for (int batch : {2, 3, 7, 16, 64})
            for (int readers : {1, 4, 16}) {
                vslam::SequenceOptions o;
                o.width = w, o.height = h, o.batch_frames = batch, o.max_corners = K, o.hypotheses = 8, o.threshold = 10.f;
                o.seed = 0xABCD0000u + (uint32_t)run, o.reader_threads = readers;
                const std::string rp = dir + "/r" + std::to_string(run) + ".bin";
                const vslam::SequenceStats st = vslam::run_sequence(vpath, rp, o);
                if (st.frames != (uint64_t)frames || st.pairs != (uint64_t)frames - 1) throw std::runtime_error("stats");
                verify(rp, o.seed, frames);
                for (int slots : {1, 3, 5}) {
                    if (batch == 64 && slots == 5) continue;
                    const std::string rq = dir + "/q" + std::to_string(run) + "_" + std::to_string(slots) + ".bin";
                    vslam::run_sequence_devices(vpath, rq, o, std::vector<int>((size_t)slots, 0));
                    verify(rq, o.seed, frames);
                }
                run++;
            }
        // every third batch reports VSLAM_ERR_CAPACITY once: the loop repeats it with whole-image lists; same records
        {
            vslam_stub_set_capacity_period(3);
            vslam::SequenceOptions o;
            o.width = w, o.height = h, o.batch_frames = 6, o.max_corners = K, o.hypotheses = 8, o.threshold = 10.f, o.seed = 0x77;
            const vslam::SequenceStats st = vslam::run_sequence(vpath, dir + "/cap.bin", o);
            if (st.batches_redone < 3) throw std::runtime_error("no batch was redone");
            verify(dir + "/cap.bin", o.seed, frames);
            const vslam::SequenceStats sd = vslam::run_sequence_devices(vpath, dir + "/capd.bin", o, {0, 0, 0});
            if (sd.batches_redone < 2) throw std::runtime_error("no batch was redone (devices)");
            verify(dir + "/capd.bin", o.seed, frames);
            vslam_stub_set_capacity_period(0);
        }
        // max_frames, and a file shorter than one pair
        vslam::SequenceOptions o;
        o.width = w, o.height = h, o.batch_frames = 5, o.max_corners = K, o.hypotheses = 8, o.threshold = 10.f, o.seed = 5, o.max_frames = 11;
        vslam::run_sequence(vpath, dir + "/m.bin", o);
        verify(dir + "/m.bin", o.seed, 11);
        if (truncate(vpath.c_str(), (off_t)fb) != 0) return 4;
        o.max_frames = 0;
        const vslam::SequenceStats one = vslam::run_sequence(vpath, dir + "/one.bin", o);
        if (one.pairs != 0) throw std::runtime_error("a one-frame file has no pairs");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
    std::printf("%d\n", checked);
    return 0;
}

// The reference's capture loop without the map and the viewer, on the device front-end:
// raw BGR24 frames in, one record per consecutive frame pair out, then the records read back.
//
//   g++ -std=c++17 -O2 examples/capture_loop.cpp -Iinclude -Lvslam_amd -lvslam_host -lvslam_amd \
//       -Wl,-rpath,$PWD/vslam_amd -o capture_loop
//   ffmpeg -i clip.mp4 -f rawvideo -pix_fmt bgr24 clip.bgr
//   ./capture_loop clip.bgr 1280 720 clip.rec            (one device)
//   ./capture_loop clip.bgr 1280 720 clip.rec 0 1 2 3    (the file's pairs sharded over these devices; same record file)
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <vector>

#include "vslam/Ingest.h"

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s frames.bgr width height out.rec [device ...]\n", argv[0]);
        return 2;
    }
    try {
        vslam::SequenceOptions o;
        o.width = std::atoi(argv[2]);
        o.height = std::atoi(argv[3]);
        o.batch_frames = 64;      // frames per device batch
        o.max_corners = 3000;     // goodFeaturesToTrack(..., 3000, 0.01, 3), src/Frame.cpp:61
        o.hypotheses = 100;       // RansacFilter rf(8, 100, 10), src/vslam.cpp:19
        o.threshold = 10.f;
        o.seed = 1;               // pair i draws its 8-subsets from seed ^ i
        std::vector<int> devices;
        for (int i = 5; i < argc; i++) devices.push_back(std::atoi(argv[i]));
        const vslam::SequenceStats st = devices.empty() ? vslam::run_sequence(argv[1], argv[4], o)
                                                        : vslam::run_sequence_devices(argv[1], argv[4], o, devices);
        std::printf("%llu frames, %llu pairs in %.3f s\n", (unsigned long long)st.frames, (unsigned long long)st.pairs, st.seconds);

        vslam::RecordReader rd(argv[4]);
        vslam::PairRecord r;
        while (rd.next(r))
            std::printf("pair %llu: hypothesis %d, %d inliers, %zu matches, F[8] = %g\n", (unsigned long long)r.first_frame,
                        r.winner, r.inliers, r.matches.size(), (double)r.F[8]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

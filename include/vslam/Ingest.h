// Video ingest and result records for the device front-end.
//
// The reference reads frames with cv::VideoCapture into a host cv::Mat, one per loop iteration, and keeps
// its results in memory for the viewer (reference: src/vslam.cpp:24-26,54-77,286).  A device front-end wants
// the opposite shape: frames arrive in batches through page-locked buffers while the previous batch is being
// processed, every frame is extracted once and matched against its predecessor, and what leaves the GPU is
// one small record per frame pair.  This header is that host layer; it sits on the C ABI
// (include/vslam_amd.h: vslam_host_alloc, vslam_upload_async/fence/wait, vslam_frontend_sequence).
//
// Input: raw BGR24 frames back to back (width * height * 3 bytes each), e.g. the output of
//   ffmpeg -i clip.mp4 -f rawvideo -pix_fmt bgr24 clip.bgr
// (decoding itself stays outside: this image has no codec library, and cv::VideoCapture hands the reference
// exactly such BGR frames).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace vslam {

// One record per pair (frame i, frame i + 1): what match_features leaves behind (src/Frame.cpp:82-105) plus the
// bookkeeping find_fundamental keeps (src/RansacFilter.cpp:36-67).
struct PairRecord {
    uint64_t first_frame = 0;      // i
    int32_t winner = -1;           // index of the accepted hypothesis, -1 if none was accepted
    int32_t inliers = 0;           // its inlier count
    float score = 0.f;             // its residual sum
    float F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<std::pair<int32_t, int32_t>> matches;   // inlier matches: keypoint index in frame i, in frame i + 1
};

// File layout (little endian, no padding):
//   header  : "VSLAMREC", u32 version (1), u32 width, u32 height, u32 max_corners, u32 hypotheses,
//             f32 threshold, u32 seed, u32 reserved (0)                                          = 40 bytes
//   records : u64 first_frame, i32 winner, i32 inliers, f32 score, f32 F[9], u32 n, n x (i32, i32)
struct RecordHeader {
    uint32_t version = 1, width = 0, height = 0, max_corners = 0, hypotheses = 0;
    float threshold = 0.f;
    uint32_t seed = 0;
};

class RecordWriter {
public:
    RecordWriter(const std::string &path, const RecordHeader &header);
    ~RecordWriter();
    RecordWriter(const RecordWriter &) = delete;
    RecordWriter &operator=(const RecordWriter &) = delete;
    void append(const PairRecord &r);
    void close();

private:
    void *file_ = nullptr;
};

class RecordReader {
public:
    explicit RecordReader(const std::string &path);
    ~RecordReader();
    RecordReader(const RecordReader &) = delete;
    RecordReader &operator=(const RecordReader &) = delete;
    const RecordHeader &header() const { return header_; }
    bool next(PairRecord &r);   // false at end of file; throws on a truncated record

private:
    void *file_ = nullptr;
    RecordHeader header_;
};

struct SequenceOptions {
    int width = 0, height = 0;
    int batch_frames = 64;       // frames per device batch (>= 2); consecutive batches share one frame
    int max_corners = 3000;      // src/Frame.cpp:61
    int hypotheses = 100;        // RansacFilter rf(8, 100, 10), src/vslam.cpp:19
    float threshold = 10.f;
    uint32_t seed = 0;           // pair i draws its sets from seed ^ i (the reference seeds from random_device)
    uint64_t max_frames = 0;     // stop after this many frames (0 = whole file)
    int reader_threads = 12;     // threads that read a batch's frames from the file in parallel (pread)
};

struct SequenceStats {
    uint64_t frames = 0, pairs = 0, batches = 0;
    double seconds = 0;          // wall time of the loop, file reads and uploads included
    // Batches in which more frames overflowed the corner detector's bounded candidate lists (response plateaus, pure noise:
    // not image data) than its whole-image fallback pool holds (max(4, frames / 16) sets).  The device reports that as
    // VSLAM_ERR_CAPACITY when the batch is waited for; the loop then repeats the batch with every list sized for the whole
    // image (VSLAM_OPT_CORNER_LIST_CAP = -1), so the records are exact either way -- such a batch costs twice its time.
    uint64_t batches_redone = 0;
};

// The reference's capture loop without the map and the viewer: read `video_path`, run the front-end on every
// consecutive frame pair, write one record per pair to `record_path`.  The result does not depend on
// batch_frames.  Throws std::runtime_error on I/O or device errors.
SequenceStats run_sequence(const std::string &video_path, const std::string &record_path, const SequenceOptions &options);

// The same file over several devices of this process: entry r of `devices` (a device may repeat: several contexts on one
// device) gets its own context, host thread and reader, and the pairs [lo, hi) of vslam_shard_range's split, i.e. frames
// [lo, hi] -- the frame between two neighbouring slices is read and extracted by both, nothing else is shared and no device
// talks to another.  Pair i keeps its global number and its seed (seed ^ i), so the record file is byte for byte the one
// run_sequence writes.  Records wait in host memory until every slice is done (about 60 + 8 * inliers bytes per pair).
// `video_path` must be a regular file (slices are read at offsets).  stats.seconds is the wall time of the whole call.
// The slot contexts are kept for the next call (as the process-wide one of run_sequence is); calls are serialised.
SequenceStats run_sequence_devices(const std::string &video_path, const std::string &record_path, const SequenceOptions &options,
                                   const std::vector<int> &devices);

}  // namespace vslam

// C entry point for hosts without C++ (ctypes): returns 0 on success, -1 with a message in err on failure.
extern "C" int vslam_host_run_sequence(const char *video_path, const char *record_path, int width, int height,
                                       int batch_frames, int max_corners, int hypotheses, float threshold,
                                       uint32_t seed, uint64_t max_frames, uint64_t *frames_out, uint64_t *pairs_out,
                                       double *seconds_out, char *err, int err_cap);
extern "C" int vslam_host_run_sequence_devices(const char *video_path, const char *record_path, int width, int height,
                                               int batch_frames, int max_corners, int hypotheses, float threshold,
                                               uint32_t seed, uint64_t max_frames, const int *devices, int n_devices,
                                               uint64_t *frames_out, uint64_t *pairs_out, double *seconds_out, char *err,
                                               int err_cap);
// SequenceStats::batches_redone of the last call through one of the two entry points above
extern "C" uint64_t vslam_host_last_batches_redone(void);

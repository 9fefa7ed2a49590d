// Frame ingest + result records on top of the C ABI (include/vslam/Ingest.h).
//
// run_sequence keeps three things busy at once: a reader (a few threads, each pread()ing its share of the batch's
// frames: one thread copies 5-9 GB/s out of the page cache, the upload takes 57 GB/s) fills one page-locked buffer
// from the file while the copy stream uploads the other and the compute stream works on the batch before it.  A host
// buffer goes back to the reader as soon as its upload is done, so reads and uploads run side by side all the time.
// Batch k holds frames [k * (B - 1), k * (B - 1) + B): consecutive batches share one frame, so every
// consecutive pair is computed exactly once and no feature has to survive a batch (re-extracting the shared
// frame costs 1 / B of the extraction).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vslam/Ingest.h"
#include "host_internal.h"

namespace vslam {
namespace {

void put(std::FILE *f, const void *p, size_t n) {
    if (std::fwrite(p, 1, n, f) != n) throw std::runtime_error("vslam records: write failed");
}
bool get(std::FILE *f, void *p, size_t n, bool eof_ok) {
    const size_t r = std::fread(p, 1, n, f);
    if (r == n) return true;
    if (r == 0 && eof_ok) return false;
    throw std::runtime_error("vslam records: truncated file");
}
const char kMagic[8] = {'V', 'S', 'L', 'A', 'M', 'R', 'E', 'C'};

// a failed call on `ctx` -> std::runtime_error carrying that context's own message (detail::check reads the process-wide one)
void check(vslam_ctx *ctx, int rc, const char *what) {
    if (rc != VSLAM_OK)
        throw std::runtime_error(std::string("vslam_amd: ") + what + " failed (rc=" + std::to_string(rc) + "): " + vslam_last_error(ctx));
}

struct Pinned {   // page-locked host buffer
    vslam_ctx *ctx;
    uint8_t *p = nullptr;
    Pinned(vslam_ctx *c, size_t bytes) : ctx(c) { check(ctx, vslam_host_alloc(ctx, bytes, reinterpret_cast<void **>(&p)), "host_alloc"); }
    ~Pinned() { (void)vslam_host_free(ctx, p); }
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
};
template <typename T>
struct Dev {
    vslam_ctx *ctx;
    T *p = nullptr;
    Dev(vslam_ctx *c, size_t count) : ctx(c) { check(ctx, vslam_dev_alloc(ctx, sizeof(T) * (count ? count : 1), reinterpret_cast<void **>(&p)), "dev_alloc"); }
    ~Dev() { (void)vslam_dev_free(ctx, p); }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
};

}  // namespace

// ------------------------------------------------------------------------------------------ records
RecordWriter::RecordWriter(const std::string &path, const RecordHeader &h) {
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("vslam records: cannot create " + path);
    file_ = f;
    const uint32_t words[5] = {h.version, h.width, h.height, h.max_corners, h.hypotheses};
    const uint32_t tail[2] = {h.seed, 0u};
    put(f, kMagic, 8);
    put(f, words, sizeof words);
    put(f, &h.threshold, 4);
    put(f, tail, sizeof tail);
}
RecordWriter::~RecordWriter() {
    if (file_) std::fclose(static_cast<std::FILE *>(file_));
}
void RecordWriter::close() {
    if (file_ && std::fclose(static_cast<std::FILE *>(file_)) != 0) {
        file_ = nullptr;
        throw std::runtime_error("vslam records: close failed");
    }
    file_ = nullptr;
}
void RecordWriter::append(const PairRecord &r) {
    std::FILE *f = static_cast<std::FILE *>(file_);
    if (!f) throw std::runtime_error("vslam records: writer is closed");
    const uint32_t n = (uint32_t)r.matches.size();
    put(f, &r.first_frame, 8);
    put(f, &r.winner, 4);
    put(f, &r.inliers, 4);
    put(f, &r.score, 4);
    put(f, r.F, 36);
    put(f, &n, 4);
    std::vector<int32_t> flat(2 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        flat[2 * i] = r.matches[i].first;
        flat[2 * i + 1] = r.matches[i].second;
    }
    if (n) put(f, flat.data(), 8 * (size_t)n);
}

RecordReader::RecordReader(const std::string &path) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("vslam records: cannot open " + path);
    file_ = f;
    char magic[8];
    uint32_t words[5], tail[2];
    get(f, magic, 8, false);
    if (std::memcmp(magic, kMagic, 8) != 0) throw std::runtime_error("vslam records: not a record file: " + path);
    get(f, words, sizeof words, false);
    get(f, &header_.threshold, 4, false);
    get(f, tail, sizeof tail, false);
    header_.version = words[0];
    header_.width = words[1];
    header_.height = words[2];
    header_.max_corners = words[3];
    header_.hypotheses = words[4];
    header_.seed = tail[0];
    if (header_.version != 1) throw std::runtime_error("vslam records: unknown version");
}
RecordReader::~RecordReader() {
    if (file_) std::fclose(static_cast<std::FILE *>(file_));
}
bool RecordReader::next(PairRecord &r) {
    std::FILE *f = static_cast<std::FILE *>(file_);
    if (!get(f, &r.first_frame, 8, true)) return false;
    uint32_t n = 0;
    get(f, &r.winner, 4, false);
    get(f, &r.inliers, 4, false);
    get(f, &r.score, 4, false);
    get(f, r.F, 36, false);
    get(f, &n, 4, false);
    if (n > header_.max_corners) throw std::runtime_error("vslam records: corrupt record (more matches than max_corners)");
    std::vector<int32_t> flat(2 * (size_t)n);
    if (n) get(f, flat.data(), 8 * (size_t)n, false);
    r.matches.resize(n);
    for (uint32_t i = 0; i < n; i++) r.matches[i] = {flat[2 * i], flat[2 * i + 1]};
    return true;
}

// ------------------------------------------------------------------------------------------ the capture loop
namespace {

struct Input {   // the opened video
    int fd = -1;
    bool regular = false;          // a regular file is read at offsets by several threads; anything else is streamed
    uint64_t frames = UINT64_MAX;  // whole frames to process (UINT64_MAX: a stream, until it ends)
    Input(const std::string &path, const SequenceOptions &o, const char *who) {
        if (o.width <= 0 || o.height <= 0 || o.batch_frames < 2 || o.max_corners <= 0 || o.hypotheses <= 0)
            throw std::invalid_argument(std::string(who) + ": bad options");
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error(std::string(who) + ": cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw std::runtime_error(std::string(who) + ": cannot stat " + path);
        }
        // A FIFO, pipe or /dev/stdin reports st_size 0 and cannot be read at offsets: it is streamed with read() by the
        // one reader thread until it ends.
        regular = S_ISREG(st.st_mode);
        const size_t frame_bytes = (size_t)o.width * o.height * 3;
        if (regular) frames = (uint64_t)st.st_size / frame_bytes;   // a trailing partial frame is dropped
        if (o.max_frames && frames > o.max_frames) frames = o.max_frames;
    }
    ~Input() { ::close(fd); }
    Input(const Input &) = delete;
    Input &operator=(const Input &) = delete;
};

RecordHeader header_of(const SequenceOptions &o) {
    RecordHeader head;
    head.width = (uint32_t)o.width;
    head.height = (uint32_t)o.height;
    head.max_corners = (uint32_t)o.max_corners;
    head.hypotheses = (uint32_t)o.hypotheses;
    head.threshold = o.threshold;
    head.seed = o.seed;
    return head;
}

// Frames [first_frame, first_frame + file_frames) of `in` on `ctx`: every consecutive pair inside the range goes to `sink`
// in order.  Pair i keeps its global number (record.first_frame, and its seed o.seed ^ i), so a range computes exactly the
// records the whole file would hold for those pairs.  A stream (not regular) starts at frame 0.
template <typename Sink>
SequenceStats run_range(vslam_ctx *ctx, int in, bool regular, uint64_t first_frame, uint64_t file_frames,
                        const SequenceOptions &o, Sink &&sink) {
    const int B = o.batch_frames, K = o.max_corners;
    const size_t frame_bytes = (size_t)o.width * o.height * 3;
    int readers = o.reader_threads;
    if (const char *e = std::getenv("VSLAM_READER_THREADS")) readers = std::atoi(e);   // tuning
    readers = readers < 1 ? 1 : (readers > 16 ? 16 : readers);
    Pinned hbuf0(ctx, frame_bytes * B), hbuf1(ctx, frame_bytes * B);
    uint8_t *hbuf[2] = {hbuf0.p, hbuf1.p};
    Dev<uint8_t> dbuf0(ctx, frame_bytes * B), dbuf1(ctx, frame_bytes * B);
    uint8_t *dbuf[2] = {dbuf0.p, dbuf1.p};
    Dev<float> d_xy(ctx, 2 * (size_t)B * K), d_F(ctx, 9 * (size_t)B);
    Dev<uint8_t> d_desc(ctx, 32 * (size_t)B * K);
    Dev<int32_t> d_nodes(ctx, (size_t)B * K), d_n(ctx, B), d_matches(ctx, 2 * (size_t)B * K), d_best(ctx, 4 * (size_t)B);
    Dev<uint32_t> d_seeds(ctx, B);
    Dev<int8_t> d_pat(ctx, 1024);
    check(ctx, vslam_copy_h2d(ctx, d_pat.p, detail::brief_pattern().data(), 1024), "pattern upload");
    vslam_extract_params params;
    detail::fill_extract_params(params, K, d_pat.p);

    // ---- reader thread: batch k -> hbuf[k & 1]; state[b]: 0 free, 1 filled (count[b] frames), 2 end of stream
    std::mutex mu;
    std::condition_variable cv;
    int state[2] = {0, 0}, count[2] = {0, 0};
    bool stop = false;
    std::string reader_error;
    std::thread reader([&] {
        try {
            uint64_t read_total = 0;
            for (uint64_t k = 0;; k++) {
                const int b = (int)(k & 1);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return state[b] == 0 || stop; });
                    if (stop) return;
                }
                int have = 0;
                if (k > 0) {   // the frame this batch shares with the previous one
                    std::memcpy(hbuf[b], hbuf[b ^ 1] + frame_bytes * (size_t)(B - 1), frame_bytes);
                    have = 1;
                }
                {   // frames read_total .. of the file -> slots have .. of the buffer, shared out over the reader threads
                    const uint64_t left = file_frames - read_total;
                    int want = (int)std::min<uint64_t>((uint64_t)(B - have), left);
                    if (!regular) {   // sequential stream: whole frames until end of input
                        int j = 0;
                        for (; j < want; j++) {
                            uint8_t *dst = hbuf[b] + frame_bytes * (size_t)(have + j);
                            size_t got = 0;
                            while (got < frame_bytes) {
                                const ssize_t r = ::read(in, dst + got, frame_bytes - got);
                                if (r < 0 && errno == EINTR) continue;
                                if (r < 0) throw std::runtime_error("read failed");
                                if (r == 0) break;
                                got += (size_t)r;
                            }
                            if (got < frame_bytes) break;   // end of input (a trailing partial frame is dropped)
                        }
                        if (j < want) file_frames = read_total + (uint64_t)j;
                        want = j;
                    }
                    std::atomic<bool> failed{false};
                    auto share = [&](int t) {
                        for (int j = t; j < want; j += readers) {
                            uint8_t *dst = hbuf[b] + frame_bytes * (size_t)(have + j);
                            size_t got = 0;
                            const off_t at = (off_t)((first_frame + read_total + (uint64_t)j) * frame_bytes);
                            while (got < frame_bytes) {
                                const ssize_t r = ::pread(in, dst + got, frame_bytes - got, at + (off_t)got);
                                if (r <= 0) {
                                    failed = true;
                                    return;
                                }
                                got += (size_t)r;
                            }
                        }
                    };
                    if (regular) {
                        // joined on every path out of this block: a thread that fails to start (std::system_error) must
                        // reach the catch below, not std::terminate through ~thread of the ones already running
                        struct Pool {
                            std::vector<std::thread> threads;
                            ~Pool() {
                                for (auto &th : threads)
                                    if (th.joinable()) th.join();
                            }
                        } pool;
                        pool.threads.reserve((size_t)readers);
                        for (int t = 1; t < readers && t < want; t++) pool.threads.emplace_back(share, t);
                        share(0);
                    }
                    if (failed) throw std::runtime_error("read failed (file truncated while in use?)");
                    have += want;
                    read_total += (uint64_t)want;
                }
                const bool last = have < B;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    count[b] = have;
                    state[b] = (last && have < 2) ? 2 : 1;   // fewer than two frames: nothing left to pair
                    if (last && have >= 2) count[b] = -have;  // negative: final batch
                }
                cv.notify_all();
                if (last) return;
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            reader_error = e.what();
            state[0] = state[1] = 2;
            cv.notify_all();
        }
    });
    struct Joiner {
        std::thread &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~Joiner() {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
    } joiner{reader, mu, cv, stop};

    auto wait_filled = [&](int b, int &frames, bool &final_batch) -> bool {   // false: stream ended before this batch
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state[b] != 0; });
        if (!reader_error.empty()) throw std::runtime_error("run_sequence: reader: " + reader_error);
        if (state[b] == 2) return false;
        final_batch = count[b] < 0;
        frames = final_batch ? -count[b] : count[b];
        return true;
    };
    auto release = [&](int b) {
        {
            std::lock_guard<std::mutex> lk(mu);
            state[b] = 0;
        }
        cv.notify_all();
    };

    SequenceStats stats;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int32_t> h_best(4 * (size_t)B), h_matches(2 * (size_t)B * K);
    std::vector<float> h_F(9 * (size_t)B);
    std::vector<uint32_t> h_seeds(B);

    int frames = 0, next_frames = 0;
    bool final_batch = false, next_final = false;
    bool have = wait_filled(0, frames, final_batch);
    if (have) {
        check(ctx, vslam_upload_async(ctx, dbuf[0], hbuf[0], frame_bytes * (size_t)frames), "upload");
        check(ctx, vslam_upload_fence(ctx), "upload fence");
        check(ctx, vslam_upload_wait(ctx), "upload wait");
        release(0);
    }
    const bool trace = std::getenv("VSLAM_INGEST_TRACE") != nullptr;   // per-batch host timeline on stderr (milliseconds)
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    for (uint64_t k = 0; have; k++) {
        const auto t_iter = std::chrono::steady_clock::now();
        double t_enq = 0, t_filled = 0, t_down = 0, t_rec = 0;
        const int b = (int)(k & 1);
        const uint64_t first = first_frame + k * (uint64_t)(B - 1);   // global index of this batch's first frame = of its first pair
        const int pairs = frames - 1;
        for (int i = 0; i < pairs; i++) h_seeds[i] = o.seed ^ (uint32_t)(first + (uint64_t)i);
        check(ctx, vslam_copy_h2d(ctx, d_seeds.p, h_seeds.data(), sizeof(uint32_t) * (size_t)pairs), "seed upload");
        auto enqueue_batch = [&] {
            check(ctx, vslam_frontend_sequence(ctx, dbuf[b], frames, o.width, o.height, 3 * o.width, &params, K, d_seeds.p,
                                                  o.hypotheses, o.threshold, d_xy.p, d_desc.p, d_nodes.p, d_n.p, d_matches.p,
                                                  d_best.p, d_F.p),
                          "frontend_sequence");
        };
        enqueue_batch();
        t_enq = ms_since(t_iter);
        // the next batch goes up while this one is computed
        bool have_next = false;
        if (!final_batch) {
            have_next = wait_filled(b ^ 1, next_frames, next_final);
            if (have_next) check(ctx, vslam_upload_async(ctx, dbuf[b ^ 1], hbuf[b ^ 1], frame_bytes * (size_t)next_frames), "upload");
        }
        t_filled = ms_since(t_iter);
        {
            // The batch's status.  The one failure a well-formed call can meet is VSLAM_ERR_CAPACITY: more frames of this batch
            // overflowed the corner detector's bounded lists (plateaus, pure noise) than its whole-image fallback pool holds;
            // those frames came back without corners.  The frames are still on the device: do the batch again with every list
            // sized for the whole image (nothing can overflow; 16 bytes per pixel and frame of workspace, for this batch
            // shape from now on) -- the records are then what an unbounded run gives, and the file goes on.
            const int rc = vslam_ctx_synchronize(ctx);
            if (rc == VSLAM_ERR_CAPACITY) {
                check(ctx, vslam_ctx_set_option(ctx, VSLAM_OPT_CORNER_LIST_CAP, -1), "set_option");
                enqueue_batch();
                const int rc2 = vslam_ctx_synchronize(ctx);
                (void)vslam_ctx_set_option(ctx, VSLAM_OPT_CORNER_LIST_CAP, 0);
                check(ctx, rc2, "frontend_sequence (whole-image corner lists)");
                stats.batches_redone++;
            } else {
                check(ctx, rc, "frontend_sequence");
            }
        }
        check(ctx, vslam_copy_d2h(ctx, h_best.data(), d_best.p, sizeof(int32_t) * 4 * (size_t)pairs), "download");
        check(ctx, vslam_copy_d2h(ctx, h_F.data(), d_F.p, sizeof(float) * 9 * (size_t)pairs), "download");
        check(ctx, vslam_copy_d2h(ctx, h_matches.data(), d_matches.p, sizeof(int32_t) * 2 * (size_t)pairs * K), "download");
        t_down = ms_since(t_iter);
        for (int i = 0; i < pairs; i++) {
            PairRecord r;
            r.first_frame = first + (uint64_t)i;
            r.winner = h_best[4 * i + 0];
            r.inliers = h_best[4 * i + 1];
            std::memcpy(&r.score, &h_best[4 * i + 2], 4);
            // nothing accepted (fewer than 8 matches, e.g. a blank frame, or every sum NaN): the device leaves F
            // untouched, as find_fundamental leaves `fundamental` (src/RansacFilter.cpp:59-65); the record keeps zeros
            if (r.winner >= 0) std::memcpy(r.F, &h_F[9 * (size_t)i], 36);
            const int n = h_best[4 * i + 3];
            r.matches.resize((size_t)n);
            for (int j = 0; j < n; j++) r.matches[(size_t)j] = {h_matches[2 * ((size_t)i * K + j)], h_matches[2 * ((size_t)i * K + j) + 1]};
            sink(std::move(r));
        }
        stats.pairs += (uint64_t)pairs;
        stats.frames = first - first_frame + (uint64_t)frames;
        stats.batches++;
        t_rec = ms_since(t_iter);
        if (have_next) {
            check(ctx, vslam_upload_wait(ctx), "upload wait");   // hbuf[b ^ 1] is on the device
            check(ctx, vslam_upload_fence(ctx), "upload fence");
            // A host buffer is free again the moment its upload is done, not when its batch has been computed: the reader
            // fills hbuf[b] with batch k + 2 while hbuf[b ^ 1] goes up, so file reads and uploads overlap each other as well
            // as the kernels.  (Only the reader writes the buffers: the frame batch k + 2 shares with k + 1 is still in
            // hbuf[b ^ 1] when it copies it, whatever the state of that buffer.)
            release(b ^ 1);
        }
        if (trace)
            std::fprintf(stderr, "batch %llu: enqueued %.2f, next batch filled + upload issued %.2f, results down %.2f, records %.2f, upload done %.2f\n",
                         (unsigned long long)k, t_enq, t_filled, t_down, t_rec, ms_since(t_iter));
        have = have_next;
        frames = next_frames;
        final_batch = next_final;
    }
    stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return stats;
}

}  // namespace

SequenceStats run_sequence(const std::string &video_path, const std::string &record_path, const SequenceOptions &o) {
    Input input(video_path, o, "run_sequence");
    RecordWriter writer(record_path, header_of(o));
    SequenceStats stats = run_range(detail::context(), input.fd, input.regular, 0, input.frames, o,
                                    [&](PairRecord &&r) { writer.append(r); });
    writer.close();
    return stats;
}

namespace {
// Contexts of run_sequence_devices, kept for the next call like the process-wide one (a context owns its streams and its
// workspace arena: creating one and growing the arena again costs about 0.1 s per slot).  Entry (device, n) is the n-th
// slot a call places on `device`; a slot thread has it to itself for the length of the call (g_slot_mu is held by the call).
std::mutex g_slot_mu;
std::vector<std::pair<std::pair<int, int>, vslam_ctx *>> g_slot_ctx;
vslam_ctx *slot_context(int device, int nth) {   // on the thread that will use the context; the caller serialises
    for (auto &e : g_slot_ctx)
        if (e.first == std::make_pair(device, nth)) {
            check(e.second, vslam_ctx_make_current(e.second), "ctx_make_current");
            return e.second;
        }
    vslam_ctx *ctx = nullptr;
    if (vslam_ctx_create(device, &ctx) != VSLAM_OK || !ctx)
        throw std::runtime_error("cannot create a context on device " + std::to_string(device));
    g_slot_ctx.push_back({{device, nth}, ctx});
    return ctx;
}
}  // namespace

SequenceStats run_sequence_devices(const std::string &video_path, const std::string &record_path, const SequenceOptions &o,
                                   const std::vector<int> &devices) {
    if (devices.empty()) throw std::invalid_argument("run_sequence_devices: no devices");
    Input input(video_path, o, "run_sequence_devices");
    if (!input.regular) throw std::invalid_argument("run_sequence_devices: the input must be a regular file (a stream cannot be read at offsets)");
    const auto t0 = std::chrono::steady_clock::now();
    const int slots = (int)devices.size();
    (void)detail::brief_pattern();   // resolved (settings / VSLAM_BRIEF_PATTERN / learned table) once, here: the slot threads only read it
    const int64_t pairs = input.frames >= 2 ? (int64_t)input.frames - 1 : 0;
    struct Slot {
        std::vector<PairRecord> records;
        SequenceStats stats;
        std::string error;
    };
    std::vector<Slot> slot((size_t)slots);
    std::lock_guard<std::mutex> one_call(g_slot_mu);   // the kept contexts belong to one call at a time
    {
        struct Pool {   // joined on every path out, as in the reader above
            std::vector<std::thread> threads;
            ~Pool() {
                for (auto &th : threads)
                    if (th.joinable()) th.join();
            }
        } pool;
        pool.threads.reserve((size_t)slots);
        std::mutex table_mu;   // g_slot_ctx itself, between this call's threads
        for (int r = 0; r < slots; r++) {
            // vslam_shard_range's rule in 64 bits: the first pairs % slots slots get one pair more
            const int64_t base = pairs / slots, extra = pairs % slots;
            const int64_t lo = base * r + std::min<int64_t>(r, extra), hi = lo + base + (r < extra ? 1 : 0);
            if (hi <= lo) continue;   // more slots than pairs
            int nth = 0;
            for (int q = 0; q < r; q++) nth += devices[(size_t)q] == devices[(size_t)r];
            pool.threads.emplace_back([&, r, lo, hi, nth] {
                Slot &me = slot[(size_t)r];
                try {
                    vslam_ctx *ctx = nullptr;
                    {   // makes devices[r] this thread's current device; everything below is allocated there
                        std::lock_guard<std::mutex> lk(table_mu);
                        ctx = slot_context(devices[(size_t)r], nth);
                    }
                    me.records.reserve((size_t)(hi - lo));
                    // pairs [lo, hi) need frames [lo, hi]: neighbouring slots both read (and extract) the frame between them
                    me.stats = run_range(ctx, input.fd, true, (uint64_t)lo, (uint64_t)(hi - lo) + 1, o,
                                         [&](PairRecord &&rec) { me.records.push_back(std::move(rec)); });
                } catch (const std::exception &e) {
                    me.error = std::string("slot ") + std::to_string(r) + ": " + e.what();
                }
            });
        }
    }
    for (const Slot &s : slot)
        if (!s.error.empty()) throw std::runtime_error("run_sequence_devices: " + s.error);
    RecordWriter writer(record_path, header_of(o));
    SequenceStats stats;
    for (const Slot &s : slot) {
        for (const PairRecord &r : s.records) writer.append(r);
        stats.pairs += s.stats.pairs;
        stats.batches += s.stats.batches;
        stats.batches_redone += s.stats.batches_redone;
    }
    writer.close();
    stats.frames = stats.pairs ? stats.pairs + 1 : std::min<uint64_t>(input.frames, 1);
    stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return stats;
}

}  // namespace vslam

namespace {
std::atomic<uint64_t> g_last_redone{0};   // SequenceStats::batches_redone of the last call through a C entry point
}
extern "C" uint64_t vslam_host_last_batches_redone(void) { return g_last_redone.load(); }

extern "C" int vslam_host_run_sequence(const char *video_path, const char *record_path, int width, int height,
                                       int batch_frames, int max_corners, int hypotheses, float threshold,
                                       uint32_t seed, uint64_t max_frames, uint64_t *frames_out, uint64_t *pairs_out,
                                       double *seconds_out, char *err, int err_cap) {
    try {
        vslam::SequenceOptions o;
        o.width = width;
        o.height = height;
        o.batch_frames = batch_frames;
        o.max_corners = max_corners;
        o.hypotheses = hypotheses;
        o.threshold = threshold;
        o.seed = seed;
        o.max_frames = max_frames;
        const vslam::SequenceStats s = vslam::run_sequence(video_path ? video_path : "", record_path ? record_path : "", o);
        g_last_redone = s.batches_redone;
        if (frames_out) *frames_out = s.frames;
        if (pairs_out) *pairs_out = s.pairs;
        if (seconds_out) *seconds_out = s.seconds;
        return 0;
    } catch (const std::exception &e) {
        if (err && err_cap > 0) {
            std::strncpy(err, e.what(), (size_t)err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
}

extern "C" int vslam_host_run_sequence_devices(const char *video_path, const char *record_path, int width, int height,
                                               int batch_frames, int max_corners, int hypotheses, float threshold,
                                               uint32_t seed, uint64_t max_frames, const int *devices, int n_devices,
                                               uint64_t *frames_out, uint64_t *pairs_out, double *seconds_out, char *err,
                                               int err_cap) {
    try {
        if (!devices || n_devices <= 0) throw std::invalid_argument("run_sequence_devices: no devices");
        vslam::SequenceOptions o;
        o.width = width;
        o.height = height;
        o.batch_frames = batch_frames;
        o.max_corners = max_corners;
        o.hypotheses = hypotheses;
        o.threshold = threshold;
        o.seed = seed;
        o.max_frames = max_frames;
        const vslam::SequenceStats s = vslam::run_sequence_devices(video_path ? video_path : "", record_path ? record_path : "", o,
                                                                   std::vector<int>(devices, devices + n_devices));
        g_last_redone = s.batches_redone;
        if (frames_out) *frames_out = s.frames;
        if (pairs_out) *pairs_out = s.pairs;
        if (seconds_out) *seconds_out = s.seconds;
        return 0;
    } catch (const std::exception &e) {
        if (err && err_cap > 0) {
            std::strncpy(err, e.what(), (size_t)err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
}

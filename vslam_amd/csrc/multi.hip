// Several devices behind the C ABI (north_star: "a batch of independent frame-pairs shards embarrassingly
// across the 8 GPUs of one node with RCCL over xGMI only for the final match/pose gather").
//
//  * One process that owns N devices: vslam_multi_* -- one context and one host thread per device, contiguous slices of
//    the batch (vslam_shard_range), per-pair seeds base ^ GLOBAL pair index, every slice's result records written
//    straight into the caller's host array at the slice's offset.  Nothing crosses between devices.
//  * One process per device (torch.distributed-style launch, or any launcher): each rank runs its slice through
//    vslam_frontend_pairs + vslam_pack_records on its own context and the fixed-size records are exchanged once with
//    vslam_gather_records = ncclAllGather on the context's stream (RCCL, loaded on first use: the library has no link-time
//    dependency on it, and a process that never gathers never loads it).
// Host code only; no kernel lives here.
#include "ctx.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <thread>

// The few RCCL (= NCCL API) types and prototypes this file uses, declared here so that the library builds where the RCCL
// headers are not installed; the entry points themselves are looked up in librccl.so at run time (a missing library is
// VSLAM_ERR_COMM).  These are the stable public ABI of nccl.h / rccl.h: a 128-byte unique id, an opaque communicator
// pointer, int-sized enums with ncclSuccess = 0 and ncclInt32 = 2.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2 } ncclDataType_t;
typedef ncclResult_t (*vs_ncclGetUniqueId_t)(ncclUniqueId *);
typedef ncclResult_t (*vs_ncclCommInitRank_t)(ncclComm_t *, int, ncclUniqueId, int);
typedef ncclResult_t (*vs_ncclCommDestroy_t)(ncclComm_t);
typedef ncclResult_t (*vs_ncclCommCount_t)(const ncclComm_t, int *);
typedef ncclResult_t (*vs_ncclCommUserRank_t)(const ncclComm_t, int *);
typedef ncclResult_t (*vs_ncclAllGather_t)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*vs_ncclSend_t)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*vs_ncclRecv_t)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*vs_ncclGroup_t)(void);
typedef const char *(*vs_ncclGetErrorString_t)(ncclResult_t);
}

extern "C" int vslam_shard_range(int items, int rank, int world, int *lo, int *hi) {
    if (items < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return VSLAM_ERR_INVALID;
    const int base = items / world, extra = items % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------ one process, N devices
namespace {
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
struct Member {
    int device = 0;
    vslam_ctx *ctx = nullptr;
    DevBuf bgr, seeds, xy, desc, nodes, n, matches, best, F, rec, pattern;
    std::vector<uint32_t> h_seeds;   // lives here: an upload may still be reading it when the worker returns on an error path
    int rc = VSLAM_OK;
    std::string err;
};
int grow(Member &m, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes) return VSLAM_OK;
    if (b.p) {
        int rc = vslam_dev_free(m.ctx, b.p);
        if (rc) return rc;
        b.p = nullptr;
        b.bytes = 0;
    }
    int rc = vslam_dev_alloc(m.ctx, bytes, &b.p);
    if (rc == VSLAM_OK) b.bytes = bytes;
    return rc;
}

// What one call asks of every slot.  Frames come either from host memory (h_last / h_cur: the whole batch, pair order) or
// are already on the slots' devices (d_bgr[r]: slot r's slice in vslam_frontend_pairs' layout).
struct MultiJob {
    const uint8_t *h_last = nullptr, *h_cur = nullptr;
    const uint8_t *const *d_bgr = nullptr;
    int pairs = 0, width = 0, height = 0, row_stride = 0, kp_stride = 0, hyp = 0;
    const vslam_extract_params *params = nullptr;
    const int8_t *h_pattern = nullptr;
    uint32_t base_seed = 0;
    float threshold = 0;
    int32_t *h_records = nullptr, *h_n_keypoints = nullptr;
};

// Host frames reach the device in chunks of this many pairs: chunk k + 1 is uploaded (copy stream) while chunk k is
// computed, so a slice costs max(upload, compute) + one chunk instead of their sum.  A pair's result does not depend on
// its batch (tests/test_gpu_batch_properties.py), so the chunking cannot be seen in the records.
constexpr int kChunkPairs = 64;
}  // namespace

struct vslam_multi {
    std::vector<Member> members;
    std::string err;
    std::mutex mu;   // one batch at a time per object
};

namespace {
void slot_work_body(std::vector<Member> &members, const MultiJob &job, int r) {
    Member &mb = members[(size_t)r];
    const int world = (int)members.size();
    int lo = 0, hi = 0;
    vslam_shard_range(job.pairs, r, world, &lo, &hi);
    const int ps = hi - lo;
    if (ps == 0) return;
    auto fail = [&](int rc, const char *what) {
        // every exit path: nothing this call queued may still be reading host memory the caller is about to reuse
        (void)vslam_upload_wait(mb.ctx);
        (void)vslam_ctx_wait(mb.ctx);
        mb.rc = rc;
        mb.err = std::string(what) + ": " + vslam_last_error(mb.ctx);
    };
    if (hipSetDevice(mb.device) != hipSuccess) {
        mb.rc = VSLAM_ERR_HIP;
        mb.err = "hipSetDevice failed";
        return;
    }
    int rc;
    const size_t P = (size_t)ps, K = (size_t)job.kp_stride, words = 13 + K;
    const size_t frame_bytes = (size_t)job.height * job.row_stride;
    const bool resident = job.d_bgr != nullptr;
    const int chunk = resident ? ps : (ps < kChunkPairs ? ps : kChunkPairs);   // pairs per vslam_frontend_pairs call
    const size_t C = (size_t)chunk;
    // two frame buffers of one chunk each when the frames come from the host; outputs for the whole slice
    if ((!resident && (rc = grow(mb, mb.bgr, 2 * (2 * C * frame_bytes)))) || (rc = grow(mb, mb.seeds, 4 * P)) ||
        (rc = grow(mb, mb.xy, 8 * 2 * C * K)) || (rc = grow(mb, mb.desc, 32 * 2 * C * K)) || (rc = grow(mb, mb.nodes, 4 * 2 * C * K)) ||
        (rc = grow(mb, mb.n, 4 * 2 * P)) || (rc = grow(mb, mb.matches, 8 * C * K)) || (rc = grow(mb, mb.best, 16 * C)) ||
        (rc = grow(mb, mb.F, 36 * C)) || (rc = grow(mb, mb.rec, 4 * P * words)) || (rc = grow(mb, mb.pattern, 1024)))
        return fail(rc, "device buffers");
    mb.h_seeds.resize(P);
    for (int i = 0; i < ps; i++) mb.h_seeds[(size_t)i] = job.base_seed ^ (uint32_t)(lo + i);   // GLOBAL pair index: results do not depend on the split
    if ((rc = vslam_upload_async(mb.ctx, mb.seeds.p, mb.h_seeds.data(), 4 * P))) return fail(rc, "seed upload");
    vslam_extract_params p = *job.params;
    if (job.h_pattern) {
        if ((rc = vslam_upload_async(mb.ctx, mb.pattern.p, job.h_pattern, 1024))) return fail(rc, "pattern upload");
        p.d_pattern = static_cast<const int8_t *>(mb.pattern.p);
    }
    uint8_t *buf[2] = {static_cast<uint8_t *>(mb.bgr.p), resident ? nullptr : static_cast<uint8_t *>(mb.bgr.p) + 2 * C * frame_bytes};
    // chunk c's frames: "last" ones first, "current" ones behind them (vslam_frontend_pairs' layout), on the copy stream
    auto upload = [&](int c0, int cn, uint8_t *dst) {
        int u = vslam_upload_async(mb.ctx, dst, job.h_last + (size_t)(lo + c0) * frame_bytes, (size_t)cn * frame_bytes);
        if (u == VSLAM_OK)
            u = vslam_upload_async(mb.ctx, dst + (size_t)cn * frame_bytes, job.h_cur + (size_t)(lo + c0) * frame_bytes, (size_t)cn * frame_bytes);
        return u;
    };
    if (!resident && (rc = upload(0, chunk < ps ? chunk : ps, buf[0]))) return fail(rc, "frame upload");
    int32_t *d_n = static_cast<int32_t *>(mb.n.p);
    int32_t *d_rec = static_cast<int32_t *>(mb.rec.p);
    for (int c0 = 0, k = 0; c0 < ps; c0 += chunk, k++) {
        const int cn = ps - c0 < chunk ? ps - c0 : chunk;
        const uint8_t *d_frames = resident ? job.d_bgr[r] : buf[k & 1];
        // the compute stream waits for everything uploaded so far (this chunk's frames; seeds and table the first time)
        if ((rc = vslam_upload_fence(mb.ctx))) return fail(rc, "upload fence");
        if (!resident && c0 + chunk < ps) {
            // the next chunk goes into the other buffer, whose last reader (chunk k - 1) must be done first: compute is
            // stream-ordered, so waiting for the compute stream here (the host is ahead of it by one chunk) is enough
            if (k >= 1 && (rc = vslam_ctx_wait(mb.ctx))) return fail(rc, "wait");
            const int nn = ps - (c0 + chunk) < chunk ? ps - (c0 + chunk) : chunk;
            if ((rc = upload(c0 + chunk, nn, buf[(k + 1) & 1]))) return fail(rc, "frame upload");
        }
        // per-frame counts of the chunk land in scratch laid out [last | current] for cn pairs; records go to their place
        int32_t *n_chunk = nullptr;
        if ((rc = vs_arena_get(mb.ctx, "multi.n_chunk", 4 * 2 * C, (void **)&n_chunk))) return fail(rc, "scratch");
        if ((rc = vslam_frontend_pairs(mb.ctx, d_frames, cn, job.width, job.height, job.row_stride, &p, job.kp_stride,
                                       static_cast<const uint32_t *>(mb.seeds.p) + c0, job.hyp, job.threshold, static_cast<float *>(mb.xy.p),
                                       static_cast<uint8_t *>(mb.desc.p), static_cast<int32_t *>(mb.nodes.p), n_chunk,
                                       static_cast<int32_t *>(mb.matches.p), static_cast<int32_t *>(mb.best.p), static_cast<float *>(mb.F.p))))
            return fail(rc, "frontend_pairs");
        if ((rc = vslam_pack_records(mb.ctx, static_cast<const float *>(mb.F.p), static_cast<const int32_t *>(mb.best.p),
                                     static_cast<const int32_t *>(mb.matches.p), cn, job.kp_stride, d_rec + (size_t)c0 * words)))
            return fail(rc, "pack_records");
        if (hipMemcpyAsync(d_n + c0, n_chunk, 4 * (size_t)cn, hipMemcpyDeviceToDevice, mb.ctx->stream) != hipSuccess ||
            hipMemcpyAsync(d_n + ps + c0, n_chunk + cn, 4 * (size_t)cn, hipMemcpyDeviceToDevice, mb.ctx->stream) != hipSuccess) {
            mb.ctx->err = "hipMemcpyAsync (keypoint counts)";
            return fail(VSLAM_ERR_HIP, "counts");
        }
    }
    // the "gather": this slice's records land at its offset of the caller's array
    if ((rc = vslam_copy_d2h(mb.ctx, job.h_records + (size_t)lo * words, mb.rec.p, 4 * P * words))) return fail(rc, "records download");
    if (job.h_n_keypoints) {
        if ((rc = vslam_copy_d2h(mb.ctx, job.h_n_keypoints + lo, d_n, 4 * P)) ||
            (rc = vslam_copy_d2h(mb.ctx, job.h_n_keypoints + job.pairs + lo, d_n + ps, 4 * P)))
            return fail(rc, "counts download");
    }
    if ((rc = vslam_upload_wait(mb.ctx)) || (rc = vslam_ctx_synchronize(mb.ctx))) return fail(rc, "synchronize");
}

// a worker thread must not let an exception escape (std::terminate would take the host process down)
void slot_work(std::vector<Member> &members, const MultiJob &job, int r) {
    Member &mb = members[(size_t)r];
    mb.rc = VSLAM_OK;
    mb.err.clear();
    try {
        slot_work_body(members, job, r);
    } catch (const std::exception &e) {
        (void)vslam_upload_wait(mb.ctx);
        (void)vslam_ctx_wait(mb.ctx);
        mb.rc = VSLAM_ERR_HIP;
        mb.err = std::string("exception in the slot's worker: ") + e.what();
    }
}

int run_job(vslam_multi *m, const MultiJob &job) {
    const int world = (int)m->members.size();
    int callers_device = -1;   // slice 0 runs on the calling thread: its current device is put back afterwards
    (void)hipGetDevice(&callers_device);
    struct DeviceGuard {
        int dev;
        ~DeviceGuard() {
            if (dev >= 0) (void)hipSetDevice(dev);
        }
    } device_guard{callers_device};
    {
        // joined on every path: a thread that cannot be started must not leave the running ones behind
        struct Pool {
            std::vector<std::thread> t;
            ~Pool() {
                for (auto &th : t)
                    if (th.joinable()) th.join();
            }
        } pool;
        pool.t.reserve((size_t)world);
        try {
            for (int r = 1; r < world; r++) pool.t.emplace_back([&, r] { slot_work(m->members, job, r); });
        } catch (const std::exception &e) {
            m->err = std::string("vslam_multi: cannot start a worker thread: ") + e.what();
        }
        if (m->err.empty()) slot_work(m->members, job, 0);
    }
    if (!m->err.empty()) return VSLAM_ERR_HIP;
    for (int r = 0; r < world; r++) {
        const Member &mb = m->members[(size_t)r];
        if (mb.rc) {
            m->err = "device slot " + std::to_string(r) + " (device " + std::to_string(mb.device) + "): " + mb.err;
            return mb.rc;
        }
    }
    return VSLAM_OK;
}
}  // namespace

extern "C" {

int vslam_multi_create(const int *devices, int n_devices, vslam_multi **out) {
    if (!devices || n_devices <= 0 || !out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    auto *m = new vslam_multi();
    m->members.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        m->members[(size_t)i].device = devices[i];
        const int rc = vslam_ctx_create(devices[i], &m->members[(size_t)i].ctx);
        if (rc) {
            for (int j = 0; j < i; j++) vslam_ctx_destroy(m->members[(size_t)j].ctx);
            delete m;
            return rc;
        }
    }
    *out = m;
    return VSLAM_OK;
}

int vslam_multi_destroy(vslam_multi *m) {
    if (!m) return VSLAM_ERR_INVALID;
    for (auto &mb : m->members) {
        (void)hipSetDevice(mb.device);
        for (DevBuf *b : {&mb.bgr, &mb.seeds, &mb.xy, &mb.desc, &mb.nodes, &mb.n, &mb.matches, &mb.best, &mb.F, &mb.rec, &mb.pattern})
            if (b->p) vslam_dev_free(mb.ctx, b->p);
        vslam_ctx_destroy(mb.ctx);
    }
    delete m;
    return VSLAM_OK;
}

int vslam_multi_size(const vslam_multi *m) { return m ? (int)m->members.size() : 0; }
vslam_ctx *vslam_multi_ctx(vslam_multi *m, int i) { return (m && i >= 0 && i < (int)m->members.size()) ? m->members[(size_t)i].ctx : nullptr; }
const char *vslam_multi_last_error(vslam_multi *m) { return m ? m->err.c_str() : "null object"; }

int vslam_multi_frontend_pairs(vslam_multi *m, const uint8_t *h_bgr_last, const uint8_t *h_bgr_cur, int pairs, int width,
                               int height, int row_stride, const vslam_extract_params *params, const int8_t *h_pattern,
                               int kp_stride, uint32_t base_seed, int hyp, float threshold, int32_t *h_records,
                               int32_t *h_n_keypoints) {
    if (!m) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(m->mu);
    m->err.clear();
    if (!h_bgr_last || !h_bgr_cur || !params || !h_records || pairs <= 0 || width <= 0 || height <= 0 ||
        row_stride < 3 * width || kp_stride <= 0 || hyp <= 0 || params->d_pattern) {
        m->err = "vslam_multi_frontend_pairs: bad argument (params->d_pattern must be NULL here: the table is h_pattern)";
        return VSLAM_ERR_INVALID;
    }
    MultiJob job;
    job.h_last = h_bgr_last;
    job.h_cur = h_bgr_cur;
    job.pairs = pairs, job.width = width, job.height = height, job.row_stride = row_stride, job.kp_stride = kp_stride, job.hyp = hyp;
    job.params = params, job.h_pattern = h_pattern, job.base_seed = base_seed, job.threshold = threshold;
    job.h_records = h_records, job.h_n_keypoints = h_n_keypoints;
    return run_job(m, job);
}

int vslam_multi_frontend_pairs_resident(vslam_multi *m, const uint8_t *const *d_bgr, int pairs, int width, int height,
                                        int row_stride, const vslam_extract_params *params, const int8_t *h_pattern,
                                        int kp_stride, uint32_t base_seed, int hyp, float threshold, int32_t *h_records,
                                        int32_t *h_n_keypoints) {
    if (!m) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(m->mu);
    m->err.clear();
    bool ok = d_bgr && params && h_records && pairs > 0 && width > 0 && height > 0 && row_stride >= 3 * width && kp_stride > 0 &&
              hyp > 0 && !params->d_pattern;
    const int world = (int)m->members.size();
    for (int r = 0; ok && r < world; r++) {
        int lo = 0, hi = 0;
        vslam_shard_range(pairs, r, world, &lo, &hi);
        if (hi > lo && !d_bgr[r]) ok = false;   // a slot with an empty slice needs no frames
    }
    if (!ok) {
        m->err = "vslam_multi_frontend_pairs_resident: bad argument (one device pointer per slot with a non-empty slice; "
                 "params->d_pattern must be NULL: the table is h_pattern)";
        return VSLAM_ERR_INVALID;
    }
    MultiJob job;
    job.d_bgr = d_bgr;
    job.pairs = pairs, job.width = width, job.height = height, job.row_stride = row_stride, job.kp_stride = kp_stride, job.hyp = hyp;
    job.params = params, job.h_pattern = h_pattern, job.base_seed = base_seed, job.threshold = threshold;
    job.h_records = h_records, job.h_n_keypoints = h_n_keypoints;
    return run_job(m, job);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ one process per device: RCCL
namespace {
struct Rccl {
    void *lib = nullptr;
    vs_ncclGetUniqueId_t get_id = nullptr;
    vs_ncclCommInitRank_t init_rank = nullptr;
    vs_ncclCommDestroy_t destroy = nullptr;
    vs_ncclCommCount_t count = nullptr;
    vs_ncclCommUserRank_t user_rank = nullptr;
    vs_ncclAllGather_t all_gather = nullptr;
    vs_ncclSend_t send = nullptr;
    vs_ncclRecv_t recv = nullptr;
    vs_ncclGroup_t group_start = nullptr, group_end = nullptr;
    vs_ncclGetErrorString_t err_string = nullptr;
    std::string why;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.why = std::string("librccl.so not found: ") + dlerror();
            return;
        }
        r.get_id = reinterpret_cast<decltype(r.get_id)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.init_rank = reinterpret_cast<decltype(r.init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.destroy = reinterpret_cast<decltype(r.destroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.count = reinterpret_cast<decltype(r.count)>(dlsym(r.lib, "ncclCommCount"));
        r.user_rank = reinterpret_cast<decltype(r.user_rank)>(dlsym(r.lib, "ncclCommUserRank"));
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
        r.send = reinterpret_cast<decltype(r.send)>(dlsym(r.lib, "ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(dlsym(r.lib, "ncclRecv"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(r.lib, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(r.lib, "ncclGroupEnd"));
        r.err_string = reinterpret_cast<decltype(r.err_string)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.get_id || !r.init_rank || !r.destroy || !r.count || !r.user_rank || !r.all_gather || !r.send || !r.recv ||
            !r.group_start || !r.group_end || !r.err_string)
            r.why = "librccl.so lacks an entry point";
    });
    return r;
}
static_assert(sizeof(ncclUniqueId) == VSLAM_COMM_ID_BYTES, "VSLAM_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
}  // namespace

struct vslam_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
    int device = 0;   // the device of the context it was made on: RCCL calls want it current
    // One communicator may serve several contexts of its device (a pipeline's batches in flight, each on its own stream).  Its
    // collectives then follow one another in issue order whatever stream they are on: each waits for the event recorded behind
    // the one before it, so no two of them are ever in progress at once -- nothing is left to how RCCL orders them internally.
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
};

// in front of a collective on `stream`: behind the communicator's previous one if that went to another stream
static hipError_t comm_order_before(vslam_comm *c, hipStream_t stream) {
    if (c->last && c->last_stream != stream) return hipStreamWaitEvent(stream, c->last, 0);
    return hipSuccess;
}
static hipError_t comm_order_after(vslam_comm *c, hipStream_t stream) {
    if (!c->last) {
        const hipError_t e = hipEventCreateWithFlags(&c->last, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    c->last_stream = stream;
    return hipEventRecord(c->last, stream);
}

extern "C" {

int vslam_comm_unique_id(void *id_out) {
    if (!id_out) return VSLAM_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.why.empty()) return VSLAM_ERR_COMM;
    ncclUniqueId id;
    if (r.get_id(&id) != ncclSuccess) return VSLAM_ERR_COMM;
    std::memcpy(id_out, &id, sizeof(id));
    return VSLAM_OK;
}

int vslam_comm_create(vslam_ctx *ctx, const void *id, int world, int rank, vslam_comm **out) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, id && out && world > 0 && rank >= 0 && rank < world, VSLAM_ERR_INVALID);
    *out = nullptr;
    Rccl &r = rccl();
    if (!r.why.empty()) {
        ctx->err = r.why;
        return VSLAM_ERR_COMM;
    }
    VS_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    auto *c = new vslam_comm();
    c->world = world;
    c->rank = rank;
    c->device = ctx->device;
    const ncclResult_t rc = r.init_rank(&c->comm, world, uid, rank);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + r.err_string(rc);
        delete c;
        return VSLAM_ERR_COMM;
    }
    *out = c;
    return VSLAM_OK;
}

int vslam_comm_destroy(vslam_comm *c) {
    if (!c) return VSLAM_ERR_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        rccl().destroy(c->comm);
    }
    if (c->last) (void)hipEventDestroy(c->last);
    delete c;
    return VSLAM_OK;
}

int vslam_comm_info(vslam_comm *c, int *world_out, int *rank_out) {
    if (!c || !c->comm) return VSLAM_ERR_INVALID;
    int n = 0, r = 0;   // asked of RCCL, not of what vslam_comm_create was told
    if (rccl().count(c->comm, &n) != ncclSuccess || rccl().user_rank(c->comm, &r) != ncclSuccess) return VSLAM_ERR_COMM;
    if (world_out) *world_out = n;
    if (rank_out) *rank_out = r;
    return VSLAM_OK;
}

int vslam_gather_records(vslam_ctx *ctx, vslam_comm *comm, const int32_t *d_records, size_t words_per_rank, int32_t *d_all) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, comm && comm->comm && d_records && d_all && words_per_rank > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, comm->device == ctx->device, VSLAM_ERR_INVALID);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, comm_order_before(comm, ctx->stream));
    const ncclResult_t rc = rccl().all_gather(d_records, d_all, words_per_rank, ncclInt32, comm->comm, ctx->stream);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclAllGather: ") + rccl().err_string(rc);
        return VSLAM_ERR_COMM;
    }
    VS_HIP(ctx, comm_order_after(comm, ctx->stream));
    return VSLAM_OK;
}

// Uneven slices (vslam_shard_range gives the first items % world ranks one pair more) and the rooted form: every rank r
// contributes h_words[r] words; with root < 0 every rank receives all of them (an all-gather with counts), with root >= 0
// only that rank does -- each peer sends its block once, straight to the root, over its own xGMI link (a
// ring would pass the same bytes over one link per hop).  One group of point-to-point sends and receives.
int vslam_gather_records_v(vslam_ctx *ctx, vslam_comm *comm, const int32_t *d_records, const size_t *h_words, int root,
                           int32_t *d_all) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, comm && comm->comm && h_words && root < comm->world, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, comm->device == ctx->device, VSLAM_ERR_INVALID);
    const int world = comm->world, me = comm->rank;
    const bool receiver = root < 0 || root == me;
    VS_REQUIRE(ctx, (h_words[me] == 0 || d_records) && (!receiver || d_all), VSLAM_ERR_INVALID);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    Rccl &r = rccl();
    size_t my_off = 0;
    for (int q = 0; q < me; q++) my_off += h_words[q];
    if (receiver && h_words[me])   // the own block does not travel
        VS_HIP(ctx, hipMemcpyAsync(d_all + my_off, d_records, 4 * h_words[me], hipMemcpyDeviceToDevice, ctx->stream));
    if (world == 1) return VSLAM_OK;
    VS_HIP(ctx, comm_order_before(comm, ctx->stream));
    ncclResult_t rc = r.group_start();
    size_t off = 0;
    for (int q = 0; q < world && rc == ncclSuccess; q++) {
        if (q != me) {
            if (h_words[me] && (root < 0 || root == q)) rc = r.send(d_records, h_words[me], ncclInt32, q, comm->comm, ctx->stream);
            if (rc == ncclSuccess && receiver && h_words[q]) rc = r.recv(d_all + off, h_words[q], ncclInt32, q, comm->comm, ctx->stream);
        }
        off += h_words[q];
    }
    const ncclResult_t rc_end = r.group_end();   // always closed, or the communicator stays in group mode
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) {
        ctx->err = std::string("vslam_gather_records_v (ncclSend / ncclRecv): ") + r.err_string(rc);
        return VSLAM_ERR_COMM;
    }
    VS_HIP(ctx, comm_order_after(comm, ctx->stream));
    return VSLAM_OK;
}

}  // extern "C"

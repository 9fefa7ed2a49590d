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
#include <rccl/rccl.h>   // types and prototypes only: the entry points are looked up in librccl.so at run time

#include <cstring>
#include <mutex>
#include <thread>

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
}  // namespace

struct vslam_multi {
    std::vector<Member> members;
    std::string err;
    std::mutex mu;   // one batch at a time per object
};

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
    const int world = (int)m->members.size();
    int callers_device = -1;   // slice 0 runs on the calling thread: its current device is put back afterwards
    (void)hipGetDevice(&callers_device);
    struct DeviceGuard {
        int dev;
        ~DeviceGuard() {
            if (dev >= 0) (void)hipSetDevice(dev);
        }
    } device_guard{callers_device};
    const size_t frame_bytes = (size_t)height * row_stride;
    const size_t words = 13 + (size_t)kp_stride;
    auto work = [&](int r) {
        Member &mb = m->members[(size_t)r];
        mb.rc = VSLAM_OK;
        mb.err.clear();
        int lo = 0, hi = 0;
        vslam_shard_range(pairs, r, world, &lo, &hi);
        const int ps = hi - lo;
        if (ps == 0) return;
        auto fail = [&](int rc, const char *what) {
            mb.rc = rc;
            mb.err = std::string(what) + ": " + vslam_last_error(mb.ctx);
        };
        if (hipSetDevice(mb.device) != hipSuccess) return fail(VSLAM_ERR_HIP, "hipSetDevice");
        int rc;
        const size_t P = (size_t)ps, K = (size_t)kp_stride;
        if ((rc = grow(mb, mb.bgr, 2 * P * frame_bytes)) || (rc = grow(mb, mb.seeds, 4 * P)) || (rc = grow(mb, mb.xy, 8 * 2 * P * K)) ||
            (rc = grow(mb, mb.desc, 32 * 2 * P * K)) || (rc = grow(mb, mb.nodes, 4 * 2 * P * K)) || (rc = grow(mb, mb.n, 4 * 2 * P)) ||
            (rc = grow(mb, mb.matches, 8 * P * K)) || (rc = grow(mb, mb.best, 16 * P)) || (rc = grow(mb, mb.F, 36 * P)) ||
            (rc = grow(mb, mb.rec, 4 * P * words)) || (rc = grow(mb, mb.pattern, 1024)))
            return fail(rc, "device buffers");
        // the slice's frames: "last" ones first, "current" ones behind them (vslam_frontend_pairs' layout); uploads run on
        // the copy stream, the compute stream waits for them
        uint8_t *d_bgr = static_cast<uint8_t *>(mb.bgr.p);
        if ((rc = vslam_upload_async(mb.ctx, d_bgr, h_bgr_last + (size_t)lo * frame_bytes, P * frame_bytes)) ||
            (rc = vslam_upload_async(mb.ctx, d_bgr + P * frame_bytes, h_bgr_cur + (size_t)lo * frame_bytes, P * frame_bytes)))
            return fail(rc, "frame upload");
        std::vector<uint32_t> seeds(P);
        for (int i = 0; i < ps; i++) seeds[(size_t)i] = base_seed ^ (uint32_t)(lo + i);   // GLOBAL pair index: results do not depend on the split
        if ((rc = vslam_upload_async(mb.ctx, mb.seeds.p, seeds.data(), 4 * P))) return fail(rc, "seed upload");
        vslam_extract_params p = *params;
        if (h_pattern) {
            if ((rc = vslam_upload_async(mb.ctx, mb.pattern.p, h_pattern, 1024))) return fail(rc, "pattern upload");
            p.d_pattern = static_cast<const int8_t *>(mb.pattern.p);
        }
        if ((rc = vslam_upload_wait(mb.ctx))) return fail(rc, "upload");   // `seeds` is about to go out of scope; the frames may be pageable
        if ((rc = vslam_frontend_pairs(mb.ctx, d_bgr, ps, width, height, row_stride, &p, kp_stride, static_cast<const uint32_t *>(mb.seeds.p),
                                       hyp, threshold, static_cast<float *>(mb.xy.p), static_cast<uint8_t *>(mb.desc.p),
                                       static_cast<int32_t *>(mb.nodes.p), static_cast<int32_t *>(mb.n.p),
                                       static_cast<int32_t *>(mb.matches.p), static_cast<int32_t *>(mb.best.p), static_cast<float *>(mb.F.p))))
            return fail(rc, "frontend_pairs");
        if ((rc = vslam_pack_records(mb.ctx, static_cast<const float *>(mb.F.p), static_cast<const int32_t *>(mb.best.p),
                                     static_cast<const int32_t *>(mb.matches.p), ps, kp_stride, static_cast<int32_t *>(mb.rec.p))))
            return fail(rc, "pack_records");
        // the "gather": this slice's records land at its offset of the caller's array
        if ((rc = vslam_copy_d2h(mb.ctx, h_records + (size_t)lo * words, mb.rec.p, 4 * P * words))) return fail(rc, "records download");
        if (h_n_keypoints) {
            if ((rc = vslam_copy_d2h(mb.ctx, h_n_keypoints + lo, mb.n.p, 4 * P)) ||
                (rc = vslam_copy_d2h(mb.ctx, h_n_keypoints + pairs + lo, static_cast<const int32_t *>(mb.n.p) + ps, 4 * P)))
                return fail(rc, "counts download");
        }
        if ((rc = vslam_ctx_synchronize(mb.ctx))) return fail(rc, "synchronize");
    };
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
            for (int r = 1; r < world; r++) pool.t.emplace_back(work, r);
        } catch (const std::exception &e) {
            m->err = std::string("vslam_multi_frontend_pairs: cannot start a worker thread: ") + e.what();
        }
        if (m->err.empty()) work(0);
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

}  // extern "C"

// ------------------------------------------------------------------------------------------ one process per device: RCCL
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) err_string = nullptr;
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
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
        r.err_string = reinterpret_cast<decltype(r.err_string)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.get_id || !r.init_rank || !r.destroy || !r.all_gather || !r.err_string) r.why = "librccl.so lacks an entry point";
    });
    return r;
}
static_assert(sizeof(ncclUniqueId) == VSLAM_COMM_ID_BYTES, "VSLAM_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
}  // namespace

struct vslam_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

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
    if (c->comm) rccl().destroy(c->comm);
    delete c;
    return VSLAM_OK;
}

int vslam_gather_records(vslam_ctx *ctx, vslam_comm *comm, const int32_t *d_records, size_t words_per_rank, int32_t *d_all) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, comm && comm->comm && d_records && d_all && words_per_rank > 0, VSLAM_ERR_INVALID);
    const ncclResult_t rc = rccl().all_gather(d_records, d_all, words_per_rank, ncclInt32, comm->comm, ctx->stream);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclAllGather: ") + rccl().err_string(rc);
        return VSLAM_ERR_COMM;
    }
    return VSLAM_OK;
}

}  // extern "C"

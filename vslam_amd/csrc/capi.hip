// C ABI (include/vslam_amd.h): context, memory, event timing, and the entry points that chain
// the stage launchers.  No CPU fallback anywhere: every entry point needs a live HIP device.
#include "ctx.h"
#include "../../include/vslam_brief_pattern_31.h"
#include <cstdlib>

#include <cstring>

// ------------------------------------------------------------------------------------------
// arena + profiling helpers
// ------------------------------------------------------------------------------------------
int vs_arena_get(vslam_ctx *ctx, const char *name, size_t bytes, void **out) {
    vslam_ctx::Buf &buf = ctx->arena[name];
    if (buf.bytes < bytes) {
        if (buf.ptr) {
            // the old block may still be in use by queued kernels
            VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            VS_HIP(ctx, hipFree(buf.ptr));
            buf.ptr = nullptr;
            buf.bytes = 0;
        }
        VS_HIP(ctx, hipMalloc(&buf.ptr, bytes));
        buf.bytes = bytes;
    }
    *out = buf.ptr;
    return VSLAM_OK;
}

// ORB's learned rBRIEF table (include/vslam_brief_pattern_31.h) on the device: what a NULL d_pattern means
static int vs_default_pattern(vslam_ctx *ctx, const int8_t **out) {
    const bool fresh = ctx->arena.find("ctx.brief_pattern_31") == ctx->arena.end();
    void *p = nullptr;
    int rc = vs_arena_get(ctx, "ctx.brief_pattern_31", 1024, &p);
    if (rc) return rc;
    if (fresh) VS_HIP(ctx, hipMemcpyAsync(p, vslam_brief_pattern_31_table, 1024, hipMemcpyHostToDevice, ctx->stream));
    *out = (const int8_t *)p;
    return VSLAM_OK;
}

int vs_aux_job_point(vslam_ctx *ctx, int point) {
    if (ctx->aux_job_at != point || !ctx->aux_job) return VSLAM_OK;
    ctx->aux_job_at = 0;
    VS_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    VS_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->aux_stream;
    const int rc = ctx->aux_job();
    ctx->stream = main_stream;
    ctx->aux_job = nullptr;
    if (rc) return rc;
    VS_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
    return VSLAM_OK;
}

// Hand the k-d build of `frames` frames to the matching stages (see ctx.h: aux_job).  Where it is forked: beside the
// matcher it takes the matcher's wave slots and registers; behind it, it runs beside the set mapping (one workgroup per
// pair) and the first solves.  Measured (tools/ab_step.py, one process, alternating blocks): at C3 (2000 keypoints: the
// build outlasts the FP4 matcher, 0.29 against 0.19 ms) 2.91 ms in front of the matcher, 2.85 behind it, 2.86 behind the set
// mapping, 2.89 behind the solves, 2.95 behind the screen (2.78 with no trees at all); at C5 (4000 keypoints: the matcher
// is the longer one, 1.30 against 1.08 ms) 15.26 in front, 15.31 behind.  So: behind the matcher up to 2048 keypoint slots.
static int vs_defer_tree_build(vslam_ctx *ctx, const float *d_xy, const int32_t *d_n, int frames, int kp_stride, int32_t *d_nodes) {
    ctx->aux_job = [=]() { return vs_launch_kdtree_build(ctx, d_xy, d_n, frames, kp_stride, d_nodes); };
    ctx->aux_job_at = ctx->tree_fork >= 0 ? ctx->tree_fork : (kp_stride <= 2048 ? 1 : 0);
    return vs_aux_job_point(ctx, 0);
}
struct VsAuxGuard {   // a job that was never forked (an error return in between) must not outlive the call that made it
    vslam_ctx *c;
    ~VsAuxGuard() {
        c->aux_job = nullptr;
        c->aux_job_at = 0;
    }
};

// what a non-zero device-side error word says (vslam_ctx_synchronize, vslam_pipeline_wait)
std::string vs_errflag_message(int32_t flag) {
    std::string m = "a fixed-size device list overflowed (flag " + std::to_string(flag) + ")";
    if (flag & 4)
        m += ": more frames of one call needed the corner detector's whole-image fallback than its pool holds; "
             "those frames got no corners (VSLAM_OPT_CORNER_LIST_CAP = -1 sizes every list for the whole image)";
    return m;
}

int vs_device_errflag(vslam_ctx *ctx, int32_t **out) {
    const bool fresh = ctx->arena.find("ctx.errflag") == ctx->arena.end();
    int rc = vs_arena_get(ctx, "ctx.errflag", sizeof(int32_t), (void **)out);
    if (rc) return rc;
    if (fresh) VS_HIP(ctx, hipMemsetAsync(*out, 0, sizeof(int32_t), ctx->stream));
    return VSLAM_OK;
}

static hipEvent_t vs_event_take(vslam_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

VsProfScope::VsProfScope(vslam_ctx *c, const char *name) : ctx(c), active(c && c->prof) {
    if (!active) return;
    auto it = ctx->prof_index.find(name);
    if (it == ctx->prof_index.end()) {
        p.slot = (int)ctx->prof_slots.size();
        ctx->prof_index[name] = p.slot;
        vslam_prof_slot s;
        s.name = name;
        ctx->prof_slots.push_back(s);
    } else {
        p.slot = it->second;
    }
    p.start = vs_event_take(ctx);
    p.stop = vs_event_take(ctx);
    (void)hipEventRecord(p.start, ctx->stream);
}

VsProfScope::~VsProfScope() {
    if (!active) return;
    (void)hipEventRecord(p.stop, ctx->stream);
    ctx->prof_pending.push_back(p);
}

static int vs_prof_fold(vslam_ctx *ctx) {
    if (ctx->prof_pending.empty()) return VSLAM_OK;
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->prof_pending) {
        float ms = 0;
        VS_HIP(ctx, hipEventElapsedTime(&ms, p.start, p.stop));
        ctx->prof_slots[p.slot].total_ms += ms;
        ctx->prof_slots[p.slot].launches += 1;
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->prof_pending.clear();
    return VSLAM_OK;
}

// Known-byte-count streaming copies used only to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE
// counters for this repo's two access widths (4 B and 16 B per lane), as MI355X_MICROARCH.md asks.
__global__ __launch_bounds__(256) void pmc_calib_copy4_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                              size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void pmc_calib_copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                               size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// A kernel of KNOWN vector-pipe occupancy, to calibrate rocprofv3's SQ_ACTIVE_INST_VALU the way the copies above calibrate
// FETCH_SIZE: eight waves on every SIMD, each issuing `iters` x 16 independent v_fma_f32 and next to nothing else.  By
// construction the vector pipes are busy for the whole launch, so (SQ_ACTIVE_INST_VALU of this kernel) / (its GRBM_GUI_ACTIVE)
// is the counter ratio that means "100 %" -- tools/sq_summary.py divides every other kernel's ratio by it.
__global__ __launch_bounds__(256) void pmc_calib_valu_kernel(float *__restrict__ out, float seed, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + (float)i + (float)threadIdx.x;
    const float c = seed * 0.999f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += a[i];
    if (acc == 12345.678f) out[0] = acc;   // never true: keeps the chain alive
}

extern "C" {

int vslam_debug_valu_calib(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    float *out = nullptr;
    int rc = vs_arena_get(ctx, "ctx.valu_calib", 256, (void **)&out);
    if (rc) return rc;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    VsProfScope ps(ctx, "pmc_calib_valu_kernel");
    pmc_calib_valu_kernel<<<cus * 8, 256, 0, ctx->stream>>>(out, 1.25f, 4000);   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vslam_debug_stream_copy(vslam_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int bytes_per_lane) {
    if (!ctx || !d_src || !d_dst || (bytes_per_lane != 4 && bytes_per_lane != 16) || bytes % 16) return VSLAM_ERR_INVALID;
    if (bytes_per_lane == 4) {
        VsProfScope ps(ctx, "pmc_calib_copy4_kernel");
        pmc_calib_copy4_kernel<<<2048, 256, 0, ctx->stream>>>((const uint32_t *)d_src, (uint32_t *)d_dst, bytes / 4);
    } else {
        VsProfScope ps(ctx, "pmc_calib_copy16_kernel");
        pmc_calib_copy16_kernel<<<2048, 256, 0, ctx->stream>>>((const uint4 *)d_src, (uint4 *)d_dst, bytes / 16);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

const char *vslam_version(void) { return "vslam_amd 0.1 (gfx950)"; }
const int8_t *vslam_brief_pattern_31(void) { return vslam_brief_pattern_31_table; }

int vslam_ctx_create(int device, vslam_ctx **out) { return vs_ctx_create(device, false, out); }

}  // extern "C"

// shared_chip: the context is one of several that keep batches in flight on the same device (vslam_pipeline_create).  What
// pays for one batch at a time -- the blur forked onto a low-priority auxiliary stream so that it fills the holes of the
// batch's own latency-bound stages -- does not with several: the holes are filled by the other batches, and every fork / join
// between queues is a hand-over of 50-120 us.  Such a context keeps its blur on the main stream, runs its auxiliary stream
// (generator, k-d build) at the main stream's priority, and makes no stream it may never use (vs_copy_stream).
// Measured with FRESH PROCESSES per configuration (tools/ab_proc.sh; which hardware queue a stream lands on depends on what
// the process created before it, and that alone moves a step by up to 9 %, so configurations compared inside one process
// are not comparable), SURVEY 8(d) data, ms per batch, four in flight: C3 2.61 against 2.69-2.74 for plain contexts, C2 0.304
// against 0.35-0.37, C5 14.2-14.5 either way.  Variations at C3, three in flight (this arrangement 2.65): blur forked 2.68,
// k-d build in front of the matcher 2.69, behind the set mapping 2.70, behind the solves 2.73, in line with the generator
// in line 2.72-2.76; the main streams at high priority (a hardware-queue pool of their own) 2.79; an idle fourth stream per
// context 2.85; GPU_MAX_HW_QUEUES 2 / 3 / 6 instead of the runtime's 4: 2.81 / 2.73 / 2.71.

// The copy stream of a context that keeps batches in flight is made on first use.  Every stream a process holds takes a
// place in the runtime's pool of hardware queues (GPU_MAX_HW_QUEUES = 4 per priority level, streams beyond that share the
// least used one), so streams that are never used still decide which of the USED streams end up sharing a queue -- and two
// streams on one hardware queue run one after the other.  Measured in fresh processes (tools/ab_proc.sh, three batches in
// flight at C3): the same kernels and forks 2.69 ms per batch with three streams per context, 2.85 with a fourth that is idle.
static int vs_copy_stream(vslam_ctx *ctx, hipStream_t *out) {
    if (!ctx->copy_stream) {
        // on the context's device, whatever the caller's thread has current (two pipelines on two devices in one thread);
        // the caller's device is put back
        int cur = -1;
        (void)hipGetDevice(&cur);
        VS_HIP(ctx, hipSetDevice(ctx->device));
        const hipError_t e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
        if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
        VS_HIP(ctx, e);
    }
    *out = ctx->copy_stream;
    return VSLAM_OK;
}

int vs_ctx_create(int device, bool shared_chip, vslam_ctx **out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_SHARED_CHIP")) shared_chip = shared_chip && e[0] != '0';   // A/B: pipeline contexts as plain ones
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return VSLAM_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return VSLAM_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return VSLAM_ERR_HIP;
    vslam_ctx *ctx = new vslam_ctx();
    ctx->device = device;
    // The auxiliary stream carries work that fills holes beside the main chain (blur, k-d build, generator): the main
    // stream's workgroups go first whenever both have some ready (VSLAM_STREAM_PRIORITY=0: both at the default priority).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_STREAM_PRIORITY")) {
        if (e[0] == '0') prio_least = prio_greatest = 0;
        if (e[0] == '2') prio_greatest = 0;   // main at the default priority, auxiliary below it
    }
    bool shared_main_high = false;
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_STREAM_PRIORITY")) shared_main_high = e[0] == '3';   // A/B: shared-chip mains above their auxiliaries
    if (shared_chip) {
        prio_least = 0;   // the auxiliary stream level with the main one, the main streams of all contexts at the default priority
        if (!shared_main_high) prio_greatest = 0;
    }
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
        delete ctx;
        return VSLAM_ERR_HIP;
    }
    ctx->own_stream = true;
    if (hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return VSLAM_ERR_HIP;
    }
    ctx->lazy_streams = shared_chip;   // streams a context may never use are made on first use (see vs_copy_stream)
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_LAZY_STREAMS")) ctx->lazy_streams = e[0] != '0';
    if ((!ctx->lazy_streams && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&ctx->ev_raw, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return VSLAM_ERR_HIP;
    }
    ctx->shared_chip = shared_chip;
    if (shared_chip) ctx->overlap_blur = 0;
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_OVERLAP_BLUR")) ctx->overlap_blur = e[0] - '0';
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_SETS_PREFETCH")) ctx->sets_prefetch = e[0] != '0';
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_TREE_FORK")) ctx->tree_fork = atoi(e);   // A/B across processes (the option does the same)
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_RANSAC_SOLVE_SPLIT")) ctx->solve_split = atoi(e);
    *out = ctx;
    return VSLAM_OK;
}

extern "C" {

int vslam_ctx_make_current(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    return VSLAM_OK;
}

int vslam_ctx_destroy(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->arena)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto &p : ctx->prof_pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_raw) (void)hipEventDestroy(ctx->ev_raw);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VSLAM_OK;
}

int vslam_ctx_set_stream(vslam_ctx *ctx, void *hip_stream) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    // taken literally: NULL is HIP's default (null) stream, which is what torch hands out as
    // its default current stream
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return VSLAM_OK;
}

int vslam_ctx_synchronize(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto it = ctx->arena.find("ctx.errflag");
    if (it != ctx->arena.end() && it->second.ptr) {
        int32_t flag = 0;
        VS_HIP(ctx, hipMemcpy(&flag, it->second.ptr, sizeof(flag), hipMemcpyDeviceToHost));
        if (flag) {
            VS_HIP(ctx, hipMemset(it->second.ptr, 0, sizeof(flag)));
            ctx->err = vs_errflag_message(flag);
            return VSLAM_ERR_CAPACITY;
        }
    }
    return VSLAM_OK;
}

int vslam_ctx_wait(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VSLAM_OK;
}

const char *vslam_last_error(vslam_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// diagnostics of the corner detector's last batch (bench.py's data regimes): waits for the stream, copies the counters
int vslam_corner_stats(vslam_ctx *ctx, uint64_t *h_stats) {
    if (!ctx || !h_stats) return VSLAM_ERR_INVALID;
    for (int i = 0; i < 5; i++) h_stats[i] = 0;
    VS_REQUIRE(ctx, ctx->stat_counts && ctx->stat_pool_count && ctx->stat_frames > 0, VSLAM_ERR_INVALID);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> counts((size_t)ctx->stat_frames);
    int32_t tickets = 0;
    VS_HIP(ctx, hipMemcpy(counts.data(), ctx->stat_counts, sizeof(uint32_t) * counts.size(), hipMemcpyDeviceToHost));
    VS_HIP(ctx, hipMemcpy(&tickets, ctx->stat_pool_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    uint64_t listed = 0;
    for (uint32_t v : counts) listed += v;
    h_stats[0] = (uint64_t)ctx->stat_frames;
    h_stats[1] = (uint64_t)ctx->stat_px;
    h_stats[2] = listed;
    h_stats[3] = (uint64_t)(tickets < 0 ? 0 : tickets);
    h_stats[4] = (uint64_t)ctx->stat_pool_slots;
    return VSLAM_OK;
}

int vslam_ctx_workspace_bytes(vslam_ctx *ctx, size_t *bytes_out) {
    if (!ctx || !bytes_out) return VSLAM_ERR_INVALID;
    size_t total = 0;
    for (const auto &kv : ctx->arena) total += kv.second.bytes;
    *bytes_out = total;
    return VSLAM_OK;
}

int vslam_ctx_set_option(vslam_ctx *ctx, int option, int value) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (option == VSLAM_OPT_RANSAC_ALL_SUMS) {
        ctx->ransac_all_sums = value != 0;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_RANSAC_MIN_MATCHES) {
        VS_REQUIRE(ctx, value >= 1 && value <= VSLAM_SET_SIZE, VSLAM_ERR_INVALID);
        ctx->ransac_min_matches = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_RANSAC_MIN_ITEMS) {
        VS_REQUIRE(ctx, value >= 0 && value <= VSLAM_SET_SIZE, VSLAM_ERR_INVALID);
        ctx->ransac_min_items = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_RANSAC_SOLVER) {
        VS_REQUIRE(ctx, value == 0 || value == 1, VSLAM_ERR_INVALID);
        ctx->ransac_solver = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_CORNER_WINDOW_PCT) {
        VS_REQUIRE(ctx, value >= 0 && value <= 100000, VSLAM_ERR_INVALID);
        ctx->corner_window_pct = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_TREE_FORK) {
        VS_REQUIRE(ctx, value >= -1 && value <= 5, VSLAM_ERR_INVALID);
        ctx->tree_fork = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_MATCH_FORM) {
        VS_REQUIRE(ctx, value >= 0 && value <= 2, VSLAM_ERR_INVALID);
#ifndef VSLAM_EXPERIMENTS
        VS_REQUIRE(ctx, value <= 1 && "the int8 form of the matcher exists in the experiments build only", VSLAM_ERR_INVALID);
#endif
        ctx->match_form = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_CORNER_LIST_CAP) {
        VS_REQUIRE(ctx, value >= -1, VSLAM_ERR_INVALID);
        ctx->corner_list_cap = value;
        return VSLAM_OK;
    }
    if (option == VSLAM_OPT_MATCH_SHAPE) {
        VS_REQUIRE(ctx, value >= 0 && value <= 2, VSLAM_ERR_INVALID);
#ifndef VSLAM_EXPERIMENTS
        VS_REQUIRE(ctx, value == 0 && "the matcher's other workgroup shapes exist in the experiments build only", VSLAM_ERR_INVALID);
#endif
        ctx->match_shape = value;
        return VSLAM_OK;
    }
    VS_REQUIRE(ctx, false && "unknown option", VSLAM_ERR_INVALID);
    return VSLAM_OK;
}

int vslam_dev_alloc(vslam_ctx *ctx, size_t bytes, void **d_out) {
    VS_REQUIRE(ctx, ctx && d_out, VSLAM_ERR_INVALID);
    VS_HIP(ctx, hipMalloc(d_out, bytes ? bytes : 1));
    return VSLAM_OK;
}
int vslam_dev_free(vslam_ctx *ctx, void *d_ptr) {
    VS_REQUIRE(ctx, ctx, VSLAM_ERR_INVALID);
    if (d_ptr) VS_HIP(ctx, hipFree(d_ptr));
    return VSLAM_OK;
}
int vslam_copy_h2d(vslam_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    VS_REQUIRE(ctx, ctx && (bytes == 0 || (d_dst && h_src)), VSLAM_ERR_INVALID);
    if (bytes) {
        VS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VSLAM_OK;
}
int vslam_copy_d2h(vslam_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    VS_REQUIRE(ctx, ctx && (bytes == 0 || (h_dst && d_src)), VSLAM_ERR_INVALID);
    if (bytes) {
        VS_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VSLAM_OK;
}

int vslam_host_alloc(vslam_ctx *ctx, size_t bytes, void **h_out) {
    if (!ctx || !h_out) return VSLAM_ERR_INVALID;
    VS_HIP(ctx, hipHostMalloc(h_out, bytes ? bytes : 1, hipHostMallocDefault));
    return VSLAM_OK;
}
int vslam_host_free(vslam_ctx *ctx, void *h_ptr) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (h_ptr) VS_HIP(ctx, hipHostFree(h_ptr));
    return VSLAM_OK;
}
int vslam_upload_async(vslam_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, d_dst && h_src, VSLAM_ERR_INVALID);
    hipStream_t cs = nullptr;
    if (int rc = vs_copy_stream(ctx, &cs)) return rc;
    if (bytes) VS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, cs));
    return VSLAM_OK;
}
int vslam_upload_fence(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (!ctx->copy_stream) return VSLAM_OK;   // nothing was ever uploaded
    VS_HIP(ctx, hipEventRecord(ctx->ev_upload, ctx->copy_stream));
    VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));
    return VSLAM_OK;
}
int vslam_download_async(vslam_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, bytes == 0 || (h_dst && d_src), VSLAM_ERR_INVALID);
    if (bytes) VS_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return VSLAM_OK;
}
int vslam_upload_wait(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (ctx->copy_stream) VS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return VSLAM_OK;
}

int vslam_prof_enable(vslam_ctx *ctx, int on) {
    if (!ctx) return VSLAM_ERR_INVALID;
    int rc = vs_prof_fold(ctx);
    ctx->prof = on != 0;
    return rc;
}
int vslam_prof_reset(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    int rc = vs_prof_fold(ctx);
    for (auto &s : ctx->prof_slots) {
        s.total_ms = 0;
        s.launches = 0;
    }
    return rc;
}
int vslam_prof_count(vslam_ctx *ctx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    int rc = vs_prof_fold(ctx);
    if (rc) return rc;
    return (int)ctx->prof_slots.size();
}
int vslam_prof_get(vslam_ctx *ctx, int i, char *name, int name_cap, double *total_ms, int64_t *launches) {
    if (!ctx || i < 0 || i >= (int)ctx->prof_slots.size()) return VSLAM_ERR_INVALID;
    const vslam_prof_slot &s = ctx->prof_slots[i];
    if (name && name_cap > 0) {
        std::strncpy(name, s.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------
// stage entry points
// ------------------------------------------------------------------------------------------
int vslam_match_knn2_ratio(vslam_ctx *ctx, const uint8_t *d_desc1, const int32_t *d_n1,
                           const uint8_t *d_desc2, const int32_t *d_n2, int batch, int kp_stride,
                           int32_t *d_pairs, int32_t *d_m, int32_t *d_knn) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_match(ctx, d_desc1, d_n1, d_desc2, d_n2, batch, kp_stride, d_pairs, d_m, d_knn);
}

int vslam_ransac_sets(vslam_ctx *ctx, const uint32_t *d_seeds, const int32_t *d_m, int batch, int hyp,
                      int32_t *d_sets, uint32_t *d_draw_scratch) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_ransac_sets(ctx, d_seeds, d_m, batch, hyp, d_sets, d_draw_scratch);
}

int vslam_ransac_fundamental(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2,
                             const int32_t *d_pairs, const int32_t *d_m, const int32_t *d_sets,
                             int batch, int kp_stride, int hyp, float threshold, float *d_F,
                             uint8_t *d_mask, int32_t *d_best, int32_t *d_matches, float *d_hypF,
                             int32_t *d_hyp_count, float *d_hyp_sum) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_ransac(ctx, d_xy1, d_xy2, d_pairs, d_m, d_sets, batch, kp_stride, hyp, threshold,
                            d_F, d_mask, d_best, d_matches, d_hypF, d_hyp_count, d_hyp_sum);
}

int vslam_ransac_solve(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2, const int32_t *d_pairs,
                       const int32_t *d_m, const int32_t *d_sets, int batch, int kp_stride, int hyp,
                       float *d_hypF) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_ransac_solve(ctx, d_xy1, d_xy2, d_pairs, d_m, d_sets, batch, kp_stride, hyp, d_hypF);
}

int vslam_ransac_evaluate(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2,
                          const int32_t *d_pairs, const int32_t *d_m, const float *d_hypF, int batch,
                          int kp_stride, int hyp, float threshold, float *d_F, uint8_t *d_mask,
                          int32_t *d_best, int32_t *d_matches, int32_t *d_hyp_count, float *d_hyp_sum) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_ransac_evaluate(ctx, d_xy1, d_xy2, d_pairs, d_m, d_hypF, batch, kp_stride, hyp, threshold,
                                     d_F, d_mask, d_best, d_matches, d_hyp_count, d_hyp_sum);
}

int vslam_kdtree_build(vslam_ctx *ctx, const float *d_xy, const int32_t *d_n, int batch, int kp_stride,
                       int32_t *d_nodes) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_kdtree_build(ctx, d_xy, d_n, batch, kp_stride, d_nodes);
}

int vslam_kdtree_radius(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy, const int32_t *d_n,
                        int batch, int kp_stride, const float *d_queries, const int32_t *d_nq,
                        int q_stride, float radius, int32_t *d_hits, int32_t *d_counts, int hit_cap) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_kdtree_radius(ctx, d_nodes, d_xy, d_n, batch, kp_stride, d_queries, d_nq, q_stride,
                                   radius, d_hits, d_counts, hit_cap);
}

int vslam_kdtree_nearest(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy, const int32_t *d_n,
                         int batch, int kp_stride, const float *d_queries, const int32_t *d_nq,
                         int q_stride, float max_distance_sq, int32_t *d_best_idx) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_kdtree_nearest(ctx, d_nodes, d_xy, d_n, batch, kp_stride, d_queries, d_nq, q_stride,
                                    max_distance_sq, d_best_idx);
}

int vslam_kdtree_cell_table(vslam_ctx *ctx, const int32_t *d_nodes, const float *d_xy, const int32_t *d_n, int batch,
                            int kp_stride, int slots, uint32_t *d_table, int32_t *d_ok) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_kdtree_cell_table(ctx, d_nodes, d_xy, d_n, batch, kp_stride, slots, d_table, d_ok);
}

// State that one entry point arms for a later stage of the same call (the rotated rBRIEF table queued ahead of the
// description stage; the raw generator outputs queued ahead of vslam_match_features) must not outlive that call: on
// an error return in between, the next call would otherwise skip work it needs.  Cleared on every exit.
struct VsTableGuard {
    vslam_ctx *c;
    ~VsTableGuard() { c->rbrief_table_ready = false; c->fork_after_eigen = false; c->img_pitch = 0; }
};
struct VsPrefetchGuard {
    vslam_ctx *c;
    ~VsPrefetchGuard() { c->raw_seeds = nullptr; }
};
// rows for a width the dword kernels do not take as it is (vslam_ctx::img_pitch); 0: the width is fine (or too small to mirror)
static inline int vs_padded_pitch(int width) { return (width % 4 != 0 && width >= 64) ? (width + 3 + 15) & ~15 : 0; }

int vslam_bgr2gray(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                   int row_stride, uint8_t *d_gray) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_bgr2gray(ctx, d_bgr, frames, width, height, row_stride, d_gray);
}

int vslam_min_eigen(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                    float *d_eig) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_min_eigen(ctx, d_gray, frames, width, height, d_eig, nullptr);
}

int vslam_good_features(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                        int max_corners, double quality, double min_distance, int kp_stride,
                        float *d_xy, int32_t *d_n) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (d_gray && vs_padded_pitch(width) && frames > 0 && height > 0) {   // see vslam_extract_features: the caller's rows, copied into padded ones
        VsTableGuard guard{ctx};
        const int pitch = vs_padded_pitch(width);
        uint8_t *padded = nullptr;
        if (int rc = vs_arena_get(ctx, "stage.gray_padded", (size_t)frames * pitch * height, (void **)&padded)) return rc;
        if (int rc = vs_launch_gray_pad(ctx, d_gray, frames, width, height, padded, pitch)) return rc;
        ctx->img_pitch = pitch;
        return vs_launch_good_features(ctx, padded, frames, width, height, max_corners, quality, min_distance, kp_stride, d_xy, d_n);
    }
    return vs_launch_good_features(ctx, d_gray, frames, width, height, max_corners, quality,
                                   min_distance, kp_stride, d_xy, d_n);
}

int vslam_gaussian7(vslam_ctx *ctx, const uint8_t *d_gray, int frames, int width, int height,
                    uint8_t *d_out) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (d_gray && d_out && vs_padded_pitch(width) && frames > 0 && height >= 4) {
        VsTableGuard guard{ctx};
        const int pitch = vs_padded_pitch(width);
        uint8_t *padded = nullptr, *blurred = nullptr;
        if (int rc = vs_arena_get(ctx, "stage.gray_padded", (size_t)frames * pitch * height, (void **)&padded)) return rc;
        if (int rc = vs_arena_get(ctx, "stage.blur_padded", (size_t)frames * pitch * height, (void **)&blurred)) return rc;
        if (int rc = vs_launch_gray_pad(ctx, d_gray, frames, width, height, padded, pitch)) return rc;
        ctx->img_pitch = pitch;
        if (int rc = vs_launch_gaussian7(ctx, padded, frames, width, height, blurred)) return rc;
        return vs_launch_gray_unpad(ctx, blurred, frames, width, height, pitch, d_out);
    }
    return vs_launch_gaussian7(ctx, d_gray, frames, width, height, d_out);
}

int vslam_orb_describe(vslam_ctx *ctx, const uint8_t *d_blurred, int frames, int width, int height,
                       const float *d_xy_in, const int32_t *d_n_in, int kp_stride, float cos_a,
                       float sin_a, const int8_t *d_pattern, float *d_xy_out, uint8_t *d_desc,
                       int32_t *d_n_out) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (!d_pattern)
        if (int rc = vs_default_pattern(ctx, &d_pattern)) return rc;
    return vs_launch_orb_describe(ctx, d_blurred, frames, width, height, d_xy_in, d_n_in, kp_stride,
                                  cos_a, sin_a, d_pattern, d_xy_out, d_desc, d_n_out);
}

#ifdef VSLAM_EXPERIMENTS
// (experiments build only) the detection half of vslam_extract_features by itself: cvtColor + goodFeaturesToTrack from the 3-byte
// image, the corners BEFORE ORB::compute's border filter -- what a parity hunt near the image border needs to look at
int vslam_debug_detect(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height, int row_stride, int max_corners,
                       int kp_stride, float *d_xy, int32_t *d_n) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VsTableGuard table_guard{ctx};
    ctx->img_pitch = vs_padded_pitch(width);
    uint8_t *gray = nullptr;
    if (int rc = vs_arena_get(ctx, "extract.gray", (size_t)frames * vs_pitch(ctx, width) * height, (void **)&gray)) return rc;
    const VsBgrSource src{d_bgr, row_stride};
    return vs_launch_good_features(ctx, gray, frames, width, height, max_corners, 0.01, 3.0, kp_stride, d_xy, d_n, &src);
}
#endif

// extract_features(Frame&), src/Frame.cpp:53-80
int vslam_extract_features(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                           int row_stride, const vslam_extract_params *params, int kp_stride,
                           float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                           int32_t *d_n_detected) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VsTableGuard table_guard{ctx};
    VS_REQUIRE(ctx, d_bgr && params && d_xy && d_desc && d_n, VSLAM_ERR_INVALID);
    vslam_extract_params with_table;
    if (!params->d_pattern) {   // the default: ORB's learned table
        with_table = *params;
        if (int prc = vs_default_pattern(ctx, &with_table.d_pattern)) return prc;
        params = &with_table;
    }
    VS_REQUIRE(ctx, params->max_corners > 0 && params->max_corners <= kp_stride, VSLAM_ERR_INVALID);
    // A width that is no multiple of 4 leaves the dword kernels (two-tier detector, streaming blur, tile-staged descriptors)
    // without aligned rows.  The internal gray and blurred planes then get rows of a multiple of 16 bytes, at least 3 longer
    // than the image is wide, the tail holding the row's BORDER_REFLECT_101 continuation (written by cvtColor): the same
    // kernels run on those rows and what they produce below column `width` is what the image alone would give
    // (blur.hip, response.hip: the one place that needs a correction is the sign of a mirrored x-derivative).
    ctx->img_pitch = vs_padded_pitch(width);
    const size_t px = (size_t)frames * vs_pitch(ctx, width) * height;
    uint8_t *gray = nullptr, *blur = nullptr;
    float *xy_det = nullptr;
    int32_t *n_det = nullptr;
    int rc;
    if ((rc = vs_arena_get(ctx, "extract.gray", px, (void **)&gray))) return rc;
    if ((rc = vs_arena_get(ctx, "extract.blur", px, (void **)&blur))) return rc;
    if ((rc = vs_arena_get(ctx, "extract.xy_det", sizeof(float) * 2 * (size_t)frames * kp_stride, (void **)&xy_det))) return rc;
    if (d_n_detected) n_det = d_n_detected;
    else if ((rc = vs_arena_get(ctx, "extract.n_det", sizeof(int32_t) * (size_t)frames, (void **)&n_det))) return rc;

    // cvtColor (:56) is the detector's first kernel (or a launch of its own in front of it, for layouts that one does not take).
    // The blur needs only the gray image: run it on the auxiliary stream beside corner detection, whose
    // selection stage is latency-bound and leaves most of the chip idle (not while per-kernel timing is on).
    const bool overlap = ctx->overlap_blur > 0 && !ctx->prof;
    const VsBgrSource src{d_bgr, row_stride};
    ctx->fork_after_eigen = overlap;
    rc = vs_launch_good_features(ctx, gray, frames, width, height, params->max_corners,                  // :56, :61
                                 params->quality, params->min_distance, kp_stride, xy_det, n_det, &src);
    ctx->fork_after_eigen = false;
    if (rc) return rc;
    {
        hipStream_t main_stream = ctx->stream;
        if (overlap) ctx->stream = ctx->aux_stream;
        // The table the description stage will want: off the main stream's critical path.  It sits in FRONT of the blur on
        // purpose.  The 8-wave kernel finds every CU held by the selection's 1024-thread workgroups and waits about 65 us
        // for a slot (profiles/r04_step_timeline.txt), which is what lets the selection take its places before the blur's
        // waves arrive: queued at the head of the auxiliary stream instead (round 5, tools/ab_lib.py, same process,
        // alternating) the blur starts at the fork and the step is unchanged with one batch in flight (2.950 vs 2.950 ms) and
        // SLOWER with three (2.764 -> 2.800 ms on the hard data, 2.604 -> 2.664 on the easy data).
        if (overlap && vs_pitch(ctx, width) % 4 == 0)
            rc = vs_launch_rbrief_rotate(ctx, params->d_pattern, params->cos_a, params->sin_a);
        if (rc == VSLAM_OK)
            rc = vs_launch_gaussian7(ctx, gray, frames, width, height, blur);                            // ORB::compute
        ctx->stream = main_stream;
        if (rc) return rc;
        if (overlap) {
            VS_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
            VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    }
    if ((rc = vs_launch_orb_describe(ctx, blur, frames, width, height, xy_det, n_det, kp_stride,         // :68-72
                                     params->cos_a, params->sin_a, params->d_pattern, d_xy, d_desc, d_n)))
        return rc;
    if (d_nodes)
        if ((rc = vs_launch_kdtree_build(ctx, d_xy, d_n, frames, kp_stride, d_nodes))) return rc;       // :76
    return VSLAM_OK;
}

// extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51
int vslam_extract_features_grid(vslam_ctx *ctx, uint8_t *d_bgr, int frames, int width, int height, int row_stride,
                                int nrows, int ncols, const int8_t *d_pattern, int kp_stride, float *d_xy,
                                uint8_t *d_desc, float *d_angle_octave, int32_t *d_n) {
    if (!ctx) return VSLAM_ERR_INVALID;
    if (!d_pattern)
        if (int rc = vs_default_pattern(ctx, &d_pattern)) return rc;
    return vs_launch_extract_grid(ctx, d_bgr, frames, width, height, row_stride, nrows, ncols, d_pattern, kp_stride,
                                  d_xy, d_desc, d_angle_octave, d_n);
}

// triangulate(p1, p2, c1, c2, points_4d), src/helpers.cpp:37-80, as declared (any camera matrices, n point pairs)
int vslam_triangulate_points(vslam_ctx *ctx, const float *d_p1, const float *d_p2, int n, const float *h_c1,
                             const float *h_c2, float *d_points4d) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_triangulate_points(ctx, d_p1, d_p2, n, h_c1, h_c2, d_points4d);
}

// extract_Rt + camera matrix, src/helpers.cpp:3-35, src/vslam.cpp:83-85,125
int vslam_extract_Rt(vslam_ctx *ctx, const float *d_F, const int32_t *d_best, int batch, const float *h_K, float *d_R,
                     float *d_t, float *d_c2) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_extract_Rt(ctx, d_F, d_best, batch, h_K, d_R, d_t, d_c2);
}

// triangulate, src/helpers.cpp:37-80
int vslam_triangulate(vslam_ctx *ctx, const float *d_xy1, const float *d_xy2, const int32_t *d_matches,
                      const int32_t *d_best, int batch, int kp_stride, const float *h_K, const float *d_c2,
                      float *d_points4d) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_triangulate(ctx, d_xy1, d_xy2, d_matches, d_best, batch, kp_stride, h_K, d_c2, d_points4d);
}

// reprojection-error filter, src/vslam.cpp:192-251
int vslam_reprojection_filter(vslam_ctx *ctx, const float *d_points4d, const float *d_xy1, const float *d_xy2,
                              const int32_t *d_matches, const int32_t *d_best, int batch, int kp_stride, const float *h_K,
                              const float *d_c2, const int32_t *d_map_point_ids, float threshold_sq, int32_t *d_inlier_idx,
                              int32_t *d_n_inliers, double *d_error) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_reproj_filter(ctx, d_points4d, d_xy1, d_xy2, d_matches, d_best, batch, kp_stride, h_K, d_c2,
                                   d_map_point_ids, threshold_sq, d_inlier_idx, d_n_inliers, d_error);
}

// map association, src/vslam.cpp:129-161 + orb_distance (src/PointMap.cpp:36-46)
int vslam_associate_map_points(vslam_ctx *ctx, const float *d_map_points, const int32_t *d_n_map, int batch, int map_stride,
                               const float *d_c2, int img_w, int img_h, const int32_t *d_nodes, const float *d_xy,
                               const uint8_t *d_desc, const int32_t *d_n, int kp_stride, const int32_t *d_obs_offsets,
                               const uint8_t *d_obs_desc, int obs_stride, float radius, uint32_t dist_threshold,
                               int32_t *d_map_point_ids, int32_t *d_claim) {
    if (!ctx) return VSLAM_ERR_INVALID;
    return vs_launch_associate(ctx, d_map_points, d_n_map, batch, map_stride, d_c2, img_w, img_h, d_nodes, d_xy, d_desc, d_n,
                               kp_stride, d_obs_offsets, d_obs_desc, obs_stride, radius, dist_threshold, d_map_point_ids,
                               d_claim);
}

// match_features, src/Frame.cpp:82-105
int vslam_match_features(vslam_ctx *ctx, const float *d_xy1, const uint8_t *d_desc1,
                         const int32_t *d_n1, const float *d_xy2, const uint8_t *d_desc2,
                         const int32_t *d_n2, int batch, int kp_stride, const uint32_t *d_seeds,
                         int hyp, float threshold, int32_t *d_matches, int32_t *d_best, float *d_F,
                         int32_t *d_prelim_m) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, d_xy1 && d_desc1 && d_n1 && d_xy2 && d_desc2 && d_n2 && d_seeds, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, d_matches && d_best && d_F, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && kp_stride > 0 && hyp > 0, VSLAM_ERR_INVALID);
    int32_t *pairs = nullptr, *m = nullptr, *sets = nullptr, *hyp_count = nullptr;
    uint32_t *draws = nullptr;
    float *hypF = nullptr, *hyp_sum = nullptr;
    uint8_t *mask = nullptr;
    int rc;
    const size_t bk = (size_t)batch * kp_stride, bh = (size_t)batch * hyp;
    if ((rc = vs_arena_get(ctx, "mf.pairs", sizeof(int32_t) * 2 * bk, (void **)&pairs))) return rc;
    if (d_prelim_m) m = d_prelim_m;
    else if ((rc = vs_arena_get(ctx, "mf.m", sizeof(int32_t) * (size_t)batch, (void **)&m))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.sets", sizeof(int32_t) * 8 * bh, (void **)&sets))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.draws", sizeof(uint32_t) * 8 * bh, (void **)&draws))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.hypF", sizeof(float) * 9 * bh, (void **)&hypF))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.hyp_count", sizeof(int32_t) * bh, (void **)&hyp_count))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.hyp_sum", sizeof(float) * bh, (void **)&hyp_sum))) return rc;
    if ((rc = vs_arena_get(ctx, "mf.mask", bk, (void **)&mask))) return rc;

    if ((rc = vs_launch_match(ctx, d_desc1, d_n1, d_desc2, d_n2, batch, kp_stride, pairs, m, nullptr))) return rc;
    if ((rc = vs_aux_job_point(ctx, 1))) return rc;
    if (ctx->raw_seeds == d_seeds && ctx->raw_batch == batch && ctx->raw_hyp == hyp) {
        // the raw generator outputs were produced ahead of time (vs_sets_prefetch): only the mapping is left
        uint32_t *raw = nullptr;
        if ((rc = vs_arena_get(ctx, "mf.raw", sizeof(uint32_t) * vs_ransac_raw_words(hyp) * (size_t)batch, (void **)&raw))) return rc;
        ctx->raw_seeds = nullptr;
        VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_raw, 0));
        if ((rc = vs_launch_ransac_map(ctx, m, batch, hyp, raw, sets, draws))) return rc;
    } else if ((rc = vs_launch_ransac_sets(ctx, d_seeds, m, batch, hyp, sets, draws))) {
        return rc;
    }
    if ((rc = vs_aux_job_point(ctx, 2))) return rc;
    return vs_launch_ransac(ctx, d_xy1, d_xy2, pairs, m, sets, batch, kp_stride, hyp, threshold, d_F, mask,
                            d_best, d_matches, hypF, hyp_count, hyp_sum);
}

// The mt19937 outputs RANSAC will draw its sets from depend on the seeds alone: generate them on the auxiliary stream
// while the frames are being extracted.  vslam_match_features picks them up when called with the same (seeds, batch, hyp).
static int vs_sets_prefetch(vslam_ctx *ctx, const uint32_t *d_seeds, int batch, int hyp) {
    ctx->raw_seeds = nullptr;
    if (!d_seeds || batch <= 0 || hyp <= 0) return VSLAM_OK;   // the entry point proper reports bad arguments
    if (!ctx->sets_prefetch) return VSLAM_OK;                  // the generator then runs in line, in front of the mapping
    uint32_t *raw = nullptr;
    int rc = vs_arena_get(ctx, "mf.raw", sizeof(uint32_t) * vs_ransac_raw_words(hyp) * (size_t)batch, (void **)&raw);
    if (rc) return rc;
    hipStream_t main_stream = ctx->stream;
    if (!ctx->prof) {   // everything queued so far (the previous step's mapping kernel reads `raw`) comes first
        VS_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        VS_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->aux_stream;
    }
    rc = vs_launch_ransac_mt(ctx, d_seeds, batch, hyp, raw);
    ctx->stream = main_stream;
    if (rc) return rc;
    VS_HIP(ctx, hipEventRecord(ctx->ev_raw, ctx->prof ? ctx->stream : ctx->aux_stream));
    ctx->raw_seeds = d_seeds;
    ctx->raw_batch = batch;
    ctx->raw_hyp = hyp;
    return VSLAM_OK;
}

// extract both frames of every pair, then match_features on (frame p, frame pairs + p)
int vslam_frontend_pairs(vslam_ctx *ctx, const uint8_t *d_bgr, int pairs, int width, int height,
                         int row_stride, const vslam_extract_params *params, int kp_stride,
                         const uint32_t *d_seeds, int hyp, float threshold, float *d_xy,
                         uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n, int32_t *d_matches,
                         int32_t *d_best, float *d_F) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VsPrefetchGuard prefetch_guard{ctx};
    VS_REQUIRE(ctx, pairs > 0, VSLAM_ERR_INVALID);
    // The k-d trees are an output of the path but not an input of match/RANSAC: build them on the
    // auxiliary stream beside the matching stages (fork after extraction, join at the end).  With
    // per-kernel timing on, everything stays on one stream so the event brackets are clean.
    const bool overlap = d_nodes && !ctx->prof && ctx->tree_fork != 5;   // 5: in line on the main stream, at the end of extraction
    int rc = vs_sets_prefetch(ctx, d_seeds, pairs, hyp);
    if (rc) return rc;
    rc = vslam_extract_features(ctx, d_bgr, 2 * pairs, width, height, row_stride, params, kp_stride,
                                    d_xy, d_desc, overlap ? nullptr : d_nodes, d_n, nullptr);
    if (rc) return rc;
    VsAuxGuard aux_guard{ctx};
    if (overlap)
        if ((rc = vs_defer_tree_build(ctx, d_xy, d_n, 2 * pairs, kp_stride, d_nodes))) return rc;
    const size_t half = (size_t)pairs * kp_stride;
    rc = vslam_match_features(ctx, d_xy, d_desc, d_n, d_xy + 2 * half, d_desc + VSLAM_DESC_BYTES * half,
                              d_n + pairs, pairs, kp_stride, d_seeds, hyp, threshold, d_matches, d_best,
                              d_F, nullptr);
    // a fork point the matching stages never passed (point 4 lies in the branch VSLAM_OPT_RANSAC_ALL_SUMS does not take): the
    // build still has to run, or the join below waits on an event of an earlier call and d_nodes stays unwritten
    if (rc == VSLAM_OK && overlap && ctx->aux_job) rc = vs_aux_job_point(ctx, ctx->aux_job_at);
    if (overlap) VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return rc;
}

// extract + match + RANSAC + extract_Rt + triangulate + reprojection filter, src/vslam.cpp:60-88,120-125,186-251
int vslam_frontend_pairs_pose(vslam_ctx *ctx, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                              const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                              float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                              int32_t *d_matches, int32_t *d_best, float *d_F, const float *h_K,
                              const int32_t *d_map_point_ids, float reproj_threshold_sq, const vslam_pose_outputs *pose) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, h_K && pose, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, pose->d_R && pose->d_t && pose->d_c2 && pose->d_points4d && pose->d_inlier_idx && pose->d_n_inliers && pose->d_error,
               VSLAM_ERR_INVALID);
    int rc = vslam_frontend_pairs(ctx, d_bgr, pairs, width, height, row_stride, params, kp_stride, d_seeds, hyp, threshold, d_xy,
                                  d_desc, d_nodes, d_n, d_matches, d_best, d_F);
    if (rc) return rc;
    const size_t half = (size_t)pairs * kp_stride;
    const float *xy1 = d_xy, *xy2 = d_xy + 2 * half;
    if (!d_map_point_ids) {   // nothing assigned yet: -1 everywhere (all bits set)
        int32_t *ids = nullptr;
        if ((rc = vs_arena_get(ctx, "pose.no_ids", sizeof(int32_t) * half, (void **)&ids))) return rc;
        VS_HIP(ctx, hipMemsetAsync(ids, 0xFF, sizeof(int32_t) * half, ctx->stream));
        d_map_point_ids = ids;
    }
    if ((rc = vs_launch_extract_Rt(ctx, d_F, d_best, pairs, h_K, pose->d_R, pose->d_t, pose->d_c2))) return rc;
    if ((rc = vs_launch_triangulate(ctx, xy1, xy2, d_matches, d_best, pairs, kp_stride, h_K, pose->d_c2, pose->d_points4d))) return rc;
    return vs_launch_reproj_filter(ctx, pose->d_points4d, xy1, xy2, d_matches, d_best, pairs, kp_stride, h_K, pose->d_c2,
                                   d_map_point_ids, reproj_threshold_sq, pose->d_inlier_idx, pose->d_n_inliers, pose->d_error);
}

// result records for the gather of the sharded path
__global__ __launch_bounds__(256) void pack_records_kernel(const float *__restrict__ F, const int32_t *__restrict__ best,
                                                           const int32_t *__restrict__ matches, int kp_stride,
                                                           int32_t *__restrict__ rec) {
    const int p = blockIdx.y;
    const int words = 13 + kp_stride;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    int32_t v;
    if (i < 9) v = __float_as_int(F[(size_t)p * 9 + i]);
    else if (i < 13) v = best[(size_t)p * 4 + (i - 9)];
    else {
        const int2 m = reinterpret_cast<const int2 *>(matches)[(size_t)p * kp_stride + (i - 13)];
        v = m.x | (m.y << 16);
    }
    rec[(size_t)p * words + i] = v;
}

int vslam_pack_records(vslam_ctx *ctx, const float *d_F, const int32_t *d_best, const int32_t *d_matches,
                       int pairs, int kp_stride, int32_t *d_records) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VS_REQUIRE(ctx, d_F && d_best && d_matches && d_records, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, pairs > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    dim3 grid(vs_div_up(13 + kp_stride, 256), pairs);
    pack_records_kernel<<<grid, 256, 0, ctx->stream>>>(d_F, d_best, d_matches, kp_stride, d_records);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// consecutive frames: extract once, pair i = (frame i, frame i + 1)
int vslam_frontend_sequence(vslam_ctx *ctx, const uint8_t *d_bgr, int frames, int width, int height,
                            int row_stride, const vslam_extract_params *params, int kp_stride,
                            const uint32_t *d_seeds, int hyp, float threshold, float *d_xy,
                            uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n, int32_t *d_matches,
                            int32_t *d_best, float *d_F) {
    if (!ctx) return VSLAM_ERR_INVALID;
    VsPrefetchGuard prefetch_guard{ctx};
    VS_REQUIRE(ctx, frames >= 2, VSLAM_ERR_INVALID);
    const bool overlap = d_nodes && !ctx->prof && ctx->tree_fork != 5;   // k-d trees beside the matching stages, as in vslam_frontend_pairs
    int rc = vs_sets_prefetch(ctx, d_seeds, frames - 1, hyp);
    if (rc) return rc;
    rc = vslam_extract_features(ctx, d_bgr, frames, width, height, row_stride, params, kp_stride, d_xy, d_desc,
                                    overlap ? nullptr : d_nodes, d_n, nullptr);
    if (rc) return rc;
    VsAuxGuard aux_guard{ctx};
    if (overlap)
        if ((rc = vs_defer_tree_build(ctx, d_xy, d_n, frames, kp_stride, d_nodes))) return rc;
    const size_t one = (size_t)kp_stride;
    rc = vslam_match_features(ctx, d_xy, d_desc, d_n, d_xy + 2 * one, d_desc + VSLAM_DESC_BYTES * one, d_n + 1,
                              frames - 1, kp_stride, d_seeds, hyp, threshold, d_matches, d_best, d_F, nullptr);
    if (rc == VSLAM_OK && overlap && ctx->aux_job) rc = vs_aux_job_point(ctx, ctx->aux_job_at);   // see vslam_frontend_pairs
    if (overlap) VS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return rc;
}

}  // extern "C"

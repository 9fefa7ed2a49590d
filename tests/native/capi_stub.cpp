// TEST INFRASTRUCTURE: a CPU stand-in for the dozen C-ABI entry points vslam_amd/host/ingest.cpp calls, so that the capture
// loop's host-side threading (reader pool, page-locked double buffer, upload / compute / refill hand-over, the per-device
// slot threads of run_sequence_devices) can run under ThreadSanitizer and AddressSanitizer on a box without a GPU
// (tests/test_sanitizers.py).  Never part of the product: libvslam_host.so links libvslam_amd.so and nothing else.
//
// What it keeps of the real thing is the ASYNCHRONY that makes the loop's hand-over matter: vslam_upload_async returns at
// once and a worker thread copies later (so a host buffer refilled too early is a data race TSan sees, and a wrong record);
// vslam_frontend_sequence runs on the same worker, behind the uploads it was fenced on.  What it "computes" is a checksum
// of each frame pair, which the driver recomputes from the file.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/vslam_amd.h"

struct vslam_ctx {
    std::string err;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> copy_q, compute_q;   // the two "streams"
    int copy_busy = 0, compute_busy = 0;
    uint64_t copy_issued = 0, copy_done = 0, compute_waits_for = 0;
    bool stop = false;
    bool whole_image_lists = false;
    std::thread copy_thread, compute_thread;
};

namespace {
void copy_loop(vslam_ctx *c) {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop || !c->copy_q.empty(); });
            if (c->copy_q.empty()) return;
            job = std::move(c->copy_q.front());
            c->copy_q.pop_front();
            c->copy_busy = 1;
        }
        job();
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->copy_busy = 0;
            c->copy_done++;
        }
        c->cv.notify_all();
    }
}
void compute_loop(vslam_ctx *c) {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop || (!c->compute_q.empty() && c->copy_done >= c->compute_waits_for); });
            if (c->compute_q.empty()) return;
            job = std::move(c->compute_q.front());
            c->compute_q.pop_front();
            c->compute_busy = 1;
        }
        job();
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->compute_busy = 0;
        }
        c->cv.notify_all();
    }
}
void compute_sync(vslam_ctx *c) {
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv.wait(lk, [&] { return c->compute_q.empty() && !c->compute_busy; });
}
uint32_t frame_hash(const uint8_t *p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 16777619u;
    return h;
}
}  // namespace

extern "C" {

// what the stub "computes" for the pair (frame a, frame b) with seed s: exported so that the driver can predict the records
uint32_t vslam_stub_pair_value(const uint8_t *a, const uint8_t *b, size_t frame_bytes, uint32_t seed) {
    return frame_hash(a, frame_bytes) * 31u + frame_hash(b, frame_bytes) * 7u + seed;
}

int vslam_ctx_create(int device, vslam_ctx **out) {
    if (!out || device < 0 || device > 7) return VSLAM_ERR_INVALID;
    auto *c = new vslam_ctx();
    c->copy_thread = std::thread(copy_loop, c);
    c->compute_thread = std::thread(compute_loop, c);
    *out = c;
    return VSLAM_OK;
}
int vslam_ctx_destroy(vslam_ctx *c) {
    if (!c) return VSLAM_ERR_INVALID;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv.notify_all();
    c->copy_thread.join();
    c->compute_thread.join();
    delete c;
    return VSLAM_OK;
}
int vslam_ctx_make_current(vslam_ctx *c) { return c ? VSLAM_OK : VSLAM_ERR_INVALID; }
// Every `period`-th batch status reads VSLAM_ERR_CAPACITY while the corner lists are bounded (the driver sets the period):
// the capture loop must then repeat the batch with VSLAM_OPT_CORNER_LIST_CAP = -1, under the same sanitizers.
static std::atomic<int> g_capacity_period{0}, g_status_calls{0};
void vslam_stub_set_capacity_period(int period) { g_capacity_period = period; }
int vslam_ctx_set_option(vslam_ctx *c, int option, int value) {
    if (!c) return VSLAM_ERR_INVALID;
    if (option == VSLAM_OPT_CORNER_LIST_CAP) {
        std::lock_guard<std::mutex> lk(c->mu);
        c->whole_image_lists = value == -1;
    }
    return VSLAM_OK;
}
int vslam_ctx_synchronize(vslam_ctx *c) {
    if (!c) return VSLAM_ERR_INVALID;
    compute_sync(c);
    const int period = g_capacity_period.load();
    bool bounded;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        bounded = !c->whole_image_lists;
    }
    if (period > 0 && bounded && (++g_status_calls % period) == 0) {
        c->err = "stub: corner pool exhausted";
        return VSLAM_ERR_CAPACITY;
    }
    return VSLAM_OK;
}
const char *vslam_last_error(vslam_ctx *c) { return c ? c->err.c_str() : "null context"; }
int vslam_host_alloc(vslam_ctx *, size_t bytes, void **h_out) { return (*h_out = std::malloc(bytes ? bytes : 1)) ? VSLAM_OK : VSLAM_ERR_HIP; }
int vslam_host_free(vslam_ctx *, void *p) {
    std::free(p);
    return VSLAM_OK;
}
int vslam_dev_alloc(vslam_ctx *, size_t bytes, void **d_out) { return (*d_out = std::malloc(bytes ? bytes : 1)) ? VSLAM_OK : VSLAM_ERR_HIP; }
int vslam_dev_free(vslam_ctx *, void *p) {
    std::free(p);
    return VSLAM_OK;
}
// synchronous copies on the compute stream, as in the library
int vslam_copy_h2d(vslam_ctx *c, void *d, const void *h, size_t n) {
    compute_sync(c);
    std::memcpy(d, h, n);
    return VSLAM_OK;
}
int vslam_copy_d2h(vslam_ctx *c, void *h, const void *d, size_t n) {
    compute_sync(c);
    std::memcpy(h, d, n);
    return VSLAM_OK;
}
int vslam_upload_async(vslam_ctx *c, void *d, const void *h, size_t n) {
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->copy_q.push_back([=] { std::memcpy(d, h, n); });
        c->copy_issued++;
    }
    c->cv.notify_all();
    return VSLAM_OK;
}
int vslam_upload_fence(vslam_ctx *c) {   // later compute work waits for every upload issued so far
    std::lock_guard<std::mutex> lk(c->mu);
    c->compute_waits_for = c->copy_issued;
    return VSLAM_OK;
}
int vslam_upload_wait(vslam_ctx *c) {
    std::unique_lock<std::mutex> lk(c->mu);
    const uint64_t want = c->copy_issued;
    c->cv.wait(lk, [&] { return c->copy_done >= want; });
    return VSLAM_OK;
}
int vslam_frontend_sequence(vslam_ctx *c, const uint8_t *d_bgr, int frames, int width, int height, int row_stride,
                            const vslam_extract_params *, int kp_stride, const uint32_t *d_seeds, int, float, float *, uint8_t *,
                            int32_t *, int32_t *d_n, int32_t *d_matches, int32_t *d_best, float *d_F) {
    if (!c || frames < 2) return VSLAM_ERR_INVALID;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->compute_q.push_back([=] {
            const size_t fb = (size_t)height * row_stride;
            (void)width;
            for (int i = 0; i < frames; i++) d_n[i] = kp_stride;
            for (int i = 0; i + 1 < frames; i++) {
                const uint32_t v = vslam_stub_pair_value(d_bgr + fb * (size_t)i, d_bgr + fb * (size_t)(i + 1), fb, d_seeds[i]);
                const int n = (int)(v % (uint32_t)(kp_stride + 1));
                d_best[4 * i + 0] = (int32_t)(v % 5u) - 1;   // winner -1 now and then
                d_best[4 * i + 1] = n;
                d_best[4 * i + 2] = (int32_t)(v >> 3);
                d_best[4 * i + 3] = n;
                for (int k = 0; k < 9; k++) d_F[9 * i + k] = (float)((v >> k) & 1023u);
                for (int j = 0; j < n; j++) {
                    d_matches[2 * ((size_t)i * kp_stride + j)] = (int32_t)((v + (uint32_t)j) % (uint32_t)kp_stride);
                    d_matches[2 * ((size_t)i * kp_stride + j) + 1] = (int32_t)((v * 3u + (uint32_t)j) % (uint32_t)kp_stride);
                }
            }
        });
    }
    c->cv.notify_all();
    return VSLAM_OK;
}

}  // extern "C"

// Batches in flight on ONE device behind the C ABI (include/vslam_amd.h: vslam_pipeline_*).
//
// The reference's capture loop (src/vslam.cpp:53-77) handles one frame pair from first to last call before it looks at the
// next.  On the device a batch goes through stages of very different shape: kernels that fill the chip (the detector, the
// 8-point solves) and kernels of one workgroup per frame or pair that leave most of it idle (selection, the closing RANSAC
// stages).  Every entry point is stream-ordered and a context owns its streams and workspaces, so the idle parts of one
// batch can be filled by the arithmetic of another: k contexts, batches handed to them round-robin, each batch's outputs
// in buffers the caller owns.  Measured at the headline shape, SURVEY 8(d) data: 2.90 ms per batch on one context, 2.61 with four in flight.
//
// A ticket is one batch.  acquire() hands out the next context (waiting for the batch that used it k tickets ago, which
// bounds the queue at k batches), the caller enqueues whatever belongs to the batch on it, commit() closes the batch: the
// context's device-side error word is copied to a page-locked word of the slot and cleared IN STREAM ORDER, then an event
// is recorded.  So a batch's status is its own: an overflow in batch t is reported by wait(t) and nowhere else, and the
// batches before and behind it on the same context are untouched.  Host code only; no kernel lives here.
#include "ctx.h"

#include <algorithm>
#include <deque>
#include <mutex>

namespace {
// what vslam_pipeline_submit_pairs / _sequence queued for a ticket: enough to queue it again (the caller's buffers stay valid
// until the ticket has been waited for)
struct Submitted {
    int kind = 0;   // 0: nothing recorded (a ticket built with acquire / commit), 1: pairs, 2: sequence, 3: pairs + pose chain
    const uint8_t *d_bgr = nullptr;
    int count = 0, width = 0, height = 0, row_stride = 0, kp_stride = 0, hyp = 0;
    vslam_extract_params params{};
    const uint32_t *d_seeds = nullptr;
    float threshold = 0.f;
    float *d_xy = nullptr;
    uint8_t *d_desc = nullptr;
    int32_t *d_nodes = nullptr, *d_n = nullptr, *d_matches = nullptr, *d_best = nullptr, *d_records = nullptr;
    float *d_F = nullptr;
    // kind 3 (vslam_frontend_pairs_pose)
    float K[9] = {};
    const int32_t *d_map_point_ids = nullptr;
    float reproj_threshold_sq = 0.f;
    vslam_pose_outputs pose{};
};
struct Slot {
    vslam_ctx *ctx = nullptr;
    hipEvent_t done = nullptr;
    int32_t *h_flag = nullptr;   // page-locked: the batch's copy of the context's error word
    int32_t *d_flag = nullptr;
    int64_t ticket = -1;         // batch in flight on this slot (committed, not yet retired); -1: none
    bool open = false;           // acquired, not yet committed
    Submitted sub;               // of the batch in flight, when it came through submit_*
};
struct Failure {
    int64_t ticket;
    int rc;
    std::string err;
};
}  // namespace

struct vslam_pipeline {
    int device = 0;
    std::vector<Slot> slots;
    int64_t next_ticket = 0;
    int64_t redone = 0;             // batches queued a second time with whole-image corner lists (see retire)
    std::deque<Failure> failures;   // retired batches that failed and have not been asked about yet (bounded)
    std::string err;
    std::mutex mu;
};

std::string vs_errflag_message(int32_t flag);   // capi.hip

namespace {
constexpr size_t kMaxFailures = 256;

// wait for the batch in flight on `s` (if any) and file its status
int retire(vslam_pipeline *p, Slot &s) {
    if (s.ticket < 0) return VSLAM_OK;
    const int64_t t = s.ticket;
    s.ticket = -1;
    int rc = VSLAM_OK;
    std::string err;
    const hipError_t e = hipEventSynchronize(s.done);
    if (e != hipSuccess) {
        rc = VSLAM_ERR_HIP;
        err = std::string("hipEventSynchronize: ") + hipGetErrorString(e);
    } else if (*s.h_flag) {
        rc = VSLAM_ERR_CAPACITY;
        err = vs_errflag_message(*s.h_flag);
        // The one overflow a well-formed batch can meet: more of its frames needed the corner detector's whole-image fallback
        // than the pool holds (plateaus, pure noise), and those frames came back without corners.  A batch that came through
        // submit_* is queued ONCE more, here, with every list sized for the whole image (VSLAM_OPT_CORNER_LIST_CAP = -1: nothing
        // can overflow; 16 bytes per pixel and frame of workspace for this context from now on) -- its outputs are then what
        // an unbounded run gives, in the caller's same buffers -- and only what that second run reports is filed.  (The context
        // is idle at this point: its only batch in flight has just completed.)
        if ((*s.h_flag & 4) && s.sub.kind != 0 && s.ctx->corner_list_cap != -1) {
            const Submitted &q = s.sub;
            const int cap = s.ctx->corner_list_cap;
            s.ctx->corner_list_cap = -1;
            int rc2 = q.kind == 1 ? vslam_frontend_pairs(s.ctx, q.d_bgr, q.count, q.width, q.height, q.row_stride, &q.params, q.kp_stride,
                                                          q.d_seeds, q.hyp, q.threshold, q.d_xy, q.d_desc, q.d_nodes, q.d_n, q.d_matches,
                                                          q.d_best, q.d_F)
                      : q.kind == 3 ? vslam_frontend_pairs_pose(s.ctx, q.d_bgr, q.count, q.width, q.height, q.row_stride, &q.params,
                                                                q.kp_stride, q.d_seeds, q.hyp, q.threshold, q.d_xy, q.d_desc, q.d_nodes,
                                                                q.d_n, q.d_matches, q.d_best, q.d_F, q.K, q.d_map_point_ids,
                                                                q.reproj_threshold_sq, &q.pose)
                                  : vslam_frontend_sequence(s.ctx, q.d_bgr, q.count, q.width, q.height, q.row_stride, &q.params,
                                                            q.kp_stride, q.d_seeds, q.hyp, q.threshold, q.d_xy, q.d_desc, q.d_nodes, q.d_n,
                                                            q.d_matches, q.d_best, q.d_F);
            if (rc2 == VSLAM_OK && q.d_records)
                rc2 = vslam_pack_records(s.ctx, q.d_F, q.d_best, q.d_matches, q.kind == 2 ? q.count - 1 : q.count, q.kp_stride, q.d_records);
            if (rc2 == VSLAM_OK) rc2 = vslam_ctx_synchronize(s.ctx);   // waits, reads and clears the word: the context is this slot's alone
            else (void)vslam_ctx_wait(s.ctx);
            s.ctx->corner_list_cap = cap;
            p->redone++;
            if (rc2 == VSLAM_OK) {
                rc = VSLAM_OK;
            } else {
                rc = rc2;
                err = std::string("queued again with whole-image corner lists: ") + vslam_last_error(s.ctx);
            }
        }
    }
    s.sub.kind = 0;
    if (rc != VSLAM_OK) {
        if (p->failures.size() >= kMaxFailures) p->failures.pop_front();
        p->failures.push_back({t, rc, err});
    }
    return VSLAM_OK;
}

void remember(vslam_pipeline *p, int64_t ticket, int kind, const uint8_t *d_bgr, int count, int width, int height, int row_stride,
              const vslam_extract_params &params, int kp_stride, const uint32_t *d_seeds, int hyp, float threshold, float *d_xy,
              uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n, int32_t *d_matches, int32_t *d_best, float *d_F, int32_t *d_records) {
    Submitted &q = p->slots[(size_t)(ticket % (int64_t)p->slots.size())].sub;
    q.kind = kind; q.d_bgr = d_bgr; q.count = count; q.width = width; q.height = height; q.row_stride = row_stride;
    q.params = params; q.kp_stride = kp_stride; q.d_seeds = d_seeds; q.hyp = hyp; q.threshold = threshold;
    q.d_xy = d_xy; q.d_desc = d_desc; q.d_nodes = d_nodes; q.d_n = d_n; q.d_matches = d_matches; q.d_best = d_best; q.d_F = d_F;
    q.d_records = d_records;
}

int acquire_locked(vslam_pipeline *p, vslam_ctx **ctx_out, int64_t *ticket_out) {
    const int n = (int)p->slots.size();
    Slot &s = p->slots[(size_t)(p->next_ticket % n)];
    if (s.open) {
        p->err = "vslam_pipeline_acquire: the previous ticket of this slot was never committed";
        return VSLAM_ERR_INVALID;
    }
    if (hipSetDevice(p->device) != hipSuccess) {
        p->err = "hipSetDevice failed";
        return VSLAM_ERR_HIP;
    }
    retire(p, s);
    s.open = true;
    s.sub.kind = 0;
    *ctx_out = s.ctx;
    *ticket_out = p->next_ticket++;
    return VSLAM_OK;
}

int commit_locked(vslam_pipeline *p, int64_t ticket) {
    const int n = (int)p->slots.size();
    if (ticket < 0 || ticket >= p->next_ticket) {
        p->err = "vslam_pipeline_commit: no such ticket";
        return VSLAM_ERR_INVALID;
    }
    Slot &s = p->slots[(size_t)(ticket % n)];
    if (!s.open || ticket + n < p->next_ticket) {
        p->err = "vslam_pipeline_commit: ticket is not open";
        return VSLAM_ERR_INVALID;
    }
    s.open = false;
    s.ticket = ticket;
    vslam_ctx *c = s.ctx;
    // the batch's own copy of the error word, then a clean word for the next batch -- both behind the batch's last kernel
    hipError_t e = hipMemcpyAsync(s.h_flag, s.d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s.d_flag, 0, sizeof(int32_t), c->stream);
    if (e == hipSuccess) e = hipEventRecord(s.done, c->stream);
    if (e != hipSuccess) {
        // The slot is already marked in flight.  Without its event it must not look finished while kernels may still run, and
        // the ticket must not report a stale flag: drain the stream here and file the failure under the ticket.
        p->err = std::string("vslam_pipeline_commit: ") + hipGetErrorString(e);
        (void)hipStreamSynchronize(c->stream);
        (void)hipEventRecord(s.done, c->stream);
        *s.h_flag = 0;
        if (p->failures.size() >= kMaxFailures) p->failures.pop_front();
        p->failures.push_back({ticket, VSLAM_ERR_HIP, p->err});
        s.ticket = -1;   // drained and filed: nothing of this batch is in flight any more
        return VSLAM_ERR_HIP;
    }
    return VSLAM_OK;
}
}  // namespace

extern "C" {

int vslam_pipeline_create(int device, int n_ctx, vslam_pipeline **out) {
    if (!out || n_ctx < 1 || n_ctx > 16) return VSLAM_ERR_INVALID;
    *out = nullptr;
    auto *p = new vslam_pipeline();
    p->device = device;
    p->slots.resize((size_t)n_ctx);
    int rc = VSLAM_OK;
    for (int i = 0; i < n_ctx && rc == VSLAM_OK; i++) {
        Slot &s = p->slots[(size_t)i];
        rc = vs_ctx_create(device, n_ctx > 1, &s.ctx);   // with company: the arrangement for a shared chip (capi.hip)
        if (rc) break;
        if ((rc = vs_device_errflag(s.ctx, &s.d_flag))) break;
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void **)&s.h_flag, sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
            rc = VSLAM_ERR_HIP;
        else
            *s.h_flag = 0;
    }
    if (rc) {
        vslam_pipeline_destroy(p);
        return rc;
    }
    *out = p;
    return VSLAM_OK;
}

int vslam_pipeline_destroy(vslam_pipeline *p) {
    if (!p) return VSLAM_ERR_INVALID;
    (void)hipSetDevice(p->device);
    for (Slot &s : p->slots) {
        if (s.ctx) (void)hipStreamSynchronize(s.ctx->stream);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.h_flag) (void)hipHostFree(s.h_flag);
        if (s.ctx) vslam_ctx_destroy(s.ctx);
    }
    delete p;
    return VSLAM_OK;
}

int vslam_pipeline_size(const vslam_pipeline *p) { return p ? (int)p->slots.size() : 0; }

vslam_ctx *vslam_pipeline_ctx(vslam_pipeline *p, int slot) {
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? p->slots[(size_t)slot].ctx : nullptr;
}

const char *vslam_pipeline_last_error(vslam_pipeline *p) { return p ? p->err.c_str() : "null pipeline"; }

int vslam_pipeline_set_option(vslam_pipeline *p, int option, int value) {
    if (!p) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    for (Slot &s : p->slots) {
        const int rc = vslam_ctx_set_option(s.ctx, option, value);
        if (rc) {
            p->err = vslam_last_error(s.ctx);
            return rc;
        }
    }
    return VSLAM_OK;
}

int vslam_pipeline_acquire(vslam_pipeline *p, vslam_ctx **ctx_out, int64_t *ticket_out) {
    if (!p || !ctx_out || !ticket_out) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    return acquire_locked(p, ctx_out, ticket_out);
}

int vslam_pipeline_commit(vslam_pipeline *p, int64_t ticket) {
    if (!p) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    return commit_locked(p, ticket);
}

int vslam_pipeline_submit_pairs(vslam_pipeline *p, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                                const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                int32_t *d_matches, int32_t *d_best, float *d_F, int32_t *d_records, int64_t *ticket_out) {
    if (!p || !ticket_out) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    *ticket_out = -1;
    vslam_ctx *c = nullptr;
    int64_t t = -1;
    int rc = acquire_locked(p, &c, &t);
    if (rc) return rc;
    rc = vslam_frontend_pairs(c, d_bgr, pairs, width, height, row_stride, params, kp_stride, d_seeds, hyp, threshold, d_xy,
                              d_desc, d_nodes, d_n, d_matches, d_best, d_F);
    if (rc == VSLAM_OK && d_records) rc = vslam_pack_records(c, d_F, d_best, d_matches, pairs, kp_stride, d_records);
    if (rc == VSLAM_OK && params)
        remember(p, t, 1, d_bgr, pairs, width, height, row_stride, *params, kp_stride, d_seeds, hyp, threshold, d_xy, d_desc, d_nodes, d_n,
                 d_matches, d_best, d_F, d_records);
    // close the batch either way: whatever part of it was queued has to be waited for before the slot is used again (the
    // caller has the error in hand: it is not filed under the ticket as well)
    const int crc = commit_locked(p, t);
    if (rc) {
        p->err = std::string("ticket ") + std::to_string(t) + ": " + vslam_last_error(c);
        return rc;
    }
    if (crc) return crc;
    *ticket_out = t;
    return VSLAM_OK;
}

// vslam_frontend_pairs_pose as a ticket: the step + extract_Rt + triangulate + reprojection filter on the slot's context; a batch
// that exhausts the corner pool is queued once more like a submit_pairs batch, pose stages included.
int vslam_pipeline_submit_pairs_pose(vslam_pipeline *p, const uint8_t *d_bgr, int pairs, int width, int height, int row_stride,
                                     const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                     float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                     int32_t *d_matches, int32_t *d_best, float *d_F, const float *h_K,
                                     const int32_t *d_map_point_ids, float reproj_threshold_sq, const vslam_pose_outputs *pose,
                                     int32_t *d_records, int64_t *ticket_out) {
    if (!p || !ticket_out || !h_K || !pose) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    *ticket_out = -1;
    vslam_ctx *c = nullptr;
    int64_t t = -1;
    int rc = acquire_locked(p, &c, &t);
    if (rc) return rc;
    rc = vslam_frontend_pairs_pose(c, d_bgr, pairs, width, height, row_stride, params, kp_stride, d_seeds, hyp, threshold, d_xy,
                                   d_desc, d_nodes, d_n, d_matches, d_best, d_F, h_K, d_map_point_ids, reproj_threshold_sq, pose);
    if (rc == VSLAM_OK && d_records) rc = vslam_pack_records(c, d_F, d_best, d_matches, pairs, kp_stride, d_records);
    if (rc == VSLAM_OK && params) {
        remember(p, t, 3, d_bgr, pairs, width, height, row_stride, *params, kp_stride, d_seeds, hyp, threshold, d_xy, d_desc, d_nodes, d_n,
                 d_matches, d_best, d_F, d_records);
        Submitted &q = p->slots[(size_t)(t % (int64_t)p->slots.size())].sub;
        for (int i = 0; i < 9; i++) q.K[i] = h_K[i];   // the caller's array need not outlive the call
        q.d_map_point_ids = d_map_point_ids;
        q.reproj_threshold_sq = reproj_threshold_sq;
        q.pose = *pose;
    }
    const int crc = commit_locked(p, t);   // (as in vslam_pipeline_submit_pairs)
    if (rc) {
        p->err = std::string("ticket ") + std::to_string(t) + ": " + vslam_last_error(c);
        return rc;
    }
    if (crc) return crc;
    *ticket_out = t;
    return VSLAM_OK;
}

int vslam_pipeline_submit_sequence(vslam_pipeline *p, const uint8_t *d_bgr, int frames, int width, int height, int row_stride,
                                   const vslam_extract_params *params, int kp_stride, const uint32_t *d_seeds, int hyp,
                                   float threshold, float *d_xy, uint8_t *d_desc, int32_t *d_nodes, int32_t *d_n,
                                   int32_t *d_matches, int32_t *d_best, float *d_F, int32_t *d_records, int64_t *ticket_out) {
    if (!p || !ticket_out) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    *ticket_out = -1;
    vslam_ctx *c = nullptr;
    int64_t t = -1;
    int rc = acquire_locked(p, &c, &t);
    if (rc) return rc;
    rc = vslam_frontend_sequence(c, d_bgr, frames, width, height, row_stride, params, kp_stride, d_seeds, hyp, threshold, d_xy,
                                 d_desc, d_nodes, d_n, d_matches, d_best, d_F);
    if (rc == VSLAM_OK && d_records) rc = vslam_pack_records(c, d_F, d_best, d_matches, frames - 1, kp_stride, d_records);
    if (rc == VSLAM_OK && params)
        remember(p, t, 2, d_bgr, frames, width, height, row_stride, *params, kp_stride, d_seeds, hyp, threshold, d_xy, d_desc, d_nodes, d_n,
                 d_matches, d_best, d_F, d_records);
    const int crc = commit_locked(p, t);
    if (rc) {
        p->err = std::string("ticket ") + std::to_string(t) + ": " + vslam_last_error(c);
        return rc;
    }
    if (crc) return crc;
    *ticket_out = t;
    return VSLAM_OK;
}

int vslam_pipeline_poll(vslam_pipeline *p, int64_t ticket) {
    if (!p) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    if (ticket < 0 || ticket >= p->next_ticket) return VSLAM_ERR_INVALID;
    Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    if (s.open && ticket + (int64_t)p->slots.size() >= p->next_ticket) return 0;   // not even committed
    if (s.ticket != ticket) return 1;                                               // retired long ago
    return hipEventQuery(s.done) == hipSuccess ? 1 : 0;
}

int vslam_pipeline_wait(vslam_pipeline *p, int64_t ticket) {
    if (!p) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    if (ticket < 0 || ticket >= p->next_ticket) {
        p->err = "vslam_pipeline_wait: no such ticket";
        return VSLAM_ERR_INVALID;
    }
    Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    if (s.open && ticket + (int64_t)p->slots.size() >= p->next_ticket) {
        p->err = "vslam_pipeline_wait: ticket was acquired but not committed";
        return VSLAM_ERR_INVALID;
    }
    if (s.ticket == ticket) {
        (void)hipSetDevice(p->device);
        retire(p, s);
    }
    auto it = std::find_if(p->failures.begin(), p->failures.end(), [&](const Failure &f) { return f.ticket == ticket; });
    if (it == p->failures.end()) return VSLAM_OK;
    const int rc = it->rc;
    p->err = "ticket " + std::to_string(ticket) + ": " + it->err;
    p->failures.erase(it);
    return rc;
}

int64_t vslam_pipeline_batches_redone(vslam_pipeline *p) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lk(p->mu);
    return p->redone;
}

int vslam_pipeline_drain(vslam_pipeline *p) {
    if (!p) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(p->mu);
    (void)hipSetDevice(p->device);
    for (Slot &s : p->slots) {
        if (s.open) {
            p->err = "vslam_pipeline_drain: a ticket is still open (acquired, not committed)";
            return VSLAM_ERR_INVALID;
        }
        retire(p, s);
    }
    if (p->failures.empty()) return VSLAM_OK;
    const Failure f = p->failures.front();
    p->err = "ticket " + std::to_string(f.ticket) + ": " + f.err +
             (p->failures.size() > 1 ? " (+" + std::to_string(p->failures.size() - 1) + " more failed batches)" : "");
    p->failures.clear();
    return f.rc;
}

}  // extern "C"

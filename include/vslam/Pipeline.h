// Batches in flight on one device, for a C++ consumer (C ABI: vslam_pipeline_*, include/vslam_amd.h).
//
// The reference's capture loop takes a frame, extracts, matches it against the previous one and only then looks at the
// next (src/vslam.cpp:53-77).  A device wants several batches queued at once: the stages of one batch that leave most of
// the chip idle are filled by the arithmetic of another (2.90 -> 2.61 ms per batch of 256 pairs at 1280x720 with four
// in flight).  vslam::Pipeline keeps `in_flight` contexts and, per context, the device and page-locked buffers of one
// batch; submit_pairs() returns at once with a ticket, collect() waits for that batch and hands back its records.
//
//   vslam::Pipeline pipe(0, 3);
//   std::deque<int64_t> q;
//   while (capture(last, cur)) {                       // host frames, [pairs][height][row_stride] BGR each
//       q.push_back(pipe.submit_pairs(last, cur, pairs, w, h, 3 * w, 3000, 100, 10.f, seed));
//       if ((int)q.size() == pipe.size()) { consume(pipe.collect(q.front())); q.pop_front(); }
//   }
//   while (!q.empty()) { consume(pipe.collect(q.front())); q.pop_front(); }
//
// A batch that fails (VSLAM_ERR_CAPACITY: more frames of it overflowed the corner detector's lists than the fallback pool
// holds) throws from ITS collect(); the batches around it are unaffected.  Header-only, one submitting thread.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vslam_amd.h"
#include "Ingest.h"

namespace vslam {

class Pipeline {
public:
    explicit Pipeline(int device = 0, int in_flight = 3) {
        if (vslam_pipeline_create(device, in_flight, &p_) != VSLAM_OK)
            throw std::runtime_error("vslam::Pipeline: cannot create the contexts (no HIP device? there is no CPU fallback)");
        slots_.resize((size_t)vslam_pipeline_size(p_));
    }
    ~Pipeline() {
        if (!p_) return;
        (void)vslam_pipeline_drain(p_);
        for (size_t i = 0; i < slots_.size(); i++) {
            vslam_ctx *c = vslam_pipeline_ctx(p_, (int)i);
            Slot &s = slots_[i];
            for (Buf *b : {&s.bgr, &s.seeds, &s.xy, &s.desc, &s.nodes, &s.n, &s.matches, &s.best, &s.F, &s.rec, &s.pattern})
                if (b->p) vslam_dev_free(c, b->p);
            if (s.h_rec) vslam_host_free(c, s.h_rec);
        }
        vslam_pipeline_destroy(p_);
    }
    Pipeline(const Pipeline &) = delete;
    Pipeline &operator=(const Pipeline &) = delete;

    int size() const { return (int)slots_.size(); }
    vslam_pipeline *handle() { return p_; }
    void set_option(int option, int value) { check(vslam_pipeline_set_option(p_, option, value)); }

    // extract_features on both frames of every pair + match_features (src/Frame.cpp:53-105) with
    // RansacFilter(8, hypotheses, threshold) seeded seed ^ (first_pair + i).  last / cur: host memory (page-locked memory
    // from vslam_host_alloc keeps the upload asynchronous), valid until collect(ticket) returns.
    int64_t submit_pairs(const uint8_t *last, const uint8_t *cur, int pairs, int width, int height, int row_stride,
                         int max_corners, int hypotheses, float threshold, uint32_t seed, uint64_t first_pair = 0,
                         const int8_t *pattern = nullptr, float keypoint_angle_deg = -1.0f) {
        if (!last || !cur || pairs <= 0) throw std::invalid_argument("vslam::Pipeline::submit_pairs: bad argument");
        vslam_ctx *c = nullptr;
        int64_t t = -1;
        check(vslam_pipeline_acquire(p_, &c, &t));   // the slot's previous batch is complete from here on
        Slot &s = slots_[(size_t)(t % (int64_t)slots_.size())];
        int rc = enqueue(c, s, last, cur, pairs, width, height, row_stride, max_corners, hypotheses, threshold, seed, first_pair,
                         pattern, keypoint_angle_deg);
        const std::string why = rc ? vslam_last_error(c) : "";
        const int crc = vslam_pipeline_commit(p_, t);   // closes whatever part of the batch was queued
        if (rc) throw std::runtime_error("vslam::Pipeline::submit_pairs: " + why);
        check(crc);
        s.ticket = t;
        s.pairs = pairs;
        s.words = 13 + (size_t)max_corners;
        s.first_pair = first_pair;
        return t;
    }

    bool done(int64_t ticket) { return vslam_pipeline_poll(p_, ticket) == 1; }

    // Waits for the batch and returns its records in pair order.  A ticket can be collected once, and before `in_flight`
    // later batches have been submitted (its slot's buffers are reused then).
    std::vector<PairRecord> collect(int64_t ticket) {
        if (ticket < 0) throw std::invalid_argument("vslam::Pipeline::collect: bad ticket");
        Slot &s = slots_[(size_t)(ticket % (int64_t)slots_.size())];
        if (s.ticket != ticket) throw std::runtime_error("vslam::Pipeline::collect: this ticket's buffers have been reused (or it was collected)");
        check(vslam_pipeline_wait(p_, ticket));
        s.ticket = -1;
        std::vector<PairRecord> out((size_t)s.pairs);
        for (int i = 0; i < s.pairs; i++) {
            const int32_t *r = s.h_rec + (size_t)i * s.words;
            PairRecord &o = out[(size_t)i];
            o.first_frame = s.first_pair + (uint64_t)i;
            std::memcpy(o.F, r, 36);
            o.winner = r[9];
            o.inliers = r[10];
            std::memcpy(&o.score, r + 11, 4);
            const int n = r[12];
            o.matches.resize((size_t)(n > 0 ? n : 0));
            for (int k = 0; k < n; k++) o.matches[(size_t)k] = {r[13 + k] & 0xFFFF, (r[13 + k] >> 16) & 0xFFFF};
            if (o.winner < 0) std::memset(o.F, 0, 36);   // nothing accepted: the device leaves F untouched (stale)
        }
        return out;
    }

private:
    struct Buf {
        void *p = nullptr;
        size_t bytes = 0;
    };
    struct Slot {
        Buf bgr, seeds, xy, desc, nodes, n, matches, best, F, rec, pattern;
        int32_t *h_rec = nullptr;
        size_t h_rec_bytes = 0;
        std::vector<uint32_t> h_seeds;
        int64_t ticket = -1;
        int pairs = 0;
        size_t words = 0;
        uint64_t first_pair = 0;
    };
    void check(int rc) {
        if (rc != VSLAM_OK) throw std::runtime_error(std::string("vslam::Pipeline: ") + vslam_pipeline_last_error(p_));
    }
    static int grow(vslam_ctx *c, Buf &b, size_t bytes) {
        if (b.bytes >= bytes) return VSLAM_OK;
        if (b.p) {
            const int rc = vslam_dev_free(c, b.p);
            if (rc) return rc;
            b.p = nullptr;
            b.bytes = 0;
        }
        const int rc = vslam_dev_alloc(c, bytes, &b.p);
        if (rc == VSLAM_OK) b.bytes = bytes;
        return rc;
    }
    int enqueue(vslam_ctx *c, Slot &s, const uint8_t *last, const uint8_t *cur, int pairs, int width, int height, int row_stride,
                int max_corners, int hypotheses, float threshold, uint32_t seed, uint64_t first_pair, const int8_t *pattern,
                float keypoint_angle_deg) {
        const size_t P = (size_t)pairs, K = (size_t)max_corners, fb = (size_t)height * row_stride, words = 13 + K;
        int rc;
        if ((rc = grow(c, s.bgr, 2 * P * fb)) || (rc = grow(c, s.seeds, 4 * P)) || (rc = grow(c, s.xy, 8 * 2 * P * K)) ||
            (rc = grow(c, s.desc, 32 * 2 * P * K)) || (rc = grow(c, s.nodes, 4 * 2 * P * K)) || (rc = grow(c, s.n, 4 * 2 * P)) ||
            (rc = grow(c, s.matches, 8 * P * K)) || (rc = grow(c, s.best, 16 * P)) || (rc = grow(c, s.F, 36 * P)) ||
            (rc = grow(c, s.rec, 4 * P * words)) || (rc = grow(c, s.pattern, 1024)))
            return rc;
        if (s.h_rec_bytes < 4 * P * words) {
            if (s.h_rec && (rc = vslam_host_free(c, s.h_rec))) return rc;
            s.h_rec = nullptr;
            s.h_rec_bytes = 0;
            if ((rc = vslam_host_alloc(c, 4 * P * words, (void **)&s.h_rec))) return rc;
            s.h_rec_bytes = 4 * P * words;
        }
        s.h_seeds.resize(P);   // lives in the slot: the upload may still be reading it when this call returns
        for (size_t i = 0; i < P; i++) s.h_seeds[i] = seed ^ (uint32_t)(first_pair + i);
        uint8_t *d_bgr = static_cast<uint8_t *>(s.bgr.p);
        vslam_extract_params p;
        p.max_corners = max_corners;
        p.quality = 0.01;        // src/Frame.cpp:61
        p.min_distance = 3.0;
        // cv::KeyPoint(p, 20) leaves angle = -1 (degrees) and ORB::compute does not recompute it for given keypoints
        const float a = keypoint_angle_deg * (float)(3.14159265358979323846 / 180.f);
        p.cos_a = (float)std::cos((double)a);
        p.sin_a = (float)std::sin((double)a);
        p.d_pattern = nullptr;
        if ((rc = vslam_upload_async(c, d_bgr, last, P * fb)) || (rc = vslam_upload_async(c, d_bgr + P * fb, cur, P * fb)) ||
            (rc = vslam_upload_async(c, s.seeds.p, s.h_seeds.data(), 4 * P)))
            return rc;
        if (pattern) {
            if ((rc = vslam_upload_async(c, s.pattern.p, pattern, 1024))) return rc;
            p.d_pattern = static_cast<const int8_t *>(s.pattern.p);
        }
        if ((rc = vslam_upload_fence(c))) return rc;
        if ((rc = vslam_frontend_pairs(c, d_bgr, pairs, width, height, row_stride, &p, max_corners, static_cast<const uint32_t *>(s.seeds.p),
                                       hypotheses, threshold, static_cast<float *>(s.xy.p), static_cast<uint8_t *>(s.desc.p),
                                       static_cast<int32_t *>(s.nodes.p), static_cast<int32_t *>(s.n.p), static_cast<int32_t *>(s.matches.p),
                                       static_cast<int32_t *>(s.best.p), static_cast<float *>(s.F.p))))
            return rc;
        if ((rc = vslam_pack_records(c, static_cast<const float *>(s.F.p), static_cast<const int32_t *>(s.best.p),
                                     static_cast<const int32_t *>(s.matches.p), pairs, max_corners, static_cast<int32_t *>(s.rec.p))))
            return rc;
        return vslam_download_async(c, s.h_rec, s.rec.p, 4 * P * words);
    }

    vslam_pipeline *p_ = nullptr;
    std::vector<Slot> slots_;
};

}  // namespace vslam

// A batch of independent frame pairs over several devices of one node.
//
// The reference runs one pair at a time on one core (src/vslam.cpp:60-77); its per-pair state is per call
// (src/RansacFilter.cpp:38), so pairs shard without any exchange: contiguous slices, one per device, per-pair seeds
// seed ^ global pair index, and the result records of all slices come back in pair order.  Header-only, on the C ABI
// (include/vslam_amd.h: vslam_multi_*): one context and one host thread per device inside the library.
//   vslam::DevicePool pool({0, 1, 2, 3, 4, 5, 6, 7});
//   std::vector<vslam::PairRecord> out = pool.frontend_pairs(last, cur, pairs, w, h, 3 * w, 3000, 100, 10.f, seed);
// For one process per device (ranks of a launcher) the same records travel through vslam_gather_records instead.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vslam_amd.h"
#include "Ingest.h"

namespace vslam {

class DevicePool {
public:
    // one entry per context; a device may be listed more than once (batches in flight on one device)
    explicit DevicePool(const std::vector<int> &devices) {
        if (vslam_multi_create(devices.data(), (int)devices.size(), &m_) != VSLAM_OK)
            throw std::runtime_error("vslam::DevicePool: cannot create a context on every listed device");
    }
    ~DevicePool() {
        if (m_) vslam_multi_destroy(m_);
    }
    DevicePool(const DevicePool &) = delete;
    DevicePool &operator=(const DevicePool &) = delete;

    int size() const { return vslam_multi_size(m_); }
    vslam_ctx *context(int i) { return vslam_multi_ctx(m_, i); }   // per-context options: vslam_ctx_set_option

    // extract_features on both frames of every pair + match_features (src/Frame.cpp:53-105) with
    // RansacFilter(8, hypotheses, threshold) seeded seed ^ pair index.  last / cur: [pairs][height][row_stride] BGR in host
    // memory.  pattern: 256 x 4 int8 rBRIEF table, nullptr for ORB's learned one.
    std::vector<PairRecord> frontend_pairs(const uint8_t *last, const uint8_t *cur, int pairs, int width, int height,
                                           int row_stride, int max_corners, int hypotheses, float threshold, uint32_t seed,
                                           const int8_t *pattern = nullptr, float keypoint_angle_deg = -1.0f) {
        vslam_extract_params p;
        p.max_corners = max_corners;
        p.quality = 0.01;        // src/Frame.cpp:61
        p.min_distance = 3.0;
        // cv::KeyPoint(p, 20) leaves angle = -1 (degrees) and ORB::compute does not recompute it for given keypoints
        const float a = keypoint_angle_deg * (float)(3.14159265358979323846 / 180.f);   // angle *= CV_PI / 180
        p.cos_a = (float)std::cos((double)a);
        p.sin_a = (float)std::sin((double)a);
        p.d_pattern = nullptr;
        const size_t words = 13 + (size_t)max_corners;
        std::vector<int32_t> rec((size_t)pairs * words);
        const int rc = vslam_multi_frontend_pairs(m_, last, cur, pairs, width, height, row_stride, &p, pattern, max_corners, seed,
                                                  hypotheses, threshold, rec.data(), nullptr);
        if (rc != VSLAM_OK) throw std::runtime_error(std::string("vslam::DevicePool: ") + vslam_multi_last_error(m_));
        std::vector<PairRecord> out((size_t)pairs);
        for (int i = 0; i < pairs; i++) {
            const int32_t *r = rec.data() + (size_t)i * words;
            PairRecord &o = out[(size_t)i];
            o.first_frame = (uint64_t)i;
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
    vslam_multi *m_ = nullptr;
};

}  // namespace vslam

// Drop-in for the reference's include/RansacFilter.h: same class, same public fields and methods,
// the hypothesis loop runs on the MI355X through include/vslam_amd.h.
#pragma once
#include <utility>
#include <vector>

#include "cvlite.h"
#include "vslam_internal.h"

namespace vslam {
namespace detail {
struct RansacAccess;   // the device-side paths in libvslam_host (match_features chains match -> sets -> RANSAC on the GPU)
}
}  // namespace vslam

class RansacFilter {
   public:
    const int min_items;        // reference: include/RansacFilter.h:12-14
    const int max_iterations;
    const float threshold;

    RansacFilter(const int min_items = 8, const int max_iterations = 100, const float threshold = 0.2);

    // reference :18, src/RansacFilter.cpp:6-34.  The reference seeds std::mt19937 from
    // std::random_device; so does this, unless a seed was fixed with set_seed().
    void initialize_sets(const int n_matches);
    // reference :19, src/RansacFilter.cpp:36-67
    void find_fundamental(const std::vector<cv::Point2f> &p1, const std::vector<cv::Point2f> &p2,
                          const std::vector<std::pair<int, int>> &matches, std::vector<bool> &inliers,
                          cv::Mat &fundamental);
    // reference :20, src/RansacFilter.cpp:69-103 (8-point sets)
    void compute_fundamental(const std::vector<cv::Point2f> &p1_set, const std::vector<cv::Point2f> &p2_set,
                             cv::Mat &temp_F);
    // reference :21, src/RansacFilter.cpp:105-140
    std::pair<int, float> compute_fundamental_residual(const std::vector<cv::Point2f> &p1,
                                                       const std::vector<cv::Point2f> &p2,
                                                       const std::vector<std::pair<int, int>> &matches,
                                                       const cv::Mat &F, std::vector<bool> &inliers);

    // addition (the reference has no seed parameter): fix the mt19937 seed of the next draws
    void set_seed(u32 seed) { seed_ = seed; has_seed_ = true; }
    const std::vector<std::vector<int>> &sets() const { return ransac_sets; }

   private:
    friend struct vslam::detail::RansacAccess;
    std::vector<std::vector<int>> ransac_sets;   // reference :24
    u32 seed_ = 0;
    bool has_seed_ = false;
    u32 next_seed();
};

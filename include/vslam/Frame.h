// Drop-in for the reference's include/Frame.h: same struct, same free functions; extraction,
// matching and RANSAC run on the MI355X through include/vslam_amd.h.
#pragma once
#include <utility>
#include <vector>

#include "KDTree.h"
#include "RansacFilter.h"
#include "cvlite.h"
#include "vslam_internal.h"

struct Frame {   // reference: include/Frame.h:11-27
    cv::Mat image;
    cv::Mat pose, R_t;
    std::vector<cv::Point2f> points;
    std::vector<s32> map_point_ids;
    cv::Mat descriptors;   // N x 32, CV_8U
    frame_kdtree kdtree;
    u64 id;
};

// reference: include/Frame.h:29, src/Frame.cpp:3-6
void initialize_frame(Frame &frame, const cv::Mat &image, long frame_id);
// reference: include/Frame.h:33, src/Frame.cpp:53-80 (Shi-Tomasi + rBRIEF + k-d tree)
void extract_features(Frame &frame);
// reference: include/Frame.h:32, src/Frame.cpp:16-51 — the grid ORB/FAST extractor (its only call is
// commented out at src/vslam.cpp:63).  Draws the cell outlines into frame.image like the reference,
// fills points + descriptors, builds no k-d tree and leaves map_point_ids alone.
void extract_features(Frame &frame, int nrows, int ncols);
// reference: include/Frame.h:34, src/Frame.cpp:82-105
void match_features(const Frame &frame1, const Frame &frame2, RansacFilter &rf,
                    std::vector<std::pair<int, int>> &matches, cv::Mat &F);
// reference: include/Frame.h:30, src/Frame.cpp:8-13; called by the capture loop (src/vslam.cpp:91).  Host code: the image is
// copied and every keypoint gets cv::circle(annotated, p, 2, Scalar(0, 255, 0)) -- cv::circle itself where OpenCV is
// installed, its radius-2 outline (the eight pixels of OpenCV's midpoint circle, clipped to the image) otherwise.
void draw(const Frame &frame, cv::Mat &annotated);

namespace vslam {
struct Settings {
    int device = 0;              // env VSLAM_DEVICE
    int max_corners = 3000;      // src/Frame.cpp:61
    double quality = 0.01;
    double min_distance = 3.0;
    float keypoint_angle_deg = -1.0f;   // cv::KeyPoint(p, 20) default angle
    // 256 x (x0,y0,x1,y1) int8 rBRIEF test pairs.  Left empty (the default) the adapters describe with ORB's learned
    // table (vslam_brief_pattern_31(), OpenCV's bit_pattern_31_: what cv::ORB::compute samples, src/Frame.cpp:57,68);
    // another table can be set here or loaded from a file (env VSLAM_BRIEF_PATTERN, 1024 raw int8).
    std::vector<s8> brief_pattern;
};
Settings &settings();
}  // namespace vslam

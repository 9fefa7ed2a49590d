// Drop-in for the reference's include/helpers.h: the same two free functions and the two inline matrix helpers, computed on
// the MI355X through include/vslam_amd.h (vslam_extract_Rt / vslam_triangulate_points), plus the batch form of the capture
// loop's map-association block.  With this header src/vslam.cpp:82,186 reach the device versions unchanged.
#pragma once
#include <iostream>
#include <vector>

#include "Frame.h"
#include "cvlite.h"
#include "vslam_internal.h"

#ifndef VSLAM_HAVE_OPENCV
// cv::Mat's stream output in OpenCV's default format ("[a, b;\n c, d]"), for print_matrix on builds without OpenCV
inline std::ostream &operator<<(std::ostream &os, const cv::Mat &m) {
    os << '[';
    for (int r = 0; r < m.rows; r++) {
        for (int c = 0; c < m.cols * m.channels(); c++) {
            if (c) os << ", ";
            if (m.depth() == CV_32F) os << m.ptr<float>(r)[c];
            else os << (int)m.ptr<unsigned char>(r)[c];
        }
        if (r + 1 < m.rows) os << ";\n ";
    }
    return os << ']';
}
#endif

// reference: include/helpers.h:9-11
inline void print_matrix(const cv::Mat &mat, const char *name) {
    std::cout << name << '\n' << mat << '\n' << '\n';
}

#ifdef VSLAM_HAVE_OPENCV
// reference: include/helpers.h:13-15 (cv::FileStorage exists only where OpenCV is installed)
inline void write_matrix(const cv::Mat &mat, const char *name, cv::FileStorage &fs) {
    fs << name << mat;
}
#endif

// reference: include/helpers.h:17, src/helpers.cpp:3-35.  fundamental, K: 3 x 3 CV_32F; rotation 3 x 3, translation 3 x 1.
void extract_Rt(const cv::Mat &fundamental, const cv::Mat &K, cv::Mat &rotation, cv::Mat &translation);

// reference: include/helpers.h:19, src/helpers.cpp:37-80.  p1, p2: N x 2 CV_32F (continuous, as the reference reads them);
// c1, c2: 3 x 4 CV_32F; points_4d: N x 4 CV_32F rows (x, y, z, 1).
void triangulate(const cv::Mat &p1, const cv::Mat &p2, const cv::Mat &c1, const cv::Mat &c2, cv::Mat &points_4d);

namespace vslam {
// The map-association block of the capture loop (src/vslam.cpp:129-161 with orb_distance, src/PointMap.cpp:36-46) in one
// device call -- what radius_search_batch is to a single radius_search.  map_points: N x 4 CV_32F rows (x, y, z, 1)
// (pm.points.rowRange(0, pm.size)); c2: 3 x 4.  Map point i is projected, dropped unless it lands inside [0, W) x [0, H),
// searched for in frame.kdtree with `radius` (2 in the reference), and given the FIRST hit in radius_search's order whose
// map_point_ids entry is still < 0 and whose descriptor is closer than dist_threshold (DISTANCE_THRESHOLD, 64) to one of the
// map point's observations; lower map indices claim first, as in the sequential loop.  The observations of map point i are
// rows obs_offsets[i] .. obs_offsets[i + 1] - 1 of obs_desc (M x 32 CV_8U): the descriptors
// pm.frames[pm.frame_ids[i][k]].descriptors.row(pm.frame_point_ids[i][k]) in k order.
// frame.map_point_ids is updated in place (:155); the return value holds, per map point, the keypoint it claimed or -1 -- the
// caller appends frame.id / that index to pm.frame_ids[i] / pm.frame_point_ids[i] (:156-157).
std::vector<s32> associate_map_points(Frame &frame, const cv::Mat &map_points, const cv::Mat &c2, int W, int H,
                                      const std::vector<u32> &obs_offsets, const cv::Mat &obs_desc, float radius = 2.f,
                                      u32 dist_threshold = 64);
}  // namespace vslam

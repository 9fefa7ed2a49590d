// Drop-in for the reference's include/KDTree.h: same structs, same free functions, computed on the
// MI355X through include/vslam_amd.h.  `root` is one malloc() block in pre-order that the caller
// releases with free() (src/vslam.cpp:295-297, tests/test_kdtree.cpp:88,142), and both structs stay
// trivially copyable (std::vector<Frame> growth copies them bitwise, src/vslam.cpp:56).
#pragma once
#include <cmath>
#include <vector>

#include "cvlite.h"
#include "vslam_internal.h"

#ifndef SQ
#define SQ(x) ((x) * (x))
#endif

struct KDTree {   // reference: include/KDTree.h:13-23
    struct KDTreeNode {
        cv::Point2f pt;
        KDTreeNode *left;
        KDTreeNode *right;
    };
    KDTreeNode *root;
    u32 size = 0;
    u8 height = 0;
};

struct frame_kdtree {   // reference: include/KDTree.h:47-57
    struct KDTreeNode {
        usize pt_index;
        KDTreeNode *left;
        KDTreeNode *right;
    };
    KDTreeNode *root;
    u32 size = 0;
    u8 height = 0;
};

// reference: include/KDTree.h:25, src/KDTree.cpp:25-35
void construct_kdtree(KDTree &kdtree, const std::vector<cv::Point2f> &points);
// reference: include/KDTree.h:30, src/KDTree.cpp:37-43
cv::Point2f nearest(const KDTree &kdtree, const cv::Point2f &query_pt, float max_distance_sq = INFINITY);
// reference: include/KDTree.h:44, src/KDTree.cpp:73-78
std::vector<cv::Point2f> radius_search(const KDTree &kdtree, const cv::Point2f &query_pt, float radius);

// reference: include/KDTree.h:60, src/KDTree.cpp:107-121
void construct_kdtree(frame_kdtree &kdtree, const std::vector<cv::Point2f> &points);
// reference: include/KDTree.h:79, src/KDTree.cpp:145-150 (tree passed by value, as there)
std::vector<usize> radius_search(const frame_kdtree kdtree, const std::vector<cv::Point2f> &points,
                                 const cv::Point2f &query_pt, float radius);

namespace vslam {
// One device round trip for a whole set of queries (what the loop at src/vslam.cpp:146-160 should
// call): result[i] = radius_search(kdtree, points, queries[i], radius), same order.
std::vector<std::vector<usize>> radius_search_batch(const frame_kdtree &kdtree,
                                                    const std::vector<cv::Point2f> &points,
                                                    const std::vector<cv::Point2f> &queries, float radius);
// drops the device-side copy kept for a tree (call before free(root) in long runs)
void forget_kdtree(const void *root);
}  // namespace vslam

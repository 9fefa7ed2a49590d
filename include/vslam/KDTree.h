// Drop-in for the reference's include/KDTree.h: same structs, same free functions, computed on the
// MI355X through include/vslam_amd.h.  `root` is one malloc() block in pre-order that the caller
// releases with free() (src/vslam.cpp:295-297, tests/test_kdtree.cpp:88,142), and both structs stay
// trivially copyable (std::vector<Frame> growth copies them bitwise, src/vslam.cpp:56).
#pragma once
#include <cmath>
#include <vector>

#include "cvlite.h"
#include "vslam_internal.h"

// reference: include/KDTree.h:9-11 (public macros; consumers may use them)
#ifndef SQ
#define SQ(x) ((x) * (x))
#endif
#ifndef ABS
#define ABS(x) (((x) > 0) ? x : -x)
#endif
#ifndef P
#define P(pt, i) ((float *)&(pt))[i]
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

// The reference's recursive helpers are public too (include/KDTree.h:26-28,31-32,45,61-63,80).  They take a node
// POINTER into a host-resident, pointer-linked subtree (any node, any starting axis), append to caller-owned storage
// and are what the top-level functions above recurse through in the reference.  Here the top-level functions go to
// the device; these walk the host structure they are handed (vslam_amd/host/kdtree_nodes.cpp) with the reference's
// visiting order, comparisons and tie behaviour, so mixing them with device-built trees gives the same answers.
// construct_kdtree(tree, points, l, r, axis) appends nodes at tree.root[tree.size++] in pre-order: the caller provides
// the storage, as the reference's top-level function does before recursing (src/KDTree.cpp:29-32,112-118).
KDTree::KDTreeNode *construct_kdtree(KDTree &kdtree, std::vector<cv::Point2f> &points,
                                     const std::vector<cv::Point2f>::iterator l,
                                     const std::vector<cv::Point2f>::iterator r, int axis);
void nearest(KDTree::KDTreeNode *node, const cv::Point2f &query_pt, int axis, cv::Point2f *best_pt,
             float *best_distance_sq);
void radius_search(KDTree::KDTreeNode *node, const cv::Point2f &query_pt, std::vector<cv::Point2f> &pts, float radius,
                   float radius_sq, int axis);
frame_kdtree::KDTreeNode *construct_kdtree(frame_kdtree &kdtree, const std::vector<cv::Point2f> &points,
                                           std::vector<usize> &point_indices, const std::vector<usize>::iterator l,
                                           const std::vector<usize>::iterator r, int axis);
void radius_search(frame_kdtree::KDTreeNode *node, const std::vector<cv::Point2f> &points, const cv::Point2f &query_pt,
                   std::vector<usize> &indices, float radius, float radius_sq, int axis);
// Declared by the reference but defined nowhere in it (include/KDTree.h:34-37,65-72): nearest_approx (both trees) and
// nearest(frame_kdtree ...).  Nothing can link against them there either; they are not declared here.

namespace vslam {
// One device round trip for a whole set of queries (what the loop at src/vslam.cpp:146-160 should
// call): result[i] = radius_search(kdtree, points, queries[i], radius), same order.
std::vector<std::vector<usize>> radius_search_batch(const frame_kdtree &kdtree,
                                                    const std::vector<cv::Point2f> &points,
                                                    const std::vector<cv::Point2f> &queries, float radius);
// drops the device-side copy kept for a tree (call before free(root) in long runs)
void forget_kdtree(const void *root);
}  // namespace vslam

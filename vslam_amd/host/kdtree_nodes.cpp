// The node-pointer overloads of the reference's KDTree.h (include/KDTree.h:26-28,31-32,45,61-63,80;
// src/KDTree.cpp:3-23,45-71,80-101,122-143,151-171).
//
// These are the public recursive helpers: they are handed a POINTER to a node of a host-resident, pointer-linked tree
// (any subtree, any starting axis) plus caller-owned output storage.  There is no batch to put on a GPU and no device
// representation of "the subtree under this host pointer", so they walk the host structure directly — a few dozen
// pointer hops per call.  The batch-capable top-level entry points (construct_kdtree(tree, points),
// radius_search(tree, ...), nearest(tree, ...), vslam::radius_search_batch) are the device path (adapters.cpp).
// Semantics restated from the reference: inclusive |split| <= r descent, strict d^2 < r^2 hit test, left before
// right, nearest's strict '<' update and split^2 < best far-side test, median at l + len/2 by std::nth_element,
// nodes appended at root[size++] before recursing (pre-order).
#include <algorithm>

#include "../../include/vslam/KDTree.h"

namespace {
inline float coord(const cv::Point2f &p, int axis) { return axis == 0 ? p.x : p.y; }
inline float dist_sq(const cv::Point2f &a, const cv::Point2f &b) {
    const float dx = a.x - b.x, dy = a.y - b.y;
    return dx * dx + dy * dy;
}
}  // namespace

KDTree::KDTreeNode *construct_kdtree(KDTree &kdtree, std::vector<cv::Point2f> &points,
                                     const std::vector<cv::Point2f>::iterator l,
                                     const std::vector<cv::Point2f>::iterator r, int axis) {
    (void)points;
    if (!(l < r)) return nullptr;
    const auto mid = l + (r - l) / 2;
    std::nth_element(l, mid, r, [axis](const cv::Point2f &a, const cv::Point2f &b) { return coord(a, axis) < coord(b, axis); });
    KDTree::KDTreeNode &node = kdtree.root[kdtree.size++];
    node.pt = *mid;
    node.left = construct_kdtree(kdtree, points, l, mid, 1 - axis);
    node.right = construct_kdtree(kdtree, points, mid + 1, r, 1 - axis);
    return &node;
}

void nearest(KDTree::KDTreeNode *node, const cv::Point2f &query_pt, int axis, cv::Point2f *best_pt,
             float *best_distance_sq) {
    if (!node) return;
    const float split = coord(query_pt, axis) - coord(node->pt, axis);
    KDTree::KDTreeNode *near_side = split < 0 ? node->left : node->right;
    KDTree::KDTreeNode *far_side = split < 0 ? node->right : node->left;
    nearest(near_side, query_pt, 1 - axis, best_pt, best_distance_sq);
    const float d = dist_sq(node->pt, query_pt);
    if (d < *best_distance_sq) {
        *best_distance_sq = d;
        *best_pt = node->pt;
    }
    if (split * split < *best_distance_sq) nearest(far_side, query_pt, 1 - axis, best_pt, best_distance_sq);
}

void radius_search(KDTree::KDTreeNode *node, const cv::Point2f &query_pt, std::vector<cv::Point2f> &pts, float radius,
                   float radius_sq, int axis) {
    if (!node) return;
    const float split = coord(query_pt, axis) - coord(node->pt, axis);
    if ((split < 0 ? -split : split) <= radius) {
        if (dist_sq(query_pt, node->pt) < radius_sq) pts.push_back(node->pt);
        radius_search(node->left, query_pt, pts, radius, radius_sq, 1 - axis);
        radius_search(node->right, query_pt, pts, radius, radius_sq, 1 - axis);
    } else if (split < 0) {
        radius_search(node->left, query_pt, pts, radius, radius_sq, 1 - axis);
    } else {
        radius_search(node->right, query_pt, pts, radius, radius_sq, 1 - axis);
    }
}

frame_kdtree::KDTreeNode *construct_kdtree(frame_kdtree &kdtree, const std::vector<cv::Point2f> &points,
                                           std::vector<usize> &point_indices, const std::vector<usize>::iterator l,
                                           const std::vector<usize>::iterator r, int axis) {
    (void)point_indices;
    if (!(l < r)) return nullptr;
    const auto mid = l + (r - l) / 2;
    std::nth_element(l, mid, r, [&points, axis](usize a, usize b) { return coord(points[a], axis) < coord(points[b], axis); });
    frame_kdtree::KDTreeNode &node = kdtree.root[kdtree.size++];
    node.pt_index = *mid;
    node.left = construct_kdtree(kdtree, points, point_indices, l, mid, 1 - axis);
    node.right = construct_kdtree(kdtree, points, point_indices, mid + 1, r, 1 - axis);
    return &node;
}

void radius_search(frame_kdtree::KDTreeNode *node, const std::vector<cv::Point2f> &points, const cv::Point2f &query_pt,
                   std::vector<usize> &indices, float radius, float radius_sq, int axis) {
    if (!node) return;
    const cv::Point2f &pt = points[node->pt_index];
    const float split = coord(query_pt, axis) - coord(pt, axis);
    if ((split < 0 ? -split : split) <= radius) {
        if (dist_sq(query_pt, pt) < radius_sq) indices.push_back(node->pt_index);
        radius_search(node->left, points, query_pt, indices, radius, radius_sq, 1 - axis);
        radius_search(node->right, points, query_pt, indices, radius, radius_sq, 1 - axis);
    } else if (split < 0) {
        radius_search(node->left, points, query_pt, indices, radius, radius_sq, 1 - axis);
    } else {
        radius_search(node->right, points, query_pt, indices, radius, radius_sq, 1 - axis);
    }
}

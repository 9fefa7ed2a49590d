// ORACLE (test infrastructure only) — 2-D k-d trees of the reference, restated.
//
// Follows /root/reference/src/KDTree.cpp:
//   frame_kdtree build   :107-143   (index-storing tree used by the live path)
//   KDTree build         :3-35      (point-storing twin used by the reference's test)
//   nearest              :37-71
//   radius_search        :73-101 and :145-171
// The median split uses the host's std::nth_element exactly as the reference does, so
// tie placement is libstdc++'s.  Nodes are appended to one array before recursing, so
// the array is in pre-order; the exported form is that array's payload column.
// Parity: PINNED by tests/test_oracle_kdtree_replay.py (replay of the reference's own
// tests/test_kdtree.cpp procedure).
#include "vso.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

struct Pt { float x, y; };
static inline float coord(const Pt &p, int axis) { return axis == 0 ? p.x : p.y; }

// node arrays use int links (-1 == NULL) instead of raw pointers
struct FNode { int32_t idx; int left, right; };
struct PNode { Pt pt; int left, right; };

struct FrameTree { std::vector<FNode> nodes; int root = -1; };
struct PointTree { std::vector<PNode> nodes; int root = -1; };

// src/KDTree.cpp:122-143
int build_frame(FrameTree &t, const std::vector<Pt> &pts, std::vector<size_t> &order,
                size_t l, size_t r, int axis) {
    if (l < r) {
        const size_t len = r - l;
        const size_t m = l + len / 2;
        std::nth_element(order.begin() + l, order.begin() + m, order.begin() + r,
                         [&pts, axis](const size_t &a, const size_t &b) {
                             return coord(pts[a], axis) < coord(pts[b], axis);
                         });
        const int me = (int)t.nodes.size();
        t.nodes.push_back({(int32_t)order[m], -1, -1});
        const int lc = build_frame(t, pts, order, l, m, 1 - axis);
        const int rc = build_frame(t, pts, order, m + 1, r, 1 - axis);
        t.nodes[me].left = lc;
        t.nodes[me].right = rc;
        return me;
    }
    return -1;
}

// src/KDTree.cpp:3-23
int build_points(PointTree &t, std::vector<Pt> &pts, size_t l, size_t r, int axis) {
    if (l < r) {
        const size_t len = r - l;
        const size_t m = l + len / 2;
        if (axis == 0)
            std::nth_element(pts.begin() + l, pts.begin() + m, pts.begin() + r,
                             [](const Pt &a, const Pt &b) { return a.x < b.x; });
        else
            std::nth_element(pts.begin() + l, pts.begin() + m, pts.begin() + r,
                             [](const Pt &a, const Pt &b) { return a.y < b.y; });
        const int me = (int)t.nodes.size();
        t.nodes.push_back({pts[m], -1, -1});
        const int lc = build_points(t, pts, l, m, 1 - axis);
        const int rc = build_points(t, pts, m + 1, r, 1 - axis);
        t.nodes[me].left = lc;
        t.nodes[me].right = rc;
        return me;
    }
    return -1;
}

// Rebuild links of a pre-order array: the left subtree of a range of `len` nodes has
// len/2 nodes and the right one len - len/2 - 1 (src/KDTree.cpp:127,138-139).
template <class Node>
int link_preorder(std::vector<Node> &nodes, int pos, int len) {
    if (len <= 0) return -1;
    const int nl = len / 2, nr = len - nl - 1;
    nodes[pos].left = link_preorder(nodes, pos + 1, nl);
    nodes[pos].right = link_preorder(nodes, pos + 1 + nl, nr);
    return pos;
}

// src/KDTree.cpp:151-171
void radius_frame(const FrameTree &t, int node, const Pt *pts, const Pt &q,
                  std::vector<int32_t> &hits, float radius, float radius_sq, int axis) {
    if (node < 0) return;
    const FNode &nd = t.nodes[node];
    const Pt &pt = pts[nd.idx];
    const float split = coord(q, axis) - coord(pt, axis);
    const float abs_split = (split > 0) ? split : -split;   // ABS macro, include/KDTree.h:10
    if (abs_split <= radius) {
        const float dx = q.x - pt.x, dy = q.y - pt.y;
        const float d2 = dx * dx + dy * dy;                 // cv::Point2f::dot in f32
        if (d2 < radius_sq) hits.push_back(nd.idx);
        radius_frame(t, nd.left, pts, q, hits, radius, radius_sq, 1 - axis);
        radius_frame(t, nd.right, pts, q, hits, radius, radius_sq, 1 - axis);
    } else if (split < 0) {
        radius_frame(t, nd.left, pts, q, hits, radius, radius_sq, 1 - axis);
    } else {
        radius_frame(t, nd.right, pts, q, hits, radius, radius_sq, 1 - axis);
    }
}

// src/KDTree.cpp:80-101
void radius_points(const PointTree &t, int node, const Pt &q, std::vector<Pt> &hits,
                   float radius, float radius_sq, int axis) {
    if (node < 0) return;
    const PNode &nd = t.nodes[node];
    const Pt pt = nd.pt;
    const float split = coord(q, axis) - coord(pt, axis);
    const float abs_split = (split > 0) ? split : -split;
    if (abs_split <= radius) {
        const float dx = q.x - pt.x, dy = q.y - pt.y;
        const float d2 = dx * dx + dy * dy;
        if (d2 < radius_sq) hits.push_back(pt);
        radius_points(t, nd.left, q, hits, radius, radius_sq, 1 - axis);
        radius_points(t, nd.right, q, hits, radius, radius_sq, 1 - axis);
    } else if (split < 0) {
        radius_points(t, nd.left, q, hits, radius, radius_sq, 1 - axis);
    } else {
        radius_points(t, nd.right, q, hits, radius, radius_sq, 1 - axis);
    }
}

// src/KDTree.cpp:45-71
void nearest_points(const PointTree &t, int node, const Pt &q, int axis, Pt *best,
                    float *best_d2) {
    if (node < 0) return;
    const PNode &nd = t.nodes[node];
    const Pt &pt = nd.pt;
    const float split = coord(q, axis) - coord(pt, axis);
    int opposite;
    if (split < 0) {
        nearest_points(t, nd.left, q, 1 - axis, best, best_d2);
        opposite = nd.right;
    } else {
        nearest_points(t, nd.right, q, 1 - axis, best, best_d2);
        opposite = nd.left;
    }
    const float dx = pt.x - q.x, dy = pt.y - q.y;
    const float cur = dx * dx + dy * dy;
    if (cur < *best_d2) {
        *best_d2 = cur;
        *best = pt;
    }
    if (split * split < *best_d2) nearest_points(t, opposite, q, 1 - axis, best, best_d2);
}

}  // namespace

extern "C" {

int vso_kdtree_height(int n) {
    if (n <= 0) return 0;
    return (int)(std::floor(std::log2((double)n)) + 1);   // src/KDTree.cpp:33,119
}

int vso_kdtree_build_frame(const float *xy, int n, int32_t *out_idx) {
    if (n < 0 || (n > 0 && (!xy || !out_idx))) return -1;
    if (n == 0) return 0;                                  // root = NULL, src/KDTree.cpp:109-110
    std::vector<Pt> pts(n);
    for (int i = 0; i < n; i++) pts[i] = {xy[2 * i], xy[2 * i + 1]};
    std::vector<size_t> order(n);
    for (int i = 0; i < n; i++) order[i] = (size_t)i;      // :113-117
    FrameTree t;
    t.nodes.reserve(n);
    t.root = build_frame(t, pts, order, 0, (size_t)n, 0);
    for (int i = 0; i < n; i++) out_idx[i] = t.nodes[i].idx;
    return 0;
}

int vso_kdtree_build_points(const float *xy, int n, float *out_xy) {
    if (n < 0 || (n > 0 && (!xy || !out_xy))) return -1;
    if (n == 0) return 0;
    std::vector<Pt> pts(n);                                // points_copy, src/KDTree.cpp:31
    for (int i = 0; i < n; i++) pts[i] = {xy[2 * i], xy[2 * i + 1]};
    PointTree t;
    t.nodes.reserve(n);
    t.root = build_points(t, pts, 0, (size_t)n, 0);
    for (int i = 0; i < n; i++) {
        out_xy[2 * i] = t.nodes[i].pt.x;
        out_xy[2 * i + 1] = t.nodes[i].pt.y;
    }
    return 0;
}

int vso_kdtree_radius_frame(const int32_t *pre_idx, const float *xy, int n, float qx, float qy,
                            float radius, int32_t *out_idx, int cap) {
    if (n < 0 || cap < 0) return -1;
    FrameTree t;
    t.nodes.resize(n);
    for (int i = 0; i < n; i++) t.nodes[i] = {pre_idx[i], -1, -1};
    t.root = link_preorder(t.nodes, 0, n);
    std::vector<int32_t> hits;
    const Pt q{qx, qy};
    const float radius_sq = radius * radius;               // SQ(radius), src/KDTree.cpp:146
    radius_frame(t, t.root, reinterpret_cast<const Pt *>(xy), q, hits, radius, radius_sq, 0);
    for (int i = 0; i < (int)hits.size() && i < cap; i++) out_idx[i] = hits[i];
    return (int)hits.size();
}

int vso_kdtree_radius_points(const float *pre_xy, int n, float qx, float qy, float radius,
                             float *out_xy, int cap) {
    if (n < 0 || cap < 0) return -1;
    PointTree t;
    t.nodes.resize(n);
    for (int i = 0; i < n; i++) t.nodes[i] = {{pre_xy[2 * i], pre_xy[2 * i + 1]}, -1, -1};
    t.root = link_preorder(t.nodes, 0, n);
    std::vector<Pt> hits;
    const Pt q{qx, qy};
    const float radius_sq = radius * radius;
    radius_points(t, t.root, q, hits, radius, radius_sq, 0);
    for (int i = 0; i < (int)hits.size() && i < cap; i++) {
        out_xy[2 * i] = hits[i].x;
        out_xy[2 * i + 1] = hits[i].y;
    }
    return (int)hits.size();
}

int vso_kdtree_nearest_points(const float *pre_xy, int n, float qx, float qy,
                              float max_distance_sq, float *out_xy) {
    if (n < 0 || !out_xy) return -1;
    PointTree t;
    t.nodes.resize(n);
    for (int i = 0; i < n; i++) t.nodes[i] = {{pre_xy[2 * i], pre_xy[2 * i + 1]}, -1, -1};
    t.root = link_preorder(t.nodes, 0, n);
    Pt best{0.f, 0.f};                                      // default-constructed, :40
    float best_d2 = max_distance_sq;
    nearest_points(t, t.root, Pt{qx, qy}, 0, &best, &best_d2);
    out_xy[0] = best.x;
    out_xy[1] = best.y;
    return 0;
}

}  // extern "C"

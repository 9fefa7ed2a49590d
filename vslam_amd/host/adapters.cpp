// Host side of the drop-in: implements the reference's Frame / KDTree / RansacFilter free functions
// and class (include/vslam/*.h) on top of the C ABI (include/vslam_amd.h).  Everything that
// computes runs on the device; this file only marshals std::vector / cv::Mat data across the
// boundary and rebuilds the pointer-linked node arrays the consumers expect.  No CPU fallback:
// if no device is available the first call throws.
//
// Latency shape of one call: the reference's consumers call these one frame / one query at a time, so
// what a call costs here is round trips, not arithmetic.  Every entry point therefore
//   * takes its device and page-locked staging memory from a grow-only scratch owned by the process-wide
//     context (no hipMalloc / hipFree per call),
//   * packs all of its inputs into ONE host->device copy and all of its outputs into ONE device->host copy,
//   * chains its kernels on the device (match -> sets -> RANSAC without the matches visiting the host).
// Entry points are serialised by one lock: one context, one stream, one scratch (the reference's hot path is
// single-threaded, src/vslam.cpp:53-294).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vslam/Frame.h"
#include "../../include/vslam/helpers.h"
#ifdef VSLAM_HAVE_OPENCV
#include <opencv2/imgproc.hpp>
#endif
#include "../../include/vslam_amd.h"
#include "host_internal.h"

namespace {

struct Block {
    void *p = nullptr;
    size_t bytes = 0;
};

void drop_cached_trees();   // defined with the tree cache below

struct Device {
    vslam_ctx *ctx = nullptr;
    std::map<std::string, Block> dev, pinned;   // grow-only scratch, released with the context
    std::vector<Block> tree_blocks;             // device blocks of dropped trees, reused by the next ones
    Device() {
        const char *d = std::getenv("VSLAM_DEVICE");
        vslam::settings().device = d ? std::atoi(d) : vslam::settings().device;
        const int rc = vslam_ctx_create(vslam::settings().device, &ctx);
        if (rc != VSLAM_OK)
            throw std::runtime_error("vslam_amd: no usable HIP device (vslam_ctx_create rc=" + std::to_string(rc) +
                                     "); the front-end has no CPU fallback");
    }
    ~Device() {
        if (!ctx) return;
        drop_cached_trees();
        for (auto &b : tree_blocks)
            if (b.p) vslam_dev_free(ctx, b.p);
        for (auto &kv : dev)
            if (kv.second.p) vslam_dev_free(ctx, kv.second.p);
        for (auto &kv : pinned)
            if (kv.second.p) vslam_host_free(ctx, kv.second.p);
        vslam_ctx_destroy(ctx);
    }
};

Device &device() {
    static Device dev;
    return dev;
}
vslam_ctx *ctx() { return device().ctx; }

std::recursive_mutex g_mu;
using Lock = std::lock_guard<std::recursive_mutex>;

void check(int rc, const char *what) {
    if (rc != VSLAM_OK)
        throw std::runtime_error(std::string("vslam_amd: ") + what + " failed (rc=" + std::to_string(rc) + "): " +
                                 vslam_last_error(ctx()));
}

// grow-only named scratch: device memory and page-locked host staging
void *dev_scratch(const char *name, size_t bytes) {
    Block &b = device().dev[name];
    if (b.bytes < bytes) {
        if (b.p) check(vslam_dev_free(ctx(), b.p), "dev_free");
        b.p = nullptr;
        const size_t want = bytes + bytes / 4 + 256;
        check(vslam_dev_alloc(ctx(), want, &b.p), "dev_alloc");
        b.bytes = want;
    }
    return b.p;
}
void *pinned_scratch(const char *name, size_t bytes) {
    Block &b = device().pinned[name];
    if (b.bytes < bytes) {
        if (b.p) check(vslam_host_free(ctx(), b.p), "host_free");
        b.p = nullptr;
        const size_t want = bytes + bytes / 4 + 256;
        check(vslam_host_alloc(ctx(), want, &b.p), "host_alloc");
        b.bytes = want;
    }
    return b.p;
}

// offsets of the pieces of one packed transfer (256-byte aligned: every piece can be read with 16-byte loads)
struct Layout {
    size_t total = 0;
    size_t add(size_t bytes) {
        const size_t off = (total + 255) & ~(size_t)255;
        total = off + bytes;
        return off;
    }
};

// One call's memory: `in` travels host -> device in one copy, `out` device -> host in one copy, `work` stays there.
struct Call {
    uint8_t *h_in = nullptr, *d_in = nullptr, *h_out = nullptr, *d_out = nullptr, *d_work = nullptr;
    // Small calls (a single k-d query: a few bytes each way) skip the copies altogether: page-locked host memory is
    // addressable from the device, so the kernel reads its arguments from and writes its answer to the staging
    // buffers directly and the call costs one launch and one stream wait.
    static constexpr size_t kZeroCopyBytes = 4096;
    Call(const Layout &in, const Layout &out, const Layout &work) {
        h_in = static_cast<uint8_t *>(pinned_scratch("call.in", in.total + 1));
        h_out = static_cast<uint8_t *>(pinned_scratch("call.out", out.total + 1));
        zero_copy = in.total <= kZeroCopyBytes && out.total <= kZeroCopyBytes;
        d_in = zero_copy ? h_in : static_cast<uint8_t *>(dev_scratch("call.in", in.total + 1));
        d_out = zero_copy ? h_out : static_cast<uint8_t *>(dev_scratch("call.out", out.total + 1));
        d_work = static_cast<uint8_t *>(dev_scratch("call.work", work.total + 1));
        n_in = in.total;
        n_out = out.total;
    }
    void upload() {
        if (n_in && !zero_copy) check(vslam_copy_h2d(ctx(), d_in, h_in, n_in), "copy_h2d");
    }
    void download() {
        if (zero_copy) check(vslam_ctx_wait(ctx()), "ctx_wait");
        else if (n_out) check(vslam_copy_d2h(ctx(), h_out, d_out, n_out), "copy_d2h");
    }
    template <class T>
    T *hin(size_t off) { return reinterpret_cast<T *>(h_in + off); }
    template <class T>
    T *din(size_t off) { return reinterpret_cast<T *>(d_in + off); }
    template <class T>
    T *hout(size_t off) { return reinterpret_cast<T *>(h_out + off); }
    template <class T>
    T *dout(size_t off) { return reinterpret_cast<T *>(d_out + off); }
    template <class T>
    T *work(size_t off) { return reinterpret_cast<T *>(d_work + off); }

   private:
    size_t n_in = 0, n_out = 0;
    bool zero_copy = false;
};

// ---------------------------------------------------------------------------------------- device copies of trees
// A tree handed to radius_search / nearest lives in host memory the caller owns; its device copy is cached under the
// root pointer and VALIDATED on every use by a hash of what the kernels would read (the pre-order pt_index column and
// the points): a tree freed and rebuilt at the same address, or points edited in place, re-uploads instead of
// answering for the old tree.
struct DevTree {
    void *block = nullptr;   // [n : int32][nodes : int32 x stride][xy : float x 2 stride]
    size_t block_bytes = 0;
    int32_t *n = nullptr, *nodes = nullptr;
    float *xy = nullptr;
    int count = 0, stride = 0;
    uint64_t hash = 0, stamp = 0;
    // Single radius queries (src/vslam.cpp:149 asks one per map point) are answered from a table the device built when
    // the tree was constructed — the tree's points filed by pixel cell, vslam_kdtree_cell_table — copied to the host
    // once.  Valid for the very point vector the tree was built from.  Checked per query: the vector's identity, a
    // sample of its contents, and that every probed point still lies in the cell it was filed under; checked in full
    // (every point and every node's pt_index against the hash taken at build time) on the first query and then every
    // kFullCheckEvery-th one.  A mismatch drops the table and the query takes the device path, which re-validates
    // everything on every call.  An in-place edit of a point that no probe touches can therefore go unnoticed for at
    // most kFullCheckEvery - 1 single queries; vslam::forget_kdtree drops the table at once.
    std::vector<uint32_t> cells;      // [slots][2]
    std::vector<int32_t> pre;         // pre-order position -> point index
    uint32_t cell_mask = 0;
    const void *pts_ptr = nullptr;
    size_t pts_size = 0;
    uint64_t pts_sample = 0;
    uint32_t since_full_check = 0;    // single queries answered since the last full validation
    ~DevTree() {   // the block goes back to the pool, not to hipFree
        if (block) device().tree_blocks.push_back(Block{block, block_bytes});
    }
};
std::map<const void *, std::shared_ptr<DevTree>> g_trees;
void drop_cached_trees() { g_trees.clear(); }
uint64_t g_stamp = 0;
constexpr size_t kTreeCache = 64;
constexpr uint32_t kFullCheckEvery = 256;

inline uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
uint64_t hash_points(uint64_t h, const cv::Point2f *p, size_t n) {
    static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f is two floats");
    for (size_t i = 0; i < n; i++) {
        uint64_t v;
        std::memcpy(&v, &p[i], 8);
        h = mix(h, v);
    }
    return h;
}
// nodes a tree really has (its array holds exactly these, in pre-order): the caller's point vector may have grown
// since construct_kdtree, and the reference only ever dereferences pt_index values
template <class Node>
int node_count(const Node *nd) { return nd ? 1 + node_count(nd->left) + node_count(nd->right) : 0; }

uint64_t hash_tree(const frame_kdtree &kd, int nodes, const std::vector<cv::Point2f> &points) {
    uint64_t h = mix(0x5EED, (uint64_t)nodes);
    for (int i = 0; i < nodes; i++) h = mix(h, (uint64_t)kd.root[i].pt_index);
    return hash_points(h, points.data(), points.size());
}

// n nodes over npts points (npts >= n: the nodes index into the caller's point vector, which may have grown)
std::shared_ptr<DevTree> alloc_tree(int n, int npts = -1) {
    auto t = std::make_shared<DevTree>();
    t->count = n;
    t->stride = std::max(std::max(n, npts), 1);
    Layout L;
    const size_t o_n = L.add(4), o_nodes = L.add(4 * (size_t)t->stride), o_xy = L.add(8 * (size_t)t->stride);
    auto &pool = device().tree_blocks;
    for (size_t i = 0; i < pool.size() && !t->block; i++)
        if (pool[i].bytes >= L.total) {
            t->block = pool[i].p;
            t->block_bytes = pool[i].bytes;
            pool.erase(pool.begin() + (long)i);
        }
    if (!t->block) {
        t->block_bytes = L.total + L.total / 4;   // head room: frames have similar keypoint counts
        check(vslam_dev_alloc(ctx(), t->block_bytes, &t->block), "dev_alloc");
    }
    uint8_t *base = static_cast<uint8_t *>(t->block);
    t->n = reinterpret_cast<int32_t *>(base + o_n);
    t->nodes = reinterpret_cast<int32_t *>(base + o_nodes);
    t->xy = reinterpret_cast<float *>(base + o_xy);
    return t;
}

void remember_tree(const void *root, const std::shared_ptr<DevTree> &t) {
    t->stamp = ++g_stamp;
    if (g_trees.size() >= kTreeCache && g_trees.find(root) == g_trees.end()) {   // drop the least recently used
        auto old = g_trees.begin();
        for (auto it = g_trees.begin(); it != g_trees.end(); ++it)
            if (it->second->stamp < old->second->stamp) old = it;
        g_trees.erase(old);
    }
    g_trees[root] = t;
}

// upload (count, pre-order index column, points) in one copy
std::shared_ptr<DevTree> upload_tree(const int32_t *pre_idx, const cv::Point2f *pts, int n, int npts = -1) {
    if (npts < n) npts = n;
    auto t = alloc_tree(n, npts);
    const size_t bytes = (size_t)(reinterpret_cast<uint8_t *>(t->xy) - static_cast<uint8_t *>(t->block)) + 8 * (size_t)t->stride;
    uint8_t *h = static_cast<uint8_t *>(pinned_scratch("tree.stage", bytes));
    std::memset(h, 0, 4);
    const int32_t nn = n;
    std::memcpy(h, &nn, 4);
    if (n > 0) {
        std::memcpy(h + (reinterpret_cast<uint8_t *>(t->nodes) - static_cast<uint8_t *>(t->block)), pre_idx, 4 * (size_t)n);
        std::memcpy(h + (reinterpret_cast<uint8_t *>(t->xy) - static_cast<uint8_t *>(t->block)), pts, 8 * (size_t)npts);
    }
    check(vslam_copy_h2d(ctx(), t->block, h, bytes), "copy_h2d");
    return t;
}

uint64_t sample_points(const std::vector<cv::Point2f> &p) {
    uint64_t h = mix(0xC311, p.size());
    const size_t n = p.size(), step = n > 16 ? n / 16 : 1;
    for (size_t i = 0; i < n; i += step) {
        uint64_t v;
        std::memcpy(&v, &p[i], 8);
        h = mix(h, v);
    }
    return h;
}

// file the tree's points by pixel cell on the device and keep the table on the host (one kernel, one copy)
void build_cell_table(DevTree &t, const std::vector<int32_t> &pre, const std::vector<cv::Point2f> &points) {
    const int n = t.count;
    t.cells.clear();
    if (n <= 0) return;
    int slots = 64;
    while (slots < 2 * t.stride) slots <<= 1;
    Layout in, res, work;
    const size_t o_tab = res.add(8 * (size_t)slots), o_ok = res.add(4);
    Call c(in, res, work);
    check(vslam_kdtree_cell_table(ctx(), t.nodes, t.xy, t.n, 1, t.stride, slots, c.dout<uint32_t>(o_tab), c.dout<int32_t>(o_ok)),
          "kdtree_cell_table");
    c.download();
    if (*c.hout<int32_t>(o_ok) == 0) return;   // coordinates outside the key range: single queries take the device path
    t.cells.assign(c.hout<uint32_t>(o_tab), c.hout<uint32_t>(o_tab) + 2 * (size_t)slots);
    t.cell_mask = (uint32_t)slots - 1u;
    t.pre = pre;
    t.pts_ptr = points.data();
    t.pts_size = points.size();
    t.pts_sample = sample_points(points);
}

// radius_search from the cell table: the hits in the reference's visit order.  false = not answerable here.
bool cell_query(const DevTree &t, const std::vector<cv::Point2f> &points, const cv::Point2f &q, float radius,
                std::vector<usize> &out) {
    if (t.cells.empty() || t.pts_ptr != points.data() || t.pts_size != points.size()) return false;
    if (!(radius >= 0.f && radius <= 8.f) || !(std::fabs(q.x) < 30000.f && std::fabs(q.y) < 30000.f)) return false;
    if (t.pts_sample != sample_points(points)) return false;
    const float r2 = radius * radius;   // SQ(radius), src/KDTree.cpp:146
    const int x0 = (int)std::floor(q.x - radius), x1 = (int)std::floor(q.x + radius);
    const int y0 = (int)std::floor(q.y - radius), y1 = (int)std::floor(q.y + radius);
    std::pair<uint32_t, int32_t> hits[64];
    int nh = 0;
    const uint32_t *T = t.cells.data();
    for (int cy = y0; cy <= y1; cy++)
        for (int cx = x0; cx <= x1; cx++) {
            const uint32_t key = ((uint32_t)(cy + 32768) << 16) | (uint32_t)(cx + 32768);
            uint32_t slot = (key * 2654435761u) >> 7;
            while (true) {
                slot &= t.cell_mask;
                const uint32_t k = T[2 * slot];
                if (k == 0xFFFFFFFFu) break;
                if (k == key) {
                    const uint32_t rank = T[2 * slot + 1];
                    const int32_t idx = t.pre[rank];
                    const cv::Point2f &pt = points[(size_t)idx];
                    if ((int)std::floor(pt.x) != cx || (int)std::floor(pt.y) != cy) return false;   // moved since it was filed
                    const float dx = q.x - pt.x, dy = q.y - pt.y;
                    const float xx = dx * dx, yy = dy * dy;
                    if (xx + yy < r2) {   // strict, src/KDTree.cpp:161
                        if (nh == 64) return false;   // a crowd: let the device path size its answer
                        hits[nh++] = {rank, idx};
                    }
                }
                slot++;
            }
        }
    std::sort(hits, hits + nh);   // pre-order position = the order the reference pushes them in
    out.resize((size_t)nh);
    for (int i = 0; i < nh; i++) out[(size_t)i] = (usize)hits[i].second;
    return true;
}

// build on the device; returns the pre-order pt_index column, keeps the device copy
std::vector<int32_t> device_build(const std::vector<cv::Point2f> &points, std::shared_ptr<DevTree> *keep) {
    const int n = (int)points.size();
    std::vector<int32_t> pre(n);
    auto t = alloc_tree(n);
    // count + points up in one copy (the node column in between is written by the kernel)
    const size_t xy_off = (size_t)(reinterpret_cast<uint8_t *>(t->xy) - static_cast<uint8_t *>(t->block));
    const size_t bytes = xy_off + 8 * (size_t)t->stride;
    uint8_t *h = static_cast<uint8_t *>(pinned_scratch("tree.stage", bytes));
    const int32_t nn = n;
    std::memcpy(h, &nn, 4);
    if (n > 0) std::memcpy(h + xy_off, points.data(), 8 * (size_t)n);
    check(vslam_copy_h2d(ctx(), t->block, h, bytes), "copy_h2d");
    if (n > 0) {
        check(vslam_kdtree_build(ctx(), t->xy, t->n, 1, t->stride, t->nodes), "kdtree_build");
        check(vslam_copy_d2h(ctx(), pre.data(), t->nodes, 4 * (size_t)n), "copy_d2h");
    }
    if (keep) *keep = t;
    return pre;
}

// links of a pre-order array: left subtree len/2 nodes, right len - len/2 - 1 (src/KDTree.cpp:127,138-139)
template <class Node>
Node *link_preorder(Node *base, int pos, int len) {
    if (len <= 0) return nullptr;
    const int nl = len / 2, nr = len - nl - 1;
    base[pos].left = link_preorder(base, pos + 1, nl);
    base[pos].right = link_preorder(base, pos + 1 + nl, nr);
    return base + pos;
}

u8 tree_height(int n) { return (u8)(std::floor(std::log2((double)n)) + 1); }   // src/KDTree.cpp:33,119

std::shared_ptr<DevTree> device_tree_for(const frame_kdtree &kd, const std::vector<cv::Point2f> &points) {
    const int n = node_count(kd.root);
    const uint64_t h = hash_tree(kd, n, points);
    auto it = g_trees.find(kd.root);
    std::shared_ptr<DevTree> table_of;   // an entry that so far only carries the host-side cell table (extract_features)
    if (it != g_trees.end() && it->second->count == n && it->second->hash == h) {
        if (it->second->block) {
            it->second->stamp = ++g_stamp;
            return it->second;
        }
        table_of = it->second;
    }
    // built elsewhere, forgotten, not on the device yet, or changed since it was uploaded: its array is in pre-order, upload it
    std::vector<int32_t> pre(n);
    for (int i = 0; i < n; i++) {
        pre[i] = (int32_t)kd.root[i].pt_index;
        if (pre[i] < 0 || (size_t)pre[i] >= points.size()) throw std::runtime_error("frame_kdtree: pt_index outside points");
    }
    auto t = upload_tree(pre.data(), points.data(), n, (int)points.size());
    t->hash = h;
    if (table_of) {
        t->cells.swap(table_of->cells);
        t->pre.swap(table_of->pre);
        t->cell_mask = table_of->cell_mask;
        t->pts_ptr = table_of->pts_ptr;
        t->pts_size = table_of->pts_size;
        t->pts_sample = table_of->pts_sample;
    }
    remember_tree(kd.root, t);
    return t;
}

// a KDTree carries its points inside the nodes (array order == pre-order): identity index column over them
std::shared_ptr<DevTree> device_tree_for(const KDTree &kd, std::vector<cv::Point2f> &pts) {
    const int n = node_count(kd.root);
    pts.resize(n);
    for (int i = 0; i < n; i++) pts[i] = kd.root[i].pt;
    const uint64_t h = hash_points(mix(0x7EE, (uint64_t)n), pts.data(), (size_t)n);
    auto it = g_trees.find(kd.root);
    if (it != g_trees.end() && it->second->count == n && it->second->hash == h) {
        it->second->stamp = ++g_stamp;
        return it->second;
    }
    std::vector<int32_t> pre(n);
    for (int i = 0; i < n; i++) pre[i] = i;
    auto t = upload_tree(pre.data(), pts.data(), n);
    t->hash = h;
    remember_tree(kd.root, t);
    return t;
}

const std::vector<s8> &pattern() {
    auto &st = vslam::settings();
    if (st.brief_pattern.size() == 1024) return st.brief_pattern;
    if (const char *path = std::getenv("VSLAM_BRIEF_PATTERN")) {
        std::ifstream f(path, std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (raw.size() != 1024) throw std::runtime_error("VSLAM_BRIEF_PATTERN must hold 1024 int8 values");
        st.brief_pattern.assign(raw.begin(), raw.end());
        return st.brief_pattern;
    }
    // the default: ORB's learned table, what cv::ORB::compute samples (src/Frame.cpp:57,68)
    const int8_t *t = vslam_brief_pattern_31();
    st.brief_pattern.assign(t, t + 1024);
    return st.brief_pattern;
}

// the pattern on the device, uploaded when it changes
const int8_t *device_pattern() {
    static std::vector<s8> uploaded;
    const std::vector<s8> &pat = pattern();
    int8_t *d = static_cast<int8_t *>(dev_scratch("pattern", 1024));
    if (uploaded != pat) {
        check(vslam_copy_h2d(ctx(), d, pat.data(), 1024), "pattern upload");
        uploaded = pat;
    }
    return d;
}

void fill_params(vslam_extract_params &p, int max_corners, const int8_t *d_pattern) {
    auto &st = vslam::settings();
    p.max_corners = max_corners;
    p.quality = st.quality;
    p.min_distance = st.min_distance;
    const float a = st.keypoint_angle_deg * (float)(3.14159265358979323846 / 180.f);   // angle *= CV_PI/180
    p.cos_a = (float)std::cos((double)a);
    p.sin_a = (float)std::sin((double)a);
    p.d_pattern = d_pattern;
}

}  // namespace

namespace vslam {
Settings &settings() {
    static Settings s;
    return s;
}

namespace detail {
vslam_ctx *context() { return ctx(); }
void check(int rc, const char *what) { ::check(rc, what); }
const std::vector<s8> &brief_pattern() { return pattern(); }
void fill_extract_params(vslam_extract_params &p, int max_corners, const int8_t *d_pattern) { fill_params(p, max_corners, d_pattern); }

// RansacFilter's device-side paths (friend of the class: they read its seed and refresh its private ransac_sets)
struct RansacAccess {
    // sets drawn on the device for `m` matches (src/RansacFilter.cpp:6-34) straight into d_sets; host copy refreshed by the caller
    static u32 seed(RansacFilter &rf) { return rf.next_seed(); }
    static void store_sets(RansacFilter &rf, const int32_t *flat) {
        rf.ransac_sets.assign(rf.max_iterations, std::vector<int>(8, 0));
        for (int i = 0; i < rf.max_iterations; i++)
            for (int j = 0; j < 8; j++) rf.ransac_sets[i][j] = flat[(size_t)i * 8 + j];
    }
};
}  // namespace detail

void forget_kdtree(const void *root) {
    Lock lk(g_mu);
    g_trees.erase(root);
}

std::vector<std::vector<usize>> radius_search_batch(const frame_kdtree &kdtree,
                                                    const std::vector<cv::Point2f> &points,
                                                    const std::vector<cv::Point2f> &queries, float radius) {
    std::vector<std::vector<usize>> out(queries.size());
    if (queries.empty() || points.empty() || kdtree.root == nullptr) return out;
    Lock lk(g_mu);
    auto t = device_tree_for(kdtree, points);
    const int q = (int)queries.size();
    int cap = 16;
    while (true) {
        Layout in, res, work;
        const size_t o_nq = in.add(4), o_q = in.add(8 * (size_t)q);
        const size_t o_cnt = res.add(4 * (size_t)q), o_hits = res.add(4 * (size_t)q * cap);
        Call c(in, res, work);
        *c.hin<int32_t>(o_nq) = q;
        std::memcpy(c.hin<float>(o_q), queries.data(), 8 * (size_t)q);
        c.upload();
        check(vslam_kdtree_radius(ctx(), t->nodes, t->xy, t->n, 1, t->stride, c.din<float>(o_q), c.din<int32_t>(o_nq), q, radius,
                                  c.dout<int32_t>(o_hits), c.dout<int32_t>(o_cnt), cap),
              "kdtree_radius");
        c.download();
        const int32_t *cnt = c.hout<int32_t>(o_cnt), *hits = c.hout<int32_t>(o_hits);
        int mx = 0;
        for (int i = 0; i < q; i++) mx = cnt[i] > mx ? cnt[i] : mx;
        if (mx > cap) {   // rare: more hits than slots, retry with room for all
            cap = mx;
            continue;
        }
        for (int i = 0; i < q; i++) out[i].assign(hits + (size_t)i * cap, hits + (size_t)i * cap + cnt[i]);
        return out;
    }
}
}  // namespace vslam

// ------------------------------------------------------------------------------------ KDTree.h
void construct_kdtree(frame_kdtree &kdtree, const std::vector<cv::Point2f> &points) {
    const usize N = points.size();
    if (N == 0) {
        kdtree.root = nullptr;   // src/KDTree.cpp:109-110
        return;
    }
    Lock lk(g_mu);
    std::shared_ptr<DevTree> dev;
    const std::vector<int32_t> pre = device_build(points, &dev);
    auto *nodes = static_cast<frame_kdtree::KDTreeNode *>(std::malloc(N * sizeof(frame_kdtree::KDTreeNode)));
    for (usize i = 0; i < N; i++) nodes[i].pt_index = (usize)pre[i];
    link_preorder(nodes, 0, (int)N);
    kdtree.root = nodes;
    kdtree.size += (u32)N;   // the reference never resets size
    kdtree.height = tree_height((int)N);
    dev->hash = hash_tree(kdtree, (int)N, points);
    build_cell_table(*dev, pre, points);
    remember_tree(nodes, dev);
}

void construct_kdtree(KDTree &kdtree, const std::vector<cv::Point2f> &points) {
    const int N = (int)points.size();
    if (N == 0) {
        kdtree.root = nullptr;
        return;
    }
    Lock lk(g_mu);
    // the point-storing twin makes the same comparisons on the same keys, so it is the index tree
    // with the points substituted (src/KDTree.cpp:3-35)
    const std::vector<int32_t> pre = device_build(points, nullptr);
    auto *nodes = static_cast<KDTree::KDTreeNode *>(std::malloc((size_t)N * sizeof(KDTree::KDTreeNode)));
    for (int i = 0; i < N; i++) nodes[i].pt = points[pre[i]];
    link_preorder(nodes, 0, N);
    kdtree.root = nodes;
    kdtree.size += (u32)N;
    kdtree.height = tree_height(N);
}

std::vector<usize> radius_search(const frame_kdtree kdtree, const std::vector<cv::Point2f> &points,
                                 const cv::Point2f &query_pt, float radius) {
    if (kdtree.root != nullptr) {   // the tree this library built, asked about the points it was built from
        Lock lk(g_mu);
        auto it = g_trees.find(kdtree.root);
        std::vector<usize> out;
        if (it != g_trees.end() && !it->second->cells.empty()) {
            DevTree &t = *it->second;
            if (t.since_full_check == 0 && (points.size() < (size_t)t.count || t.hash != hash_tree(kdtree, t.count, points)))
                t.cells.clear();   // edited since the build: the device path below sees the current points and nodes
            if (++t.since_full_check >= kFullCheckEvery) t.since_full_check = 0;
        }
        if (it != g_trees.end() && cell_query(*it->second, points, query_pt, radius, out)) return out;
    }
    return vslam::radius_search_batch(kdtree, points, std::vector<cv::Point2f>{query_pt}, radius)[0];
}

std::vector<cv::Point2f> radius_search(const KDTree &kdtree, const cv::Point2f &query_pt, float radius) {
    std::vector<cv::Point2f> out;
    if (!kdtree.root) return out;
    Lock lk(g_mu);
    std::vector<cv::Point2f> pts;
    auto t = device_tree_for(kdtree, pts);
    const int n = t->count;
    Layout in, res, work;
    const size_t o_nq = in.add(4), o_q = in.add(8);
    const size_t o_cnt = res.add(4), o_hits = res.add(4 * (size_t)n);
    Call c(in, res, work);
    *c.hin<int32_t>(o_nq) = 1;
    std::memcpy(c.hin<float>(o_q), &query_pt, 8);
    c.upload();
    check(vslam_kdtree_radius(ctx(), t->nodes, t->xy, t->n, 1, t->stride, c.din<float>(o_q), c.din<int32_t>(o_nq), 1, radius,
                              c.dout<int32_t>(o_hits), c.dout<int32_t>(o_cnt), n),
          "kdtree_radius");
    c.download();
    const int32_t cnt = *c.hout<int32_t>(o_cnt), *hits = c.hout<int32_t>(o_hits);
    out.reserve(cnt);
    for (int i = 0; i < cnt; i++) out.push_back(pts[hits[i]]);
    return out;
}

cv::Point2f nearest(const KDTree &kdtree, const cv::Point2f &query_pt, float max_distance_sq) {
    cv::Point2f r;   // default {0,0} when nothing qualifies, src/KDTree.cpp:38-42
    if (!kdtree.root) return r;
    Lock lk(g_mu);
    std::vector<cv::Point2f> pts;
    auto t = device_tree_for(kdtree, pts);
    Layout in, res, work;
    const size_t o_nq = in.add(4), o_q = in.add(8);
    const size_t o_best = res.add(4);
    Call c(in, res, work);
    *c.hin<int32_t>(o_nq) = 1;
    std::memcpy(c.hin<float>(o_q), &query_pt, 8);
    c.upload();
    check(vslam_kdtree_nearest(ctx(), t->nodes, t->xy, t->n, 1, t->stride, c.din<float>(o_q), c.din<int32_t>(o_nq), 1,
                               max_distance_sq, c.dout<int32_t>(o_best)),
          "kdtree_nearest");
    c.download();
    const int32_t best = *c.hout<int32_t>(o_best);
    if (best >= 0) r = pts[best];
    return r;
}

// ------------------------------------------------------------------------------- RansacFilter.h
RansacFilter::RansacFilter(const int min_items_, const int max_iterations_, const float threshold_)
    : min_items(min_items_), max_iterations(max_iterations_), threshold(threshold_) {}

u32 RansacFilter::next_seed() {
    if (has_seed_) return seed_;
    std::random_device rd;   // src/RansacFilter.cpp:15
    return (u32)rd();
}

namespace {
// The reference draws min_items indices into sets that are 8 wide whatever min_items is (src/RansacFilter.cpp:17,22):
// min_items < 8 leaves the other entries 0 (so every hypothesis also uses match 0), min_items > 8 writes past the set
// (undefined), and fewer matches than min_items builds a distribution over (0, -1) (:24, undefined).
int drawn_items(const RansacFilter &rf) { return rf.min_items < 0 ? 0 : rf.min_items; }
void require_sets(const RansacFilter &rf, int n_matches) {
    if (rf.min_items > VSLAM_SET_SIZE) throw std::invalid_argument("RansacFilter: min_items > 8 overruns the 8-wide sets (undefined in the reference)");
    if (n_matches < std::max(drawn_items(rf), 1)) throw std::invalid_argument("RansacFilter: fewer matches than min_items (undefined in the reference)");
}
// the context's set-drawing and minimum-match options follow the filter for the duration of one adapter call
struct FilterOptions {
    explicit FilterOptions(const RansacFilter &rf) {
        check(vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_ITEMS, drawn_items(rf)), "set_option");
        check(vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_MATCHES, std::max(drawn_items(rf), 1)), "set_option");
    }
    ~FilterOptions() {
        vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_ITEMS, VSLAM_SET_SIZE);
        vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_MATCHES, VSLAM_SET_SIZE);
    }
};
}  // namespace

void RansacFilter::initialize_sets(const int n_matches) {
    require_sets(*this, n_matches);
    Lock lk(g_mu);
    FilterOptions opts(*this);
    const size_t H = (size_t)max_iterations;
    Layout in, res, work;
    const size_t o_seed = in.add(4), o_m = in.add(4);
    const size_t o_sets = res.add(32 * H);
    const size_t o_draw = work.add(32 * H);
    Call c(in, res, work);
    *c.hin<uint32_t>(o_seed) = next_seed();
    *c.hin<int32_t>(o_m) = n_matches;
    c.upload();
    check(vslam_ransac_sets(ctx(), c.din<uint32_t>(o_seed), c.din<int32_t>(o_m), 1, max_iterations, c.dout<int32_t>(o_sets),
                            c.work<uint32_t>(o_draw)),
          "ransac_sets");
    c.download();
    vslam::detail::RansacAccess::store_sets(*this, c.hout<int32_t>(o_sets));
}

void RansacFilter::find_fundamental(const std::vector<cv::Point2f> &p1, const std::vector<cv::Point2f> &p2,
                                    const std::vector<std::pair<int, int>> &matches, std::vector<bool> &inliers,
                                    cv::Mat &fundamental) {
    const int H = max_iterations, M = (int)matches.size();
    require_sets(*this, M);   // initialize_sets(matches.size()), src/RansacFilter.cpp:38
    Lock lk(g_mu);
    FilterOptions opts(*this);
    const int stride = (int)std::max(std::max(p1.size(), p2.size()), (size_t)M);
    Layout in, res, work;
    const size_t o_seed = in.add(4), o_m = in.add(4), o_xy1 = in.add(8 * (size_t)stride), o_xy2 = in.add(8 * (size_t)stride),
                 o_pairs = in.add(8 * (size_t)stride);
    const size_t o_best = res.add(16), o_F = res.add(36), o_mask = res.add((size_t)stride), o_sets = res.add(32 * (size_t)H);
    const size_t o_draw = work.add(32 * (size_t)H), o_hypF = work.add(36 * (size_t)H), o_cnt = work.add(4 * (size_t)H),
                 o_sum = work.add(4 * (size_t)H), o_match = work.add(8 * (size_t)stride);
    Call c(in, res, work);
    *c.hin<uint32_t>(o_seed) = next_seed();
    *c.hin<int32_t>(o_m) = M;
    if (!p1.empty()) std::memcpy(c.hin<float>(o_xy1), p1.data(), 8 * p1.size());
    if (!p2.empty()) std::memcpy(c.hin<float>(o_xy2), p2.data(), 8 * p2.size());
    int32_t *flat = c.hin<int32_t>(o_pairs);
    for (int i = 0; i < M; i++) {
        flat[2 * i] = matches[i].first;
        flat[2 * i + 1] = matches[i].second;
    }
    c.upload();
    // sets are drawn on the device and go straight into the hypothesis loop
    check(vslam_ransac_sets(ctx(), c.din<uint32_t>(o_seed), c.din<int32_t>(o_m), 1, H, c.dout<int32_t>(o_sets), c.work<uint32_t>(o_draw)),
          "ransac_sets");
    check(vslam_ransac_fundamental(ctx(), c.din<float>(o_xy1), c.din<float>(o_xy2), c.din<int32_t>(o_pairs), c.din<int32_t>(o_m),
                                   c.dout<int32_t>(o_sets), 1, stride, H, threshold, c.dout<float>(o_F), c.dout<uint8_t>(o_mask),
                                   c.dout<int32_t>(o_best), c.work<int32_t>(o_match), c.work<float>(o_hypF), c.work<int32_t>(o_cnt),
                                   c.work<float>(o_sum)),
          "ransac_fundamental");
    c.download();
    vslam::detail::RansacAccess::store_sets(*this, c.hout<int32_t>(o_sets));
    const int32_t *best = c.hout<int32_t>(o_best);
    if (best[0] < 0) return;   // nothing accepted: `fundamental` and `inliers` stay as they were (:59-65)
    fundamental.create(3, 3, CV_32FC1);
    std::memcpy(fundamental.ptr<float>(), c.hout<float>(o_F), 36);
    const uint8_t *mask = c.hout<uint8_t>(o_mask);
    inliers.assign(M, false);
    for (int i = 0; i < M; i++) inliers[i] = mask[i] != 0;
}

void RansacFilter::compute_fundamental(const std::vector<cv::Point2f> &p1_set, const std::vector<cv::Point2f> &p2_set,
                                       cv::Mat &temp_F) {
    if (p1_set.size() != 8 || p2_set.size() != 8) throw std::invalid_argument("compute_fundamental: the device solver takes 8-point sets");
    Lock lk(g_mu);
    Layout in, res, work;
    const size_t o_m = in.add(4), o_xy1 = in.add(64), o_xy2 = in.add(64), o_pairs = in.add(64), o_set = in.add(32);
    const size_t o_F = res.add(36);
    Call c(in, res, work);
    *c.hin<int32_t>(o_m) = 8;
    std::memcpy(c.hin<float>(o_xy1), p1_set.data(), 64);
    std::memcpy(c.hin<float>(o_xy2), p2_set.data(), 64);
    for (int i = 0; i < 8; i++) {
        c.hin<int32_t>(o_pairs)[2 * i] = c.hin<int32_t>(o_pairs)[2 * i + 1] = i;
        c.hin<int32_t>(o_set)[i] = i;
    }
    c.upload();
    check(vslam_ransac_solve(ctx(), c.din<float>(o_xy1), c.din<float>(o_xy2), c.din<int32_t>(o_pairs), c.din<int32_t>(o_m),
                             c.din<int32_t>(o_set), 1, 8, 1, c.dout<float>(o_F)),
          "ransac_solve");
    c.download();
    temp_F.create(3, 3, CV_32FC1);
    std::memcpy(temp_F.ptr<float>(), c.hout<float>(o_F), 36);
}

std::pair<int, float> RansacFilter::compute_fundamental_residual(const std::vector<cv::Point2f> &p1,
                                                                 const std::vector<cv::Point2f> &p2,
                                                                 const std::vector<std::pair<int, int>> &matches,
                                                                 const cv::Mat &F, std::vector<bool> &inliers) {
    const int M = (int)matches.size();
    inliers.resize(M);   // :107
    if (M == 0) return {0, 0.f};   // nothing to evaluate: no inliers, cv::sum of an empty row is 0 (:128-138)
    Lock lk(g_mu);
    const int stride = (int)std::max(std::max(p1.size(), p2.size()), (size_t)M);
    Layout in, res, work;
    const size_t o_m = in.add(4), o_xy1 = in.add(8 * (size_t)stride), o_xy2 = in.add(8 * (size_t)stride),
                 o_pairs = in.add(8 * (size_t)stride), o_hypF = in.add(36);
    const size_t o_cnt = res.add(4), o_sum = res.add(4), o_mask = res.add((size_t)stride);
    const size_t o_best = work.add(16), o_F = work.add(36), o_match = work.add(8 * (size_t)stride);
    Call c(in, res, work);
    *c.hin<int32_t>(o_m) = M;
    if (!p1.empty()) std::memcpy(c.hin<float>(o_xy1), p1.data(), 8 * p1.size());
    if (!p2.empty()) std::memcpy(c.hin<float>(o_xy2), p2.data(), 8 * p2.size());
    int32_t *flat = c.hin<int32_t>(o_pairs);
    for (int i = 0; i < M; i++) {
        flat[2 * i] = matches[i].first;
        flat[2 * i + 1] = matches[i].second;
    }
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 3; col++) c.hin<float>(o_hypF)[r * 3 + col] = F.at<float>(r, col);
    c.upload();
    // a GIVEN F is scored on any number of matches (the reference's method has no lower limit, :105-140); with one
    // hypothesis its count is the maximum, so its sum is computed, and the winner's mask is this hypothesis' mask
    // whenever anything is an inlier (otherwise it is all false, which is also this hypothesis' mask)
    check(vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_MATCHES, 1), "set_option");
    const int rc = vslam_ransac_evaluate(ctx(), c.din<float>(o_xy1), c.din<float>(o_xy2), c.din<int32_t>(o_pairs), c.din<int32_t>(o_m),
                                         c.din<float>(o_hypF), 1, stride, 1, threshold, c.work<float>(o_F), c.dout<uint8_t>(o_mask),
                                         c.work<int32_t>(o_best), c.work<int32_t>(o_match), c.dout<int32_t>(o_cnt), c.dout<float>(o_sum));
    check(vslam_ctx_set_option(ctx(), VSLAM_OPT_RANSAC_MIN_MATCHES, VSLAM_SET_SIZE), "set_option");
    check(rc, "ransac_evaluate");
    c.download();
    const uint8_t *mask = c.hout<uint8_t>(o_mask);
    for (int i = 0; i < M; i++) inliers[i] = mask[i] != 0;
    return {*c.hout<int32_t>(o_cnt), *c.hout<float>(o_sum)};
}

// -------------------------------------------------------------------------------------- Frame.h
void initialize_frame(Frame &frame, const cv::Mat &image, long frame_id) {
    frame.image = image;   // shallow, aliases the capture buffer (src/Frame.cpp:4)
    frame.id = (u64)frame_id;
}

void draw(const Frame &frame, cv::Mat &annotated) {   // src/Frame.cpp:8-13; display only, nothing here touches the device
    frame.image.copyTo(annotated);
#ifdef VSLAM_HAVE_OPENCV
    for (const auto &p : frame.points) cv::circle(annotated, p, 2, cv::Scalar(0, 255, 0));
#else
    if (annotated.empty()) return;
    // cv::circle, radius 2, thickness 1: the midpoint walk plots (+-2, 0), (0, +-2) and then (+-1, +-1)
    static const int ring[8][2] = {{2, 0}, {-2, 0}, {0, 2}, {0, -2}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    const int cn = annotated.channels();
    for (const auto &p : frame.points) {
        const cv::Point c(p);   // cvRound, as cv::circle's Point parameter converts
        for (const auto &d : ring) {
            const int x = c.x + d[0], y = c.y + d[1];
            if (x < 0 || y < 0 || x >= annotated.cols || y >= annotated.rows) continue;
            unsigned char *px = annotated.ptr<unsigned char>(y) + (size_t)x * cn;
            if (annotated.depth() != CV_8U) continue;
            px[0] = 0;                       // Scalar(0, 255, 0): B, G, R
            if (cn > 1) px[1] = 255;
            if (cn > 2) px[2] = 0;
        }
    }
#endif
}

void extract_features(Frame &frame, int nrows, int ncols) {
    cv::Mat &img = frame.image;
    if (img.empty() || img.type() != CV_8UC3) throw std::invalid_argument("extract_features: expects a CV_8UC3 BGR image");
    Lock lk(g_mu);
    const int w = img.cols, h = img.rows;
    const int K = 500 * nrows * ncols + 4096;   // 500 per cell plus room for response ties
    const size_t img_bytes = (size_t)h * img.step;
    Layout in, res, work;
    in.add(0);
    const size_t o_n = res.add(4), o_xy = res.add(8 * (size_t)K), o_desc = res.add(32 * (size_t)K);
    Call c(in, res, work);
    // the image goes up and (outlines drawn, src/Frame.cpp:32) comes back through its own buffer
    uint8_t *dimg = static_cast<uint8_t *>(dev_scratch("frame.image", img_bytes));
    check(vslam_copy_h2d(ctx(), dimg, img.data, img_bytes), "image upload");
    check(vslam_extract_features_grid(ctx(), dimg, 1, w, h, (int)img.step, nrows, ncols, device_pattern(), K, c.dout<float>(o_xy),
                                      c.dout<uint8_t>(o_desc), nullptr, c.dout<int32_t>(o_n)),
          "extract_features_grid");
    check(vslam_copy_d2h(ctx(), img.data, dimg, img_bytes), "image download");
    c.download();
    const int32_t n = *c.hout<int32_t>(o_n);
    frame.descriptors.create(n, 32, CV_8UC1);
    if (n) std::memcpy(frame.descriptors.data, c.hout<uint8_t>(o_desc), (size_t)n * 32);
    const size_t old = frame.points.size();
    frame.points.resize(old + n);   // push_back loop, :47-49; no k-d tree, no map_point_ids (as the reference)
    if (n) std::memcpy(static_cast<void *>(frame.points.data() + old), c.hout<float>(o_xy), 8 * (size_t)n);
}

void extract_features(Frame &frame) {
    const cv::Mat &img = frame.image;
    if (img.empty() || img.type() != CV_8UC3) throw std::invalid_argument("extract_features: expects a CV_8UC3 BGR image");
    Lock lk(g_mu);
    auto &st = vslam::settings();
    const int w = img.cols, h = img.rows, K = st.max_corners;
    const size_t img_bytes = (size_t)h * img.step;
    Layout in, res, work;
    in.add(0);
    const size_t o_n = res.add(4), o_nd = res.add(4), o_xy = res.add(8 * (size_t)K), o_desc = res.add(32 * (size_t)K),
                 o_nodes = res.add(4 * (size_t)K);
    int slots = 64;   // the tree's points by pixel cell, for the single radius queries of src/vslam.cpp:149
    while (slots < 2 * K) slots <<= 1;
    const size_t o_tab = res.add(8 * (size_t)slots), o_ok = res.add(4);
    Call c(in, res, work);
    uint8_t *dimg = static_cast<uint8_t *>(dev_scratch("frame.image", img_bytes));
    check(vslam_copy_h2d(ctx(), dimg, img.data, img_bytes), "image upload");
    vslam_extract_params p;
    fill_params(p, K, device_pattern());
    check(vslam_extract_features(ctx(), dimg, 1, w, h, (int)img.step, &p, K, c.dout<float>(o_xy), c.dout<uint8_t>(o_desc),
                                 c.dout<int32_t>(o_nodes), c.dout<int32_t>(o_n), c.dout<int32_t>(o_nd)),
          "extract_features");
    check(vslam_kdtree_cell_table(ctx(), c.dout<int32_t>(o_nodes), c.dout<float>(o_xy), c.dout<int32_t>(o_n), 1, K, slots,
                                  c.dout<uint32_t>(o_tab), c.dout<int32_t>(o_ok)),
          "kdtree_cell_table");
    c.download();
    const int32_t n = *c.hout<int32_t>(o_n), nd = *c.hout<int32_t>(o_nd);
    const size_t old = frame.points.size();
    frame.points.resize(old + n);   // push_back loop, src/Frame.cpp:69-72
    if (n) std::memcpy(static_cast<void *>(frame.points.data() + old), c.hout<float>(o_xy), 8 * (size_t)n);
    frame.descriptors.create(n, 32, CV_8UC1);
    if (n) std::memcpy(frame.descriptors.data, c.hout<uint8_t>(o_desc), (size_t)n * 32);
    frame.map_point_ids.resize(nd, -1);   // sized from the PRE-filter count, :73
    // k-d tree (:76): the device already built it over the kept points
    if (old == 0 && n > 0) {
        const int32_t *pre = c.hout<int32_t>(o_nodes);
        auto *nodes = static_cast<frame_kdtree::KDTreeNode *>(std::malloc((size_t)n * sizeof(frame_kdtree::KDTreeNode)));
        for (int i = 0; i < n; i++) nodes[i].pt_index = (usize)pre[i];
        link_preorder(nodes, 0, n);
        frame.kdtree.root = nodes;
        frame.kdtree.size += (u32)n;
        frame.kdtree.height = tree_height(n);
        // its device copy is made by the first batched query that needs it; single queries (src/vslam.cpp:149) are
        // answered from the cell table that came down with this call
        auto t = std::make_shared<DevTree>();
        t->count = n;
        t->stride = n;
        t->hash = hash_tree(frame.kdtree, n, frame.points);
        if (*c.hout<int32_t>(o_ok) != 0) {
            t->cells.assign(c.hout<uint32_t>(o_tab), c.hout<uint32_t>(o_tab) + 2 * (size_t)slots);
            t->cell_mask = (uint32_t)slots - 1u;
            t->pre.assign(pre, pre + n);
            t->pts_ptr = frame.points.data();
            t->pts_size = frame.points.size();
            t->pts_sample = sample_points(frame.points);
        }
        remember_tree(nodes, t);
    } else {
        construct_kdtree(frame.kdtree, frame.points);
    }
}

void match_features(const Frame &frame1, const Frame &frame2, RansacFilter &rf,
                    std::vector<std::pair<int, int>> &matches, cv::Mat &F) {
    const int n1 = (int)frame1.points.size(), n2 = (int)frame2.points.size();
    if (frame1.descriptors.rows != n1 || frame2.descriptors.rows != n2)
        throw std::invalid_argument("match_features: descriptors and points disagree");
    if (rf.min_items > VSLAM_SET_SIZE) throw std::invalid_argument("RansacFilter: min_items > 8 overruns the 8-wide sets (undefined in the reference)");
    Lock lk(g_mu);
    FilterOptions opts(rf);
    const int K = std::max(std::max(n1, n2), 1), H = rf.max_iterations;
    // knnMatch + ratio (src/Frame.cpp:83-94), then rf.find_fundamental (:97) with rf's seed, iterations and threshold,
    // then the inlier filter (:98-102) — one upload, the kernels chained on the device, one download
    Layout in, res, work;
    const size_t o_n1 = in.add(4), o_n2 = in.add(4), o_seed = in.add(4), o_d1 = in.add(32 * (size_t)K), o_d2 = in.add(32 * (size_t)K),
                 o_xy1 = in.add(8 * (size_t)K), o_xy2 = in.add(8 * (size_t)K);
    const size_t o_m = res.add(4), o_best = res.add(16), o_F = res.add(36), o_kept = res.add(8 * (size_t)K), o_sets = res.add(32 * (size_t)H);
    const size_t o_pairs = work.add(8 * (size_t)K), o_mask = work.add((size_t)K), o_draw = work.add(32 * (size_t)H),
                 o_hypF = work.add(36 * (size_t)H), o_cnt = work.add(4 * (size_t)H), o_sum = work.add(4 * (size_t)H);
    Call c(in, res, work);
    *c.hin<int32_t>(o_n1) = n1;
    *c.hin<int32_t>(o_n2) = n2;
    *c.hin<uint32_t>(o_seed) = vslam::detail::RansacAccess::seed(rf);
    if (n1) {
        std::memcpy(c.hin<uint8_t>(o_d1), frame1.descriptors.data, (size_t)n1 * 32);
        std::memcpy(c.hin<float>(o_xy1), frame1.points.data(), 8 * (size_t)n1);
    }
    if (n2) {
        std::memcpy(c.hin<uint8_t>(o_d2), frame2.descriptors.data, (size_t)n2 * 32);
        std::memcpy(c.hin<float>(o_xy2), frame2.points.data(), 8 * (size_t)n2);
    }
    c.upload();
    check(vslam_match_knn2_ratio(ctx(), c.din<uint8_t>(o_d1), c.din<int32_t>(o_n1), c.din<uint8_t>(o_d2), c.din<int32_t>(o_n2), 1, K,
                                 c.work<int32_t>(o_pairs), c.dout<int32_t>(o_m), nullptr),
          "match_knn2_ratio");
    check(vslam_ransac_sets(ctx(), c.din<uint32_t>(o_seed), c.dout<int32_t>(o_m), 1, H, c.dout<int32_t>(o_sets), c.work<uint32_t>(o_draw)),
          "ransac_sets");
    check(vslam_ransac_fundamental(ctx(), c.din<float>(o_xy1), c.din<float>(o_xy2), c.work<int32_t>(o_pairs), c.dout<int32_t>(o_m),
                                   c.dout<int32_t>(o_sets), 1, K, H, rf.threshold, c.dout<float>(o_F), c.work<uint8_t>(o_mask),
                                   c.dout<int32_t>(o_best), c.dout<int32_t>(o_kept), c.work<float>(o_hypF), c.work<int32_t>(o_cnt),
                                   c.work<float>(o_sum)),
          "ransac_fundamental");
    c.download();
    if (*c.hout<int32_t>(o_m) < std::max(drawn_items(rf), 1))
        throw std::invalid_argument("RansacFilter: fewer matches than min_items (undefined in the reference)");
    vslam::detail::RansacAccess::store_sets(rf, c.hout<int32_t>(o_sets));
    const int32_t *best = c.hout<int32_t>(o_best);
    if (best[0] < 0) return;   // nothing accepted: F untouched, no inliers (src/RansacFilter.cpp:59-65, src/Frame.cpp:98)
    F.create(3, 3, CV_32FC1);
    std::memcpy(F.ptr<float>(), c.hout<float>(o_F), 36);
    const int32_t *kept = c.hout<int32_t>(o_kept);
    for (int i = 0; i < best[3]; i++) matches.emplace_back(kept[2 * i], kept[2 * i + 1]);   // appended, not cleared (:98-102)
}

// ------------------------------------------------------------------------------------ helpers.h
namespace {
// a 32-bit float matrix of the given shape, row by row into `dst` (rows may be padded: cv::Mat::step)
void read_f32(const cv::Mat &m, int rows, int cols, float *dst, const char *what) {
    if (m.empty() || m.rows != rows || m.cols != cols || m.depth() != CV_32F || m.channels() != 1)
        throw std::invalid_argument(std::string("vslam_amd: ") + what + " must be a " + std::to_string(rows) + " x " +
                                    std::to_string(cols) + " CV_32F matrix");
    for (int r = 0; r < rows; r++) std::memcpy(dst + (size_t)r * cols, m.ptr<float>(r), sizeof(float) * (size_t)cols);
}
}  // namespace

// reference: src/helpers.cpp:3-35
void extract_Rt(const cv::Mat &fundamental, const cv::Mat &K, cv::Mat &rotation, cv::Mat &translation) {
    Lock lk(g_mu);
    float Kh[9];
    read_f32(K, 3, 3, Kh, "K");
    Layout in, out, work;
    const size_t o_F = in.add(sizeof(float) * 9);
    const size_t o_R = out.add(sizeof(float) * 9), o_t = out.add(sizeof(float) * 3), o_c2 = out.add(sizeof(float) * 12);
    Call c(in, out, work);
    read_f32(fundamental, 3, 3, c.hin<float>(o_F), "fundamental");
    c.upload();
    check(vslam_extract_Rt(ctx(), c.din<float>(o_F), nullptr, 1, Kh, c.dout<float>(o_R), c.dout<float>(o_t), c.dout<float>(o_c2)),
          "extract_Rt");
    c.download();
    rotation.create(3, 3, CV_32FC1);
    translation.create(3, 1, CV_32FC1);
    for (int r = 0; r < 3; r++) {
        std::memcpy(rotation.ptr<float>(r), c.hout<float>(o_R) + 3 * r, sizeof(float) * 3);
        translation.ptr<float>(r)[0] = c.hout<float>(o_t)[r];
    }
}

// reference: src/helpers.cpp:37-80
void triangulate(const cv::Mat &p1, const cv::Mat &p2, const cv::Mat &c1, const cv::Mat &c2, cv::Mat &points_4d) {
    Lock lk(g_mu);
    float c1h[12], c2h[12];
    read_f32(c1, 3, 4, c1h, "c1");
    read_f32(c2, 3, 4, c2h, "c2");
    const int n = p1.rows;
    points_4d.create(n, 4, CV_32FC1);   // points_4d = cv::Mat(N, 4, CV_32F), :41
    if (n == 0) return;
    Layout in, out, work;
    const size_t o_p1 = in.add(sizeof(float) * 2 * (size_t)n), o_p2 = in.add(sizeof(float) * 2 * (size_t)n);
    const size_t o_pts = out.add(sizeof(float) * 4 * (size_t)n);
    Call c(in, out, work);
    read_f32(p1, n, 2, c.hin<float>(o_p1), "p1");
    read_f32(p2, n, 2, c.hin<float>(o_p2), "p2");
    c.upload();
    check(vslam_triangulate_points(ctx(), c.din<float>(o_p1), c.din<float>(o_p2), n, c1h, c2h, c.dout<float>(o_pts)),
          "triangulate_points");
    c.download();
    for (int r = 0; r < n; r++) std::memcpy(points_4d.ptr<float>(r), c.hout<float>(o_pts) + 4 * (size_t)r, sizeof(float) * 4);
}

namespace vslam {
// reference: src/vslam.cpp:129-161, src/PointMap.cpp:36-46
std::vector<s32> associate_map_points(Frame &frame, const cv::Mat &map_points, const cv::Mat &c2, int W, int H,
                                      const std::vector<u32> &obs_offsets, const cv::Mat &obs_desc, float radius,
                                      u32 dist_threshold) {
    const int n_map = map_points.rows;
    std::vector<s32> claim((size_t)n_map, -1);
    const int n_kp = (int)frame.points.size();
    if (n_map == 0 || n_kp == 0 || frame.kdtree.root == nullptr) return claim;
    if (obs_offsets.size() != (size_t)n_map + 1)
        throw std::invalid_argument("vslam_amd: obs_offsets needs one entry per map point plus one");
    if (frame.map_point_ids.size() < (size_t)n_kp || frame.descriptors.rows < n_kp || frame.descriptors.cols != VSLAM_DESC_BYTES)
        throw std::invalid_argument("vslam_amd: frame.map_point_ids / frame.descriptors do not cover frame.points");
    const int n_obs = (int)obs_offsets.back();
    if (obs_desc.rows < n_obs || (n_obs > 0 && obs_desc.cols != VSLAM_DESC_BYTES))
        throw std::invalid_argument("vslam_amd: obs_desc does not hold the rows obs_offsets names");
    Lock lk(g_mu);
    auto t = device_tree_for(frame.kdtree, frame.points);
    const int kp_stride = t->stride, obs_stride = n_obs > 0 ? n_obs : 1;
    float c2h[12];
    read_f32(c2, 3, 4, c2h, "c2");
    Layout in, out, work;
    const size_t o_nmap = in.add(4), o_mp = in.add(sizeof(float) * 4 * (size_t)n_map), o_c2 = in.add(sizeof(float) * 12);
    const size_t o_desc = in.add((size_t)kp_stride * VSLAM_DESC_BYTES), o_off = in.add(4 * ((size_t)n_map + 1));
    const size_t o_obs = in.add((size_t)obs_stride * VSLAM_DESC_BYTES), o_ids_in = in.add(4 * (size_t)kp_stride);
    const size_t o_claim = out.add(4 * (size_t)n_map);
    Call c(in, out, work);
    *c.hin<int32_t>(o_nmap) = n_map;
    read_f32(map_points, n_map, 4, c.hin<float>(o_mp), "map_points");
    std::memcpy(c.hin<float>(o_c2), c2h, sizeof(c2h));
    for (int r = 0; r < n_kp; r++) std::memcpy(c.hin<uint8_t>(o_desc) + (size_t)r * VSLAM_DESC_BYTES, frame.descriptors.ptr<uint8_t>(r), VSLAM_DESC_BYTES);
    for (int i = 0; i <= n_map; i++) c.hin<int32_t>(o_off)[i] = (int32_t)obs_offsets[(size_t)i];
    for (int r = 0; r < n_obs; r++) std::memcpy(c.hin<uint8_t>(o_obs) + (size_t)r * VSLAM_DESC_BYTES, obs_desc.ptr<uint8_t>(r), VSLAM_DESC_BYTES);
    for (int i = 0; i < kp_stride; i++) c.hin<int32_t>(o_ids_in)[i] = i < n_kp ? frame.map_point_ids[(size_t)i] : 0;
    c.upload();
    // map_point_ids is in/out: the kernel updates the uploaded copy in place, which then comes back by itself
    check(vslam_associate_map_points(ctx(), c.din<float>(o_mp), c.din<int32_t>(o_nmap), 1, n_map, c.din<float>(o_c2), W, H, t->nodes,
                                     t->xy, c.din<uint8_t>(o_desc), t->n, kp_stride, c.din<int32_t>(o_off), c.din<uint8_t>(o_obs),
                                     obs_stride, radius, dist_threshold, c.din<int32_t>(o_ids_in), c.dout<int32_t>(o_claim)),
          "associate_map_points");
    c.download();
    if (c.din<int32_t>(o_ids_in) != c.hin<int32_t>(o_ids_in))   // (a small call runs on the page-locked staging memory itself)
        check(vslam_copy_d2h(ctx(), c.hin<int32_t>(o_ids_in), c.din<int32_t>(o_ids_in), 4 * (size_t)kp_stride), "copy_d2h");
    check(vslam_ctx_synchronize(ctx()), "associate_map_points (more than 16 acceptable hits for one map point)");
    for (int i = 0; i < n_kp; i++) frame.map_point_ids[(size_t)i] = c.hin<int32_t>(o_ids_in)[i];
    for (int i = 0; i < n_map; i++) claim[(size_t)i] = c.hout<int32_t>(o_claim)[i];
    return claim;
}
}  // namespace vslam

// Host side of the drop-in: implements the reference's Frame / KDTree / RansacFilter free functions
// and class (include/vslam/*.h) on top of the C ABI (include/vslam_amd.h).  Everything that
// computes runs on the device; this file only marshals std::vector / cv::Mat data across the
// boundary and rebuilds the pointer-linked node arrays the consumers expect.  No CPU fallback:
// if no device is available the first call throws.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>

#include "../../include/vslam/Frame.h"
#include "../../include/vslam_amd.h"
#include "host_internal.h"

namespace {

struct Device {
    vslam_ctx *ctx = nullptr;
    Device() {
        const char *d = std::getenv("VSLAM_DEVICE");
        vslam::settings().device = d ? std::atoi(d) : vslam::settings().device;
        const int rc = vslam_ctx_create(vslam::settings().device, &ctx);
        if (rc != VSLAM_OK)
            throw std::runtime_error("vslam_amd: no usable HIP device (vslam_ctx_create rc=" + std::to_string(rc) +
                                     "); the front-end has no CPU fallback");
    }
    ~Device() {
        if (ctx) vslam_ctx_destroy(ctx);
    }
};

vslam_ctx *ctx() {
    static Device dev;
    return dev.ctx;
}

void check(int rc, const char *what) {
    if (rc != VSLAM_OK)
        throw std::runtime_error(std::string("vslam_amd: ") + what + " failed (rc=" + std::to_string(rc) + "): " +
                                 vslam_last_error(ctx()));
}

// RAII device buffer
template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() {}
    explicit DBuf(size_t count) { alloc(count); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    void alloc(size_t count) {
        release();
        n = count;
        check(vslam_dev_alloc(ctx(), sizeof(T) * (count ? count : 1), reinterpret_cast<void **>(&p)), "dev_alloc");
    }
    void release() {
        if (p) vslam_dev_free(ctx(), p);
        p = nullptr;
    }
    ~DBuf() { release(); }
    void upload(const T *h, size_t count) { check(vslam_copy_h2d(ctx(), p, h, sizeof(T) * count), "copy_h2d"); }
    void download(T *h, size_t count) const { check(vslam_copy_d2h(ctx(), h, p, sizeof(T) * count), "copy_d2h"); }
};

// device copies of trees built through this adapter, keyed by the host root pointer
struct DevTree {
    DBuf<int32_t> nodes;
    DBuf<float> xy;
    DBuf<int32_t> n;
    int count = 0, stride = 0;
};
std::mutex g_mu;
std::map<const void *, std::shared_ptr<DevTree>> g_trees;

std::shared_ptr<DevTree> upload_tree(const std::vector<int32_t> &pre_idx, const float *xy, int n) {
    auto t = std::make_shared<DevTree>();
    t->count = n;
    t->stride = n > 0 ? n : 1;
    t->nodes.alloc(t->stride);
    t->xy.alloc(2 * (size_t)t->stride);
    t->n.alloc(1);
    if (n > 0) {
        t->nodes.upload(pre_idx.data(), n);
        t->xy.upload(xy, 2 * (size_t)n);
    }
    const int32_t nn = n;
    t->n.upload(&nn, 1);
    return t;
}

// build on the device, return the pre-order pt_index column
std::vector<int32_t> device_build(const std::vector<cv::Point2f> &points, std::shared_ptr<DevTree> *keep) {
    const int n = (int)points.size();
    std::vector<int32_t> pre(n);
    auto t = std::make_shared<DevTree>();
    t->count = n;
    t->stride = n > 0 ? n : 1;
    t->nodes.alloc(t->stride);
    t->xy.alloc(2 * (size_t)t->stride);
    t->n.alloc(1);
    const int32_t nn = n;
    t->n.upload(&nn, 1);
    if (n > 0) {
        t->xy.upload(reinterpret_cast<const float *>(points.data()), 2 * (size_t)n);
        check(vslam_kdtree_build(ctx(), t->xy.p, t->n.p, 1, t->stride, t->nodes.p), "kdtree_build");
        t->nodes.download(pre.data(), n);
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
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_trees.find(kd.root);
        if (it != g_trees.end() && it->second->count == (int)points.size()) return it->second;
    }
    // tree built elsewhere (or forgotten): its array is already in pre-order, re-upload it
    const int n = (int)points.size();
    std::vector<int32_t> pre(n);
    for (int i = 0; i < n; i++) pre[i] = (int32_t)kd.root[i].pt_index;
    auto t = upload_tree(pre, reinterpret_cast<const float *>(points.data()), n);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_trees.size() > 64) g_trees.clear();
    g_trees[kd.root] = t;
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
    // built-in seeded stand-in (NOT OpenCV's learned table): Gaussian offsets, sigma 31/5, clipped to +-13
    std::mt19937 g(0xB21EF);
    std::normal_distribution<float> nd(0.f, 31.f / 5.f);
    st.brief_pattern.resize(1024);
    for (auto &v : st.brief_pattern) {
        float x = std::nearbyint(nd(g));
        v = (s8)(x < -13 ? -13 : (x > 13 ? 13 : x));
    }
    return st.brief_pattern;
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
void fill_extract_params(vslam_extract_params &p, int max_corners, const int8_t *d_pattern) {
    auto &st = vslam::settings();
    p.max_corners = max_corners;
    p.quality = st.quality;
    p.min_distance = st.min_distance;
    const float a = st.keypoint_angle_deg * (float)(3.14159265358979323846 / 180.f);   // angle *= CV_PI/180
    p.cos_a = (float)std::cos((double)a);
    p.sin_a = (float)std::sin((double)a);
    p.d_pattern = d_pattern;
}
}  // namespace detail

void forget_kdtree(const void *root) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_trees.erase(root);
}

std::vector<std::vector<usize>> radius_search_batch(const frame_kdtree &kdtree,
                                                    const std::vector<cv::Point2f> &points,
                                                    const std::vector<cv::Point2f> &queries, float radius) {
    std::vector<std::vector<usize>> out(queries.size());
    if (queries.empty() || points.empty() || kdtree.root == nullptr) return out;
    auto t = device_tree_for(kdtree, points);
    const int q = (int)queries.size();
    int cap = 16;
    while (true) {
        DBuf<float> dq(2 * (size_t)q);
        DBuf<int32_t> dnq(1), dhits((size_t)q * cap), dcnt(q);
        dq.upload(reinterpret_cast<const float *>(queries.data()), 2 * (size_t)q);
        const int32_t nq = q;
        dnq.upload(&nq, 1);
        check(vslam_kdtree_radius(ctx(), t->nodes.p, t->xy.p, t->n.p, 1, t->stride, dq.p, dnq.p, q, radius, dhits.p,
                                  dcnt.p, cap),
              "kdtree_radius");
        std::vector<int32_t> hits((size_t)q * cap), cnt(q);
        dhits.download(hits.data(), hits.size());
        dcnt.download(cnt.data(), q);
        int mx = 0;
        for (int c : cnt) mx = c > mx ? c : mx;
        if (mx > cap) {   // rare: more hits than slots, retry with room for all
            cap = mx;
            continue;
        }
        for (int i = 0; i < q; i++) out[i].assign(hits.begin() + (size_t)i * cap, hits.begin() + (size_t)i * cap + cnt[i]);
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
    std::shared_ptr<DevTree> dev;
    const std::vector<int32_t> pre = device_build(points, &dev);
    auto *nodes = static_cast<frame_kdtree::KDTreeNode *>(std::malloc(N * sizeof(frame_kdtree::KDTreeNode)));
    for (usize i = 0; i < N; i++) nodes[i].pt_index = (usize)pre[i];
    link_preorder(nodes, 0, (int)N);
    kdtree.root = nodes;
    kdtree.size += (u32)N;   // the reference never resets size
    kdtree.height = tree_height((int)N);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_trees.size() > 64) g_trees.clear();
    g_trees[nodes] = dev;
}

void construct_kdtree(KDTree &kdtree, const std::vector<cv::Point2f> &points) {
    const int N = (int)points.size();
    if (N == 0) {
        kdtree.root = nullptr;
        return;
    }
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
    return vslam::radius_search_batch(kdtree, points, std::vector<cv::Point2f>{query_pt}, radius)[0];
}

namespace {
// a KDTree carries its points inside the nodes: flatten them (array order == pre-order)
void flatten(const KDTree &kd, int n, std::vector<cv::Point2f> &pts, std::vector<int32_t> &pre) {
    pts.resize(n);
    pre.resize(n);
    for (int i = 0; i < n; i++) {
        pts[i] = kd.root[i].pt;
        pre[i] = i;
    }
}
int node_count(const KDTree::KDTreeNode *nd) { return nd ? 1 + node_count(nd->left) + node_count(nd->right) : 0; }
}  // namespace

std::vector<cv::Point2f> radius_search(const KDTree &kdtree, const cv::Point2f &query_pt, float radius) {
    std::vector<cv::Point2f> out;
    if (!kdtree.root) return out;
    const int n = node_count(kdtree.root);
    std::vector<cv::Point2f> pts;
    std::vector<int32_t> pre;
    flatten(kdtree, n, pts, pre);
    auto t = upload_tree(pre, reinterpret_cast<const float *>(pts.data()), n);
    DBuf<float> dq(2);
    DBuf<int32_t> dnq(1), dcnt(1), dhits(n);
    dq.upload(&query_pt.x, 2);
    const int32_t one = 1;
    dnq.upload(&one, 1);
    check(vslam_kdtree_radius(ctx(), t->nodes.p, t->xy.p, t->n.p, 1, t->stride, dq.p, dnq.p, 1, radius, dhits.p, dcnt.p, n),
          "kdtree_radius");
    int32_t cnt = 0;
    dcnt.download(&cnt, 1);
    std::vector<int32_t> hits(cnt);
    if (cnt) dhits.download(hits.data(), cnt);
    for (int32_t h : hits) out.push_back(pts[h]);
    return out;
}

cv::Point2f nearest(const KDTree &kdtree, const cv::Point2f &query_pt, float max_distance_sq) {
    cv::Point2f r;   // default {0,0} when nothing qualifies, src/KDTree.cpp:38-42
    if (!kdtree.root) return r;
    const int n = node_count(kdtree.root);
    std::vector<cv::Point2f> pts;
    std::vector<int32_t> pre;
    flatten(kdtree, n, pts, pre);
    auto t = upload_tree(pre, reinterpret_cast<const float *>(pts.data()), n);
    DBuf<float> dq(2);
    DBuf<int32_t> dnq(1), dbest(1);
    dq.upload(&query_pt.x, 2);
    const int32_t one = 1;
    dnq.upload(&one, 1);
    check(vslam_kdtree_nearest(ctx(), t->nodes.p, t->xy.p, t->n.p, 1, t->stride, dq.p, dnq.p, 1, max_distance_sq, dbest.p),
          "kdtree_nearest");
    int32_t best = -1;
    dbest.download(&best, 1);
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

void RansacFilter::initialize_sets(const int n_matches) {
    if (min_items != VSLAM_SET_SIZE) throw std::invalid_argument("RansacFilter: the device path draws 8-subsets (min_items == 8)");
    if (n_matches < VSLAM_SET_SIZE) throw std::invalid_argument("RansacFilter: fewer than 8 matches (undefined in the reference)");
    const u32 seed = next_seed();
    DBuf<uint32_t> dseed(1), ddraw((size_t)max_iterations * 8);
    DBuf<int32_t> dm(1), dsets((size_t)max_iterations * 8);
    dseed.upload(&seed, 1);
    const int32_t m = n_matches;
    dm.upload(&m, 1);
    check(vslam_ransac_sets(ctx(), dseed.p, dm.p, 1, max_iterations, dsets.p, ddraw.p), "ransac_sets");
    std::vector<int32_t> flat((size_t)max_iterations * 8);
    dsets.download(flat.data(), flat.size());
    ransac_sets.assign(max_iterations, std::vector<int>(8, 0));
    for (int i = 0; i < max_iterations; i++)
        for (int j = 0; j < 8; j++) ransac_sets[i][j] = flat[(size_t)i * 8 + j];
}

namespace {
struct PairUpload {
    DBuf<float> xy1, xy2;
    DBuf<int32_t> pairs, m;
    int stride = 1;
    PairUpload(const std::vector<cv::Point2f> &p1, const std::vector<cv::Point2f> &p2,
               const std::vector<std::pair<int, int>> &matches) {
        stride = (int)std::max(std::max(p1.size(), p2.size()), std::max(matches.size(), (size_t)1));
        xy1.alloc(2 * (size_t)stride);
        xy2.alloc(2 * (size_t)stride);
        pairs.alloc(2 * (size_t)stride);
        m.alloc(1);
        if (!p1.empty()) xy1.upload(reinterpret_cast<const float *>(p1.data()), 2 * p1.size());
        if (!p2.empty()) xy2.upload(reinterpret_cast<const float *>(p2.data()), 2 * p2.size());
        std::vector<int32_t> flat(2 * matches.size());
        for (size_t i = 0; i < matches.size(); i++) {
            flat[2 * i] = matches[i].first;
            flat[2 * i + 1] = matches[i].second;
        }
        if (!flat.empty()) pairs.upload(flat.data(), flat.size());
        const int32_t mm = (int32_t)matches.size();
        m.upload(&mm, 1);
    }
};
}  // namespace

void RansacFilter::find_fundamental(const std::vector<cv::Point2f> &p1, const std::vector<cv::Point2f> &p2,
                                    const std::vector<std::pair<int, int>> &matches, std::vector<bool> &inliers,
                                    cv::Mat &fundamental) {
    initialize_sets((int)matches.size());   // src/RansacFilter.cpp:38
    const int H = max_iterations, M = (int)matches.size();
    PairUpload up(p1, p2, matches);
    std::vector<int32_t> flat((size_t)H * 8);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < 8; j++) flat[(size_t)i * 8 + j] = ransac_sets[i][j];
    DBuf<int32_t> dsets(flat.size()), dbest(4), dmatches(2 * (size_t)up.stride), dcount(H);
    DBuf<float> dF(9), dhypF((size_t)H * 9), dsum(H);
    DBuf<uint8_t> dmask(up.stride);
    dsets.upload(flat.data(), flat.size());
    check(vslam_ransac_fundamental(ctx(), up.xy1.p, up.xy2.p, up.pairs.p, up.m.p, dsets.p, 1, up.stride, H, threshold, dF.p,
                                   dmask.p, dbest.p, dmatches.p, dhypF.p, dcount.p, dsum.p),
          "ransac_fundamental");
    int32_t best[4];
    dbest.download(best, 4);
    if (best[0] < 0) return;   // nothing accepted: `fundamental` and `inliers` stay as they were (:59-65)
    float F[9];
    dF.download(F, 9);
    fundamental.create(3, 3, CV_32FC1);
    std::memcpy(fundamental.ptr<float>(), F, sizeof(F));
    std::vector<uint8_t> mask(M);
    dmask.download(mask.data(), M);
    inliers.assign(M, false);
    for (int i = 0; i < M; i++) inliers[i] = mask[i] != 0;
}

void RansacFilter::compute_fundamental(const std::vector<cv::Point2f> &p1_set, const std::vector<cv::Point2f> &p2_set,
                                       cv::Mat &temp_F) {
    if (p1_set.size() != 8 || p2_set.size() != 8) throw std::invalid_argument("compute_fundamental: the device solver takes 8-point sets");
    std::vector<std::pair<int, int>> ident(8);
    for (int i = 0; i < 8; i++) ident[i] = {i, i};
    PairUpload up(p1_set, p2_set, ident);
    const int32_t set[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    DBuf<int32_t> dsets(8);
    DBuf<float> dhypF(9);
    dsets.upload(set, 8);
    check(vslam_ransac_solve(ctx(), up.xy1.p, up.xy2.p, up.pairs.p, up.m.p, dsets.p, 1, up.stride, 1, dhypF.p), "ransac_solve");
    temp_F.create(3, 3, CV_32FC1);
    dhypF.download(temp_F.ptr<float>(), 9);
}

std::pair<int, float> RansacFilter::compute_fundamental_residual(const std::vector<cv::Point2f> &p1,
                                                                 const std::vector<cv::Point2f> &p2,
                                                                 const std::vector<std::pair<int, int>> &matches,
                                                                 const cv::Mat &F, std::vector<bool> &inliers) {
    const int M = (int)matches.size();
    inliers.resize(M);   // :107
    if (M < 8) throw std::invalid_argument("compute_fundamental_residual: the device path needs >= 8 matches");
    PairUpload up(p1, p2, matches);
    DBuf<float> dhypF(9), dF(9), dsum(1);
    DBuf<int32_t> dbest(4), dmatches(2 * (size_t)up.stride), dcount(1);
    DBuf<uint8_t> dmask(up.stride);
    float f[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) f[r * 3 + c] = F.at<float>(r, c);
    dhypF.upload(f, 9);
    // threshold -inf..: evaluate reports count and sum for the given F irrespective of acceptance
    check(vslam_ransac_evaluate(ctx(), up.xy1.p, up.xy2.p, up.pairs.p, up.m.p, dhypF.p, 1, up.stride, 1, threshold, dF.p,
                                dmask.p, dbest.p, dmatches.p, dcount.p, dsum.p),
          "ransac_evaluate");
    int32_t count = 0;
    float sum = 0;
    dcount.download(&count, 1);
    dsum.download(&sum, 1);
    // the mask of THIS hypothesis: recompute-free because with hyp == 1 the winner is hypothesis 0
    // whenever anything is accepted; otherwise every e is NaN/over threshold and the mask is all false
    std::vector<uint8_t> mask(M);
    dmask.download(mask.data(), M);
    for (int i = 0; i < M; i++) inliers[i] = mask[i] != 0;
    return {count, sum};
}

// -------------------------------------------------------------------------------------- Frame.h
void initialize_frame(Frame &frame, const cv::Mat &image, long frame_id) {
    frame.image = image;   // shallow, aliases the capture buffer (src/Frame.cpp:4)
    frame.id = (u64)frame_id;
}

void extract_features(Frame &frame, int nrows, int ncols) {
    cv::Mat &img = frame.image;
    if (img.empty() || img.type() != CV_8UC3) throw std::invalid_argument("extract_features: expects a CV_8UC3 BGR image");
    const int w = img.cols, h = img.rows;
    const int K = 500 * nrows * ncols + 4096;   // 500 per cell plus room for response ties
    const std::vector<s8> &pat = pattern();
    DBuf<uint8_t> dimg((size_t)h * img.step), ddesc((size_t)K * 32);
    DBuf<int8_t> dpat(1024);
    DBuf<float> dxy(2 * (size_t)K);
    DBuf<int32_t> dn(1);
    dimg.upload(img.data, (size_t)h * img.step);
    dpat.upload(reinterpret_cast<const int8_t *>(pat.data()), 1024);
    check(vslam_extract_features_grid(ctx(), dimg.p, 1, w, h, (int)img.step, nrows, ncols, dpat.p, K, dxy.p, ddesc.p, nullptr,
                                      dn.p),
          "extract_features_grid");
    dimg.download(img.data, (size_t)h * img.step);   // cv::rectangle drew into frame.image (src/Frame.cpp:32)
    int32_t n = 0;
    dn.download(&n, 1);
    frame.descriptors.create(n, 32, CV_8UC1);
    if (n) ddesc.download(frame.descriptors.data, (size_t)n * 32);
    const size_t old = frame.points.size();
    frame.points.resize(old + n);   // push_back loop, :47-49; no k-d tree, no map_point_ids (as the reference)
    if (n) dxy.download(reinterpret_cast<float *>(frame.points.data() + old), 2 * (size_t)n);
}

void extract_features(Frame &frame) {
    const cv::Mat &img = frame.image;
    if (img.empty() || img.type() != CV_8UC3) throw std::invalid_argument("extract_features: expects a CV_8UC3 BGR image");
    auto &st = vslam::settings();
    const int w = img.cols, h = img.rows, K = st.max_corners;
    const std::vector<s8> &pat = pattern();
    DBuf<uint8_t> dimg((size_t)h * img.step), ddesc((size_t)K * 32);
    DBuf<int8_t> dpat(1024);
    DBuf<float> dxy(2 * (size_t)K);
    DBuf<int32_t> dnodes(K), dn(1), dnd(1);
    dimg.upload(img.data, (size_t)h * img.step);
    dpat.upload(reinterpret_cast<const int8_t *>(pat.data()), 1024);
    vslam_extract_params p;
    p.max_corners = K;
    p.quality = st.quality;
    p.min_distance = st.min_distance;
    const float a = st.keypoint_angle_deg * (float)(3.14159265358979323846 / 180.f);   // angle *= CV_PI/180
    p.cos_a = (float)std::cos((double)a);
    p.sin_a = (float)std::sin((double)a);
    p.d_pattern = dpat.p;
    check(vslam_extract_features(ctx(), dimg.p, 1, w, h, (int)img.step, &p, K, dxy.p, ddesc.p, dnodes.p, dn.p, dnd.p),
          "extract_features");
    int32_t n = 0, nd = 0;
    dn.download(&n, 1);
    dnd.download(&nd, 1);
    const size_t old = frame.points.size();
    frame.points.resize(old + n);   // push_back loop, src/Frame.cpp:69-72
    if (n) dxy.download(reinterpret_cast<float *>(frame.points.data() + old), 2 * (size_t)n);
    frame.descriptors.create(n, 32, CV_8UC1);
    if (n) ddesc.download(frame.descriptors.data, (size_t)n * 32);
    frame.map_point_ids.resize(nd, -1);   // sized from the PRE-filter count, :73
    // k-d tree (:76): the device already built it over the kept points
    if (old == 0 && n > 0) {
        std::vector<int32_t> pre(n);
        dnodes.download(pre.data(), n);
        auto *nodes = static_cast<frame_kdtree::KDTreeNode *>(std::malloc((size_t)n * sizeof(frame_kdtree::KDTreeNode)));
        for (int i = 0; i < n; i++) nodes[i].pt_index = (usize)pre[i];
        link_preorder(nodes, 0, n);
        frame.kdtree.root = nodes;
        frame.kdtree.size += (u32)n;
        frame.kdtree.height = tree_height(n);
    } else {
        construct_kdtree(frame.kdtree, frame.points);
    }
}

void match_features(const Frame &frame1, const Frame &frame2, RansacFilter &rf,
                    std::vector<std::pair<int, int>> &matches, cv::Mat &F) {
    const int n1 = (int)frame1.points.size(), n2 = (int)frame2.points.size();
    if (frame1.descriptors.rows != n1 || frame2.descriptors.rows != n2)
        throw std::invalid_argument("match_features: descriptors and points disagree");
    const int K = std::max(std::max(n1, n2), 1);
    DBuf<uint8_t> dd1((size_t)K * 32), dd2((size_t)K * 32);
    DBuf<float> dxy1(2 * (size_t)K), dxy2(2 * (size_t)K), dF(9);
    DBuf<int32_t> dn1(1), dn2(1), dpairs(2 * (size_t)K), dm(1);
    if (n1) {
        dd1.upload(frame1.descriptors.data, (size_t)n1 * 32);
        dxy1.upload(reinterpret_cast<const float *>(frame1.points.data()), 2 * (size_t)n1);
    }
    if (n2) {
        dd2.upload(frame2.descriptors.data, (size_t)n2 * 32);
        dxy2.upload(reinterpret_cast<const float *>(frame2.points.data()), 2 * (size_t)n2);
    }
    const int32_t a = n1, b = n2;
    dn1.upload(&a, 1);
    dn2.upload(&b, 1);
    // knnMatch + ratio (src/Frame.cpp:83-94), then rf.find_fundamental (:97) with rf's own sets/seed
    check(vslam_match_knn2_ratio(ctx(), dd1.p, dn1.p, dd2.p, dn2.p, 1, K, dpairs.p, dm.p, nullptr), "match_knn2_ratio");
    int32_t m = 0;
    dm.download(&m, 1);
    std::vector<int32_t> flat(2 * (size_t)m);
    if (m) dpairs.download(flat.data(), flat.size());
    std::vector<std::pair<int, int>> i_matches(m);
    for (int i = 0; i < m; i++) i_matches[i] = {flat[2 * i], flat[2 * i + 1]};
    std::vector<bool> inliers;
    rf.find_fundamental(frame1.points, frame2.points, i_matches, inliers, F);
    for (size_t i = 0; i < inliers.size(); i++)
        if (inliers[i]) matches.emplace_back(i_matches[i]);   // appended, not cleared (:98-102)
}

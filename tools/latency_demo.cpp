// One frame at a time through the reference's own API (include/vslam/Frame.h, KDTree.h, RansacFilter.h), at the
// reference's parameters: 3000 corners (src/Frame.cpp:61), RansacFilter rf(8, 100, 10) (src/vslam.cpp:19), radius 2
// queries (src/vslam.cpp:149).  Prints one JSON object with per-call wall times in microseconds.
//
// usage: latency_demo <in.bin> [reps]     in: int32 w, h ; 2 BGR frames ; 1024 int8 pattern
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vslam/Frame.h"

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    FILE *fi = fopen(argv[1], "rb");
    int hdr[2];
    if (!fi || fread(hdr, 4, 2, fi) != 2) return 3;
    const int w = hdr[0], h = hdr[1];
    std::vector<unsigned char> img[2];
    for (auto &b : img) {
        b.resize((size_t)w * h * 3);
        if (fread(b.data(), 1, b.size(), fi) != b.size()) return 3;
    }
    vslam::settings().brief_pattern.resize(1024);
    if (fread(vslam::settings().brief_pattern.data(), 1, 1024, fi) != 1024) return 3;
    fclose(fi);
    vslam::settings().max_corners = 3000;

    auto extract = [&](int i, Frame &fr) {
        fr.kdtree.root = nullptr;
        cv::Mat image(h, w, CV_8UC3, img[i].data());
        initialize_frame(fr, image, i);
        extract_features(fr);
    };
    Frame a, b;
    extract(0, a);   // warm-up: context creation, scratch growth, kernel load
    extract(1, b);
    RansacFilter rf(8, 100, 10);
    rf.set_seed(1);
    {
        std::vector<std::pair<int, int>> m;
        cv::Mat F;
        match_features(a, b, rf, m, F);
    }
    std::vector<double> t_ext, t_match;
    size_t n_matches = 0;
    for (int r = 0; r < reps; r++) {
        Frame f;
        const auto t0 = clk::now();
        extract(r & 1, f);
        const auto t1 = clk::now();
        t_ext.push_back(us(t0, t1));
        vslam::forget_kdtree(f.kdtree.root);
        free(f.kdtree.root);
        std::vector<std::pair<int, int>> m;
        cv::Mat F;
        const auto t2 = clk::now();
        match_features(a, b, rf, m, F);
        const auto t3 = clk::now();
        t_match.push_back(us(t2, t3));
        n_matches = m.size();
    }
    // map association style queries: 1000 single calls vs one batched call
    std::vector<cv::Point2f> qs;
    for (size_t k = 0; k < 1000; k++) {
        const cv::Point2f &p = b.points[k % b.points.size()];
        qs.push_back(cv::Point2f(p.x + 0.75f, p.y - 1.25f));
    }
    radius_search(b.kdtree, b.points, qs[0], 2);   // first use uploads the tree
    const auto s0 = clk::now();
    size_t hits_single = 0;
    for (auto &q : qs) hits_single += radius_search(b.kdtree, b.points, q, 2).size();
    const auto s1 = clk::now();
    auto batch = vslam::radius_search_batch(b.kdtree, b.points, qs, 2);
    const auto s2 = clk::now();
    size_t hits_batch = 0;
    for (auto &v : batch) hits_batch += v.size();
    std::vector<double> t_build;
    for (int r = 0; r < reps; r++) {
        frame_kdtree t;
        t.root = nullptr;
        const auto c0 = clk::now();
        construct_kdtree(t, b.points);
        const auto c1 = clk::now();
        t_build.push_back(us(c0, c1));
        vslam::forget_kdtree(t.root);
        free(t.root);
    }
    printf("{\"width\": %d, \"height\": %d, \"keypoints\": [%zu, %zu], \"inlier_matches\": %zu, \"reps\": %d, "
           "\"extract_features_us\": %.1f, \"match_features_us\": %.1f, \"construct_kdtree_us\": %.1f, "
           "\"radius_search_1000_single_calls_us\": %.1f, \"radius_search_batch_1000_us\": %.1f, \"hits\": [%zu, %zu]}\n",
           w, h, a.points.size(), b.points.size(), n_matches, reps, median(t_ext), median(t_match), median(t_build), us(s0, s1), us(s1, s2),
           hits_single, hits_batch);
    free(a.kdtree.root);
    free(b.kdtree.root);
    return hits_single == hits_batch ? 0 : 1;
}

// Uses the drop-in C++ surfaces (include/vslam/Frame.h, KDTree.h, RansacFilter.h) the way the
// reference's consumers do (src/vslam.cpp:56-77,149; tests/test_kdtree.cpp), on frames read from a
// raw file, and dumps everything it computed so the Python test can compare with the oracle.
//
// usage: adapter_demo <in.bin> <out.bin>
//   in : int32 w, h, max_corners, hyp, seed ; then 2 BGR frames (h*w*3 bytes each) ; 1024 int8 pattern
//   out: flat int32/float32 records, see the writes below
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vslam/Frame.h"

static void wr(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }
static void wr_i(FILE *f, int v) { wr(f, &v, 4); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    int hdr[5];
    if (!fi || fread(hdr, 4, 5, fi) != 5) return 3;
    const int w = hdr[0], h = hdr[1], maxc = hdr[2], hyp = hdr[3];
    const unsigned seed = (unsigned)hdr[4];
    std::vector<unsigned char> img[2];
    for (auto &b : img) {
        b.resize((size_t)w * h * 3);
        if (fread(b.data(), 1, b.size(), fi) != b.size()) return 3;
    }
    vslam::settings().brief_pattern.resize(1024);
    if (fread(vslam::settings().brief_pattern.data(), 1, 1024, fi) != 1024) return 3;
    fclose(fi);
    vslam::settings().max_corners = maxc;

    std::vector<Frame> frames;                       // pm.frames, src/vslam.cpp:56
    for (int i = 0; i < 2; i++) {
        frames.emplace_back();
        Frame &fr = frames.back();
        fr.kdtree.root = nullptr;
        cv::Mat image(h, w, CV_8UC3, img[i].data());
        initialize_frame(fr, image, i);              // :60
        extract_features(fr);                        // :64
    }
    RansacFilter rf(8, hyp, 10);                     // :19
    rf.set_seed(seed);
    std::vector<std::pair<int, int>> matches;
    cv::Mat fundamental;
    match_features(frames[0], frames[1], rf, matches, fundamental);   // :77

    FILE *fo = fopen(argv[2], "wb");
    for (int i = 0; i < 2; i++) {
        const Frame &fr = frames[i];
        wr_i(fo, (int)fr.points.size());
        wr_i(fo, (int)fr.map_point_ids.size());
        wr(fo, fr.points.data(), fr.points.size() * 8);
        wr(fo, fr.descriptors.data, (size_t)fr.descriptors.rows * 32);
        for (size_t k = 0; k < fr.points.size(); k++) wr_i(fo, (int)fr.kdtree.root[k].pt_index);   // array order = pre-order
        wr_i(fo, (int)fr.kdtree.height);
    }
    wr_i(fo, (int)matches.size());
    for (auto &m : matches) { wr_i(fo, m.first); wr_i(fo, m.second); }
    wr_i(fo, fundamental.empty() ? 0 : 1);
    if (!fundamental.empty()) wr(fo, fundamental.ptr<float>(), 36);

    // map-association style queries (src/vslam.cpp:149): radius 2 around jittered keypoints of frame 1
    const Frame &f1 = frames[1];
    std::vector<cv::Point2f> qs;
    for (size_t k = 0; k < f1.points.size() && k < 64; k++) qs.push_back(cv::Point2f(f1.points[k].x + 0.75f, f1.points[k].y - 1.25f));
    wr_i(fo, (int)qs.size());
    for (auto &q : qs) {
        std::vector<usize> idx = radius_search(f1.kdtree, f1.points, q, 2);   // single-query signature
        wr_i(fo, (int)idx.size());
        for (usize v : idx) wr_i(fo, (int)v);
    }
    auto batch = vslam::radius_search_batch(f1.kdtree, f1.points, qs, 2);
    int same = 1;
    for (size_t k = 0; k < qs.size(); k++) same &= (batch[k] == radius_search(f1.kdtree, f1.points, qs[k], 2));
    wr_i(fo, same);

    // the point-storing KDTree the reference's own test drives (tests/test_kdtree.cpp)
    KDTree kd;
    kd.root = nullptr;
    construct_kdtree(kd, f1.points);
    cv::Point2f nn = nearest(kd, cv::Point2f(100.5f, 80.25f));
    wr(fo, &nn, 8);
    std::vector<cv::Point2f> near = radius_search(kd, cv::Point2f(100.5f, 80.25f), 12.f);
    wr_i(fo, (int)near.size());
    wr(fo, near.data(), near.size() * 8);

    // RansacFilter's other public methods
    std::vector<cv::Point2f> s1, s2;
    for (int j = 0; j < 8; j++) { s1.push_back(frames[0].points[matches[j].first]); s2.push_back(frames[1].points[matches[j].second]); }
    cv::Mat F8;
    rf.compute_fundamental(s1, s2, F8);
    wr(fo, F8.ptr<float>(), 36);
    std::vector<bool> inl;
    auto r = rf.compute_fundamental_residual(frames[0].points, frames[1].points, matches, F8, inl);
    wr_i(fo, r.first);
    wr(fo, &r.second, 4);
    for (size_t k = 0; k < inl.size(); k++) wr_i(fo, inl[k] ? 1 : 0);
    // the grid ORB/FAST extractor (src/Frame.cpp:16-51; call site commented out at src/vslam.cpp:63)
    {
        std::vector<unsigned char> copy = img[0];
        Frame g;
        g.kdtree.root = nullptr;
        cv::Mat image(h, w, CV_8UC3, copy.data());
        initialize_frame(g, image, 7);
        extract_features(g, 3, 4);
        wr_i(fo, (int)g.points.size());
        wr(fo, g.points.data(), g.points.size() * 8);
        wr(fo, g.descriptors.data, (size_t)g.descriptors.rows * 32);
        wr(fo, copy.data(), copy.size());            // frame.image aliases the buffer: outlines must be in it
    }
    fclose(fo);
    for (auto &fr : frames) free(fr.kdtree.root);    // src/vslam.cpp:295-297
    free(kd.root);
    return 0;
}

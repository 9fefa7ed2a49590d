// Uses the drop-in C++ surfaces (include/vslam/Frame.h, KDTree.h, RansacFilter.h) the way the
// reference's consumers do (src/vslam.cpp:56-77,149; tests/test_kdtree.cpp), on frames read from a
// raw file, and dumps everything it computed so the Python test can compare with the oracle.
//
// usage: adapter_demo <in.bin> <out.bin>
//   in : int32 w, h, max_corners, hyp, seed ; then 2 BGR frames (h*w*3 bytes each) ; optionally a 1024 int8 pattern
//   out: flat int32/float32 records, see the writes below
#include <cmath>
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
    // an optional 1024-byte table after the frames; without one the adapters' default is used: ORB's learned table
    {
        std::vector<s8> table(1024);
        if (fread(table.data(), 1, 1024, fi) == 1024) vslam::settings().brief_pattern = table;
    }
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
    {   // draw (src/Frame.cpp:8-13, src/vslam.cpp:91): a copy of the image with a green radius-2 ring on every keypoint
        cv::Mat annotated;
        draw(frames[0], annotated);
        if (annotated.rows != h || annotated.cols != w || annotated.data == frames[0].image.data) return 21;
        size_t changed = 0, green = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const unsigned char *a = annotated.ptr<unsigned char>(y) + 3 * x, *b = frames[0].image.ptr<unsigned char>(y) + 3 * x;
                if (a[0] == 0 && a[1] == 255 && a[2] == 0) green++;
                if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) changed++;
            }
        if (frames[0].points.empty() || green < 4 || green > 8 * frames[0].points.size() || changed > green) return 22;
        const cv::Point c(frames[0].points[0]);   // corners lie at least a few pixels inside the image
        const unsigned char *r = annotated.ptr<unsigned char>(c.y) + 3 * (c.x + 2), *m = annotated.ptr<unsigned char>(c.y) + 3 * c.x;
        const unsigned char *m0 = frames[0].image.ptr<unsigned char>(c.y) + 3 * c.x;
        if (!(r[0] == 0 && r[1] == 255 && r[2] == 0) || m[0] != m0[0] || m[1] != m0[1] || m[2] != m0[2]) return 23;
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
    // single queries are answered from the device-built cell table: many of them, several radii (on and off the grid,
    // exactly r away from a keypoint: the test is strict), against the host walk of the same nodes (src/KDTree.cpp:151-171
    // through the public node-pointer overload) and against the batched device query
    {
        int ok = 1, nonempty = 0;
        unsigned lcg = 12345u;
        auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) * (1.0f / 16777216.0f); };
        const float radii[6] = {2.f, 0.f, 0.5f, 1.5f, 3.75f, 8.f};
        for (int t = 0; t < 3000; t++) {
            const float r = radii[t % 6];
            const cv::Point2f &kp = f1.points[(size_t)(rnd() * f1.points.size()) % f1.points.size()];
            cv::Point2f q;
            switch (t % 5) {
                case 0: q = cv::Point2f(kp.x + r, kp.y); break;                                   // exactly r away: not a hit
                case 1: q = cv::Point2f(kp.x + (rnd() - 0.5f) * 6.f, kp.y + (rnd() - 0.5f) * 6.f); break;
                case 2: q = cv::Point2f(std::floor(kp.x + rnd() * 4.f - 2.f), std::floor(kp.y + rnd() * 4.f - 2.f)); break;
                case 3: q = cv::Point2f(rnd() * 330.f - 5.f, rnd() * 250.f - 5.f); break;         // anywhere, also outside the frame
                default: q = kp; break;
            }
            std::vector<usize> want;
            radius_search(f1.kdtree.root, f1.points, q, want, r, r * r, 0);
            const std::vector<usize> got = radius_search(f1.kdtree, f1.points, q, r);
            ok &= got == want ? 1 : 0;
            nonempty += want.empty() ? 0 : 1;
            if (t % 97 == 0) ok &= vslam::radius_search_batch(f1.kdtree, f1.points, std::vector<cv::Point2f>{q}, r)[0] == want ? 1 : 0;
        }
        wr_i(fo, ok);
        wr_i(fo, nonempty);
    }

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
    // --- compute_fundamental_residual on fewer than 8 matches (the reference's method has no lower limit)
    {
        std::vector<std::pair<int, int>> few(matches.begin(), matches.begin() + 5);
        std::vector<bool> in5;
        auto r5 = rf.compute_fundamental_residual(frames[0].points, frames[1].points, few, F8, in5);
        wr_i(fo, r5.first);
        wr(fo, &r5.second, 4);
        for (size_t k = 0; k < in5.size(); k++) wr_i(fo, in5[k] ? 1 : 0);
        std::vector<std::pair<int, int>> none;
        auto r0 = rf.compute_fundamental_residual(frames[0].points, frames[1].points, none, F8, in5);
        wr_i(fo, r0.first == 0 && r0.second == 0.f && in5.empty() ? 1 : 0);
    }
    // --- the device copy of a tree is validated, not trusted: (1) points edited in place, (2) a tree freed and another
    // one allocated at the same address with the same point count
    {
        const size_t half = f1.points.size() / 2;   // two disjoint point sets of the same size
        std::vector<cv::Point2f> pts(f1.points.begin(), f1.points.begin() + half);
        frame_kdtree t1;
        t1.root = nullptr;
        construct_kdtree(t1, pts);
        const cv::Point2f q(pts[7].x + 0.5f, pts[7].y);
        std::vector<usize> before = radius_search(t1, pts, q, 2);
        int ok = 1;
        bool had7 = false;
        for (usize v : before) had7 |= v == 7;
        ok &= had7 ? 1 : 0;
        // (0) the same kind of edit IN PLACE, on the vector the single-query table belongs to: the probe finds point 7 filed
        // under a cell it no longer lies in and the query is answered by the device walk over the current coordinates;
        // moved back, the answers are the original ones again
        {
            const cv::Point2f keep = pts[7];
            pts[7] = cv::Point2f(keep.x + 300.f, keep.y + 200.f);
            std::vector<usize> walked;
            radius_search(t1.root, pts, q, walked, 2.f, 4.f, 0);
            ok &= radius_search(t1, pts, q, 2) == walked ? 1 : 0;
            const cv::Point2f q_new(pts[7].x + 0.5f, pts[7].y);        // where the point went: a cell nothing was filed under
            walked.clear();
            radius_search(t1.root, pts, q_new, walked, 2.f, 4.f, 0);
            for (int rep = 0; rep < 300; rep++) (void)radius_search(t1, pts, q_new, 2);   // past the periodic full validation
            ok &= radius_search(t1, pts, q_new, 2) == walked ? 1 : 0;
            pts[7] = keep;
            ok &= radius_search(t1, pts, q, 2) == before ? 1 : 0;
        }
        // (1) move point 7 far away without rebuilding: the reference's search would now test the moved coordinates
        std::vector<cv::Point2f> moved = pts;
        moved[7] = cv::Point2f(pts[7].x + 500.f, pts[7].y + 500.f);
        std::vector<usize> after;
        radius_search(t1.root, moved, q, after, 2.f, 4.f, 0);                  // host walk of the same nodes = ground truth
        ok &= radius_search(t1, moved, q, 2) == after ? 1 : 0;
        // (2) free, then build trees over different points until one lands on the old address
        void *old_root = t1.root;
        free(t1.root);
        std::vector<cv::Point2f> other(f1.points.begin() + half, f1.points.begin() + 2 * half);
        int reused = 0;
        for (int attempt = 0; attempt < 8 && !reused; attempt++) {
            frame_kdtree t2;
            t2.root = nullptr;
            // hand-built (not through construct_kdtree, which would refresh the cache entry): the public recursive overload
            t2.root = static_cast<frame_kdtree::KDTreeNode *>(malloc(other.size() * sizeof(frame_kdtree::KDTreeNode)));
            std::vector<usize> idx(other.size());
            for (usize k = 0; k < idx.size(); k++) idx[k] = k;
            t2.size = 0;
            construct_kdtree(t2, other, idx, idx.begin(), idx.end(), 0);
            if (t2.root == old_root) {
                reused = 1;
                const cv::Point2f q2(other[3].x, other[3].y + 0.25f);
                std::vector<usize> want;
                radius_search(t2.root, other, q2, want, 2.f, 4.f, 0);
                ok &= radius_search(t2, other, q2, 2) == want ? 1 : 0;
                bool has3 = false;
                for (usize v : want) has3 |= v == 3;
                ok &= has3 ? 1 : 0;
            }
            free(t2.root);
        }
        wr_i(fo, ok);
        wr_i(fo, reused);
    }
    // --- the public node-pointer overloads agree with the device entry points, and the macros exist
    {
        int ok = 1;
        const cv::Point2f q(100.5f, 80.25f);
        cv::Point2f bp;
        float bd = INFINITY;
        nearest(kd.root, q, 0, &bp, &bd);
        ok &= bp == nn ? 1 : 0;
        std::vector<cv::Point2f> host_hits;
        radius_search(kd.root, q, host_hits, 12.f, SQ(12.f), 0);
        ok &= host_hits == near ? 1 : 0;
        KDTree kd2;
        kd2.root = static_cast<KDTree::KDTreeNode *>(malloc(f1.points.size() * sizeof(KDTree::KDTreeNode)));
        kd2.size = 0;
        std::vector<cv::Point2f> copy = f1.points;
        construct_kdtree(kd2, copy, copy.begin(), copy.end(), 0);
        for (size_t k = 0; k < f1.points.size(); k++) ok &= kd2.root[k].pt == kd.root[k].pt ? 1 : 0;   // same pre-order array as the device build
        free(kd2.root);
        float neg = -2.5f;
        cv::Point2f pp(3.f, 4.f);
        ok &= (ABS(neg) == 2.5f && P(pp, 1) == 4.f) ? 1 : 0;
        wr_i(fo, ok);
    }
    // --- RansacFilter(min_items < 8): 5 indices drawn into 8-wide sets (src/RansacFilter.cpp:17,22), then the hypothesis loop
    {
        RansacFilter rf5(5, 64, 10);
        rf5.set_seed(0xABCD);
        std::vector<bool> in5;
        cv::Mat F5;
        rf5.find_fundamental(frames[0].points, frames[1].points, matches, in5, F5);
        wr(fo, F5.ptr<float>(), 36);
        int cnt = 0;
        for (size_t k = 0; k < in5.size(); k++) cnt += in5[k] ? 1 : 0;
        wr_i(fo, cnt);
        int threw = 0;
        try {
            RansacFilter rf9(9, 64, 10);
            rf9.find_fundamental(frames[0].points, frames[1].points, matches, in5, F5);
        } catch (const std::invalid_argument &) { threw = 1; }
        wr_i(fo, threw);
    }
    fclose(fo);
    for (auto &fr : frames) free(fr.kdtree.root);    // src/vslam.cpp:295-297
    free(kd.root);
    return 0;
}

// Uses the drop-in include/vslam/helpers.h the way the reference's capture loop does (src/vslam.cpp:77-88,120-125,129-161,
// 186): match_features -> extract_Rt -> R_t / camera matrices -> triangulate, then the map-association block through its batch
// form, on frames read from a raw file; dumps what it computed so the Python test can hold it to the oracle.
//
// usage: helpers_demo <in.bin> <out.bin>     (in.bin as for adapter_demo: int32 w, h, max_corners, hyp, seed; 2 BGR frames)
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <vector>

#include "vslam/Frame.h"
#include "vslam/helpers.h"

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
    fclose(fi);
    vslam::settings().max_corners = maxc;
    std::vector<Frame> frames;
    for (int i = 0; i < 2; i++) {
        frames.emplace_back();
        Frame &fr = frames.back();
        fr.kdtree.root = nullptr;
        cv::Mat image(h, w, CV_8UC3, img[i].data());
        initialize_frame(fr, image, i);
        extract_features(fr);
    }
    RansacFilter rf(8, hyp, 10);
    rf.set_seed(seed);
    std::vector<std::pair<int, int>> matches;
    cv::Mat fundamental;
    match_features(frames[0], frames[1], rf, matches, fundamental);   // src/vslam.cpp:77
    if (fundamental.empty() || matches.size() < 8) return 4;

    // src/vslam.cpp:32: K
    const float F_ = 525.f;
    cv::Mat K(3, 3, CV_32FC1);
    const float kv[9] = {F_, 0, (float)(w / 2), 0, F_, (float)(h / 2), 0, 0, 1};
    for (int i = 0; i < 9; i++) K.ptr<float>(i / 3)[i % 3] = kv[i];
    cv::Mat rotation, translation;
    extract_Rt(fundamental, K, rotation, translation);                // :82
    if (rotation.rows != 3 || rotation.cols != 3 || translation.rows != 3 || translation.cols != 1) return 5;

    // :83-85 R_t = [R | t], :123-125 c1 = [K | 0], c2 = K * R_t.rowRange(0, 3) (float products summed left to right, as cv::Mat's
    // small-matrix multiply does)
    float Rt[12], c1v[12], c2v[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rt[r * 4 + c] = rotation.ptr<float>(r)[c];
        Rt[r * 4 + 3] = translation.ptr<float>(r)[0];
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            c1v[r * 4 + c] = c < 3 ? kv[r * 3 + c] : 0.f;
            float t = kv[r * 3 + 0] * Rt[0 * 4 + c];
            t = t + kv[r * 3 + 1] * Rt[1 * 4 + c];
            t = t + kv[r * 3 + 2] * Rt[2 * 4 + c];
            c2v[r * 4 + c] = t;
        }
    cv::Mat c1(3, 4, CV_32FC1, c1v), c2(3, 4, CV_32FC1, c2v);

    // :101-118 the matched points, interleaved
    const int n = (int)matches.size();
    cv::Mat p1(n, 2, CV_32FC1), p2(n, 2, CV_32FC1);
    for (int i = 0; i < n; i++) {
        p1.ptr<float>(i)[0] = frames[0].points[matches[i].first].x;
        p1.ptr<float>(i)[1] = frames[0].points[matches[i].first].y;
        p2.ptr<float>(i)[0] = frames[1].points[matches[i].second].x;
        p2.ptr<float>(i)[1] = frames[1].points[matches[i].second].y;
    }
    cv::Mat points_4d;
    triangulate(p1, p2, c1, c2, points_4d);                           // :186
    if (points_4d.rows != n || points_4d.cols != 4) return 6;

    // :129-161 the association block through its batch form: the triangulated points as the map; map point i was seen in
    // frame 0 at keypoint matches[i].first (every third one twice: the second observation is the matched keypoint of frame 1)
    Frame &cur = frames[1];
    cur.map_point_ids.assign(cur.points.size(), -1);
    for (size_t k = 0; k < cur.map_point_ids.size(); k += 11) cur.map_point_ids[k] = 5;   // already propagated by matching (:111-115)
    std::vector<u32> offs((size_t)n + 1, 0);
    std::vector<unsigned char> obs;
    for (int i = 0; i < n; i++) {
        const unsigned char *d0 = frames[0].descriptors.ptr<unsigned char>(matches[i].first);
        obs.insert(obs.end(), d0, d0 + 32);
        if (i % 3 == 0) {
            const unsigned char *d1 = frames[1].descriptors.ptr<unsigned char>(matches[i].second);
            obs.insert(obs.end(), d1, d1 + 32);
        }
        offs[(size_t)i + 1] = (u32)(obs.size() / 32);
    }
    cv::Mat obs_desc((int)(obs.size() / 32), 32, CV_8UC1, obs.data());
    const std::vector<s32> ids_before = cur.map_point_ids;
    const std::vector<s32> claim = vslam::associate_map_points(cur, points_4d, c2, w, h, offs, obs_desc, 6.f, 64);   // (a wider radius than the loop's 2: more of this synthetic pair's points find a keypoint)
    if ((int)claim.size() != n) return 7;

    // print_matrix goes to stdout in OpenCV's default format
    {
        std::ostringstream os;
        std::streambuf *old = std::cout.rdbuf(os.rdbuf());
        print_matrix(K, "K");
        std::cout.rdbuf(old);
        if (os.str().find("K\n[525, 0, ") != 0) return 8;
    }

    FILE *fo = fopen(argv[2], "wb");
    wr_i(fo, n);
    for (auto &m : matches) { wr_i(fo, m.first); wr_i(fo, m.second); }
    wr(fo, fundamental.ptr<float>(), 36);
    for (int r = 0; r < 3; r++) wr(fo, rotation.ptr<float>(r), 12);
    for (int r = 0; r < 3; r++) wr(fo, translation.ptr<float>(r), 4);
    wr(fo, c2v, 48);
    for (int i = 0; i < n; i++) wr(fo, points_4d.ptr<float>(i), 16);
    wr_i(fo, (int)cur.points.size());
    wr(fo, ids_before.data(), ids_before.size() * 4);
    wr(fo, cur.map_point_ids.data(), cur.map_point_ids.size() * 4);
    wr(fo, claim.data(), claim.size() * 4);
    wr_i(fo, (int)(obs.size() / 32));
    for (int i = 0; i <= n; i++) wr_i(fo, (int)offs[(size_t)i]);
    wr(fo, obs.data(), obs.size());
    fclose(fo);
    for (auto &fr : frames) free(fr.kdtree.root);
    return 0;
}

// OpenCV-side vector dumper: the only way the OpenCV-dependent rows of the hot path (SURVEY.md 8 a2, a3, a4, a8,
// a11-a13, and the 8f pose helpers) can ever be pinned.  It runs the reference's own OpenCV calls on the seeded inputs of tests/golden/make_golden.py
// and writes what they return, so that tests/test_opencv_pin.py can hold the oracle (and through it the kernels) to
// real OpenCV output.  NOT part of the product, never built by build(), never shipped to the GPU box; it builds only
// on a machine that has OpenCV 4 (the reference's dependency, makefile:4,7):
//
//     python tools/opencv_case.py export /tmp/case.bin                       # inputs, from tests/golden/frontend_v1.npz
//     g++ -O2 -std=c++17 tools/opencv_dump.cpp -o /tmp/opencv_dump $(pkg-config --cflags --libs opencv4)
//     /tmp/opencv_dump /tmp/case.bin /tmp/dump.bin
//     python tools/opencv_case.py import /tmp/dump.bin tests/golden/opencv_v1.npz   # commit the npz
//
// Each block below names the reference line whose OpenCV call it repeats.  Container format (both files): a sequence of
// records { u32 name_len, name, u32 dtype (0 u8, 1 i32, 2 f32, 3 f64), u32 ndim, u32 dims[ndim], data }.
#if !__has_include(<opencv2/core.hpp>)
#error "tools/opencv_dump.cpp needs OpenCV 4 headers (it is a maintainer-side tool, not part of the build)"
#else
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>
#include <stdexcept>
#include <string>
#include <vector>

struct Arr {
    uint32_t dtype = 0;
    std::vector<uint32_t> dims;
    std::vector<uint8_t> data;
    size_t count() const {
        size_t n = 1;
        for (uint32_t d : dims) n *= d;
        return n;
    }
    template <class T>
    const T *as() const { return reinterpret_cast<const T *>(data.data()); }
};
static const size_t kElem[4] = {1, 4, 4, 8};

static std::map<std::string, Arr> read_all(const char *path) {
    std::map<std::string, Arr> out;
    FILE *f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    uint32_t nl;
    while (fread(&nl, 4, 1, f) == 1) {
        std::string name(nl, ' ');
        Arr a;
        uint32_t nd;
        if (fread(&name[0], 1, nl, f) != nl || fread(&a.dtype, 4, 1, f) != 1 || fread(&nd, 4, 1, f) != 1) throw std::runtime_error("truncated case file");
        a.dims.resize(nd);
        if (nd && fread(a.dims.data(), 4, nd, f) != nd) throw std::runtime_error("truncated case file");
        a.data.resize(a.count() * kElem[a.dtype]);
        if (!a.data.empty() && fread(a.data.data(), 1, a.data.size(), f) != a.data.size()) throw std::runtime_error("truncated case file");
        out[name] = a;
    }
    fclose(f);
    return out;
}

static FILE *g_out = nullptr;
static void put(const std::string &name, uint32_t dtype, std::vector<uint32_t> dims, const void *data) {
    const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size();
    size_t n = kElem[dtype];
    for (uint32_t d : dims) n *= d;
    fwrite(&nl, 4, 1, g_out);
    fwrite(name.data(), 1, nl, g_out);
    fwrite(&dtype, 4, 1, g_out);
    fwrite(&nd, 4, 1, g_out);
    if (nd) fwrite(dims.data(), 4, nd, g_out);
    if (n) fwrite(data, 1, n, g_out);
}
static void put_mat(const std::string &name, const cv::Mat &m) {
    cv::Mat c = m.isContinuous() ? m : m.clone();
    const uint32_t dtype = c.depth() == CV_8U ? 0 : c.depth() == CV_32S ? 1 : c.depth() == CV_32F ? 2 : 3;
    if (c.depth() != CV_8U && c.depth() != CV_32S && c.depth() != CV_32F && c.depth() != CV_64F) throw std::runtime_error("put_mat: depth");
    put(name, dtype, {(uint32_t)c.rows, (uint32_t)(c.cols * c.channels())}, c.data);
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: opencv_dump <case.bin> <dump.bin>\n");
        return 2;
    }
    try {
        auto in = read_all(argv[1]);
        g_out = fopen(argv[2], "wb");
        if (!g_out) throw std::runtime_error("cannot write the dump");
        const std::string ver = CV_VERSION;
        put("opencv_version", 0, {(uint32_t)ver.size()}, ver.data());
#ifdef HAVE_LAPACK
        const uint8_t lapack = 1;   // cv::SVDecomp goes through sgesdd then: the built-in Jacobi is NOT what this build runs
#else
        const uint8_t lapack = 0;
#endif
        put("opencv_have_lapack_macro", 0, {1}, &lapack);

        // ---------------------------------------------------------------- extraction, src/Frame.cpp:53-80
        // `prefix` "" = the seeded synthetic pair of frontend_v1.npz; "p_" = the photographic frames of real_v1.npz
        // (round 5: plateaus, gradients and JPEG structure, which the synthetic textures lack)
        auto extraction = [&](const std::string &prefix, const Arr &bgr, int maxc, std::vector<cv::Mat> *descs) {
            const int frames = (int)bgr.dims[0], h = (int)bgr.dims[1], w = (int)bgr.dims[2];
            for (int f = 0; f < frames; f++) {
                const std::string t = std::to_string(f);
                cv::Mat image(h, w, CV_8UC3, const_cast<uint8_t *>(bgr.as<uint8_t>()) + (size_t)f * h * w * 3);
                cv::Mat gray;
                cv::cvtColor(image, gray, cv::COLOR_BGR2GRAY);                                  // :56
                put_mat("cv_" + prefix + "gray" + t, gray);
                cv::Mat eig;
                cv::cornerMinEigenVal(gray, eig, 3, 3);                                          // inside goodFeaturesToTrack
                put_mat("cv_" + prefix + "eig" + t, eig);
                std::vector<cv::Point2f> corners;
                cv::goodFeaturesToTrack(gray, corners, maxc, 0.01, 3);                           // :61 (3000 there)
                put("cv_" + prefix + "corners" + t, 2, {(uint32_t)corners.size(), 2}, corners.data());
                cv::Mat blurred;
                cv::GaussianBlur(gray, blurred, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);    // ORB::compute's blur
                put_mat("cv_" + prefix + "blur" + t, blurred);
                std::vector<cv::KeyPoint> kps;
                for (const cv::Point2f &p : corners) kps.push_back(cv::KeyPoint(p, 20));          // :64-67
                cv::Ptr<cv::ORB> orb = cv::ORB::create();                                         // :57
                cv::Mat desc;
                orb->compute(gray, kps, desc);                                                    // :68 (may drop border keypoints)
                std::vector<cv::Point2f> kept;
                for (const cv::KeyPoint &k : kps) kept.push_back(k.pt);                           // :69-72
                put("cv_" + prefix + "kept_xy" + t, 2, {(uint32_t)kept.size(), 2}, kept.data());
                put_mat("cv_" + prefix + "desc" + t, desc);                                       // OpenCV's own learned pattern
                if (descs) descs->push_back(desc.clone());
            }
        };
        extraction("", in.at("e_bgr"), in.at("e_maxc").as<int32_t>()[0], nullptr);
        if (in.count("p_bgr")) {   // [2 P][480][640][3]: frames [0, P) "last", [P, 2 P) "current"; knnMatch + ratio test per pair (:83-94)
            std::vector<cv::Mat> descs;
            extraction("p_", in.at("p_bgr"), in.at("p_maxc").as<int32_t>()[0], &descs);
            const int P = (int)descs.size() / 2;
            for (int i = 0; i < P; i++) {
                cv::Ptr<cv::BFMatcher> matcher = cv::BFMatcher::create(cv::NORM_HAMMING);
                std::vector<std::vector<cv::DMatch>> knn;
                matcher->knnMatch(descs[(size_t)i], descs[(size_t)(P + i)], knn, 2);
                std::vector<int32_t> pairs;
                for (auto &m : knn)
                    if (m.size() == 2 && m[0].distance < m[1].distance * 0.7) {
                        pairs.push_back(m[0].queryIdx);
                        pairs.push_back(m[0].trainIdx);
                    }
                put("cv_p_pairs" + std::to_string(i), 1, {(uint32_t)(pairs.size() / 2), 2}, pairs.data());
            }
        }

        // ---------------------------------------------------------------- matching, src/Frame.cpp:83-94
        {
            const Arr &d1 = in.at("m_d1"), &d2 = in.at("m_d2");
            cv::Mat m1((int)d1.dims[0], 32, CV_8UC1, const_cast<uint8_t *>(d1.as<uint8_t>()));
            cv::Mat m2((int)d2.dims[0], 32, CV_8UC1, const_cast<uint8_t *>(d2.as<uint8_t>()));
            cv::Ptr<cv::BFMatcher> matcher = cv::BFMatcher::create(cv::NORM_HAMMING);         // :83
            std::vector<std::vector<cv::DMatch>> knn;
            matcher->knnMatch(m1, m2, knn, 2);                                                // :85
            std::vector<int32_t> flat;   // idx0, dist0, idx1, dist1 per query
            std::vector<int32_t> pairs;
            for (auto &m : knn) {
                flat.push_back(m[0].trainIdx);
                flat.push_back((int32_t)m[0].distance);
                flat.push_back(m[1].trainIdx);
                flat.push_back((int32_t)m[1].distance);
                if (m[0].distance < m[1].distance * 0.7) {                                    // :91
                    pairs.push_back(m[0].queryIdx);
                    pairs.push_back(m[0].trainIdx);
                }
            }
            put("cv_knn", 1, {(uint32_t)knn.size(), 4}, flat.data());
            put("cv_pairs", 1, {(uint32_t)(pairs.size() / 2), 2}, pairs.data());
        }

        // ---------------------------------------------------------------- RANSAC arithmetic, src/RansacFilter.cpp:69-140
        {
            const Arr &p1 = in.at("r_p1"), &p2 = in.at("r_p2"), &pr = in.at("r_pairs"), &sets = in.at("r_fsets");
            const int M = (int)pr.dims[0], H = (int)sets.dims[0];
            const float thr = in.at("r_thr").as<float>()[0];
            const float *P1 = p1.as<float>(), *P2 = p2.as<float>();
            const int32_t *PR = pr.as<int32_t>(), *S = sets.as<int32_t>();
            std::vector<float> allF((size_t)H * 9), allD8((size_t)H * 8), allVt8((size_t)H * 81), allD3((size_t)H * 3), allSum(H);
            std::vector<int32_t> allCount(H);
            cv::Mat x1(3, M, CV_32FC1), x2(3, M, CV_32FC1);                                   // :108-117
            for (int i = 0; i < M; i++) {
                x1.at<float>(0, i) = P1[2 * PR[2 * i]];
                x1.at<float>(1, i) = P1[2 * PR[2 * i] + 1];
                x1.at<float>(2, i) = 1;
                x2.at<float>(0, i) = P2[2 * PR[2 * i + 1]];
                x2.at<float>(1, i) = P2[2 * PR[2 * i + 1] + 1];
                x2.at<float>(2, i) = 1;
            }
            for (int hh = 0; hh < H; hh++) {
                cv::Mat A(8, 9, CV_32FC1);                                                    // :75-90
                for (int i = 0; i < 8; i++) {
                    const int idx = S[hh * 8 + i];
                    const float u1 = P1[2 * PR[2 * idx]], v1 = P1[2 * PR[2 * idx] + 1];
                    const float u2 = P2[2 * PR[2 * idx + 1]], v2 = P2[2 * PR[2 * idx + 1] + 1];
                    float *r = A.ptr<float>(i);
                    r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2;
                    r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2;
                    r[6] = u1;      r[7] = v1;      r[8] = 1;
                }
                cv::Mat D, U, V_t;
                cv::SVDecomp(A, D, U, V_t, cv::SVD::MODIFY_A | cv::SVD::FULL_UV);              // :94
                std::memcpy(&allD8[(size_t)hh * 8], D.ptr<float>(), 32);
                cv::Mat Vc = V_t.isContinuous() ? V_t : V_t.clone();
                std::memcpy(&allVt8[(size_t)hh * 81], Vc.ptr<float>(), 81 * 4);
                cv::Mat temp_F = V_t.row(8).reshape(0, 3);                                    // :95
                cv::SVDecomp(temp_F, D, U, V_t, cv::SVD::MODIFY_A | cv::SVD::FULL_UV);         // :98
                std::memcpy(&allD3[(size_t)hh * 3], D.ptr<float>(), 12);
                D.at<float>(2) = 0;                                                           // :99
                temp_F = U * cv::Mat::diag(D) * V_t;                                          // :101
                cv::Mat Fc = temp_F.isContinuous() ? temp_F : temp_F.clone();
                std::memcpy(&allF[(size_t)hh * 9], Fc.ptr<float>(), 36);
                // compute_fundamental_residual, :119-138
                cv::Mat F_x1 = temp_F * x1;
                cv::Mat F_t_x2 = temp_F.t() * x2;
                cv::Mat x2_t_F_x1 = x2.mul(F_x1);
                cv::reduce(x2_t_F_x1, x2_t_F_x1, 0, cv::REDUCE_SUM);
                cv::Mat e_sq = x2_t_F_x1.mul(x2_t_F_x1) / F_x1.row(0).mul(F_x1.row(0)) + F_x1.row(1).mul(F_x1.row(1)) +
                               F_t_x2.row(0).mul(F_t_x2.row(0)) + F_t_x2.row(1).mul(F_t_x2.row(1));
                int n_in = 0;
                for (int i = 0; i < M; i++) n_in += e_sq.at<float>(i) <= thr ? 1 : 0;         // :130
                allCount[hh] = n_in;
                allSum[hh] = (float)cv::sum(e_sq)[0];                                         // :138
                if (hh == 0) put_mat("cv_esq_h0", e_sq);
            }
            put("cv_hypF", 2, {(uint32_t)H, 9}, allF.data());
            put("cv_svd8_D", 2, {(uint32_t)H, 8}, allD8.data());
            put("cv_svd8_Vt", 2, {(uint32_t)H, 81}, allVt8.data());
            put("cv_svd3_D", 2, {(uint32_t)H, 3}, allD3.data());
            put("cv_hyp_count", 1, {(uint32_t)H}, allCount.data());
            put("cv_hyp_sum", 2, {(uint32_t)H}, allSum.data());
        }
        // ---------------------------------------------------------------- grid ORB/FAST extractor, src/Frame.cpp:16-51
        if (in.count("g_bgr")) {
            const Arr &gb = in.at("g_bgr");   // [h][w][3] u8
            const int gh = (int)gb.dims[0], gw = (int)gb.dims[1];
            const int nrows = in.at("g_grid").as<int32_t>()[0], ncols = in.at("g_grid").as<int32_t>()[1];
            cv::Mat image = cv::Mat(gh, gw, CV_8UC3, const_cast<uint8_t *>(gb.as<uint8_t>())).clone();   // the extractor draws into it
            {   // the pieces the restatement is built from, on the untouched frame
                cv::Mat gray;
                cv::cvtColor(image, gray, cv::COLOR_BGR2GRAY);
                std::vector<cv::KeyPoint> fk;
                cv::FAST(gray, fk, 20, true);                                                 // FAST-9/16 + 3x3 non-max, as ORB's detector runs it
                std::vector<float> ff;
                for (auto &k : fk) { ff.push_back(k.pt.x); ff.push_back(k.pt.y); ff.push_back(k.response); }
                put("cv_g_fast20", 2, {(uint32_t)fk.size(), 3}, ff.data());
                cv::Mat small;
                cv::resize(gray, small, cv::Size(213, 160), 0, 0, cv::INTER_LINEAR_EXACT);     // ORB's pyramid resampling
                put_mat("cv_g_resized", small);
            }
            const int nfeatures = 500;                                                        // :19
            const int cw = image.cols / ncols, ch = image.rows / nrows;                        // :20
            cv::Ptr<cv::ORB> det = cv::ORB::create(nfeatures, 1.2f, 8, 31, 0, 2, cv::ORB::HARRIS_SCORE, 31, 20);   // :22
            cv::Ptr<cv::ORB> fallback = cv::ORB::create(nfeatures, 1.2f, 8, 31, 0, 2, cv::ORB::HARRIS_SCORE, 31, 5);   // :23
            std::vector<cv::KeyPoint> keypoints;
            for (int i = 0; i < ncols; i++)
                for (int j = 0; j < nrows; j++) {                                             // :27-41
                    std::vector<cv::KeyPoint> temp;
                    const int sx = i * cw, sy = j * ch;
                    const cv::Rect rect(sx, sy, cw, ch);
                    cv::rectangle(image, rect, cv::Scalar(0, 0, 0));                          // :32
                    det->detect(image(rect), temp);                                           // :33
                    if (temp.size() < (size_t)nfeatures) fallback->detect(image(rect), temp); // :34-36
                    for (auto &kp : temp) keypoints.emplace_back(sx + kp.pt.x, sy + kp.pt.y, kp.size, kp.angle, kp.response, kp.octave, kp.class_id);
                }
            cv::Mat gdesc;
            det->compute(image, keypoints, gdesc);                                            // :43
            std::vector<float> gxy, gao;
            for (auto &kp : keypoints) {
                gxy.push_back(kp.pt.x); gxy.push_back(kp.pt.y);
                gao.push_back(kp.angle); gao.push_back((float)kp.octave);
            }
            put_mat("cv_g_outlined", image);
            put("cv_g_xy", 2, {(uint32_t)keypoints.size(), 2}, gxy.data());
            put("cv_g_angle_octave", 2, {(uint32_t)keypoints.size(), 2}, gao.data());
            put_mat("cv_g_desc", gdesc);                                                      // OpenCV's own learned pattern
        }

        // ---------------------------------------------------------------- pose helpers, src/helpers.cpp:3-80
        if (in.count("t_F")) {
            cv::Mat Fm(3, 3, CV_32FC1, const_cast<float *>(in.at("t_F").as<float>()));
            cv::Mat K(3, 3, CV_32FC1, const_cast<float *>(in.at("t_K").as<float>()));
            // extract_Rt, :3-35
            cv::Mat E = K.t() * Fm * K;                                                        // :4
            cv::Mat U, D, V_t;
            cv::SVD::compute(E, D, U, V_t);                                                    // :7
            cv::Mat translation;
            U.col(2).copyTo(translation);                                                      // :9
            translation /= cv::norm(translation);                                              // :11
            cv::Mat W = cv::Mat::zeros(3, 3, CV_32FC1);
            W.at<float>(0, 1) = -1; W.at<float>(1, 0) = 1; W.at<float>(2, 2) = 1;             // :13-16
            cv::Mat R_1 = U * W * V_t;                                                         // :18
            if (cv::determinant(R_1) < 0) R_1 = -R_1;
            cv::Mat R_2 = U * W.t() * V_t;                                                     // :23
            if (cv::determinant(R_2) < 0) R_2 = -R_2;
            cv::Mat rotation = (R_1.at<float>(0, 0) + R_1.at<float>(1, 1) + R_1.at<float>(2, 2) < 0) ? R_2 : R_1;   // :29
            if (translation.at<float>(2) < 0) translation *= -1;                               // :31-33
            put_mat("cv_t_E", E);
            put_mat("cv_t_R", rotation);
            put_mat("cv_t_t", translation);
            // camera matrices as the driver forms them (src/vslam.cpp:83-88,125): c1 = K [I | 0], c2 = K [R | t]
            cv::Mat Rt1 = cv::Mat::eye(3, 4, CV_32FC1), Rt2(3, 4, CV_32FC1);
            rotation.copyTo(Rt2(cv::Rect(0, 0, 3, 3)));
            translation.copyTo(Rt2(cv::Rect(3, 0, 1, 3)));
            cv::Mat c1 = K * Rt1, c2 = K * Rt2;
            put_mat("cv_t_c2", c2);
            // triangulate, :37-80
            const Arr &a1 = in.at("t_p1"), &a2 = in.at("t_p2");
            const int N = (int)a1.dims[0];
            const float *p1d = a1.as<float>(), *p2d = a2.as<float>();
            cv::Mat points_4d(N, 4, CV_32F), A(4, 4, CV_32F);
            for (int i = 0; i < N; i++) {
                A.row(0) = p1d[2 * i] * c1.row(2) - c1.row(0);                                 // :50-53
                A.row(1) = p1d[2 * i + 1] * c1.row(2) - c1.row(1);
                A.row(2) = p2d[2 * i] * c2.row(2) - c2.row(0);
                A.row(3) = p2d[2 * i + 1] * c2.row(2) - c2.row(1);
                cv::Mat U4, D4, Vt4;
                cv::SVD::compute(A, D4, U4, Vt4, cv::SVD::MODIFY_A | cv::SVD::FULL_UV);        // :60
                const float *v = Vt4.ptr<float>(3);
                float *o = points_4d.ptr<float>(i);
                o[0] = v[0] / v[3]; o[1] = v[1] / v[3]; o[2] = v[2] / v[3]; o[3] = 1;          // :73-76
            }
            put_mat("cv_t_points4d", points_4d);
        }
        fclose(g_out);
        printf("wrote %s (OpenCV %s)\n", argv[2], CV_VERSION);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "opencv_dump: %s\n", e.what());
        return 1;
    }
}
#endif

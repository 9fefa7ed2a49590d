// The reference's only test, run against the PRODUCT: the procedure of
// /root/reference/tests/test_kdtree.cpp:47-151 (test_nearest_neighbor(1000, 2500, 3000) and
// test_radius_search(1000, 10, 100, 2500, 3000), :148-151) driven through the drop-in surfaces of
// include/vslam/KDTree.h -> libvslam_host.so -> C ABI -> HIP kernels.  Same random stream (glibc rand(),
// never seeded, two values burned per generated point, :42), same sizes, same acceptance rules
// (:72-79 nearest: same point or tied squared distance; :119-129 radius: equal sorted hit sets).
// The reference prints "1000 successes out of 1000 trials" twice; so must this.
//
// usage: kdtree_ref_procedure [n_trials]   -> prints "<nn_successes> <radius_successes> <n_trials>"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vslam/KDTree.h"

static void random_points(std::vector<cv::Point2f> &arr, int size) {
    arr.reserve(size);
    for (int j = 0; j < size; j++) {
        (void)rand();
        (void)rand();                                   // the unused local at :42
        const float second = (float)(rand() % 100);     // g++ evaluates emplace_back's arguments right to left
        const float first = (float)(rand() % 100);
        arr.emplace_back(first, second);
    }
}

static float int_dist_sq(const cv::Point2f &a, const cv::Point2f &b) {   // cv::Point diff = a - b; diff.dot(diff)
    const cv::Point d = a - b;
    return (float)d.dot(d);
}

int main(int argc, char **argv) {
    const int n_trials = argc > 1 ? atoi(argv[1]) : 1000;
    const int min_size = 2500, max_size = 3000;
    std::vector<cv::Point2f> arr;
    int nn_ok = 0, rad_ok = 0;

    for (int t = 0; t < n_trials; t++) {
        const int size = rand() % (max_size - min_size) + min_size;
        random_points(arr, size);
        KDTree kdtree;
        kdtree.root = nullptr;
        construct_kdtree(kdtree, arr);
        cv::Point2f qp;
        qp.x = (float)(rand() % 100);
        qp.y = (float)(rand() % 100);
        const cv::Point2f nn = nearest(kdtree, qp);
        float best = INFINITY;
        cv::Point2f actual;
        for (const cv::Point2f &pt : arr) {
            const float cur = int_dist_sq(qp, pt);
            if (cur < best) {
                best = cur;
                actual = pt;
            }
        }
        if (nn == actual || int_dist_sq(qp, nn) == best) nn_ok++;
        free(kdtree.root);
        arr.clear();
    }

    const float min_radius = 10, max_radius = 100;
    for (int t = 0; t < n_trials; t++) {
        const int size = rand() % (max_size - min_size) + min_size;
        const float radius = (float)rand() / ((float)RAND_MAX / (max_radius - min_radius)) + min_radius;
        const float radius_sq = SQ(radius);
        random_points(arr, size);
        KDTree kdtree;
        kdtree.root = nullptr;
        construct_kdtree(kdtree, arr);
        cv::Point2f qp;
        qp.x = (float)(rand() % 100);
        qp.y = (float)(rand() % 100);
        std::vector<cv::Point2f> found = radius_search(kdtree, qp, radius);
        std::vector<cv::Point2f> want;
        for (const cv::Point2f &pt : arr)
            if (int_dist_sq(qp, pt) < radius_sq) want.push_back(pt);
        bool fail = found.size() != want.size();
        if (!fail) {
            const auto lex = [](const cv::Point2f &a, const cv::Point2f &b) { return (a.x == b.x) ? a.y < b.y : a.x < b.x; };
            std::sort(found.begin(), found.end(), lex);
            std::sort(want.begin(), want.end(), lex);
            for (size_t i = 0; i < found.size(); i++) fail |= found[i] != want[i];
        }
        if (!fail) rad_ok++;
        free(kdtree.root);
        arr.clear();
    }
    printf("%d %d %d\n", nn_ok, rad_ok, n_trials);
    return 0;
}

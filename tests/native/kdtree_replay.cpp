// Replays the PROCEDURE of the reference's only test, /root/reference/tests/test_kdtree.cpp,
// against the oracle's KDTree restatement (oracle/vso_kdtree.cpp).  It is the one known-answer
// check the reference holds for the hot path: `1000 successes out of 1000 trials`, twice
// (test_kdtree.cpp:148-151).  Same random stream: glibc rand(), never seeded (:42-43,51,56),
// two extra rand() burned per generated point (:42), both tests in one process.
//
// usage: kdtree_replay [n_trials]   -> prints "<nn_successes> <radius_successes> <n_trials>"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../oracle/vso.h"

struct P { float x, y; };

static void gen_points(std::vector<P> &arr, int size) {
    arr.clear();
    for (int j = 0; j < size; j++) {
        (void)rand(); (void)rand();                 // the unused `pt` at test_kdtree.cpp:42
        // emplace_back(rand() % 100, rand() % 100): g++ evaluates call arguments right to left
        float second = (float)(rand() % 100);
        float first = (float)(rand() % 100);
        arr.push_back({first, second});
    }
}

int main(int argc, char **argv) {
    const int n_trials = argc > 1 ? atoi(argv[1]) : 1000;
    const int min_size = 2500, max_size = 3000;     // test_kdtree.cpp:149-150
    std::vector<P> arr;
    std::vector<float> tree;
    int nn_ok = 0, rad_ok = 0;

    for (int t = 0; t < n_trials; t++) {            // test_nearest_neighbor, :47-92
        int size = rand() % (max_size - min_size) + min_size;
        gen_points(arr, size);
        tree.resize(2 * (size_t)size);
        vso_kdtree_build_points(&arr[0].x, size, tree.data());
        P qp;
        qp.x = (float)(rand() % 100);
        qp.y = (float)(rand() % 100);
        float nn[2];
        vso_kdtree_nearest_points(tree.data(), size, qp.x, qp.y, INFINITY, nn);
        float best = INFINITY;
        P actual{0, 0};
        for (const P &pt : arr) {
            int dx = (int)lrintf(qp.x - pt.x), dy = (int)lrintf(qp.y - pt.y);   // cv::Point diff, :65
            float cur = (float)(dx * dx + dy * dy);
            if (cur < best) { best = cur; actual = pt; }
        }
        if (nn[0] == actual.x && nn[1] == actual.y) nn_ok++;
        else {
            int dx = (int)lrintf(qp.x - nn[0]), dy = (int)lrintf(qp.y - nn[1]);
            if ((float)(dx * dx + dy * dy) == best) nn_ok++;                     // :72-79
        }
    }

    const float min_radius = 10, max_radius = 100;
    for (int t = 0; t < n_trials; t++) {            // test_radius_search, :94-146
        int size = rand() % (max_size - min_size) + min_size;
        float radius = (float)rand() / ((float)RAND_MAX / (max_radius - min_radius)) + min_radius;
        float radius_sq = radius * radius;
        gen_points(arr, size);
        tree.resize(2 * (size_t)size);
        vso_kdtree_build_points(&arr[0].x, size, tree.data());
        P qp;
        qp.x = (float)(rand() % 100);
        qp.y = (float)(rand() % 100);
        std::vector<P> found((size_t)size);
        int nf = vso_kdtree_radius_points(tree.data(), size, qp.x, qp.y, radius, &found[0].x, size);
        found.resize(nf);
        std::vector<P> pts;
        for (const P &pt : arr) {
            int dx = (int)lrintf(qp.x - pt.x), dy = (int)lrintf(qp.y - pt.y);
            if ((float)(dx * dx + dy * dy) < radius_sq) pts.push_back(pt);
        }
        bool fail = found.size() != pts.size();
        if (!fail) {
            auto cmp = [](const P &a, const P &b) { return (a.x == b.x) ? a.y < b.y : a.x < b.x; };
            std::sort(found.begin(), found.end(), cmp);
            std::sort(pts.begin(), pts.end(), cmp);
            for (size_t i = 0; i < found.size(); i++)
                if (found[i].x != pts[i].x || found[i].y != pts[i].y) fail = true;
        }
        if (!fail) rad_ok++;
    }
    printf("%d %d %d\n", nn_ok, rad_ok, n_trials);
    return 0;
}

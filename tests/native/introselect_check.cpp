// Host check of vslam_amd/csrc/introselect.h against the platform's std::nth_element:
// identical final permutation (not just the nth value) on tie-heavy and adversarial inputs.
// usage: introselect_check -> prints "<ok> <total> <heap_select fallbacks taken>"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static int g_heap_calls = 0;
#define VS_SEL_ON_HEAP_SELECT() (++g_heap_calls)
#include "../../vslam_amd/csrc/introselect.h"

struct Item { float key; int id; };
struct Store {
    using value_type = Item;
    using key_type = float;
    std::vector<Item> *v;
    float key(int i) const { return (*v)[i].key; }
    float key_of(const Item &x) const { return x.key; }
    Item get(int i) const { return (*v)[i]; }
    void set(int i, const Item &x) { (*v)[i] = x; }
    void swap(int i, int j) { std::swap((*v)[i], (*v)[j]); }
    bool less(float a, float b) const { return a < b; }
};

static bool run_case(std::vector<Item> base, int first, int nth, int last) {
    std::vector<Item> a = base, b = base;
    std::nth_element(a.begin() + first, a.begin() + nth, a.begin() + last,
                     [](const Item &x, const Item &y) { return x.key < y.key; });
    Store s{&b};
    vs_sel::nth_element(s, first, nth, last);
    for (size_t i = 0; i < a.size(); i++)
        if (a[i].id != b[i].id) return false;
    return true;
}

// median-of-3 killer (Musser) so the depth limit trips and heap_select runs
static std::vector<Item> killer(int n) {
    std::vector<Item> v(n);
    int k = n / 2;
    for (int i = 1; i <= k; i++) {
        if (i % 2 == 1) { v[i - 1].key = (float)i; v[i].key = (float)(k + i); }
        v[k + i - 1].key = (float)(2 * i);
    }
    for (int i = 0; i < n; i++) v[i].id = i;
    return v;
}

int main() {
    std::mt19937 g(12345);
    int ok = 0, total = 0;
    const int sizes[] = {1, 2, 3, 4, 5, 7, 8, 16, 17, 33, 64, 100, 257, 1000, 2000, 4001};
    for (int n : sizes)
        for (int pattern = 0; pattern < 8; pattern++)
            for (int rep = 0; rep < 6; rep++) {
                std::vector<Item> v(n);
                for (int i = 0; i < n; i++) {
                    v[i].id = i;
                    switch (pattern) {
                        case 0: v[i].key = (float)(g() % 1280); break;          // pixel-like, many ties
                        case 1: v[i].key = (float)(g() % 4); break;             // extreme ties
                        case 2: v[i].key = (float)i; break;                     // sorted
                        case 3: v[i].key = (float)(n - i); break;               // reversed
                        case 4: v[i].key = 7.f; break;                          // all equal
                        case 5: v[i].key = (float)(i < n / 2 ? i : n - i); break;   // organ pipe
                        case 6: v[i].key = (float)(g() % 1000000) * 0.37f; break;   // nearly distinct
                        default: break;
                    }
                }
                if (pattern == 7) v = killer(n % 2 ? n + 1 : n);
                const int nn = (int)v.size();
                int nth = rep == 0 ? nn / 2 : (int)(g() % nn);
                total++;
                ok += run_case(v, 0, nth, nn);
                if (nn > 8) {   // interior sub-range, as the recursion produces
                    int f = (int)(g() % (nn / 2)), l = nn - (int)(g() % (nn / 4 + 1));
                    total++;
                    ok += run_case(v, f, f + (l - f) / 2, l);
                }
            }
    printf("%d %d %d\n", ok, total, g_heap_calls);
    return ok == total ? 0 : 1;
}

// Position-exact re-implementation of libstdc++'s std::nth_element (GCC 11 introselect) over an
// abstract random-access store, usable from host and device code.
//
// Why: the reference builds its k-d trees with std::nth_element (src/KDTree.cpp:10,12,128).
// Keypoint coordinates are integer-valued, so ties on the split axis are the norm, and which of
// the tied points lands left/right of the median — and in what order, which steers every deeper
// split — is decided by the exact sequence of swaps introselect performs.  To emit a tree that is
// bit-identical to the reference's, the device has to replay that sequence:
//   median-of-three to *first (first+1, mid, last-1), unguarded Hoare partition around *first,
//   keep the side holding nth, depth limit 2*floor(lg n) then heap-select, insertion sort for
//   ranges of <= 3 elements.
// tests/test_introselect.py compiles this header for the host and checks the final permutation
// against the platform's std::nth_element on tie-heavy and adversarial inputs.
//
// Store concept:  struct S { using value_type = ...; using key_type = ...;
//   value_type get(int i) const; void set(int i, const value_type&); void swap(int i, int j);
//   key_type key(int i) const;  key_type key_of(const value_type&) const;
//   bool less(key_type a, key_type b) const; }
// Comparisons only ever look at keys, so the scanning loops touch one word per element; whole
// elements move only in swaps and shifts.
#pragma once

#ifdef __HIPCC__
#define VS_HD __host__ __device__
#else
#define VS_HD
#endif

// test hook: lets the host check count how often the depth-limit fallback ran
#ifndef VS_SEL_ON_HEAP_SELECT
#define VS_SEL_ON_HEAP_SELECT()
#endif

namespace vs_sel {

VS_HD inline int floor_lg(int n) {   // std::__lg
    int k = 0;
    while (n > 1) {
        n >>= 1;
        k++;
    }
    return k;
}

template <class S>
VS_HD inline void move_median_to_first(S &s, int result, int a, int b, int c) {
    const auto va = s.key(a), vb = s.key(b), vc = s.key(c);
    if (s.less(va, vb)) {
        if (s.less(vb, vc)) s.swap(result, b);
        else if (s.less(va, vc)) s.swap(result, c);
        else s.swap(result, a);
    } else if (s.less(va, vc)) s.swap(result, a);
    else if (s.less(vb, vc)) s.swap(result, c);
    else s.swap(result, b);
}

template <class S>
VS_HD inline int unguarded_partition(S &s, int first, int last, int pivot) {
    const auto pv = s.key(pivot);   // *pivot is never moved by the swaps below (pivot < first)
    while (true) {
        while (s.less(s.key(first), pv)) ++first;
        --last;
        while (s.less(pv, s.key(last))) --last;
        if (!(first < last)) return first;
        s.swap(first, last);
        ++first;
    }
}

template <class S>
VS_HD inline int unguarded_partition_pivot(S &s, int first, int last) {
    const int mid = first + (last - first) / 2;
    move_median_to_first(s, first, first + 1, mid, last - 1);
    return unguarded_partition(s, first + 1, last, first);
}

template <class S>
VS_HD inline void insertion_sort(S &s, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const auto val = s.get(i);
        const auto vk = s.key_of(val);
        if (s.less(vk, s.key(first))) {
            for (int k = i; k > first; --k) s.set(k, s.get(k - 1));   // move_backward
            s.set(first, val);
        } else {   // __unguarded_linear_insert
            int hole = i, next = i - 1;
            while (s.less(vk, s.key(next))) {
                s.set(hole, s.get(next));
                hole = next;
                --next;
            }
            s.set(hole, val);
        }
    }
}

// heap helpers operate on [first, first+len) with indices relative to first
template <class S, class V>
VS_HD inline void push_heap_rel(S &s, int first, int hole, int top, const V &value) {
    int parent = (hole - 1) / 2;
    const auto vk = s.key_of(value);
    while (hole > top && s.less(s.key(first + parent), vk)) {
        s.set(first + hole, s.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    s.set(first + hole, value);
}

template <class S, class V>
VS_HD inline void adjust_heap(S &s, int first, int hole, int len, const V &value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (s.less(s.key(first + child), s.key(first + child - 1))) child--;
        s.set(first + hole, s.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        s.set(first + hole, s.get(first + child - 1));
        hole = child - 1;
    }
    push_heap_rel(s, first, hole, top, value);
}

template <class S>
VS_HD inline void make_heap(S &s, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    while (true) {
        const auto v = s.get(first + parent);
        adjust_heap(s, first, parent, len, v);
        if (parent == 0) return;
        parent--;
    }
}

template <class S>
VS_HD inline void heap_select(S &s, int first, int middle, int last) {
    make_heap(s, first, middle);
    for (int i = middle; i < last; ++i)
        if (s.less(s.key(i), s.key(first))) {   // __pop_heap(first, middle, i)
            const auto v = s.get(i);
            s.set(i, s.get(first));
            adjust_heap(s, first, 0, middle - first, v);
        }
}

// std::nth_element(first, nth, last)
template <class S>
VS_HD inline void nth_element(S &s, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int depth_limit = floor_lg(last - first) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            VS_SEL_ON_HEAP_SELECT();
            heap_select(s, first, nth + 1, last);
            s.swap(first, nth);
            return;
        }
        --depth_limit;
        const int cut = unguarded_partition_pivot(s, first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(s, first, last);
}

#ifdef __HIPCC__
// ---- wave-cooperative replay of std::nth_element ---------------------------------------------
// introselect's unguarded Hoare partition is a fixed pairing: with L_k the k-th position (ascending)
// whose key is not < pivot and R_k the k-th position (descending, the pivot slot itself being the
// last one) whose key is not > pivot, the loop swaps (L_k, R_k) for every k with L_k < R_k and returns
// min(L_K, R_{K-1}) for the first K with L_K >= R_K.  That is two ballot prefix scans, one parallel
// swap and no data-dependent serial chain, so one wave does a 2000-element partition in a few
// hundred cycles instead of a lane walking it element by element.  Median-of-three, the <= 3 element
// insertion sort and the (rare) heap-select fallback stay on lane 0.  Same permutation as
// nth_element, hence as libstdc++.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// I = the stopper lists' element type: positions are below 16384 here, so 16-bit lists do (they are what decides how many
// tree builds fit a CU's LDS at 4000 keypoints)
template <class S, class I>
__device__ int wave_partition(S &s, int a, int b, typename S::key_type pv, I *sl, I *sr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nL = 0, nR = 0;
    for (int p0 = a; p0 < b; p0 += 64) {   // left stoppers, ascending
        const int p = p0 + lane;
        const bool st = p < b && !s.less(s.key(p), pv);
        const unsigned long long bal = __ballot(st);
        if (st) sl[a + nL + (int)__popcll(bal & lt)] = (I)p;
        nL += (int)__popcll(bal);
    }
    for (int p0 = b - 1; p0 >= a - 1; p0 -= 64) {   // right stoppers, descending, pivot slot a-1 included
        const int p = p0 - lane;
        const bool st = p >= a - 1 && !s.less(pv, s.key(p));
        const unsigned long long bal = __ballot(st);
        if (st) sr[a + nR + (int)__popcll(bal & lt)] = (I)p;
        nR += (int)__popcll(bal);
    }
    wave_sync_lds();
    const int m = nL < nR ? nL : nR;
    int K = 0;
    for (int k0 = 0; k0 < m; k0 += 64) {   // pairs are monotone: count the leading L_k < R_k
        const int k = k0 + lane;
        const bool sw = k < m && (int)sl[a + k] < (int)sr[a + k];
        const unsigned long long bal = __ballot(sw);
        if (sw) s.swap((int)sl[a + k], (int)sr[a + k]);
        K += (int)__popcll(bal);
        if (bal != ~0ull) break;
    }
    const int cl = K < nL ? (int)sl[a + K] : 0x7FFFFFFF;
    const int cr = K > 0 ? (int)sr[a + K - 1] : 0x7FFFFFFF;
    wave_sync_lds();
    return cl < cr ? cl : cr;
}

template <class S, class I>
__device__ void wave_nth_element(S &s, int first, int nth, int last, I *sl, I *sr) {
    const int lane = threadIdx.x & 63;
    if (first == last || nth == last) return;
    int depth_limit = floor_lg(last - first) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            if (lane == 0) {
                heap_select(s, first, nth + 1, last);
                s.swap(first, nth);
            }
            wave_sync_lds();
            return;
        }
        --depth_limit;
        if (lane == 0) move_median_to_first(s, first, first + 1, first + (last - first) / 2, last - 1);
        wave_sync_lds();
        const int cut = wave_partition(s, first + 1, last, s.key(first), sl, sr);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) insertion_sort(s, first, last);
    wave_sync_lds();
}

#endif   // __HIPCC__

}  // namespace vs_sel

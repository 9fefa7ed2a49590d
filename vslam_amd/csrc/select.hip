// The selection half of cv::goodFeaturesToTrack for gfx950 (reference: src/Frame.cpp:61): exact threshold at
// quality * max, rank by (response desc, address desc), greedy min-distance suppression, first maxCorners.
// One workgroup per frame (corner_select_kernel); candidates come from response.hip as 64-bit keys.
#include "image_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// greedy min-distance suppression + first-N by rank, one workgroup per frame
// ------------------------------------------------------------------------------------------
// goodFeaturesToTrack sorts candidates by (value desc, address desc) and accepts a candidate iff
// no already-accepted corner lies within minDistance (featureselect.cpp; the cell grid there is
// only an index).  Acceptance of c depends on higher-ranked candidates within that radius only,
// so the sequential scan is the least fixpoint of
//     c accepted  <=> every higher-ranked neighbour is rejected
//     c rejected  <=> some higher-ranked neighbour is accepted
// which is reached by rounds in which every undecided candidate inspects its neighbourhood;
// each round settles at least the highest-ranked undecided candidate.  The first maxCorners
// accepted in rank order are then the reference's output, in its order.
constexpr int kST = 1024;
constexpr int kDiscMax = 15 * 15;   // (2R+1)^2 positions for R <= 7

struct SelectShared {
    uint32_t wave_cnt[kST / 64];
    uint32_t hist[256];
    uint32_t flag, fill, need, d_star, above;
    unsigned long long prefix;
};

// One fixpoint visit of candidate `off`: 2 = accepted, 3 = rejected, 1 = still blocked by an
// undecided higher-ranked neighbour.  Rank = (response desc, address desc).
__device__ __forceinline__ int nms_visit(const float *__restrict__ E, const uint8_t *S, int w, int h, uint32_t off,
                                         int R, float min_dist_sq) {
    const int y = off / w, x = off - y * w;
    const float val = E[off];
    bool blocked = false;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t noff = (uint32_t)(yy * w + xx);
            const uint8_t sn = S[noff];
            if (sn == 0 || sn == 3) continue;
            const float vn = E[noff];
            const bool higher = (vn > val) || (vn == val && noff > off);
            if (!higher) continue;
            if (sn == 2) return 3;
            blocked = true;
        }
    }
    return blocked ? 1 : 2;
}

// Key T such that exactly `want` of the n distinct keys are >= T (0 when want >= n): MSB radix
// select, 8 bits per pass, stopping as soon as a whole bucket is wanted.
__device__ unsigned long long radix_select_nth(const unsigned long long *K, uint32_t n, uint32_t want,
                                               SelectShared &sh) {
    if (want >= n) return 0ull;
    const int tid = threadIdx.x;
    unsigned long long prefix = 0;
    uint32_t need = want;
    int known_bits = 0;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        __syncthreads();
        for (int i = tid; i < 256; i += kST) sh.hist[i] = 0;
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kST) {
            const unsigned long long key = K[i];
            if (known_bits == 0 || (key >> (64 - known_bits)) == (prefix >> (64 - known_bits)))
                atomicAdd(&sh.hist[(uint32_t)(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid < 256) {   // bucket d is the one where the running count (from the top) crosses `need`
            uint32_t above = 0;
            for (int d = tid + 1; d < 256; d++) above += sh.hist[d];
            const uint32_t mine = sh.hist[tid];
            if (above < need && need <= above + mine) {
                sh.prefix = prefix | ((unsigned long long)tid << shift);
                sh.need = need - above;
                sh.flag = (need - above == mine) ? 1u : 0u;   // whole bucket wanted: done
            }
        }
        __syncthreads();
        prefix = sh.prefix;
        need = sh.need;
        known_bits += 8;
        if (sh.flag) break;
    }
    __syncthreads();
    return prefix;
}

__device__ void bitonic_sort_desc_lds(unsigned long long *buf, int cap) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= cap; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < cap; i += kST) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = buf[i], b = buf[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        buf[i] = b;
                        buf[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
}

// The same network for cap = EPT * kST keys with EPT consecutive keys per thread held in registers:
// strides below EPT are compare-exchanges inside a thread, strides below 64 * EPT are wave shuffles, and
// only the strides that cross waves (10 of the 78 stages at cap = 4096) go through LDS and barriers.
template <int EPT>
__device__ void bitonic_sort_desc_regs(unsigned long long *buf) {
    const int tid = threadIdx.x;
    constexpr int cap = EPT * kST;
    unsigned long long v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) v[e] = buf[EPT * tid + e];
    for (int k = 2; k <= cap; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64 * EPT; j >>= 1) {   // partner in another wave
            const int tj = j / EPT;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; e++) buf[EPT * tid + e] = v[e];
            __syncthreads();
            const bool lower = (tid & tj) == 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const unsigned long long p = buf[EPT * (tid ^ tj) + e];
                const bool desc = ((EPT * tid + e) & k) == 0;
                const bool keep_max = desc == lower;
                v[e] = keep_max ? (v[e] > p ? v[e] : p) : (v[e] < p ? v[e] : p);
            }
        }
        for (; j >= EPT; j >>= 1) {   // partner in another lane of this wave
            const int tj = j / EPT;
            const bool lower = (tid & tj) == 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const unsigned long long p = __shfl_xor(v[e], tj, 64);
                const bool desc = ((EPT * tid + e) & k) == 0;
                const bool keep_max = desc == lower;
                v[e] = keep_max ? (v[e] > p ? v[e] : p) : (v[e] < p ? v[e] : p);
            }
        }
#pragma unroll
        for (int jj = EPT / 2; jj > 0; jj >>= 1) {   // partner in this thread
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    if ((e & jj) == 0) {
                        const bool desc = ((EPT * tid + e) & k) == 0;
                        const unsigned long long a = v[e], b = v[e | jj];
                        const bool swap = desc ? (a < b) : (a > b);
                        v[e] = swap ? b : a;
                        v[e | jj] = swap ? a : b;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; e++) buf[EPT * tid + e] = v[e];
    __syncthreads();
}

// buf[0 .. cap) sorted descending, cap a power of two; every thread of the workgroup calls it.
// EPT = cap / kST is a template parameter of the selection kernel (0: cap < kST) so that each instantiation only
// carries the registers of the network it uses.
template <int EPT>
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *buf, int cap) {
    __syncthreads();
    if (EPT == 0) bitonic_sort_desc_lds(buf, cap);
    else bitonic_sort_desc_regs<(EPT > 0 ? EPT : 1)>(buf);
}

// Gather the keys >= T into LDS (unordered) and sort them descending; returns how many.
template <int EPT>
__device__ uint32_t gather_sorted(const unsigned long long *K, uint32_t n, unsigned long long T,
                                  unsigned long long *sortbuf, int sort_cap, SelectShared &sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) sh.fill = 0;
    for (int i = tid; i < sort_cap; i += kST) sortbuf[i] = 0ull;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kST) {
        const unsigned long long key = K[i];
        if (key >= T) {
            const uint32_t p = atomicAdd(&sh.fill, 1u);
            if (p < (uint32_t)sort_cap) sortbuf[p] = key;
        }
    }
    __syncthreads();
    bitonic_sort_desc<EPT>(sortbuf, sort_cap);
    return sh.fill < (uint32_t)sort_cap ? sh.fill : (uint32_t)sort_cap;
}

// ---- rank window in two passes over the candidate list ---------------------------------------
// Every kept key has thr < response <= max, both known, so the ordered responses share the top
// L = clz(ord(thr) ^ ord(max)) bits; the next log2(bins) bits form a monotone digit.  Pass 1 histograms the
// digits of the kept keys (the histogram borrows the sort buffer), a suffix scan finds the digit d* in
// whose bin the N-th best key lies, pass 2 gathers every kept key with digit >= d* into the sort buffer,
// which is then sorted: its first N entries are the N best-ranked candidates.  Returns false when the
// gathered set would not fit the buffer (heavy ties); the caller then uses the generic radix select.
// n_kept receives the number of keys above the threshold, N_io is clamped to it.
template <int EPT>
__device__ bool rank_window_2pass(const unsigned long long *K, uint32_t n, unsigned long long tkey, uint32_t t32,
                                  uint32_t m32, uint32_t &N_io, unsigned long long *sortbuf, int sort_cap,
                                  SelectShared &sh, uint32_t &n_kept) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *hist = reinterpret_cast<uint32_t *>(sortbuf);
    const int bins = 2 * sort_cap < 4096 ? 2 * sort_cap : 4096;
    const int db = 31 - __clz(bins);
    const int L = (t32 ^ m32) ? __clz(t32 ^ m32) : 32;
    const int shift = 32 - L - db > 0 ? 32 - L - db : 0;
    const uint32_t dmask = (uint32_t)bins - 1u;
    __syncthreads();
    for (int i = tid; i < bins; i += kST) hist[i] = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * kST) {   // four independent loads in flight per thread
        unsigned long long key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * kST + tid;
            key[u] = i < n ? K[i] : 0ull;   // 0 never passes the threshold test
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (key[u] > tkey) atomicAdd(&hist[((uint32_t)(key[u] >> 32) >> shift) & dmask], 1u);
        }
    }
    __syncthreads();
    // suffix scan: thread t owns `per` consecutive bins; above = keys in bins owned by higher threads
    const int per = bins > kST ? bins / kST : 1;
    const int lo = tid * per;
    uint32_t own = 0;
    if (lo < bins)
        for (int b = 0; b < per; b++) own += hist[lo + b];
    uint32_t incl = own;   // becomes the sum over lanes >= lane of this wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += o;
    }
    if (lane == 0) sh.wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t higher = 0, total = 0;
    for (int wv = 0; wv < kST / 64; wv++) {
        const uint32_t c = sh.wave_cnt[wv];
        if (wv > wave) higher += c;
        total += c;
    }
    n_kept = total;
    uint32_t N = N_io;
    if (N > total) N = total;
    if (N > (uint32_t)sort_cap) N = (uint32_t)sort_cap;
    N_io = N;
    if (N == 0) {
        __syncthreads();
        return true;
    }
    uint32_t run = higher + incl - own;
    if (lo < bins)
        for (int b = per - 1; b >= 0; b--) {
            const uint32_t mine = hist[lo + b];
            if (run < N && N <= run + mine) {   // exactly one bin satisfies this
                sh.d_star = (uint32_t)(lo + b);
                sh.above = run;
                sh.need = mine;
            }
            run += mine;
        }
    __syncthreads();
    const uint32_t d_star = sh.d_star;
    const uint32_t gathered = sh.above + sh.need;
    __syncthreads();   // every thread is done with the histogram (and sh.*) before the buffer is reused
    if (gathered > (uint32_t)sort_cap) return false;
    if (tid == 0) sh.fill = 0;
    for (int i = tid; i < sort_cap; i += kST) sortbuf[i] = 0ull;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * kST) {
        unsigned long long key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * kST + tid;
            key[u] = i < n ? K[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (key[u] > tkey && (((uint32_t)(key[u] >> 32) >> shift) & dmask) >= d_star)
                sortbuf[atomicAdd(&sh.fill, 1u)] = key[u];
    }
    __syncthreads();
    bitonic_sort_desc<EPT>(sortbuf, sort_cap);
    return true;
}

// ---- suppression on the rank window, entirely in LDS ---------------------------------------------
// After the sort the responses are no longer needed (rank == index), so the window is compacted to
// offs[i] = pixel offset | status << 30 (status 0 undecided, 1 accepted, 2 rejected) in the first half of
// the buffer, and the second half becomes an open-addressing table of 2 * sort_cap 16-bit slots
// (pixel offset -> rank, verified against offs[]; 0xFFFF = empty; load factor <= 1/2).

__device__ __forceinline__ uint32_t slot_hash(uint32_t q, int hshift) { return (q * 2654435761u) >> hshift; }

__device__ __forceinline__ void slot_insert(uint32_t *slot32, uint32_t smask, int hshift, uint32_t q, uint32_t rank) {
    uint32_t hs = slot_hash(q, hshift);
    while (true) {
        const int sh16 = (hs & 1u) * 16;
        const uint32_t old = slot32[hs >> 1];
        if (((old >> sh16) & 0xFFFFu) == 0xFFFFu) {
            const uint32_t upd = (old & ~(0xFFFFu << sh16)) | (rank << sh16);
            if (atomicCAS(&slot32[hs >> 1], old, upd) == old) return;
            continue;   // the word changed under us (its other half, or this slot): look again
        }
        hs = (hs + 1u) & smask;
    }
}

// The better-ranked window entries (rank < i) within the distance of entry i: their count, and the ranks
// of the first four in list[] (padded with 0xFFFF).  Every candidate that can decide entry i's fate is in
// the table, so this list is all a later visit needs.
__device__ __forceinline__ int nms_collect(const uint32_t *offs, const uint32_t *slot32, uint32_t smask, int hshift,
                                           int w, int h, uint32_t i, int R, float min_dist_sq, uint32_t list[4]) {
    const uint32_t off = offs[i] & kOffMask;
    const int y = off / w, x = off - y * w;
    int cnt = 0;
    list[0] = list[1] = list[2] = list[3] = 0xFFFFu;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t q = (uint32_t)(yy * w + xx);
            uint32_t hs = slot_hash(q, hshift);
            while (true) {
                const uint32_t r = (slot32[hs >> 1] >> ((hs & 1u) * 16)) & 0xFFFFu;
                if (r == 0xFFFFu) break;
                if ((offs[r] & kOffMask) == q) {
                    if (r < i) {
                        if (cnt == 0) list[0] = r;
                        else if (cnt == 1) list[1] = r;
                        else if (cnt == 2) list[2] = r;
                        else if (cnt == 3) list[3] = r;
                        cnt++;
                    }
                    break;
                }
                hs = (hs + 1u) & smask;
            }
        }
    }
    return cnt;
}

// The same through a list of the offsets inside the distance that the host builds and passes as a kernel argument
// (disc.e[k] = dy << 16 | (dx & 0xFFFF); scalar loads, no float test per position), four positions per trip so that
// the four table probes are in flight together; the up to four ranks are packed into one 64-bit value (rank k in
// bits 16k .. 16k+15, unused ones 0xFFFF).
struct DiscTable {
    int n;                 // 0: no table (R == 0 or R > 7); padded to a multiple of 4 with entries that repeat e[0]
    int e[kDiscMax + 3];
};
// The kernel's copy in LDS: read from the kernel-argument segment inside the probing loop, every group of four
// offsets was a scalar load the wave had to wait for — 8 k cycles per trip, 62 of the workgroup's 115 us.
struct DiscLds {
    int n;
    int e[kDiscMax + 3];
};

// A 64 Kbit presence filter over the window's pixel offsets (one hash, set at insertion): most of the disc's positions
// hold no window entry, and asking the table about each of them made a wave walk, for every position, the longest
// probe chain among its 64 lanes (two dependent LDS reads per step): 62 of the workgroup's 115 us.  With the filter a
// lane probes the table only for the positions whose bit is set — its real neighbours and 4 % false positives.
constexpr int kFiltWords = 2048;
__device__ __forceinline__ uint32_t filt_hash(uint32_t q) { return (q * 0x9E3779B1u) >> 16; }

__device__ __forceinline__ int nms_collect_disc(const uint32_t *offs, const uint32_t *slot32, uint32_t smask, int hshift,
                                                const uint32_t *filt, int w, int h, uint32_t i, const DiscLds &disc,
                                                unsigned long long &packed) {
    const uint32_t off = offs[i] & kOffMask;
    const int y = off / w, x = off - y * w;
    int cnt = 0;
    packed = ~0ull;
    for (int c0 = 0; c0 < disc.n; c0 += 32) {   // 32 positions at a time: their filter bits first (independent reads)
        uint32_t hit = 0;
        const int ce = min(32, disc.n - c0);
        for (int e0 = 0; e0 < ce; e0 += 4) {
            uint32_t fw[4], fb[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pk = disc.e[c0 + e0 + u];
                const int dy = pk >> 16, dx = (int)(int16_t)(pk & 0xFFFF);
                const int yy = y + dy, xx = x + dx;
                in[u] = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && e0 + u < ce;
                fb[u] = filt_hash((uint32_t)(yy * w + xx));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) fw[u] = filt[fb[u] >> 5];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (in[u] && ((fw[u] >> (fb[u] & 31u)) & 1u)) hit |= 1u << (e0 + u);
        }
        while (hit) {   // the table, for the few positions that may hold an entry (ascending position order)
            const int e = __ffs(hit) - 1;
            hit &= hit - 1u;
            const int pk = disc.e[c0 + e];
            const uint32_t q = (uint32_t)((y + (pk >> 16)) * w + x + (int)(int16_t)(pk & 0xFFFF));
            uint32_t hh = slot_hash(q, hshift);
            uint32_t rr = (slot32[hh >> 1] >> ((hh & 1u) * 16)) & 0xFFFFu;
            while (rr != 0xFFFFu) {
                if ((offs[rr] & kOffMask) == q) {
                    if (rr < i) {
                        if (cnt < 4) packed = (packed & ~(0xFFFFull << (16 * cnt))) | ((unsigned long long)rr << (16 * cnt));
                        cnt++;
                    }
                    break;
                }
                hh = (hh + 1u) & smask;
                rr = (slot32[hh >> 1] >> ((hh & 1u) * 16)) & 0xFFFFu;
            }
        }
    }
    return cnt;
}

// One visit of window entry i through the table: 1 accepted, 2 rejected, 0 still blocked by an undecided
// better-ranked neighbour.  Statuses are read as they are at this moment (other waves publish theirs
// without a barrier); they only ever go from undecided to decided, so a stale read costs a later visit.
// The statuses other waves publish are read through a VOLATILE pointer, and it has to say "LDS" itself: the compiler's
// address-space inference leaves volatile accesses alone, and through a generic pointer every poll was a flat_load with
// system scope (49 of them in this file) instead of a ds_read.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
__device__ __forceinline__ int nms_visit_lds(const lds_vu32 *offs, const uint32_t *slot32, uint32_t smask,
                                             int hshift, int w, int h, uint32_t i, int R, float min_dist_sq) {
    const uint32_t off = offs[i] & kOffMask;
    const int y = off / w, x = off - y * w;
    bool blocked = false;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -R; dx <= R; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w || (dx == 0 && dy == 0)) continue;
            const float fx = (float)dx, fy = (float)dy;
            if (!(fx * fx + fy * fy < min_dist_sq)) continue;
            const uint32_t q = (uint32_t)(yy * w + xx);
            uint32_t hs = slot_hash(q, hshift);
            while (true) {
                const uint32_t r = (slot32[hs >> 1] >> ((hs & 1u) * 16)) & 0xFFFFu;
                if (r == 0xFFFFu) break;
                const uint32_t e = offs[r];
                if ((e & kOffMask) == q) {
                    if (r < i) {
                        const uint32_t st = e >> 30;
                        if (st == 1u) return 2;
                        if (st == 0u) blocked = true;
                    }
                    break;
                }
                hs = (hs + 1u) & smask;
            }
        }
    }
    return blocked ? 0 : 1;
}

// One workgroup per frame.
//  fast path: only the first maxCorners ACCEPTED corners in rank order are wanted, and a
//    candidate's fate depends on higher-ranked candidates only, so suppression is run on the N
//    best-ranked candidates (N a little above maxCorners), sorted in LDS; if they yield fewer than
//    maxCorners survivors N is doubled (decisions already made stay valid).
//  slow path (N would exceed the LDS sort buffer): suppression over every candidate, then select.
template <int EPT>
__global__ __launch_bounds__(kST) __attribute__((amdgpu_waves_per_eu(8, 8))) void corner_select_kernel(
    float *__restrict__ eig, const uint32_t *__restrict__ cutkey, uint32_t *__restrict__ need, int pass, int w, int h,
    uint8_t *__restrict__ state, unsigned long long *__restrict__ keys, const uint32_t *__restrict__ counts, size_t key_cap,
    int max_corners, float min_dist, float min_dist_sq, int sort_cap, float *__restrict__ out_xy,
    int32_t *__restrict__ out_n, int kp_stride, int32_t *__restrict__ overflow,
    const uint32_t *__restrict__ frame_max, double quality, int use_lists, const DiscTable disc,
    const uint32_t *__restrict__ raw_counts, const VsCornerPool pool, int32_t *__restrict__ errflag) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned long long *sortbuf = reinterpret_cast<unsigned long long *>(smem_raw);
    __shared__ SelectShared sh;
    __shared__ DiscLds sdisc;
    __shared__ uint32_t sfilt[kFiltWords];
    for (int i = threadIdx.x; i < kDiscMax + 3; i += kST) sdisc.e[i] = disc.e[i];
    if (threadIdx.x == 0) sdisc.n = disc.n;   // (the first barrier below orders these)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // pass 1 = the rerun of the frames whose first selection ran out of keys above the cut (their list is complete now)
    // pass 2 = the frames filed in the fallback pool: a workgroup per slot, inputs by slot (the plain pipeline's exact keys
    //          of every candidate, whole-image E / S), outputs by frame
    int f = blockIdx.x;
    const int slot = blockIdx.x;
    if (pass == 2) {
        if (slot >= vs_pool_used(pool)) return;
        f = pool.frame[slot];
    }
    if (pass == 1 && need[f] == 0u) return;
    float *E = eig + (size_t)slot * w * h;   // slow path only: responses of the candidates, scattered from their keys
    uint8_t *S = state + (size_t)slot * w * h;
    unsigned long long *K = keys + (size_t)slot * key_cap;
    float2 *O = reinterpret_cast<float2 *>(out_xy) + (size_t)f * kp_stride;
    uint32_t n = counts[slot];
    // Bounded lists (passes 0 and 1 of the two-tier path; E and S do not exist there): a frame whose raw list did not fit,
    // or whose selection needs the per-pixel maps, is filed in the pool and redone from scratch by pass 2; if the pool is
    // full as well the frame gets no corners and bit 2 of the context's error word is raised (VSLAM_ERR_CAPACITY).
    const bool bounded = pass != 2 && pool.count != nullptr;
    auto to_pool = [&]() {   // every thread calls it, then returns
        if (tid == 0) {
            const int t = atomicAdd(pool.count, 1);
            if (t < pool.slots) pool.frame[t] = f;
            else atomicOr(errflag, 4);
            out_n[f] = 0;
        }
    };
    if (bounded && raw_counts[f] > key_cap) {
        to_pool();
        return;
    }
    if (n > key_cap) {
        if (tid == 0) atomicAdd(overflow, 1);
        n = (uint32_t)key_cap;
    }
    const int R = min_dist >= 1.f ? (int)ceilf(min_dist) : 0;
    const uint32_t want_max = (uint32_t)max_corners;

    // exact threshold: the fused detector prefilters with a running maximum, so only keys whose response is
    // > (float)(max * quality) count (every key of an exactly-thresholded list passes)
    float thr = (float)((double)ord2f(frame_max[slot]) * quality);
    if (thr == 0.f) thr = 0.f;   // -0 -> +0 so the ordered-key compare equals the float compare
    // The two-tier detector's key list is complete only above cutkey[f] (a key below it may be outranked by a pixel that
    // was never evaluated): such keys are ignored, and if the selection then runs out of keys before it has max_corners
    // corners the frame is flagged and redone (pass 1) on the complete list.
    const uint32_t cut32 = (pass == 0 && cutkey) ? cutkey[f] : 0u;
    const bool cut_binds = cut32 > f2ord(thr);
    const uint32_t t32 = cut_binds ? cut32 - 1u : f2ord(thr), m32 = frame_max[slot];   // keys >= cut32 pass
    auto finish = [&](uint32_t found) {   // tid 0: the count, or the request for the rerun
        if (cut_binds && found < want_max) {
            need[f] = 1u;
            out_n[f] = 0;
        } else {
            out_n[f] = (int32_t)(found < want_max ? found : want_max);
        }
    };
    unsigned long long tkey = ((unsigned long long)t32 << 32) | 0xFFFFFFFFull;
    bool compacted = false;
    // drop the keys at or below the threshold from K (generic paths only; the two-pass window filters on the fly)
    auto compact_keys = [&]() {
        uint32_t kept = 0;
        for (uint32_t base = 0; base < n; base += kST) {
            const uint32_t i = base + tid;
            unsigned long long key = 0;
            bool keep = false;
            if (i < n) {
                key = K[i];
                keep = key > tkey;
            }
            const unsigned long long bal = __ballot(keep);
            __syncthreads();
            if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int wv = 0; wv < kST / 64; wv++) {
                const uint32_t c = sh.wave_cnt[wv];
                if (wv < wave) pre += c;
                tot += c;
            }
            if (keep) K[kept + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = key;
            kept += tot;
        }
        n = kept;
        tkey = 0ull;   // every remaining key passes (no key is 0: its response would be a NaN pattern)
        compacted = true;
        __syncthreads();
    };

    // ---------------------------------------------------------------- fast path
    uint32_t *offs = reinterpret_cast<uint32_t *>(sortbuf);
    uint32_t *slot32 = offs + sort_cap;
    const uint32_t smask = 2u * (uint32_t)sort_cap - 1u;
    const int hshift = 31 - (31 - __clz(sort_cap));   // 32 - log2(2 * sort_cap)
    bool done = false;
    {
        uint32_t N = want_max + want_max / 4 + 64;
        if (R == 0) N = want_max;
        while (true) {
            uint32_t got = N, n_kept = 0;
            if (!rank_window_2pass<EPT>(K, n, tkey, t32, m32, got, sortbuf, sort_cap, sh, n_kept)) {
                if (!compacted) compact_keys();
                n_kept = n;
                got = N < n ? N : n;
                if (got > (uint32_t)sort_cap) got = (uint32_t)sort_cap;
                const unsigned long long T = radix_select_nth(K, n, got, sh);
                got = gather_sorted<EPT>(K, n, T, sortbuf, sort_cap, sh);
            }
            if (R > 0) {
                // window -> offs[] (entries are read, then a barrier, then written: offs[i] overlays sortbuf[i / 2])
                for (uint32_t c0 = 0; c0 < got; c0 += kST) {
                    const uint32_t i = c0 + tid;
                    const uint32_t v = i < got ? (uint32_t)sortbuf[i] : 0u;
                    __syncthreads();
                    if (i < got) offs[i] = v;
                }
                __syncthreads();
                for (int i = tid; i < sort_cap; i += kST) slot32[i] = 0xFFFFFFFFu;
                for (int i = tid; i < kFiltWords; i += kST) sfilt[i] = 0u;
                __syncthreads();
                uint32_t pend = 0;   // bit k: window entry tid + k * kST is undecided (sort_cap <= 16 * kST)
                {
                    int k = 0;
                    for (uint32_t i = tid; i < got; i += kST, k++) {
                        slot_insert(slot32, smask, hshift, offs[i], i);
                        const uint32_t fb = filt_hash(offs[i]);
                        atomicOr(&sfilt[fb >> 5], 1u << (fb & 31u));
                        pend |= 1u << k;
                    }
                }
                __syncthreads();
                // first visit: who can decide my fate?  Nobody -> accepted; up to four -> remember their
                // ranks (when the launch provides the list region); more -> table visits every time.
                uint2 *nbr = use_lists ? reinterpret_cast<uint2 *>(smem_raw + sizeof(unsigned long long) * (size_t)sort_cap) : nullptr;
                uint32_t full = 0;
                {
                    int k = 0;
                    for (uint32_t i = tid; i < got; i += kST, k++) {
                        int cnt;
                        uint2 ranks;
                        if (disc.n) {
                            unsigned long long packed;
                            cnt = nms_collect_disc(offs, slot32, smask, hshift, sfilt, w, h, i, sdisc, packed);
                            ranks = make_uint2((uint32_t)packed, (uint32_t)(packed >> 32));
                        } else {
                            uint32_t list[4];
                            cnt = nms_collect(offs, slot32, smask, hshift, w, h, i, R, min_dist_sq, list);
                            ranks = make_uint2(list[0] | (list[1] << 16), list[2] | (list[3] << 16));
                        }
                        if (cnt == 0) {
                            offs[i] |= 1u << 30;
                            pend &= ~(1u << k);
                        } else if (nbr && cnt <= 4) {
                            nbr[i] = ranks;
                        } else {
                            full |= 1u << k;
                        }
                    }
                }
                __syncthreads();
                // Suppression fixpoint.  An entry's fate needs its better-ranked neighbours decided first, and chains
                // of such dependencies run along image edges, so the number of rounds is the longest chain: each
                // wave therefore polls on its own, without workgroup barriers (statuses live in LDS, every wave of the
                // workgroup is resident, and the rank order makes the dependency graph acyclic).
                {
                    lds_vu32 *voffs = (lds_vu32 *)offs;
                    int spins = 0;
                    while (__any(pend != 0)) {
                        int k = 0;
                        for (uint32_t i = tid; i < got; i += kST, k++) {
                            if (!((pend >> k) & 1u)) continue;
                            int d;
                            if ((full >> k) & 1u) {
                                d = nms_visit_lds(voffs, slot32, smask, hshift, w, h, i, R, min_dist_sq);
                            } else {
                                const uint2 l = nbr[i];
                                const uint32_t r[4] = {l.x & 0xFFFFu, l.x >> 16, l.y & 0xFFFFu, l.y >> 16};
                                bool blocked = false, rejected = false;
#pragma unroll
                                for (int t = 0; t < 4; t++) {
                                    if (r[t] == 0xFFFFu) continue;
                                    const uint32_t st = voffs[r[t]] >> 30;
                                    rejected |= st == 1u;
                                    blocked |= st == 0u;
                                }
                                d = rejected ? 2 : (blocked ? 0 : 1);
                            }
                            if (d) {
                                voffs[i] = voffs[i] | ((uint32_t)d << 30);
                                pend &= ~(1u << k);
                            }
                        }
                        if (++spins > (1 << 22)) {   // cannot happen (acyclic); never hang the device on a defect
                            if (lane == 0) atomicAdd(overflow, 1);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
            }
            // survivors in rank order: ordered scan over the window
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t i0 = 0; i0 < got; i0 += kST) {
                const uint32_t i = i0 + tid;
                uint32_t off = 0;
                bool acc = false;
                if (i < got) {
                    if (R == 0) {
                        off = (uint32_t)sortbuf[i];
                        acc = true;
                    } else {
                        const uint32_t e = offs[i];
                        off = e & kOffMask;
                        acc = (e >> 30) == 1u;
                    }
                }
                const unsigned long long bal = __ballot(acc);
                __syncthreads();
                if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t pre = 0, tot = 0;
                for (int wv = 0; wv < kST / 64; wv++) {
                    const uint32_t c = sh.wave_cnt[wv];
                    if (wv < wave) pre += c;
                    tot += c;
                }
                const uint32_t pos = base + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                // written speculatively: if this attempt falls short the next one rewrites the
                // same prefix with the same values (rank order does not change)
                if (acc && pos < want_max) {
                    const int y = off / w, x = off - y * w;
                    O[pos] = make_float2((float)x, (float)y);
                }
                base += tot;
            }
            if (base >= want_max || got == n_kept) {
                if (tid == 0) finish(base);
                done = true;
                break;
            }
            if (got == (uint32_t)sort_cap) break;   // cannot widen in LDS: slow path
            N *= 2;
        }
    }
    if (done) return;
    if (bounded) {   // the slow path needs whole-image maps: the pool's pass has them
        to_pool();
        return;
    }

    // ---------------------------------------------------------------- slow path (rare)
    // suppression over every candidate through a per-pixel state map in global memory
    // (0 none, 1 undecided, 2 accepted, 3 rejected), which this path initialises itself
    if (!compacted) compact_keys();
    if (R > 0) {
        for (uint32_t i = tid; i < (uint32_t)(w * h); i += kST) S[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kST) {   // nms_visit reads the response of candidates only
            const unsigned long long key = K[i];
            S[(uint32_t)key] = 1;
            E[(uint32_t)key] = ord2f((uint32_t)(key >> 32));
        }
    }
    if (R > 0) {
        while (true) {
            __syncthreads();
            if (tid == 0) sh.flag = 0;
            __syncthreads();
            bool pending = false;
            for (uint32_t i = tid; i < n; i += kST) {
                const uint32_t off = (uint32_t)K[i];
                if (S[off] != 1) continue;
                const int d = nms_visit(E, S, w, h, off, R, min_dist_sq);
                if (d == 1) pending = true;
                else S[off] = (uint8_t)d;
            }
            if (pending) sh.flag = 1;
            __syncthreads();
            if (!sh.flag) break;
        }
    }
    __syncthreads();
    // compact accepted keys to the front of K (entries are read before any write of the same round)
    uint32_t n_acc = 0;
    for (uint32_t base = 0; base < n; base += kST) {
        const uint32_t i = base + tid;
        unsigned long long key = 0;
        bool acc = false;
        if (i < n) {
            key = K[i];
            acc = (R == 0) || S[(uint32_t)key] == 2;
        }
        const unsigned long long bal = __ballot(acc);
        __syncthreads();
        if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int wv = 0; wv < kST / 64; wv++) {
            const uint32_t c = sh.wave_cnt[wv];
            if (wv < wave) pre += c;
            tot += c;
        }
        if (acc) K[n_acc + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = key;
        n_acc += tot;
    }
    __syncthreads();
    uint32_t want = n_acc < want_max ? n_acc : want_max;
    if (want > (uint32_t)sort_cap) want = (uint32_t)sort_cap;
    const unsigned long long T = radix_select_nth(K, n_acc, want, sh);
    gather_sorted<EPT>(K, n_acc, T, sortbuf, sort_cap, sh);
    for (uint32_t i = tid; i < want; i += kST) {
        const uint32_t off = (uint32_t)sortbuf[i];
        const int y = off / w, x = off - y * w;
        O[i] = make_float2((float)x, (float)y);
    }
    if (tid == 0) finish(n_acc);
}

}  // namespace

int vs_launch_good_features(vslam_ctx *ctx, const uint8_t *gray, int frames, int w, int h,
                            int max_corners, double quality, double min_distance, int kp_stride,
                            float *xy, int32_t *n, const VsBgrSource *bgr) {
    VS_REQUIRE(ctx, gray && xy && n, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w >= 3 && h >= 3, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, max_corners > 0 && max_corners <= kp_stride, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, min_distance < 64.0, VSLAM_ERR_CAPACITY);
    VS_REQUIRE(ctx, (size_t)w * h < (1u << 28), VSLAM_ERR_CAPACITY);   // pixel offsets share their word with 4 flag bits (image_common.h)
    const size_t px = (size_t)w * h;
    float *eig = nullptr;
    uint32_t *block = nullptr;
    uint8_t *state = nullptr;
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    int rc;
    // The two-tier detector (gray rows of a multiple of 4 bytes: width % 4 == 0, or padded rows) keeps BOUNDED per-frame lists: the detector lists about
    // 10 x max_corners pixels on image data, so 16 x max_corners + 4096 entries hold a frame's list with room to spare
    // (VSLAM_OPT_CORNER_LIST_CAP: another bound; -1: the whole image, which nothing can overflow).  A frame that does
    // overflow it (response plateaus, noise: every interior pixel can be listed), or whose selection needs per-pixel maps
    // (its rank window outgrew LDS), is filed in a small pool of whole-image scratch and redone from scratch by the plain
    // exact pipeline (pass 2 below): results never depend on the bound.  Should more frames of one call need the pool
    // than it has slots, those frames get no corners and vslam_ctx_synchronize reports VSLAM_ERR_CAPACITY.
    // Gray rows that are no multiple of 4 bytes (a caller's packed image of such a width that nobody padded: widths below 64)
    // run the plain pipeline on whole-image buffers for every frame.
    const bool two_tier = (vs_pitch(ctx, w) % 4 == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);   // (as vs_launch_response_candidates decides)
    size_t key_cap = px;
    if (two_tier && ctx->corner_list_cap >= 0) {
        const size_t want = ctx->corner_list_cap > 0 ? (size_t)ctx->corner_list_cap : 16 * (size_t)max_corners + 4096;
        if (want < key_cap) key_cap = want;
    }
    VsCornerPool pool;
    pool.slots = two_tier ? (frames < 4 ? frames : (frames / 16 < 4 ? 4 : (frames / 16 > 64 ? 64 : frames / 16))) : 0;
    pool.key_cap = px;
    // every per-frame counter of the pipeline in one block, so that one memset clears all of it:
    // counts[F], overflow[1], fmax[F], low[F], count2[F], count3[F], need[F], cutkey[F], hist[F][bins],
    // pool: count[1], frame[slots], counts[slots], fmax[slots]
    // (rounded up to 256 bytes: hipMemsetAsync clears a size that is not a multiple of its wide stores with a second kernel)
    const size_t F = (size_t)frames, words = (7 * F + 1 + vs_response_hist_words(frames) + 1 + 3 * (size_t)pool.slots + 63) & ~(size_t)63;
    if ((rc = vs_arena_get(ctx, "gf.counts", sizeof(uint32_t) * words, (void **)&block))) return rc;
    VsCornerCounters c;
    c.counts = block;
    int32_t *overflow = reinterpret_cast<int32_t *>(block + F);
    c.fmax = block + F + 1;
    c.low = c.fmax + F;
    c.count2 = c.low + F;
    c.count3 = c.count2 + F;
    c.need = c.count3 + F;
    c.cutkey = c.need + F;
    c.hist = c.cutkey + F;
    {
        uint32_t *pw = c.hist + vs_response_hist_words(frames);
        pool.count = reinterpret_cast<int32_t *>(pw);
        pool.frame = reinterpret_cast<int32_t *>(pw + 1);
        pool.counts = pw + 1 + pool.slots;
        pool.fmax = pw + 1 + 2 * (size_t)pool.slots;
    }
    ctx->stat_counts = two_tier ? c.counts : nullptr;
    ctx->stat_pool_count = two_tier ? pool.count : nullptr;
    ctx->stat_frames = frames;
    ctx->stat_pool_slots = pool.slots;
    ctx->stat_px = px;
    int32_t *errflag = nullptr;
    if ((rc = vs_device_errflag(ctx, &errflag))) return rc;
    if (two_tier) {
        if ((rc = vs_arena_get(ctx, "gf.pool.keys", sizeof(unsigned long long) * px * pool.slots, (void **)&pool.keys))) return rc;
        if ((rc = vs_arena_get(ctx, "gf.pool.eig", sizeof(float) * px * pool.slots, (void **)&pool.eig))) return rc;
        if ((rc = vs_arena_get(ctx, "gf.pool.state", px * pool.slots, (void **)&pool.state))) return rc;
    } else {
        if ((rc = vs_arena_get(ctx, "gf.eig", sizeof(float) * px * frames, (void **)&eig))) return rc;
        if ((rc = vs_arena_get(ctx, "gf.state", px * frames, (void **)&state))) return rc;
        pool = VsCornerPool{};
    }
    // keys = the detector's list, keys2 = the exact keys of the two-tier path
    if ((rc = vs_arena_get(ctx, "gf.keys", sizeof(unsigned long long) * key_cap * frames, (void **)&keys))) return rc;
    if (two_tier)
        if ((rc = vs_arena_get(ctx, "gf.keys2", sizeof(unsigned long long) * key_cap * frames, (void **)&keys2))) return rc;

    VS_HIP(ctx, hipMemsetAsync(block, 0, sizeof(uint32_t) * words, ctx->stream));
    // The selection ranks 1.25 x max_corners + 64 candidates (twice that if the suppression leaves it short, which it
    // rarely does): evaluate exactly that first window plus slack — the exact tier is bound by its scattered window
    // reads, about 640 B per pixel — and rerun a frame completely if it runs out.
    const uint32_t n_safe = ctx->corner_window_pct > 0
                                ? (uint32_t)((unsigned long long)max_corners * (unsigned)ctx->corner_window_pct / 100u) + 128u
                                : 0xFFFFFFFFu;   // 0: everything
    int raw_list = 0;
    if ((rc = vs_launch_response_candidates(ctx, gray, frames, w, h, quality, eig, c, keys, keys2, key_cap, n_safe, &raw_list, bgr))) return rc;
    if (ctx->fork_after_eigen) {   // the caller runs an independent stage on the auxiliary stream beside the selection
        VS_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        VS_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    }
    {
        int sort_cap = 2;
        while (sort_cap < 2 * max_corners && sort_cap < 16384) sort_cap <<= 1;
        while (sort_cap < max_corners) sort_cap <<= 1;   // at least max_corners slots
        size_t lds = sizeof(unsigned long long) * (size_t)sort_cap;
        VS_REQUIRE(ctx, lds <= 128 * 1024, VSLAM_ERR_CAPACITY);
        // second region of the same size: per window entry the ranks of up to four candidates that can suppress it
        const int use_lists = 2 * lds <= 128 * 1024;
        if (use_lists) lds *= 2;
        const float md = (float)min_distance;
        const float md2 = (float)(min_distance * min_distance);   // `minDistance *= minDistance` in double, compared as float
        DiscTable disc;   // the offsets with dx^2 + dy^2 < minDistance^2 (float arithmetic as in the kernels), dy-major
        disc.n = 0;
        const int R = md >= 1.f ? (int)ceilf(md) : 0;
        if (R > 0 && R <= 7) {
            for (int dy = -R; dy <= R; dy++)
                for (int dx = -R; dx <= R; dx++) {
                    const float fx = (float)dx, fy = (float)dy;
                    if ((dx != 0 || dy != 0) && fx * fx + fy * fy < md2) disc.e[disc.n++] = (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xFFFFu));
                }
        }
        for (int k = disc.n; k < disc.n + 3 && k < kDiscMax + 3; k++) disc.e[k] = disc.n ? disc.e[0] : 0;
#define VS_SELECT_LAUNCH(EPT, PASS, KEYS, COUNTS)                                                                          \
    do {                                                                                                                   \
        if (!ctx->attr_done["corner_select" #EPT]) {                                                                       \
            VS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(corner_select_kernel<EPT>),                     \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));                     \
            ctx->attr_done["corner_select" #EPT] = true;                                                                   \
        }                                                                                                                  \
        if (PASS == 2)                                                                                                     \
            corner_select_kernel<EPT><<<pool.slots, kST, lds, ctx->stream>>>(                                              \
                pool.eig, nullptr, c.need, 2, w, h, pool.state, pool.keys, pool.counts, pool.key_cap, max_corners, md, md2, \
                sort_cap, xy, n, kp_stride, overflow, pool.fmax, quality, use_lists, disc, nullptr, pool, errflag);         \
        else                                                                                                               \
            corner_select_kernel<EPT><<<frames, kST, lds, ctx->stream>>>(                                                  \
                eig, raw_list ? c.cutkey : nullptr, c.need, PASS, w, h, state, KEYS, COUNTS, key_cap, max_corners, md, md2, \
                sort_cap, xy, n, kp_stride, overflow, c.fmax, quality, use_lists, disc, raw_list ? c.counts : nullptr,     \
                raw_list ? pool : VsCornerPool{}, errflag);                                                                \
    } while (0)
#define VS_SELECT_DISPATCH(PASS, KEYS, COUNTS)                                                                             \
    switch (sort_cap / kST) {                                                                                              \
        case 1: VS_SELECT_LAUNCH(1, PASS, KEYS, COUNTS); break;                                                            \
        case 2: VS_SELECT_LAUNCH(2, PASS, KEYS, COUNTS); break;                                                            \
        case 4: VS_SELECT_LAUNCH(4, PASS, KEYS, COUNTS); break;                                                            \
        case 8: VS_SELECT_LAUNCH(8, PASS, KEYS, COUNTS); break;                                                            \
        case 16: VS_SELECT_LAUNCH(16, PASS, KEYS, COUNTS); break;                                                          \
        default: VS_SELECT_LAUNCH(0, PASS, KEYS, COUNTS); break; /* sort_cap < kST */                                      \
    }
        {
            VsProfScope ps(ctx, "corner_select_kernel");
            if (raw_list) VS_SELECT_DISPATCH(0, keys2, c.count2)
            else VS_SELECT_DISPATCH(0, keys, c.counts)
        }
        if (raw_list) {
            // the rerun, for the frames (if any) whose selection ran out of keys above the cut: every listed pixel is
            // evaluated, then selected from again.  Both launches return at once for the other frames.
            if ((rc = vs_launch_corner_exact(ctx, gray, frames, w, h, c, keys, keys2, key_cap, 0u, 1))) return rc;
            {
                VsProfScope ps(ctx, "corner_rerun_kernels");
                VS_SELECT_DISPATCH(1, keys2, c.count3)
            }
            // the pool's frames (usually none: three launches that return at once), redone by the plain exact pipeline
            if ((rc = vs_launch_pool_candidates(ctx, gray, w, h, quality, pool))) return rc;
            VsProfScope ps(ctx, "corner_rerun_kernels");
            VS_SELECT_DISPATCH(2, pool.keys, pool.counts)
        }
#undef VS_SELECT_DISPATCH
#undef VS_SELECT_LAUNCH
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// ORB::compute for provided keypoints on gfx950 (reference: src/Frame.cpp:57,64-72): the image-border filter and the
// steered BRIEF descriptor sampled from the blurred image.
#include "image_common.h"

#include <cstdio>
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------
// ORB::compute for provided keypoints: border filter (ordered) + steered BRIEF
// ------------------------------------------------------------------------------------------
constexpr int kKT = 256;

// Tiles of the tile-staged descriptor kernel (rbrief_tile_kernel): kTW x kTH pixels; a frame has at most kTilesMax of
// them (the launcher doubles the tile size until that holds).
constexpr int kTW = 128, kTH = 128, kTilesMax = 1024;

// KeyPointsFilter::runByImageBorder(kps, size, 31): keep pt inside [31, w-31) x [31, h-31), order kept.
// With tile_start != nullptr the kept keypoints are also filed by image tile for the descriptor kernel:
// tile_kp[f][tile_start[f][t] .. tile_start[f][t + 1]) = (x | y << 16, index into the kept list) of the keypoints of tile t.
__global__ __launch_bounds__(kKT) void keypoint_border_kernel(const float *__restrict__ xy_in,
                                                              const int32_t *__restrict__ n_in, int kp_stride,
                                                              int w, int h, float *__restrict__ xy_out,
                                                              int32_t *__restrict__ n_out, int tw, int th, int tiles_x,
                                                              int ntiles, int32_t *__restrict__ tile_start,
                                                              int32_t *__restrict__ tile_kp) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_cnt[kKT / 64];
    __shared__ int s_base;
    __shared__ int s_tile[kTilesMax + 1];
    const float2 *I = reinterpret_cast<const float2 *>(xy_in) + (size_t)f * kp_stride;
    float2 *O = reinterpret_cast<float2 *>(xy_out) + (size_t)f * kp_stride;
    const int n = n_in[f];
    const int border = 31;
    const bool any = !(h <= border * 2 || w <= border * 2);
    if (tid == 0) s_base = 0;
    if (tile_start)
        for (int t = tid; t <= ntiles; t += kKT) s_tile[t] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kKT) {
        const int i = i0 + tid;
        float2 p = make_float2(0.f, 0.f);
        bool keep = false;
        if (i < n) {
            p = I[i];
            keep = any && p.x >= (float)border && p.x < (float)(w - border) && p.y >= (float)border &&
                   p.y < (float)(h - border);
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_cnt[wave] = (int)__popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int wv = 0; wv < wave; wv++) off += s_cnt[wv];
        if (keep) {
            O[off + (int)__popcll(bal & ((1ull << lane) - 1ull))] = p;
            if (tile_start) atomicAdd(&s_tile[((int)rintf(p.y) / th) * tiles_x + (int)rintf(p.x) / tw + 1], 1);   // count, shifted by one
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int wv = 0; wv < kKT / 64; wv++) t += s_cnt[wv];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) n_out[f] = s_base;
    if (!tile_start) return;
    // inclusive scan of the shifted counts = the tiles' start offsets (ntiles + 1 values); one wave, 16 values per lane
    if (wave == 0) {
        int run = 0;
        for (int b0 = 0; b0 <= ntiles; b0 += 64) {
            const int t = b0 + lane;
            int v = t <= ntiles ? s_tile[t] : 0;
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(v, o, 64);
                if (lane >= o) v += u;
            }
            if (t <= ntiles) s_tile[t] = v + run;
            run += __shfl(v, 63, 64);
        }
    }
    __syncthreads();
    int32_t *TS = tile_start + (size_t)f * (kTilesMax + 1);
    for (int t = tid; t <= ntiles; t += kKT) TS[t] = s_tile[t];
    __syncthreads();
    // scatter: s_tile[t] becomes the fill position of tile t (the start offsets are already in memory)
    const int kept = s_base;
    int2 *TK = reinterpret_cast<int2 *>(tile_kp) + (size_t)f * kp_stride;
    for (int i = tid; i < kept; i += kKT) {
        const float2 p = O[i];   // written above by this workgroup (same addresses, after barriers)
        const int x = (int)rintf(p.x), y = (int)rintf(p.y);   // cvRound(pt): where ORB samples
        TK[atomicAdd(&s_tile[(y / th) * tiles_x + x / tw], 1)] = make_int2(x | (y << 16), i);
    }
}

// One lane per (keypoint, descriptor byte): 8 tests = 16 gathers from the blurred image.
// The 512 rotated sample offsets are the same for every keypoint (one global angle), so each
// workgroup rotates the pattern once into LDS:  x = px*a - py*b, y = px*b + py*a, cvRound.
__global__ __launch_bounds__(kKT) void rbrief_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                     const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
                                                     int kp_stride, float ca, float sa,
                                                     const int8_t *__restrict__ pattern,
                                                     uint8_t *__restrict__ desc, int frames, int per_frame) {
    __shared__ int s_off[512];
    const int tid = threadIdx.x;
    int f, bx;
    vs_xcd_item_block(blockIdx.x, per_frame, f, bx);   // a frame's patches are gathered through one XCD's L2
    if (f >= frames) return;
    const int n = n_arr[f];
    const int kp0 = bx * (kKT / 32);
    if (kp0 >= n) return;
    for (int i = tid; i < 512; i += kKT) {
        const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
        const float a1 = px * ca, a2 = py * sa, b1 = px * sa, b2 = py * ca;
        const float rx = a1 - a2, ry = b1 + b2;
        const int ix = (int)rintf(rx), iy = (int)rintf(ry);
        s_off[i] = iy * w + ix;
    }
    __syncthreads();
    const int kp = kp0 + (tid >> 5), byte = tid & 31;
    if (kp >= n) return;
    const float2 p = reinterpret_cast<const float2 *>(xy)[(size_t)f * kp_stride + kp];
    const int cx = (int)rintf(p.x), cy = (int)rintf(p.y);
    const uint8_t *center = blurred + (size_t)f * w * h + (size_t)cy * w + cx;
    uint32_t val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int t0 = center[s_off[(byte * 8 + bit) * 2]];
        const int t1 = center[s_off[(byte * 8 + bit) * 2 + 1]];
        val |= (uint32_t)(t0 < t1) << bit;
    }
    desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + byte] = (uint8_t)val;
}

// The same descriptor with the keypoint's patch staged through LDS (width % 4 == 0).
// The direct form is a chain of dependent round trips per workgroup of 8 keypoints (pattern, keypoint, 16
// scattered byte loads per lane, each 64 separate addresses for the texture addresser).  Here
//  * the pattern is rotated once per launch (rbrief_rotate_kernel) instead of once per workgroup,
//  * a workgroup walks through 64 keypoints, 8 at a time; the 32 lanes of a keypoint copy its patch rows
//    (2R+1 rows of up to 36 bytes, R = the largest rotated offset; dword loads, one or two cache lines per row) into
//    LDS and take the 16 samples per lane from there,
//  * the patch loads of the next 8 keypoints are in flight while the current 8 are sampled (two LDS buffers).
// Sample offsets are kept transposed ([sample][byte]) so the 32 lanes of a keypoint read 32 consecutive words.
constexpr int kRPitch = 36;     // bytes per staged patch row (9 dwords: 2 * 16 + 1 columns plus alignment slack)
constexpr int kRMax = 16;       // largest |offset| staged: covers the 31x31 pattern at the angles cv::KeyPoint's default
                                // (-1 degree) and small rotations give; wider rotated patterns are sampled directly.
                                // Keeping the patch small is what lets 7 workgroups share a CU's LDS.
constexpr int kRRows = 2 * kRMax + 1;
constexpr int kRGroups = 8;     // groups of 8 keypoints per workgroup
constexpr int kRLoads = (kRRows * (kRPitch / 4) + 31) / 32;   // patch dwords per lane, worst case

// table[i] = (iy << 16) | (ix & 0xFFFF) for sample i (test i / 2, side i & 1), table[512] = max(|ix|, |iy|)
__global__ __launch_bounds__(512) void rbrief_rotate_kernel(const int8_t *__restrict__ pattern, float ca, float sa,
                                                            int32_t *__restrict__ table) {
    __shared__ int s_R;
    const int i = threadIdx.x;
    if (i == 0) s_R = 0;
    __syncthreads();
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float a1 = px * ca, a2 = py * sa, b1 = px * sa, b2 = py * ca;
    const float rx = a1 - a2, ry = b1 + b2;
    const int ix = (int)rintf(rx), iy = (int)rintf(ry);
    table[i] = (int32_t)(((uint32_t)iy << 16) | ((uint32_t)ix & 0xFFFFu));
    atomicMax(&s_R, max(abs(ix), abs(iy)));
    __syncthreads();
    if (i == 0) table[512] = s_R;
}

// What bounds this kernel is the latency of the patch loads: a workgroup's eight groups are a chain of
// load -> LDS -> sample steps, and with the loads of only the next group in flight while one group is sampled (round
// 2's form, 113 VGPRs, 4 waves / SIMD) every step waited about a memory round trip: 0.29 ms for a kernel whose LDS
// and vector work is a third of that.  Here the loads run TWO groups ahead (two register sets, the loop unrolled by
// two so that their roles are static), rows are staged at a fixed pitch in linear dword order (LDS address = index,
// no per-load address registers), and the sample offsets live in registers.
__device__ __forceinline__ void rbrief_fetch(uint32_t (&pre)[kRLoads], const uint8_t *img, int w, int cx, int cy, int R, int total,
                                             int l32, bool live) {
    if (!live) return;
    const uint8_t *src = img + (size_t)(cy - R) * w + ((cx - R) & ~3);
#pragma unroll
    for (int k = 0; k < kRLoads; k++) {
        const int idx = l32 + 32 * k;                    // dword idx of the patch: row idx / 9, column idx % 9
        const int r = (idx * 57) >> 9, c = idx - 9 * r;  // exact for idx < 460
        if (idx < total) pre[k] = *reinterpret_cast<const uint32_t *>(src + r * w + 4 * c);
    }
}

__global__ __launch_bounds__(kKT) void rbrief_lds_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                         const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
                                                         int kp_stride, const int32_t *__restrict__ table,
                                                         uint8_t *__restrict__ desc, int frames, int per_frame) {
    __shared__ uint32_t s_patch[2][kKT / 32][kRRows * (kRPitch / 4)];
    const int tid = threadIdx.x;
    int f, bx;
    vs_xcd_item_block(blockIdx.x, per_frame, f, bx);
    if (f >= frames) return;
    const int n = n_arr[f];
    const int kp0 = bx * (kRGroups * (kKT / 32));
    if (kp0 >= n) return;   // whole workgroup
    const int R = table[512];
    const int slot = tid >> 5, l32 = tid & 31;
    if (R > kRMax) {   // a pattern that rotates out of the staged patch (not the reference's angle): sample the image directly
        const int4 *t4 = reinterpret_cast<const int4 *>(table) + 4 * l32;
        for (int g = 0; g < kRGroups; g++) {
            const int kp = kp0 + g * (kKT / 32) + slot;
            if (kp >= n) break;
            const float2 p = reinterpret_cast<const float2 *>(xy)[(size_t)f * kp_stride + kp];
            const uint8_t *center = blurred + (size_t)f * w * h + (size_t)((int)rintf(p.y)) * w + (int)rintf(p.x);
            uint32_t val = 0;
            for (int q = 0; q < 4; q++) {
                const int4 e4 = t4[q];
                const int e[4] = {e4.x, e4.y, e4.z, e4.w};
                for (int u = 0; u < 4; u += 2) {
                    const int t0 = center[(e[u] >> 16) * w + (int)(int16_t)(e[u] & 0xFFFF)];
                    const int t1 = center[(e[u + 1] >> 16) * w + (int)(int16_t)(e[u + 1] & 0xFFFF)];
                    val |= (uint32_t)(t0 < t1) << (2 * q + u / 2);
                }
            }
            desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + l32] = (uint8_t)val;
        }
        return;
    }
    // this lane's 16 sample offsets (byte l32 of the descriptor: tests 8 l32 .. 8 l32 + 7, two sides each) in the staged patch
    int off[16];
    {
        const int4 *t4 = reinterpret_cast<const int4 *>(table) + 4 * l32;   // entries 16 l32 .. 16 l32 + 15
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 e4 = t4[q];
            const int e[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int u = 0; u < 4; u++) off[4 * q + u] = (e[u] >> 16) * kRPitch + (int)(int16_t)(e[u] & 0xFFFF);   // index 2 bit + side
        }
    }
    const uint8_t *img = blurred + (size_t)f * w * h;
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)f * kp_stride;
    auto keypoint = [&](int g) -> float2 {   // this slot's keypoint of group g (its 32 lanes read the same one)
        const int kp = kp0 + g * (kKT / 32) + slot;
        return (g < kRGroups && kp < n) ? P[kp] : make_float2(-1.f, -1.f);   // x < 0: no keypoint
    };
    const int total = (2 * R + 1) * (kRPitch / 4);
    uint32_t preA[kRLoads], preB[kRLoads];
    float2 pa = keypoint(0), pb2 = keypoint(1);
    int cxa = (int)rintf(pa.x), cya = (int)rintf(pa.y), cxb = (int)rintf(pb2.x), cyb = (int)rintf(pb2.y);
    bool la = pa.x >= 0.f, lb = pb2.x >= 0.f;
    rbrief_fetch(preA, img, w, cxa, cya, R, total, l32, la);
    rbrief_fetch(preB, img, w, cxb, cyb, R, total, l32, lb);
    float2 pn = keypoint(2);   // the keypoint whose patch is requested next

    auto stage = [&](const uint32_t (&pre)[kRLoads], int buf, bool live) {
        if (!live) return;
        uint32_t *dst = s_patch[buf][slot] + l32;
#pragma unroll
        for (int k = 0; k < kRLoads; k++)
            if (l32 + 32 * k < total) dst[32 * k] = pre[k];
    };
    auto sample = [&](int g, int buf, int cx, bool live) {
        if (!live) return;
        const int kp = kp0 + g * (kKT / 32) + slot;
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(s_patch[buf][slot]) + R * kRPitch + (cx - ((cx - R) & ~3));
        uint32_t val = 0;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const int t0 = pb[off[2 * bit]];
            const int t1 = pb[off[2 * bit + 1]];
            val |= (uint32_t)(t0 < t1) << bit;
        }
        desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + l32] = (uint8_t)val;
    };
#pragma unroll 1
    for (int g = 0; g < kRGroups; g += 2) {
        // group g: its patch is in set A
        stage(preA, 0, la);
        __syncthreads();
        {
            const int cx = cxa;
            const bool live = la;
            cxa = (int)rintf(pn.x);
            cya = (int)rintf(pn.y);
            la = pn.x >= 0.f;
            rbrief_fetch(preA, img, w, cxa, cya, R, total, l32, la);   // group g + 2
            pn = keypoint(g + 3);
            sample(g, 0, cx, live);
        }
        // group g + 1: set B
        stage(preB, 1, lb);
        __syncthreads();
        {
            const int cx = cxb;
            const bool live = lb;
            cxb = (int)rintf(pn.x);
            cyb = (int)rintf(pn.y);
            lb = pn.x >= 0.f;
            rbrief_fetch(preB, img, w, cxb, cyb, R, total, l32, lb);   // group g + 3
            pn = keypoint(g + 4);
            sample(g + 1, 1, cx, live);
        }
    }
}

// The descriptor with the IMAGE TILE staged instead of one patch per keypoint.  The per-keypoint staging above moves
// (2R + 1) rows of 36 bytes per keypoint through the L1s, 3.3 GB of cache lines for 1 M keypoints, in a chain of
// load -> LDS -> sample steps per workgroup; but a frame's 2000 patches overlap — every pixel is in two of them on
// average.  Here a workgroup owns one kTW x kTH tile of the image: it loads the tile plus a margin of kRMax pixels once
// (16-byte pieces, coalesced rows) and describes every keypoint that keypoint_border_kernel filed under the tile, eight
// at a time, from LDS; the list entry of the next keypoint is in flight while one is sampled.  Same samples, same
// comparisons, same bytes.  0.29 -> 0.215 ms at C3, and 0.19 once the samples were LDS reads (see the two loops at the end)
// and the staging loads were in flight together (128 x 128 tiles; 64 x 64: 0.33, 128 x 64: 0.23, 256 x 128: 0.19); what is
// left is the 16 byte reads per lane and keypoint from LDS at random positions in the patch (bank conflicts) and the
// workgroup's chain of staging, barrier and four or five trips of entry -> samples -> store.
__global__ __launch_bounds__(kKT) void rbrief_tile_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                          int kp_stride,
                                                          const int32_t *__restrict__ table, const int32_t *__restrict__ tile_start,
                                                          const int32_t *__restrict__ tile_kp, uint8_t *__restrict__ desc,
                                                          int frames, int tw, int th, int tiles_x, int ntiles) {
    extern __shared__ __align__(16) uint8_t s_tile_px[];   // (th + 2 kRMax) rows of pitch = tw + 2 kRMax + 16 bytes
    const int tid = threadIdx.x;
    int f, t;
    vs_xcd_item_block(blockIdx.x, ntiles, f, t);
    if (f >= frames) return;
    const int32_t *TS = tile_start + (size_t)f * (kTilesMax + 1);
    const int k0 = TS[t], k1 = TS[t + 1];
    if (k0 == k1) return;   // nothing to describe here
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    // staged region: 16-byte columns x0 .. x0 + 16 nc (x0 a multiple of 16), rows y0 .. y0 + nr, clipped to the image
    // (w % 16 == 0: a 16-byte piece never leaves its row)
    const int x0 = max(tx * tw - kRMax, 0) & ~15, x1 = min(tx * tw + tw + kRMax, w);
    const int y0 = max(ty * th - kRMax, 0), y1 = min(ty * th + th + kRMax, h);
    const int nc = (x1 - x0 + 15) >> 4, nr = y1 - y0;
    // a pattern whose rotated offsets reach beyond the margin (not the reference's angle, or a wider user pattern) is
    // sampled from the image itself: same code, the tile is simply not staged
    const bool staged = table[512] <= kRMax;
    const int pitch = staged ? tw + 2 * kRMax + 16 : w;   // bytes; a multiple of 16 when staged
    const uint8_t *img = blurred + (size_t)f * w * h;
    const uint32_t magic = (65536u + (uint32_t)nc - 1u) / (uint32_t)nc;   // i / nc == (i * magic) >> 16 for the i that occur (< 8000)
    // eight pieces per thread in flight (a 128 x 128 tile with its margins is 7.5 per thread): one piece per trip of a plain
    // loop is a load, a wait and a store, i.e. eight memory round trips in a row in front of the barrier
    constexpr int kStageDepth = 8;
    for (int i0 = tid; staged && i0 < nr * nc; i0 += kKT * kStageDepth) {
        uint4 v[kStageDepth];
#pragma unroll
        for (int u = 0; u < kStageDepth; u++) {
            const int i = min(i0 + kKT * u, nr * nc - 1);   // unconditional loads (a load under a condition is waited for at its join)
            const int r = (int)(((uint32_t)i * magic) >> 16), c = i - r * nc;
            v[u] = *reinterpret_cast<const uint4 *>(img + (size_t)(y0 + r) * w + x0 + 16 * c);
        }
        __builtin_amdgcn_sched_barrier(0);   // or the scheduler pairs every load with its store again
#pragma unroll
        for (int u = 0; u < kStageDepth; u++) {
            const int i = min(i0 + kKT * u, nr * nc - 1);   // the surplus writes the last piece again: no branches here either
            const int r = (int)(((uint32_t)i * magic) >> 16), c = i - r * nc;
            *reinterpret_cast<uint4 *>(s_tile_px + r * pitch + 16 * c) = v[u];
        }
    }
    const int slot = tid >> 5, l32 = tid & 31;
    int off[16];   // this lane's 16 sample offsets in the staged tile
    {
        const int4 *t4 = reinterpret_cast<const int4 *>(table) + 4 * l32;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 e4 = t4[q];
            const int e[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int u = 0; u < 4; u++) off[4 * q + u] = (e[u] >> 16) * pitch + (int)(int16_t)(e[u] & 0xFFFF);
        }
    }
    const int2 *TK = reinterpret_cast<const int2 *>(tile_kp) + (size_t)f * kp_stride;
    int2 ent = k0 + slot < k1 ? TK[k0 + slot] : make_int2(0, 0);   // requested before the tile is waited for
    __syncthreads();
    // Two copies of the loop, one per address space: with one pointer that is either into the tile or into the image the
    // samples are FLAT loads — a 64-bit address each, the slow path to LDS, and a wait that covers the list entry in flight
    // as well — instead of ds_read_u8 at a 32-bit address.
    auto describe = [&](auto sample_base) {
        for (int k = k0 + slot; k < k1; k += kKT / 32) {
            const int2 cur = ent;
            if (k + kKT / 32 < k1) ent = TK[k + kKT / 32];   // the next one is in flight while this one is sampled
            const int kx = cur.x & 0xFFFF, ky = cur.x >> 16, kp = cur.y;
            const auto pb = sample_base(kx, ky);
            uint32_t val = 0;
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const int t0 = pb[off[2 * bit]];
                const int t1 = pb[off[2 * bit + 1]];
                val |= (uint32_t)(t0 < t1) << bit;
            }
            desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + l32] = (uint8_t)val;
        }
    };
    if (staged)
        describe([&](int kx, int ky) { return s_tile_px + (ky - y0) * pitch + (kx - x0); });
    else
        describe([&](int kx, int ky) { return img + (size_t)ky * w + kx; });
}

}  // namespace

// The rotated pattern table depends on the pattern and the angle alone: a caller that has an idle stream ahead of the
// description stage (vslam_extract_features: the auxiliary stream, in front of the blur) launches it there and
// vs_launch_orb_describe skips its own launch.  The caller guarantees the ordering (it joins that stream before describing).
int vs_launch_rbrief_rotate(vslam_ctx *ctx, const int8_t *pattern, float ca, float sa) {
    int32_t *table = nullptr;
    int rc = vs_arena_get(ctx, "rbrief.table", sizeof(int32_t) * 513, (void **)&table);
    if (rc) return rc;
    rbrief_rotate_kernel<<<1, 512, 0, ctx->stream>>>(pattern, ca, sa, table);
    VS_HIP(ctx, hipGetLastError());
    ctx->rbrief_table_ready = true;
    return VSLAM_OK;
}

int vs_launch_orb_describe(vslam_ctx *ctx, const uint8_t *blurred, int frames, int w, int h,
                           const float *xy_in, const int32_t *n_in, int kp_stride, float ca, float sa,
                           const int8_t *pattern, float *xy_out, uint8_t *desc, int32_t *n_out) {
    VS_REQUIRE(ctx, blurred && xy_in && n_in && pattern && xy_out && desc && n_out, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && kp_stride > 0, VSLAM_ERR_INVALID);
    // the tile-staged descriptor kernel (VSLAM_RBRIEF_PATCH=1 selects round 2's per-keypoint staging, for A/B timing)
    static const bool patch_form = VS_EXPERIMENT_ENV("VSLAM_RBRIEF_PATCH") != nullptr;
    int tw = kTW, th = kTH;
    if (const char *e = VS_EXPERIMENT_ENV("VSLAM_RBRIEF_TILE")) sscanf(e, "%dx%d", &tw, &th);   // A/B timing
    // Rows of the blurred plane may be longer than the image is wide (vslam_ctx::img_pitch: a mirrored tail nobody samples,
    // keypoints keep 31 pixels from the image's border).  The kernels address and tile the plane by its rows, `wl`; only the
    // border rule below sees the image's width.
    const int w_img = w;
    w = vs_pitch(ctx, w);
    while (vs_div_up(w, tw) * vs_div_up(h, th) > kTilesMax) {
        tw *= 2;
        th *= 2;
    }
    const int tiles_x = vs_div_up(w, tw), ntiles = tiles_x * vs_div_up(h, th);
    const size_t tile_lds = (size_t)(th + 2 * kRMax) * (tw + 2 * kRMax + 16);
    // the tile lists pack a keypoint as x | y << 16 (unpacked with & 0xFFFF and an arithmetic >> 16): x < 65536, y < 32768
    const bool tiled = w % 16 == 0 && (reinterpret_cast<uintptr_t>(blurred) & 15) == 0 && tile_lds <= 64 * 1024 && !patch_form &&
                       w <= 65536 && h <= 32768;
    int32_t *tile_start = nullptr, *tile_kp = nullptr;
    if (tiled) {
        int rc;
        if ((rc = vs_arena_get(ctx, "rbrief.tile_start", sizeof(int32_t) * (size_t)frames * (kTilesMax + 1), (void **)&tile_start))) return rc;
        if ((rc = vs_arena_get(ctx, "rbrief.tile_kp", sizeof(int32_t) * 2 * (size_t)frames * kp_stride, (void **)&tile_kp))) return rc;
    }
    {
        VsProfScope ps(ctx, "keypoint_border_kernel");
        keypoint_border_kernel<<<frames, kKT, 0, ctx->stream>>>(xy_in, n_in, kp_stride, w_img, h, xy_out, n_out, tw, th, tiles_x, ntiles,
                                                                tile_start, tile_kp);
    }
    {
        VsProfScope ps(ctx, "rbrief_kernel");
        const int per_frame = vs_div_up(kp_stride, kKT / 32);
        if (w % 4 == 0 && (reinterpret_cast<uintptr_t>(blurred) & 3) == 0) {
            int32_t *table = nullptr;
            int rc = vs_arena_get(ctx, "rbrief.table", sizeof(int32_t) * 513, (void **)&table);
            if (rc) return rc;
            if (!ctx->rbrief_table_ready) rbrief_rotate_kernel<<<1, 512, 0, ctx->stream>>>(pattern, ca, sa, table);
            if (tiled) {
                rbrief_tile_kernel<<<vs_xcd_grid(frames, ntiles), kKT, tile_lds, ctx->stream>>>(
                    blurred, w, h, kp_stride, table, tile_start, tile_kp, desc, frames, tw, th, tiles_x, ntiles);
            } else {
                const int per_frame_lds = vs_div_up(kp_stride, kRGroups * (kKT / 32));
                rbrief_lds_kernel<<<vs_xcd_grid(frames, per_frame_lds), kKT, 0, ctx->stream>>>(
                    blurred, w, h, xy_out, n_out, kp_stride, table, desc, frames, per_frame_lds);
            }
        } else {
            rbrief_kernel<<<vs_xcd_grid(frames, per_frame), kKT, 0, ctx->stream>>>(blurred, w, h, xy_out, n_out, kp_stride, ca,
                                                                                    sa, pattern, desc, frames, per_frame);
        }
    }
    ctx->rbrief_table_ready = false;   // a caller's early launch (vs_launch_rbrief_rotate) covers one describe call
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}


// ORB::compute for provided keypoints on gfx950 (reference: src/Frame.cpp:57,64-72): the image-border filter and the
// steered BRIEF descriptor sampled from the blurred image.
#include "image_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// ORB::compute for provided keypoints: border filter (ordered) + steered BRIEF
// ------------------------------------------------------------------------------------------
constexpr int kKT = 256;

// KeyPointsFilter::runByImageBorder(kps, size, 31): keep pt inside [31, w-31) x [31, h-31), order kept
__global__ __launch_bounds__(kKT) void keypoint_border_kernel(const float *__restrict__ xy_in,
                                                              const int32_t *__restrict__ n_in, int kp_stride,
                                                              int w, int h, float *__restrict__ xy_out,
                                                              int32_t *__restrict__ n_out) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_cnt[kKT / 64];
    __shared__ int s_base;
    const float2 *I = reinterpret_cast<const float2 *>(xy_in) + (size_t)f * kp_stride;
    float2 *O = reinterpret_cast<float2 *>(xy_out) + (size_t)f * kp_stride;
    const int n = n_in[f];
    const int border = 31;
    const bool any = !(h <= border * 2 || w <= border * 2);
    if (tid == 0) s_base = 0;
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
        if (keep) O[off + (int)__popcll(bal & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int wv = 0; wv < kKT / 64; wv++) t += s_cnt[wv];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) n_out[f] = s_base;
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

__global__ __launch_bounds__(kKT) void rbrief_lds_kernel(const uint8_t *__restrict__ blurred, int w, int h,
                                                         const float *__restrict__ xy, const int32_t *__restrict__ n_arr,
                                                         int kp_stride, const int32_t *__restrict__ table,
                                                         uint8_t *__restrict__ desc, int frames, int per_frame) {
    __shared__ int s_off[16][32];   // [2 * bit + side][byte]: offset relative to the centre, in the staged patch (or the image)
    __shared__ uint32_t s_patch[2][kKT / 32][kRRows * (kRPitch / 4)];
    const int tid = threadIdx.x;
    int f, bx;
    vs_xcd_item_block(blockIdx.x, per_frame, f, bx);
    if (f >= frames) return;
    const int n = n_arr[f];
    const int kp0 = bx * (kRGroups * (kKT / 32));
    if (kp0 >= n) return;   // whole workgroup
    const int R = table[512];
    const bool staged = R <= kRMax;
    for (int i = tid; i < 512; i += kKT) {
        const int32_t e = table[i];
        const int ix = (int)(int16_t)(e & 0xFFFF), iy = e >> 16;
        const int t = i >> 1, byte = t >> 3, bit = t & 7;
        s_off[2 * bit + (i & 1)][byte] = staged ? iy * kRPitch + ix : iy * w + ix;
    }
    const int slot = tid >> 5, l32 = tid & 31;
    const uint8_t *img = blurred + (size_t)f * w * h;
    const float2 *P = reinterpret_cast<const float2 *>(xy) + (size_t)f * kp_stride;
    auto keypoint = [&](int g) -> float2 {   // this slot's keypoint of group g (its 32 lanes read the same one)
        const int kp = kp0 + g * (kKT / 32) + slot;
        return (g < kRGroups && kp < n) ? P[kp] : make_float2(-1.f, -1.f);   // x < 0: no keypoint
    };
    // dwords per staged row: columns (cx - R) & ~3 .. cx + R, at most 2R + 4 bytes; rows keep the fixed pitch
    const int nd = staged ? (2 * R + 3) / 4 + 1 : 1;
    const int total = (2 * R + 1) * nd;
    const int r0 = l32 / nd, c0 = l32 - r0 * nd, dq = 32 / nd, dr = 32 - dq * nd;   // dword l32 + 32k = row r, column c
    uint32_t pre[kRLoads];
    auto prefetch = [&](int cx, int cy, bool live) {
        if (!staged || !live) return;
        const uint8_t *src = img + (size_t)(cy - R) * w + ((cx - R) & ~3);
        int r = r0, c = c0;
#pragma unroll
        for (int k = 0; k < kRLoads; k++) {
            if (l32 + 32 * k < total) pre[k] = *reinterpret_cast<const uint32_t *>(src + (size_t)r * w + 4 * c);
            r += dq;
            c += dr;
            if (c >= nd) {
                c -= nd;
                r++;
            }
        }
    };
    // keypoints are read two groups ahead and patches one group ahead, so no load waits on another inside the loop
    float2 p0 = keypoint(0), p1 = keypoint(1);
    int cx = (int)rintf(p0.x), cy = (int)rintf(p0.y);
    bool live = p0.x >= 0.f;
    prefetch(cx, cy, live);
#pragma unroll 1   // one copy of the body: the fully unrolled form needs twice the registers and halves the occupancy
    for (int g = 0; g < kRGroups; g++) {
        if (staged && live) {
            int r = r0, c = c0;
#pragma unroll
            for (int k = 0; k < kRLoads; k++) {
                if (l32 + 32 * k < total) s_patch[g & 1][slot][r * (kRPitch / 4) + c] = pre[k];
                r += dq;
                c += dr;
                if (c >= nd) {
                    c -= nd;
                    r++;
                }
            }
        }
        __syncthreads();   // also orders the first s_off reads after their writes
        const float2 p2 = keypoint(g + 2);
        const int cx_n = (int)rintf(p1.x), cy_n = (int)rintf(p1.y);
        const bool live_n = p1.x >= 0.f;
        prefetch(cx_n, cy_n, live_n);
        if (live) {
            const int kp = kp0 + g * (kKT / 32) + slot;
            const uint8_t *pb = staged ? reinterpret_cast<const uint8_t *>(s_patch[g & 1][slot]) + R * kRPitch + (cx - ((cx - R) & ~3))
                                       : img + (size_t)cy * w + cx;
            uint32_t val = 0;
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const int t0 = pb[s_off[2 * bit][l32]];
                const int t1 = pb[s_off[2 * bit + 1][l32]];
                val |= (uint32_t)(t0 < t1) << bit;
            }
            desc[((size_t)f * kp_stride + kp) * VSLAM_DESC_BYTES + l32] = (uint8_t)val;
        }
        cx = cx_n;
        cy = cy_n;
        live = live_n;
        p1 = p2;
    }
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
    {
        VsProfScope ps(ctx, "keypoint_border_kernel");
        keypoint_border_kernel<<<frames, kKT, 0, ctx->stream>>>(xy_in, n_in, kp_stride, w, h, xy_out, n_out);
    }
    {
        VsProfScope ps(ctx, "rbrief_kernel");
        const int per_frame = vs_div_up(kp_stride, kKT / 32);
        if (w % 4 == 0 && (reinterpret_cast<uintptr_t>(blurred) & 3) == 0) {
            int32_t *table = nullptr;
            int rc = vs_arena_get(ctx, "rbrief.table", sizeof(int32_t) * 513, (void **)&table);
            if (rc) return rc;
            if (!ctx->rbrief_table_ready) rbrief_rotate_kernel<<<1, 512, 0, ctx->stream>>>(pattern, ca, sa, table);
            const int per_frame_lds = vs_div_up(kp_stride, kRGroups * (kKT / 32));
            rbrief_lds_kernel<<<vs_xcd_grid(frames, per_frame_lds), kKT, 0, ctx->stream>>>(
                blurred, w, h, xy_out, n_out, kp_stride, table, desc, frames, per_frame_lds);
        } else {
            rbrief_kernel<<<vs_xcd_grid(frames, per_frame), kKT, 0, ctx->stream>>>(blurred, w, h, xy_out, n_out, kp_stride, ca,
                                                                                    sa, pattern, desc, frames, per_frame);
        }
    }
    ctx->rbrief_table_ready = false;   // a caller's early launch (vs_launch_rbrief_rotate) covers one describe call
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}


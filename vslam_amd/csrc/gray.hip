// cv::cvtColor(BGR2GRAY) for gfx950 (reference: src/Frame.cpp:56).  HBM-bound: 3 bytes in, 1 byte out per pixel.
#include "image_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// cvtColor(BGR2GRAY), 8U: (b*3735 + g*19235 + r*9798 + 2^14) >> 15
// ------------------------------------------------------------------------------------------
constexpr int kGrayIter = 4;

__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t *__restrict__ bgr, int w, int h,
                                                       int stride, uint8_t *__restrict__ gray,
                                                       int aligned) {
    const int f = blockIdx.y;
    const int qpr = (w + 3) >> 2;   // 4-pixel groups per row
    // kGrayIter consecutive 256-group chunks per workgroup: a quarter of the workgroups to dispatch, and a lane's
    // loads of all its chunks are in flight together
#pragma unroll
    for (int it = 0; it < kGrayIter; it++) {
        const int q = (blockIdx.x * kGrayIter + it) * 256 + threadIdx.x;
        if (q >= qpr * h) return;
        const int y = q / qpr, x = (q - y * qpr) * 4;
        const uint8_t *src = bgr + ((size_t)f * h + y) * stride + 3 * x;
        uint8_t *dst = gray + ((size_t)f * h + y) * w + x;
        if (aligned && x + 3 < w) {
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
            const uint32_t a = s4[0], b = s4[1], c = s4[2];   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            const uint32_t g0 = gray_of(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF);
            const uint32_t g1 = gray_of(a >> 24, b & 0xFF, (b >> 8) & 0xFF);
            const uint32_t g2 = gray_of((b >> 16) & 0xFF, b >> 24, c & 0xFF);
            const uint32_t g3 = gray_of((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24);
            *reinterpret_cast<uint32_t *>(dst) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else {
            for (int i = 0; i < 4 && x + i < w; i++) dst[i] = (uint8_t)gray_of(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
        }
    }
}

// The same conversion into rows of `pitch` >= w + 3 bytes (pitch % 4 == 0): columns [w, pitch) hold the row's
// BORDER_REFLECT_101 continuation (column w + k = column w - 2 - k), which is what a filter reading past the last column
// would be handed.  For widths that are no multiple of 4: the dword kernels downstream run on the padded rows and what they
// produce for the columns below w is what they would produce for the image alone.  One lane = 4 output bytes.
__global__ __launch_bounds__(256) void bgr2gray_padded_kernel(const uint8_t *__restrict__ bgr, int w, int h, int stride,
                                                              uint8_t *__restrict__ gray, int pitch) {
    const int f = blockIdx.y;
    const int qpr = pitch >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * h) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const uint8_t *row = bgr + ((size_t)f * h + y) * stride;
    uint32_t out;
    if (x + 3 < w) {   // twelve bytes of four whole pixels, wherever they start
        uint32_t d[3];
        __builtin_memcpy(d, row + 3 * x, 12);
        const uint32_t a = d[0], b = d[1], c = d[2];   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        out = gray_of(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF) | (gray_of(a >> 24, b & 0xFF, (b >> 8) & 0xFF) << 8) |
              (gray_of((b >> 16) & 0xFF, b >> 24, c & 0xFF) << 16) | (gray_of((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24) << 24);
    } else {
        out = 0;
        for (int i = 0; i < 4; i++) {
            const int sx = reflect101(x + i, w);
            out |= gray_of(row[3 * sx], row[3 * sx + 1], row[3 * sx + 2]) << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t *>(gray + ((size_t)f * h + y) * pitch + x) = out;
}

// A caller's gray image (rows of w bytes) into rows of `pitch` bytes with the mirrored tail, and the first w bytes of such rows
// back into packed rows: what lets the stage-level entry points (vslam_good_features, vslam_gaussian7) use the dword kernels
// for widths that are no multiple of 4.  A lane = 4 bytes of a padded row.
__global__ __launch_bounds__(256) void gray_pad_kernel(const uint8_t *__restrict__ src, int w, int h, uint8_t *__restrict__ dst,
                                                       int pitch) {
    const int f = blockIdx.y;
    const int qpr = pitch >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * h) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const uint8_t *row = src + ((size_t)f * h + y) * w;
    uint32_t out;
    if (x + 3 < w) {
        __builtin_memcpy(&out, row + x, 4);
    } else {
        out = 0;
        for (int i = 0; i < 4; i++) out |= (uint32_t)row[reflect101(x + i, w)] << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(dst + ((size_t)f * h + y) * pitch + x) = out;
}

__global__ __launch_bounds__(256) void gray_unpad_kernel(const uint8_t *__restrict__ src, int w, int h, int pitch,
                                                         uint8_t *__restrict__ dst) {
    const int f = blockIdx.y;
    const int qpr = (w + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * h) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(src + ((size_t)f * h + y) * pitch + x);
    uint8_t *out = dst + ((size_t)f * h + y) * w + x;
    if (x + 3 < w) {
        __builtin_memcpy(out, &v, 4);
    } else {
        for (int i = 0; i < 4 && x + i < w; i++) out[i] = (uint8_t)(v >> (8 * i));
    }
}

}  // namespace

int vs_launch_gray_pad(vslam_ctx *ctx, const uint8_t *src, int frames, int w, int h, uint8_t *dst, int pitch) {
    VS_REQUIRE(ctx, src && dst && frames > 0 && pitch % 4 == 0 && pitch >= w + 3 && pitch - w < w - 1, VSLAM_ERR_INVALID);
    gray_pad_kernel<<<dim3(vs_div_up((pitch / 4) * h, 256), frames), 256, 0, ctx->stream>>>(src, w, h, dst, pitch);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_gray_unpad(vslam_ctx *ctx, const uint8_t *src, int frames, int w, int h, int pitch, uint8_t *dst) {
    VS_REQUIRE(ctx, src && dst && frames > 0 && pitch % 4 == 0 && pitch >= w, VSLAM_ERR_INVALID);
    gray_unpad_kernel<<<dim3(vs_div_up(((w + 3) / 4) * h, 256), frames), 256, 0, ctx->stream>>>(src, w, h, pitch, dst);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

int vs_launch_bgr2gray(vslam_ctx *ctx, const uint8_t *bgr, int frames, int w, int h, int stride,
                       uint8_t *gray) {
    VS_REQUIRE(ctx, bgr && gray, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, frames > 0 && w > 0 && h > 0 && stride >= 3 * w, VSLAM_ERR_INVALID);
    const int aligned = (stride % 4 == 0) && (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(bgr) & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(gray) & 3) == 0) && (((size_t)h * stride) % 4 == 0);
    VsProfScope ps(ctx, "bgr2gray_kernel");
    const int pitch = vs_pitch(ctx, w);
    if (pitch != w) {   // padded rows (vslam_ctx::img_pitch)
        VS_REQUIRE(ctx, pitch % 4 == 0 && pitch >= w + 3 && pitch - w < w - 1 && (reinterpret_cast<uintptr_t>(gray) & 3) == 0, VSLAM_ERR_INVALID);
        dim3 pgrid(vs_div_up((pitch / 4) * h, 256), frames);
        bgr2gray_padded_kernel<<<pgrid, 256, 0, ctx->stream>>>(bgr, w, h, stride, gray, pitch);
        VS_HIP(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    dim3 grid(vs_div_up(((w + 3) / 4) * h, 256 * kGrayIter), frames);
    bgr2gray_kernel<<<grid, 256, 0, ctx->stream>>>(bgr, w, h, stride, gray, aligned);
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}


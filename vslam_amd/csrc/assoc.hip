// Map association for gfx950.
//
// Replaces the loop at /root/reference/src/vslam.cpp:129-161 together with orb_distance
// (src/PointMap.cpp:36-46): project every map point with c2 = K [R|t], keep the ones that land in the
// image, radius_search(frame.kdtree, frame.points, q, 2), and give the map point the first hit that is
// still unassigned and whose minimum Hamming distance to the map point's stored observations is < 64.
//
// Split in two kernels:
//   assoc_candidates_kernel  one lane per map point: projection (double-accumulated like OpenCV's
//                            GEMM_2_T), tree walk in visit order, min-Hamming per hit; emits the
//                            ordered list of ACCEPTABLE hits (distance under threshold).
//   assoc_resolve_kernel     the reference loop is sequential (a claim hides the keypoint from later
//                            map points); one workgroup per item reaches the same assignment in a few
//                            parallel rounds (see the kernel's comment for why the result is identical).
#include "ctx.h"

namespace {

constexpr int kCandCap = 16;
constexpr int kAT = 256, kAStack = 18;

__device__ __forceinline__ uint32_t ham256(const uint4 *a, const uint4 *b) {
    const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    return __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) + __builtin_popcount(a0.z ^ b0.z) +
           __builtin_popcount(a0.w ^ b0.w) + __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
           __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
}

__global__ __launch_bounds__(kAT) void assoc_candidates_kernel(
    const float *__restrict__ map_points, const int32_t *__restrict__ n_map, int map_stride, const float *__restrict__ c2_all,
    int img_w, int img_h, const int32_t *__restrict__ nodes, const float *__restrict__ xy, const uint8_t *__restrict__ desc,
    const int32_t *__restrict__ n_kp, int kp_stride, const int32_t *__restrict__ obs_offsets,
    const uint8_t *__restrict__ obs_desc, int obs_stride, float radius, uint32_t dist_threshold,
    int32_t *__restrict__ cand, int32_t *__restrict__ cand_cnt, int32_t *__restrict__ errflag) {
    __shared__ uint32_t stack[kAStack * kAT];
    __shared__ int32_t stack_idx[kAStack * kAT];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * kAT + tid;
    if (i >= n_map[b]) return;
    const float4 P = reinterpret_cast<const float4 *>(map_points)[(size_t)b * map_stride + i];
    const float *c2 = c2_all + (size_t)b * 12;
    float pr[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {   // pm.points * c2.t(): exact double products, (s0+s1+s2+s3) then one rounding
        const double s0 = (double)P.x * (double)c2[r * 4 + 0], s1 = (double)P.y * (double)c2[r * 4 + 1];
        const double s2 = (double)P.z * (double)c2[r * 4 + 2], s3 = (double)P.w * (double)c2[r * 4 + 3];
        pr[r] = (float)(((s0 + s1) + s2) + s3);
    }
    const float qx = pr[0] / pr[2], qy = pr[1] / pr[2];
    int cnt = 0;
    int32_t *out = cand + ((size_t)b * map_stride + i) * kCandCap;
    if (qx >= 0 && qx < (float)img_w && qy >= 0 && qy < (float)img_h) {
        const int n = min(n_kp[b], kp_stride);
        const float2 *Pt = reinterpret_cast<const float2 *>(xy) + (size_t)b * kp_stride;
        const int32_t *T = nodes + (size_t)b * kp_stride;
        const uint8_t *D = desc + (size_t)b * kp_stride * VSLAM_DESC_BYTES;
        const int32_t *OO = obs_offsets + (size_t)b * (map_stride + 1);
        const uint8_t *OD = obs_desc + (size_t)b * obs_stride * VSLAM_DESC_BYTES;
        const int o0 = OO[i], o1 = OO[i + 1];
        const float radius_sq = radius * radius;
        // A visit costs one memory round trip, not two: the children's positions follow from (pos, len) alone, so their
        // point indices are fetched beside the node's own point and travel on the stack with the entries.
        int sp = 0;
        if (n > 0) {
            stack[tid] = 0u | ((uint32_t)n << 15);
            stack_idx[tid] = T[0];
            sp = 1;
        }
        while (sp > 0) {
            --sp;
            const uint32_t e = stack[sp * kAT + tid];
            const int idx = stack_idx[sp * kAT + tid];
            const int pos = (int)(e & 0x7FFFu), len = (int)((e >> 15) & 0x7FFFu), axis = (int)(e >> 30);
            const int nl = len / 2, nr = len - nl - 1;
            const int il = nl > 0 ? T[pos + 1] : 0, ir = nr > 0 ? T[pos + 1 + nl] : 0;
            const float2 pt = Pt[idx];
            const float split = (axis == 0 ? qx : qy) - (axis == 0 ? pt.x : pt.y);
            const uint32_t nax = (uint32_t)(1 - axis) << 30;
            const uint32_t le = (uint32_t)(pos + 1) | ((uint32_t)nl << 15) | nax;
            const uint32_t re = (uint32_t)(pos + 1 + nl) | ((uint32_t)nr << 15) | nax;
            const float abs_split = (split > 0) ? split : -split;
            const bool both = abs_split <= radius;
            if (both) {
                const float dx = qx - pt.x, dy = qy - pt.y;
                if (dx * dx + dy * dy < radius_sq) {
                    uint32_t mn = 0xFFFFFFFFu;   // orb_distance: min over the stored observations
                    for (int o = o0; o < o1; o++) {
                        const uint32_t cur = ham256(reinterpret_cast<const uint4 *>(D + (size_t)idx * 32),
                                                    reinterpret_cast<const uint4 *>(OD + (size_t)o * 32));
                        mn = cur < mn ? cur : mn;
                    }
                    if (mn < dist_threshold) {
                        if (cnt < kCandCap) out[cnt] = idx;
                        else atomicOr(errflag, 1);   // more acceptable hits than slots: reported at synchronize
                        cnt++;
                    }
                }
            }
            if ((both || !(split < 0)) && nr > 0) {   // left is visited first: it goes on the stack last
                stack[sp * kAT + tid] = re;
                stack_idx[sp * kAT + tid] = ir;
                sp++;
            }
            if ((both || split < 0) && nl > 0) {
                stack[sp * kAT + tid] = le;
                stack_idx[sp * kAT + tid] = il;
                sp++;
            }
        }
    }
    cand_cnt[(size_t)b * map_stride + i] = cnt < kCandCap ? cnt : kCandCap;
}

constexpr int kRT = 256;   // threads of assoc_resolve_kernel: one workgroup per item

// The reference loop is sequential: map point i takes the first acceptable hit no earlier map point holds.
// The same assignment falls out of rounds that are parallel inside. U = map points still undecided,
// taken = keypoints assigned before or by a decided claim, owner[k] = the lowest index in U that lists k
// among its free hits. A map point whose first free hit k has owner[k] == itself is decided: every
// lower-indexed map point that lists k is decided already and did not take it, and every hit it skipped
// is held by a lower index (a higher index can never be decided on a keypoint a lower undecided one
// lists). A map point with no free hit left is decided as unassociated, since `taken` only grows. The
// lowest index in U is always decided, so the rounds end; on real maps two or three rounds settle
// everything where the one-at-a-time walk needed one step per map point.
__global__ __launch_bounds__(kRT) void assoc_resolve_kernel(const int32_t *__restrict__ n_map, int map_stride,
                                                            const int32_t *__restrict__ n_kp, int kp_stride,
                                                            const int32_t *__restrict__ cand,
                                                            const int32_t *__restrict__ cand_cnt,
                                                            int32_t *__restrict__ map_point_ids, int32_t *__restrict__ claim) {
    extern __shared__ uint32_t lds[];
    const int words = (kp_stride + 31) / 32;
    uint32_t *taken = lds;            // one bit per keypoint
    int32_t *owner = reinterpret_cast<int32_t *>(lds + ((words + 1) & ~1));   // one entry per keypoint
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = min(n_map[b], map_stride), nk = min(n_kp[b], kp_stride);   // a count beyond its stride would reach into the next item
    int32_t *ids = map_point_ids + (size_t)b * kp_stride;
    int32_t *mine = claim + (size_t)b * map_stride;
    const int32_t *cnts = cand_cnt + (size_t)b * map_stride;
    const int4 *lists = reinterpret_cast<const int4 *>(cand + (size_t)b * map_stride * kCandCap);
    constexpr int32_t kUndecided = -2, kNobody = 0x7FFFFFFF;

    for (int base = 0; base < words * 32; base += kRT) {   // `if (frame.map_point_ids[idx] >= 0) continue`
        const int idx = base + tid;
        const unsigned long long set = __ballot(idx < nk && ids[idx] >= 0);
        if (lane == 0) {
            const int wd = (base + (tid & ~63)) >> 5;
            if (wd < words) taken[wd] = (uint32_t)set;
            if (wd + 1 < words) taken[wd + 1] = (uint32_t)(set >> 32);
        }
        if (idx < kp_stride) owner[idx] = kNobody;
    }
    for (int i = tid; i < n; i += kRT) mine[i] = cnts[i] > 0 ? kUndecided : -1;
    __syncthreads();

    for (;;) {
        for (int i = tid; i < n; i += kRT) {   // lowest undecided index per free keypoint
            if (mine[i] != kUndecided) continue;
            const int cnt = cnts[i];
            int4 q[kCandCap / 4];
#pragma unroll
            for (int k = 0; k < kCandCap / 4; k++) q[k] = lists[(size_t)i * (kCandCap / 4) + k];
            const int32_t *c = reinterpret_cast<const int32_t *>(q);
#pragma unroll
            for (int k = 0; k < kCandCap; k++) {
                if (k >= cnt) continue;
                const int idx = c[k];
                if (!(taken[idx >> 5] & (1u << (idx & 31)))) atomicMin(&owner[idx], i);
            }
        }
        __syncthreads();
        int undecided = 0;
        for (int i = tid; i < n; i += kRT) {
            if (mine[i] != kUndecided) continue;
            const int cnt = cnts[i];
            int4 q[kCandCap / 4];
#pragma unroll
            for (int k = 0; k < kCandCap / 4; k++) q[k] = lists[(size_t)i * (kCandCap / 4) + k];
            const int32_t *c = reinterpret_cast<const int32_t *>(q);
            int first = -1;
#pragma unroll
            for (int k = 0; k < kCandCap; k++) {
                if (k >= cnt || first >= 0) continue;
                const int idx = c[k];
                if (!(taken[idx >> 5] & (1u << (idx & 31)))) first = idx;
            }
            if (first < 0) {
                mine[i] = -1;
            } else if (owner[first] == i) {
                atomicOr(&taken[first >> 5], 1u << (first & 31));
                mine[i] = first;
                ids[first] = i;   // frame.map_point_ids[idx] = i, src/vslam.cpp:154
            } else {
                undecided = 1;
            }
        }
        if (!__syncthreads_or(undecided)) break;
        for (int idx = tid; idx < kp_stride; idx += kRT) owner[idx] = kNobody;   // owners are per round
        __syncthreads();
    }
}

}  // namespace

int vs_launch_associate(vslam_ctx *ctx, const float *map_points, const int32_t *n_map, int batch, int map_stride,
                        const float *c2, int img_w, int img_h, const int32_t *nodes, const float *xy, const uint8_t *desc,
                        const int32_t *n_kp, int kp_stride, const int32_t *obs_offsets, const uint8_t *obs_desc,
                        int obs_stride, float radius, uint32_t dist_threshold, int32_t *map_point_ids, int32_t *claim) {
    VS_REQUIRE(ctx, map_points && n_map && c2 && nodes && xy && desc && n_kp && obs_offsets && obs_desc && map_point_ids && claim,
               VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, batch > 0 && map_stride > 0 && kp_stride > 0 && obs_stride > 0, VSLAM_ERR_INVALID);
    VS_REQUIRE(ctx, kp_stride <= VSLAM_MAX_KP, VSLAM_ERR_CAPACITY);
    int32_t *cand = nullptr, *cand_cnt = nullptr, *errflag = nullptr;
    int rc;
    if ((rc = vs_arena_get(ctx, "assoc.cand", sizeof(int32_t) * (size_t)batch * map_stride * kCandCap, (void **)&cand))) return rc;
    if ((rc = vs_arena_get(ctx, "assoc.cnt", sizeof(int32_t) * (size_t)batch * map_stride, (void **)&cand_cnt))) return rc;
    if ((rc = vs_device_errflag(ctx, &errflag))) return rc;
    {
        VsProfScope ps(ctx, "assoc_candidates_kernel");
        assoc_candidates_kernel<<<dim3(vs_div_up(map_stride, kAT), batch), kAT, 0, ctx->stream>>>(
            map_points, n_map, map_stride, c2, img_w, img_h, nodes, xy, desc, n_kp, kp_stride, obs_offsets, obs_desc, obs_stride,
            radius, dist_threshold, cand, cand_cnt, errflag);
    }
    {
        VsProfScope ps(ctx, "assoc_resolve_kernel");
        const size_t words = (size_t)((kp_stride + 31) / 32);
        const size_t lds = sizeof(uint32_t) * (((words + 1) & ~(size_t)1) + (size_t)kp_stride);
        assoc_resolve_kernel<<<batch, kRT, lds, ctx->stream>>>(n_map, map_stride, n_kp, kp_stride, cand, cand_cnt, map_point_ids,
                                                              claim);
    }
    VS_HIP(ctx, hipGetLastError());
    return VSLAM_OK;
}

// placeholder until the extraction kernels land (next commit)
#include "ctx.h"
int vs_launch_bgr2gray(vslam_ctx *ctx, const uint8_t *, int, int, int, int, uint8_t *) { ctx->err = "extract: not built"; return VSLAM_ERR_INVALID; }
int vs_launch_min_eigen(vslam_ctx *ctx, const uint8_t *, int, int, int, float *, uint32_t *) { ctx->err = "extract: not built"; return VSLAM_ERR_INVALID; }
int vs_launch_good_features(vslam_ctx *ctx, const uint8_t *, int, int, int, int, double, double, int, float *, int32_t *) { ctx->err = "extract: not built"; return VSLAM_ERR_INVALID; }
int vs_launch_gaussian7(vslam_ctx *ctx, const uint8_t *, int, int, int, uint8_t *) { ctx->err = "extract: not built"; return VSLAM_ERR_INVALID; }
int vs_launch_orb_describe(vslam_ctx *ctx, const uint8_t *, int, int, int, const float *, const int32_t *, int, float, float, const int8_t *, float *, uint8_t *, int32_t *) { ctx->err = "extract: not built"; return VSLAM_ERR_INVALID; }

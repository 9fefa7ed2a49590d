// Shared between the host-side translation units (adapters.cpp, ingest.cpp); not installed.
#pragma once
#include <vector>

#include "../../include/vslam/vslam_internal.h"
#include "../../include/vslam_amd.h"

namespace vslam {
namespace detail {
vslam_ctx *context();                          // the process-wide device context; throws if there is no device
void check(int rc, const char *what);          // throws std::runtime_error with vslam_last_error on rc != VSLAM_OK
const std::vector<s8> &brief_pattern();        // the 256 x 4 int8 test pattern in use (see vslam::Settings)
void fill_extract_params(vslam_extract_params &p, int max_corners, const int8_t *d_pattern);
}  // namespace detail
}  // namespace vslam

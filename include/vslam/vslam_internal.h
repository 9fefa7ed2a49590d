// Scalar aliases the reference's class surfaces are written in (reference: include/vslam_internal.h:9-27).
#pragma once
#include <cfloat>
#include <climits>
#include <cstddef>
#include <cstdint>

using s8 = std::int8_t;
using s16 = std::int16_t;
using s32 = std::int32_t;
using s64 = std::int64_t;
using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using usize = std::size_t;
using f32 = float;
using f64 = double;

#ifndef u32_max
#define u32_max ((u32)-1)
#endif
#ifndef f32_maximum
#define f32_maximum FLT_MAX
#endif

// the reference's three names for `static` (include/vslam_internal.h:29-32; defined there, used nowhere in its tree): kept so
// that this header can stand in for that one whatever a consumer does with them
#ifndef internal_function
#define internal_function static
#endif
#ifndef local_persist
#define local_persist static
#endif
#ifndef global_variable
#define global_variable static
#endif

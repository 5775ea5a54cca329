// hagrid/common.h -- scalar helpers shared by host code and the gfx950 kernels.
//
// API mirror of the reference's src/common.h (names, argument meaning and results are the same so
// that code written against the reference compiles unchanged); the implementation is new.  The
// CUDA-only parts of the reference header (DEBUG_SYNC, CHECK_CUDA_CALL, set_global: common.h:95-126)
// have no counterpart here: the gfx950 kernels take their parameters as kernel arguments, and
// runtime errors are reported through the C ABI (include/hagrid_amd.h).
//
// HOST / DEVICE are supplied by the build: empty for plain C++ translation units (the way the
// reference compiles main.cpp, src/CMakeLists.txt:42), __host__ / __device__ for hipcc.
#ifndef HAGRID_COMMON_H
#define HAGRID_COMMON_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>

#ifndef HOST
#define HOST
#endif
#ifndef DEVICE
#define DEVICE
#endif

namespace hagrid {

/// Milliseconds elapsed on the device while f runs (reference: common.h:15, profile.cu:5-18).
/// Defined inline in hagrid/mem_manager.h on top of hagrid_profile_begin/end.
inline float profile(std::function<void()> f);

/// Smallest q with q * j >= i (reference: common.h:18-20).
HOST DEVICE inline int round_div(int i, int j) { return (i + j - 1) / j; }

// The comparison forms matter for NaN propagation, so they are kept: min is "a < b ? a : b".
template <typename T> HOST DEVICE inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> HOST DEVICE inline T max(T a, T b) { return a > b ? a : b; }
template <typename T> HOST DEVICE inline T clamp(T v, T lo, T hi) { return min(hi, max(lo, v)); }
template <typename T> HOST DEVICE inline void swap(T& a, T& b) { T t = a; a = b; b = t; }

/// Bit cast (reference: common.h:31-37).
template <typename U, typename T>
HOST DEVICE inline U as(T t) {
    static_assert(sizeof(U) == sizeof(T), "as<> needs equally sized types");
    U u;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_memcpy(&u, &t, sizeof(U));
#else
    std::memcpy(&u, &t, sizeof(U));
#endif
    return u;
}

/// 1/x, or an infinity carrying the sign of x when x is +-0 (reference: common.h:40-42).
HOST DEVICE inline float safe_rcp(float x) {
    if (x != 0.0f) return 1.0f / x;
    return as<float>(0x7f800000u | (as<uint32_t>(x) & 0x80000000u));
}

/// x with its sign flipped when y is negative (reference: common.h:45-47).
HOST DEVICE inline float prodsign(float x, float y) {
    return as<float>(as<uint32_t>(x) ^ (as<uint32_t>(y) & 0x80000000u));
}

/// Number of bits needed for t (0 for t <= 1): the radix-sort key width of the reference
/// (common.h:81-93, used at build.cu:691).  Identical values, computed with clz.
template <typename T>
HOST DEVICE inline int ilog2(T t) {
    unsigned long long v = (unsigned long long)t;
    if (sizeof(T) < 8) v &= (1ull << (sizeof(T) * 8 % 64)) - 1ull;
    if (v <= 1) return 0;
    return 64 - __builtin_clzll(v);
}

/// Deterministic cube root: the same IEEE operation sequence on host and device, so both sides agree
/// on integer grid dimensions (the reference calls cbrtf: grid.h:99, evaluated by libm on the host
/// and by the CUDA math library on the device).  Correctly rounded for all tested inputs.
HOST DEVICE inline float det_cbrtf(float v) {
    if (v == 0.0f || v != v) return v;
    double x = v < 0 ? -(double)v : (double)v;
    if (x > 1.7e308) return v;
    uint64_t i = as<uint64_t>(x);
    i = i / 3 + 0x2A9F7893782DA1CEull;
    double y = as<double>(i);
    for (int k = 0; k < 6; k++) {
        double y2 = y * y;
        y = y - (y2 * y - x) / (3.0 * y2);
    }
    float r = (float)y;
    return v < 0 ? -r : r;
}

} // namespace hagrid

#endif // HAGRID_COMMON_H

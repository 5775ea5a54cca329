// scan_kat.hip -- known-answer hook for the device-wide scans of wave_prims.h (tests only).
//
// The scans replace cub::DeviceScan::ExclusiveSum of the reference (parallel.cuh:31-42); the construction passes only ever
// exercise them indirectly, so tests/test_scan_gpu.py drives them directly: tile-boundary sizes, the device carry chain and
// the two-word (Int2) publish of the look-back form.
#define HG_SCAN_THREE_KERNELS              // the reduce / spine / apply form next to the look-back form: instantiated here only
#include "../ctx.h"
#include "../wave_prims.h"
#include "hagrid_amd_kat.h"

using namespace hagrid_impl;

namespace {
struct IntIn { const int* v; __device__ int operator()(int i) const { return v[i]; } };
struct IntOut { int* v; __device__ void operator()(int i, int s) const { v[i] = s; } };
struct PairIn { const int* v; __device__ Int2 operator()(int i) const { return Int2{v[2 * size_t(i)], v[2 * size_t(i) + 1]}; } };
struct PairOut { int* v; __device__ void operator()(int i, Int2 s) const { v[2 * size_t(i)] = s.a; v[2 * size_t(i) + 1] = s.b; } };
} // namespace

extern "C" int hagrid_kat_scan(hagrid_ctx* ctx, const int32_t* values, int n, int words, const int32_t* carry_in, int lookback,
                               int32_t* out, int32_t* total) {
    if (!ctx || n < 0 || (words != 1 && words != 2) || (n > 0 && (!values || !out))) return HAGRID_EINVAL;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    PoolTemps tmp(ctx);
    const size_t bytes = size_t(n) * size_t(words) * sizeof(int);
    int* d_in = tmp.get<int>(size_t(n) * words + 2);
    const bool in_place = (lookback & 4) != 0;             // the scan overwrites its input (merge.hip tile sums, ray_order.hip bin table, trav_image.hip sizes do)
    lookback &= 3;
    int* d_out = in_place ? d_in : tmp.get<int>(size_t(n) * words + 2);
    int* partials = tmp.get<int>(size_t(words) * (size_t(scan_num_tiles(n)) + 1));
    int* scalars = tmp.get<int>(8);                          // [0..1] carry in, [2..3] total out
    if (!d_in || !d_out || !partials || !scalars) return HAGRID_ENOMEM;
    if (bytes) HG_TRY(hagrid_mem_copy_h2d(ctx, d_in, values, bytes));
    int zero[4] = {0, 0, 0, 0};
    if (carry_in) { zero[0] = carry_in[0]; if (words == 2) zero[1] = carry_in[1]; }
    HG_TRY(hagrid_mem_copy_h2d(ctx, scalars, zero, sizeof(zero)));
    const int saved = ctx->opt_lookback;
    ctx->opt_lookback = lookback == 2 ? 2 : (lookback ? 1 : 0);      // 2: the helping path at the first miss
    bool ok;
    if (words == 1) ok = ctx_scan<int>(ctx, IntIn{d_in}, IntOut{d_out}, n, partials, carry_in ? scalars : (const int*)nullptr, scalars + 2);
    else ok = ctx_scan<Int2>(ctx, PairIn{d_in}, PairOut{d_out}, n, reinterpret_cast<Int2*>(partials),
                             carry_in ? reinterpret_cast<const Int2*>(scalars) : (const Int2*)nullptr, reinterpret_cast<Int2*>(scalars + 2));
    ctx->opt_lookback = saved;
    if (!ok) return HAGRID_ENOMEM;
    HG_HIP(ctx, hipGetLastError());
    if (bytes) HG_TRY(hagrid_mem_copy_d2h(ctx, out, d_out, bytes));
    if (total) HG_TRY(hagrid_mem_copy_d2h(ctx, total, scalars + 2, size_t(words) * sizeof(int)));
    return HAGRID_OK;
}

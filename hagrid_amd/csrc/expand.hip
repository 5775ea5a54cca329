// expand.hip -- hagrid_expand_grid: grow each cell's integer box over face neighbours whose reference set
// is a subset of its own, on gfx950.
//
// Replaces the reference's expand.cu: expand (:199-223), expansion_iter (:184-197), overlap_step<axis> (:145-182),
// find_overlap (:60-143), is_subset (:21-36), with subset_only = true (the reference's compiled setting, :159).
// Bit-identical to the CPU oracle.  Cells that are not processed in a step are copied through to the new
// buffer (the reference leaves them stale, expand.cu:154-155,181 -- DESIGN.md D2).
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct ExpandK { ivec3 dims; ivec3 top; int shift; };   // expand.cu:5-9 (only what subset_only needs)
struct CellRec { ivec3 lo; int begin; ivec3 hi; int end; };

__device__ __forceinline__ CellRec load_cell(const Cell* cells, int i) {
    const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
    const int4 a = p[0], b = p[1];
    CellRec c; c.lo = ivec3(a.x, a.y, a.z); c.begin = a.w; c.hi = ivec3(b.x, b.y, b.z); c.end = b.w;
    return c;
}
__device__ __forceinline__ int comp(const ivec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// expand.cu:21-36
__device__ __forceinline__ bool is_subset(const int* __restrict__ p0, int c0, const int* __restrict__ p1, int c1) {
    if (c1 > c0) return false;
    if (c1 == 0) return true;
    int i = 0, j = 0;
    do {
        const int a = p0[i], b = p1[j];
        if (b < a) return false;
        j += (a == b);
        i++;
    } while ((i < c0) & (j < c1));
    return j == c1;
}

// find_overlap (expand.cu:60-143)
template <int axis, bool dir>
__device__ __forceinline__ int find_overlap(const ExpandK& k, const Entry* __restrict__ entries, const int* __restrict__ refs,
                                            const Cell* __restrict__ cells, const CellRec& cell, bool& continue_overlap) {
    constexpr int axis1 = (axis + 1) % 3, axis2 = (axis + 2) % 3;
    if (dir) { if (!(comp(cell.hi, axis) < comp(k.dims, axis))) return 0; }     // overlap_possible, expand.cu:12-18
    else     { if (!(comp(cell.lo, axis) > 0)) return 0; }
    int d = dir ? comp(k.dims, axis) : -comp(k.dims, axis);
    int k2 = comp(k.dims, axis2);
    int i = comp(cell.lo, axis1), j = comp(cell.lo, axis2);
    int max_d = d;
    const int a = dir ? comp(cell.hi, axis) : comp(cell.lo, axis) - 1;
    for (;;) {
        const ivec3 np = axis == 0 ? ivec3(a, i, j) : (axis == 1 ? ivec3(j, a, i) : ivec3(i, j, a));
        const CellRec next = load_cell(cells, int(lookup_entry(entries, k.shift, k.top, np)));
        max_d = dir ? min(max_d, comp(next.hi, axis) - comp(cell.hi, axis)) : max(max_d, comp(next.lo, axis) - comp(cell.lo, axis));
        d = dir ? min(d, max_d) : max(d, max_d);
        if (!is_subset(refs + cell.begin, cell.end - cell.begin, refs + next.begin, next.end - next.begin)) { d = 0; break; }
        const int k1 = comp(next.hi, axis1) - i;
        k2 = min(k2, comp(next.hi, axis2) - j);
        i += k1;
        if (i >= comp(cell.hi, axis1)) {
            i = comp(cell.lo, axis1);
            j += k2;
            k2 = comp(k.dims, axis2);
            if (j >= comp(cell.hi, axis2)) break;
        }
    }
    continue_overlap |= d == max_d;
    return d;
}

// overlap_step (expand.cu:145-182)
template <int axis>
__global__ void __launch_bounds__(kBlock) overlap_step(ExpandK k, const Entry* __restrict__ entries, const int* __restrict__ refs,
                                                       const Cell* __restrict__ cells, Cell* __restrict__ new_cells,
                                                       int* __restrict__ cell_flags, int num_cells) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= num_cells) return;
    const int flags = cell_flags[id];
    int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
    if ((flags & (1 << axis)) == 0) {      // copy through
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(id);
        const int4 a = p[0], b = p[1];
        out[0] = a; out[1] = b;
        return;
    }
    CellRec cell = load_cell(cells, id);
    bool flag = false;
    const int ov1 = find_overlap<axis, false>(k, entries, refs, cells, cell, flag);
    const int ov2 = find_overlap<axis, true>(k, entries, refs, cells, cell, flag);
    if (axis == 0) { cell.lo.x += ov1; cell.hi.x += ov2; }
    if (axis == 1) { cell.lo.y += ov1; cell.hi.y += ov2; }
    if (axis == 2) { cell.lo.z += ov1; cell.hi.z += ov2; }
    cell_flags[id] = (flag ? 1 << axis : 0) | (flags & ~(1 << axis));
    out[0] = make_int4(cell.lo.x, cell.lo.y, cell.lo.z, cell.begin);
    out[1] = make_int4(cell.hi.x, cell.hi.y, cell.hi.z, cell.end);
}

} // namespace

extern "C" int hagrid_expand_grid(hagrid_ctx* ctx, hagrid_grid* grid, const void* tris, int iters) {
    (void)tris;                                 // only the subset_only = false variant reads primitives
    if (!ctx || !grid) return HAGRID_EINVAL;
    if (iters <= 0) return HAGRID_OK;
    if (!grid->cells || !grid->entries || !grid->ref_ids) HG_FAIL(ctx, HAGRID_EINVAL, "expand_grid: incomplete (or compressed) grid");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ExpandK k;
    k.top = ivec3(grid->dims[0], grid->dims[1], grid->dims[2]);
    k.dims = k.top << grid->shift;
    k.shift = grid->shift;
    const int n = grid->num_cells;
    Cell* cells = static_cast<Cell*>(grid->cells);
    Cell* other = pool_alloc<Cell>(ctx, size_t(n));
    int* flags = pool_alloc<int>(ctx, size_t(n));
    if (!other || !flags) { hagrid_mem_free(ctx, other); hagrid_mem_free(ctx, flags); return HAGRID_ENOMEM; }
    HG_HIP(ctx, hipMemsetAsync(flags, 0xFF, size_t(n) * sizeof(int), st));              // expand.cu:206
    const Entry* entries = static_cast<const Entry*>(grid->entries);
    const int* refs = static_cast<const int*>(grid->ref_ids);
    const int blocks = grid_blocks(n, kBlock);
    for (int it = 0; it < iters; it++) {                                               // expansion_iter, expand.cu:184-197
        overlap_step<0><<<blocks, kBlock, 0, st>>>(k, entries, refs, cells, other, flags, n); std::swap(cells, other);
        overlap_step<1><<<blocks, kBlock, 0, st>>>(k, entries, refs, cells, other, flags, n); std::swap(cells, other);
        overlap_step<2><<<blocks, kBlock, 0, st>>>(k, entries, refs, cells, other, flags, n); std::swap(cells, other);
    }
    hipError_t e = hipGetLastError();
    HG_HIP(ctx, hipStreamSynchronize(st));
    hagrid_mem_free(ctx, flags);
    hagrid_mem_free(ctx, other);
    grid->cells = cells;
    if (e != hipSuccess) HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e));
    return HAGRID_OK;
}

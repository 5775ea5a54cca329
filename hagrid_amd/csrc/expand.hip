// expand.hip -- hagrid_expand_grid: grow each cell's integer box over face neighbours whose reference set
// is a subset of its own, on gfx950.
//
// Replaces the reference's expand.cu: expand (:199-223), expansion_iter (:184-197), overlap_step<axis> (:145-182),
// find_overlap (:60-143), is_subset (:21-36), compute_overlap (:39-57).  subset_only = true is the reference's compiled setting
// (:159) and the default; the precise mode (subset_only = false) is selected with hagrid_set_option("expand.subset_only", 0).
// Bit-identical to the CPU oracle.  Cells that are not processed in a step are copied through to the new
// buffer (the reference leaves them stale, expand.cu:154-155,181 -- DESIGN.md D2).
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"
#include "hagrid/prims.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct ExpandK { ivec3 dims; ivec3 top; int shift; vec3 gmin, cell_size, grid_inv; };   // expand.cu:5-9
constexpr int kChanged = 1 << 8;     // cell_flags: the cell's box changed in the previous pass (bits 0-2: expand.cu:145-182)
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

// compute_overlap (expand.cu:39-57): how far the cell may grow along `axis` before it would have to reference `prim`
template <int axis, bool dir>
__device__ __forceinline__ int compute_overlap(const ExpandK& k, const Tri& prim, const CellRec& cell, const BBox& cb, int d) {
    constexpr int axis1 = (axis + 1) % 3, axis2 = (axis + 2) % 3;
    const BBox pb = prim.bbox();
    if (get<axis1>(pb.min) <= get<axis1>(cb.max) && get<axis1>(pb.max) >= get<axis1>(cb.min) &&
        get<axis2>(pb.min) <= get<axis2>(cb.max) && get<axis2>(pb.max) >= get<axis2>(cb.min)) {
        const int prim_d = int(((dir ? get<axis>(pb.min) : get<axis>(pb.max)) - get<axis>(k.gmin)) * get<axis>(k.grid_inv));
        d = dir ? min(d, prim_d - comp(cell.hi, axis)) : max(d, prim_d - comp(cell.lo, axis) + 1);
        d = dir ? max(d, 0) : min(d, 0);
    }
    return d;
}

__device__ __forceinline__ Tri load_tri(const float4* __restrict__ tris, int i) {
    const float4* p = tris + 3 * size_t(i);
    const float4 a = p[0], b = p[1], c = p[2];
    return Tri(vec3(a.x, a.y, a.z), a.w, vec3(b.x, b.y, b.z), b.w, vec3(c.x, c.y, c.z), c.w);
}

// find_overlap (expand.cu:60-143)
template <int axis, bool dir, bool SUBSET_ONLY>
__device__ __forceinline__ int find_overlap(const ExpandK& k, const Entry* __restrict__ entries, const int* __restrict__ refs, const float4* __restrict__ tris,
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
        if (SUBSET_ONLY) {
            if (!is_subset(refs + cell.begin, cell.end - cell.begin, refs + next.begin, next.end - next.begin)) { d = 0; break; }
        } else {
            // expand.cu:96-127: references of the neighbour that the cell does not hold limit the growth
            if (next.begin < next.end) {
                const BBox cb(k.gmin + k.cell_size * vec3(cell.lo), k.gmin + k.cell_size * vec3(cell.hi));
                int p1 = cell.begin, p2 = next.begin;
                int ref2 = refs[p2];
                for (;;) {
                    while (p1 < cell.end) {
                        const int ref1 = refs[p1];
                        if (ref1 > ref2) break;
                        if (ref1 == ref2) {
                            if (++p2 >= next.end) break;
                            ref2 = refs[p2];
                        }
                        p1++;
                    }
                    if (p2 >= next.end) break;
                    d = compute_overlap<axis, dir>(k, load_tri(tris, ref2), cell, cb, d);
                    if (d == 0 || ++p2 >= next.end) break;
                    ref2 = refs[p2];
                }
            }
            if (d == 0) break;
        }
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

template <int axis, bool SUBSET_ONLY>
__device__ __forceinline__ void grow_cell(const ExpandK& k, const Entry* __restrict__ entries, const int* __restrict__ refs, const float4* __restrict__ tris,
                                          const Cell* __restrict__ cells, Cell* __restrict__ new_cells, int* __restrict__ cell_flags, int id, int flags) {
    CellRec cell = load_cell(cells, id);
    bool flag = false;
    const int ov1 = find_overlap<axis, false, SUBSET_ONLY>(k, entries, refs, tris, cells, cell, flag);
    const int ov2 = find_overlap<axis, true, SUBSET_ONLY>(k, entries, refs, tris, cells, cell, flag);
    if (axis == 0) { cell.lo.x += ov1; cell.hi.x += ov2; }
    if (axis == 1) { cell.lo.y += ov1; cell.hi.y += ov2; }
    if (axis == 2) { cell.lo.z += ov1; cell.hi.z += ov2; }
    cell_flags[id] = (flag ? 1 << axis : 0) | (flags & ~((1 << axis) | kChanged)) | ((ov1 | ov2) ? kChanged : 0);
    int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
    out[0] = make_int4(cell.lo.x, cell.lo.y, cell.lo.z, cell.begin);
    out[1] = make_int4(cell.hi.x, cell.hi.y, cell.hi.z, cell.end);
}

// overlap_step (expand.cu:145-182)
template <int axis, bool SUBSET_ONLY>
__global__ void __launch_bounds__(kBlock) overlap_step(ExpandK k, const Entry* __restrict__ entries, const int* __restrict__ refs,
                                                       const float4* __restrict__ tris, const Cell* __restrict__ cells, Cell* __restrict__ new_cells,
                                                       int* __restrict__ cell_flags, int num_cells) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= num_cells) return;
    const int flags = cell_flags[id];
    if ((flags & (1 << axis)) == 0) {      // copy through
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(id);
        int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
        const int4 a = p[0], b = p[1];
        out[0] = a; out[1] = b;
        if (flags & kChanged) cell_flags[id] = flags & ~kChanged;
        return;
    }
    grow_cell<axis, SUBSET_ONLY>(k, entries, refs, tris, cells, new_cells, cell_flags, id, flags);
}

// ---- passes of the later iterations: dense over the cells that are still growing ------------------------------------
// After the first iteration only a part of the cells is still flagged for an axis (1M-triangle soup: 51 / 39 / 32 % in the
// second iteration, 12 / 11 / 9 % in the third), scattered so evenly that almost every wavefront of overlap_step holds at
// least one of them: the all-cells kernel then runs at the latency of the face walk whatever the fraction (137-145 us for
// the 10 % passes against 164-225 us for the full ones), and the rest is a 270 MB copy-through.  Instead:
//   expand_select   lists the flagged cells (tile-wise compaction in LDS, one atomic per 8192 cells) and copies through only
//                   the unflagged cells that CHANGED in the previous pass -- the output buffer is the one of two passes ago,
//                   so every other unflagged cell already holds its current value there;
//   expand_listed   one thread per listed cell, same arithmetic as overlap_step.
// Results are the ones of overlap_step bit for bit (same input snapshot per pass, every cell written by one thread).
constexpr int kSelectItems = 16;                      // cells per thread
constexpr int kSelectTile = kBlock * kSelectItems;    // cells per workgroup

template <int axis>
__global__ void __launch_bounds__(kBlock) expand_select(const Cell* __restrict__ cells, Cell* __restrict__ new_cells, int* __restrict__ cell_flags,
                                                        int num_cells, int* __restrict__ list, int* __restrict__ count) {
    __shared__ int tile_list[kSelectTile];
    __shared__ int tile_count, tile_base;
    if (threadIdx.x == 0) tile_count = 0;
    __syncthreads();
    const int base = blockIdx.x * kSelectTile;
    int fl[kSelectItems];
    #pragma unroll
    for (int j = 0; j < kSelectItems; j++) {                     // all loads in flight before anything depends on them
        const int id = base + j * kBlock + threadIdx.x;
        fl[j] = id < num_cells ? cell_flags[id] : 0;
    }
    #pragma unroll
    for (int j = 0; j < kSelectItems; j++) {
        const int id = base + j * kBlock + threadIdx.x;
        const int flags = fl[j];
        const bool listed = (flags & (1 << axis)) != 0;
        const unsigned long long m = __ballot(listed);            // one LDS atomic per wavefront
        int wbase = 0;
        if (lane_id() == 0 && m) wbase = atomicAdd(&tile_count, __popcll(m));
        wbase = __shfl(wbase, 0, 64);
        if (listed) {
            tile_list[wbase + __popcll(m & ((1ull << lane_id()) - 1ull))] = id;
        } else if (flags & kChanged) {
            const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(id);
            int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
            const int4 a = p[0], b = p[1];
            out[0] = a; out[1] = b;
            cell_flags[id] = flags & ~kChanged;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_base = tile_count ? atomicAdd(count, tile_count) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < tile_count; i += kBlock) list[tile_base + i] = tile_list[i];
}

template <int axis, bool SUBSET_ONLY>
__global__ void __launch_bounds__(kBlock) expand_listed(ExpandK k, const Entry* __restrict__ entries, const int* __restrict__ refs,
                                                        const float4* __restrict__ tris, const Cell* __restrict__ cells, Cell* __restrict__ new_cells,
                                                        int* __restrict__ cell_flags, const int* __restrict__ list, const int* __restrict__ count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *count) return;
    const int id = list[i];
    grow_cell<axis, SUBSET_ONLY>(k, entries, refs, tris, cells, new_cells, cell_flags, id, cell_flags[id]);
}

template <int axis, bool SUBSET_ONLY>
void listed_step(hipStream_t st, const ExpandK& k, const Entry* entries, const int* refs, const float4* tris, const Cell* cells, Cell* other,
                 int* flags, int n, int* list, int* count) {
    expand_select<axis><<<grid_blocks(n, kSelectTile), kBlock, 0, st>>>(cells, other, flags, n, list, count);
    expand_listed<axis, SUBSET_ONLY><<<grid_blocks(n, kBlock), kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, list, count);
}

} // namespace

extern "C" int hagrid_expand_grid(hagrid_ctx* ctx, hagrid_grid* grid, const void* tris_v, int iters) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
    const bool subset_only = ctx->opt_expand_subset_only != 0;      // the reference's compiled setting (expand.cu:159) is true
    if (!subset_only && !tris_v && iters > 0) HG_FAIL(ctx, HAGRID_EINVAL, "expand_grid: the precise mode needs the triangles");
    const float4* tris = static_cast<const float4*>(tris_v);
    if (iters <= 0) return HAGRID_OK;
    if (!grid->cells || !grid->entries || !grid->ref_ids) HG_FAIL(ctx, HAGRID_EINVAL, "expand_grid: incomplete (or compressed) grid");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ExpandK k;
    k.top = ivec3(grid->dims[0], grid->dims[1], grid->dims[2]);
    k.dims = k.top << grid->shift;
    k.shift = grid->shift;
    {   // expand.cu:208-216
        const vec3 lo(grid->bbox_min[0], grid->bbox_min[1], grid->bbox_min[2]), ext = vec3(grid->bbox_max[0], grid->bbox_max[1], grid->bbox_max[2]) - lo;
        k.gmin = lo; k.cell_size = ext / vec3(k.dims); k.grid_inv = vec3(k.dims) / ext;
    }
    const int n = grid->num_cells;
    Cell* cells = static_cast<Cell*>(grid->cells);
    Cell* other = pool_alloc<Cell>(ctx, size_t(n));
    int* flags = pool_alloc<int>(ctx, size_t(n));
    if (!other || !flags) { hagrid_mem_free(ctx, other); hagrid_mem_free(ctx, flags); return HAGRID_ENOMEM; }
    (void)hipMemsetAsync(flags, 0xFF, size_t(n) * sizeof(int), st);                     // expand.cu:206 (errors surface at the final check)
    const Entry* entries = static_cast<const Entry*>(grid->entries);
    const int* refs = static_cast<const int*>(grid->ref_ids);
    const int blocks = grid_blocks(n, kBlock);
    int* list = iters > 1 && ctx->opt_expand_listed ? pool_alloc<int>(ctx, size_t(n)) : nullptr;
    int* counts = ctx->dscratch + 160;             // one list length per listed pass
    const int max_listed = 48;
    if (list) (void)hipMemsetAsync(counts, 0, max_listed * sizeof(int), st);
    int listed = 0;
    for (int it = 0; it < iters; it++) {                                               // expansion_iter, expand.cu:184-197
        if (it > 0 && list && listed + 3 <= max_listed) {
            if (subset_only) {
                listed_step<0, true>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
                listed_step<1, true>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
                listed_step<2, true>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
            } else {
                listed_step<0, false>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
                listed_step<1, false>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
                listed_step<2, false>(st, k, entries, refs, tris, cells, other, flags, n, list, counts + listed++); std::swap(cells, other);
            }
            continue;
        }
        if (subset_only) {
            overlap_step<0, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
            overlap_step<1, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
            overlap_step<2, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
        } else {
            overlap_step<0, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
            overlap_step<1, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
            overlap_step<2, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); std::swap(cells, other);
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx->counts.expand_passes = 3 * iters; ctx->counts.expand_cells = n;
    hagrid_mem_free(ctx, flags);
    hagrid_mem_free(ctx, other);
    hagrid_mem_free(ctx, list);
    grid->cells = cells;
    if (e != hipSuccess) HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e));
    return HAGRID_OK;
}

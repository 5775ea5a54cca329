// expand.hip -- hagrid_expand_grid: grow each cell's integer box over face neighbours whose reference set
// is a subset of its own, on gfx950.
//
// Replaces the reference's expand.cu: expand (:199-223), expansion_iter (:184-197), overlap_step<axis> (:145-182),
// find_overlap (:60-143 -> face_growth), is_subset (:21-36 -> contains_all), compute_overlap (:39-57 -> room_before_prim).  subset_only = true is the reference's compiled setting
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

struct ExpandK { ivec3 dims; ivec3 top; int shift; vec3 gmin, cell_size, grid_inv; const int* voxel_cells; };   // expand.cu:5-9 (+ the flat map below)
constexpr int kChanged = 1 << 3;     // cell_flags (one byte per cell): the cell's box changed in the previous pass (bits 0-2: expand.cu:145-182)
struct CellRec { ivec3 lo; int begin; ivec3 hi; int end; };

__device__ __forceinline__ CellRec load_cell(const Cell* cells, int i) {
    const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
    const int4 a = p[0], b = p[1];
    CellRec c; c.lo = ivec3(a.x, a.y, a.z); c.begin = a.w; c.hi = ivec3(b.x, b.y, b.z); c.end = b.w;
    return c;
}
__device__ __forceinline__ int comp(const ivec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// ---- sorted id lists -------------------------------------------------------------------------------------------------------
// Reference lists are ascending and free of duplicates (the construction sorts them; expand.cu:20 relies on it).  They hold one
// to two ids on average and rarely more than a handful, so the common case (both lists of at most four ids) is decided in
// registers, every id against every id, without a loop or a data-dependent branch; longer lists: a binary search per id.

// position of the first element of a[0..n) that is >= x
__device__ __forceinline__ int lower_bound_in(const int* __restrict__ a, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// does list `big` contain every id of list `small`?  (the question is_subset of expand.cu:21-36 answers)
__device__ __forceinline__ bool contains_all(const int* __restrict__ big, int nbig, const int* __restrict__ small, int nsmall) {
    if (nsmall > nbig) return false;
    if (nsmall == 0) return true;
    if (nbig <= 4) {
        // both lists in registers (eight independent loads), sixteen compares, no loop: unused slots of the big list hold -1
        // (ids are >= 0: never equal), unused slots of the small list repeat its first id
        const int b0 = big[0], b1 = nbig > 1 ? big[1] : -1, b2 = nbig > 2 ? big[2] : -1, b3 = nbig > 3 ? big[3] : -1;
        const int s0 = small[0], s1 = nsmall > 1 ? small[1] : s0, s2 = nsmall > 2 ? small[2] : s0, s3 = nsmall > 3 ? small[3] : s0;
        auto in_big = [&](int x) { return (x == b0) | (x == b1) | (x == b2) | (x == b3); };
        return in_big(s0) & in_big(s1) & in_big(s2) & in_big(s3);
    }
    for (int j = 0; j < nsmall; j++) {
        const int x = small[j], at = lower_bound_in(big, nbig, x);
        if (at == nbig || big[at] != x) return false;
    }
    return true;
}

__device__ __forceinline__ Tri load_tri(const float4* __restrict__ tris, int i) {
    const float4* p = tris + 3 * size_t(i);
    const float4 a = p[0], b = p[1], c = p[2];
    return Tri(vec3(a.x, a.y, a.z), a.w, vec3(b.x, b.y, b.z), b.w, vec3(c.x, c.y, c.z), c.w);
}

// The expansion looks up the cell of a voxel for every neighbour of every face walk -- nine passes over the cells, a dozen look-ups per cell and
// pass -- and the grid does not change its cells' numbers while it expands: the voxel map is resolved ONCE into one word per voxel (the cell that
// holds it), top-level cell by top-level cell with x fastest inside (a face walk stays inside a few lines).  A look-up is then one load instead of
// the chain through the voxel map's levels (two dependent loads after flatten_grid).  Grids whose virtual resolution exceeds 2^28 voxels keep the chain.
__global__ void __launch_bounds__(kBlock) fill_voxel_cells(ExpandK k, const Entry* __restrict__ entries, int* __restrict__ voxel_cells, int voxels) {
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= voxels) return;
    const int s = k.shift, m = (1 << s) - 1, top = v >> (3 * s), local = v & ((1 << (3 * s)) - 1);
    const int tx = top % k.top.x, ty = (top / k.top.x) % k.top.y, tz = top / (k.top.x * k.top.y);
    const ivec3 at((tx << s) + (local & m), (ty << s) + ((local >> s) & m), (tz << s) + (local >> (2 * s)));
    voxel_cells[v] = int(lookup_entry(entries, s, k.top, at));
}
__device__ __forceinline__ int cell_of_voxel(const ExpandK& k, const Entry* __restrict__ entries, const ivec3& at) {
    if (!k.voxel_cells) return int(lookup_entry(entries, k.shift, k.top, at));
    const int s = k.shift, m = (1 << s) - 1;
    const int top = (at.x >> s) + k.top.x * ((at.y >> s) + k.top.y * (at.z >> s));
    return k.voxel_cells[(top << (3 * s)) + (at.x & m) + (((at.y & m) + ((at.z & m) << s)) << s)];
}

// Precise mode (expand.cu:39-57): a triangle the neighbour references and the cell does not limits the growth to the voxel layer
// where its bounding box begins -- if the box overlaps the cell's cross-section at all.  `room` is the number of layers the face
// may still move (a magnitude, whatever the direction).
template <int AXIS>
__device__ __forceinline__ int room_before_prim(const ExpandK& k, const Tri& prim, const CellRec& cell, const BBox& cross, int room, bool UP) {
    constexpr int A1 = (AXIS + 1) % 3, A2 = (AXIS + 2) % 3;
    const BBox pb = prim.bbox();
    const bool overlaps = get<A1>(pb.min) <= get<A1>(cross.max) && get<A1>(pb.max) >= get<A1>(cross.min) &&
                          get<A2>(pb.min) <= get<A2>(cross.max) && get<A2>(pb.max) >= get<A2>(cross.min);
    if (!overlaps) return room;
    const int layer = int(((UP ? get<AXIS>(pb.min) : get<AXIS>(pb.max)) - get<AXIS>(k.gmin)) * get<AXIS>(k.grid_inv));
    const int free_layers = UP ? layer - comp(cell.hi, AXIS) : comp(cell.lo, AXIS) - layer - 1;
    return max(min(room, free_layers), 0);
}

// One growth direction of one cell (find_overlap, expand.cu:60-143).  Returns the signed number of voxel layers the face moves;
// sets `again` when the move used all the room the neighbours leave, i.e. the cell may grow further in the next iteration.
// The face is swept row by row in the order the reference sweeps it -- a neighbour's (possibly already expanded) box decides how
// far the sweep jumps, so the set of neighbours looked at is part of the result.  Quantities are kept as magnitudes:
//   reach = the least extent of a neighbour seen so far beyond the face, room = min(reach, limits from references).
// The direction is a run-time argument: in the listed passes the two directions of a cell run in two neighbouring lanes, in lock step.
template <int AXIS, bool SUBSET_ONLY>
__device__ __forceinline__ int face_growth(const ExpandK& k, const Entry* __restrict__ entries, const int* __restrict__ refs, const float4* __restrict__ tris,
                                           const Cell* __restrict__ cells, const CellRec& cell, bool UP, bool& again) {
    constexpr int A1 = (AXIS + 1) % 3, A2 = (AXIS + 2) % 3;
    const int face = UP ? comp(cell.hi, AXIS) : comp(cell.lo, AXIS);
    const int extent = comp(k.dims, AXIS);
    if (UP ? face >= extent : face <= 0) return 0;                       // the face lies on the grid boundary (expand.cu:12-18)
    const int layer = UP ? face : face - 1;                               // the voxel layer just beyond the face
    const int lo1 = comp(cell.lo, A1), hi1 = comp(cell.hi, A1), lo2 = comp(cell.lo, A2), hi2 = comp(cell.hi, A2);
    const int own = cell.end - cell.begin;
    int reach = extent, room = extent;
    int u = lo1, v = lo2, row_step = comp(k.dims, A2);
    for (;;) {
        const ivec3 at = AXIS == 0 ? ivec3(layer, u, v) : (AXIS == 1 ? ivec3(v, layer, u) : ivec3(u, v, layer));
        const CellRec nb = load_cell(cells, cell_of_voxel(k, entries, at));
        reach = min(reach, UP ? comp(nb.hi, AXIS) - face : face - comp(nb.lo, AXIS));
        room = min(room, reach);
        const int theirs = nb.end - nb.begin;
        if (SUBSET_ONLY) {
            if (!contains_all(refs + cell.begin, own, refs + nb.begin, theirs)) { room = 0; break; }
        } else if (theirs > 0) {
            // every id of the neighbour that the cell does not hold limits the growth (expand.cu:96-127); both lists ascend,
            // so one cursor into the cell's list is enough
            const BBox cross(k.gmin + k.cell_size * vec3(cell.lo), k.gmin + k.cell_size * vec3(cell.hi));
            int mine = 0;
            for (int j = 0; j < theirs && room > 0; j++) {
                const int id = refs[nb.begin + j];
                while (mine < own && refs[cell.begin + mine] < id) mine++;
                if (mine < own && refs[cell.begin + mine] == id) continue;
                room = room_before_prim<AXIS>(k, load_tri(tris, id), cell, cross, room, UP);
            }
            if (room == 0) break;
        }
        // next neighbour of the row, or the next row (a row advances by the smallest extent of the neighbours it crossed)
        row_step = min(row_step, comp(nb.hi, A2) - v);
        u = comp(nb.hi, A1);
        if (u >= hi1) {
            u = lo1; v += row_step; row_step = comp(k.dims, A2);
            if (v >= hi2) break;
        }
    }
    again |= room == reach;
    return UP ? room : -room;
}

// PAIRED: a cell's two growth directions are independent walks of dependent gathers (voxel map -> neighbour cell -> its list);
// in the passes over the LISTED cells (few threads, every one a serial chain) they run in two neighbouring lanes (lane & 1 =
// direction), the even lane collects both results and writes the cell: -10 % on those passes.  The pass over ALL cells is bound
// by the gather rate, not by the chains (two lanes per cell: +12 %), and keeps one thread per cell.
template <int axis, bool SUBSET_ONLY, bool PAIRED>
__device__ __forceinline__ void grow_cell(const ExpandK& k, const Entry* __restrict__ entries, const int* __restrict__ refs, const float4* __restrict__ tris,
                                          const Cell* __restrict__ cells, Cell* __restrict__ new_cells, unsigned char* __restrict__ cell_flags, int id, int flags, bool up) {
    CellRec cell = load_cell(cells, id);
    bool flag = false;
    int ov1, ov2;
    if (PAIRED) {                                                      // `id` and `flags` are the same in both lanes of a pair
        bool again = false;
        const int mine = face_growth<axis, SUBSET_ONLY>(k, entries, refs, tris, cells, cell, up, again);
        const int other = __shfl_xor(mine, 1, 64);
        flag = again | (__shfl_xor(int(again), 1, 64) != 0);
        if (up) return;
        ov1 = mine; ov2 = other;                                       // this lane walked down, its neighbour up
    } else {
        ov1 = face_growth<axis, SUBSET_ONLY>(k, entries, refs, tris, cells, cell, false, flag);
        ov2 = face_growth<axis, SUBSET_ONLY>(k, entries, refs, tris, cells, cell, true, flag);
    }
    if (axis == 0) { cell.lo.x += ov1; cell.hi.x += ov2; }
    if (axis == 1) { cell.lo.y += ov1; cell.hi.y += ov2; }
    if (axis == 2) { cell.lo.z += ov1; cell.hi.z += ov2; }
    cell_flags[id] = (unsigned char)((flag ? 1 << axis : 0) | (flags & ~((1 << axis) | kChanged)) | ((ov1 | ov2) ? kChanged : 0));
    int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
    out[0] = make_int4(cell.lo.x, cell.lo.y, cell.lo.z, cell.begin);
    out[1] = make_int4(cell.hi.x, cell.hi.y, cell.hi.z, cell.end);
}

// overlap_step (expand.cu:145-182)
template <int axis, bool SUBSET_ONLY>
__global__ void __launch_bounds__(kBlock) overlap_step(ExpandK k, const Entry* __restrict__ entries, const int* __restrict__ refs,
                                                       const float4* __restrict__ tris, const Cell* __restrict__ cells, Cell* __restrict__ new_cells,
                                                       unsigned char* __restrict__ cell_flags, int num_cells) {
    const int id = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;          // (the face walk reads the neighbours' cells and lists: one L2 per stretch of cells)
    if (id >= num_cells) return;
    const int flags = cell_flags[id];
    if ((flags & (1 << axis)) == 0) {      // copy through
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(id);
        int4* out = reinterpret_cast<int4*>(new_cells) + 2 * size_t(id);
        const int4 a = p[0], b = p[1];
        out[0] = a; out[1] = b;
        if (flags & kChanged) cell_flags[id] = (unsigned char)(flags & ~kChanged);
        return;
    }
    grow_cell<axis, SUBSET_ONLY, false>(k, entries, refs, tris, cells, new_cells, cell_flags, id, flags, false);
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
__global__ void __launch_bounds__(kBlock) expand_select(const Cell* __restrict__ cells, Cell* __restrict__ new_cells, unsigned char* __restrict__ cell_flags,
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
            cell_flags[id] = (unsigned char)(flags & ~kChanged);
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
                                                        unsigned char* __restrict__ cell_flags, const int* __restrict__ list, const int* __restrict__ count) {
    // (the grid is sized for all cells; the workgroups that hold listed cells -- the first ones -- take them XCD by XCD, like overlap_step)
    const int n = *count, active = (2 * n + kBlock - 1) / kBlock;
    if (int(blockIdx.x) >= active) return;
    const int t = xcd_block(blockIdx.x, active) * kBlock + threadIdx.x;
    const int i = t >> 1;                                          // two lanes per listed cell, one per direction
    if (i >= n) return;
    const int id = list[i];
    grow_cell<axis, SUBSET_ONLY, true>(k, entries, refs, tris, cells, new_cells, cell_flags, id, cell_flags[id], (t & 1) != 0);
}

template <int axis, bool SUBSET_ONLY>
void listed_step(hipStream_t st, const ExpandK& k, const Entry* entries, const int* refs, const float4* tris, const Cell* cells, Cell* other,
                 unsigned char* flags, int n, int* list, int* count) {
    expand_select<axis><<<grid_blocks(n, kSelectTile), kBlock, 0, st>>>(cells, other, flags, n, list, count);
    expand_listed<axis, SUBSET_ONLY><<<grid_blocks(2ll * n, kBlock), kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, list, count);
}

} // namespace

extern "C" int hagrid_expand_grid(hagrid_ctx* ctx, hagrid_grid* grid, const void* tris_v, int iters) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    const bool subset_only = ctx->opt_expand_subset_only != 0;      // the reference's compiled setting (expand.cu:159) is true
    if (!subset_only && !tris_v && iters > 0) HG_FAIL(ctx, HAGRID_EINVAL, "expand_grid: the precise mode needs the triangles");
    const float4* tris = static_cast<const float4*>(tris_v);
    if (iters <= 0) return HAGRID_OK;
    if (!grid->cells || !grid->entries || !grid->ref_ids) HG_FAIL(ctx, HAGRID_EINVAL, "expand_grid: incomplete (or compressed) grid");
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
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
    unsigned char* flags = pool_alloc<unsigned char>(ctx, size_t(n));
    if (!other || !flags) { hagrid_mem_free(ctx, other); hagrid_mem_free(ctx, flags); return HAGRID_ENOMEM; }
    (void)hipMemsetAsync(flags, 0xFF, size_t(n), st);                     // expand.cu:206 (errors surface at the final check)
    const Entry* entries = static_cast<const Entry*>(grid->entries);
    const int* refs = static_cast<const int*>(grid->ref_ids);
    const int blocks = grid_blocks(n, kBlock);
    int* list = iters > 1 ? pool_try_alloc<int>(ctx, size_t(n)) : nullptr;      // (without it the later iterations fall back to the all-cells pass)
    const long long voxels = (long long)k.dims.x * k.dims.y * k.dims.z;
    int* voxel_cells = (ctx->opt_expand_voxel_map && voxels <= (1ll << 28)) ? pool_try_alloc<int>(ctx, size_t(voxels)) : nullptr;   // (without it: the chain through the voxel map)
    k.voxel_cells = voxel_cells;
    if (voxel_cells) { ExpandK kf = k; kf.voxel_cells = nullptr; fill_voxel_cells<<<grid_blocks(voxels, kBlock), kBlock, 0, st>>>(kf, entries, voxel_cells, int(voxels)); HG_DBG(ctx); }
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
            overlap_step<0, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
            overlap_step<1, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
            overlap_step<2, true><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
        } else {
            overlap_step<0, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
            overlap_step<1, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
            overlap_step<2, false><<<blocks, kBlock, 0, st>>>(k, entries, refs, tris, cells, other, flags, n); HG_DBG(ctx); std::swap(cells, other);
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx->counts.expand_passes = 3 * iters; ctx->counts.expand_cells = n;
    hagrid_mem_free(ctx, flags);
    hagrid_mem_free(ctx, other);
    hagrid_mem_free(ctx, list);
    hagrid_mem_free(ctx, voxel_cells);
    grid->cells = cells;
    if (e != hipSuccess) HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e));
    return HAGRID_OK;
}

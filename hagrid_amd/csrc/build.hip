// build.hip -- hagrid_build_grid: construction of the initial (octree-structured) irregular grid on gfx950.
//
// Replaces the reference's build.cu (build :718-758, first_build_iter :470-525, build_iter :527-619,
// concat_levels :621-716 and their 20 kernels) and the CUB calls behind them (parallel.cuh).  The result
// -- entries, cells, ref_ids, dims, bbox, shift, offsets -- is bit-identical to the CPU oracle's.
//
// How it differs from the reference's pipeline (not a port):
//  * Only the ORDER OF CELLS is part of the result; the final reference list of a cell is the ascending
//    list of primitive ids.  So references never go through an ordered scan, a flagged partition or a radix
//    sort.  From the top level on the references of a level are kept GROUPED BY CELL, in cell order: a cell's
//    references are a segment of {reference, cell} pairs, its count is the segment's length -- known when its
//    parent is classified -- and a leaf's list is its segment.  One returning atomic per (primitive, top-level
//    cell) pair places the pair in its cell's segment (the same atomic counts the cell's references, which the
//    depth rule needs anyway); below the top level there are no global atomics at all: child counts and emission
//    cursors live in LDS, per workgroup tile of references aligned to cells.  Ordered scans run over cells only.
//    (Rounds 1-4 kept the references in emission order, drew a list slot per kept reference with a returning
//    atomic, scattered the kept references at the end and sorted every list: build_grid 1.32 -> 1.11 ms for the
//    1M-triangle scene, same box; profiles/NOTES.md "Round 5".)
//  * Per level there is ONE host round trip (kept references, references and cells of the next levels) instead
//    of four; the bounding box, the reference count, the shift and the first level's cells cost four more;
//    concat costs one.  All temporaries of a construction come from one pool buffer (Arena).
//  * Per-primitive bounding boxes are recomputed from the 48-byte triangle instead of being stored.
//  * The top-level cell a reference sits in is known from its index, so the SAT filter (filter_refs,
//    build.cu:139-157) is fused into the emission kernel.
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"
#include "hagrid/prims.h"

#include <cstring>
#include <vector>

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct BuildK {              // build.cu:37-40 (__constant__ there, kernel argument here)
    ivec3 dims;              // top-level resolution
    int shift;
    vec3 bmin, bmax;         // enlarged grid box
    vec3 cell_size;          // extents / (dims << shift)
};

__device__ __forceinline__ Tri load_tri(const float4* __restrict__ tris, int i) {
    const float4* p = tris + 3 * size_t(i);
    const float4 a = p[0], b = p[1], c = p[2];
    return Tri(vec3(a.x, a.y, a.z), a.w, vec3(b.x, b.y, b.z), b.w, vec3(c.x, c.y, c.z), c.w);
}
__device__ __forceinline__ void store_cell(Cell* cells, int i, ivec3 lo, int begin, ivec3 hi, int end) {
    int4* p = reinterpret_cast<int4*>(cells) + 2 * size_t(i);
    p[0] = make_int4(lo.x, lo.y, lo.z, begin);
    p[1] = make_int4(hi.x, hi.y, hi.z, end);
}
__device__ __forceinline__ ivec3 load_cell_min(const Cell* cells, int i) {
    const int4 a = reinterpret_cast<const int4*>(cells)[2 * size_t(i)];
    return ivec3(a.x, a.y, a.z);
}
__device__ __forceinline__ BBox cell_world_box(const BuildK& k, ivec3 lo, ivec3 hi) {   // build.cu:150-151
    return BBox(k.bmin + vec3(lo) * k.cell_size, k.bmin + vec3(hi) * k.cell_size);
}

// ---- scene bounding box (compute_bboxes + DeviceReduce, build.cu:725-727) ----------------------------
// one partial box per workgroup, then a single workgroup folds the partials
__global__ void __launch_bounds__(kBlock) bbox_partials(const float4* __restrict__ tris, int n, float* __restrict__ partials) {
    __shared__ float lds[kWaves][6];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const BBox b = load_tri(tris, i).bbox();
        lo[0] = min(lo[0], b.min.x); lo[1] = min(lo[1], b.min.y); lo[2] = min(lo[2], b.min.z);
        hi[0] = max(hi[0], b.max.x); hi[1] = max(hi[1], b.max.y); hi[2] = max(hi[2], b.max.z);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = wave_min_f(lo[c]); hi[c] = wave_max_f(hi[c]); }
    if (lane_id() == 0) for (int c = 0; c < 3; c++) { lds[wave_id()][c] = lo[c]; lds[wave_id()][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = lds[0][threadIdx.x];
        for (int w = 1; w < kWaves; w++) v = threadIdx.x < 3 ? min(v, lds[w][threadIdx.x]) : max(v, lds[w][threadIdx.x]);
        partials[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void __launch_bounds__(kBlock) bbox_final(const float* __restrict__ partials, int num, float* __restrict__ out) {
    __shared__ float lds[kWaves][6];
    float v[6] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = threadIdx.x; i < num; i += kBlock)
        for (int c = 0; c < 6; c++) v[c] = c < 3 ? min(v[c], partials[i * 6 + c]) : max(v[c], partials[i * 6 + c]);
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = c < 3 ? wave_min_f(v[c]) : wave_max_f(v[c]);
    if (lane_id() == 0) for (int c = 0; c < 6; c++) lds[wave_id()][c] = v[c];
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = lds[0][threadIdx.x];
        for (int w = 1; w < kWaves; w++) r = threadIdx.x < 3 ? min(r, lds[w][threadIdx.x]) : max(r, lds[w][threadIdx.x]);
        out[threadIdx.x] = r;
    }
}

// ---- top level -----------------------------------------------------------------------------------------
// count_new_refs (build.cu:57-66) + count_refs_per_cell (build.cu:246-253, counted BEFORE the SAT filter).
// A primitive that covers many top-level cells (a ground plane) is handled by the whole wavefront: lanes stride
// over its cell range -- the wave64 counterpart of the reference's 32-lane cooperative emission (build.cu:106-135).
constexpr int kCoopCells = 64;


// compute_log_dims (build.cu:256-270) + the max reduction of build.cu:508
__global__ void __launch_bounds__(kBlock) top_log_dims(const int* __restrict__ refs_per_cell, int num_top, BuildK k, float snd_density,
                                                       int* __restrict__ log_dims, int* __restrict__ max_out) {
    __shared__ int lds[kWaves];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int ld = 0;
    if (i < num_top) {
        const vec3 ext = BBox(k.bmin, k.bmax).extents() / vec3(k.dims);
        const ivec3 d = compute_grid_dims(BBox(vec3(0, 0, 0), ext), refs_per_cell[i], snd_density);
        const int max_dim = max(d.x, max(d.y, d.z));
        ld = 31 - __clz(max_dim);
        ld = (1 << ld) < max_dim ? ld + 1 : ld;
        log_dims[i] = ld;
    }
    ld = wave_max(ld);
    if (lane_id() == 0) lds[wave_id()] = ld;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = lds[0];
        for (int w = 1; w < kWaves; w++) m = max(m, lds[w]);
        if (m > 0) atomicMax(max_out, m);
    }
}


// the grid constants of the top-level emission, with the shift read on the device (SegOut below)
__device__ __forceinline__ BuildK with_device_shift(BuildK k, const int* __restrict__ shift_dev, vec3 extents) {
    k.shift = *shift_dev;
    k.cell_size = extents / vec3(k.dims << k.shift);
    return k;
}

// ---- one subdivision level -----------------------------------------------------------------------------
// scan functors: 8 children per split cell (build.cu:557-559), then update_entries (build.cu:317-329)
struct ChildCountIn {
    const uint32_t* entries;
    __device__ int operator()(int i) const { return (entries[i] & 3u) ? 8 : 0; }
    __device__ void load4(int i, int n, int* v) const {                 // a lane's four consecutive items: one 16-byte access
        if (i + 4 <= n && lb_aligned16(entries + i)) {
            const uint4 e = *reinterpret_cast<const uint4*>(entries + i);
            v[0] = (e.x & 3u) ? 8 : 0; v[1] = (e.y & 3u) ? 8 : 0; v[2] = (e.z & 3u) ? 8 : 0; v[3] = (e.w & 3u) ? 8 : 0;
        } else {
            for (int c = 0; c < 4; c++) v[c] = i + c < n ? (*this)(i + c) : 0;
        }
    }
};
struct UpdateEntriesOut {
    uint32_t* entries;
    __device__ static uint32_t word(uint32_t e, int i, int start) { const uint32_t ld = e & 3u; return ld | (uint32_t(ld ? start : i) << 2); }
    __device__ void operator()(int i, int start) const { entries[i] = word(entries[i], i, start); }
    __device__ void store4(int i, int n, const int* v) const {
        if (i + 4 <= n && lb_aligned16(entries + i)) {
            uint4* p = reinterpret_cast<uint4*>(entries + i);
            const uint4 e = *p;
            *p = make_uint4(word(e.x, i, v[0]), word(e.y, i + 1, v[1]), word(e.z, i + 2, v[2]), word(e.w, i + 3, v[3]));
        } else {
            for (int c = 0; c < 4; c++) if (i + c < n) (*this)(i + c, v[c]);
        }
    }
};

// compute_split_masks (build.cu:160-216)
__device__ __forceinline__ int split_mask(const BuildK& k, ivec3 lo, ivec3 hi, const Tri& tri) {
    const vec3 cmin = k.bmin + k.cell_size * vec3(lo);
    const vec3 cmax = k.bmin + k.cell_size * vec3(hi);
    const vec3 mid = (cmin + cmax) * 0.5f;
    int mask = 0xFF;
    const BBox rb = tri.bbox();
    if (rb.min.x > cmax.x || rb.max.x < cmin.x) mask = 0;
    if (rb.min.x > mid.x) mask &= 0xAA;
    if (rb.max.x < mid.x) mask &= 0x55;
    if (rb.min.y > cmax.y || rb.max.y < cmin.y) mask = 0;
    if (rb.min.y > mid.y) mask &= 0xCC;
    if (rb.max.y < mid.y) mask &= 0x33;
    if (rb.min.z > cmax.z || rb.max.z < cmin.z) mask = 0;
    if (rb.min.z > mid.z) mask &= 0xF0;
    if (rb.max.z < mid.z) mask &= 0x0F;
    int todo = mask;
    while (todo) {
        const int i = __ffs(todo) - 1;
        todo &= todo - 1;
        const BBox b(vec3(i & 1 ? mid.x : cmin.x, i & 2 ? mid.y : cmin.y, i & 4 ? mid.z : cmin.z),
                     vec3(i & 1 ? cmax.x : mid.x, i & 2 ? cmax.y : mid.y, i & 4 ? cmax.z : mid.z));
        if (!intersect_prim_cell(tri, b)) mask &= ~(1 << i);
    }
    return mask;
}


// ---- concatenation ---------------------------------------------------------------------------------------
// leaf flag + kept-reference count per cell, scanned over all levels in cell order
struct LeafIn {
    const uint32_t* entries; const int* cell_counts;
    __device__ static Int2 item(uint32_t e, int count) { const bool leaf = (e & 3u) == 0; return Int2{ leaf ? 1 : 0, leaf ? count : 0 }; }
    __device__ Int2 operator()(int i) const { return item(entries[i], cell_counts[i]); }
    __device__ void load4(int i, int n, Int2* v) const {
        if (i + 4 <= n && lb_aligned16(entries + i) && lb_aligned16(cell_counts + i)) {
            const uint4 e = *reinterpret_cast<const uint4*>(entries + i);
            const int4 c = *reinterpret_cast<const int4*>(cell_counts + i);
            v[0] = item(e.x, c.x); v[1] = item(e.y, c.y); v[2] = item(e.z, c.z); v[3] = item(e.w, c.w);
        } else {
            for (int k = 0; k < 4; k++) v[k] = i + k < n ? (*this)(i + k) : Int2{0, 0};
        }
    }
};
struct LeafOut {
    int* start_cell; int* ref_begin;
    __device__ void operator()(int i, Int2 v) const { start_cell[i] = v.a; ref_begin[i] = v.b; }
    __device__ void store4(int i, int n, const Int2* v) const {
        if (i + 4 <= n && lb_aligned16(start_cell + i) && lb_aligned16(ref_begin + i)) {
            *reinterpret_cast<int4*>(start_cell + i) = make_int4(v[0].a, v[1].a, v[2].a, v[3].a);
            *reinterpret_cast<int4*>(ref_begin + i) = make_int4(v[0].b, v[1].b, v[2].b, v[3].b);
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n) (*this)(i + k, v[k]);
        }
    }
};


// ---- references grouped by cell ---------------------------------------------------------------------------------------------------------------------
// (The level loop of rounds 1-4 handed a kept reference its list slot with one returning atomic, scattered it there at the end and left the references of a
// level in the order the emission tiles drew.)  The references of a level are kept grouped by cell, in cell order, from the top
// level on: a cell's references are a SEGMENT [seg_begin[c], seg_begin[c + 1]) of {reference, cell} pairs, its count is the segment's length -- known when its
// PARENT is classified -- and a leaf's list is its segment.  Per level: classify (SAT masks + the eight child counts of every splitting cell, reduced per
// wavefront and cell, then in LDS), ONE scan over the new cells (segment starts and child-cell bases together), emit (the children of a cell go to the segments
// of its child cells: LDS cursors, output staged in LDS).  The deepest level needs no pass at all, the concatenation copies segments.  Workgroups take TILES of
// references aligned to cells: tile t owns the cells whose segments start in [t * kTileRefs, (t + 1) * kTileRefs) -- tile_first[t] is the first of them, left
// there by the scan that made the segment starts.
#ifndef HG_TILE_REFS                       // (variant builds: tools/build_variant.sh)
#define HG_TILE_REFS 1024
#endif
#ifndef HG_CHUNK_CELLS
#define HG_CHUNK_CELLS 256
#endif
#ifndef HG_EMIT_CHUNK
#define HG_EMIT_CHUNK 256
#endif
#ifndef HG_STAGE
#define HG_STAGE 1024
#endif
constexpr int kTileRefs = HG_TILE_REFS;    // references per tile (nominal: a tile ends with the cell it ends in)
constexpr int kChunkCells = HG_CHUNK_CELLS; // cells a workgroup of the classification holds counters for at a time
constexpr int kEmitChunk = HG_EMIT_CHUNK;  // cells a workgroup of the emission holds cursors for at a time
constexpr int kStage = HG_STAGE;           // output slots a workgroup stages in LDS per chunk

// segment starts from per-cell reference counts; the cell a tile boundary falls into names the tile's first cell (the cell behind it)
__device__ __forceinline__ void mark_tiles(int* __restrict__ tile_first, int i, int start, int count) {
    for (int t = start / kTileRefs + 1; t <= (start + count) / kTileRefs; t++) tile_first[t] = i + 1;
}
struct SegIn {
    const int* counts;
    __device__ int operator()(int i) const { return counts[i]; }
};
// ... and emit_top_cells (build.cu:332-351) with the levels the cell may still be split (log_dims, build.cu:256-270; update_log_dims :273-278 becomes "one less per
// level"): while a level is under construction the `begin` word of its cells is free and carries that number, where classify_refs finds it in the record it loads
// anyway.  The top-level cells' voxel-map words and reference counts are cleared here as well (before the references of the level are handed out): no fill launches.
// The shift -- the deepest subdivision of any top-level cell, top_log_dims -- is read from device memory here and in emit_top_refs: the host learns it together
// with the number of cells of the next level, one round trip later.
struct SegOut {
    const int* counts; int* seg_begin; int* tile_first;
    Cell* cells; uint32_t* entries; int* cell_counts; const int* log_dims; const int* shift_dev; ivec3 dims;
    __device__ void operator()(int i, int start) const {
        seg_begin[i] = start; mark_tiles(tile_first, i, start, counts[i]);
        const int shift = *shift_dev;
        entries[i] = 0u; cell_counts[i] = 0;
        const int x = i % dims.x, y = (i / dims.x) % dims.y, z = i / (dims.x * dims.y);
        const ivec3 lo(x << shift, y << shift, z << shift);
        store_cell(cells, i, lo, log_dims[i], lo + ivec3(1 << shift), 0);
    }
};
// the new cells of a level: {references, 8 if the cell splits again} -> {segment start, first child cell}; update_entries fused as in UpdateEntriesOut
struct ChildSegIn {
    const int* counts; const uint32_t* entries;
    __device__ Int2 operator()(int i) const { return Int2{counts[i], (entries[i] & 3u) ? 8 : 0}; }
    __device__ void load4(int i, int n, Int2* v) const {
        if (i + 4 <= n && lb_aligned16(counts + i) && lb_aligned16(entries + i)) {
            const int4 c = *reinterpret_cast<const int4*>(counts + i);
            const uint4 e = *reinterpret_cast<const uint4*>(entries + i);
            v[0] = Int2{c.x, (e.x & 3u) ? 8 : 0}; v[1] = Int2{c.y, (e.y & 3u) ? 8 : 0}; v[2] = Int2{c.z, (e.z & 3u) ? 8 : 0}; v[3] = Int2{c.w, (e.w & 3u) ? 8 : 0};
        } else {
            for (int k = 0; k < 4; k++) v[k] = i + k < n ? (*this)(i + k) : Int2{0, 0};
        }
    }
};
struct ChildSegOut {
    const int* counts; uint32_t* entries; int* seg_begin; int* tile_first;
    __device__ void operator()(int i, Int2 v) const {
        seg_begin[i] = v.a;
        entries[i] = UpdateEntriesOut::word(entries[i], i, v.b);
        mark_tiles(tile_first, i, v.a, counts[i]);
    }
    __device__ void store4(int i, int n, const Int2* v) const {
        if (i + 4 <= n && lb_aligned16(counts + i) && lb_aligned16(entries + i) && lb_aligned16(seg_begin + i)) {
            const int4 c = *reinterpret_cast<const int4*>(counts + i);
            uint4* pe = reinterpret_cast<uint4*>(entries + i);
            const uint4 e = *pe;
            *reinterpret_cast<int4*>(seg_begin + i) = make_int4(v[0].a, v[1].a, v[2].a, v[3].a);
            *pe = make_uint4(UpdateEntriesOut::word(e.x, i, v[0].b), UpdateEntriesOut::word(e.y, i + 1, v[1].b), UpdateEntriesOut::word(e.z, i + 2, v[2].b), UpdateEntriesOut::word(e.w, i + 3, v[3].b));
            // (the four cells together: a tile boundary inside their references is rare -- 2048 references per tile, a handful per cell)
            if (v[0].a / kTileRefs != (v[3].a + c.w) / kTileRefs) {
                mark_tiles(tile_first, i, v[0].a, c.x); mark_tiles(tile_first, i + 1, v[1].a, c.y); mark_tiles(tile_first, i + 2, v[2].a, c.z); mark_tiles(tile_first, i + 3, v[3].a, c.w);
            }
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n) (*this)(i + k, v[k]);
        }
    }
};

// top level, pass 1: the number of top-level cells every primitive's box covers (count_new_refs, build.cu:57-66)
// Primitives whose box covers kCoopCells top-level cells or more -- the walls of a hall, a ground plane: a few triangles with thousands of cells each -- are LISTED
// and get workgroups of their own behind the others' in count_top_refs and emit_top_refs (a workgroup per primitive, a thread per cell): as guests of the wavefront their index falls into,
// ten such triangles were one wavefront's 360 dependent rounds of returning atomics / of SAT tests -- count_top_refs 311 us and emit_top_refs 561 us on the stadium
// scene against 127 / 73 us on the soup (round 6, gpurun_out/r6i, r6j).
__global__ void __launch_bounds__(kBlock) top_range_sizes(const float4* __restrict__ tris, int n, BuildK k, int* __restrict__ counts, int* __restrict__ big_list, int* __restrict__ big_count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int size = 0;
    if (i < n) { size = max(0, compute_range(k.dims, BBox(k.bmin, k.bmax), load_tri(tris, i).bbox()).size()); counts[i] = size; }
    const bool big = size >= kCoopCells;
    const int at = wave_append(big ? 1 : 0, big_count);
    if (big) big_list[at] = i;
}
// A returning atomic increment whose lanes may name the SAME counter.  A mesh puts thousands of consecutive primitives into one top-level cell (a finely
// tessellated object smaller than a cell: 40 000 triangles of the stadium's grain of dust, 150 000 of a blob of the clustered scene in eight cells), and atomics on
// one address are served one after the other by its L2 channel, ~15 ns each across the XCDs: count_top_refs 128 us on the uniform soup, 1028 us on the clustered
// scene, 865 us on the stadium (round 6, gpurun_out/r6i).  So the wavefront looks at the counter of its first active lane: when other lanes name it too, the
// lanes are served group by group -- one atomic per distinct counter, its return value shared out by lane position -- for up to sixteen groups; lanes left
// over, and wavefronts whose first counter is named once (the soup: 64 lanes, 64 cells), take the plain atomic.
__device__ __forceinline__ int wave_shared_increment(int* __restrict__ counters, int index, bool active) {
    int rank = 0;
    unsigned long long todo = __ballot(active);
    const unsigned long long below = (1ull << lane_id()) - 1ull;
    for (int round = 0; todo && round < 16; round++) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __shfl(index, leader, 64);
        const unsigned long long same = __ballot(active && index == c) & todo;
        if (round == 0 && (same & (same - 1)) == 0ull) break;           // named once: no crowd here
        int base = 0;
        if (lane_id() == leader) base = atomicAdd(counters + c, __popcll(same));
        base = __shfl(base, leader, 64);
        if ((same >> lane_id()) & 1ull) rank = base + __popcll(same & below);
        todo &= ~same;
    }
    if ((todo >> lane_id()) & 1ull) rank = atomicAdd(counters + index, 1);
    return rank;
}

// top level, pass 2: count_refs_per_cell (build.cu:246-253, counted BEFORE the SAT filter) with the count's return value kept per (primitive, cell) pair: the
// pair's place inside its cell's segment.  Pairs in the order of emit_top_refs (x fastest; large ranges spread over the wavefront).
__global__ void __launch_bounds__(kBlock) count_top_refs(const float4* __restrict__ tris, int n, BuildK k, const int* __restrict__ start_emit,
                                                                int* __restrict__ refs_per_cell, int* __restrict__ pair_rank,
                                                                const int* __restrict__ big_list, const int* __restrict__ big_count, int first_big_block) {
    if (int(blockIdx.x) >= first_big_block) {      // the blocks behind the primitives' own: a workgroup per large primitive, its cells over the threads (every cell once: plain atomics)
        const int nb = *big_count;
        for (int b = int(blockIdx.x) - first_big_block; b < nb; b += int(gridDim.x) - first_big_block) {
            const int prim = big_list[b];
            const Range r = compute_range(k.dims, BBox(k.bmin, k.bmax), load_tri(tris, prim).bbox());
            const int total = r.size(), first = start_emit[prim], sx = r.hx - r.lx + 1, sy = r.hy - r.ly + 1;
            for (int c = threadIdx.x; c < total; c += kBlock) {
                const int x = r.lx + c % sx, y = r.ly + (c / sx) % sy, z = r.lz + c / (sx * sy);
                pair_rank[first + c] = atomicAdd(refs_per_cell + (x + k.dims.x * (y + k.dims.y * z)), 1);
            }
        }
        return;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    Range r(0, 0, 0, -1, -1, -1);
    int size = 0, start = 0;
    if (i < n) {
        r = compute_range(k.dims, BBox(k.bmin, k.bmax), load_tri(tris, i).bbox());
        size = max(0, r.size());
        start = start_emit[i];
    }
    const bool coop = size >= kCoopCells;
    {   // the lanes walk their (small) ranges in step: pair j of every lane in round j, lanes that name the same cell served together
        const bool mine = size > 0 && !coop;
        // a crowd? the first cell of the wavefront's first primitive, named by another lane's first cell as well (consecutive primitives of a mesh lie next to each
        // other; those of a soup do not, and walk their ranges on their own with plain atomics -- the lock step costs them a fifth: 128 -> 158 us, gpurun_out/r6k)
        const int c0 = mine ? r.lx + k.dims.x * (r.ly + k.dims.y * r.lz) : -1 - lane_id();
        const unsigned long long any = __ballot(mine);
        const bool crowd = any != 0ull && __popcll(__ballot(c0 == __shfl(c0, __ffsll((long long)any) - 1, 64))) >= 2;
        if (!crowd) {
            if (mine) {
                int cur = start;
                for (int z = r.lz; z <= r.hz; z++)
                    for (int y = r.ly; y <= r.hy; y++)
                        for (int x = r.lx; x <= r.hx; x++) pair_rank[cur++] = atomicAdd(refs_per_cell + (x + k.dims.x * (y + k.dims.y * z)), 1);
            }
        } else {
            const int rounds = wave_max(mine ? size : 0);
            int x = r.lx, y = r.ly, z = r.lz;                      // (x fastest, as emit_top_refs walks the range)
            for (int j = 0; j < rounds; j++) {
                const bool act = mine && j < size;
                const int rank = wave_shared_increment(refs_per_cell, act ? x + k.dims.x * (y + k.dims.y * z) : 0, act);
                if (act) pair_rank[start + j] = rank;
                if (++x > r.hx) { x = r.lx; if (++y > r.hy) { y = r.ly; z++; } }
            }
        }
    }
}
// top level, pass 3: emit_new_refs + filter_refs (build.cu:69-157) into the cells' segments: a pair the SAT rejects leaves a HOLE (reference -1) in its
// cell's segment -- the counts that sized the segments are the unfiltered ones, as the reference's depth rule wants them.  One 8-byte store per pair.
__device__ __forceinline__ void place_top_ref(const BuildK& k, const Tri& tri, int prim, int x, int y, int z, int rank, const int* __restrict__ seg_begin,
                                              int2* __restrict__ refs, const int* __restrict__ log_dims, uint32_t* __restrict__ entries) {
    const int inc = 1 << k.shift;
    const ivec3 lo(x << k.shift, y << k.shift, z << k.shift);
    const bool hit = intersect_prim_cell(tri, cell_world_box(k, lo, lo + ivec3(inc)));
    const int cell = x + k.dims.x * (y + k.dims.y * z);
    refs[seg_begin[cell] + rank] = make_int2(hit ? prim : -1, cell);
    if (hit && log_dims[cell] > 0) entries[cell] = 1u;
}
__global__ void __launch_bounds__(kBlock) emit_top_refs(const float4* __restrict__ tris, int n, BuildK k0, const int* __restrict__ shift_dev, vec3 extents,
                                                                const int* __restrict__ start_emit,
                                                                const int* __restrict__ pair_rank, const int* __restrict__ seg_begin, int2* __restrict__ refs,
                                                                const int* __restrict__ log_dims, uint32_t* __restrict__ entries,
                                                                const int* __restrict__ big_list, const int* __restrict__ big_count, int first_big_block) {
    const BuildK k = with_device_shift(k0, shift_dev, extents);
    if (int(blockIdx.x) >= first_big_block) {      // a workgroup per large primitive (count_top_refs)
        const int nb = *big_count;
        for (int b = int(blockIdx.x) - first_big_block; b < nb; b += int(gridDim.x) - first_big_block) {
            const int prim = big_list[b];
            const Tri t = load_tri(tris, prim);
            const Range r = compute_range(k.dims, BBox(k.bmin, k.bmax), t.bbox());
            const int total = r.size(), first = start_emit[prim], sx = r.hx - r.lx + 1, sy = r.hy - r.ly + 1;
            for (int c = threadIdx.x; c < total; c += kBlock)
                place_top_ref(k, t, prim, r.lx + c % sx, r.ly + (c / sx) % sy, r.lz + c / (sx * sy), pair_rank[first + c], seg_begin, refs, log_dims, entries);
        }
        return;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    Range r(0, 0, 0, -1, -1, -1);
    int size = 0, start = 0;
    Tri tri;
    if (i < n) {
        tri = load_tri(tris, i);
        r = compute_range(k.dims, BBox(k.bmin, k.bmax), tri.bbox());
        size = max(0, r.size());
        start = start_emit[i];
    }
    const bool coop = size >= kCoopCells;
    if (size > 0 && !coop) {
        int cur = start;
        for (int z = r.lz; z <= r.hz; z++)
            for (int y = r.ly; y <= r.hy; y++)
                for (int x = r.lx; x <= r.hx; x++) { place_top_ref(k, tri, i, x, y, z, pair_rank[cur], seg_begin, refs, log_dims, entries); cur++; }
    }
}

// the cells of tile t and their references
struct TileRange { int c0, c1; };
__device__ __forceinline__ TileRange tile_cells(const int* __restrict__ tile_first, int t, int num_tiles, int num_cells) {
    return TileRange{ t == 0 ? 0 : tile_first[t], t + 1 >= num_tiles ? num_cells : tile_first[t + 1] };
}
// The lanes of a wavefront hold consecutive references, so the references of a cell are a RUN of lanes: `run` = the lanes of this lane's run from the run's
// first lane on (for the first lane: the whole run), `first` = that lane.
struct LaneRun { unsigned long long run; int first; bool head; };
__device__ __forceinline__ LaneRun lane_run(int cell) {
    const int l = lane_id();
    const int prev = __shfl_up(cell, 1, 64);
    LaneRun r;
    r.head = l == 0 || prev != cell;
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long upto = heads & (~0ull >> (63 - l));                  // heads at or below this lane
    r.first = 63 - __clzll((long long)upto);
    const unsigned long long above = l == 63 ? 0ull : heads >> (l + 1);          // heads above this lane
    const int next = above ? l + 1 + (__ffsll((long long)above) - 1) : 64;
    const unsigned long long below_next = next == 64 ? ~0ull : ((1ull << next) - 1ull);
    r.run = below_next & ~((1ull << r.first) - 1ull);
    return r;
}

#ifndef HG_CLASSIFY_WAVES
#define HG_CLASSIFY_WAVES 8           // (64 registers, two spilled values; at 6 wavefronts per SIMD 77 registers: 1 % slower)
#endif
// compute_split_masks (build.cu:160-216) for the references of splitting cells + the reference count of every child cell (the popcount reduction of
// build.cu:597 and the final sort's histogram: per wavefront and cell by ballots, then per cell in LDS -- no global atomics).  A child that receives references
// and has levels left is marked to split again (compute_dims, build.cu:286-302).  top: segments have holes (references the SAT filter rejected), the leaves'
// counts are taken here as well.  totals[0] += references of cells that do not split (kept).
__global__ void __launch_bounds__(kBlock, HG_CLASSIFY_WAVES) classify_refs(const int2* __restrict__ refs, const int* __restrict__ seg_begin,
                                                           int num_cells, int num_refs, const int* __restrict__ tile_first, int num_tiles, int top,
                                                           const float4* __restrict__ tris, const Cell* __restrict__ cells, const uint32_t* __restrict__ entries, BuildK k,
                                                           unsigned char* __restrict__ masks, int* __restrict__ cell_counts,
                                                           Cell* __restrict__ new_cells, int* __restrict__ new_counts, uint32_t* __restrict__ new_entries, int* __restrict__ totals) {
    __shared__ int hist[kChunkCells * 9];
    __shared__ int lds[kWaves];
    int kept = 0;
    for (int t = xcd_block(blockIdx.x, gridDim.x); t < num_tiles; t += gridDim.x) {
        const TileRange tr = tile_cells(tile_first, t, num_tiles, num_cells);
        for (int cc = tr.c0; cc < tr.c1; cc += kChunkCells) {
            const int ce = min(cc + kChunkCells, tr.c1);
            const int r0 = seg_begin[cc], r1 = ce >= num_cells ? num_refs : seg_begin[ce];
            if (r0 == r1) continue;                                    // (uniform: cells without references have nothing to count -- their children's counts are zero already)
            for (int j = threadIdx.x; j < (ce - cc) * 9; j += kBlock) hist[j] = 0;
            __syncthreads();
            for (int base = r0 + wave_id() * 64; base < r1; base += kBlock) {      // (whole wavefronts: the ballots below)
                const int i = base + lane_id();
                int2 rc = make_int2(-1, 0x7fffffff);
                if (i < r1) rc = refs[i];
                const int ref = rc.x, c = rc.y;
                int m = 0;
                bool keeps = false;
                if (ref >= 0) {
                    const uint32_t e = entries[c];
                    if ((e & 3u) == 0) keeps = true;
                    else {
                        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(c);
                        const int4 a = p[0], b = p[1];
                        m = split_mask(k, ivec3(a.x, a.y, a.z), ivec3(b.x, b.y, b.z), load_tri(tris, ref));
                    }
                }
                if (i < r1) masks[i] = (unsigned char)m;
                kept += keeps ? 1 : 0;
                const LaneRun run = lane_run(c);
                int* h = hist + (c - cc) * 9;
                const unsigned long long any = __ballot(m != 0);
                if (any) {
#pragma unroll
                    for (int child = 0; child < 8; child++) {
                        const unsigned long long with = __ballot((m >> child) & 1);
                        if (run.head && i < r1) { const int n = __popcll(with & run.run); if (n) atomicAdd(h + child, n); }
                    }
                }
                if (top) {
                    const unsigned long long with = __ballot(keeps);
                    if (run.head && i < r1) { const int n = __popcll(with & run.run); if (n) atomicAdd(h + 8, n); }
                }
            }
            __syncthreads();
            for (int j = threadIdx.x; j < (ce - cc) * 8; j += kBlock) {
                const int c = cc + (j >> 3), child = j & 7;
                const uint32_t e = entries[c];
                if (e & 3u) {
                    // emit_new_cells (build.cu:354-383) as well: eight lanes write the eight children of a cell, 256 contiguous bytes (a cell splits because it holds
                    // references, so every splitting cell passes here); a.w: the levels the cell may still split (emit_top_cells)
                    const int n = hist[(j >> 3) * 9 + child];
                    const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(c);
                    const int4 a = p[0], b = p[1];
                    const int inc = (b.x - a.x) >> 1;
                    const ivec3 lo(a.x + (child & 1) * inc, a.y + ((child >> 1) & 1) * inc, a.z + (child >> 2) * inc);
                    store_cell(new_cells, int(e >> 2) + child, lo, a.w - 1, lo + ivec3(inc), 0);
                    new_counts[(e >> 2) + child] = n;
                    new_entries[(e >> 2) + child] = (n > 0 && a.w > 1) ? 1u : 0u;
                } else if (top && child == 0) cell_counts[c] = hist[(j >> 3) * 9 + 8];
            }
            __syncthreads();
        }
    }
    kept = block_sum(kept, lds);
    if (threadIdx.x == 0 && kept) atomicAdd(totals, kept);
}

// split_refs (build.cu:219-243) into the segments of the child cells.  The children of a tile's cells are consecutive cells, so the tile's output is one
// contiguous run of the new array: slots from LDS cursors (one per child cell, starting at the child's segment; one returning LDS atomic per wavefront, cell and
// child, the lanes of a run take consecutive slots), output staged in LDS and written in full lines (lane by lane at scattered offsets the same data costs
// several times its size in HBM write traffic); what a chunk emits beyond the staging area goes out directly.
__global__ void __launch_bounds__(kBlock) emit_child_refs(const int2* __restrict__ refs, const int* __restrict__ seg_begin,
                                                                  int num_cells, int num_refs, const int* __restrict__ tile_first, int num_tiles,
                                                                  const unsigned char* __restrict__ masks, const uint32_t* __restrict__ entries,
                                                                  const int* __restrict__ new_seg_begin, int2* __restrict__ new_refs) {
    __shared__ int cursor[kEmitChunk * 8];
    __shared__ int2 stage[kStage];
    __shared__ int region[2];                        // first output slot of the chunk, one past its last
    for (int t = blockIdx.x; t < num_tiles; t += gridDim.x) {
        const TileRange tr = tile_cells(tile_first, t, num_tiles, num_cells);
        for (int cc = tr.c0; cc < tr.c1; cc += kEmitChunk) {
            const int ce = min(cc + kEmitChunk, tr.c1);
            const int r0 = seg_begin[cc], r1 = ce >= num_cells ? num_refs : seg_begin[ce];
            if (r0 == r1) continue;
            if (threadIdx.x == 0) { region[0] = 0x7fffffff; region[1] = 0; }
            __syncthreads();
            int lo = 0x7fffffff;
            for (int j = threadIdx.x; j < (ce - cc) * 8; j += kBlock) {
                const uint32_t e = entries[cc + (j >> 3)];
                const int at = (e & 3u) ? new_seg_begin[(e >> 2) + (j & 7)] : 0x7fffffff;
                cursor[j] = at;
                lo = min(lo, at);
            }
            lo = -wave_max(-lo);
            if (lane_id() == 0 && lo != 0x7fffffff) atomicMin(&region[0], lo);
            __syncthreads();
            const int base = region[0];
            int hi = 0;
            for (int rb = r0 + wave_id() * 64; rb < r1; rb += kBlock) {             // (whole wavefronts: the ballots below)
                const int i = rb + lane_id();
                int m = 0;
                int2 rc = make_int2(-1, 0x7fffffff);
                if (i < r1) { m = masks[i]; rc = refs[i]; }
                if (__ballot(m != 0) == 0ull) continue;
                const int c = rc.y;
                const int first_child = m ? int(entries[c] >> 2) : 0;
                const LaneRun run = lane_run(c);
                int* cur = cursor + (c - cc) * 8;
                const unsigned long long below = (1ull << lane_id()) - 1ull;
#pragma unroll
                for (int child = 0; child < 8; child++) {
                    const unsigned long long with = __ballot((m >> child) & 1);
                    if (!with) continue;
                    int at = 0;
                    if (run.head && i < r1) { const int n = __popcll(with & run.run); if (n) at = atomicAdd(cur + child, n); }
                    at = __shfl(at, run.first, 64);
                    if ((m >> child) & 1) {
                        const int slot = at + __popcll(with & run.run & below);
                        hi = max(hi, slot + 1);
                        const int2 v = make_int2(rc.x, first_child + child);
                        if (slot - base < kStage) stage[slot - base] = v;
                        else new_refs[slot] = v;
                    }
                }
            }
            hi = wave_max(hi);
            if (lane_id() == 0 && hi) atomicMax(&region[1], hi);
            __syncthreads();
            const int staged = min(region[1] - base, kStage);
            for (int q = threadIdx.x; q < staged; q += kBlock) new_refs[base + q] = stage[q];
            __syncthreads();
        }
    }
}

// copy_cells + copy_entries + compute_cell_ranges (build.cu:407-468) + copy_refs + remap_refs + the sort (:634-647, :681, :691) of a grouped level in ONE pass: a
// leaf's list is its segment, copied behind the lists of the leaves before it and put in ascending order by the thread that copies it (lists hold a handful
// of references).  top: the holes of a segment (references the SAT filter rejected) are skipped.
constexpr int kCoopListMin = 9, kCoopListMax = 1024;        // lists of this many references are put in order by a wavefront (concat_level), longer ones by one thread
__global__ void __launch_bounds__(kBlock) concat_level(const uint32_t* __restrict__ entries, const Cell* __restrict__ cells, const int* __restrict__ cell_counts,
                                                               const int* __restrict__ start_cell, const int* __restrict__ ref_begin, int num_cells, int level_off,
                                                               const int2* __restrict__ refs, const int* __restrict__ seg_begin, int num_refs, int top,
                                                               Cell* __restrict__ out_cells, uint32_t* __restrict__ out_entries, int* __restrict__ out_refs) {
    // Lists of more than eight references (a mesh: the stadium's deepest level averages six per cell, its grain of dust hundreds) are left to the WAVEFRONTS of the
    // workgroup, one list at a time: a thread that sorts 20 references in global memory by insertion while its neighbours sort two keeps its whole wavefront waiting
    // (concat_level 1.29 ms for the stadium's last level, 0.05 ms for the soup's: round 6, gpurun_out/r6i).
    __shared__ int long_cells[kBlock];
    __shared__ int num_long;
    __shared__ int stage[kWaves][kCoopListMax];
    if (threadIdx.x == 0) num_long = 0;
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < num_cells) {
        const uint32_t e = entries[i];
        if (e & 3u) out_entries[level_off + i] = (e & 3u) | (((e >> 2) + uint32_t(level_off + num_cells)) << 2);
        else {
            const int dst = start_cell[i], cnt = cell_counts[i], rb = ref_begin[i];
            const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
            const int4 a = p[0], b = p[1];
            store_cell(out_cells, dst, ivec3(a.x, a.y, a.z), cnt ? rb : 0, ivec3(b.x, b.y, b.z), cnt ? rb + cnt : 0);
            out_entries[level_off + i] = uint32_t(dst) << 2;
            int* r = out_refs + rb;
            const int2* src = refs + seg_begin[i];
            if (cnt == 0) { }
            else if (!top && cnt <= 4) {                        // the common list: four loads, a sorting network in registers, no list read back from memory
                const int big = 0x7fffffff;
                int v0 = src[0].x, v1 = cnt > 1 ? src[1].x : big, v2 = cnt > 2 ? src[2].x : big, v3 = cnt > 3 ? src[3].x : big;
                int t;
                if (v0 > v1) { t = v0; v0 = v1; v1 = t; }
                if (v2 > v3) { t = v2; v2 = v3; v3 = t; }
                if (v0 > v2) { t = v0; v0 = v2; v2 = t; }
                if (v1 > v3) { t = v1; v1 = v3; v3 = t; }
                if (v1 > v2) { t = v1; v1 = v2; v2 = t; }
                r[0] = v0;
                if (cnt > 1) r[1] = v1;
                if (cnt > 2) r[2] = v2;
                if (cnt > 3) r[3] = v3;
            } else if (cnt >= kCoopListMin && cnt <= kCoopListMax) {
                long_cells[atomicAdd(&num_long, 1)] = i;
            } else {
                int n = 0;
                if (top) { for (int j = 0; n < cnt; j++) { const int v = src[j].x; if (v >= 0) r[n++] = v; } }
                else for (; n < cnt; n++) r[n] = src[n].x;
                if (cnt > 24) {                                  // (Shell passes for the rare list beyond what a wavefront stages)
                    const int gaps[8] = { 1750, 701, 301, 132, 57, 23, 10, 4 };
                    for (int g = 0; g < 8; g++) {
                        const int gap = gaps[g];
                        for (int x = gap; x < cnt; x++) {
                            const int v = r[x];
                            int y = x - gap;
                            while (y >= 0 && r[y] > v) { r[y + gap] = r[y]; y -= gap; }
                            r[y + gap] = v;
                        }
                    }
                }
                for (int x = 1; x < cnt; x++) {
                    const int v = r[x];
                    int y = x - 1;
                    while (y >= 0 && r[y] > v) { r[y + 1] = r[y]; y--; }
                    r[y + 1] = v;
                }
            }
        }
    }
    __syncthreads();
    // one list per wavefront: staged in LDS (top level: without the holes the SAT filter left), every reference placed by its RANK -- the ids of a list are distinct
    // (a primitive is referenced once per cell), so the rank is the number of smaller ids
    const int nl = num_long, lane = lane_id();
    int* buf = stage[wave_id()];
    for (int q = wave_id(); q < nl; q += kWaves) {
        const int c = long_cells[q];
        const int cnt = cell_counts[c];
        int* r = out_refs + ref_begin[c];
        const int2* src = refs + seg_begin[c];
        if (top) {
            int found = 0;
            for (int base = 0; found < cnt; base += 64) {       // (the segment holds cnt references that passed: the loop ends)
                const int v = seg_begin[c] + base + lane < num_refs ? src[base + lane].x : -1;
                const unsigned long long ok = __ballot(v >= 0);
                const int at = found + __popcll(ok & ((1ull << lane) - 1ull));
                if (v >= 0 && at < cnt) buf[at] = v;
                found += __popcll(ok);
            }
        } else {
            for (int x = lane; x < cnt; x += 64) buf[x] = src[x].x;
        }
        __builtin_amdgcn_wave_barrier();
        for (int x = lane; x < cnt; x += 64) {
            const int v = buf[x];
            int rank = 0;
            for (int y = 0; y < cnt; y++) rank += buf[y] < v ? 1 : 0;
            r[rank] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct PlainIn { const int* v; __device__ int operator()(int i) const { return v[i]; } };
struct PlainOut { int* v; __device__ void operator()(int i, int s) const { v[i] = s; } };

using Temps = PoolTemps;

// ---- the level loop and the concatenation (host side) --------------------------------------------------------------------------------------------------
struct GLevel {
    int2* refs = nullptr; int num_refs = 0;     // {reference, cell}, grouped by cell, in cell order
    Cell* cells = nullptr; uint32_t* entries = nullptr; int num_cells = 0;
    int* cell_counts = nullptr;                 // references per cell (leaves: the length of the list)
    int* seg_begin = nullptr;                   // first reference of every cell
    int* tile_first = nullptr; int num_tiles = 0;
    int* start_cell = nullptr; int* ref_begin = nullptr;
};

// The temporaries of the level loop come from ONE pool buffer, sized by what the last construction of this context needed (what does not fit comes from the
// pool as before): three dozen requests of sizes that differ from level to level leave the pool's cached slots unsettled for a dozen constructions (a
// request nothing fits re-grows the largest free slot), each paying hipFree + hipMalloc: build_grid 1.43 instead of 1.24 ms until then.
struct Arena {
    hagrid_ctx* ctx; Temps& tmp; char* base = nullptr; size_t size = 0, used = 0, wanted = 0; int num_tris;
    Arena(hagrid_ctx* c, Temps& t, int n_tris) : ctx(c), tmp(t), num_tris(n_tris) {
        // the hint belongs to a scene of build_arena_tris primitives: a much smaller scene takes its share of it (a small construction after a large one must not
        // allocate the large one's buffer); the buffer is OPTIONAL -- without it every request goes to the pool, and a failed hipMalloc must leave no error behind
        size_t want = c->build_arena_hint;
        if (want && c->build_arena_tris > 0 && 2ll * n_tris < c->build_arena_tris)
            want = size_t(double(want) * 1.25 * double(std::max(n_tris, 1)) / double(c->build_arena_tris)) + 4096;
        if (want) {
            base = pool_try_alloc<char>(c, want);
            if (base) { t.ptrs.push_back(base); size = want; }
        }
    }
    ~Arena() { ctx->build_arena_hint = wanted; ctx->build_arena_tris = num_tris; }
    template <typename T> T* get(size_t n) {
        const size_t bytes = (std::max(n, size_t(1)) * sizeof(T) + 255) & ~size_t(255);
        wanted += bytes;
        if (base && used + bytes <= size) { T* p = reinterpret_cast<T*>(base + used); used += bytes; return p; }
        return tmp.get<T>(n);
    }
    void drop(void* p) { if (!base || p < base || p >= base + size) tmp.drop(p); }
};

int build_levels(hagrid_ctx* ctx, const float4* tris, int num_tris, hagrid_grid* grid, BuildK k, const BBox& gb, float snd_density, Temps& tmp) {
    hipStream_t st = ctx->stream;
    int* dsc = ctx->dscratch;
    Arena ar(ctx, tmp, num_tris);
    const ivec3 dims = k.dims;
    const int num_top = dims.x * dims.y * dims.z;

    // ---- reference counts, per-cell depth, shift (first_build_iter, build.cu:470-512) ----
    int* counts = ar.get<int>(size_t(num_tris));
    int* start_emit = ar.get<int>(size_t(num_tris));
    int* refs_per_cell = ar.get<int>(size_t(num_top));
    int* log_dims = ar.get<int>(size_t(num_top));
    int* const partials = nullptr;               // (the look-back scans need none)
    if (!counts || !start_emit || !refs_per_cell || !log_dims) return HAGRID_ENOMEM;
    HG_HIP(ctx, hipMemsetAsync(refs_per_cell, 0, size_t(num_top) * sizeof(int), st));
    int* big_list = ar.get<int>(size_t(num_tris));               // primitives of kCoopCells cells and more (top_range_sizes); their number in dsc[2]
    if (!big_list) return HAGRID_ENOMEM;
    top_range_sizes<<<grid_blocks(num_tris, kBlock), kBlock, 0, st>>>(tris, num_tris, k, counts, big_list, dsc + 2); HG_DBG(ctx);
    if (!ctx_scan<int>(ctx, PlainIn{counts}, PlainOut{start_emit}, num_tris, partials, (const int*)nullptr, dsc + 0)) return HAGRID_ENOMEM;
    int R0 = 0, num_big = 0;
    {
        int h[3];
        HG_TRY(read_back(ctx, dsc, h, sizeof(h)));
        R0 = h[0]; num_big = h[2];
    }
    if (R0 < 0 || R0 > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many top-level references");
    ar.drop(counts);
    int* pair_rank = ar.get<int>(size_t(R0) + 1);
    if (!pair_rank) return HAGRID_ENOMEM;
    const int prim_blocks = grid_blocks(num_tris, kBlock), big_blocks = std::min(num_big, 2048);      // (the large primitives' workgroups ride behind the others' in the same launches)
    count_top_refs<<<prim_blocks + big_blocks, kBlock, 0, st>>>(tris, num_tris, k, start_emit, refs_per_cell, pair_rank, big_list, dsc + 2, prim_blocks); HG_DBG(ctx);
    top_log_dims<<<grid_blocks(num_top, kBlock), kBlock, 0, st>>>(refs_per_cell, num_top, k, snd_density, log_dims, dsc + 1); HG_DBG(ctx);

    std::vector<GLevel> levels;
    GLevel L;
    L.num_refs = R0; L.num_cells = num_top;
    L.num_tiles = grid_blocks(R0, kTileRefs);
    L.refs = ar.get<int2>(size_t(R0) + 1);
    L.cells = ar.get<Cell>(size_t(num_top)); L.entries = ar.get<uint32_t>(size_t(num_top) + 1);
    L.cell_counts = ar.get<int>(size_t(num_top)); L.seg_begin = ar.get<int>(size_t(num_top) + 1);
    L.tile_first = ar.get<int>(size_t(L.num_tiles) + 2);
    L.start_cell = ar.get<int>(size_t(num_top)); L.ref_begin = ar.get<int>(size_t(num_top));
    if (!L.refs || !L.cells || !L.entries || !L.cell_counts || !L.seg_begin || !L.tile_first || !L.start_cell || !L.ref_begin) return HAGRID_ENOMEM;
    if (!ctx_scan<int>(ctx, SegIn{refs_per_cell}, SegOut{refs_per_cell, L.seg_begin, L.tile_first, L.cells, L.entries, L.cell_counts, log_dims, dsc + 1, dims}, num_top, partials,
                       (const int*)nullptr, (int*)nullptr)) return HAGRID_ENOMEM;
    hagrid_build_counts& bc = ctx->counts;
    memset(&bc, 0, sizeof(bc));
    bc.num_tris = num_tris; bc.top_cells = num_top; bc.top_refs = R0;
    emit_top_refs<<<prim_blocks + big_blocks, kBlock, 0, st>>>(tris, num_tris, k, dsc + 1, gb.extents(), start_emit, pair_rank, L.seg_begin, L.refs, log_dims, L.entries, big_list, dsc + 2, prim_blocks); HG_DBG(ctx);
    levels.push_back(L);

    // ---- subdivision, one level per iteration (build_iter, build.cu:527-619) ----
    // tot (four words per level, zeroed by the caller): {cells of the next level, kept references of this level, {references, cells} of the level after the
    // next: the total of the scan over the new cells}
    {
        if (!ctx_scan<int>(ctx, ChildCountIn{L.entries}, UpdateEntriesOut{L.entries}, num_top, (int*)nullptr, (const int*)nullptr, dsc + 8)) return HAGRID_ENOMEM;
    }
    int num_new_cells = 0, shift = 0;
    {   // the shift (dsc[1], top_log_dims) and the cells of the next level (dsc[8]) in one round trip
        int h[8];
        HG_TRY(read_back(ctx, dsc + 1, h, sizeof(h)));
        shift = h[0]; num_new_cells = h[7];
    }
    if (shift >= 24) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many levels");
    k.shift = shift;
    k.cell_size = gb.extents() / vec3(dims << shift);
    ar.drop(start_emit); ar.drop(pair_rank); ar.drop(refs_per_cell);
    for (int level = 0;; level++) {
        GLevel& P = levels.back();
        int* tot = dsc + 8 + 4 * level;
        if (num_new_cells < 0) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: level too large");
        if ((int)levels.size() >= 24) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many levels");
        const bool last = num_new_cells == 0;
        if (last && level > 0) {
            // no cell of this level splits: its references stay where they are, every count is known (build.cu:583-587)
            if (level < HAGRID_MAX_LEVELS) { bc.level_refs[level] = P.num_refs; bc.level_cells[level] = P.num_cells; bc.level_kept[level] = P.num_refs; bc.num_levels = level + 1; }
            break;
        }
        GLevel N;
        N.num_cells = num_new_cells;
        unsigned char* masks = ar.get<unsigned char>(size_t(P.num_refs) + 1);
        if (!masks) return HAGRID_ENOMEM;
        if (!last) {
            N.cells = ar.get<Cell>(size_t(num_new_cells)); N.entries = ar.get<uint32_t>(size_t(num_new_cells) + 1);
            N.cell_counts = ar.get<int>(size_t(num_new_cells)); N.seg_begin = ar.get<int>(size_t(num_new_cells) + 1);
            N.tile_first = ar.get<int>(size_t(grid_blocks(8ll * P.num_refs, kTileRefs)) + 2);
            N.start_cell = ar.get<int>(size_t(num_new_cells)); N.ref_begin = ar.get<int>(size_t(num_new_cells));
            if (!N.cells || !N.entries || !N.cell_counts || !N.seg_begin || !N.tile_first || !N.start_cell || !N.ref_begin) return HAGRID_ENOMEM;
        }
        if (P.num_refs > 0) {
            classify_refs<<<std::min(P.num_tiles, 4096), kBlock, 0, st>>>(P.refs, P.seg_begin, P.num_cells, P.num_refs, P.tile_first, P.num_tiles, level == 0 ? 1 : 0,
                                                                             tris, P.cells, P.entries, k, masks, P.cell_counts, N.cells, N.cell_counts, N.entries, tot + 1); HG_DBG(ctx);
        }
        Int2* next = reinterpret_cast<Int2*>(tot + 2);
        if (!last) {
            if (!ctx_scan<Int2>(ctx, ChildSegIn{N.cell_counts, N.entries}, ChildSegOut{N.cell_counts, N.entries, N.seg_begin, N.tile_first}, num_new_cells, (Int2*)nullptr,
                                (const Int2*)nullptr, next)) return HAGRID_ENOMEM;
        }
        int h3[3] = {0, 0, 0};                        // kept, references of the next level, cells of the level after it
        HG_TRY(read_back(ctx, tot + 1, h3, sizeof(h3)));
        if (level < HAGRID_MAX_LEVELS) { bc.level_refs[level] = P.num_refs; bc.level_cells[level] = P.num_cells; bc.level_kept[level] = h3[0]; bc.num_levels = level + 1; }
        if (last) { ar.drop(masks); break; }
        const int num_children = h3[1];
        if (num_children < 0 || num_children > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: level too large");
        N.num_refs = num_children; N.num_tiles = grid_blocks(num_children, kTileRefs);
        N.refs = ar.get<int2>(size_t(num_children) + 1);
        if (!N.refs) return HAGRID_ENOMEM;
        if (P.num_refs > 0 && num_children > 0) {
            emit_child_refs<<<std::min(P.num_tiles, 4096), kBlock, 0, st>>>(P.refs, P.seg_begin, P.num_cells, P.num_refs, P.tile_first, P.num_tiles,
                                                                                   masks, P.entries, N.seg_begin, N.refs); HG_DBG(ctx);
        }
        ar.drop(masks);
        num_new_cells = h3[2];
        levels.push_back(N);
    }
    ar.drop(log_dims);

    // ---- concat_levels (build.cu:621-716) ----
    const int num_levels = (int)levels.size();
    long long total_cells_ll = 0;
    int max_cells = 0;
    for (auto& V : levels) { total_cells_ll += V.num_cells; max_cells = std::max(max_cells, V.num_cells); }
    if (total_cells_ll > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many voxel map entries");
    const int total_cells = int(total_cells_ll);
    Int2* const part2 = nullptr;                           // (the look-back scans need no partials)
    Int2* carry = reinterpret_cast<Int2*>(dsc + 128);      // one Int2 per level, chained on the device
    for (int l = 0; l < num_levels; l++) {
        GLevel& V = levels[l];
        if (!ctx_scan<Int2>(ctx, LeafIn{V.entries, V.cell_counts}, LeafOut{V.start_cell, V.ref_begin}, V.num_cells, part2,
                            l ? carry + (l - 1) : (const Int2*)nullptr, carry + l)) return HAGRID_ENOMEM;
    }
    int hf[2];
    HG_TRY(read_back(ctx, carry + (num_levels - 1), hf, sizeof(hf)));
    const int new_total_cells = hf[0], total_refs = hf[1];
    if (new_total_cells <= 0 || total_refs < 0) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: inconsistent totals");

    Cell* out_cells = pool_alloc<Cell>(ctx, size_t(new_total_cells));
    uint32_t* out_entries = pool_alloc<uint32_t>(ctx, size_t(total_cells));
    int* out_refs = pool_alloc<int>(ctx, size_t(total_refs));
    if (!out_cells || !out_entries || !out_refs) {
        hagrid_mem_free(ctx, out_cells); hagrid_mem_free(ctx, out_entries); hagrid_mem_free(ctx, out_refs);
        return HAGRID_ENOMEM;
    }
    for (int l = 0, off = 0; l < num_levels; off += levels[l].num_cells, l++) {
        GLevel& V = levels[l];
        concat_level<<<grid_blocks(V.num_cells, kBlock), kBlock, 0, st>>>(V.entries, V.cells, V.cell_counts, V.start_cell, V.ref_begin, V.num_cells, off,
                                                                                  V.refs, V.seg_begin, V.num_refs, l == 0 ? 1 : 0, out_cells, out_entries, out_refs); HG_DBG(ctx);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // temporaries are released by the caller's Temps
    if (e != hipSuccess) {
        hagrid_mem_free(ctx, out_cells); hagrid_mem_free(ctx, out_entries); hagrid_mem_free(ctx, out_refs);
        HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e));
    }
    memset(grid, 0, sizeof(*grid));
    grid->entries = out_entries; grid->ref_ids = out_refs; grid->cells = out_cells; grid->small_cells = nullptr;
    grid->bbox_min[0] = gb.min.x; grid->bbox_min[1] = gb.min.y; grid->bbox_min[2] = gb.min.z;
    grid->bbox_max[0] = gb.max.x; grid->bbox_max[1] = gb.max.y; grid->bbox_max[2] = gb.max.z;
    grid->dims[0] = dims.x; grid->dims[1] = dims.y; grid->dims[2] = dims.z;
    grid->num_cells = new_total_cells; grid->num_entries = total_cells; grid->num_refs = total_refs;
    bc.build_cells = new_total_cells; bc.build_entries = total_cells; bc.build_refs = total_refs;
    grid->shift = shift;                         // the cell-coordinate shift
    grid->num_offsets = shift + 1;
    for (int i = 0, off = 0; i <= shift; i++) {  // build.cu:711-715, padded when the deepest level is empty
        if (i < num_levels) off += levels[i].num_cells;
        grid->offsets[i] = off;
    }
    return HAGRID_OK;
}

} // namespace

extern "C" int hagrid_build_grid(hagrid_ctx* ctx, const void* tris_v, int num_tris, hagrid_grid* grid,
                                 float top_density, float snd_density) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
    if (!tris_v || num_tris <= 0) HG_FAIL(ctx, HAGRID_EINVAL, "build_grid: no triangles");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    const float4* tris = static_cast<const float4*>(tris_v);
    hipStream_t st = ctx->stream;
    Temps tmp(ctx);
    int* dsc = ctx->dscratch;                       // device scalars
    HG_HIP(ctx, hipMemsetAsync(dsc, 0, 256 * sizeof(int), st));

    // ---- scene box, top-level resolution (build.cu:725-740) ----
    const int bb_blocks = std::min(grid_blocks(num_tris, kBlock), 1024);
    float* bb_part = tmp.get<float>(size_t(bb_blocks) * 6 + 8);
    if (!bb_part) return HAGRID_ENOMEM;
    float* bb_out = bb_part + size_t(bb_blocks) * 6;
    bbox_partials<<<bb_blocks, kBlock, 0, st>>>(tris, num_tris, bb_part); HG_DBG(ctx);
    bbox_final<<<1, kBlock, 0, st>>>(bb_part, bb_blocks, bb_out); HG_DBG(ctx);
    float hb[6];
    HG_TRY(read_back(ctx, bb_out, hb, sizeof(hb)));
    BBox gb(vec3(hb[0], hb[1], hb[2]), vec3(hb[3], hb[4], hb[5]));
    ivec3 dims = compute_grid_dims(gb, num_tris, top_density);
    dims.x += dims.x & 1; dims.y += dims.y & 1; dims.z += dims.z & 1;      // even: 8-entry blocks stay 32 B aligned
    const vec3 ext = gb.extents();
    gb.min -= ext * 0.001f;
    gb.max += ext * 0.001f;
    const long long num_top_ll = (long long)dims.x * dims.y * dims.z;
    if (num_top_ll > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: top-level grid too large");

    BuildK k;
    k.dims = dims; k.shift = 0; k.bmin = gb.min; k.bmax = gb.max; k.cell_size = vec3(0.0f);

    return build_levels(ctx, tris, num_tris, grid, k, gb, snd_density, tmp);
}

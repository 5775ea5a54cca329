// build.hip -- hagrid_build_grid: construction of the initial (octree-structured) irregular grid on gfx950.
//
// Replaces the reference's build.cu (build :718-758, first_build_iter :470-525, build_iter :527-619,
// concat_levels :621-716 and their 20 kernels) and the CUB calls behind them (parallel.cuh).  The result
// -- entries, cells, ref_ids, dims, bbox, shift, offsets -- is bit-identical to the CPU oracle's.
//
// How it differs from the reference's pipeline (not a port):
//  * Only the ORDER OF CELLS is part of the result; the final reference list of a cell is the ascending
//    list of primitive ids.  So references never go through ordered scans, a flagged partition or a radix
//    sort: kept and split references are appended with one atomic per wavefront (wave_append), per-cell
//    reference counts are accumulated with atomics while the levels are built, the final scatter uses a
//    per-cell cursor, and each (short) list is sorted in place.  Ordered scans run over cells only.
//  * Per level there is ONE host round trip (new cells, children, kept) instead of four; the bounding box,
//    the reference count and the shift cost two more; concat costs one.
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

__global__ void __launch_bounds__(kBlock) count_top_refs(const float4* __restrict__ tris, int n, BuildK k,
                                                         int* __restrict__ counts, int* __restrict__ refs_per_cell) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    Range r(0, 0, 0, -1, -1, -1);
    int size = 0;
    if (i < n) {
        r = compute_range(k.dims, BBox(k.bmin, k.bmax), load_tri(tris, i).bbox());
        size = max(0, r.size());
        counts[i] = size;
    }
    const bool coop = size >= kCoopCells;
    if (size > 0 && !coop)
        for (int z = r.lz; z <= r.hz; z++)
            for (int y = r.ly; y <= r.hy; y++)
                for (int x = r.lx; x <= r.hx; x++)
                    atomicAdd(refs_per_cell + (x + k.dims.x * (y + k.dims.y * z)), 1);
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int lx = __shfl(r.lx, src, 64), ly = __shfl(r.ly, src, 64), lz = __shfl(r.lz, src, 64);
        const int sx = __shfl(r.hx, src, 64) - lx + 1, sy = __shfl(r.hy, src, 64) - ly + 1, total = __shfl(size, src, 64);
        for (int c = lane_id(); c < total; c += 64) {
            const int x = lx + c % sx, y = ly + (c / sx) % sy, z = lz + c / (sx * sy);
            atomicAdd(refs_per_cell + (x + k.dims.x * (y + k.dims.y * z)), 1);
        }
    }
}

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

// emit_new_refs (build.cu:69-136) + filter_refs (build.cu:139-157): every (primitive, top cell) pair of the
// primitive's cell range in x-fastest order, with -1/-1 where the triangle misses the cell.  Large ranges are spread
// over the wavefront as in count_top_refs (slot = start + linear cell index, so the order is the serial one).
// A cell that receives a reference and still has levels to go is marked for splitting by whoever hands it the reference
// (compute_dims, build.cu:286-302, does that in a pass of its own over the references): entry word 1 = make_entry(1, 0), the same
// value from every writer.
__device__ __forceinline__ void emit_one_top_ref(const BuildK& k, const Tri& tri, int prim, int x, int y, int z, int slot,
                                                 int* __restrict__ ref_ids, int* __restrict__ cell_ids,
                                                 const int* __restrict__ log_dims, uint32_t* __restrict__ entries) {
    const int inc = 1 << k.shift;
    const ivec3 lo(x << k.shift, y << k.shift, z << k.shift);
    const bool hit = intersect_prim_cell(tri, cell_world_box(k, lo, lo + ivec3(inc)));
    const int cell = x + k.dims.x * (y + k.dims.y * z);
    ref_ids[slot] = hit ? prim : -1;
    cell_ids[slot] = hit ? cell : -1;
    if (hit && log_dims[cell] > 0) entries[cell] = 1u;
}

__global__ void __launch_bounds__(kBlock) emit_top_refs(const float4* __restrict__ tris, int n, BuildK k, const int* __restrict__ start_emit,
                                                        int* __restrict__ ref_ids, int* __restrict__ cell_ids,
                                                        const int* __restrict__ log_dims, uint32_t* __restrict__ entries) {
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
                for (int x = r.lx; x <= r.hx; x++) emit_one_top_ref(k, tri, i, x, y, z, cur++, ref_ids, cell_ids, log_dims, entries);
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int prim = __shfl(i, src, 64), first = __shfl(start, src, 64), total = __shfl(size, src, 64);
        const int lx = __shfl(r.lx, src, 64), ly = __shfl(r.ly, src, 64), lz = __shfl(r.lz, src, 64);
        const int sx = __shfl(r.hx, src, 64) - lx + 1, sy = __shfl(r.hy, src, 64) - ly + 1;
        const Tri t = load_tri(tris, prim);                     // same address in every lane: one broadcast load
        for (int c = lane_id(); c < total; c += 64)
            emit_one_top_ref(k, t, prim, lx + c % sx, ly + (c / sx) % sy, lz + c / (sx * sy), first + c, ref_ids, cell_ids, log_dims, entries);
    }
}

// emit_top_cells (build.cu:332-351)
// + the levels the cell may still be split (log_dims, build.cu:256-270; update_log_dims :273-278 becomes "one less per level")
// While a level is under construction the `begin` word of its cells is free: it carries the number of levels the cell may still
// be split, where classify_refs finds it in the record it loads anyway.
// The kernel that creates a level's cells also clears their voxel-map words and reference counts (it runs before the
// references of the level are handed out): no fill launches.
__global__ void __launch_bounds__(kBlock) emit_top_cells(Cell* __restrict__ cells, int num_top, BuildK k, const int* __restrict__ log_dims,
                                                         uint32_t* __restrict__ entries, int* __restrict__ cell_counts) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= num_top) return;
    entries[id] = 0u; cell_counts[id] = 0;
    const int x = id % k.dims.x, y = (id / k.dims.x) % k.dims.y, z = id / (k.dims.x * k.dims.y);
    const ivec3 lo(x << k.shift, y << k.shift, z << k.shift);
    store_cell(cells, id, lo, log_dims[id], lo + ivec3(1 << k.shift), 0);
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

constexpr int kNoRank = int(0x80000000);        // ranks[]: a reference without a cell

// mark_kept_refs (build.cu:305-314) + compute_split_masks + the popcount reduction of build.cu:597, and the
// per-cell reference count that replaces the final sort's histogram.  totals[0] += children, totals[1] += kept.
__global__ void __launch_bounds__(kBlock) classify_refs(const int* __restrict__ ref_ids, const int* __restrict__ cell_ids, int num_refs,
                                                        const float4* __restrict__ tris, const Cell* __restrict__ cells,
                                                        const uint32_t* __restrict__ entries, BuildK k,
                                                        unsigned char* __restrict__ masks, int* __restrict__ cell_counts, int* __restrict__ ranks,
                                                        int* __restrict__ totals) {
    __shared__ int lds[kWaves];
    int children = 0, kept = 0;
    // several rounds per workgroup: the two totals cost one atomic pair per workgroup (a same-word atomic per 256 references
    // would run into the ~88 atomics/us ceiling of a single L2 word).  A workgroup takes a contiguous stretch of the references and the stretches
    // go to the XCDs eighth by eighth (wave_prims.h xcd_block): the references of a cell and of its neighbours -- the same triangles -- meet in one L2.
    const int per = ((num_refs + int(gridDim.x) - 1) / int(gridDim.x) + kBlock - 1) / kBlock * kBlock;
    const long long first = (long long)xcd_block(blockIdx.x, gridDim.x) * per;
    const int last = int(first + per < num_refs ? first + per : num_refs);
    for (int i = int(first < num_refs ? first : num_refs) + threadIdx.x; i < last; i += kBlock) {
        const int c = cell_ids[i];
        int m = 0, r = kNoRank;
        if (c >= 0) {
            const uint32_t e = entries[c];
            if ((e & 3u) == 0) {
                kept++;
                // the count and the reference's slot inside its cell's list in ONE atomic: random atomics run at ~26 per ns on this
                // part whatever their scope or whether they return a value (tools/micro/atomic_scope.hip), so the scatter pass must
                // not pay for a second one per reference
                r = atomicAdd(cell_counts + c, 1);
            } else {
                const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(c);
                const int4 a = p[0], b = p[1];
                m = split_mask(k, ivec3(a.x, a.y, a.z), ivec3(b.x, b.y, b.z), load_tri(tris, ref_ids[i]));
                children += __popc(m);
                // where the cell's children start and whether they split again (a.w: levels left): emit_child_refs need not look
                // anything up
                r = -int(((e >> 2) << 1) | uint32_t(a.w > 1)) - 1;
            }
        }
        ranks[i] = r;                                  // >= 0: slot of a kept reference; < 0: not kept (scatter_kept_refs skips it unseen)
        masks[i] = (unsigned char)m;
    }
    children = block_sum(children, lds);
    kept = block_sum(kept, lds);
    if (threadIdx.x == 0) {
        if (children) atomicAdd(totals + 0, children);
        if (kept) atomicAdd(totals + 1, kept);
    }
}

// split_refs (build.cu:219-243).  Output order is irrelevant (see the header), so slots are handed out per TILE of
// 2048 references: block-wide prefix over the per-thread child counts, one atomic per tile.  Inside a tile every wavefront owns a
// contiguous range and fills it row by row (one reference per lane and row, up to eight children each) through a 4 KB staging
// area in LDS, so that the children leave in full 256-byte runs: written lane by lane, each at its own offset, the same data cost
// 3.6x its size in HBM write traffic (profiles/pmc_r2m_construction_traffic.txt) and the deepest level 142 us instead of ~60.
constexpr int kEmitItems = 8;
__global__ void __launch_bounds__(kBlock) emit_child_refs(const int* __restrict__ ref_ids, const int* __restrict__ cell_ids, int num_refs,
                                                          const unsigned char* __restrict__ masks, const int* __restrict__ ranks,
                                                          uint32_t* __restrict__ new_entries,
                                                          int* __restrict__ new_ref_ids, int* __restrict__ new_cell_ids, int* __restrict__ cursor) {
    __shared__ int lds[kWaves];
    __shared__ int tile_base;
    __shared__ int2 stage[kWaves][64 * 8];              // per wavefront: the children {reference, cell} of one row
    const int tile_size = kBlock * kEmitItems;
    int2* mine = stage[wave_id()];
    for (int base = blockIdx.x * tile_size; base < num_refs; base += gridDim.x * tile_size) {
        int m[kEmitItems];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < kEmitItems; j++) {
            const int i = base + j * kBlock + threadIdx.x;
            m[j] = i < num_refs ? masks[i] : 0;
            cnt += __popc(m[j]);
        }
        const int wave_total = wave_sum(cnt);
        if (lane_id() == 0) lds[wave_id()] = wave_total;
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { if (w < wave_id()) off += lds[w]; total += lds[w]; }
        if (threadIdx.x == 0) tile_base = total ? atomicAdd(cursor, total) : 0;
        __syncthreads();
        int row_base = tile_base + off;                  // this wavefront's range, filled row after row
#pragma unroll
        for (int j = 0; j < kEmitItems; j++) {
            int mm = m[j];
            const int c = __popc(mm);
            const int incl = wave_inclusive_scan(c);
            const int row_total = __shfl(incl, 63, 64);
            if (row_total == 0) continue;                 // (uniform)
            if (mm) {
                const int i = base + j * kBlock + threadIdx.x;
                const int ref = ref_ids[i];
                const int code = -ranks[i] - 1;                            // classify_refs left the children's first cell here
                const int begin = code >> 1;
                const bool splits_again = (code & 1) != 0;                 // the children still have a level to go
                int at = incl - c;
                while (mm) {
                    const int child = __ffs(mm) - 1;
                    mm &= mm - 1;
                    mine[at++] = make_int2(ref, begin + child);
                    if (splits_again) new_entries[begin + child] = 1u;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // LDS operations of a wavefront execute in order: only the
            __builtin_amdgcn_wave_barrier();                                // compiler must not move the reads below above the writes
            for (int q = lane_id(); q < row_total; q += 64) {
                const int2 v = mine[q];
                new_ref_ids[row_base + q] = v.x;
                new_cell_ids[row_base + q] = v.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            row_base += row_total;
        }
        __syncthreads();
    }
}

// emit_new_cells (build.cu:354-383): 8 cells x 32 B = 256 contiguous bytes per split cell
__global__ void __launch_bounds__(kBlock) emit_child_cells(const uint32_t* __restrict__ entries, const Cell* __restrict__ cells, int num_cells,
                                                           Cell* __restrict__ new_cells, uint32_t* __restrict__ new_entries, int* __restrict__ new_cell_counts) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int id = t >> 3, child = t & 7;     // 8 lanes per parent: each lane stores one child
    if (id >= num_cells) return;
    const uint32_t e = entries[id];
    if ((e & 3u) == 0) return;
    const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(id);
    const int4 a = p[0], b = p[1];
    const int inc = (b.x - a.x) >> 1;
    const ivec3 lo(a.x + (child & 1) * inc, a.y + ((child >> 1) & 1) * inc, a.z + (child >> 2) * inc);
    store_cell(new_cells, int(e >> 2) + child, lo, a.w - 1, lo + ivec3(inc), 0);       // (a.w: levels left, see emit_top_cells)
    new_entries[int(e >> 2) + child] = 0u; new_cell_counts[int(e >> 2) + child] = 0;
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

// copy_cells (build.cu:407-419) + copy_entries (:422-440) + compute_cell_ranges (:453-468)
__global__ void __launch_bounds__(kBlock) concat_level(const uint32_t* __restrict__ entries, const Cell* __restrict__ cells, const int* __restrict__ cell_counts,
                                                       const int* __restrict__ start_cell, const int* __restrict__ ref_begin, int num_cells,
                                                       int level_off, Cell* __restrict__ out_cells, uint32_t* __restrict__ out_entries) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_cells) return;
    const uint32_t e = entries[i];
    if ((e & 3u) == 0) {
        const int dst = start_cell[i], cnt = cell_counts[i], rb = ref_begin[i];
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
        const int4 a = p[0], b = p[1];
        store_cell(out_cells, dst, ivec3(a.x, a.y, a.z), cnt ? rb : 0, ivec3(b.x, b.y, b.z), cnt ? rb + cnt : 0);
        out_entries[level_off + i] = uint32_t(dst) << 2;
    } else {
        out_entries[level_off + i] = (e & 3u) | (((e >> 2) + uint32_t(level_off + num_cells)) << 2);
    }
}

// copy_refs + remap_refs + the scatter half of the sort (build.cu:634-647, :681, :691): the slot inside the cell's list is
// the rank classify_refs drew for the reference
__global__ void __launch_bounds__(kBlock) scatter_kept_refs(const int* __restrict__ ref_ids, const int* __restrict__ cell_ids, int num_refs,
                                                            const int* __restrict__ ranks, const int* __restrict__ ref_begin, int* __restrict__ out_refs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_refs) return;
    const int r = ranks[i];
    if (r < 0) return;                                 // split further or without a cell: classify_refs marked it
    out_refs[ref_begin[cell_ids[i]] + r] = ref_ids[i];
}

} // namespace

namespace {

struct PlainIn { const int* v; __device__ int operator()(int i) const { return v[i]; } };
struct PlainOut { int* v; __device__ void operator()(int i, int s) const { v[i] = s; } };

// Puts every cell's reference list in ascending order (the canonical order: primitive ids within a cell
// are distinct).  Lists are short -- a handful of references -- so each lane sorts its own cell in place:
// insertion sort, preceded by Shell passes for the rare long list.
// (one launch per level over the level's own 4-byte counts and list starts: a quarter of the bytes of the finished cells)
__global__ void __launch_bounds__(kBlock) sort_cell_refs(const int* __restrict__ cell_counts, const int* __restrict__ ref_begin, int num_cells, int* __restrict__ refs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_cells) return;
    const int n = cell_counts[i];                      // 0 for a cell that was split further
    if (n < 2) return;
    int* r = refs + ref_begin[i];
    if (n > 24) {
        const int gaps[8] = { 1750, 701, 301, 132, 57, 23, 10, 4 };
        for (int g = 0; g < 8; g++) {
            const int gap = gaps[g];
            for (int a = gap; a < n; a++) {
                const int v = r[a];
                int b = a - gap;
                while (b >= 0 && r[b] > v) { r[b + gap] = r[b]; b -= gap; }
                r[b + gap] = v;
            }
        }
    }
    for (int a = 1; a < n; a++) {
        const int v = r[a];
        int b = a - 1;
        while (b >= 0 && r[b] > v) { r[b + 1] = r[b]; b--; }
        r[b + 1] = v;
    }
}

struct Level {
    int* ref_ids = nullptr; int* cell_ids = nullptr; int num_refs = 0;
    Cell* cells = nullptr; uint32_t* entries = nullptr; int num_cells = 0;
    int* cell_counts = nullptr;                 // kept references per cell
    int* ranks = nullptr;                       // per kept reference: its slot inside its cell's list
    int* start_cell = nullptr; int* ref_begin = nullptr;
};

using Temps = PoolTemps;

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
    const int num_top = int(num_top_ll);

    BuildK k;
    k.dims = dims; k.shift = 0; k.bmin = gb.min; k.bmax = gb.max; k.cell_size = vec3(0.0f);

    // ---- reference counts, per-cell depth, shift (first_build_iter, build.cu:470-512) ----
    int* counts = tmp.get<int>(size_t(num_tris));
    int* start_emit = tmp.get<int>(size_t(num_tris));
    int* refs_per_cell = tmp.get<int>(size_t(num_top));
    int* log_dims = tmp.get<int>(size_t(num_top));
    int* partials = tmp.get<int>(2 * size_t(scan_num_tiles(std::max(num_tris, num_top)) + 1));
    if (!counts || !start_emit || !refs_per_cell || !log_dims || !partials) return HAGRID_ENOMEM;
    HG_HIP(ctx, hipMemsetAsync(refs_per_cell, 0, size_t(num_top) * sizeof(int), st));
    count_top_refs<<<grid_blocks(num_tris, kBlock), kBlock, 0, st>>>(tris, num_tris, k, counts, refs_per_cell); HG_DBG(ctx);
    if (!ctx_scan<int>(ctx, PlainIn{counts}, PlainOut{start_emit}, num_tris, partials, (const int*)nullptr, dsc + 0)) return HAGRID_ENOMEM;
    top_log_dims<<<grid_blocks(num_top, kBlock), kBlock, 0, st>>>(refs_per_cell, num_top, k, snd_density, log_dims, dsc + 1); HG_DBG(ctx);
    int h2[2];
    HG_TRY(read_back(ctx, dsc, h2, sizeof(h2)));
    const int R0 = h2[0], shift = h2[1];
    if (R0 < 0 || R0 > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many top-level references");
    if (shift >= 24) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many levels");
    k.shift = shift;
    k.cell_size = gb.extents() / vec3(dims << shift);
    tmp.drop(counts); tmp.drop(refs_per_cell);
    hagrid_build_counts& bc = ctx->counts;
    memset(&bc, 0, sizeof(bc));
    bc.num_tris = num_tris; bc.top_cells = num_top; bc.top_refs = R0;

    std::vector<Level> levels;
    {
        Level L;
        L.num_refs = R0; L.num_cells = num_top;
        L.ref_ids = tmp.get<int>(size_t(R0)); L.cell_ids = tmp.get<int>(size_t(R0));
        L.cells = tmp.get<Cell>(size_t(num_top)); L.entries = tmp.get<uint32_t>(size_t(num_top) + 1);
        L.cell_counts = tmp.get<int>(size_t(num_top)); L.ranks = tmp.get<int>(size_t(R0));
        L.start_cell = tmp.get<int>(size_t(num_top)); L.ref_begin = tmp.get<int>(size_t(num_top));
        if (!L.ref_ids || !L.cell_ids || !L.cells || !L.entries || !L.cell_counts || !L.ranks || !L.start_cell || !L.ref_begin) return HAGRID_ENOMEM;
        emit_top_cells<<<grid_blocks(num_top, kBlock), kBlock, 0, st>>>(L.cells, num_top, k, log_dims, L.entries, L.cell_counts); HG_DBG(ctx);
        emit_top_refs<<<grid_blocks(num_tris, kBlock), kBlock, 0, st>>>(tris, num_tris, k, start_emit, L.ref_ids, L.cell_ids, log_dims, L.entries); HG_DBG(ctx);
        levels.push_back(L);
    }
    tmp.drop(start_emit);

    // ---- subdivision, one level per iteration (build_iter, build.cu:527-619) ----
    for (int level = 0;; level++) {
        Level& L = levels.back();
        int* tot = dsc + 8 + 4 * level;              // {new cells, children, kept}; zeroed above
        int* part = tmp.get<int>(size_t(scan_num_tiles(L.num_cells)) + 1);
        unsigned char* masks = tmp.get<unsigned char>(size_t(L.num_refs) + 1);
        if (!part || !masks) return HAGRID_ENOMEM;
        if (!ctx_scan<int>(ctx, ChildCountIn{L.entries}, UpdateEntriesOut{L.entries}, L.num_cells, part, (const int*)nullptr, tot + 0)) return HAGRID_ENOMEM;
        if (L.num_refs > 0)
            classify_refs<<<std::min(grid_blocks(L.num_refs, kBlock), 4096), kBlock, 0, st>>>(L.ref_ids, L.cell_ids, L.num_refs, tris, L.cells, L.entries, k,
                                                                               masks, L.cell_counts, L.ranks, tot + 1); HG_DBG(ctx);
        int h3[3];
        HG_TRY(read_back(ctx, tot, h3, sizeof(h3)));
        const int num_new_cells = h3[0], num_children = h3[1];
        if (level < HAGRID_MAX_LEVELS) { bc.level_refs[level] = L.num_refs; bc.level_cells[level] = L.num_cells; bc.level_kept[level] = h3[2]; bc.num_levels = level + 1; }
        tmp.drop(part);
        if (num_new_cells == 0) { tmp.drop(masks); break; }              // build.cu:583-587
        if (num_new_cells < 0 || num_children < 0 || num_children > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: level too large");
        if ((int)levels.size() >= 24) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many levels");

        Level N;
        N.num_refs = num_children; N.num_cells = num_new_cells;
        N.ref_ids = tmp.get<int>(size_t(num_children)); N.cell_ids = tmp.get<int>(size_t(num_children));
        N.cells = tmp.get<Cell>(size_t(num_new_cells)); N.entries = tmp.get<uint32_t>(size_t(num_new_cells) + 1);
        N.cell_counts = tmp.get<int>(size_t(num_new_cells)); N.ranks = tmp.get<int>(size_t(num_children));
        N.start_cell = tmp.get<int>(size_t(num_new_cells)); N.ref_begin = tmp.get<int>(size_t(num_new_cells));
        if (!N.ref_ids || !N.cell_ids || !N.cells || !N.entries || !N.cell_counts || !N.ranks || !N.start_cell || !N.ref_begin) return HAGRID_ENOMEM;
        int* cursor = tot + 3;                                              // zeroed above
        emit_child_cells<<<grid_blocks((long long)L.num_cells * 8, kBlock), kBlock, 0, st>>>(L.entries, L.cells, L.num_cells, N.cells, N.entries, N.cell_counts); HG_DBG(ctx);
        emit_child_refs<<<std::min(grid_blocks(L.num_refs, kBlock * kEmitItems), 4096), kBlock, 0, st>>>(L.ref_ids, L.cell_ids, L.num_refs, masks, L.ranks,
                                                                             N.entries, N.ref_ids, N.cell_ids, cursor); HG_DBG(ctx);
        tmp.drop(masks);
        levels.push_back(N);
    }
    tmp.drop(log_dims);

    // ---- concat_levels (build.cu:621-716) ----
    const int num_levels = (int)levels.size();
    long long total_cells_ll = 0;
    int max_cells = 0;
    for (auto& L : levels) { total_cells_ll += L.num_cells; max_cells = std::max(max_cells, L.num_cells); }
    if (total_cells_ll > 0x3fffffff) HG_FAIL(ctx, HAGRID_ERANGE, "build_grid: too many voxel map entries");
    const int total_cells = int(total_cells_ll);
    Int2* part2 = tmp.get<Int2>(size_t(scan_num_tiles(max_cells)) + 1);
    Int2* carry = reinterpret_cast<Int2*>(dsc + 128);      // one Int2 per level, chained on the device
    if (!part2) return HAGRID_ENOMEM;
    for (int l = 0; l < num_levels; l++) {
        Level& L = levels[l];
        if (!ctx_scan<Int2>(ctx, LeafIn{L.entries, L.cell_counts}, LeafOut{L.start_cell, L.ref_begin}, L.num_cells, part2,
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
        Level& L = levels[l];
        concat_level<<<grid_blocks(L.num_cells, kBlock), kBlock, 0, st>>>(L.entries, L.cells, L.cell_counts, L.start_cell, L.ref_begin,
                                                                          L.num_cells, off, out_cells, out_entries); HG_DBG(ctx);
        if (L.num_refs > 0)
            scatter_kept_refs<<<grid_blocks(L.num_refs, kBlock), kBlock, 0, st>>>(L.ref_ids, L.cell_ids, L.num_refs, L.ranks, L.ref_begin, out_refs); HG_DBG(ctx);
        if (L.num_refs > 0)
            sort_cell_refs<<<grid_blocks(L.num_cells, kBlock), kBlock, 0, st>>>(L.cell_counts, L.ref_begin, L.num_cells, out_refs); HG_DBG(ctx);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // temporaries are released below
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

// merge.hip -- hagrid_merge_grid: SAH-guided merging of neighbouring cells on gfx950.
//
// Replaces the reference's merge.cu: merge_grid (:331-377), merge_iteration<axis> (:292-329) and the kernels
// compute_merge_counts (:91-142), compute_cell_flags (:145-170), compute_ref_counts (:173-186), merge (:189-278),
// remap_entries (:281-290).  Results are bit-identical to the CPU oracle's.
//
// Differences in structure: the axis is a kernel argument (one kernel instead of three instantiations); one host round trip per
// iteration (the three axis passes chain through device-side counts); the reference's warp-cooperative copy of unmerged runs
// (written for 32-lane warps, merge.cu:245-270) is replaced by per-cell copies -- lists hold one to two references on average.
// The passes are bound by the bytes they stream -- about 6 % of the cells merge in a pass, the rest is copied to its new place --
// so between the passes a cell is a 16-byte working record (CellFmt<true>), and the two scans of an iteration (cells kept,
// references kept) never write a value per cell: sums per 256-cell tile, one small scan over the tile sums, the scan inside a
// tile by the merge kernel's own workgroup (compute_ref_counts disappears into the item).
//
// Tried and rejected (round 1): merging in place (cells keep their index, absorbed cells become tombstones with a
// redirect, merged lists bump-allocated by a chained scan, one compaction at the end).  Bit-identical, but slower:
// 2.9 instead of 2.5 ms at 1M triangles, because every pass then runs over all 7.4M original slots while compacting
// after each pass shrinks the population to 4.2M -- the compaction pays for itself.
// Round 2, also measured and dropped (profiles/NOTES.md): the merge kernel as the output functor of the look-back scan (the
// persistent, occupancy-limited scan is the wrong host for gather work: merge 2.43 -> 3.76 ms); trying the next few array
// slots as the neighbour before walking the voxel map (+8 us per candidate and pass: the walk is cache-friendly, the extra
// load is not free); two or four cells per thread carried through compute_merge_counts in lock step (+0.15 / +0.39 ms: the
// pass is bound by gather throughput and bytes, not by the latency of one chain); {count, next} as one 8-byte record (+-0).
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct MergeK {          // merge.cu:17-19
    ivec3 dims;          // virtual resolution
    ivec3 top;           // top-level resolution
    vec3 cell_size;
    int shift;
};

struct CellRec { ivec3 lo; int begin; ivec3 hi; int end; };

// Between the passes the cells live in a 16-byte WORKING record: six u16 bounds + the first slot of the list.  The passes are
// bound by the bytes they stream (6 % of the cells merge in a pass, the rest is copied to its new place), and the lists of
// consecutive cells are contiguous in every array a pass writes -- a cell's list ends where the next one's begins, the record
// after the last cell holds the number of references -- so `end` need not be stored.  The first pass reads the caller's 32-byte
// cells, the last step converts back.  (Virtual resolutions of 65536 and more keep the 32-byte record throughout.)
template <bool NARROW> struct CellFmt;
template <> struct CellFmt<false> {
    static __device__ __forceinline__ CellRec load(const void* cells, int i) {
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
        const int4 a = p[0], b = p[1];
        CellRec c; c.lo = ivec3(a.x, a.y, a.z); c.begin = a.w; c.hi = ivec3(b.x, b.y, b.z); c.end = b.w;
        return c;
    }
    static __device__ __forceinline__ void finish(const void*, int, CellRec&) {}
    static __device__ __forceinline__ void store(void* cells, int i, ivec3 lo, int begin, ivec3 hi, int end) {
        int4* p = reinterpret_cast<int4*>(cells) + 2 * size_t(i);
        p[0] = make_int4(lo.x, lo.y, lo.z, begin);
        p[1] = make_int4(hi.x, hi.y, hi.z, end);
    }
    static __device__ __forceinline__ void store_end(void*, int, int) {}
};
template <> struct CellFmt<true> {
    // box and begin; `end` is filled by finish() -- callers that only look at the box of a neighbour skip that second access
    static __device__ __forceinline__ CellRec load(const void* cells, int i) {
        const uint4 v = reinterpret_cast<const uint4*>(cells)[i];
        CellRec c;
        c.lo = ivec3(int(v.x & 0xffffu), int(v.x >> 16), int(v.y & 0xffffu));
        c.hi = ivec3(int(v.y >> 16), int(v.z & 0xffffu), int(v.z >> 16));
        c.begin = int(v.w); c.end = int(v.w);
        return c;
    }
    static __device__ __forceinline__ void finish(const void* cells, int i, CellRec& c) { c.end = int(reinterpret_cast<const uint4*>(cells)[size_t(i) + 1].w); }
    static __device__ __forceinline__ void store(void* cells, int i, ivec3 lo, int begin, ivec3 hi, int) {
        reinterpret_cast<uint4*>(cells)[i] = make_uint4(uint32_t(lo.x) | uint32_t(lo.y) << 16, uint32_t(lo.z) | uint32_t(hi.x) << 16,
                                                        uint32_t(hi.y) | uint32_t(hi.z) << 16, uint32_t(begin));
    }
    static __device__ __forceinline__ void store_end(void* cells, int num_cells, int num_refs) {      // the record after the last cell
        reinterpret_cast<uint4*>(cells)[num_cells] = make_uint4(0u, 0u, 0u, uint32_t(num_refs));
    }
};
__device__ __forceinline__ int comp(const ivec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
__device__ __forceinline__ float comp(const vec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// merge.cu:21-31
__device__ __forceinline__ bool aligned(int axis, const CellRec& c1, const CellRec& c2) {
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    return comp(c1.hi, axis) == comp(c2.lo, axis) &&
           comp(c1.lo, a1) == comp(c2.lo, a1) && comp(c1.lo, a2) == comp(c2.lo, a2) &&
           comp(c1.hi, a1) == comp(c2.hi, a1) && comp(c1.hi, a2) == comp(c2.hi, a2);
}
// merge.cu:34-39
__device__ __forceinline__ bool merge_allowed(const MergeK& k, int empty_mask, int pos) {
    const int top_level_mask = (1 << k.shift) - 1;
    const int is_shifted = (pos >> k.shift) & empty_mask;
    const bool is_top_level = !(pos & top_level_mask);
    return !is_shifted || !is_top_level;
}
// merge.cu:42-47
__device__ __forceinline__ ivec3 next_cell_pos(int axis, const ivec3& lo, const ivec3& hi) {
    return ivec3(axis == 0 ? hi.x : lo.x, axis == 1 ? hi.y : lo.y, axis == 2 ? hi.z : lo.z);
}
// ---- sorted id lists ---------------------------------------------------------------------------------------------------
// Reference lists are ascending and free of duplicates (merge.cu:57 relies on it), one to two ids on average.
// |A u B| = |A| + |B| - |A n B| (what count_union of merge.cu:58-69 returns): lists of at most four ids each sit in registers and
// every id is compared against every id -- no loop, no data-dependent branch; otherwise the ids of the shorter list are looked
// up in the longer one by binary search.
__device__ __forceinline__ int lower_bound_in(const int* __restrict__ a, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int union_size(const int* __restrict__ a, int na, const int* __restrict__ b, int nb) {
    if (na == 0 || nb == 0) return na + nb;
    int common = 0;
    if (na <= 4 && nb <= 4) {
        // both lists in registers (eight independent loads), sixteen compares: unused slots hold -1 / -2 (ids are >= 0)
        const int a0 = a[0], a1 = na > 1 ? a[1] : -1, a2 = na > 2 ? a[2] : -1, a3 = na > 3 ? a[3] : -1;
        const int b0 = b[0], b1 = nb > 1 ? b[1] : -2, b2 = nb > 2 ? b[2] : -2, b3 = nb > 3 ? b[3] : -2;
        auto hits = [&](int x) { return int(x == b0) + int(x == b1) + int(x == b2) + int(x == b3); };
        common = hits(a0) + hits(a1) + hits(a2) + hits(a3);
    } else {
        if (nb > na) { const int* t = a; a = b; b = t; const int n = na; na = nb; nb = n; }      // b is the shorter list
        for (int j = 0; j < nb; j++) {
            const int x = b[j], at = lower_bound_in(a, na, x);
            common += at < na && a[at] == x;
        }
    }
    return na + nb - common;
}
// The union itself, ascending, n_out = union_size ids (merge_refs of merge.cu:72-88): an exhausted list reads as +infinity, the
// loop runs over the OUTPUT, so there is no tail to copy.
__device__ __forceinline__ void write_union(const int* __restrict__ a, int na, const int* __restrict__ b, int nb, int* __restrict__ out, int n_out) {
    const int inf = 0x7fffffff;
    int i = 0, j = 0;
    int x = na > 0 ? a[0] : inf, y = nb > 0 ? b[0] : inf;
    for (int o = 0; o < n_out; o++) {
        const int m = min(x, y);
        out[o] = m;
        if (x == m) { i++; x = i < na ? a[i] : inf; }
        if (y == m) { j++; y = j < nb ? b[j] : inf; }
    }
}

// The surface-area test of merge.cu:113-133 for two aligned cells with their list ranges filled in: the size of the merged list if merging
// does not raise the cost, else -1.
__device__ __forceinline__ int merged_size_if_cheaper(const MergeK& k, int axis, const CellRec& c1, const CellRec& c2, const int* __restrict__ refs) {
    const float unit_cost = 1.0f;
    const vec3 e1 = vec3(c1.hi - c1.lo) * k.cell_size;
    const vec3 e2 = vec3(c2.hi - c2.lo) * k.cell_size;
    const float a1 = e1.x * (e1.y + e1.z) + e1.y * e1.z;
    const float a2 = e2.x * (e2.y + e2.z) + e2.y * e2.z;
    const float a = a1 + a2 - comp(e1, (axis + 1) % 3) * comp(e1, (axis + 2) % 3);
    const int n1 = c1.end - c1.begin, n2 = c2.end - c2.begin;
    const float cc1 = a1 * (n1 + unit_cost), cc2 = a2 * (n2 + unit_cost);
    if (a * (max(n1, n2) + unit_cost) <= cc1 + cc2) {
        const int n = union_size(refs + c1.begin, n1, refs + c2.begin, n2);
        const float c = a * (n + unit_cost);
        if (c <= cc1 + cc2) return n;
    }
    return -1;
}

// compute_merge_counts (merge.cu:91-142)
template <bool NARROW>
__global__ void __launch_bounds__(kBlock) merge_counts_kernel(int axis, MergeK k, const Entry* __restrict__ entries, const void* __restrict__ cells,
                                                              const int* __restrict__ refs, int* __restrict__ merge_counts,
                                                              int* __restrict__ nexts, unsigned char* __restrict__ has_prev, int pass_tag, int empty_mask, int num_cells,
                                                              const int* __restrict__ n_dev) {
    using F = CellFmt<NARROW>;
    // (the workgroups that hold cells take them XCD by XCD -- wave_prims.h xcd_block: the neighbour's cell and list are then in the same L2)
    const int n = n_dev ? *n_dev : num_cells, active = (n + kBlock - 1) / kBlock;
    if (int(blockIdx.x) >= active) return;
    const int id = xcd_block(blockIdx.x, active) * kBlock + threadIdx.x;
    if (id >= n) return;
    CellRec c1 = F::load(cells, id);
    F::finish(cells, id, c1);
    const ivec3 np = next_cell_pos(axis, c1.lo, c1.hi);
    int count = -(c1.end - c1.begin + 1);
    int next_id = -1;
    if (merge_allowed(k, empty_mask, comp(c1.lo, axis)) && comp(np, axis) < comp(k.dims, axis)) {
        next_id = int(lookup_entry(entries, k.shift, k.top, np));
        CellRec c2 = F::load(cells, next_id);
        if (aligned(axis, c1, c2)) {
            F::finish(cells, next_id, c2);
            const int n = merged_size_if_cheaper(k, axis, c1, c2, refs);
            if (n >= 0) count = n;
        }
    }
    merge_counts[id] = count;
    next_id = count >= 0 ? next_id : -1;
    nexts[id] = next_id;
    // merge.cu:141 stores the predecessor's id; only "has a predecessor" is ever read (compute_cell_flags, merge.cu:152), so the
    // array holds the tag of the pass that last gave the cell one: no clearing between the passes
    if (next_id >= 0) has_prev[next_id] = (unsigned char)pass_tag;
}

// compute_cell_flags (merge.cu:145-170): chain heads mark every second cell of their chain as residue
__global__ void __launch_bounds__(kBlock) cell_flags_kernel(const int* __restrict__ nexts, const unsigned char* __restrict__ has_prev, int pass_tag,
                                                            unsigned char* __restrict__ cell_flags, int num_cells, const int* __restrict__ n_dev) {
    // four consecutive cells per thread (one 16-byte and one 4-byte load; the arrays come from the pool); flags are bytes and are
    // stored one by one because the flag of a cell with a predecessor is written by its chain's head
    const int id = (blockIdx.x * kBlock + threadIdx.x) * 4;
    const int n = n_dev ? *n_dev : num_cells;
    if (id >= n) return;
    int hp[4], nx[4];
    if (id + 4 <= n) {
        const uchar4 h = *reinterpret_cast<const uchar4*>(has_prev + id);
        const int4 x = *reinterpret_cast<const int4*>(nexts + id);
        hp[0] = h.x; hp[1] = h.y; hp[2] = h.z; hp[3] = h.w; nx[0] = x.x; nx[1] = x.y; nx[2] = x.z; nx[3] = x.w;
    } else {
        for (int c = 0; c < 4; c++) { hp[c] = id + c < n ? has_prev[id + c] : pass_tag; nx[c] = id + c < n ? nexts[id + c] : -1; }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (hp[c] == pass_tag) continue;
        int next_id = nx[c];
        cell_flags[id + c] = 1;
        int count = 1;
        while (next_id >= 0) {
            cell_flags[next_id] = (count & 1) ? 0 : 1;
            next_id = nexts[next_id];
            count++;
        }
    }
}

// The two scans of merge.cu:310-311 (cells kept, references kept; compute_ref_counts of merge.cu:173-186 is the item) in three
// steps that never write a scanned value per cell: one sum per TILE of 256 cells (a wavefront reads its tile with 16-byte
// accesses), an exclusive scan over the tile sums (a few thousand items, in place), and the scan inside the tile by the
// workgroup of the merge kernel that owns it.
constexpr int kMergeTile = kBlock;                  // cells per workgroup of merge_kernel
__device__ __forceinline__ Int2 keep_item(int flag, int count) { return Int2{ flag ? 1 : 0, flag ? (count >= 0 ? count : -(count + 1)) : 0 }; }

__global__ void __launch_bounds__(kBlock) merge_tile_sums(const unsigned char* __restrict__ cell_flags, const int* __restrict__ merge_counts, int num_cells,
                                                          const int* __restrict__ n_dev, Int2* __restrict__ sums, int num_tiles) {
    static_assert(kMergeTile == 64 * 4, "a wavefront covers one tile with four cells per lane");
    const int tile = blockIdx.x * kWaves + wave_id();
    if (tile >= num_tiles) return;
    const int n = n_dev ? min(num_cells, *n_dev) : num_cells;
    const int i = tile * kMergeTile + lane_id() * 4;
    Int2 s{0, 0};
    if (i + 4 <= n) {
        const uchar4 f = *reinterpret_cast<const uchar4*>(cell_flags + i);
        const int4 m = *reinterpret_cast<const int4*>(merge_counts + i);
        s = keep_item(f.x, m.x) + keep_item(f.y, m.y) + keep_item(f.z, m.z) + keep_item(f.w, m.w);
    } else {
        for (int c = 0; c < 4; c++) if (i + c < n) s = s + keep_item(cell_flags[i + c], merge_counts[i + c]);
    }
    s = Int2{wave_sum(s.a), wave_sum(s.b)};
    if (lane_id() == 0) sums[tile] = s;
}
struct SumsIn { const Int2* v; __device__ Int2 operator()(int i) const { return v[i]; } };
struct SumsOut { Int2* v; __device__ void operator()(int i, Int2 s) const { v[i] = s; } };

// merge (merge.cu:189-278)
template <bool IN_NARROW, bool OUT_NARROW>
__global__ void __launch_bounds__(kBlock) merge_kernel(int axis, MergeK k, const Entry* __restrict__ entries, const void* __restrict__ cells,
                                                       const int* __restrict__ refs, const unsigned char* __restrict__ cell_flags,
                                                       const Int2* __restrict__ tile_prefix, const int* __restrict__ merge_counts,
                                                       int* new_cell_ids /* holds nexts on entry */,
                                                       void* __restrict__ new_cells, int* __restrict__ new_refs, int num_cells, const int* __restrict__ n_dev,
                                                       const Int2* __restrict__ totals,
                                                       const unsigned char* __restrict__ stamp_in, unsigned char* __restrict__ stamp_out, int stamp) {
    using FI = CellFmt<IN_NARROW>;
    using FO = CellFmt<OUT_NARROW>;
    __shared__ Int2 lds[kWaves];
    const int n = n_dev ? min(num_cells, *n_dev) : num_cells, active = (n + kBlock - 1) / kBlock;
    if (int(blockIdx.x) >= active && blockIdx.x != 0) return;                       // (uniform over the workgroup; workgroup 0 also runs for n == 0: the end marker)
    const int tile = int(blockIdx.x) < active ? xcd_block(blockIdx.x, active) : 0;   // XCD by XCD, like merge_counts_kernel
    const int id = tile * kBlock + threadIdx.x;
    if (id == 0) FO::store_end(new_cells, totals->a, totals->b);
    const bool inside = id < n;
    const int flag = inside ? cell_flags[id] : 0;
    const int mc = inside ? merge_counts[id] : 0;
    // {new cell id, first slot of the list} = exclusive scan of the kept cells' items: inside the tile here, the tile's offset
    // from the scan over the tile sums
    const Int2 item = keep_item(flag, mc);
    const Int2 incl = wave_inclusive_scan(item);
    if (lane_id() == 63) lds[wave_id()] = incl;
    __syncthreads();
    if (!flag) return;
    Int2 at = tile_prefix[tile];
    for (int w = 0; w < wave_id(); w++) at = at + lds[w];
    const int new_id = at.a + incl.a - item.a, nb = at.b + incl.b - item.b;
    CellRec cell = FI::load(cells, id);
    FI::finish(cells, id, cell);
    // the array still holds `nexts` (compute_merge_counts' neighbour, = what lookup_entry would find again): only the second
    // cell of a merging pair is overwritten by another thread, and that cell is residue (flag 0) -- it never gets here
    const int next_id = new_cell_ids[id];
    new_cell_ids[id] = new_id;
    // the pass that gave the cell its present box and list (0: the construction): what the in-place iterations start from (ip_begin)
    if (stamp_out) stamp_out[new_id] = (unsigned char)(mc >= 0 ? stamp : (stamp_in ? stamp_in[id] : 0));
    const int n1 = cell.end - cell.begin;
    if (mc >= 0) {
        CellRec nc = FI::load(cells, next_id);
        FI::finish(cells, next_id, nc);
        new_cell_ids[next_id] = new_id;
        FO::store(new_cells, new_id, min(nc.lo, cell.lo), nb, max(nc.hi, cell.hi), nb + mc);
        if (nc.begin < nc.end) {
            write_union(refs + cell.begin, n1, refs + nc.begin, nc.end - nc.begin, new_refs + nb, mc);
            return;
        }
    } else {
        FO::store(new_cells, new_id, cell.lo, nb, cell.hi, nb + n1);
    }
    for (int i = 0; i < n1; i++) new_refs[nb + i] = refs[cell.begin + i];
}


// ---- iterations in place ------------------------------------------------------------------------------------------------------
// After the first iteration a pass merges a few per cent of the cells, later ones a few per mille (1M-triangle soup: 19, 17, 8, 3.3, 1.8, 0.6,
// 0.3, 0.08, 0.02 %), yet a compacting pass streams every cell, every reference and every voxel-map word to merge them.  Once an iteration has
// merged less than half of its cells ("merge.inplace_div", default 2) the remaining iterations therefore run IN PLACE on the 16-byte working records:
//   * cells keep their slot; the cell that absorbs its neighbour takes the merged box and a list appended behind the live references
//     (so a list needs an explicit end: list_end[]), the absorbed one becomes a TOMBSTONE that names its absorber (look-ups through the voxel
//     map, which is not rewritten, follow tombstones);
//   * a pass only looks at cells whose situation changed since the last pass of its axis: a cell is DIRTY for an axis when it or the cell behind
//     its face in that direction was merged (the absorber marks itself for all axes and, per axis, the one cell that can be aligned with it from
//     behind: the cell that holds the voxel in front of its lower corner).  A cell that is not dirty decided "no" before, on the same boxes and
//     lists and under a mask that has only become stricter since (merge.cu:361: 1, 3, 7, 15; from the fifth iteration on the mask is 0 and every cell is
//     looked at again once), so it would decide "no" again.
//   * slots for the merged lists come from a scan over the absorbers of the pass (no atomics); ONE compaction at the very end restores the
//     public arrays: cells in slot order (= the order repeated compactions give: the absorber keeps its place), lists in cell order, the voxel
//     map rewritten once.
// Round 3 built in-place iterations without the dirty marks (every pass still evaluated every cell: no gain, reverted); with them a late pass
// costs a few sweeps over byte flags.
constexpr uint32_t kTombLo = 0xffffu;             // lo.x of a tombstone (the working record is used below virtual resolutions of 65536: no cell has it)
__device__ __forceinline__ bool ip_is_tomb(const uint4& v) { return (v.x & 0xffffu) == kTombLo; }
__device__ __forceinline__ int ip_live(const void* cells, int id) {          // the live cell a slot stands for
    for (;;) {
        const uint4 v = reinterpret_cast<const uint4*>(cells)[id];
        if (!ip_is_tomb(v)) return id;
        id = int(v.w);
    }
}
__device__ __forceinline__ CellRec ip_load(const void* cells, const int* list_end, int id) {
    CellRec c = CellFmt<true>::load(cells, id);
    c.end = list_end[id];
    return c;
}

// The sweeps of a pass read a flag byte per slot and mostly find nothing to do: four slots per thread, one 4-byte load of the flags.
constexpr int kIpPer = 4;
constexpr int kIpTile = kBlock * kIpPer;            // slots per workgroup of the sweeps, and per tile of the slot scan
__device__ __forceinline__ uint32_t ip_flags4(const unsigned char* __restrict__ f, int id4, int slots) {
    if (id4 + 4 <= slots) return *reinterpret_cast<const uint32_t*>(f + id4);
    uint32_t v = 0;
    for (int c = 0; c < 4; c++) if (id4 + c < slots) v |= uint32_t(f[id4 + c]) << (8 * c);
    return v;
}
__device__ __forceinline__ uint32_t ip_match4(uint32_t flags, int tag) {      // bit 8c set where byte c equals tag
    uint32_t m = 0;
    for (int c = 0; c < 4; c++) if (int((flags >> (8 * c)) & 0xffu) == tag) m |= 1u << (8 * c);
    return m;
}

// The flagged slots of a workgroup's tile, compacted in LDS: a sparse pass then works with full wavefronts, one slot per thread, instead of a
// few lanes per wavefront walking up to four dependent-gather chains one after the other (1M-triangle soup, third iteration: 36 -> ~10 us per pass
// for the evaluation).  Returns the number of listed slots (in list[0 ..)); every thread of the workgroup must call it.
__device__ __forceinline__ int ip_compact_tile(uint32_t mask4 /* bit 8c: slot id4 + c is flagged */, int id4, int* __restrict__ list, int* __restrict__ count) {
    if (threadIdx.x == 0) *count = 0;
    __syncthreads();
    const int n = int((mask4 & 1u) + ((mask4 >> 8) & 1u) + ((mask4 >> 16) & 1u) + ((mask4 >> 24) & 1u));
    const int incl = wave_inclusive_scan(n);
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane_id() == 63 && total) base = atomicAdd(count, total);
    base = __shfl(base, 63, 64) + incl - n;
    for (int c = 0; c < 4; c++) if (mask4 & (1u << (8 * c))) list[base++] = id4 + c;
    __syncthreads();
    return *count;
}

// Entering the mode: explicit list ends; a cell is dirty for an axis when it was made by a merge after the last evaluation of that axis
// (stamps: the pass that gave a cell its box and list; since[a]: the pass of the last evaluation of axis a) -- and so is the cell behind
// its lower corner (ip_mark_entry).  Without stamps every cell is dirty.
__global__ void __launch_bounds__(kBlock) ip_begin(const void* __restrict__ cells, int slots, int* __restrict__ list_end, unsigned char* __restrict__ dirty,
                                                   const unsigned char* __restrict__ stamps, int since_x, int since_y, int since_z,
                                                   unsigned char* __restrict__ evaluated, unsigned char* __restrict__ absorbs, int* __restrict__ books, int num_refs, size_t dstride) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id == 0) { books[0] = num_refs; books[1] = slots; books[2] = num_refs; books[3] = 0; }      // cursor, live cells, live references, overflow
    if (id >= slots) return;
    evaluated[id] = 0; absorbs[id] = 0;                                       // the tags of the mode's passes
    list_end[id] = int(reinterpret_cast<const uint4*>(cells)[size_t(id) + 1].w);
    const int st = stamps ? stamps[id] : 255;
    dirty[id] = st >= since_x; dirty[dstride + id] = st >= since_y; dirty[2 * dstride + id] = st >= since_z;
}
__global__ void __launch_bounds__(kBlock) ip_mark_entry(MergeK k, const Entry* __restrict__ entries, const void* __restrict__ cells, int slots,
                                                        unsigned char* __restrict__ dirty, const unsigned char* __restrict__ stamps, int since_x, int since_y, int since_z, size_t dstride) {
    __shared__ int list[kIpTile];
    __shared__ int count;
    const int id4 = (blockIdx.x * kBlock + threadIdx.x) * kIpPer;
    const int since_min = min(since_x, min(since_y, since_z));
    uint32_t recent = 0;
    if (id4 < slots) {
        const uint32_t f = ip_flags4(stamps, id4, slots);
        for (int c = 0; c < 4; c++) if (id4 + c < slots && int((f >> (8 * c)) & 0xffu) >= since_min) recent |= 1u << (8 * c);
    }
    const int n = ip_compact_tile(recent, id4, list, &count);
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int id = list[i], st = stamps[id];
        const CellRec cell = CellFmt<true>::load(cells, id);
        for (int axis = 0; axis < 3; axis++) {
            if (st < (axis == 0 ? since_x : (axis == 1 ? since_y : since_z))) continue;
            ivec3 p = cell.lo;
            if (axis == 0) p.x--; else if (axis == 1) p.y--; else p.z--;
            if (comp(p, axis) < 0) continue;
            dirty[size_t(axis) * dstride + int(lookup_entry(entries, k.shift, k.top, p))] = 1;
        }
    }
}

// compute_merge_counts (merge.cu:91-142) for the dirty cells of the axis
__device__ __forceinline__ void ip_count_one(int id, int axis, const MergeK& k, const Entry* __restrict__ entries, const void* __restrict__ cells, const int* __restrict__ list_end,
                                             const int* __restrict__ refs, Int2* __restrict__ minfo, int* __restrict__ nexts, unsigned char* __restrict__ evaluated,
                                             unsigned char* __restrict__ has_prev, int pass_tag, int empty_mask) {
    if (ip_is_tomb(reinterpret_cast<const uint4*>(cells)[id])) return;
    const CellRec c1 = ip_load(cells, list_end, id);
    const ivec3 np = next_cell_pos(axis, c1.lo, c1.hi);
    if (!merge_allowed(k, empty_mask, comp(c1.lo, axis)) || comp(np, axis) >= comp(k.dims, axis)) return;
    const int next_id = ip_live(cells, int(lookup_entry(entries, k.shift, k.top, np)));
    const CellRec c2 = ip_load(cells, list_end, next_id);
    if (!aligned(axis, c1, c2)) return;
    const int n = merged_size_if_cheaper(k, axis, c1, c2, refs);
    if (n < 0) return;
    minfo[id] = Int2{ n, (c1.end - c1.begin) + (c2.end - c2.begin) - n };     // merged size, references that disappear
    nexts[id] = next_id;
    evaluated[id] = (unsigned char)pass_tag;                                 // nexts[id] / minfo[id] belong to this pass
    has_prev[next_id] = (unsigned char)pass_tag;
}
__global__ void __launch_bounds__(kBlock) ip_counts(int axis, MergeK k, const Entry* __restrict__ entries, const void* __restrict__ cells, const int* __restrict__ list_end,
                                                    const int* __restrict__ refs, int slots, unsigned char* __restrict__ dirty_axis,
                                                    Int2* __restrict__ minfo, int* __restrict__ nexts, unsigned char* __restrict__ evaluated,
                                                    unsigned char* __restrict__ has_prev, int pass_tag, int empty_mask) {
    __shared__ int list[kIpTile];
    __shared__ int count;
    const int id4 = (xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x) * kIpPer;
    uint32_t mask = 0;
    if (id4 < slots) {
        const uint32_t f = ip_flags4(dirty_axis, id4, slots);
        for (int c = 0; c < 4; c++) if ((f >> (8 * c)) & 0xffu) { mask |= 1u << (8 * c); dirty_axis[id4 + c] = 0; }
    }
    const int n = ip_compact_tile(mask, id4, list, &count);
    for (int i = threadIdx.x; i < n; i += kBlock)
        ip_count_one(list[i], axis, k, entries, cells, list_end, refs, minfo, nexts, evaluated, has_prev, pass_tag, empty_mask);
}

// compute_cell_flags (merge.cu:145-170): chain heads name the absorbers of their chain (every second cell, while it has a successor)
__global__ void __launch_bounds__(kBlock) ip_chains(int slots, const int* __restrict__ nexts, const unsigned char* __restrict__ evaluated,
                                                    const unsigned char* __restrict__ has_prev, unsigned char* __restrict__ absorbs, int pass_tag) {
    const int id4 = (blockIdx.x * kBlock + threadIdx.x) * kIpPer;
    if (id4 >= slots) return;
    const uint32_t heads = ip_match4(ip_flags4(evaluated, id4, slots), pass_tag) & ~ip_match4(ip_flags4(has_prev, id4, slots), pass_tag);
    if (!heads) return;
    for (int c = 0; c < 4; c++) {
        if (!(heads & (1u << (8 * c)))) continue;
        int cur = id4 + c, pos = 0;
        for (;;) {
            const int nxt = evaluated[cur] == pass_tag ? nexts[cur] : -1;
            if (nxt < 0) break;
            if (!(pos & 1)) absorbs[cur] = (unsigned char)pass_tag;
            cur = nxt; pos++;
        }
    }
}

// per tile of kIpTile slots: {slots its merged lists need, merges} for the scan (its total closes the books of the pass), and on the side the
// references that disappear (sums of a pass kept by atomics on a handful of words of one cache line cost 115 us per pass: the L2 serialises them)
__global__ void __launch_bounds__(kBlock) ip_tile_sums(const unsigned char* __restrict__ absorbs, int pass_tag, const Int2* __restrict__ minfo, int slots,
                                                       Int2* __restrict__ sums, int* __restrict__ removed) {
    __shared__ int lds[kWaves];
    const int id4 = (blockIdx.x * kBlock + threadIdx.x) * kIpPer;
    int need = 0, merges = 0, gone = 0;
    if (id4 < slots) {
        const uint32_t m = ip_match4(ip_flags4(absorbs, id4, slots), pass_tag);
        for (int c = 0; c < 4; c++) if (m & (1u << (8 * c))) { const Int2 v = minfo[id4 + c]; need += v.a; merges++; gone += v.b; }
    }
    need = block_sum(need, lds); merges = block_sum(merges, lds); gone = block_sum(gone, lds);
    if (threadIdx.x == 0) { sums[blockIdx.x] = Int2{need, merges}; removed[blockIdx.x] = gone; }
}

// merge (merge.cu:189-278), in place.
// The merged lists are appended behind the live references in the SAME buffer (its tail is free once the first iterations have shrunk the lists).
// A pass whose lists do not fit is not applied at all, nor is any later pass of the iteration (books[3] = 1 + its axis, sticky): the host compacts and
// runs those passes the compacting way.
__global__ void __launch_bounds__(kBlock) ip_apply(void* cells, int* list_end, int* refs, const unsigned char* __restrict__ absorbs, int pass_tag,
                                                   const Int2* __restrict__ minfo, const int* __restrict__ nexts, const Int2* __restrict__ tile_prefix,
                                                   int slots, int* __restrict__ books, const Int2* __restrict__ pass_total, int capacity, int axis) {
    __shared__ int lds[kWaves];
    const int id4 = (blockIdx.x * kBlock + threadIdx.x) * kIpPer;
    const int* cursor = books;
    const int overflow = __hip_atomic_load(books + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (overflow > 0 || (long long)*cursor + pass_total->a > (long long)capacity) {       // (the same answer in every thread: books[0] and the total do not change in this kernel)
        if (id4 == 0 && overflow <= 0) __hip_atomic_store(books + 3, -(1 + axis), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // negative: "this pass"; ip_mark makes it sticky
        return;
    }
    const uint32_t mine = id4 < slots ? ip_match4(ip_flags4(absorbs, id4, slots), pass_tag) : 0u;
    int m[4], total = 0;
    for (int c = 0; c < 4; c++) { m[c] = (mine & (1u << (8 * c))) ? minfo[id4 + c].a : 0; total += m[c]; }
    const int incl = wave_inclusive_scan(total);
    if (lane_id() == 63) lds[wave_id()] = incl;
    __syncthreads();
    if (!mine) return;
    int at = *cursor + tile_prefix[blockIdx.x].a + incl - total;
    for (int w = 0; w < wave_id(); w++) at += lds[w];
    for (int c = 0; c < 4; c++) {
        if (!(mine & (1u << (8 * c)))) continue;
        const int id = id4 + c, other = nexts[id];
        const CellRec a = ip_load(cells, list_end, id), n = ip_load(cells, list_end, other);
        write_union(refs + a.begin, a.end - a.begin, refs + n.begin, n.end - n.begin, refs + at, m[c]);
        CellFmt<true>::store(cells, id, min(n.lo, a.lo), at, max(n.hi, a.hi), 0);
        list_end[id] = at + m[c];
        reinterpret_cast<uint4*>(cells)[other] = make_uint4(kTombLo, 0u, 0u, uint32_t(id));
        at += m[c];
    }
}

// who has to look again: the absorber for every axis and, per axis, the cell behind its lower corner; block 0 closes the books of the pass
__global__ void __launch_bounds__(kBlock) ip_mark(MergeK k, const Entry* __restrict__ entries, const void* __restrict__ cells, const unsigned char* __restrict__ absorbs,
                                                  int pass_tag, int slots, unsigned char* __restrict__ dirty, const Int2* __restrict__ pass_total,
                                                  const int* __restrict__ removed, int num_tiles, int* __restrict__ books /* cursor, live cells, live refs, overflow */,
                                                  int* __restrict__ snap, size_t dstride) {
    __shared__ int lds[kWaves];
    const int id4 = (blockIdx.x * kBlock + threadIdx.x) * kIpPer;
    const int overflow = books[3];                    // (written by the previous kernel at the latest; this kernel's thread 0 only changes its sign)
    if (blockIdx.x == 0) {                            // the books of the pass: cursor, live cells, live references
        int gone = 0;
        for (int i = threadIdx.x; i < num_tiles; i += kBlock) gone += removed[i];
        gone = block_sum(gone, lds);
        if (threadIdx.x == 0) {
            if (overflow == 0) { books[0] += pass_total->a; books[1] -= pass_total->b; books[2] -= gone; }
            else if (overflow < 0) books[3] = -overflow;
            snap[0] = books[1]; snap[1] = books[2];
        }
    }
    __shared__ int list[kIpTile];
    __shared__ int count;
    const uint32_t mine = (overflow == 0 && id4 < slots) ? ip_match4(ip_flags4(absorbs, id4, slots), pass_tag) : 0u;
    const int n = ip_compact_tile(mine, id4, list, &count);
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int id = list[i];
        const CellRec cell = CellFmt<true>::load(cells, id);
        for (int axis = 0; axis < 3; axis++) {
            dirty[size_t(axis) * dstride + id] = 1;
            ivec3 p = cell.lo;
            if (axis == 0) p.x--; else if (axis == 1) p.y--; else p.z--;
            if (comp(p, axis) < 0) continue;
            dirty[size_t(axis) * dstride + ip_live(cells, int(lookup_entry(entries, k.shift, k.top, p)))] = 1;
        }
    }
}

// ---- leaving the mode: one compaction ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) ip_live_sums(const void* __restrict__ cells, const int* __restrict__ list_end, int slots, Int2* __restrict__ sums, int num_tiles) {
    const int tile = blockIdx.x * kWaves + wave_id();
    if (tile >= num_tiles) return;
    Int2 s{0, 0};
    for (int c = 0; c < 4; c++) {
        const int i = tile * kMergeTile + c * 64 + lane_id();
        if (i < slots) {
            const uint4 v = reinterpret_cast<const uint4*>(cells)[i];
            if (!ip_is_tomb(v)) s = s + Int2{1, list_end[i] - int(v.w)};
        }
    }
    s = Int2{wave_sum(s.a), wave_sum(s.b)};
    if (lane_id() == 0) sums[tile] = s;
}
// live cells -> the public 32-byte records in slot order, their lists in cell order; new_ids[slot] = the cell's new index (-1 - absorber for a tombstone)
__global__ void __launch_bounds__(kBlock) ip_compact(const void* __restrict__ cells, const int* __restrict__ list_end, const int* __restrict__ refs, int slots,
                                                     const Int2* __restrict__ tile_prefix, Cell* __restrict__ out_cells, int* __restrict__ out_refs, int* __restrict__ new_ids) {
    __shared__ Int2 lds[kWaves];
    const int id = blockIdx.x * kBlock + threadIdx.x;
    uint4 v = make_uint4(kTombLo, 0u, 0u, 0u);
    if (id < slots) v = reinterpret_cast<const uint4*>(cells)[id];
    const bool live = id < slots && !ip_is_tomb(v);
    const int end = live ? list_end[id] : 0;
    const Int2 item = live ? Int2{1, end - int(v.w)} : Int2{0, 0};
    const Int2 incl = wave_inclusive_scan(item);
    if (lane_id() == 63) lds[wave_id()] = incl;
    __syncthreads();
    if (id >= slots) return;
    if (!live) { new_ids[id] = -1 - int(v.w); return; }
    Int2 at = tile_prefix[blockIdx.x];
    for (int w = 0; w < wave_id(); w++) at = at + lds[w];
    const int new_id = at.a + incl.a - 1, nb = at.b + incl.b - item.b;
    const CellRec c = CellFmt<true>::load(cells, id);
    CellFmt<false>::store(out_cells, new_id, c.lo, nb, c.hi, nb + item.b);
    for (int i = 0; i < item.b; i++) out_refs[nb + i] = refs[c.begin + i];
    new_ids[id] = new_id;
}
__global__ void __launch_bounds__(kBlock) ip_resolve_tombs(int* __restrict__ new_ids, int slots) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= slots) return;
    int v = new_ids[id];
    if (v >= 0) return;
    // (live cells' words are final and never written here; a tombstone's word names its absorber until this thread -- and only it -- replaces it)
    int cur = -1 - v;
    for (;;) {
        const int w = __hip_atomic_load(new_ids + cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w >= 0) { v = w; break; }
        cur = -1 - w;
    }
    new_ids[id] = v;
}

// working records -> the public 32-byte cells
__global__ void __launch_bounds__(kBlock) widen_cells_kernel(const void* __restrict__ narrow, Cell* __restrict__ cells, const Int2* __restrict__ totals) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= totals->a) return;
    CellRec c = CellFmt<true>::load(narrow, id);
    CellFmt<true>::finish(narrow, id, c);
    CellFmt<false>::store(cells, id, c.lo, c.begin, c.hi, c.end);
}

// remap_entries (merge.cu:281-290)
__global__ void __launch_bounds__(kBlock) remap_entries_kernel(uint32_t* __restrict__ entries, const int* __restrict__ new_cell_ids, int num_entries) {
    // four consecutive entries per thread: one 16-byte access each way
    const int id = (blockIdx.x * kBlock + threadIdx.x) * 4;
    if (id >= num_entries) return;
    auto remap = [&](uint32_t e) { return (e & 3u) == 0 ? uint32_t(new_cell_ids[e >> 2]) << 2 : e; };
    if (id + 4 <= num_entries && lb_aligned16(entries + id)) {
        uint4* p = reinterpret_cast<uint4*>(entries + id);
        const uint4 e = *p;
        *p = make_uint4(remap(e.x), remap(e.y), remap(e.z), remap(e.w));
    } else {
        for (int i = id; i < min(id + 4, num_entries); i++) entries[i] = remap(entries[i]);
    }
}

} // namespace

extern "C" int hagrid_merge_grid(hagrid_ctx* ctx, hagrid_grid* grid, float alpha) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    if (!grid->cells || !grid->entries || !grid->ref_ids) HG_FAIL(ctx, HAGRID_EINVAL, "merge_grid: incomplete (or compressed) grid");
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
    ctx->counts.merge_passes = 0;
    ctx->counts.merged_cells = grid->num_cells; ctx->counts.merged_refs = grid->num_refs;
    if (!(alpha > 0)) return HAGRID_OK;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    MergeK k;
    k.top = ivec3(grid->dims[0], grid->dims[1], grid->dims[2]);
    k.dims = k.top << grid->shift;
    k.shift = grid->shift;
    k.cell_size = (vec3(grid->bbox_max[0], grid->bbox_max[1], grid->bbox_max[2]) - vec3(grid->bbox_min[0], grid->bbox_min[1], grid->bbox_min[2])) / vec3(k.dims);

    // buffers sized for the un-merged grid (merge.cu:334-347); cells and refs ping-pong with grid arrays
    const size_t nc0 = size_t(grid->num_cells), nr0 = size_t(grid->num_refs);
    Cell* cells_b = pool_alloc<Cell>(ctx, nc0);
    int* refs_b = pool_alloc<int>(ctx, nr0);
    int* merge_counts = pool_alloc<int>(ctx, nc0 + 1);
    int* nexts = pool_alloc<int>(ctx, nc0 + 1);
    unsigned char* prevs = pool_alloc<unsigned char>(ctx, nc0 + 4);
    unsigned char* cell_flags = pool_alloc<unsigned char>(ctx, nc0 + 4);
    // per cell: the pass that made it (0: the construction); two buffers that swap with the cells.  One allocation: the halves are 256-byte aligned
    const size_t stamp_stride = (nc0 + 4 + 255) & ~size_t(255);
    unsigned char* stamps = pool_alloc<unsigned char>(ctx, 2 * stamp_stride);
    unsigned char* stamps_other = stamps ? stamps + stamp_stride : nullptr;
    unsigned char* const stamps_base = stamps;
    const int max_tiles = grid_blocks(grid->num_cells, kMergeTile);
    Int2* tile_sums = pool_alloc<Int2>(ctx, size_t(max_tiles) + 1);
    Int2* partials = pool_alloc<Int2>(ctx, size_t(scan_num_tiles(max_tiles)) + 1);
    Int2* total = reinterpret_cast<Int2*>(ctx->dscratch);
    auto release = [&]() {
        hagrid_mem_free(ctx, merge_counts); hagrid_mem_free(ctx, nexts); hagrid_mem_free(ctx, prevs); hagrid_mem_free(ctx, cell_flags);
        hagrid_mem_free(ctx, tile_sums); hagrid_mem_free(ctx, partials); hagrid_mem_free(ctx, stamps_base);
    };
    if (!cells_b || !refs_b || !merge_counts || !nexts || !prevs || !cell_flags || !tile_sums || !partials || !stamps) {
        release(); hagrid_mem_free(ctx, cells_b); hagrid_mem_free(ctx, refs_b);
        return HAGRID_ENOMEM;
    }

    void* cells = grid->cells;
    void* cells_other = cells_b;
    int* refs = static_cast<int*>(grid->ref_ids);
    uint32_t* entries = static_cast<uint32_t*>(grid->entries);
    int num_cells = grid->num_cells, num_refs = grid->num_refs;
    const int num_entries = grid->num_entries;
    // 16-byte working records between the passes (CellFmt<true>) when the bounds fit 16 bits
    const bool narrow = ctx->opt_merge_narrow && std::max(k.dims.x, std::max(k.dims.y, k.dims.z)) < 65536;

    int rc = HAGRID_OK;
    int prev_num_cells = 0, iter = 0, pass_tag = 0;
    bool in_narrow = false;                                                // the caller's cells are 32-byte records
    (void)hipMemsetAsync(prevs, 0, nc0, st);                               // tag 0 = never had a predecessor
    hagrid_build_counts& bc = ctx->counts;                                 // sizes that entered the passes (diagnostics)
    auto record = [&](int c, int r) {
        if (bc.merge_passes < HAGRID_MAX_MERGE_PASSES) { bc.merge_cells[bc.merge_passes] = c; bc.merge_refs[bc.merge_passes] = r; bc.merge_passes++; }
    };

    // ---- iterations in place (see the kernels above): state of the mode ----
    // No buffer of its own: the explicit list ends live in `merge_counts`, the per-slot scratch of the mode (merged size / disappearing references,
    // dirty bytes per axis, "evaluated" tags) in the cell buffer that is not in use, and the merged lists go behind the live references of `refs`.
    int global_pass = 0;                                                   // passes run so far (the stamp of the cells the next one makes is global_pass + 1)
    bool have_stamps = false;                                              // `stamps` describes `cells`
    int since[3] = {0, 0, 0};                                              // the pass of the last evaluation of every axis
    bool in_place = false;
    int ip_slots = 0, ip_iters = 0, prev_mask = 0;
    int* list_end = merge_counts;
    Int2* minfo = nullptr; unsigned char* dirty = nullptr; unsigned char* evaluated = nullptr;
    int* books = ctx->dscratch + 16;                                      // cursor, live cells, live references, overflow (1 + axis of the first pass that did not fit)
    int* snap = ctx->dscratch + 20;                                       // live cells / references behind each of the three passes
    Int2* ip_total = reinterpret_cast<Int2*>(ctx->dscratch + 26);
    int* tile_removed = nullptr;                                          // references that disappear, per tile of the pass
    const int ip_capacity = int(std::min<size_t>(nr0, 0x7fffffff));       // both reference buffers hold nr0 ints
    // Enters the mode behind a compacting iteration (cells: num_cells working records, refs: num_refs references, compact).
    size_t ip_dstride = 0;                                                // bytes between the dirty flags of two axes (a multiple of 256: the sweeps load four flags at a time)
    // (returns false -- and the merge goes on compacting -- when the scratch of the mode does not fit the idle cell buffer: grids of a few dozen cells)
    auto ip_enter = [&]() -> bool {
        const auto r256 = [](size_t n) { return (n + 255) & ~size_t(255); };
        const size_t slots = size_t(num_cells), dstride = r256(slots);
        const size_t off_dirty = r256(slots * 8), off_eval = off_dirty + 3 * dstride, off_removed = off_eval + r256(slots);
        if (off_removed + r256(size_t(grid_blocks(num_cells, kIpTile)) * sizeof(int)) > nc0 * sizeof(Cell)) return false;
        ip_slots = num_cells; ip_dstride = dstride;
        char* scratch = static_cast<char*>(cells_other);                   // 32 bytes per cell of the un-merged grid: 12 per slot are used
        minfo = reinterpret_cast<Int2*>(scratch);
        dirty = reinterpret_cast<unsigned char*>(scratch + off_dirty);
        evaluated = reinterpret_cast<unsigned char*>(scratch + off_eval);
        tile_removed = reinterpret_cast<int*>(scratch + off_removed);
        // dirty cells: the ones made after the last evaluation of the axis (their stamps say so) and the cells behind their lower corners
        const unsigned char* stp = have_stamps ? stamps : nullptr;
        ip_begin<<<grid_blocks(ip_slots, kBlock), kBlock, 0, st>>>(cells, ip_slots, list_end, dirty, stp, since[0], since[1], since[2],
                                                                   evaluated, cell_flags /* the mode's `absorbs` tags */, books, num_refs, ip_dstride); HG_DBG(ctx);
        if (stp) ip_mark_entry<<<grid_blocks(ip_slots, kIpTile), kBlock, 0, st>>>(k, reinterpret_cast<const Entry*>(entries), cells, ip_slots, dirty, stp, since[0], since[1], since[2], ip_dstride); HG_DBG(ctx);
        in_place = true;
        return true;
    };
    // Leaves the mode: the live cells become public 32-byte records in `cells_other` (the scratch above is dead by then), their lists go to the
    // reference buffer that is not in use, the voxel map is rewritten once.  Afterwards the grid is in the state a compacting pass with 32-byte
    // output leaves it in.
    auto ip_leave = [&]() -> int {
        const int tiles = grid_blocks(ip_slots, kMergeTile);
        ip_live_sums<<<grid_blocks(tiles, kWaves), kBlock, 0, st>>>(cells, list_end, ip_slots, tile_sums, tiles); HG_DBG(ctx);
        if (!ctx_scan<Int2>(ctx, SumsIn{tile_sums}, SumsOut{tile_sums}, tiles, partials, (const Int2*)nullptr, ip_total)) return HAGRID_ENOMEM;
        ip_compact<<<tiles, kBlock, 0, st>>>(cells, list_end, refs, ip_slots, tile_sums, static_cast<Cell*>(cells_other), refs_b, nexts); HG_DBG(ctx);
        ip_resolve_tombs<<<grid_blocks(ip_slots, kBlock), kBlock, 0, st>>>(nexts, ip_slots); HG_DBG(ctx);
        remap_entries_kernel<<<grid_blocks((num_entries + 3) / 4, kBlock), kBlock, 0, st>>>(entries, nexts, num_entries); HG_DBG(ctx);
        std::swap(cells, cells_other);
        std::swap(refs, refs_b);
        int h[2];
        HG_TRY(read_back(ctx, ip_total, h, sizeof(h)));
        if (h[0] != num_cells) return fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, "merge_grid: the books of the in-place iterations do not add up");
        num_refs = h[1];
        in_place = false; in_narrow = false; ip_iters = 0;
        have_stamps = false;                                               // (the compaction does not carry them: a later entry starts with every cell dirty)
        return HAGRID_OK;
    };
    // The compacting passes of one iteration from `first_axis` on (merge_iteration<axis>, merge.cu:292-329).  They run back to back: the cell count
    // of the second and third pass is only known to the device (the previous pass's scan total); their kernels are launched for the count the first
    // of them starts from and read the real one.  One host round trip per iteration instead of three.
    int last_pass_in = 0, last_pass_out = 0;
    auto compacting_passes = [&](int first_axis, int mask) -> int {
        for (int axis = first_axis; axis < 3; axis++) {
            const int blocks = grid_blocks(num_cells, kBlock);
            Int2* tot = total + axis;
            const int* n_dev = axis > first_axis ? &total[axis - 1].a : nullptr;
            const Entry* ent = reinterpret_cast<const Entry*>(entries);
            if (++pass_tag == 256) { pass_tag = 1; (void)hipMemsetAsync(prevs, 0, nc0, st); }      // tags are bytes
            if (in_narrow) merge_counts_kernel<true><<<blocks, kBlock, 0, st>>>(axis, k, ent, cells, refs, merge_counts, nexts, prevs, pass_tag, mask, num_cells, n_dev);
            else           merge_counts_kernel<false><<<blocks, kBlock, 0, st>>>(axis, k, ent, cells, refs, merge_counts, nexts, prevs, pass_tag, mask, num_cells, n_dev);
            HG_DBG(ctx);
            cell_flags_kernel<<<grid_blocks((num_cells + 3) / 4, kBlock), kBlock, 0, st>>>(nexts, prevs, pass_tag, cell_flags, num_cells, n_dev); HG_DBG(ctx);
            const int tiles = grid_blocks(num_cells, kMergeTile);
            merge_tile_sums<<<grid_blocks(tiles, kWaves), kBlock, 0, st>>>(cell_flags, merge_counts, num_cells, n_dev, tile_sums, tiles); HG_DBG(ctx);
            if (!ctx_scan<Int2>(ctx, SumsIn{tile_sums}, SumsOut{tile_sums}, tiles, partials, (const Int2*)nullptr, tot)) return HAGRID_ENOMEM;
            // (new_cell_ids = nexts: dead after the flags)
            // (stamps are bytes: a merge of more than 250 passes goes on without them -- and without the in-place mode)
            global_pass++;
            const bool stamping = global_pass < 250 && (have_stamps || global_pass == 1);
            const unsigned char* st_in = stamping && have_stamps ? stamps : nullptr;
            unsigned char* st_out = stamping ? stamps_other : nullptr;
            if (in_narrow)   merge_kernel<true, true><<<tiles, kBlock, 0, st>>>(axis, k, ent, cells, refs, cell_flags, tile_sums, merge_counts, nexts, cells_other, refs_b, num_cells, n_dev, tot, st_in, st_out, global_pass);
            else if (narrow) merge_kernel<false, true><<<tiles, kBlock, 0, st>>>(axis, k, ent, cells, refs, cell_flags, tile_sums, merge_counts, nexts, cells_other, refs_b, num_cells, n_dev, tot, st_in, st_out, global_pass);
            else             merge_kernel<false, false><<<tiles, kBlock, 0, st>>>(axis, k, ent, cells, refs, cell_flags, tile_sums, merge_counts, nexts, cells_other, refs_b, num_cells, n_dev, tot, st_in, st_out, global_pass);
            HG_DBG(ctx);
            have_stamps = stamping; since[axis] = global_pass;
            std::swap(stamps, stamps_other);
            in_narrow = narrow;
            remap_entries_kernel<<<grid_blocks((num_entries + 3) / 4, kBlock), kBlock, 0, st>>>(entries, nexts, num_entries); HG_DBG(ctx);
            std::swap(cells, cells_other);
            std::swap(refs, refs_b);
        }
        int h[6];
        HG_TRY(read_back(ctx, total, h, sizeof(int) * 6));
        for (int axis = first_axis; axis < 3; axis++) {
            record(num_cells, num_refs);
            if (axis == 2) { last_pass_in = num_cells; last_pass_out = h[4]; }
            num_cells = h[2 * axis]; num_refs = h[2 * axis + 1];
        }
        return HAGRID_OK;
    };

    do {                                                                   // merge.cu:357-367
        prev_num_cells = num_cells;
        const int mask = iter > 3 ? 0 : (1 << (iter + 1)) - 1;
        if (in_place && ctx->opt_merge_inplace_iters > 0 && ip_iters >= ctx->opt_merge_inplace_iters) {      // (tests: alternate the modes)
            rc = ip_leave();
            if (rc != HAGRID_OK) break;
        }
        int first_compacting_axis = 0;
        if (in_place) {
            // every cell looks again when the mask lets merges through that it held back before (merge.cu:361: from the fifth iteration on)
            if (prev_mask & ~mask) (void)hipMemsetAsync(dirty, 1, 3 * ip_dstride, st);
            const int blocks = grid_blocks(ip_slots, kIpTile), tiles = blocks;
            const Entry* ent = reinterpret_cast<const Entry*>(entries);
            for (int axis = 0; axis < 3 && rc == HAGRID_OK; axis++) {
                global_pass++;
                if (++pass_tag == 256) {                                   // tags are bytes
                    pass_tag = 1;
                    (void)hipMemsetAsync(prevs, 0, nc0, st); (void)hipMemsetAsync(evaluated, 0, size_t(ip_slots), st); (void)hipMemsetAsync(cell_flags, 0, size_t(ip_slots), st);
                }
                ip_counts<<<blocks, kBlock, 0, st>>>(axis, k, ent, cells, list_end, refs, ip_slots, dirty + size_t(axis) * ip_dstride, minfo, nexts, evaluated, prevs, pass_tag, mask); HG_DBG(ctx);
                ip_chains<<<blocks, kBlock, 0, st>>>(ip_slots, nexts, evaluated, prevs, cell_flags, pass_tag); HG_DBG(ctx);
                ip_tile_sums<<<tiles, kBlock, 0, st>>>(cell_flags, pass_tag, minfo, ip_slots, tile_sums, tile_removed); HG_DBG(ctx);
                if (!ctx_scan<Int2>(ctx, SumsIn{tile_sums}, SumsOut{tile_sums}, tiles, partials, (const Int2*)nullptr, ip_total)) { rc = HAGRID_ENOMEM; break; }
                ip_apply<<<tiles, kBlock, 0, st>>>(cells, list_end, refs, cell_flags, pass_tag, minfo, nexts, tile_sums, ip_slots, books, ip_total,
                                                    ctx->opt_merge_inplace_room > 0 ? std::min(ip_capacity, ctx->opt_merge_inplace_room) : ip_capacity, axis); HG_DBG(ctx);
                ip_mark<<<blocks, kBlock, 0, st>>>(k, ent, cells, cell_flags, pass_tag, ip_slots, dirty, ip_total, tile_removed, tiles, books, snap + 2 * axis, ip_dstride); HG_DBG(ctx);
            }
            if (rc != HAGRID_OK) break;
            int h[10];                                                     // books (4), the three snapshots
            rc = read_back(ctx, books, h, sizeof(h));
            if (rc != HAGRID_OK) break;
            const int applied = h[3] > 0 ? h[3] - 1 : 3;                   // the passes before the first one whose lists did not fit
            for (int axis = 0; axis < applied; axis++) { record(num_cells, num_refs); num_cells = h[4 + 2 * axis]; num_refs = h[5 + 2 * axis]; }
            ip_iters++;
            if (applied < 3) {                                             // no room: back to the public arrays; the rest of the iteration compacts
                rc = ip_leave();
                if (rc != HAGRID_OK) break;
                first_compacting_axis = applied;
            }
        }
        if (!in_place) {
            if (first_compacting_axis < 3) rc = compacting_passes(first_compacting_axis, mask);
            if (rc != HAGRID_OK) break;
            // The next iterations in place: once an iteration merges less than half of its cells (working records only: below 65536 per axis).  The
            // first in-place iteration looks at the cells the last compacting one made and at the cells behind them (stamps): about two cells per merge
            // and axis.  1M-triangle soup: the first iteration merges 39 % of the cells, the second 5.7 %, the third 0.4 %; a compacting pass costs ~160 us
            // whatever it merges, a pass in place 75 - 130 us in the second iteration and ~45 us in the third (profiles/NOTES.md "Round 4").
            const int div = ctx->opt_merge_inplace_div > 0 ? ctx->opt_merge_inplace_div : 2;
            if (in_narrow && ctx->opt_merge_inplace && (long long)div * (prev_num_cells - num_cells) < prev_num_cells && num_cells < alpha * prev_num_cells) (void)ip_enter();
        }
        prev_mask = mask;
        iter++;
    } while (rc == HAGRID_OK && num_cells < alpha * prev_num_cells);
    if (rc == HAGRID_OK && in_place) rc = ip_leave();
    else if (rc == HAGRID_OK && in_narrow) {                               // back to the public record
        widen_cells_kernel<<<grid_blocks(num_cells, kBlock), kBlock, 0, st>>>(cells, static_cast<Cell*>(cells_other), total + 2); HG_DBG(ctx);
        std::swap(cells, cells_other);
    }

    if (rc == HAGRID_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, hipGetErrorString(e));
    }
    release();
    hagrid_mem_free(ctx, cells_other);   // whichever buffers are not the live ones (merge.cu:375-376)
    hagrid_mem_free(ctx, refs_b);
    if (rc != HAGRID_OK) {
        // A pass failed half way: the voxel map may already point at the new cell numbers and the cells may be 16-byte working
        // records.  Nothing of that is a grid (the reference aborts here, common.h:103-108): cells and references go back to the
        // pool and the descriptor says so; entries stay the caller's to free.
        (void)hipStreamSynchronize(st);
        hagrid_mem_free(ctx, cells); hagrid_mem_free(ctx, refs);
        grid->cells = nullptr; grid->ref_ids = nullptr; grid->num_cells = 0; grid->num_refs = 0;
        return rc;
    }
    grid->cells = cells; grid->ref_ids = refs;
    grid->num_cells = num_cells; grid->num_refs = num_refs;
    ctx->counts.merged_cells = num_cells; ctx->counts.merged_refs = num_refs;
    return rc;
}

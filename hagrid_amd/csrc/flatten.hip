// flatten.hip -- hagrid_flatten_grid: collapse uniform subtrees of the voxel map and fuse up to three
// octree levels per node, on gfx950.
//
// Replaces the reference's flatten.cu: flatten_grid (:109-175) and the kernels collapse_entries (:9-29),
// compute_depths (:32-46), copy_top_level (:49-62), flatten_level (:65-107).  Bit-identical to the oracle.
//
// Structure: collapse + depth are one kernel per level (deepest first); the per-group scans chain on the
// device through a carry word, so the whole pass needs ONE host round trip (the new entry count);
// flatten_level assigns a whole wavefront to a (2^d)^3 block instead of one workgroup per entry (the
// reference launches one 64-thread block per entry and returns immediately for leaves, flatten.cu:72-76).
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

constexpr int kFlatLevels = 3;   // (1 << Entry::LOG_DIM_BITS) - 1, flatten.cu:6

// collapse_entries + compute_depths for one level; children were finished by the previous launch
__global__ void __launch_bounds__(kBlock) collapse_and_depth(uint32_t* __restrict__ entries, int* __restrict__ depths, int first, int num) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= num) return;
    uint32_t e = entries[first + id];
    int d = 0;
    if (e & 3u) {
        const uint4* p = reinterpret_cast<const uint4*>(entries + (e >> 2));   // 8-entry blocks are 32 B aligned
        const uint4 a = p[0], b = p[1];
        if (a.x == a.y && a.x == a.z && a.x == a.w && a.x == b.x && b.x == b.y && b.x == b.z && b.x == b.w) {
            e = a.x;
            entries[first + id] = e;
        }
    }
    if (e & 3u) {
        const int4* q = reinterpret_cast<const int4*>(depths + (e >> 2));
        const int4 a = q[0], b = q[1];
        d = 1 + max(max(max(a.x, b.x), max(a.y, b.y)), max(max(a.z, b.z), max(a.w, b.w)));
    }
    depths[first + id] = d;
}

struct BlockSizeIn {     // flatten.cu:136-138
    const int* depths;
    __device__ int operator()(int i) const { const int d = depths[i]; return d > 0 ? 1 << (min(d, kFlatLevels) * 3) : 0; }
};
struct StartOut {
    int* start;
    __device__ void operator()(int i, int s) const { start[i] = s; }
};

// copy_top_level (flatten.cu:49-62); start[] already includes the level's offset
__global__ void __launch_bounds__(kBlock) copy_top(const uint32_t* __restrict__ entries, const int* __restrict__ start, const int* __restrict__ depths,
                                                   uint32_t* __restrict__ out, int num) {
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= num) return;
    uint32_t e = entries[id];
    if (e & 3u) e = uint32_t(min(depths[id], kFlatLevels)) | (uint32_t(start[id]) << 2);
    out[id] = e;
}

// flatten_level (flatten.cu:65-107): one wavefront per subtree root, lanes iterate Morton codes
__global__ void __launch_bounds__(kBlock) flatten_level(const uint32_t* __restrict__ entries, const int* __restrict__ start, const int* __restrict__ depths,
                                                        uint32_t* __restrict__ out, int first, int num) {
    const int id = blockIdx.x * kWaves + wave_id();
    if (id >= num) return;
    const int d = min(depths[first + id], kFlatLevels);
    if (d == 0) return;
    const int nsub = 1 << (3 * d);
    const int base = start[first + id];
    const uint32_t root = entries[first + id];
    for (int m = lane_id(); m < nsub; m += 64) {
        int x = 0, y = 0, z = 0, next_id = first + id;
        uint32_t e = root;
        for (int cur = d - 1; cur >= 0; cur--) {
            const int pos = m >> (cur * 3);
            x += (pos & 1) << cur;
            y += ((pos >> 1) & 1) << cur;
            z += ((pos >> 2) & 1) << cur;
            if (e & 3u) { next_id = int(e >> 2) + (pos & 7); e = entries[next_id]; }
        }
        if (e & 3u) e = uint32_t(min(depths[next_id], kFlatLevels)) | (uint32_t(start[next_id]) << 2);
        out[base + x + ((y + (z << d)) << d)] = e;
    }
}

} // namespace

extern "C" int hagrid_flatten_grid(hagrid_ctx* ctx, hagrid_grid* grid) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    if (!grid->entries || grid->num_offsets != grid->shift + 1) HG_FAIL(ctx, HAGRID_EINVAL, "flatten_grid: needs the un-flattened voxel map of build_grid/merge_grid");
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int shift = grid->shift, num_entries = grid->num_entries;
    uint32_t* entries = static_cast<uint32_t*>(grid->entries);
    auto first_of = [&](int i) { return i > 0 ? grid->offsets[i - 1] : 0; };

    int* depths = pool_alloc<int>(ctx, size_t(num_entries) + 8);
    int* start = pool_alloc<int>(ctx, size_t(num_entries) + 8);
    int max_level = 0;
    for (int i = 0; i <= shift; i++) max_level = std::max(max_level, grid->offsets[i] - first_of(i));
    int* partials = pool_alloc<int>(ctx, size_t(scan_num_tiles(max_level)) + 1);
    auto release = [&]() { hagrid_mem_free(ctx, depths); hagrid_mem_free(ctx, start); hagrid_mem_free(ctx, partials); };
    if (!depths || !start || !partials) { release(); return HAGRID_ENOMEM; }

    // collapse + depths, deepest level first (flatten.cu:115-124)
    for (int i = shift; i >= 0; i--) {
        const int first = first_of(i), num = grid->offsets[i] - first;
        if (num > 0) collapse_and_depth<<<grid_blocks(num, kBlock), kBlock, 0, st>>>(entries, depths, first, num);
    }
    // insertion position of every flattened block (flatten.cu:127-141): the scans of levels 0, 3, 6, ... chain
    // through a device carry initialised with the number of top-level entries
    int* carry = ctx->dscratch;                 // carry[0] = offsets[0], carry[1 + g] = end of group g
    const int top_entries = grid->offsets[0];
    HG_HIP(ctx, hipMemcpyAsync(carry, &top_entries, sizeof(int), hipMemcpyHostToDevice, st));
    int groups = 0;
    for (int i = 0; i < shift; i += kFlatLevels, groups++) {
        const int first = first_of(i), num = grid->offsets[i] - first;
        if (!ctx_scan<int>(ctx, BlockSizeIn{depths + first}, StartOut{start + first}, num, partials, carry + groups, carry + groups + 1)) { release(); return HAGRID_ENOMEM; }
    }
    int h[HAGRID_MAX_LEVELS + 2];
    int rc = read_back(ctx, carry, h, sizeof(int) * size_t(groups + 1));
    if (rc != HAGRID_OK) { release(); return rc; }
    const int total_entries = h[groups];
    if (total_entries < top_entries || total_entries > 0x3fffffff) { release(); HG_FAIL(ctx, HAGRID_ERANGE, "flatten_grid: voxel map too large"); }

    uint32_t* out = pool_alloc<uint32_t>(ctx, size_t(total_entries));
    if (!out) { release(); return HAGRID_ENOMEM; }
    copy_top<<<grid_blocks(top_entries, kBlock), kBlock, 0, st>>>(entries, start, depths, out, top_entries); HG_DBG(ctx);
    int new_offsets[HAGRID_MAX_LEVELS], num_new = 0;
    for (int i = 0, g = 0; i < shift; i += kFlatLevels, g++) {
        const int first = first_of(i), num = grid->offsets[i] - first;
        if (num > 0) flatten_level<<<grid_blocks(num, kWaves), kBlock, 0, st>>>(entries, start, depths, out, first, num);
        new_offsets[num_new++] = h[g];          // level_offsets[i]
    }
    new_offsets[num_new++] = total_entries;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { release(); hagrid_mem_free(ctx, out); HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e)); }
    release();
    hagrid_mem_free(ctx, entries);               // flatten.cu:168-170
    ctx->counts.flatten_entries_in = num_entries; ctx->counts.flatten_entries_out = total_entries;
    grid->entries = out;
    grid->num_entries = total_entries;
    grid->num_offsets = num_new;
    for (int i = 0; i < num_new; i++) grid->offsets[i] = new_offsets[i];
    return HAGRID_OK;
}

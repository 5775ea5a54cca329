// kat.hip -- known-answer hooks: the device versions of the L0 functions for the golden-vector tests, the lane <-> ray assignment of
// the tile packets, the records of a traversal image read back voxel by voxel, and a timed instantiation of the headline kernel.
// TEST / DEV INFRASTRUCTURE: built into libhagrid_amd_kat.so (hagrid_amd/build.py), which links against the product library and is
// loaded only by tests/ and tools/dev_*.py; the product library carries none of this.  Declarations: kat/hagrid_amd_kat.h.
#include "../trav_kernels.h"
#include "hagrid_amd_kat.h"

#include <cstring>

using namespace hagrid;
using namespace hagrid_impl;
using namespace hagrid_trav;


namespace {

__global__ void kat_prim_ray(const Tri* tris, const Ray* rays, const int* idx, int n, int* ret, int* hid, float* ht) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h(-1, rays[i].tmax, 0, 0);
    ret[i] = intersect_prim_ray(tris[idx[i]], rays[i], idx[i], h) ? 1 : 0;
    hid[i] = h.id; ht[i] = h.t;
}
__global__ void kat_prim_ray_uvs(const Tri* tris, const Ray* rays, const int* idx, int n, int* ret, int* hid, float* ht, float* hu, float* hv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h(-1, rays[i].tmax, 0, 0);
    ret[i] = intersect_prim_ray_uvs(tris[idx[i]], rays[i], idx[i], h) ? 1 : 0;
    hid[i] = h.id; ht[i] = h.t; hu[i] = h.u; hv[i] = h.v;
}
__global__ void kat_prim_cell(const Tri* tris, const BBox* boxes, const int* idx, int n, int* ret) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ret[i] = intersect_prim_cell(tris[idx[i]], boxes[i]) ? 1 : 0;
}
__global__ void kat_range(const int* dims, const BBox* gbb, const BBox* obb, int n, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Range r = compute_range(ivec3(dims[3 * i], dims[3 * i + 1], dims[3 * i + 2]), gbb[i], obb[i]);
    out[6 * i + 0] = r.lx; out[6 * i + 1] = r.ly; out[6 * i + 2] = r.lz;
    out[6 * i + 3] = r.hx; out[6 * i + 4] = r.hy; out[6 * i + 5] = r.hz;
}
__global__ void kat_grid_dims(const BBox* bb, const int* np, const float* dens, int n, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ivec3 d = compute_grid_dims(bb[i], np[i], dens[i]);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}
__global__ void kat_lookup(const Entry* entries, int shift, ivec3 top, const int* vox, int n, uint32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = lookup_entry(entries, shift, top, ivec3(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]));
}

__global__ void kat_tile_slots(TraverseArgs a, int* out) {     // lane <-> ray assignment of v2, one wavefront per block
    const int w = tile_packet_row_len(a);
    const int b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk_log2) : xcd_split(blockIdx.x, gridDim.x);
    out[blockIdx.x * 64 + threadIdx.x] = tile_packet_slot(a, w, b, threadIdx.x);
}

__global__ void kat_image_records(TraverseArgs a, const int* vox, int n, uint32_t* out, int flat, int slim, int slim_uniform, int general) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int vx = vox[3 * i], vy = vox[3 * i + 1], vz = vox[3 * i + 2];
    const int top = (vx >> a.shift) + a.top_x * ((vy >> a.shift) + a.top_y * (vz >> a.shift));
    const uint2 tab = general ? make_uint2(0u, 0u) : a.img_table[top];
    {     // the slim record, brought into an explicit form: bounds lo | hi << 16 per axis, list length (bit 31: by index, bit 30: through a link), ids or first index
        uint4 r;
        int region = a.shift;                   // general layout: log2 of the region the record's bound bytes count from
        uint32_t links = 0;
        if (!general) {
            const int d = int(tab.y & 3u), sh = a.shift - d, m = (1 << d) - 1;             // (uniform layout: d == shift)
            r = reinterpret_cast<const uint4*>(a.img_blocks)[size_t(tab.x) + size_t(((vx >> sh) & m) + ((((vy >> sh) & m) + (((vz >> sh) & m) << d)) << d))];
        } else {
            // general layout: from the top-level record down through the links, as lookup_entry walks the voxel map (grid.h:103-116)
            const uint32_t none_ = (1u << slim) - 1u; const int last_ = 48 + (80 / slim - 1) * slim;
            // (from the image's virtual top level, where the kernels' look-ups start: a.gen_*)
            region = a.gen_shift;
            r = reinterpret_cast<const uint4*>(a.img_blocks)[size_t(a.gen_base) + size_t(vx >> region) + size_t(a.gen_x) * size_t(vy >> region) + size_t(a.gen_xy) * size_t(vz >> region)];
            while (((r.w >> (last_ - 96)) & none_) == none_ - 2u) {
                const int k = int((r.z >> 16) & 3u), m = (1 << k) - 1;
                region += int((r.z >> 18) & 3u) - k; links++;
                const uint32_t first = (r.y >> 16) | (r.z << 16);
                r = reinterpret_cast<const uint4*>(a.img_blocks)[size_t(first) + size_t(((vx >> region) & m) + ((((vy >> region) & m) + (((vz >> region) & m) << k)) << k))];
            }
        }
        const uint32_t w[5] = {r.x, r.y, r.z, r.w, 0u};
        auto field = [&](int pos, int nb) -> uint32_t {
            const int wi = pos >> 5, o = pos & 31;
            unsigned long long v = (static_cast<unsigned long long>(w[wi + 1]) << 32 | w[wi]) >> o;
            return nb == 32 ? uint32_t(v) : uint32_t(v) & ((1u << nb) - 1u);
        };
        const int ni = 80 / slim;
        const uint32_t none = (1u << slim) - 1u;
        uint32_t* o = out + 8 * size_t(i);
        if (slim_uniform) {
            o[0] = uint32_t(vx - int(field(0, 8))) | uint32_t(vx + int(field(8, 8))) << 16;
            o[1] = uint32_t(vy - int(field(16, 8))) | uint32_t(vy + int(field(24, 8))) << 16;
            o[2] = uint32_t(vz - int(field(32, 8))) | uint32_t(vz + int(field(40, 8))) << 16;
        } else if (!general) {       // table layout: biased offsets from the origin of the top-level cell
            const int om = ~((1 << a.shift) - 1);
            o[0] = uint32_t((vx & om) + int(field(0, 8)) - 128) | uint32_t((vx & om) + int(field(8, 8)) - 128) << 16;
            o[1] = uint32_t((vy & om) + int(field(16, 8)) - 128) | uint32_t((vy & om) + int(field(24, 8)) - 128) << 16;
            o[2] = uint32_t((vz & om) + int(field(32, 8)) - 128) | uint32_t((vz & om) + int(field(40, 8)) - 128) << 16;
        } else {
            const int om = int(~0u << region);
            o[0] = uint32_t((vx & om) - int(field(0, 8))) | uint32_t((vx & om) + int(field(8, 8))) << 16;
            o[1] = uint32_t((vy & om) - int(field(16, 8))) | uint32_t((vy & om) + int(field(24, 8))) << 16;
            o[2] = uint32_t((vz & om) - int(field(32, 8))) | uint32_t((vz & om) + int(field(40, 8))) << 16;
        }
        const uint32_t marker = field(48 + (ni - 1) * slim, slim);
        const bool wide = !slim_uniform && marker == none - 3u;
        if (marker == none - 1u || wide) {
            const uint32_t cnt = field(80, 20);
            uint32_t first = field(48, 32);
            if (wide) {       // the cell's bounds and first reference index come from its wide record
                const uint4 wr = reinterpret_cast<const uint4*>(a.img_table)[first];
                o[0] = wr.x; o[1] = wr.y; o[2] = wr.z; first = wr.w;
            }
            // (the explicit form holds lists of at most four ids inline: read them through the index)
            o[3] = cnt | (cnt > 4 ? 0x80000000u : 0u) | (links ? 0x40000000u : 0u);
            if (cnt > 4) { o[4] = first; o[5] = o[6] = o[7] = 0u; }
            else for (uint32_t j = 0; j < 4; j++) o[4 + j] = j < cnt ? uint32_t(a.refs[first + j]) : ~0u;
        } else {
            uint32_t cnt = 0;
            for (int j = 0; j < 4; j++) {
                const uint32_t id = j < ni ? field(48 + j * slim, slim) : none;
                o[4 + j] = id == none ? ~0u : id;
                if (id != none) cnt++;
            }
            o[3] = cnt | (links ? 0x40000000u : 0u);        // bit 30: came through a link
        }
        return;
    }
}

struct Staged {   // host array staged on the device through the pool
    hagrid_ctx* ctx; void* d = nullptr; size_t bytes;
    Staged(hagrid_ctx* c, const void* h, size_t b) : ctx(c), bytes(b) {
        d = hagrid_mem_alloc(ctx, b);
        if (d && h) hagrid_mem_copy_h2d(ctx, d, h, b);
    }
    ~Staged() { hagrid_mem_free(ctx, d); }
    int fetch(void* h) { return hagrid_mem_copy_d2h(ctx, h, d, bytes); }
};

} // namespace

extern "C" int hagrid_kat_intersect_prim_ray(hagrid_ctx* ctx, const void* tris, const void* rays, const int32_t* tri_index,
                                             int n, int32_t* ret, int32_t* hit_id, float* hit_t) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    int max_idx = 0;
    for (int i = 0; i < n; i++) max_idx = std::max(max_idx, tri_index[i]);
    Staged t(ctx, tris, size_t(max_idx + 1) * 48), r(ctx, rays, size_t(n) * 32), ix(ctx, tri_index, size_t(n) * 4);
    Staged o0(ctx, nullptr, size_t(n) * 4), o1(ctx, nullptr, size_t(n) * 4), o2(ctx, nullptr, size_t(n) * 4);
    kat_prim_ray<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Tri*)t.d, (const Ray*)r.d, (const int*)ix.d, n, (int*)o0.d, (int*)o1.d, (float*)o2.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    HG_TRY(o0.fetch(ret)); HG_TRY(o1.fetch(hit_id)); HG_TRY(o2.fetch(hit_t));
    return HAGRID_OK;
}

extern "C" int hagrid_kat_intersect_prim_ray_uvs(hagrid_ctx* ctx, const void* tris, const void* rays, const int32_t* tri_index,
                                                 int n, int32_t* ret, int32_t* hit_id, float* hit_t, float* hit_u, float* hit_v) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    int max_idx = 0;
    for (int i = 0; i < n; i++) max_idx = std::max(max_idx, tri_index[i]);
    Staged t(ctx, tris, size_t(max_idx + 1) * 48), r(ctx, rays, size_t(n) * 32), ix(ctx, tri_index, size_t(n) * 4);
    Staged o0(ctx, nullptr, size_t(n) * 4), o1(ctx, nullptr, size_t(n) * 4), o2(ctx, nullptr, size_t(n) * 4), o3(ctx, nullptr, size_t(n) * 4), o4(ctx, nullptr, size_t(n) * 4);
    kat_prim_ray_uvs<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Tri*)t.d, (const Ray*)r.d, (const int*)ix.d, n, (int*)o0.d, (int*)o1.d, (float*)o2.d, (float*)o3.d, (float*)o4.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    HG_TRY(o0.fetch(ret)); HG_TRY(o1.fetch(hit_id)); HG_TRY(o2.fetch(hit_t)); HG_TRY(o3.fetch(hit_u)); HG_TRY(o4.fetch(hit_v));
    return HAGRID_OK;
}

extern "C" int hagrid_kat_intersect_prim_cell(hagrid_ctx* ctx, const void* tris, const void* boxes, const int32_t* tri_index, int n, int32_t* ret) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    int max_idx = 0;
    for (int i = 0; i < n; i++) max_idx = std::max(max_idx, tri_index[i]);
    Staged t(ctx, tris, size_t(max_idx + 1) * 48), b(ctx, boxes, size_t(n) * 32), ix(ctx, tri_index, size_t(n) * 4), o(ctx, nullptr, size_t(n) * 4);
    kat_prim_cell<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Tri*)t.d, (const BBox*)b.d, (const int*)ix.d, n, (int*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(ret);
}

extern "C" int hagrid_kat_compute_range(hagrid_ctx* ctx, const int32_t* dims3, const void* grid_bb, const void* obj_bb, int n, int32_t* out6) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    Staged d(ctx, dims3, size_t(n) * 12), g(ctx, grid_bb, size_t(n) * 32), ob(ctx, obj_bb, size_t(n) * 32), o(ctx, nullptr, size_t(n) * 24);
    kat_range<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const int*)d.d, (const BBox*)g.d, (const BBox*)ob.d, n, (int*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out6);
}

extern "C" int hagrid_kat_compute_grid_dims(hagrid_ctx* ctx, const void* bb, const int32_t* num_prims, const float* density, int n, int32_t* out3) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    Staged b(ctx, bb, size_t(n) * 32), np(ctx, num_prims, size_t(n) * 4), de(ctx, density, size_t(n) * 4), o(ctx, nullptr, size_t(n) * 12);
    kat_grid_dims<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const BBox*)b.d, (const int*)np.d, (const float*)de.d, n, (int*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out3);
}

extern "C" int hagrid_kat_lookup_entry(hagrid_ctx* ctx, const uint32_t* entries, int num_entries, int shift, const int32_t* top_dims3,
                                       const int32_t* voxels3, int n, uint32_t* out) {
    if (!ctx || n <= 0 || num_entries <= 0) return HAGRID_EINVAL;
    Staged e(ctx, entries, size_t(num_entries) * 4), v(ctx, voxels3, size_t(n) * 12), o(ctx, nullptr, size_t(n) * 4);
    kat_lookup<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Entry*)e.d, shift, ivec3(top_dims3[0], top_dims3[1], top_dims3[2]), (const int*)v.d, n, (uint32_t*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out);
}

extern "C" int hagrid_kat_detect_ray_rows(hagrid_ctx* ctx, const void* rays_dev, int num_rays, float bbox_diag, int32_t* row_len) {
    if (!ctx || !rays_dev || num_rays < 0 || !row_len) return HAGRID_EINVAL;
    int* d = ctx->dscratch + 232;
    TraverseArgs a;
    memset(&a, 0, sizeof(a));
    a.rays = static_cast<const float4*>(rays_dev);
    a.max_x = bbox_diag; a.min_x = 0.0f;                 // only the diagonal matters
    launch_detect(ctx, a, num_rays, d, 65536);            // the test hook runs the origin criterion from 64k rays on
    HG_HIP(ctx, hipGetLastError());
    HG_TRY(read_back(ctx, d, row_len, sizeof(int)));
    *row_len &= kRowLenMask;                               // (bit 30: found from the origins alone)
    return HAGRID_OK;
}

extern "C" int hagrid_kat_tile_slots(hagrid_ctx* ctx, int num_rays, int row_len, int super_log2, int xcd_chunk_log2, int32_t* slots) {
    if (!ctx || num_rays <= 0 || !slots || super_log2 < 0 || (super_log2 & 0xff) > 8) return HAGRID_EINVAL;          // (bits 8..: rows of super-tiles per band)
    const int blocks = grid_blocks(num_rays, 64);
    TraverseArgs a;
    memset(&a, 0, sizeof(a));
    a.num_rays = num_rays; a.row_len_hint = row_len; a.super_log2 = super_log2 & 0xff; a.band_rows = super_log2 >> 8; a.xcd_chunk_log2 = xcd_chunk_log2;
    Staged o(ctx, nullptr, size_t(blocks) * 64 * 4);
    if (!o.d) return HAGRID_ENOMEM;
    kat_tile_slots<<<blocks, 64, 0, ctx->stream>>>(a, (int*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(slots);
}

extern "C" int hagrid_kat_image_records(hagrid_ctx* ctx, const hagrid_grid* grid, const int32_t* voxels3, int n, uint32_t* records8, int64_t* image_bytes) {
    if (!ctx || !grid || n < 0) return HAGRID_EINVAL;
    if (!trav_image_matches(ctx, grid)) HG_FAIL(ctx, HAGRID_EINVAL, "no traversal image for this grid");
    if (image_bytes) *image_bytes = (int64_t)ctx->image.block_bytes + (int64_t)ctx->image.table_bytes;
    if (n == 0) return HAGRID_OK;
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, nullptr, nullptr, nullptr, 0, a));
    image_args(ctx, a);
    // staging must not disturb the image: these buffers are not grid arrays
    Staged v(ctx, voxels3, size_t(n) * 12), o(ctx, nullptr, size_t(n) * 32);
    if (!v.d || !o.d) return HAGRID_ENOMEM;
    kat_image_records<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>(a, (const int*)v.d, n, (uint32_t*)o.d, 1, ctx->image.slim, ctx->image.uniform ? 1 : 0, ctx->image.general ? 1 : 0); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(records8);
}

// One launch of the headline kernel (uniform slim-20 image, nearest hit) in its TIMES instantiation: times[2 b] / times[2 b + 1] = wall
// clock (100 MHz) at the start of block b and when its last lane left; tile_order (optional) = the tile block b traverses.
// tail = 0: the plain slim kernel.  row_len = the image width of the batch (tile packets).  tools/dev_wave_timeline.py.
extern "C" int hagrid_kat_traverse_timed(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, const void* rays, void* hits, int num_rays,
                                         int row_len, int tail, unsigned long long* times_dev, const int* tile_order_dev) {
    if (!ctx || !grid || !times_dev || num_rays <= 0) return HAGRID_EINVAL;
    if (!trav_image_matches(ctx, grid) || !(ctx->image.uniform && ctx->image.slim == 20))
        HG_FAIL(ctx, HAGRID_EINVAL, "kat_traverse_timed: needs the table-free image with 20-bit slim records of this grid");
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, tris, rays, hits, num_rays, a));
    image_args(ctx, a);
    a.row_len_hint = row_len; a.wave_times = times_dev; a.tile_order = tile_order_dev;
    const int blocks = grid_blocks(num_rays, 64);
    if (tail) traverse_kernel_tail<20, true><<<blocks, 64, 0, ctx->stream>>>(a);
    else      traverse_kernel_img<0, 20, 0, true><<<blocks, 64, 0, ctx->stream>>>(a);
    HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return HAGRID_OK;
}

// Which of several equivalent code paths runs: forced by the parity tests (every path must give the oracle's hits / arrays) and swept by
// tools/dev_option_sweep.py.  Not options of the product (include/hagrid_amd.h: hagrid_set_option); hits never depend on them.
extern "C" int hagrid_kat_set_option(hagrid_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return HAGRID_EINVAL;
    struct { const char* name; int* dst; int lo, hi; } table[] = {
        {"traverse.variant", &ctx->opt_variant, 0, 4},              {"traverse.narrow", &ctx->opt_narrow, 0, 1},
        {"traverse.image_uniform", &ctx->opt_image_uniform, 0, 2},  {"traverse.image_slim", &ctx->opt_image_slim, 1, 2}, {"traverse.image_general", &ctx->opt_image_general, 0, 2}, {"traverse.image_vtop", &ctx->opt_image_vtop, 0, 1}, {"traverse.quad_head", &ctx->opt_quad_head, 0, 1000}, {"ctx.fast_readback", &ctx->opt_fast_readback, 0, 1},
        {"traverse.tail", &ctx->opt_tail, 0, 1},                    {"traverse.quad_tail", &ctx->opt_quad_tail, -1, 100},
        {"traverse.tail_dual", &ctx->opt_tail_dual, -1, 1},        
        {"traverse.super_tile", &ctx->opt_super_log2, 0, 8},        {"traverse.xcd_chunk", &ctx->opt_xcd_chunk_log2, -2, 16},
        {"traverse.row_cache", &ctx->opt_row_cache, 0, 1},          {"traverse.lds_pad", &ctx->opt_lds_pad, 0, 65536},
        {"merge.narrow_cells", &ctx->opt_merge_narrow, 0, 1},       {"scan.lookback", &ctx->opt_lookback, 1, 2},
        {"traverse.order_gate", &ctx->opt_order_gate, 0, 1}, {"traverse.share_trial", &ctx->opt_share_trial, 0, 1}, {"traverse.band_rows", &ctx->opt_band_rows, 0, 1 << 16}, {"traverse.mailbox", &ctx->opt_mailbox, -1, 1},
        {"merge.inplace", &ctx->opt_merge_inplace, 0, 1},           {"merge.inplace_iters", &ctx->opt_merge_inplace_iters, 0, 1 << 20}, {"merge.inplace_room", &ctx->opt_merge_inplace_room, 0, 0x7fffffff}, {"merge.inplace_div", &ctx->opt_merge_inplace_div, 0, 1 << 20},
        {"expand.voxel_map", &ctx->opt_expand_voxel_map, 0, 1},
    };
    for (auto& t : table)
        if (!strcmp(key, t.name)) {
            if (value < t.lo || value > t.hi || (t.dst == &ctx->opt_variant && value == 3)) HG_FAIL(ctx, HAGRID_EINVAL, "kat_set_option: value out of range");
            *t.dst = value;
            return HAGRID_OK;
        }
    return hagrid_set_option(ctx, key, value);              // the product's own options
}

extern "C" int hagrid_kat_order_state(hagrid_ctx* ctx, const void* rays, int32_t* out12, float* ms2 /* 4 floats */) {
    if (!ctx || !out12) return HAGRID_EINVAL;
    for (int i = 0; i < 12; i++) out12[i] = 0;
    out12[0] = -1;
    for (int i = 0; i < hagrid_ctx::kRayHints; i++) {
        const hagrid_ctx::RayHints& h = ctx->hints[i];
        if (h.key_rays != rays) continue;
        const int v[12] = {i, h.lpt_valid, h.cooldown > 0, h.lpt_rot, h.head_disabled, h.n_base, h.n_head, h.share_choice >= 0 ? h.share_cands[h.share_choice] : -1, h.share_done + 100 * h.share_issued + 10000 * int(h.order_loses), h.n_all + 10 * int(h.learned_all) + 100 * h.cooldown, __atomic_load_n(ctx->mailbox + kMbxHeadSuggest + i, __ATOMIC_RELAXED), h.share_launches};
        for (int k = 0; k < 12; k++) out12[k] = v[k];
        if (ms2) { ms2[0] = h.t_base; ms2[1] = h.t_head; ms2[2] = h.t_all; ms2[3] = h.share_choice >= 0 ? h.share_t[h.share_choice] : 0.0f; }
        return HAGRID_OK;
    }
    return HAGRID_OK;
}

extern "C" int hagrid_kat_forget_hints(hagrid_ctx* ctx) {
    if (!ctx) return HAGRID_EINVAL;
    for (auto& h : ctx->hints) { h.key_rays = nullptr; h.key_n = 0; h.used = 0; h.lpt_rays = nullptr; h.lpt_valid = false; h.rowlen_rays = nullptr; h.rowlen_seen = 0; h.share_ncand = 0; h.share_shape_nc = 0; h.share_choice = -1; }
    return HAGRID_OK;
}

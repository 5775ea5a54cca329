// traverse.hip -- gfx950 ray traversal of the irregular grid.
//
// Replaces the reference's traverse.cu: setup_traversal (:97-109), traverse_grid (:111-117) and the
// traverse<CellT, Tri> kernel (:27-95) with intersect_ray_box (:14-21) and compute_voxel (:23-25).
// Results per ray are identical to the CPU oracle's (same IEEE operation sequence, contraction off):
// the primitive id of the nearest hit (-1 on a miss) and its distance t.
//
// Design notes (MI355X): one ray per lane, 64-lane wavefronts, 256-thread workgroups.  The grid
// constants travel as kernel arguments (scalar registers), not as __constant__ symbols, so several
// grids / contexts can traverse concurrently.  Ray and hit records are 32 B / 16 B per lane and are
// moved as 16-byte vector accesses.  See DESIGN.md for the algorithmic-byte accounting.
#include "ctx.h"

#include "hagrid/grid.h"
#include "hagrid/prims.h"
#include "hagrid/ray.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct TraverseArgs {
    const uint32_t* __restrict__ entries;
    const void* __restrict__ cells;
    const int* __restrict__ refs;
    const float4* __restrict__ tris;
    const float4* __restrict__ rays;
    float4* __restrict__ hits;
    int* __restrict__ steps;                 // optional per-ray step counter
    unsigned long long* __restrict__ stats;  // optional 7 batch counters
    int num_rays;
    int shift;
    int dims_x, dims_y, dims_z;   // virtual resolution dims << shift
    int top_x, top_y;             // top-level resolution (x, y)
    float min_x, min_y, min_z;    // grid box
    float max_x, max_y, max_z;
    float cs_x, cs_y, cs_z;       // cell size
    float inv_x, inv_y, inv_z;    // 1 / cell size (as dims / extents)
};

struct CellBox { int lx, ly, lz, hx, hy, hz, begin, end; };

template <bool SMALL>
__device__ __forceinline__ CellBox load_cell_box(const void* __restrict__ cells, uint32_t index) {
    CellBox c;
    if (SMALL) {
        const uint4 w = reinterpret_cast<const uint4*>(cells)[index];
        c.lx = int(w.x & 0xffffu); c.ly = int(w.x >> 16); c.lz = int(w.y & 0xffffu);
        c.hx = int(w.y >> 16); c.hy = int(w.z & 0xffffu); c.hz = int(w.z >> 16);
        c.begin = int(w.w); c.end = 0;
    } else {
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(index);
        const int4 a = p[0], b = p[1];
        c.lx = a.x; c.ly = a.y; c.lz = a.z; c.begin = a.w;
        c.hx = b.x; c.hy = b.y; c.hz = b.z; c.end = b.w;
    }
    return c;
}

__device__ __forceinline__ Tri load_tri(const float4* __restrict__ tris, int ref) {
    const float4* p = tris + 3 * size_t(ref);
    const float4 a = p[0], b = p[1], c = p[2];
    return Tri(vec3(a.x, a.y, a.z), a.w, vec3(b.x, b.y, b.z), b.w, vec3(c.x, c.y, c.z), c.w);
}

template <bool SMALL, bool STATS>
__global__ void __launch_bounds__(256) traverse_kernel(const TraverseArgs a) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= a.num_rays) return;

    const float4 r0 = a.rays[2 * size_t(id)], r1 = a.rays[2 * size_t(id) + 1];
    const vec3 org(r0.x, r0.y, r0.z), dir(r1.x, r1.y, r1.z);
    const float tmin = r0.w, tmax = r1.w;
    const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 csize(a.cs_x, a.cs_y, a.cs_z), ginv(a.inv_x, a.inv_y, a.inv_z);
    const bool px = dir.x >= 0.0f, py = dir.y >= 0.0f, pz = dir.z >= 0.0f;

    // slab test against the grid box
    const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
    const vec3 t0 = min(ta, tb), t1 = max(ta, tb);
    const float tstart = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), tmin);
    const float tend = detail::fmin2(detail::fmin2(t1.x, detail::fmin2(t1.y, t1.z)), tmax);

    Hit hit(-1, tmax, 0.0f, 0.0f);
    int steps = 0;
    unsigned n_cells = 0, n_words = 0, n_refs = 0, n_sent = 0;

    if (!(tstart > tend)) {
        const vec3 fv = (tstart * dir + org - gmin) * ginv;
        int vx = min(max(int(fv.x), 0), a.dims_x - 1);
        int vy = min(max(int(fv.y), 0), a.dims_y - 1);
        int vz = min(max(int(fv.z), 0), a.dims_z - 1);

        for (;;) {
            // voxel map walk
            uint32_t w = a.entries[(vx >> a.shift) + a.top_x * ((vy >> a.shift) + a.top_y * (vz >> a.shift))];
            int depth = 0;
            if (STATS) n_words++;
            while (w & 3u) {
                const int k = int(w & 3u);
                depth += k;
                const int s = a.shift - depth, m = (1 << k) - 1;
                w = a.entries[(w >> 2) + ((vx >> s) & m) + ((((vy >> s) & m) + (((vz >> s) & m) << k)) << k)];
                if (STATS) n_words++;
            }
            const CellBox c = load_cell_box<SMALL>(a.cells, w >> 2);

            // exit plane of the cell along the ray
            const int cx = px ? c.hx : c.lx, cy = py ? c.hy : c.ly, cz = pz ? c.hz : c.lz;
            const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
            const float texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));

            // next voxel, never moving backwards
            const vec3 ev = (texit * dir + org - gmin) * ginv;
            const int nx = texit == tcell.x ? cx + (px ? 0 : -1) : int(ev.x);
            const int ny = texit == tcell.y ? cy + (py ? 0 : -1) : int(ev.y);
            const int nz = texit == tcell.z ? cz + (pz ? 0 : -1) : int(ev.z);
            vx = px ? max(nx, vx) : min(nx, vx);
            vy = py ? max(ny, vy) : min(ny, vy);
            vz = pz ? max(nz, vz) : min(nz, vz);

            // the cell's triangles
            int consumed = 0;
            if (SMALL) {
                if (c.begin >= 0) {
                    int cur = c.begin;
                    int ref = a.refs[cur++];
                    while (ref >= 0) {
                        const int next = a.refs[cur++];
                        intersect_prim_ray(load_tri(a.tris, ref), Ray(org, tmin, dir, hit.t), ref, hit);
                        ref = next;
                    }
                    consumed = cur - c.begin;
                    if (STATS) { n_refs += unsigned(consumed - 1); n_sent++; }
                }
            } else {
                int cur = c.begin;
                int ref = cur < c.end ? a.refs[cur++] : -1;
                while (ref >= 0) {
                    const int next = cur < c.end ? a.refs[cur++] : -1;
                    intersect_prim_ray(load_tri(a.tris, ref), Ray(org, tmin, dir, hit.t), ref, hit);
                    ref = next;
                }
                consumed = c.end - c.begin;
                if (STATS) n_refs += unsigned(consumed);
            }
            steps += 1 + consumed;
            if (STATS) n_cells++;

            if (hit.t <= texit || ((vx < 0) | (vx >= a.dims_x) | (vy < 0) | (vy >= a.dims_y) | (vz < 0) | (vz >= a.dims_z))) break;
        }
    }

    a.hits[id] = make_float4(__int_as_float(hit.id), hit.t, 0.0f, 0.0f);

    if (STATS) {
        if (a.steps) a.steps[id] = steps;
        if (a.stats) {
            atomicAdd(a.stats + 0, 1ull);
            atomicAdd(a.stats + 1, (unsigned long long)(!(tstart > tend)));
            atomicAdd(a.stats + 2, (unsigned long long)n_cells);
            atomicAdd(a.stats + 3, (unsigned long long)n_words);
            atomicAdd(a.stats + 4, (unsigned long long)n_refs);
            atomicAdd(a.stats + 5, (unsigned long long)n_sent);
            atomicAdd(a.stats + 6, (unsigned long long)(hit.id >= 0));
        }
    }
}

int make_args(hagrid_ctx* ctx, const hagrid_grid* g, const void* tris, const void* rays, void* hits, int num_rays, TraverseArgs& a) {
    if (!g || !g->entries || !g->ref_ids || (!g->cells && !g->small_cells)) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: incomplete grid");
    if (num_rays < 0 || (num_rays > 0 && (!rays || !hits || !tris))) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: null buffer");
    if (g->shift < 0 || g->shift > 15) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: bad shift");
    // setup_traversal (traverse.cu:97-109)
    const vec3 lo(g->bbox_min[0], g->bbox_min[1], g->bbox_min[2]), hi(g->bbox_max[0], g->bbox_max[1], g->bbox_max[2]);
    const vec3 ext = hi - lo;
    const ivec3 dims = ivec3(g->dims[0], g->dims[1], g->dims[2]) << g->shift;
    const vec3 ginv = vec3(dims) / ext;
    const vec3 cs = ext / vec3(dims);
    a.entries = static_cast<const uint32_t*>(g->entries);
    a.cells = g->small_cells ? g->small_cells : g->cells;
    a.refs = static_cast<const int*>(g->ref_ids);
    a.tris = static_cast<const float4*>(tris);
    a.rays = static_cast<const float4*>(rays);
    a.hits = static_cast<float4*>(hits);
    a.steps = nullptr; a.stats = nullptr;
    a.num_rays = num_rays; a.shift = g->shift;
    a.dims_x = dims.x; a.dims_y = dims.y; a.dims_z = dims.z;
    a.top_x = g->dims[0]; a.top_y = g->dims[1];
    a.min_x = lo.x; a.min_y = lo.y; a.min_z = lo.z;
    a.max_x = hi.x; a.max_y = hi.y; a.max_z = hi.z;
    a.cs_x = cs.x; a.cs_y = cs.y; a.cs_z = cs.z;
    a.inv_x = ginv.x; a.inv_y = ginv.y; a.inv_z = ginv.z;
    return HAGRID_OK;
}

} // namespace

extern "C" int hagrid_setup_traversal(hagrid_ctx* ctx, const hagrid_grid* grid) {
    if (!ctx) return HAGRID_EINVAL;
    TraverseArgs a;
    return make_args(ctx, grid, nullptr, nullptr, nullptr, 0, a);
}

extern "C" int hagrid_traverse_grid(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                                    const void* rays, void* hits, int num_rays) {
    if (!ctx) return HAGRID_EINVAL;
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, tris, rays, hits, num_rays, a));
    if (num_rays == 0) return HAGRID_OK;
    const int blocks = grid_blocks(num_rays, 256);
    if (grid->small_cells) traverse_kernel<true, false><<<blocks, 256, 0, ctx->stream>>>(a);
    else                   traverse_kernel<false, false><<<blocks, 256, 0, ctx->stream>>>(a);
    HG_HIP(ctx, hipGetLastError());
    return HAGRID_OK;
}

extern "C" int hagrid_traverse_grid_stats(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                                          const void* rays, void* hits, int num_rays,
                                          void* steps, hagrid_traversal_stats* stats) {
    if (!ctx) return HAGRID_EINVAL;
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, tris, rays, hits, num_rays, a));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (num_rays == 0) return HAGRID_OK;
    unsigned long long* dstats = nullptr;
    if (stats) {
        dstats = pool_alloc<unsigned long long>(ctx, 8);
        if (!dstats) return HAGRID_ENOMEM;
        HG_HIP(ctx, hipMemsetAsync(dstats, 0, 8 * sizeof(unsigned long long), ctx->stream));
    }
    a.steps = static_cast<int*>(steps);
    a.stats = dstats;
    const int blocks = grid_blocks(num_rays, 256);
    if (grid->small_cells) traverse_kernel<true, true><<<blocks, 256, 0, ctx->stream>>>(a);
    else                   traverse_kernel<false, true><<<blocks, 256, 0, ctx->stream>>>(a);
    HG_HIP(ctx, hipGetLastError());
    if (stats) {
        unsigned long long h[8];
        HG_TRY(read_back(ctx, dstats, h, sizeof(h)));
        stats->rays = (int64_t)h[0]; stats->rays_hit_grid = (int64_t)h[1]; stats->cells = (int64_t)h[2];
        stats->entry_words = (int64_t)h[3]; stats->refs = (int64_t)h[4]; stats->sentinels = (int64_t)h[5];
        stats->hits = (int64_t)h[6];
        hagrid_mem_free(ctx, dstats);
    } else {
        HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HAGRID_OK;
}

// ---- known-answer hooks: the device versions of the L0 functions, for the golden-vector tests --------

namespace {

__global__ void kat_prim_ray(const Tri* tris, const Ray* rays, const int* idx, int n, int* ret, int* hid, float* ht) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h(-1, rays[i].tmax, 0, 0);
    ret[i] = intersect_prim_ray(tris[idx[i]], rays[i], idx[i], h) ? 1 : 0;
    hid[i] = h.id; ht[i] = h.t;
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
    kat_prim_ray<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Tri*)t.d, (const Ray*)r.d, (const int*)ix.d, n, (int*)o0.d, (int*)o1.d, (float*)o2.d);
    HG_HIP(ctx, hipGetLastError());
    HG_TRY(o0.fetch(ret)); HG_TRY(o1.fetch(hit_id)); HG_TRY(o2.fetch(hit_t));
    return HAGRID_OK;
}

extern "C" int hagrid_kat_intersect_prim_cell(hagrid_ctx* ctx, const void* tris, const void* boxes, const int32_t* tri_index, int n, int32_t* ret) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    int max_idx = 0;
    for (int i = 0; i < n; i++) max_idx = std::max(max_idx, tri_index[i]);
    Staged t(ctx, tris, size_t(max_idx + 1) * 48), b(ctx, boxes, size_t(n) * 32), ix(ctx, tri_index, size_t(n) * 4), o(ctx, nullptr, size_t(n) * 4);
    kat_prim_cell<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Tri*)t.d, (const BBox*)b.d, (const int*)ix.d, n, (int*)o.d);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(ret);
}

extern "C" int hagrid_kat_compute_range(hagrid_ctx* ctx, const int32_t* dims3, const void* grid_bb, const void* obj_bb, int n, int32_t* out6) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    Staged d(ctx, dims3, size_t(n) * 12), g(ctx, grid_bb, size_t(n) * 32), ob(ctx, obj_bb, size_t(n) * 32), o(ctx, nullptr, size_t(n) * 24);
    kat_range<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const int*)d.d, (const BBox*)g.d, (const BBox*)ob.d, n, (int*)o.d);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out6);
}

extern "C" int hagrid_kat_compute_grid_dims(hagrid_ctx* ctx, const void* bb, const int32_t* num_prims, const float* density, int n, int32_t* out3) {
    if (!ctx || n <= 0) return HAGRID_EINVAL;
    Staged b(ctx, bb, size_t(n) * 32), np(ctx, num_prims, size_t(n) * 4), de(ctx, density, size_t(n) * 4), o(ctx, nullptr, size_t(n) * 12);
    kat_grid_dims<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const BBox*)b.d, (const int*)np.d, (const float*)de.d, n, (int*)o.d);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out3);
}

extern "C" int hagrid_kat_lookup_entry(hagrid_ctx* ctx, const uint32_t* entries, int num_entries, int shift, const int32_t* top_dims3,
                                       const int32_t* voxels3, int n, uint32_t* out) {
    if (!ctx || n <= 0 || num_entries <= 0) return HAGRID_EINVAL;
    Staged e(ctx, entries, size_t(num_entries) * 4), v(ctx, voxels3, size_t(n) * 12), o(ctx, nullptr, size_t(n) * 4);
    kat_lookup<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>((const Entry*)e.d, shift, ivec3(top_dims3[0], top_dims3[1], top_dims3[2]), (const int*)v.d, n, (uint32_t*)o.d);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(out);
}

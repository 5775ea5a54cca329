// trav_plain.hip -- traversal of the construction format (entries -> cells -> ref_ids), for grids without a traversal image
// (virtual resolution above 65535 per axis, compressed grids deeper than six levels, "traverse.image" = 0) and for what walks that
// format by definition: the statistics entry point and the reference binary's Hit.id = step count (traverse.cu:80,93).
//
// Replaces the reference's traverse<CellT, Tri> kernel (traverse.cu:27-95) with intersect_ray_box (:14-21) and compute_voxel (:23-25).
// Results per ray are identical to the CPU oracle's (same IEEE operation sequence, contraction off).
#include "trav_common.h"

using namespace hagrid;
using namespace hagrid_impl;
using namespace hagrid_trav;

namespace {

// One instantiation per cell format: the counters are kept in registers whether or not the caller asked for them (a.steps / a.stats null, "traverse.variant" = 1).
template <bool SMALL>
__global__ void __launch_bounds__(256) traverse_kernel(const TraverseArgs a) {
    constexpr bool STATS = true;
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
    unsigned n_cells = 0, n_words = 0, n_refs = 0, n_sent = 0, n_long = 0;

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
                    if (STATS) { n_refs += unsigned(consumed - 1); n_sent++; if (consumed - 1 > 4) n_long += unsigned(consumed - 1); }
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
                if (STATS) { n_refs += unsigned(consumed); if (consumed > 4) n_long += unsigned(consumed); }
            }
            steps += 1 + consumed;
            if (STATS) n_cells++;

            if (hit.t <= texit || ((vx < 0) | (vx >= a.dims_x) | (vy < 0) | (vy >= a.dims_y) | (vz < 0) | (vz >= a.dims_z))) break;
        }
    }

    a.hits[id] = make_float4(__int_as_float((STATS && a.id_is_steps) ? steps : hit.id), hit.t, 0.0f, 0.0f);

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
            atomicAdd(a.stats + 7, (unsigned long long)n_long);
        }
    }
}


// ---- v2: latency-oriented kernel ---------------------------------------------------------------------------------
// A 1M-ray batch is bound by the critical path of its longest rays (hundreds of cell steps, each a chain of
// dependent loads: top entry -> sub entry -> cell -> ref id -> triangle), not by throughput.  v2 shortens that chain:
//   * the NEXT cell's voxel-map walk and cell load are issued before the current cell's triangles are tested
//     (they are independent of the tests; if the ray terminates in this cell the loads are simply dropped);
//   * the top-level entry is kept in a register while the ray stays inside the same top-level cell;
//   * loads are issued unconditionally with clamped addresses so that independent chains overlap instead of
//     being serialised by divergent branches;
//   * one wavefront per workgroup (a finished wave frees its slot at once) and an XCD-aware block -> ray-range map:
//     consecutive ray ranges run on the same XCD, so each of the 8 private L2s caches one band of the scene.
// Same arithmetic per ray as v1 (and the oracle): identical hits.
// a.mode (HAGRID_TRAVERSE_ANY_HIT | HAGRID_TRAVERSE_UVS) is read at run time: the barycentrics are computed with every accepted hit (two multiplies; id and t are
// the same operations either way) and stored where asked for -- one instantiation per cell format and addressing instead of four.
template <bool SMALL, int BLOCK, bool NARROW>
__global__ void __launch_bounds__(BLOCK, 8) traverse_kernel_v2(const TraverseArgs a) {
    const bool ANY = (a.mode & HAGRID_TRAVERSE_ANY_HIT) != 0, UVS = (a.mode & HAGRID_TRAVERSE_UVS) != 0;
    const int* perm = (a.perm && (!a.perm_flag || __builtin_amdgcn_readfirstlane(*a.perm_flag))) ? a.perm : nullptr;
    const int w = (BLOCK == 64 && !perm) ? tile_packet_row_len(a) : 0;
    const int b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk_log2) : xcd_split(blockIdx.x, gridDim.x);
    const int slot = w ? tile_packet_slot(a, w, b, threadIdx.x) : b * BLOCK + threadIdx.x;
    if (slot >= a.num_rays) return;
    const int id = perm ? perm[slot] : slot;

    const float4 r0 = nt_load4(a.rays + 2 * size_t(id)), r1 = nt_load4(a.rays + 2 * size_t(id) + 1);
    const vec3 org(r0.x, r0.y, r0.z), dir(r1.x, r1.y, r1.z);
    const float tmin = r0.w, tmax = r1.w;
    const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 csize(a.cs_x, a.cs_y, a.cs_z), ginv(a.inv_x, a.inv_y, a.inv_z);
    const bool px = dir.x >= 0.0f, py = dir.y >= 0.0f, pz = dir.z >= 0.0f;

    const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
    const vec3 t0 = min(ta, tb), t1 = max(ta, tb);
    const float tstart = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), tmin);
    const float tend = detail::fmin2(detail::fmin2(t1.x, detail::fmin2(t1.y, t1.z)), tmax);

    Hit hit(-1, tmax, 0.0f, 0.0f);

    if (!(tstart > tend)) {
        const vec3 fv = (tstart * dir + org - gmin) * ginv;
        int vx = min(max(int(fv.x), 0), a.dims_x - 1);
        int vy = min(max(int(fv.y), 0), a.dims_y - 1);
        int vz = min(max(int(fv.z), 0), a.dims_z - 1);

        auto walk = [&](uint32_t w, int x, int y, int z) -> uint32_t {   // sub-levels of the voxel map
            int depth = 0;
            while (w & 3u) {
                const int k = int(w & 3u);
                depth += k;
                const int s = a.shift - depth, m = (1 << k) - 1;
                const uint32_t e = (w >> 2) + ((x >> s) & m) + ((((y >> s) & m) + (((z >> s) & m) << k)) << k);
                w = NARROW ? gather32<uint32_t>(a.entries, e << 2) : a.entries[e];
            }
            return w;
        };

        auto top_index = [&](int x, int y, int z) -> int {
            if (NARROW) return int(uint32_t(x >> a.shift) + __umul24(uint32_t(a.top_x), uint32_t(y >> a.shift)) + __umul24(uint32_t(a.top_xy), uint32_t(z >> a.shift)));
            return (x >> a.shift) + a.top_x * ((y >> a.shift) + a.top_y * (z >> a.shift));
        };
        auto entry = [&](int i) -> uint32_t { return NARROW ? gather32<uint32_t>(a.entries, uint32_t(i) << 2) : a.entries[i]; };
        auto ref_at = [&](int i) -> int { return NARROW ? gather32<int>(a.refs, uint32_t(i) << 2) : a.refs[i]; };
        auto cell_at = [&](uint32_t i) -> CellBox {
            if (!NARROW) return load_cell_box<SMALL>(a.cells, i);
            CellBox c;
            if (SMALL) {
                const uint4 w = gather32<uint4>(a.cells, i << 4);
                c.lx = int(w.x & 0xffffu); c.ly = int(w.x >> 16); c.lz = int(w.y & 0xffffu);
                c.hx = int(w.y >> 16); c.hy = int(w.z & 0xffffu); c.hz = int(w.z >> 16);
                c.begin = int(w.w); c.end = 0;
            } else {
                const int4 lo = gather32<int4>(a.cells, i << 5), hi = gather32<int4>(a.cells, (i << 5) + 16u);
                c.lx = lo.x; c.ly = lo.y; c.lz = lo.z; c.begin = lo.w;
                c.hx = hi.x; c.hy = hi.y; c.hz = hi.z; c.end = hi.w;
            }
            return c;
        };
        auto tri_at = [&](int ref) -> Tri {
            if (!NARROW) return load_tri(a.tris, ref);
            if (HG_SOLO && __ballot(ref != __builtin_amdgcn_readfirstlane(ref)) == 0ull) return load_tri_scalar(a.tris, ref);
            // ref * 48 as two full-rate instructions (the compiler turns the shift-add back into a quarter-rate 32-bit multiply)
            uint32_t r3, o;
            asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r3) : "v"(ref));
            asm("v_lshlrev_b32 %0, 4, %1" : "=v"(o) : "v"(r3));
            const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.tris) + o);
            const float4 p0 = p[0], p1 = p[1], p2 = p[2];
            return Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w);
        };

        int top_idx = top_index(vx, vy, vz);
        uint32_t topw = entry(top_idx);
        CellBox c = cell_at(walk(topw, vx, vy, vz) >> 2);

        for (;;) {
            const int cx = px ? c.hx : c.lx, cy = py ? c.hy : c.ly, cz = pz ? c.hz : c.lz;
            const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
            const float texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
            const vec3 ev = (texit * dir + org - gmin) * ginv;
            const int nx = texit == tcell.x ? cx + (px ? 0 : -1) : int(ev.x);
            const int ny = texit == tcell.y ? cy + (py ? 0 : -1) : int(ev.y);
            const int nz = texit == tcell.z ? cz + (pz ? 0 : -1) : int(ev.z);
            vx = px ? max(nx, vx) : min(nx, vx);
            vy = py ? max(ny, vy) : min(ny, vy);
            vz = pz ? max(nz, vz) : min(nz, vz);
            const bool outside = NARROW ? (uint32_t(vx) >= uint32_t(a.dims_x)) | (uint32_t(vy) >= uint32_t(a.dims_y)) | (uint32_t(vz) >= uint32_t(a.dims_z))
                                        : (vx < 0) | (vx >= a.dims_x) | (vy < 0) | (vy >= a.dims_y) | (vz < 0) | (vz >= a.dims_z);

            // first reference of this cell and the next cell's top entry: two independent loads in flight
            const int begin = c.begin;
            const bool nonempty = SMALL ? begin >= 0 : begin < c.end;
            int cur = nonempty ? begin : 0;
            int ref = ref_at(cur);
            cur++;
            if (!nonempty) ref = -1;
            const int ntop = outside ? top_idx : top_index(vx, vy, vz);
            if (ntop != top_idx) { topw = entry(ntop); top_idx = ntop; }
            // next cell: walk + load, overlapping the triangle tests below
            const CellBox nc = cell_at(walk(topw, vx, vy, vz) >> 2);

            while (ref >= 0) {
                const int next = SMALL ? ref_at(cur) : (cur < c.end ? ref_at(cur) : -1);
                cur++;
                const bool got = intersect_prim_ray_uvs(tri_at(ref), Ray(org, tmin, dir, hit.t), ref, hit);
                ref = (ANY && got) ? -1 : next;
            }
            if ((ANY && hit.id >= 0) || hit.t <= texit || outside) break;
            c = nc;
        }
    }
    nt_store4(a.hits + id, __int_as_float(hit.id), hit.t, UVS ? hit.u : 0.0f, UVS ? hit.v : 0.0f);
}


} // namespace

void hagrid_trav::launch_v2(hipStream_t st, int blocks, bool small, bool narrow, unsigned mode, const TraverseArgs& a0) {
    TraverseArgs a = a0;
    a.mode = mode & 3u;
    if (small) { if (narrow) traverse_kernel_v2<true, 64, true><<<blocks, 64, 0, st>>>(a); else traverse_kernel_v2<true, 64, false><<<blocks, 64, 0, st>>>(a); }
    else       { if (narrow) traverse_kernel_v2<false, 64, true><<<blocks, 64, 0, st>>>(a); else traverse_kernel_v2<false, 64, false><<<blocks, 64, 0, st>>>(a); }
}


void hagrid_trav::launch_plain(hipStream_t st, int num_rays, bool small, const TraverseArgs& a) {
    const int blocks = grid_blocks(num_rays, 256);
    if (small) traverse_kernel<true><<<blocks, 256, 0, st>>>(a);
    else       traverse_kernel<false><<<blocks, 256, 0, st>>>(a);
}

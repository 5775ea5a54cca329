// trav_kernels.h -- the kernels that traverse the traversal image (16-byte slim records in three layouts, trav_image.hip): traverse_kernel_img (any-hit /
// barycentric variants, and the nearest hit without the tail mode) and traverse_kernel_tail (nearest hit: the default of every BASELINE configuration).  Templates, instantiated by
// traverse.hip (the product) and, with TIMES = true, by kat/kat.hip (wavefront timelines for tools/dev_wave_timeline.py).
#pragma once

#include "trav_common.h"

namespace hagrid_trav {

// Every gather is base (scalar registers) + unsigned 32-bit byte offset: the host launches these kernels when image, triangles and references are below 4 GB
// and the top-level resolution fits 23 bits per axis (traverse.hip; larger grids are traversed in the construction format).
// LAYOUT: 0 uniform (record of a voxel by arithmetic; bounds as offsets from the voxel), 1 table (blocks per top-level cell through the table; bounds from the
// top-level cell's origin; wide records), 2 general (a record per voxel-map entry: trav_common.h GenWalk; links, wide records) -- trav_image.hip
// SLIM: bits per packed reference id (20 or 26)
// TIMES: diagnostic instantiation that records the wall clock at the start and the end of every wavefront (tools/dev_wave_timeline.py)
// UVS: barycentrics stored with the hit (HAGRID_TRAVERSE_UVS: two more registers); the any-hit rule (HAGRID_TRAVERSE_ANY_HIT) is a uniform flag of the call, a.mode
template <bool UVS, int SLIM, int LAYOUT, bool TIMES = false>
__global__ void __launch_bounds__(64, 8) traverse_kernel_img(const TraverseArgs a) {
    const bool ANY = (a.mode & HAGRID_TRAVERSE_ANY_HIT) != 0;
    constexpr bool UNIFORM = LAYOUT == 0, TABLE = LAYOUT == 1, GENERAL = LAYOUT == 2;
    constexpr int NONE = (1 << SLIM) - 1;          // the id field of an unused list slot
    constexpr int NI = 80 / SLIM, LAST = 48 + (NI - 1) * SLIM;
    struct Stamp {
        unsigned long long* p;
        __device__ Stamp(unsigned long long* q) : p(q) { if (TIMES && threadIdx.x == 0) p[0] = wall_clock64(); }
        __device__ ~Stamp() { if (TIMES) { const unsigned long long t = wall_clock64(); atomicMax(p + 1, t); } }   // the last lane to leave
    } stamp(TIMES ? a.wave_times + 2 * size_t(blockIdx.x) : nullptr);
    const int* perm = (a.perm && (!a.perm_flag || __builtin_amdgcn_readfirstlane(*a.perm_flag))) ? a.perm : nullptr;
    const int w = !perm ? tile_packet_row_len(a) : 0;
    const int b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk_log2) : xcd_split(blockIdx.x, gridDim.x);
    const int slot = w ? tile_packet_slot(a, w, (TIMES && a.tile_order) ? a.tile_order[b] : b, threadIdx.x) : b * 64 + threadIdx.x;
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

        auto top_index = [&](int x, int y, int z) -> int {
            return int(uint32_t(x >> a.shift) + __umul24(uint32_t(a.top_x), uint32_t(y >> a.shift)) + __umul24(uint32_t(a.top_xy), uint32_t(z >> a.shift)));
        };
        auto table_at = [&](int t) -> uint2 { return gather32<uint2>(a.img_table, uint32_t(t) << 3); };
        GenWalk<SLIM> gw;
        gw.restart(a);
        // record of a voxel: one address computation off the scalar base
        auto record = [&](uint2 tab, int x, int y, int z, uint32_t moved = 0u) -> uint4 {
            if (UNIFORM) {
                const int d = a.shift, m = (1 << d) - 1;
                const uint32_t idx = uint32_t(x & m) + (uint32_t((y & m) + ((z & m) << d)) << d);
                return *reinterpret_cast<const uint4*>(a.img_blocks + (((uint32_t(top_index(x, y, z)) << (3 * d)) + idx) << 4));
            }
            if (GENERAL) return gw.lookup(a, x, y, z, moved);          // from the block of the last look-up, or from the top level (a link is resolved behind the tests)
            const int d = int(tab.y & 3u), s = a.shift - d, m = (1 << d) - 1;          // table layout: block offset in records, depth of the block
            const uint32_t idx = uint32_t((x >> s) & m) + (uint32_t(((y >> s) & m) + (((z >> s) & m) << d)) << d);
            return *reinterpret_cast<const uint4*>(a.img_blocks + ((tab.x + idx) << 4));
        };
        auto tri_at = [&](int ref) -> Tri {
            uint32_t r3, o;
            asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r3) : "v"(ref));
            asm("v_lshlrev_b32 %0, 4, %1" : "=v"(o) : "v"(r3));
            const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.tris) + o);
            const float4 p0 = p[0], p1 = p[1], p2 = p[2];
            return Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w);
        };
        auto tri_for = [&](int ref) -> Tri {
            if (HG_SOLO) {
                const int r0 = __builtin_amdgcn_readfirstlane(ref);
                const unsigned long long others = __ballot(ref != r0);
                if (others == 0ull) return load_tri_scalar(a.tris, r0);
            }
            return tri_at(ref);
        };
        // which byte of the bounds is the exit plane, and the direction the offset counts in
        const uint32_t ox = px ? 8u : 0u, oy = py ? 24u : 16u, oz = pz ? 8u : 0u;
        const int sgx = px ? 1 : -1, sgy = py ? 1 : -1, sgz = pz ? 1 : -1;
        const int bx = px ? 0 : -1, by = py ? 0 : -1, bz = pz ? 0 : -1;               // the voxel just past it
        const int lim_x = px ? 0x7fffffff : int(0x80000000), lim_y = py ? 0x7fffffff : int(0x80000000), lim_z = pz ? 0x7fffffff : int(0x80000000);
        int top_idx = TABLE ? top_index(vx, vy, vz) : 0;
        uint2 tab = TABLE ? table_at(top_idx) : make_uint2(0u, 0u);
        uint4 ca = record(tab, vx, vy, vz);
        if (GENERAL) gw.descend(a, ca, vx, vy, vz);

        for (;;) {
            int cx, cy, cz;
            bool wide_cell = false;
            uint32_t wide_begin = 0u;
            if (UNIFORM) {     // byte offsets from the voxel the record belongs to
                // voxel +- offset as ONE multiply-add with the ray's sign (the compiler expands a plain multiply by +-1 into negate + select)
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cx) : "v"(sgx), "v"(__builtin_amdgcn_ubfe(ca.x, ox, 8u)), "v"(vx));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cy) : "v"(sgy), "v"(__builtin_amdgcn_ubfe(ca.x, oy, 8u)), "v"(vy));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cz) : "v"(sgz), "v"(__builtin_amdgcn_ubfe(ca.y, oz, 8u)), "v"(vz));
            } else {
                if (GENERAL) {     // byte offsets from the origin of the record's region
                    const int org_mask = int(~0u << gw.region_shift());
                    cx = (vx & org_mask) + sgx * int(__builtin_amdgcn_ubfe(ca.x, ox, 8u));
                    cy = (vy & org_mask) + sgy * int(__builtin_amdgcn_ubfe(ca.x, oy, 8u));
                    cz = (vz & org_mask) + sgz * int(__builtin_amdgcn_ubfe(ca.y, oz, 8u));
                } else {           // table layout: biased byte offsets from the origin of the top-level cell
                    const int org_mask = ~((1 << a.shift) - 1);
                    cx = (vx & org_mask) + int(__builtin_amdgcn_ubfe(ca.x, ox, 8u)) - 128;
                    cy = (vy & org_mask) + int(__builtin_amdgcn_ubfe(ca.x, oy, 8u)) - 128;
                    cz = (vz & org_mask) + int(__builtin_amdgcn_ubfe(ca.y, oz, 8u)) - 128;
                }
                wide_cell = GenWalk<SLIM>::is_wide(ca);          // a cell the bytes cannot hold: absolute bounds in its wide record
                if (wide_cell) {
                    const uint4 wr = GenWalk<SLIM>::wide_at(a, ca);
                    cx = int(__builtin_amdgcn_ubfe(wr.x, px ? 16u : 0u, 16u)); cy = int(__builtin_amdgcn_ubfe(wr.y, py ? 16u : 0u, 16u)); cz = int(__builtin_amdgcn_ubfe(wr.z, pz ? 16u : 0u, 16u));
                    wide_begin = wr.w;
                }
            }
            const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
            const float texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
            const vec3 ev = (texit * dir + org - gmin) * ginv;
            const int nx = texit == tcell.x ? cx + bx : int(ev.x);
            const int ny = texit == tcell.y ? cy + by : int(ev.y);
            const int nz = texit == tcell.z ? cz + bz : int(ev.z);
            // never backwards: max with the current voxel along a positive direction, min along a negative one -- the median of
            // (new, current, +-infinity), one instruction per axis
            const int pvx = vx, pvy = vy, pvz = vz;
            vx = med3_i32(nx, vx, lim_x); vy = med3_i32(ny, vy, lim_y); vz = med3_i32(nz, vz, lim_z);
            const bool outside = (uint32_t(vx) >= uint32_t(a.dims_x)) | (uint32_t(vy) >= uint32_t(a.dims_y)) | (uint32_t(vz) >= uint32_t(a.dims_z));

            // next cell: table entry (only when the top-level cell changes) -> record, in flight during the tests below
            if (TABLE) {
                const int ntop = outside ? top_idx : top_index(vx, vy, vz);
                if (ntop != top_idx) { tab = table_at(ntop); top_idx = ntop; }
            }
            uint4 na = make_uint4(0u, 0u, 0u, 0u);
            if (UNIFORM) { const int sx = outside ? 0 : vx, sy = outside ? 0 : vy, sz = outside ? 0 : vz; na = record(tab, sx, sy, sz); }
            else if (!outside) na = record(tab, vx, vy, vz, uint32_t((vx ^ pvx) | (vy ^ pvy) | (vz ^ pvz)));

            // Lists: inline ids (up to four, unused fields NONE) are consumed front to back; a list given by index (more ids than a record holds, wide cells)
            // fetches the id of the next test one test ahead, as v2 does.
            auto ref_at = [&](uint32_t i) -> int { return gather32<int>(a.refs, i << 2); };
            auto field = [&](int pos, int n) -> uint32_t {              // pos, n are constants after inlining
                const uint32_t wd[4] = {ca.x, ca.y, ca.z, ca.w};
                const int i = pos >> 5, o = pos & 31;
                uint32_t v = wd[i] >> o;
                if (o + n > 32) v |= wd[i + 1] << (32 - o);
                return n == 32 ? v : (v & ((1u << n) - 1u));
            };
            const bool by_index = field(LAST, SLIM) == uint32_t(NONE - 1) || wide_cell;
            int ref = int(field(48, SLIM));
            uint32_t q1 = NI > 1 ? field(48 + SLIM, SLIM) : uint32_t(NONE), q2 = NI > 2 ? field(48 + 2 * SLIM, SLIM) : uint32_t(NONE),
                     q3 = NI > 3 ? field(48 + 3 * SLIM, SLIM) : uint32_t(NONE);
            const uint32_t li_begin = wide_cell ? wide_begin : field(48, 32), li_count = field(80, 20);
            if (UNIFORM && __ballot(by_index) == 0ull) {
                // shallow grids: lists of more than four ids are rare (1.5 % of the visited cells of the 1M-triangle soup), so a
                // wavefront normally holds inline lists only and runs this loop: no index bookkeeping, no masked branches
#pragma unroll 1
                while (ref != NONE) {
                    const bool got = UVS ? intersect_prim_ray_uvs(tri_for(ref), Ray(org, tmin, dir, hit.t), ref, hit)
                                         : intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit.t), ref, hit);
                    ref = (ANY && got) ? NONE : int(q1);
                    q1 = q2; q2 = q3; q3 = uint32_t(NONE);
                }
            } else {
                // One loop for both list forms, so a wavefront whose lanes hold both pays the longest list, not the sum of the two longest.
                if (by_index) {                                             // by index: q1 = index of the next id, q2 = end of the list
                    q1 = li_begin; q2 = li_begin + li_count;
                    ref = NONE;
                    if (q1 < q2) ref = ref_at(q1);
                    q1++;
                }
#pragma unroll 1
                while (ref != NONE) {
                    int next;
                    if (UNIFORM) {
                        // shallow grids, long lists are rare: the fewest instructions for the inline form
                        if (by_index) { next = q1 < q2 ? ref_at(q1) : NONE; q1++; }
                        else { next = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE); }
                    }
                    int pre = NONE;
                    if (!UNIFORM && by_index && q1 < q2) pre = ref_at(q1);      // in flight during the test; nothing reads it before
                    const bool got = UVS ? intersect_prim_ray_uvs(tri_for(ref), Ray(org, tmin, dir, hit.t), ref, hit)
                                         : intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit.t), ref, hit);
                    if (!UNIFORM) {
                        if (by_index) { next = pre; q1++; }
                        else { next = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE); }
                    }
                    ref = (ANY && got) ? NONE : next;
                }
            }
            if ((ANY && hit.id >= 0) || hit.t <= texit || outside) break;
            ca = na;
            if (GENERAL) gw.descend(a, ca, vx, vy, vz);
        }
    }
    nt_store4(a.hits + id, __int_as_float(hit.id), hit.t, UVS ? hit.u : 0.0f, UVS ? hit.v : 0.0f);
}


// ---- image kernel with a tail mode ---------------------------------------------------------------------------------------
// The 1M-ray launch ends when its longest rays end (half of it is the drain of wavefronts that hold a handful of live
// rays, each lock-step iteration a serial chain of dependent instructions and one dependent gather), and a wavefront of few live rays
// still pays a triangle round per id of its longest list.  Rays only finish, so the number of live rays of a wavefront only falls:
// once it is at most 16 the wavefront COMPACTS -- live ray r moves to lanes 4r .. 4r + 3 (one LDS rendezvous, one ds_bpermute per
// register, once per wavefront) -- and from then on a cell step tests the up to four inline ids of a list in ONE round, lane s of the
// group taking id s, and the cell step itself is split over the four lanes (one axis each, see phase 2 below).  The reference's sequential rule
// (every test sees the tmax the accepted tests before it left, prims.h:266-295) is kept exactly: a lane computes everything that
// does not depend on tmax -- the barycentric test, t >= |det| * tmin, t and |det| -- and the group then replays the acceptance
// `|det| * tmax > t` in list order on quad broadcasts (DPP), so hit ids and t stay bit-identical.  Table-free layout with slim
// records, nearest hit, narrow addressing; everything else runs traverse_kernel_img.
struct TriCand { float t, abs_det; bool ok; };
__device__ __forceinline__ TriCand tri_candidate(const Tri& tri, const vec3& org, const vec3& dir, float tmin) {   // prims.h:266-283, up to the comparison with tmax
    const vec3 n = tri.normal();
    const vec3 c = tri.v0 - org;
    const vec3 r = cross(dir, c);
    const float det = dot(n, dir);
    const float abs_det = detail::fabs1(det);
    const float u = prodsign(dot(r, tri.e2), det);
    const float v = prodsign(dot(r, tri.e1), det);
    const float w = abs_det - u - v;
    const float eps = 1e-9f;
    TriCand cd; cd.t = 0.0f; cd.abs_det = abs_det; cd.ok = false;
    if (u >= -eps && v >= -eps && w >= -eps) {
        const float t = prodsign(dot(n, c), det);
        if (t >= abs_det * tmin) { cd.t = t; cd.ok = true; }
    }
    return cd;
}
template <int K> __device__ __forceinline__ int quad_bcast_i(int x) { return __builtin_amdgcn_update_dpp(x, x, K * 0x55, 0xf, 0xf, true); }   // (every lane is written: no `old` value to set up)
template <int K> __device__ __forceinline__ float quad_bcast_f(float x) { return __int_as_float(quad_bcast_i<K>(__float_as_int(x))); }
// lane s of a quad reads lane (CTRL >> 2s) & 3 of the same quad: 9 = [1,2,0,0], 82 = [2,0,1,1]
template <int CTRL> __device__ __forceinline__ int quad_perm_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float quad_perm_f(float x) { return __int_as_float(quad_perm_i<CTRL>(__float_as_int(x))); }

constexpr int kTailRays = 16;        // live rays at which a wavefront compacts (64 lanes / 4 lanes per ray)

// Sensitivity experiments (tools/dev_ab.sh: HAGRID_HIPCC_EXTRA=-DHG_EXTRA_NOPS=20 builds ab/libB.so): this many v_nop per triangle round of the
// tail kernel -- what the launch pays for VALU issue slots.  Not defined in the product.
#ifdef HG_EXTRA_NOPS
#define HG_NOPS() asm volatile(".rept %0\n v_nop\n .endr" :: "n"(HG_EXTRA_NOPS))
#else
#define HG_NOPS() do { } while (0)
#endif

// TIMES: diagnostic instantiation that records the wall clock at the start and the end of every wavefront (tools/dev_wave_timeline.py)
// UNIFORM = false: the table layout of slim records (grids of at most three levels whose top-level cells differ in depth): the block of a
// top-level cell is found through its table entry, kept while the ray stays inside the cell; bounds count from that cell's origin.
// DUAL: phase 1 tests the ids of an inline list two per round trip (see test_list)
// COST: the wavefront counts its iterations and leaves them at its tile (the tile order of traverse.hip); the table layout runs that bookkeeping at seven resident
// wavefronts per SIMD (HG_TABLE_WAVES below: at eight it spills a dozen registers around its loops), so launches of it that keep no costs run the instantiation without
// MAILBOX: every ray remembers the last four triangles it was tested against (a FIFO of ids per lane in LDS: the kernel has no registers for it) and skips
// a test it has made before.  A triangle is referenced by the several cells it overlaps and a ray that passes it crosses several of them: on the oracle's
// traces a quarter of the tests of a ray repeat one of its last four (15 % one of its last two), primary and incoherent batches alike.  A repeated test
// is a no-op by construction -- rejected before the tmax comparison: rejected again (the ray has not changed); rejected by it: tmax has only fallen since;
// accepted: it either writes the same id and t again or is rejected -- so the hits stay bit-identical.  What it saves is the three lane accesses of the
// triangle in the vector L1 and its L2 / HBM sector: the resources the incoherent and the beyond-cache batches are bound by (profiles/r4a).  It costs an LDS
// round trip in front of every triangle round.
#ifndef HG_TABLE_WAVES
// resident wavefronts per SIMD of the table-layout instantiations WITH cost bookkeeping (launches that learn or follow a tile order: a few rounds, latency-bound).  At
// eight they keep 64 registers and spill 9 - 11 values AROUND their loops (none inside: the compiler's listing, tools/kernel_resources.py); at seven (70 registers) none.
// Same box, round 6 (gpurun_out/r6g): config 3's grid 1024^2 in a learned order 0.1618 -> 0.1582 ms (-2.2 %), with wide records 0.166 -> 0.164.  The instantiations
// WITHOUT costs (launches beyond 25 rounds: throughput-bound) stay at eight: the soup at --snd-density 5, 4096^2, 1.327 -> 1.372 ms (+3.4 %) at seven, although the
// wide-record one reloads a spilled value once per iteration of its four-lanes-per-ray phase.
#define HG_TABLE_WAVES 7
#endif
#ifndef HG_GENERAL_WAVES
#define HG_GENERAL_WAVES 7          // resident wavefronts per SIMD of the general-layout instantiations (8: a dozen spilled registers around the loops)
#endif
template <int SLIM, bool TIMES = false, bool UNIFORM = true, bool DUAL = false, bool COST = UNIFORM, bool MAILBOX = false, bool GENERAL = false, bool WIDE = GENERAL>
__global__ void __launch_bounds__(64, MAILBOX ? 7 : (GENERAL ? HG_GENERAL_WAVES : ((!UNIFORM && COST) ? HG_TABLE_WAVES : 8))) traverse_kernel_tail(const TraverseArgs a) {
    // WIDE: the image holds wide records (always possible in the general layout; a table-layout image without any runs the instantiation without the checks)
    static_assert(!WIDE || !UNIFORM, "the uniform layout has no wide records");
    static_assert(!GENERAL || !UNIFORM, "a layout is uniform, table (blocks per top-level cell) or general (a record per voxel-map entry)");
    constexpr bool TABLE = !UNIFORM && !GENERAL;
    constexpr int NONE = (1 << SLIM) - 1, NI = 80 / SLIM, LAST = 48 + (NI - 1) * SLIM;
    static_assert(!(MAILBOX && DUAL), "the mailbox is built into the one-id-per-round loops");
    __shared__ int lanes_of[64];
    __shared__ int4 mailbox[MAILBOX ? 64 : 1];            // MAILBOX: the last four ids tested for the ray in this lane (phase 2: for the ray of the group, in its first lane's slot)
    __shared__ int cost_bonus;                            // iterations counted twice at the end (blocks without a one-ray-per-lane phase)
    __shared__ int* cost_at;                              // where this wavefront leaves its cost (its tile's word of a.tile_cost), nullptr: nowhere
    __shared__ float4 tri_lds[DUAL ? 3 * 64 : 1];         // DUAL: a lane's second triangle of a round, written by the load itself (LDS-DMA)
    const int lane = threadIdx.x;
    if (MAILBOX) mailbox[lane] = make_int4(-1, -1, -1, -1);          // (ids are >= 0; a wavefront is its own workgroup: LDS operations of a wavefront execute in order)
    struct Stamp {
        unsigned long long* p;
        __device__ Stamp(unsigned long long* q) : p(q) { if (TIMES && threadIdx.x == 0) p[0] = wall_clock64(); }
        __device__ ~Stamp() { if (TIMES) { const unsigned long long t = wall_clock64(); atomicMax(p + 1, t); } }
    } stamp(TIMES ? a.wave_times + 2 * size_t(blockIdx.x) : nullptr);
    const int* perm = (a.perm && (!a.perm_flag || __builtin_amdgcn_readfirstlane(*a.perm_flag))) ? a.perm : nullptr;
    const int w = !perm ? tile_packet_row_len(a) : 0;
    // The last tiles in dispatch order are traversed with four lanes per ray from their first cell on (phase 2 below): a tile is then
    // four blocks of 16 rays (its 4 x 4 pixel quadrants).  They are the wavefronts that start when the machine begins to drain, where
    // wavefront slots are free and what counts is how long the longest ray of a wavefront takes.
    // With a.quad_head > 0 those blocks are dispatched FIRST (the grid rotated by 4 * quad_head blocks) and the tile order is rotated the same way (its last quad_head
    // positions hold the longest tiles, ray_order.hip): the tiles the launch is as long as start at once and take half as many iterations per ray.
    const int bid = a.quad_head ? (int(blockIdx.x) < 4 * a.quad_head ? int(blockIdx.x) + (int(gridDim.x) - 4 * a.quad_head) : int(blockIdx.x) - 4 * a.quad_head) : int(blockIdx.x);
    const bool quad_start = bid >= a.quad_first_block;
    const int group = lane >> 2, sub = lane & 3;
    int b, lane_in_tile = lane;
    if (quad_start) {
        const int q = bid - a.quad_first_block, nq = int(gridDim.x) - a.quad_first_block;
        const int lq = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(q, nq, a.xcd_chunk_log2 + 2) : xcd_split(q, nq);
        b = a.quad_first_block + (lq >> 2);
        lane_in_tile = ((((lq >> 1) & 1) << 2) + (group >> 2)) * 8 + ((lq & 1) << 2) + (group & 3);
    } else {
        const int nb = min(int(gridDim.x), a.quad_first_block);
        b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(bid, nb, a.xcd_chunk_log2) : xcd_split(bid, nb);
    }
    // Which tile: position b of the dispatch order, or -- when the previous launch over this ray buffer left its costs -- the tile that
    // order names (longest first, traverse.hip "tile order").  The wavefront leaves its own cost (iterations = cells of its longest ray) behind.
    int iters = 0;                                        // iterations this wavefront ran, those of phase 1 counted twice: its cost
    int slot;
    {
        int tile = b;
        if (COST && w && a.tile_order) {
            tile = a.tile_order[b];
            // (an order learned on other rays than the buffer holds now -- refilled, another buffer at a recycled address, the camera moved -- is
            // reported by the first wavefront of the launch and not used again: a stale order is slower than none)
            if (a.order_samples && blockIdx.x == 0) order_check(a, lane);
        }
        slot = w ? tile_packet_slot(a, w, tile, lane_in_tile) : b * 64 + lane_in_tile;
        // (a block that starts with four lanes per ray has no one-ray-per-lane phase to count twice: the first dozen of its iterations are doubled
        // instead -- about what that phase lasts)
        if (COST && lane == 0) { cost_at = (a.tile_cost && w) ? a.tile_cost + tile : nullptr; cost_bonus = quad_start ? 12 : 0; }      // (in LDS: the kernel has no register to spare for the whole traversal)
    }
    const bool valid = slot < a.num_rays;
    int id = valid ? (perm ? perm[slot] : slot) : 0;
    bool pending = valid;                                  // this lane still owes its ray's hit to the hit buffer

    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = make_float4(1.0f, 1.0f, 1.0f, -1.0f);
    if (valid) { r0 = nt_load4(a.rays + 2 * size_t(id)); r1 = nt_load4(a.rays + 2 * size_t(id) + 1); }
    vec3 org(r0.x, r0.y, r0.z), dir(r1.x, r1.y, r1.z);
    float tmin = r0.w;
    const float tmax = r1.w;
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 csize(a.cs_x, a.cs_y, a.cs_z), ginv(a.inv_x, a.inv_y, a.inv_z);
    float hit_t = tmax;
    int hit_id = -1;
    int vx = 0, vy = 0, vz = 0;
    bool alive = false;
    {
        const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
        const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
        const vec3 t0 = min(ta, tb), t1 = max(ta, tb);
        const float tstart = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), tmin);
        const float tend = detail::fmin2(detail::fmin2(t1.x, detail::fmin2(t1.y, t1.z)), tmax);
        if (valid && !(tstart > tend)) {
            const vec3 fv = (tstart * dir + org - gmin) * ginv;
            vx = min(max(int(fv.x), 0), a.dims_x - 1);
            vy = min(max(int(fv.y), 0), a.dims_y - 1);
            vz = min(max(int(fv.z), 0), a.dims_z - 1);
            alive = true;
        }
    }
    GenWalk<SLIM> gw;                                      // general layout: the innermost block the ray's last look-up ended in
    gw.restart(a);
    uint32_t tab_off = 0u, tab_d = 0u;                     // table layout: block offset (records) and depth of the top-level cell the ray is in
    int top_idx = -1;
    auto load_record = [&](int x, int y, int z) -> uint4 {          // uniform layout: the record of a voxel is arithmetic on the voxel; table layout: through the table entry of its top-level cell
        const uint32_t top = uint32_t(x >> a.shift) + __umul24(uint32_t(a.top_x), uint32_t(y >> a.shift)) + __umul24(uint32_t(a.top_xy), uint32_t(z >> a.shift));
        if (UNIFORM) {
            const int d = a.shift, m = (1 << d) - 1;
            const uint32_t idx = uint32_t(x & m) + (uint32_t((y & m) + ((z & m) << d)) << d);
            return *reinterpret_cast<const uint4*>(a.img_blocks + (((top << (3 * d)) + idx) << 4));
        }
        if (int(top) != top_idx) {
            const uint2 t = gather32<uint2>(a.img_table, top << 3);
            tab_off = t.x; tab_d = t.y & 3u; top_idx = int(top);
        }
        const int d = int(tab_d), sh = a.shift - d, m = (1 << d) - 1;
        const uint32_t idx = uint32_t((x >> sh) & m) + (uint32_t(((y >> sh) & m) + (((z >> sh) & m) << d)) << d);
        return *reinterpret_cast<const uint4*>(a.img_blocks + ((tab_off + idx) << 4));
    };
    auto tri_ptr = [&](int ref) -> const float4* {
        uint32_t r3, o;
        asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r3) : "v"(ref));
        asm("v_lshlrev_b32 %0, 4, %1" : "=v"(o) : "v"(r3));
        return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.tris) + o);
    };
    auto tri_vec = [&](int ref) -> Tri {
        const float4* p = tri_ptr(ref);
        const float4 p0 = p[0], p1 = p[1], p2 = p[2];
        return Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w);
    };
    auto tri_for = [&](int ref) -> Tri {                   // every live lane the same triangle: through the scalar cache
        const int f = __builtin_amdgcn_readfirstlane(ref);
        if (HG_SOLO && __ballot(ref != f) == 0ull) return load_tri_scalar(a.tris, f);
        return tri_vec(ref);
    };
    auto field = [&](const uint4& rec, int pos, int n) -> uint32_t {
        const uint32_t wd[4] = {rec.x, rec.y, rec.z, rec.w};
        const int i = pos >> 5, o = pos & 31;
        uint32_t v = wd[i] >> o;
        if (o + n > 32) v |= wd[i + 1] << (32 - o);
        return n == 32 ? v : (v & ((1u << n) - 1u));
    };
    auto ref_at = [&](uint32_t i) -> int { return gather32<int>(a.refs, i << 2); };

    // One cell step of the ray in this lane (traverse.cu:61-78): exit plane of the cell `rec` describes, next voxel, next record.
    float texit = 0.0f;
    bool outside = false;
    // The form of the current cell's list, found once per step: by index (more ids than a record holds) -- or, in the table and general layouts, a WIDE cell, whose
    // list is by index as well -- in this lane, and in any lane of the wavefront.  Wide cells are looked for only in steps in which some lane is by index:
    // the common step (inline lists everywhere) pays one OR for them.
    bool bi_cur = false;
    unsigned long long bi_any = 0ull;
    auto cell_step = [&](const uint4& rec, const vec3& inv_dir) -> uint4 {
        const bool px = dir.x >= 0.0f, py = dir.y >= 0.0f, pz = dir.z >= 0.0f;
        const uint32_t marker = field(rec, LAST, SLIM);
        bi_cur = !WIDE ? marker == uint32_t(NONE - 1) : (marker | 2u) == uint32_t(NONE - 1);        // NONE - 1: by index, NONE - 3: wide
        bi_any = __ballot(bi_cur);
        int cx, cy, cz;
        if (UNIFORM) {
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cx) : "v"(px ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, px ? 8u : 0u, 8u)), "v"(vx));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cy) : "v"(py ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, py ? 24u : 16u, 8u)), "v"(vy));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cz) : "v"(pz ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.y, pz ? 8u : 0u, 8u)), "v"(vz));
        } else if (TABLE) {     // biased byte offsets from the origin of the top-level cell
            const int org_mask = ~((1 << a.shift) - 1);
            cx = (vx & org_mask) + int(__builtin_amdgcn_ubfe(rec.x, px ? 8u : 0u, 8u)) - 128;
            cy = (vy & org_mask) + int(__builtin_amdgcn_ubfe(rec.x, py ? 24u : 16u, 8u)) - 128;
            cz = (vz & org_mask) + int(__builtin_amdgcn_ubfe(rec.y, pz ? 8u : 0u, 8u)) - 128;
            if (WIDE && bi_any != 0ull && marker == uint32_t(NONE - 3)) {          // a cell the bytes cannot hold: absolute bounds in its wide record (the large cells of empty space)
                const uint4 wr = GenWalk<SLIM>::wide_at(a, rec);
                cx = int(__builtin_amdgcn_ubfe(wr.x, px ? 16u : 0u, 16u)); cy = int(__builtin_amdgcn_ubfe(wr.y, py ? 16u : 0u, 16u)); cz = int(__builtin_amdgcn_ubfe(wr.z, pz ? 16u : 0u, 16u));
            }
        } else {
            // general layout: byte offsets from the origin of the record's region (2^s voxels wide) -- or, for the few cells that reach further, a wide
            // record with absolute 16-bit bounds (one more dependent gather, in steps through the large cells of empty space only)
            const int org_mask = int(~0u << gw.region_shift());
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cx) : "v"(px ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, px ? 8u : 0u, 8u)), "v"(vx & org_mask));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cy) : "v"(py ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, py ? 24u : 16u, 8u)), "v"(vy & org_mask));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cz) : "v"(pz ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.y, pz ? 8u : 0u, 8u)), "v"(vz & org_mask));
            if (WIDE && bi_any != 0ull && marker == uint32_t(NONE - 3)) {
                const uint4 wr = GenWalk<SLIM>::wide_at(a, rec);
                cx = int(__builtin_amdgcn_ubfe(wr.x, px ? 16u : 0u, 16u)); cy = int(__builtin_amdgcn_ubfe(wr.y, py ? 16u : 0u, 16u)); cz = int(__builtin_amdgcn_ubfe(wr.z, pz ? 16u : 0u, 16u));
            }
        }
        const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
        texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
        const vec3 ev = (texit * dir + org - gmin) * ginv;
        const int nx = texit == tcell.x ? cx + (px ? 0 : -1) : int(ev.x);
        const int ny = texit == tcell.y ? cy + (py ? 0 : -1) : int(ev.y);
        const int nz = texit == tcell.z ? cz + (pz ? 0 : -1) : int(ev.z);
        uint32_t moved = 0u;                                       // general layout: the bits in which the voxel changes
        { const int t = med3_i32(nx, vx, px ? 0x7fffffff : int(0x80000000)); if (GENERAL) moved = uint32_t(t ^ vx); vx = t; }
        { const int t = med3_i32(ny, vy, py ? 0x7fffffff : int(0x80000000)); if (GENERAL) moved |= uint32_t(t ^ vy); vy = t; }
        { const int t = med3_i32(nz, vz, pz ? 0x7fffffff : int(0x80000000)); if (GENERAL) moved |= uint32_t(t ^ vz); vz = t; }
        outside = (uint32_t(vx) >= uint32_t(a.dims_x)) | (uint32_t(vy) >= uint32_t(a.dims_y)) | (uint32_t(vz) >= uint32_t(a.dims_z));
        uint4 next = make_uint4(0u, 0u, 0u, 0u);                  // a ray that left the grid requests nothing
        if (!outside) next = GENERAL ? gw.lookup(a, vx, vy, vz, moved) : load_record(vx, vy, vz);       // (general layout: possibly a link, resolved behind the tests)
        return next;
    };
    // The list of the cell `rec` describes, tested front to back by this lane alone (the plain loop of traverse_kernel_img).
    auto test_list = [&](const uint4& rec) {
        const bool by_index = bi_cur;                          // (cell_step of this record has looked)
        int ref = int(field(rec, 48, SLIM));
        uint32_t q1 = NI > 1 ? field(rec, 48 + SLIM, SLIM) : uint32_t(NONE), q2 = NI > 2 ? field(rec, 48 + 2 * SLIM, SLIM) : uint32_t(NONE),
                 q3 = NI > 3 ? field(rec, 48 + 3 * SLIM, SLIM) : uint32_t(NONE);
        if (DUAL && bi_any == 0ull) {
            // Two ids per round trip.  A list is tested front to back and every test waits for its triangle; with 64 registers a lane
            // has room for one triangle, so the SECOND id of a round is requested straight into the lane's slots of `tri_lds`
            // (global_load_lds: no registers) together with the first one's ordinary loads, and read from there when the first test is
            // done.  Same tests, same order, same tmax at every test; a list of up to four ids costs two gather latencies instead of four.
            typedef __attribute__((address_space(1))) const void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll 1
            while (ref != NONE) {
                const int second = int(q1);
                if (second != NONE) {
                    const float4* p = tri_ptr(second);
                    // (the instruction's offset counts in the global AND in the LDS address: the LDS bases are 1024 - 16 and 2048 - 32 bytes)
                    // (and the address arithmetic is done on the LDS pointer: a generic pointer would be checked for null on its way to LDS)
                    typedef __attribute__((address_space(3))) float4 lds_f4;
                    lds_f4* const slots = (lds_f4*)tri_lds;
                    __builtin_amdgcn_global_load_lds((gptr_t)(p), (lptr_t)(slots), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(p), (lptr_t)(slots + 63), 16, 16, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(p), (lptr_t)(slots + 126), 16, 32, 0);
                }
                Hit h(hit_id, hit_t, 0.0f, 0.0f);
                (void)intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit_t), ref, h);
                if (second != NONE) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the requests landed: they were issued before the first triangle's loads)
                    const float4 p0 = tri_lds[lane], p1 = tri_lds[64 + lane], p2 = tri_lds[128 + lane];
                    (void)intersect_prim_ray(Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w), Ray(org, tmin, dir, h.t), second, h);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the slots are read before the next round's request may overwrite them)
                }
                hit_t = h.t; hit_id = h.id;
                ref = int(q2); q1 = q3; q2 = uint32_t(NONE); q3 = uint32_t(NONE);
            }
        } else if (bi_any == 0ull) {
#pragma unroll 1
            while (ref != NONE) {
                bool fresh = true;
                if (MAILBOX) {
                    const int4 m = mailbox[lane];
                    fresh = !(ref == m.x || ref == m.y || ref == m.z || ref == m.w);
                    if (fresh) mailbox[lane] = make_int4(m.y, m.z, m.w, ref);
                }
                if (fresh) {
                    Hit h(hit_id, hit_t, 0.0f, 0.0f);
                    (void)intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit_t), ref, h);
                    hit_t = h.t; hit_id = h.id;
                }
                ref = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE);
                HG_NOPS();
            }
        } else {
            if (by_index) {
                q1 = (WIDE && GenWalk<SLIM>::is_wide(rec)) ? GenWalk<SLIM>::wide_at(a, rec).w : field(rec, 48, 32); q2 = q1 + field(rec, 80, 20);      // (the wide record again: the kernel has no register to carry its list index across the step)
                ref = NONE;
                if (q1 < q2) ref = ref_at(q1);
                q1++;
            }
#pragma unroll 1
            while (ref != NONE) {
                int next;
                if (by_index) { next = q1 < q2 ? ref_at(q1) : NONE; q1++; }
                else { next = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE); }
                bool fresh = true;
                if (MAILBOX) {
                    const int4 m = mailbox[lane];
                    fresh = !(ref == m.x || ref == m.y || ref == m.z || ref == m.w);
                    if (fresh) mailbox[lane] = make_int4(m.y, m.z, m.w, ref);
                }
                if (fresh) {
                    Hit h(hit_id, hit_t, 0.0f, 0.0f);
                    (void)intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit_t), ref, h);
                    hit_t = h.t; hit_id = h.id;
                }
                ref = next;
            }
        }
    };

    uint4 ca = make_uint4(0u, 0u, 0u, 0u);
    if (alive) {
        if (GENERAL) { ca = gw.lookup(a, vx, vy, vz, 0u); gw.descend(a, ca, vx, vy, vz); }
        else ca = load_record(vx, vy, vz);
    }
    unsigned long long live = __ballot(alive);

    if (quad_start) pending = valid && sub == 0;           // (the four lanes of a group hold the same ray: one of them stores its hit)
    else {
    // ---- phase 1: one ray per lane, while the wavefront holds more than kTailRays live rays -------------------------------
    {
        vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
        while (__popcll(live) > kTailRays) {
            if (alive) {
                const uint4 na = cell_step(ca, inv_dir);
                test_list(ca);
                if (hit_t <= texit || outside) alive = false;
                ca = na;
                if (GENERAL && alive) gw.descend(a, ca, vx, vy, vz);          // general layout: a link leads on to the child block (its first gather was in flight during the tests)
            }
            live = __ballot(alive);
            if (COST) iters += 2;              // (an iteration of this phase runs its lists' rounds one after the other: it weighs about two of the other phase's)
        }
    }
    if (live == 0ull) {
        if (pending) nt_store4(a.hits + id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
        if (COST && lane == 0 && cost_at) atomicMax(cost_at, iters + min(iters, cost_bonus));
        return;
    }

    // ---- compaction: finished lanes hand in their hits; live ray r moves to lanes 4r .. 4r + 3 ----------------------------------
    if (pending && !alive) nt_store4(a.hits + id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
    const int nlive = __popcll(live);
    if (alive) lanes_of[__popcll(live & ((1ull << lane) - 1ull))] = lane;
    __syncthreads();
    alive = group < nlive;
    pending = alive && sub == 0;
    const int src4 = lanes_of[alive ? group : 0] << 2;
    auto pull_i = [&](int v) -> int { return __builtin_amdgcn_ds_bpermute(src4, v); };
    auto pull_f = [&](float v) -> float { return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v))); };
    org = vec3(pull_f(org.x), pull_f(org.y), pull_f(org.z));
    dir = vec3(pull_f(dir.x), pull_f(dir.y), pull_f(dir.z));
    tmin = pull_f(tmin); hit_t = pull_f(hit_t); hit_id = pull_i(hit_id); id = pull_i(id);
    vx = pull_i(vx); vy = pull_i(vy); vz = pull_i(vz);
    ca = make_uint4(uint32_t(pull_i(int(ca.x))), uint32_t(pull_i(int(ca.y))), uint32_t(pull_i(int(ca.z))), uint32_t(pull_i(int(ca.w))));
    if (GENERAL) { gw.blk = uint32_t(pull_i(int(gw.blk))); gw.bks = uint32_t(pull_i(int(gw.bks))); }
    if (TABLE) { tab_off = uint32_t(pull_i(int(tab_off))); tab_d = uint32_t(pull_i(int(tab_d))); top_idx = pull_i(top_idx); }
    if (MAILBOX) {                                         // the mailbox moves with its ray: the group's first lane's slot holds it from here on
        const int4 m = mailbox[lanes_of[alive ? group : 0]];
        __syncthreads();                                   // (every slot is read before any is overwritten)
        if (alive && sub == 0) mailbox[lane] = m;
        __syncthreads();
    }
    }

    // ---- phase 2: four lanes per ray ------------------------------------------------------------------------------------------
    {
        const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
        // The cell step is split over the lanes of a group instead of being repeated by them.  Lane s owns axis
        // min(s, 2) (lane 3 doubles z): it computes its axis' exit plane, the exit parameter is the minimum over the group, the lane
        // finds its own coordinate of the next voxel and its share of the record's address, and the shares are added over the group
        // (quad permutes [1,2,0,0] and [2,0,1,1]: every lane sees the other two axes).  The same operations on the same values as
        // cell_step, a third of them per lane: ~45 instead of ~100 VALU instructions per step, address included.
        const int ax = sub < 2 ? sub : 2;
        const float m_dir = ax == 0 ? dir.x : (ax == 1 ? dir.y : dir.z), m_org = ax == 0 ? org.x : (ax == 1 ? org.y : org.z);
        const float m_inv = ax == 0 ? inv_dir.x : (ax == 1 ? inv_dir.y : inv_dir.z);
        const float m_cs = ax == 0 ? a.cs_x : (ax == 1 ? a.cs_y : a.cs_z), m_gmin = ax == 0 ? a.min_x : (ax == 1 ? a.min_y : a.min_z);
        const float m_ginv = ax == 0 ? a.inv_x : (ax == 1 ? a.inv_y : a.inv_z);
        const int m_dims = ax == 0 ? a.dims_x : (ax == 1 ? a.dims_y : a.dims_z);
        const bool m_pos = m_dir >= 0.0f;
        const uint32_t m_bit = (ax == 1 ? 16u : 0u) + (m_pos ? 8u : 0u);                       // where the record holds this axis' bound byte
        const uint32_t m_stride = ax == 0 ? 1u : (ax == 1 ? uint32_t(GENERAL ? a.gen_x : a.top_x) : uint32_t(GENERAL ? a.gen_xy : a.top_xy)), m_lsh = uint32_t(ax * a.shift);
        int m_v = ax == 0 ? vx : (ax == 1 ? vy : vz);
        auto quad_sum = [&](uint32_t x) -> uint32_t { return x + uint32_t(quad_perm_i<9>(int(x))) + uint32_t(quad_perm_i<82>(int(x))); };
        auto quad_or = [&](int o) -> bool { return (o | quad_perm_i<9>(o) | quad_perm_i<82>(o)) != 0; };
        // general layout, this lane's share of a child index: ((v >> s) & (2^k - 1)) << (axis * k)
        auto child_share = [&](uint32_t v, uint32_t k, uint32_t s) -> uint32_t { return ((v >> s) & ((1u << k) - 1u)) << __umul24(uint32_t(ax), k); };
        auto quad_descend = [&](uint4& rec) {          // GenWalk::descend with the voxel spread over the lanes of the group (all of them hold the same record)
            while (GenWalk<SLIM>::is_link(rec)) {
                const uint32_t k = (rec.z >> 16) & 3u, s = GenWalk<SLIM>::link_region(rec, gw.bks);
                gw.blk = GenWalk<SLIM>::word48(rec); gw.bks = k | s << 2;
                rec = GenWalk<SLIM>::rec_at(a, gw.blk + quad_sum(child_share(uint32_t(m_v), k, s)));
            }
        };
        auto quad_step = [&](const uint4& rec, bool maybe_wide /* some lane of the wavefront holds a list by index: wave-uniform */) -> uint4 {
            int c;
            const uint32_t bound = __builtin_amdgcn_ubfe(ax == 2 ? rec.y : rec.x, m_bit, 8u);
            if (UNIFORM) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c) : "v"(m_pos ? 1 : -1), "v"(bound), "v"(m_v));
            else if (TABLE) {             // table layout: bounds count from the top-level cell's origin; a wide cell has absolute bounds in its wide record
                c = (m_v & ~((1 << a.shift) - 1)) + int(bound) - 128;
                if (WIDE && maybe_wide && GenWalk<SLIM>::is_wide(rec)) {
                    const uint4 wr = GenWalk<SLIM>::wide_at(a, rec);
                    c = int(__builtin_amdgcn_ubfe(ax == 0 ? wr.x : (ax == 1 ? wr.y : wr.z), m_pos ? 16u : 0u, 16u));
                }
            } else {
                // general layout: the byte counts from the origin of the record's region; a wide cell has absolute bounds in its wide record
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c) : "v"(m_pos ? 1 : -1), "v"(bound), "v"(m_v & int(~0u << gw.region_shift())));
                if (maybe_wide && GenWalk<SLIM>::is_wide(rec)) {
                    const uint4 wr = GenWalk<SLIM>::wide_at(a, rec);
                    c = int(__builtin_amdgcn_ubfe(ax == 0 ? wr.x : (ax == 1 ? wr.y : wr.z), m_pos ? 16u : 0u, 16u));
                }
            }
            const float tc = (float(c) * m_cs + m_gmin - m_org) * m_inv;
            texit = detail::fmin2(detail::fmin2(tc, quad_perm_f<9>(tc)), quad_perm_f<82>(tc));
            const float ev = (texit * m_dir + m_org - m_gmin) * m_ginv;
            const int n_exit = c + (m_pos ? 0 : -1), n_other = int(ev);      // both, then one select: no divergent branch on the step's chain
            const int n = texit == tc ? n_exit : n_other;
            const int o_v = m_v;
            m_v = med3_i32(n, m_v, m_pos ? 0x7fffffff : int(0x80000000));
            outside = quad_or(uint32_t(m_v) >= uint32_t(m_dims) ? 1 : 0);
            const uint32_t v = outside ? 0u : uint32_t(m_v);
            if (UNIFORM) {
                const uint32_t d = uint32_t(a.shift);
                const uint32_t rec_idx = quad_sum((__umul24(v >> d, m_stride) << (3u * d)) + ((v & ((1u << d) - 1u)) << m_lsh));
                return *reinterpret_cast<const uint4*>(a.img_blocks + (rec_idx << 4));
            }
            if (TABLE) {
                const uint32_t top = quad_sum(__umul24(v >> uint32_t(a.shift), m_stride));
                if (int(top) != top_idx) {
                    const uint2 t = gather32<uint2>(a.img_table, top << 3);
                    tab_off = t.x; tab_d = t.y & 3u; top_idx = int(top);
                }
                const uint32_t idx = quad_sum(((v >> (uint32_t(a.shift) - tab_d)) & ((1u << tab_d) - 1u)) << __umul24(uint32_t(ax), tab_d));
                return *reinterpret_cast<const uint4*>(a.img_blocks + ((tab_off + idx) << 4));
            }
            if (outside) return make_uint4(0u, 0u, 0u, 0u);
            const uint32_t k = gw.bks & 3u, s = gw.bks >> 2;
            // still inside the block of the last look-up (no axis left its region)?  then one gather; else from the top level again
            if (gw.blk != ~0u && !quad_or(((uint32_t(m_v) ^ uint32_t(o_v)) >> (s + k)) != 0u ? 1 : 0)) return GenWalk<SLIM>::rec_at(a, gw.blk + quad_sum(child_share(v, k, s)));
            gw.restart(a);                   // from the image's virtual top level
            return GenWalk<SLIM>::rec_at(a, a.gen_base + quad_sum(__umul24(v >> uint32_t(a.gen_shift), m_stride)));
        };
        const int my_word = (48 + (sub < NI ? sub : 0) * SLIM) >> 5;
        const uint32_t my_shift = uint32_t(48 + (sub < NI ? sub : 0) * SLIM) & 31u;
        live = __ballot(alive);
        while (live) {
            if (alive) {                                                   // (whole groups: the four lanes of a ray finish together)
                const uint32_t marker = field(ca, LAST, SLIM);
                const bool by_index = !WIDE ? marker == uint32_t(NONE - 1) : (marker | 2u) == uint32_t(NONE - 1);        // NONE - 1: by index, NONE - 3: wide (by index as well)
                const unsigned long long any_by_index = __ballot(by_index);
                const uint4 na = quad_step(ca, any_by_index != 0ull);
                const bool wide_cell = WIDE && marker == uint32_t(NONE - 3);
                const int i0 = int(field(ca, 48, SLIM)), i1 = NI > 1 ? int(field(ca, 48 + SLIM, SLIM)) : NONE,
                          i2 = NI > 2 ? int(field(ca, 48 + 2 * SLIM, SLIM)) : NONE, i3 = NI > 3 ? int(field(ca, 48 + 3 * SLIM, SLIM)) : NONE;
                // this lane's id: field `sub` of the 80 id bits from bit 48 on -- two words chosen by the lane's constants, one funnel shift
                const uint32_t id_lo = my_word == 1 ? ca.y : (my_word == 2 ? ca.z : ca.w), id_hi = my_word == 1 ? ca.z : (my_word == 2 ? ca.w : 0u);
                const int mine_id = sub < NI ? int(__builtin_amdgcn_alignbit(id_hi, id_lo, my_shift) & uint32_t(NONE)) : NONE;
                const int inl = by_index ? NONE : mine_id;
                auto accept = [&](int ok, float t, float ad, int ref) {            // prims.h:284-292 with the tmax of this moment
                    if (ok && ad * hit_t > t) { const float inv_det = 1.0f / ad; hit_t = t * inv_det; hit_id = ref; }
                };
                if (any_by_index == 0ull) {
                    // the common step: inline lists only, one round
                    TriCand cd; cd.t = 0.0f; cd.abs_det = 0.0f; cd.ok = false;
                    int mine_now = inl;
                    if (MAILBOX) {
                        // the group's mailbox (its first lane's slot): a lane whose id is in it skips its test; the ids of this cell then replace the
                        // oldest entries -- the first lane of the group stores them (one dword per id: no registers for the old entries)
                        const int4 m = mailbox[lane & ~3];
                        if (inl != NONE && (inl == m.x || inl == m.y || inl == m.z || inl == m.w)) mine_now = NONE;
                        if (sub == 0 && i0 != NONE) {
                            int* mb = reinterpret_cast<int*>(&mailbox[lane]);
                            if (i3 != NONE)      { mb[0] = i0; mb[1] = i1; mb[2] = i2; mb[3] = i3; }
                            else if (i2 != NONE) { mb[0] = m.w; mb[1] = i0; mb[2] = i1; mb[3] = i2; }
                            else if (i1 != NONE) { mb[0] = m.z; mb[1] = m.w; mb[2] = i0; mb[3] = i1; }
                            else                 { mb[0] = m.y; mb[1] = m.z; mb[2] = m.w; mb[3] = i0; }
                        }
                    }
                    if (mine_now != NONE) cd = tri_candidate(tri_vec(mine_now), org, dir, tmin);
                    HG_NOPS();
                    const unsigned long long cand = __ballot(cd.ok);
                    if (cand != 0ull) {
                        // replay the acceptance in list order; every lane of the group computes the same.  A list position at which no
                        // group of the wavefront holds a candidate is skipped (bit s of every nibble of `cand` = position s).
                        const int okv = cd.ok ? 1 : 0;
                        if (cand & 0x1111111111111111ull) { const int ok = quad_bcast_i<0>(okv); const float t = quad_bcast_f<0>(cd.t), ad = quad_bcast_f<0>(cd.abs_det); accept(ok, t, ad, i0); }
                        if (NI > 1 && (cand & 0x2222222222222222ull)) { const int ok = quad_bcast_i<1>(okv); const float t = quad_bcast_f<1>(cd.t), ad = quad_bcast_f<1>(cd.abs_det); accept(ok, t, ad, i1); }
                        if (NI > 2 && (cand & 0x4444444444444444ull)) { const int ok = quad_bcast_i<2>(okv); const float t = quad_bcast_f<2>(cd.t), ad = quad_bcast_f<2>(cd.abs_det); accept(ok, t, ad, i2); }
                        if (NI > 3 && (cand & 0x8888888888888888ull)) { const int ok = quad_bcast_i<3>(okv); const float t = quad_bcast_f<3>(cd.t), ad = quad_bcast_f<3>(cd.abs_det); accept(ok, t, ad, i3); }
                    }
                } else {
                    // some list of the wavefront is given by index (more ids than a record holds): four ids per round as well, lane s takes
                    // ids s, s + 4, ...; the groups with inline lists take part in the first round
                    const uint32_t li_begin = wide_cell ? GenWalk<SLIM>::wide_at(a, ca).w : field(ca, 48, 32), li_count = by_index ? field(ca, 80, 20) : 0u;
                    int mine = inl;
                    if (uint32_t(sub) < li_count) mine = ref_at(li_begin + uint32_t(sub));
#pragma unroll 1
                    for (uint32_t next = 4u + uint32_t(sub); __ballot(mine != NONE) != 0ull; next += 4u) {
                        int ahead = NONE;
                        if (next < li_count) ahead = ref_at(li_begin + next);       // the id of the next round, in flight during this one
                        TriCand cd; cd.t = 0.0f; cd.abs_det = 0.0f; cd.ok = false;
                        if (mine != NONE) cd = tri_candidate(tri_for(mine), org, dir, tmin);
                        if (__ballot(cd.ok) != 0ull) {
                            const int okv = cd.ok ? 1 : 0;
                            { const int ok = quad_bcast_i<0>(okv); const float t = quad_bcast_f<0>(cd.t), ad = quad_bcast_f<0>(cd.abs_det); accept(ok, t, ad, quad_bcast_i<0>(mine)); }
                            { const int ok = quad_bcast_i<1>(okv); const float t = quad_bcast_f<1>(cd.t), ad = quad_bcast_f<1>(cd.abs_det); accept(ok, t, ad, quad_bcast_i<1>(mine)); }
                            { const int ok = quad_bcast_i<2>(okv); const float t = quad_bcast_f<2>(cd.t), ad = quad_bcast_f<2>(cd.abs_det); accept(ok, t, ad, quad_bcast_i<2>(mine)); }
                            { const int ok = quad_bcast_i<3>(okv); const float t = quad_bcast_f<3>(cd.t), ad = quad_bcast_f<3>(cd.abs_det); accept(ok, t, ad, quad_bcast_i<3>(mine)); }
                        }
                        mine = ahead;
                    }
                }
                if (hit_t <= texit || outside) alive = false;
                ca = na;
                if (GENERAL && alive) quad_descend(ca);
            }
            live = __ballot(alive);
            if (COST) iters++;
        }
    }
    if (pending) nt_store4(a.hits + id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
    if (COST && lane == 0 && cost_at) atomicMax(cost_at, iters + min(iters, cost_bonus));
}



} // namespace hagrid_trav

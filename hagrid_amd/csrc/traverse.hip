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
#include "wave_prims.h"

#include <cstdlib>
#include <cstring>

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
    unsigned long long* __restrict__ stats;  // optional 8 batch counters
    const int* __restrict__ perm;            // optional traversal order (ray binning): slot i processes ray perm[i]
    const int* __restrict__ perm_flag;       // optional, device: 0 = ignore perm (automatic binning decided against it)
    const int* __restrict__ row_len;         // optional, device: row length found by detect_ray_rows (0 = none)
    int row_len_hint;                        // > 0: row length given by the caller ("traverse.image_width")
    int super_log2;                          // tile packets: tiles per super-tile edge, log2
    int xcd_chunk_log2;                      // tile packets: blocks per XCD chunk, log2 (< 0: one eighth of the range per XCD)
    const int* __restrict__ tile_order;      // diagnostic (hagrid_kat_tile_order): packet b processes tile tile_order[b]
    unsigned long long* __restrict__ wave_times; // diagnostic (hagrid_kat_wave_times): start / end of every wavefront, 100 MHz wall clock
    const uint2* __restrict__ img_table;     // traversal image (trav_image.hip) or null
    const unsigned char* __restrict__ img_blocks;
    int num_rays;
    int lds_pad;                  // host only: dynamic LDS bytes per block of the tail kernel (experiments: fewer resident wavefronts)
    int quad_first_block;         // tail kernel: blocks from this index on start with four lanes per ray (16 rays each, four blocks per tile); INT_MAX: none
    int id_is_steps;              // statistics kernel: Hit.id receives the step count, as the reference's kernel writes it (traverse.cu:93)
    int shift;
    int dims_x, dims_y, dims_z;   // virtual resolution dims << shift
    int top_x, top_y;             // top-level resolution (x, y)
    int top_xy;                   // top_x * top_y when it fits 24 bits (NARROW kernels), else 0
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
// streaming (read-once / write-once) accesses for rays and hits: keep them out of the way of the grid in L2
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, float x, float y, float z, float w) {
    f32x4 v; v.x = x; v.y = y; v.z = z; v.w = w;
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}

// ---- tile packets ---------------------------------------------------------------------------------------------------
// A batch of camera rays arrives in image order (gen_rays, main.cpp:55-66: ray y * w + x), so 64 consecutive rays are
// a 64 x 1 pixel strip: the lanes of a wavefront fan out over 64 pixel columns and share few cells.  An 8 x 8 pixel
// tile per wavefront keeps the packet compact in both image directions (the vector L1 serves fewer distinct lines per
// load instruction), and listing the tiles along a Z curve inside super-tiles keeps neighbouring wavefronts -- and the
// contiguous block range each XCD receives -- compact as well (L2).  Measured on MI355X, soup-1M, unchanged kernel,
// rays reordered on the host (tools/dev_tile_order.py): 1024^2 rays 0.406 -> 0.355 ms, 4096^2 rays 3.36 -> 2.10 ms.
// The ray buffer stays in the reference's order and every hit goes to its ray's slot: only the lane <-> ray assignment
// changes, so results are identical.  The row length w comes from the caller ("traverse.image_width") or from
// detect_ray_rows below; w must be a multiple of 8; rows beyond the last multiple of 8 and rays beyond the last full
// row keep the identity assignment.
__device__ __forceinline__ uint32_t compact1by1(uint32_t v) {   // even bits of v, packed
    v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
    return (v | (v >> 8)) & 0x0000ffffu;
}

// Blocks are dispatched round-robin over the 8 XCDs (private L2 each).  split: XCD x runs the x-th eighth of the logical
// block range (bands of a batch in buffer order).  chunked: XCD x runs the logical chunks x, x + 8, x + 16, ... of
// 2^k blocks each -- along the Z curve an aligned run of 4^j tiles is a compact square, so every L2 serves compact
// squares while the 8 XCDs work side by side on neighbouring ones: an image whose cost is concentrated in one region
// (scene in the middle, sky around it) still loads them evenly.
__device__ __forceinline__ int xcd_split(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
__device__ __forceinline__ int xcd_chunked(int b, int nb, int chunk_log2) {
    const int full = (nb >> (chunk_log2 + 3)) << (chunk_log2 + 3);       // blocks in complete groups of 8 chunks
    if (b >= full) return full + xcd_split(b - full, nb - full);
    const int xcd = b & 7, j = b >> 3;
    return ((((j >> chunk_log2) << 3) + xcd) << chunk_log2) + (j & ((1 << chunk_log2) - 1));
}

__device__ __forceinline__ int tile_packet_row_len(const TraverseArgs& a) {      // 0: buffer order
    const int w = a.row_len_hint > 0 ? a.row_len_hint : (a.row_len ? __builtin_amdgcn_readfirstlane(*a.row_len) : 0);
    return (w < 8 || (w & 7) || a.num_rays / w < 8) ? 0 : w;
}

__device__ __forceinline__ int tile_packet_slot(const TraverseArgs& a, int w, int b, int lane) {
    const int identity = b * 64 + lane;
    if (!w) return identity;
    const int tiles_x = w >> 3, tiles_y = (a.num_rays / w) >> 3;
    if (b >= tiles_x * tiles_y) return identity;             // ragged rows at the bottom, rays past the last full row
    const int S = 1 << a.super_log2;
    const int band = b / (tiles_x * S), in_band = b - band * tiles_x * S;
    const int hb = min(S, tiles_y - band * S);               // tile rows in this band of super-tiles
    const int col = in_band / (S * hb), in_super = in_band - col * S * hb;
    const int wc = min(S, tiles_x - col * S);                // tile columns in this super-tile
    int tx, ty;
    if (wc == S && hb == S) { tx = int(compact1by1(uint32_t(in_super))); ty = int(compact1by1(uint32_t(in_super) >> 1)); }
    else                    { ty = in_super / wc; tx = in_super - ty * wc; }
    const int px = ((col * S + tx) << 3) + (lane & 7), py = ((band * S + ty) << 3) + (lane >> 3);
    return py * w + px;
}

// Row length of an image-ordered batch, or 0: the (origin, direction) of consecutive rays advances by a constant step
// s = ray[1] - ray[0] along a row (perspective: the direction; orthographic: the origin) and jumps at a row break.
// w = index of the first break; accepted if it is a multiple of 8, the second row starts with the same step and, when
// there is a third row, ray 2w is a break too.  One workgroup; the answer stays on the device (no host round trip).
// A wrong answer can only cost speed: any row length gives a valid lane <-> ray assignment.
constexpr int kDetectBlock = 1024;
constexpr int kDetectLimit = 1 << 16;

__device__ __forceinline__ float ray_step_dev2(const float4* __restrict__ rays, int i, const float (&s)[6]) {
    // squared distance between (ray[i+1] - ray[i]) and s over origin and direction
    const float4 a0 = rays[2 * size_t(i)], a1 = rays[2 * size_t(i) + 1], b0 = rays[2 * size_t(i) + 2], b1 = rays[2 * size_t(i) + 3];
    const float d[6] = {b0.x - a0.x - s[0], b0.y - a0.y - s[1], b0.z - a0.z - s[2], b1.x - a1.x - s[3], b1.y - a1.y - s[4], b1.z - a1.z - s[5]};
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
}

// Second criterion, for image-ordered batches whose directions are not a function of the pixel (bounce rays leaving the
// primary hit points): the ORIGINS of vertically neighbouring pixels are close.  Every candidate row length w = 64, 72, ...
// gets the clamped mean squared distance |org[i + w] - org[i]|^2 / tau^2 over 256 sampled i (tau = 1/64 of the grid
// diagonal); the true row length is the minimum (one pixel apart; w +- 8 is eight pixels apart, 2w two rows).  Accepted if
// it stands out from the mean over all candidates and horizontally neighbouring origins are as close (but not all identical).  Blocks 1.. of the same
// launch do the scoring, the block that finishes last picks -- no extra launch, nothing waits; used only when the first
// criterion found nothing.  Like the first one it can only cost speed if it is wrong.
constexpr int kRowCandidates = 2048;                 // w = 8 * (c + 8): 64 .. 16440
constexpr int kRowSamples = 256;

__global__ void __launch_bounds__(kDetectBlock) detect_ray_rows(const float4* __restrict__ rays, int n, int* __restrict__ out,
                                                                int* __restrict__ scores, float inv_tau2, int origins_only) {
    __shared__ int first_break;
    __shared__ int lds_score[kDetectBlock / 64];
    __shared__ int ticket;
    __shared__ unsigned long long best[kDetectBlock / 64];
    __shared__ long long sums[kDetectBlock / 64];
    __shared__ int counts[kDetectBlock / 64];
    // origins_only: a second launch after the first criterion; nothing to do if that one found the row length
    if (origins_only && __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) return;
    if (blockIdx.x == 0 && !origins_only) {
        if (threadIdx.x == 0) { first_break = 0x7fffffff; out[0] = 0; }
        int w1 = 0;
        if (n >= 128) {
            const float4 a0 = rays[0], a1 = rays[1], b0 = rays[2], b1 = rays[3];
            const float s[6] = {b0.x - a0.x, b0.y - a0.y, b0.z - a0.z, b1.x - a1.x, b1.y - a1.y, b1.z - a1.z};
            const float s2 = s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
            if ((s2 > 0.0f) && (s2 < 3.0e38f)) {
                const float tol = 0.25f * s2;
                const int limit = min(n - 1, kDetectLimit);              // pairs (i, i + 1) with i < limit
                __syncthreads();
                for (int base = 1; base < limit; base += kDetectBlock) {
                    const int i = base + int(threadIdx.x);
                    if (i < limit && !(ray_step_dev2(rays, i, s) <= tol)) atomicMin(&first_break, i + 1);
                    __syncthreads();
                    const int found = first_break;
                    __syncthreads();
                    if (found != 0x7fffffff) break;
                }
                if (threadIdx.x == 0) {
                    const int w = first_break;
                    bool ok = w != 0x7fffffff && w >= 8 && (w & 7) == 0 && n / w >= 8;
                    if (ok) ok = ray_step_dev2(rays, w, s) <= tol;                                   // second row advances like the first
                    if (ok && n > 2 * w) ok = !(ray_step_dev2(rays, 2 * w - 1, s) <= tol);            // and ends where the first did
                    w1 = ok ? w : 0;
                }
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(out, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gridDim.x == 1) return;
    } else if (blockIdx.x != 0) {
        // four candidates per block, one per group of 256 threads; candidate kRowCandidates is the horizontal neighbour (w = 1)
        const int c = (int(blockIdx.x) - 1) * 4 + int(threadIdx.x >> 8);
        const int w = c < kRowCandidates ? 8 * (c + 8) : 1;
        int v = 1024;
        if (c <= kRowCandidates && n - w > 0) {
            uint32_t h = uint32_t(c) * 2654435761u + (threadIdx.x & 255u) * 40503u + 12345u;
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
            const int i = int(h % uint32_t(n - w));
            const float4 p = rays[2 * size_t(i)], q = rays[2 * size_t(i + w)];
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float d2 = (dx * dx + dy * dy + dz * dz) * inv_tau2;
            v = d2 < 1.0f ? int(d2 * 1024.0f) : 1024;            // NaN -> 1024
        }
        v = wave_sum(v);
        if (lane_id() == 0) lds_score[wave_id()] = v;
        __syncthreads();
        if ((threadIdx.x & 255) == 0 && c <= kRowCandidates) {
            const int g = int(threadIdx.x >> 8) * 4;
            __hip_atomic_store(scores + c, lds_score[g] + lds_score[g + 1] + lds_score[g + 2] + lds_score[g + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the block that finishes last picks.  Scores and out[0] travel as agent-scope atomics (a __threadfence per thread costs an
    // L2 write-back each on this part: 150 us for the launch); the ticket is the release / acquire point.
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(scores + kRowCandidates + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != int(gridDim.x) - 1) return;
    unsigned long long m = ~0ull;                                    // (score << 32 | w), minimum
    long long total = 0; int counted = 0;                            // mean score of the candidates
    for (int c = int(threadIdx.x); c < kRowCandidates; c += kDetectBlock) {
        const int w = 8 * (c + 8);
        if (n / w >= 8) {
            const unsigned sc = (unsigned)__hip_atomic_load(scores + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long key = ((unsigned long long)sc << 32) | unsigned(w);
            m = key < m ? key : m;
            total += sc; counted++;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64); m = o < m ? o : m;
        total += __shfl_xor(total, d, 64); counted += __shfl_xor(counted, d, 64);
    }
    if (lane_id() == 0) { best[wave_id()] = m; sums[wave_id()] = total; counts[wave_id()] = counted; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kDetectBlock / 64; i++) { m = best[i] < m ? best[i] : m; total += sums[i]; counted += counts[i]; }
        // The row length stands out: its score lies clearly (8 % of the clamp) below the mean of all candidates, and so does the
        // score of horizontally neighbouring origins.  Unrelated origins score ~1.0 everywhere; rows of hit points with
        // silhouettes and rays that left the scene 0.2-0.9.  horizontal == 0: all origins coincide (a pinhole camera) -- no information.
        const int full = kRowSamples * 1024;
        const long long mean = counted ? total / counted : 0;
        const int horizontal = __hip_atomic_load(scores + kRowCandidates, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int w1 = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long bar = mean - full * 8 / 100;
        if (w1 == 0 && m != ~0ull && (long long)(m >> 32) < bar && horizontal < bar && horizontal > 0) out[0] = int(unsigned(m));
        scores[kRowCandidates + 1] = 0;                              // ready for the next batch
    }
}

// intersect_prim_ray as the reference compiles it with COMPUTE_UVS (prims.h:266-295, :285-288): same test, the accepted
// hit also stores its barycentrics.  (include/hagrid/prims.h has the same code behind the same macro; the kernels need both
// forms in one translation unit.)
__device__ __forceinline__ bool intersect_prim_ray_uvs(const Tri& tri, const Ray& ray, int id, Hit& hit) {
    const vec3 n = tri.normal();
    const vec3 c = tri.v0 - ray.org;
    const vec3 r = cross(ray.dir, c);
    const float det = dot(n, ray.dir);
    const float abs_det = detail::fabs1(det);
    const float u = prodsign(dot(r, tri.e2), det);
    const float v = prodsign(dot(r, tri.e1), det);
    const float w = abs_det - u - v;
    const float eps = 1e-9f;
    if (u >= -eps && v >= -eps && w >= -eps) {
        const float t = prodsign(dot(n, c), det);
        if (t >= abs_det * ray.tmin && abs_det * ray.tmax > t) {
            const float inv_det = 1.0f / abs_det;
            hit.t = t * inv_det;
            hit.u = u * inv_det;
            hit.v = v * inv_det;
            hit.id = id;
            return true;
        }
    }
    return false;
}

// NARROW: every gather is base (scalar registers) + unsigned 32-bit byte offset (one VALU shift instead of a sign
// extension and a 64-bit add), index products are 24-bit multiplies (full rate; 32-bit multiplies are quarter rate) and the
// range test is three unsigned compares.  The host selects it when every array it indexes is smaller than 4 GB and the
// top-level resolution fits 23 bits per axis.
__device__ __forceinline__ int med3_i32(int a, int b, int c) {          // the median of three (one VALU instruction)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <typename T>
__device__ __forceinline__ T gather32(const void* base, uint32_t byte_offset) {
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_offset);
}

// MODE: HAGRID_TRAVERSE_ANY_HIT (the ray is done at its first accepted intersection: shadow rays) and / or
// HAGRID_TRAVERSE_UVS (barycentrics stored with the hit) -- SURVEY.md 8(f) row 4; 0 is the reference's traversal.
// A triangle round in which every live lane tests the SAME triangle -- one live lane (23 % of the rounds of the 1M-ray batch,
// profiles/dev_r2_generations.txt items 10-12), or neighbouring rays in the same cell at the same place of its list (common in dense
// batches) -- still costs the CU's vector-memory path its fixed ~12 cycles per load instruction and a cycle per lane: there the
// triangle comes through the scalar cache (constant address space + a uniform address = s_load), no vector-memory instruction at
// all, and the test reads it from scalar registers.  HG_SOLO=0 compiles the path out.
#ifndef HG_SOLO
#define HG_SOLO 1
#endif
__device__ __forceinline__ Tri load_tri_scalar(const float4* tris, int ref) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const f4 __attribute__((address_space(4)))* const_f4;
    const_f4 p = (const_f4)(reinterpret_cast<uintptr_t>(tris) + size_t(uint32_t(__builtin_amdgcn_readfirstlane(ref))) * 48u);
    const f4 p0 = p[0], p1 = p[1], p2 = p[2];
    return Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w);
}

template <bool SMALL, int BLOCK, bool NARROW, unsigned MODE>
__global__ void __launch_bounds__(BLOCK, 8) traverse_kernel_v2(const TraverseArgs a) {
    constexpr bool ANY = (MODE & HAGRID_TRAVERSE_ANY_HIT) != 0, UVS = (MODE & HAGRID_TRAVERSE_UVS) != 0;
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
                const bool got = UVS ? intersect_prim_ray_uvs(tri_at(ref), Ray(org, tmin, dir, hit.t), ref, hit)
                                     : intersect_prim_ray(tri_at(ref), Ray(org, tmin, dir, hit.t), ref, hit);
                ref = (ANY && got) ? -1 : next;
            }
            if ((ANY && hit.id >= 0) || hit.t <= texit || outside) break;
            c = nc;
        }
    }
    nt_store4(a.hits + id, __int_as_float(hit.id), hit.t, UVS ? hit.u : 0.0f, UVS ? hit.v : 0.0f);
}


// ---- image kernel: v2 over the traversal image ------------------------------------------------------------------------
// Same ray arithmetic as v1 / v2 / the oracle; the cell comes from the traversal image (trav_image.hip): the table entry
// of the top-level cell (kept in registers while the ray stays inside it), one slot byte, one 32-byte record that carries
// the bounds and -- for lists of up to four -- the reference ids themselves.  The next cell's slot + record are fetched
// before the current cell's triangles are tested, as in v2.
template <bool FLAT>
__device__ __forceinline__ const uint4* image_record(const TraverseArgs& a, uint2 tab, int vx, int vy, int vz) {
    const uint32_t meta = tab.y;
    const int d = int(meta & 3u), w = int((meta >> 2) & 1u);
    const unsigned char* base = a.img_blocks + size_t(tab.x) * 128u;
    const int s = a.shift - d, m = (1 << d) - 1;
    const int idx = ((vx >> s) & m) + ((((vy >> s) & m) + (((vz >> s) & m) << d)) << d);   // 0 when d == 0
    if (FLAT) return reinterpret_cast<const uint4*>(base + uint32_t(idx) * 32u);             // the record itself: one gather per step
    uint32_t slot = base[idx << w];                                                           // d == 0: a byte of the record, ignored
    if (w) slot |= uint32_t(base[(idx << 1) + 1]) << 8;
    uint32_t ebytes = (1u << (3 * d)) << w;
    ebytes = d ? (ebytes < 32u ? 32u : ebytes) : 0u;
    if (!d) slot = 0;
    return reinterpret_cast<const uint4*>(base + ebytes + slot * 32u);
}

// A `deep` record: the block does not resolve this voxel; continue the walk of the construction format at the entry the
// record names and bring the cell into record form (list by index, never inline).
__device__ __forceinline__ void image_resolve_deep(const TraverseArgs& a, int vx, int vy, int vz, uint4& ca, uint4& cb) {
    uint32_t w = a.entries[cb.x];
    int depth = int(cb.y);
    while (w & 3u) {
        const int k = int(w & 3u);
        depth += k;
        const int s = a.shift - depth, m = (1 << k) - 1;
        w = a.entries[(w >> 2) + ((vx >> s) & m) + ((((vy >> s) & m) + (((vz >> s) & m) << k)) << k)];
    }
    const int4* p = reinterpret_cast<const int4*>(a.cells) + 2 * size_t(w >> 2);
    const int4 lo = p[0], hi = p[1];
    ca.x = uint32_t(lo.x) | (uint32_t(hi.x) << 16);
    ca.y = uint32_t(lo.y) | (uint32_t(hi.y) << 16);
    ca.z = uint32_t(lo.z) | (uint32_t(hi.z) << 16);
    ca.w = uint32_t(hi.w - lo.w) | 0x80000000u;
    cb.x = uint32_t(lo.w);
}

// Records that are links: 0xfffffffe = nested block (three more levels of the same flat form), 0xffffffff = deep (construction
// format).  Dense spots of very non-uniform scenes only; the common record never gets here.
__device__ __forceinline__ void image_resolve_links(const TraverseArgs& a, int vx, int vy, int vz, uint4& ca, uint4& cb, uint32_t& nest_off, uint32_t& nest_meta) {
    while (ca.w == 0xfffffffeu) {
        nest_off = cb.x; nest_meta = cb.y;
        const int d = int(cb.y & 3u), s = a.shift - int(cb.y >> 8) - d, m = (1 << d) - 1;
        const uint32_t idx = uint32_t((vx >> s) & m) + (uint32_t(((vy >> s) & m) + (((vz >> s) & m) << d)) << d);
        const uint4* p = reinterpret_cast<const uint4*>(a.img_blocks + size_t(cb.x) * 128u + size_t(idx) * 32u);
        ca = p[0]; cb = p[1];
    }
    if (ca.w == 0xffffffffu) image_resolve_deep(a, vx, vy, vz, ca, cb);
}

// NARROW: 32-bit offsets off scalar bases as in v2 (the host checks that image, triangles, entries and cells are < 4 GB)
// UNIFORM (with FLAT and NARROW): every block has (2^shift)^3 records and block T starts at T * (2^shift)^3 -- no table
// TIMES: diagnostic instantiation that records the wall clock at the start and the end of every wavefront (tools/dev_wave_timeline.py)
// SLIM (with FLAT and NARROW, grids of at most three levels): 16-byte records, SLIM = bits per packed reference id (trav_image.hip,
// "Slim records"); 0 = 32-byte records.  UNIFORM: bounds as offsets from the voxel; table layout: from the top-level cell's origin.
template <int BLOCK, bool FLAT, bool NARROW, bool UNIFORM, unsigned MODE, bool TIMES = false, int SLIM = 0>
__global__ void __launch_bounds__(BLOCK, 8) traverse_kernel_img(const TraverseArgs a) {
    constexpr bool ANY = (MODE & HAGRID_TRAVERSE_ANY_HIT) != 0, UVS = (MODE & HAGRID_TRAVERSE_UVS) != 0;
    static_assert(SLIM == 0 || (FLAT && NARROW), "slim records are read by the flat narrow kernels only");
    constexpr int NONE = SLIM ? (1 << (SLIM ? SLIM : 1)) - 1 : -1;          // the id field of an unused list slot
    struct Stamp {
        unsigned long long* p;
        __device__ Stamp(unsigned long long* q) : p(q) { if (TIMES && threadIdx.x == 0) p[0] = wall_clock64(); }
        __device__ ~Stamp() { if (TIMES) { const unsigned long long t = wall_clock64(); atomicMax(p + 1, t); } }   // the last lane to leave
    } stamp(TIMES ? a.wave_times + 2 * size_t(blockIdx.x) : nullptr);
    const int* perm = (a.perm && (!a.perm_flag || __builtin_amdgcn_readfirstlane(*a.perm_flag))) ? a.perm : nullptr;
    const int w = (BLOCK == 64 && !perm) ? tile_packet_row_len(a) : 0;
    const int b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk_log2) : xcd_split(blockIdx.x, gridDim.x);
    const int slot = w ? tile_packet_slot(a, w, (TIMES && a.tile_order) ? a.tile_order[b] : b, threadIdx.x) : b * BLOCK + threadIdx.x;
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
            if (NARROW) return int(uint32_t(x >> a.shift) + __umul24(uint32_t(a.top_x), uint32_t(y >> a.shift)) + __umul24(uint32_t(a.top_xy), uint32_t(z >> a.shift)));
            return (x >> a.shift) + a.top_x * ((y >> a.shift) + a.top_y * (z >> a.shift));
        };
        auto table_at = [&](int t) -> uint2 { return NARROW ? gather32<uint2>(a.img_table, uint32_t(t) << 3) : a.img_table[t]; };
        uint32_t nest = ~0u;                                                    // innermost nested block the ray is inside (FLAT + NARROW, table layout)
        int nest_x = 0, nest_y = 0, nest_z = 0;                                 // ... and the voxel that led there
        // record of a voxel: FLAT + NARROW is one address computation off the scalar base
        auto record = [&](uint2 tab, int x, int y, int z, uint4& ra, uint4& rb) {
            if (UNIFORM) {
                const int d = a.shift, m = (1 << d) - 1;
                const uint32_t idx = uint32_t(x & m) + (uint32_t((y & m) + ((z & m) << d)) << d);
                const uint32_t o = ((uint32_t(top_index(x, y, z)) << (3 * d)) + idx) << (SLIM ? 4 : 5);
                const uint4* p = reinterpret_cast<const uint4*>(a.img_blocks + o);
                ra = p[0];
                if (!SLIM) rb = p[1];
            } else if (FLAT && NARROW && SLIM) {          // table layout, no links: block offset in records, depth of the block
                const int d = int(tab.y & 3u), s = a.shift - d, m = (1 << d) - 1;
                const uint32_t idx = uint32_t((x >> s) & m) + (uint32_t(((y >> s) & m) + (((z >> s) & m) << d)) << d);
                ra = *reinterpret_cast<const uint4*>(a.img_blocks + ((tab.x + idx) << 4));
            } else if (FLAT && NARROW) {
                int d = int(tab.y & 3u), s = a.shift - d;
                uint32_t base = tab.x;
                if (nest != ~0u) {
                    const int sr = a.shift - int(nest >> 27);               // finest-level voxels per root cell of the nested block, log2
                    if ((((x ^ nest_x) | (y ^ nest_y) | (z ^ nest_z)) >> sr) == 0) { d = int((nest >> 25) & 3u); s = sr - d; base = nest & 0x1ffffffu; }
                }
                const int m = (1 << d) - 1;
                const uint32_t idx = uint32_t((x >> s) & m) + (uint32_t(((y >> s) & m) + (((z >> s) & m) << d)) << d);
                const uint32_t o = (base << 7) + (idx << 5);
                const uint4* p = reinterpret_cast<const uint4*>(a.img_blocks + o);
                ra = p[0]; rb = p[1];
            } else {
                const uint4* p = image_record<FLAT>(a, tab, x, y, z);
                ra = p[0]; rb = p[1];
            }
        };
        auto tri_at = [&](int ref) -> Tri {
            if (!NARROW) return load_tri(a.tris, ref);
            uint32_t r3, o;
            asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r3) : "v"(ref));
            asm("v_lshlrev_b32 %0, 4, %1" : "=v"(o) : "v"(r3));
            const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.tris) + o);
            const float4 p0 = p[0], p1 = p[1], p2 = p[2];
            return Tri(vec3(p0.x, p0.y, p0.z), p0.w, vec3(p1.x, p1.y, p1.z), p1.w, vec3(p2.x, p2.y, p2.z), p2.w);
        };

        auto tri_for = [&](int ref) -> Tri {
            if (HG_SOLO && NARROW) {
                const int r0 = __builtin_amdgcn_readfirstlane(ref);
                const unsigned long long others = __ballot(ref != r0);
                if (others == 0ull) return load_tri_scalar(a.tris, r0);
            }
            return tri_at(ref);
        };
        // which half of a bounds word is the exit plane (slim records: which byte, and the direction the offset counts in)
        const uint32_t ox = SLIM ? (px ? 8u : 0u) : (px ? 16u : 0u), oy = SLIM ? (py ? 24u : 16u) : (py ? 16u : 0u), oz = SLIM ? (pz ? 8u : 0u) : (pz ? 16u : 0u);
        const int sgx = px ? 1 : -1, sgy = py ? 1 : -1, sgz = pz ? 1 : -1;
        const int bx = px ? 0 : -1, by = py ? 0 : -1, bz = pz ? 0 : -1;               // the voxel just past it
        const int lim_x = px ? 0x7fffffff : int(0x80000000), lim_y = py ? 0x7fffffff : int(0x80000000), lim_z = pz ? 0x7fffffff : int(0x80000000);
        int top_idx = UNIFORM ? 0 : top_index(vx, vy, vz);
        uint2 tab = UNIFORM ? make_uint2(0u, 0u) : table_at(top_idx);
        uint4 ca, cb = make_uint4(0u, 0u, 0u, 0u);
        record(tab, vx, vy, vz, ca, cb);

        for (;;) {
            if (!UNIFORM && !SLIM && ca.w >= 0xfffffffeu) {                 // (the table-free layout and slim records need shift <= 3: every block resolves its cell fully)
                // remember the innermost nested block and the voxel that led there: while the ray stays inside that block's root
                // cell the next records are fetched from it directly (one gather per step again)
                uint32_t off = ~0u, meta = 0u;
                image_resolve_links(a, vx, vy, vz, ca, cb, off, meta);
                if (!UNIFORM && FLAT && NARROW && off != ~0u) {
                    nest = off | (meta & 3u) << 25 | (meta >> 8) << 27;        // offset < 2^25 units (NARROW), depth of the block, depth of its root
                    nest_x = vx; nest_y = vy; nest_z = vz;
                }
            }
            // lo or hi of every axis: one bit-field extract per axis (offset 0 or 16, fixed per ray)
            int cx, cy, cz;
            if (SLIM && !UNIFORM) {     // table layout: biased byte offsets from the origin of the top-level cell
                const int org_mask = ~((1 << a.shift) - 1);
                cx = (vx & org_mask) + int(__builtin_amdgcn_ubfe(ca.x, ox, 8u)) - 128;
                cy = (vy & org_mask) + int(__builtin_amdgcn_ubfe(ca.x, oy, 8u)) - 128;
                cz = (vz & org_mask) + int(__builtin_amdgcn_ubfe(ca.y, oz, 8u)) - 128;
            } else if (SLIM) {     // byte offsets from the voxel the record belongs to
                // voxel +- offset as ONE multiply-add with the ray's sign (the compiler expands a plain multiply by +-1 into negate + select)
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cx) : "v"(sgx), "v"(__builtin_amdgcn_ubfe(ca.x, ox, 8u)), "v"(vx));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cy) : "v"(sgy), "v"(__builtin_amdgcn_ubfe(ca.x, oy, 8u)), "v"(vy));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cz) : "v"(sgz), "v"(__builtin_amdgcn_ubfe(ca.y, oz, 8u)), "v"(vz));
            } else { cx = int(__builtin_amdgcn_ubfe(ca.x, ox, 16u)); cy = int(__builtin_amdgcn_ubfe(ca.y, oy, 16u)); cz = int(__builtin_amdgcn_ubfe(ca.z, oz, 16u)); }
            const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
            const float texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
            const vec3 ev = (texit * dir + org - gmin) * ginv;
            const int nx = texit == tcell.x ? cx + bx : int(ev.x);
            const int ny = texit == tcell.y ? cy + by : int(ev.y);
            const int nz = texit == tcell.z ? cz + bz : int(ev.z);
            // never backwards: max with the current voxel along a positive direction, min along a negative one -- the median of
            // (new, current, +-infinity), one instruction per axis
            if (UNIFORM) { vx = med3_i32(nx, vx, lim_x); vy = med3_i32(ny, vy, lim_y); vz = med3_i32(nz, vz, lim_z); }
            else { vx = px ? max(nx, vx) : min(nx, vx); vy = py ? max(ny, vy) : min(ny, vy); vz = pz ? max(nz, vz) : min(nz, vz); }   // (the table layouts have no registers to spare)
            const bool outside = (uint32_t(vx) >= uint32_t(a.dims_x)) | (uint32_t(vy) >= uint32_t(a.dims_y)) | (uint32_t(vz) >= uint32_t(a.dims_z));

            // next cell: table entry (only when the top-level cell changes) -> record, in flight during the tests below
            if (!UNIFORM) {
                const int ntop = outside ? top_idx : top_index(vx, vy, vz);
                if (ntop != top_idx) { tab = table_at(ntop); top_idx = ntop; }
            }
            uint4 na, nb = make_uint4(0u, 0u, 0u, 0u);
            if (UNIFORM) { const int sx = outside ? 0 : vx, sy = outside ? 0 : vy, sz = outside ? 0 : vz; record(tab, sx, sy, sz, na, nb); }
            else record(tab, vx, vy, vz, na, nb);

            // Lists: inline ids (up to four, unused slots -1) are consumed front to back; a list given by index (bit 31: more
            // than four ids, deep cells) fetches the id of the next test one test ahead, as v2 does.
            auto ref_at = [&](uint32_t i) -> int { return NARROW ? gather32<int>(a.refs, i << 2) : a.refs[i]; };
            bool by_index;
            uint32_t q1, q2, q3, li_begin, li_count;
            int ref;
            if (SLIM) {
                // id fields of SLIM bits from bit 48 on; the last field = NONE - 1 marks a list given by index
                constexpr int NI = 80 / (SLIM ? SLIM : 80), LAST = 48 + (NI - 1) * SLIM;
                auto field = [&](int pos, int n) -> uint32_t {              // pos, n are constants after inlining
                    const uint32_t w[4] = {ca.x, ca.y, ca.z, ca.w};
                    const int i = pos >> 5, o = pos & 31;
                    uint32_t v = w[i] >> o;
                    if (o + n > 32) v |= w[i + 1] << (32 - o);
                    return n == 32 ? v : (v & ((1u << n) - 1u));
                };
                by_index = field(LAST, SLIM) == uint32_t(NONE - 1);
                ref = int(field(48, SLIM));
                q1 = NI > 1 ? field(48 + SLIM, SLIM) : uint32_t(NONE);
                q2 = NI > 2 ? field(48 + 2 * SLIM, SLIM) : uint32_t(NONE);
                q3 = NI > 3 ? field(48 + 3 * SLIM, SLIM) : uint32_t(NONE);
                li_begin = field(48, 32); li_count = field(80, 20);
            } else {
                by_index = int(ca.w) < 0;
                q1 = cb.y; q2 = cb.z; q3 = cb.w;                            // inline: the ids still to test
                ref = int(cb.x);                                            // inline: the first id, or -1 for an empty list
                li_begin = cb.x; li_count = ca.w & 0x7fffffffu;
            }
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
            ca = na; cb = nb;
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

// TIMES: diagnostic instantiation that records the wall clock at the start and the end of every wavefront (tools/dev_wave_timeline.py)
// UNIFORM = false: the table layout of slim records (grids of at most three levels whose top-level cells differ in depth): the block of a
// top-level cell is found through its table entry, kept while the ray stays inside the cell; bounds count from that cell's origin.
template <int SLIM, bool TIMES = false, bool UNIFORM = true>
__global__ void __launch_bounds__(64, 8) traverse_kernel_tail(const TraverseArgs a) {
    constexpr int NONE = (1 << SLIM) - 1, NI = 80 / SLIM, LAST = 48 + (NI - 1) * SLIM;
    __shared__ int lanes_of[64];
    const int lane = threadIdx.x;
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
    const bool quad_start = int(blockIdx.x) >= a.quad_first_block;
    const int group = lane >> 2, sub = lane & 3;
    int b, lane_in_tile = lane;
    if (quad_start) {
        const int q = int(blockIdx.x) - a.quad_first_block, nq = int(gridDim.x) - a.quad_first_block;
        const int lq = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(q, nq, a.xcd_chunk_log2 + 2) : xcd_split(q, nq);
        b = a.quad_first_block + (lq >> 2);
        lane_in_tile = ((((lq >> 1) & 1) << 2) + (group >> 2)) * 8 + ((lq & 1) << 2) + (group & 3);
    } else {
        const int nb = min(int(gridDim.x), a.quad_first_block);
        b = (w && a.xcd_chunk_log2 >= 0) ? xcd_chunked(blockIdx.x, nb, a.xcd_chunk_log2) : xcd_split(blockIdx.x, nb);
    }
    const int slot = w ? tile_packet_slot(a, w, (TIMES && a.tile_order) ? a.tile_order[b] : b, lane_in_tile) : b * 64 + lane_in_tile;
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
    uint32_t tab_off = 0u, tab_d = 0u;                     // table layout: block offset (records) and depth of the top-level cell the ray is in
    int top_idx = -1;
    auto load_record = [&](int x, int y, int z) -> uint4 {
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
    auto cell_step = [&](const uint4& rec, const vec3& inv_dir) -> uint4 {
        const bool px = dir.x >= 0.0f, py = dir.y >= 0.0f, pz = dir.z >= 0.0f;
        int cx, cy, cz;
        if (UNIFORM) {
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cx) : "v"(px ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, px ? 8u : 0u, 8u)), "v"(vx));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cy) : "v"(py ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.x, py ? 24u : 16u, 8u)), "v"(vy));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(cz) : "v"(pz ? 1 : -1), "v"(__builtin_amdgcn_ubfe(rec.y, pz ? 8u : 0u, 8u)), "v"(vz));
        } else {
            const int org_mask = ~((1 << a.shift) - 1);
            cx = (vx & org_mask) + int(__builtin_amdgcn_ubfe(rec.x, px ? 8u : 0u, 8u)) - 128;
            cy = (vy & org_mask) + int(__builtin_amdgcn_ubfe(rec.x, py ? 24u : 16u, 8u)) - 128;
            cz = (vz & org_mask) + int(__builtin_amdgcn_ubfe(rec.y, pz ? 8u : 0u, 8u)) - 128;
        }
        const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
        texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
        const vec3 ev = (texit * dir + org - gmin) * ginv;
        const int nx = texit == tcell.x ? cx + (px ? 0 : -1) : int(ev.x);
        const int ny = texit == tcell.y ? cy + (py ? 0 : -1) : int(ev.y);
        const int nz = texit == tcell.z ? cz + (pz ? 0 : -1) : int(ev.z);
        vx = med3_i32(nx, vx, px ? 0x7fffffff : int(0x80000000));
        vy = med3_i32(ny, vy, py ? 0x7fffffff : int(0x80000000));
        vz = med3_i32(nz, vz, pz ? 0x7fffffff : int(0x80000000));
        outside = (uint32_t(vx) >= uint32_t(a.dims_x)) | (uint32_t(vy) >= uint32_t(a.dims_y)) | (uint32_t(vz) >= uint32_t(a.dims_z));
        uint4 next = make_uint4(0u, 0u, 0u, 0u);                  // a ray that left the grid requests nothing
        if (!outside) next = load_record(vx, vy, vz);
        return next;
    };
    // The list of the cell `rec` describes, tested front to back by this lane alone (the plain loop of traverse_kernel_img).
    auto test_list = [&](const uint4& rec) {
        const bool by_index = field(rec, LAST, SLIM) == uint32_t(NONE - 1);
        int ref = int(field(rec, 48, SLIM));
        uint32_t q1 = NI > 1 ? field(rec, 48 + SLIM, SLIM) : uint32_t(NONE), q2 = NI > 2 ? field(rec, 48 + 2 * SLIM, SLIM) : uint32_t(NONE),
                 q3 = NI > 3 ? field(rec, 48 + 3 * SLIM, SLIM) : uint32_t(NONE);
        if (__ballot(by_index) == 0ull) {
#pragma unroll 1
            while (ref != NONE) {
                Hit h(hit_id, hit_t, 0.0f, 0.0f);
                (void)intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit_t), ref, h);
                hit_t = h.t; hit_id = h.id;
                ref = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE);
            }
        } else {
            if (by_index) {
                q1 = field(rec, 48, 32); q2 = q1 + field(rec, 80, 20);
                ref = NONE;
                if (q1 < q2) ref = ref_at(q1);
                q1++;
            }
#pragma unroll 1
            while (ref != NONE) {
                int next;
                if (by_index) { next = q1 < q2 ? ref_at(q1) : NONE; q1++; }
                else { next = int(q1); q1 = q2; q2 = q3; q3 = uint32_t(NONE); }
                Hit h(hit_id, hit_t, 0.0f, 0.0f);
                (void)intersect_prim_ray(tri_for(ref), Ray(org, tmin, dir, hit_t), ref, h);
                hit_t = h.t; hit_id = h.id;
                ref = next;
            }
        }
    };

    uint4 ca = make_uint4(0u, 0u, 0u, 0u);
    if (alive) ca = load_record(vx, vy, vz);
    unsigned long long live = __ballot(alive);

    if (quad_start) pending = valid && sub == 0;           // (the four lanes of a group hold the same ray: one of them stores its hit)
    else {
    // ---- phase 1: one ray per lane, while the wavefront holds more than kTailRays live rays -------------------------------
    {
        const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
        while (__popcll(live) > kTailRays) {
            if (alive) {
                const uint4 na = cell_step(ca, inv_dir);
                test_list(ca);
                if (hit_t <= texit || outside) alive = false;
                ca = na;
            }
            live = __ballot(alive);
        }
    }
    if (live == 0ull) {
        if (pending) nt_store4(a.hits + id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
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
    if (!UNIFORM) { tab_off = uint32_t(pull_i(int(tab_off))); tab_d = uint32_t(pull_i(int(tab_d))); top_idx = pull_i(top_idx); }
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
        const uint32_t m_stride = ax == 0 ? 1u : (ax == 1 ? uint32_t(a.top_x) : uint32_t(a.top_xy)), m_lsh = uint32_t(ax * a.shift);
        int m_v = ax == 0 ? vx : (ax == 1 ? vy : vz);
        auto quad_sum = [&](uint32_t x) -> uint32_t { return x + uint32_t(quad_perm_i<9>(int(x))) + uint32_t(quad_perm_i<82>(int(x))); };
        auto quad_step = [&](const uint4& rec) -> uint4 {
            int c;
            const uint32_t bound = __builtin_amdgcn_ubfe(ax == 2 ? rec.y : rec.x, m_bit, 8u);
            if (UNIFORM) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c) : "v"(m_pos ? 1 : -1), "v"(bound), "v"(m_v));
            else c = (m_v & ~((1 << a.shift) - 1)) + int(bound) - 128;             // table layout: bounds count from the top-level cell's origin
            const float tc = (float(c) * m_cs + m_gmin - m_org) * m_inv;
            texit = detail::fmin2(detail::fmin2(tc, quad_perm_f<9>(tc)), quad_perm_f<82>(tc));
            const float ev = (texit * m_dir + m_org - m_gmin) * m_ginv;
            const int n = texit == tc ? c + (m_pos ? 0 : -1) : int(ev);
            m_v = med3_i32(n, m_v, m_pos ? 0x7fffffff : int(0x80000000));
            const int o = uint32_t(m_v) >= uint32_t(m_dims) ? 1 : 0;
            outside = (o | quad_perm_i<9>(o) | quad_perm_i<82>(o)) != 0;
            const uint32_t v = outside ? 0u : uint32_t(m_v);
            if (UNIFORM) {
                const uint32_t d = uint32_t(a.shift);
                const uint32_t rec_idx = quad_sum((__umul24(v >> d, m_stride) << (3u * d)) + ((v & ((1u << d) - 1u)) << m_lsh));
                return *reinterpret_cast<const uint4*>(a.img_blocks + (rec_idx << 4));
            }
            const uint32_t top = quad_sum(__umul24(v >> uint32_t(a.shift), m_stride));
            if (int(top) != top_idx) {
                const uint2 t = gather32<uint2>(a.img_table, top << 3);
                tab_off = t.x; tab_d = t.y & 3u; top_idx = int(top);
            }
            const uint32_t idx = quad_sum(((v >> (uint32_t(a.shift) - tab_d)) & ((1u << tab_d) - 1u)) << __umul24(uint32_t(ax), tab_d));
            return *reinterpret_cast<const uint4*>(a.img_blocks + ((tab_off + idx) << 4));
        };
        live = __ballot(alive);
        while (live) {
            if (alive) {                                                   // (whole groups: the four lanes of a ray finish together)
                const uint4 na = quad_step(ca);
                const bool by_index = field(ca, LAST, SLIM) == uint32_t(NONE - 1);
                const int i0 = int(field(ca, 48, SLIM)), i1 = NI > 1 ? int(field(ca, 48 + SLIM, SLIM)) : NONE,
                          i2 = NI > 2 ? int(field(ca, 48 + 2 * SLIM, SLIM)) : NONE, i3 = NI > 3 ? int(field(ca, 48 + 3 * SLIM, SLIM)) : NONE;
                const int inl = by_index ? NONE : (sub == 0 ? i0 : (sub == 1 ? i1 : (sub == 2 ? i2 : i3)));
                auto accept = [&](int ok, float t, float ad, int ref) {            // prims.h:284-292 with the tmax of this moment
                    if (ok && ad * hit_t > t) { const float inv_det = 1.0f / ad; hit_t = t * inv_det; hit_id = ref; }
                };
                if (__ballot(by_index) == 0ull) {
                    // the common step: inline lists only, one round
                    TriCand cd; cd.t = 0.0f; cd.abs_det = 0.0f; cd.ok = false;
                    if (inl != NONE) cd = tri_candidate(tri_for(inl), org, dir, tmin);
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
                    const uint32_t li_begin = field(ca, 48, 32), li_count = by_index ? field(ca, 80, 20) : 0u;
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
            }
            live = __ballot(alive);
        }
    }
    if (pending) nt_store4(a.hits + id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
}


// ---- v3: persistent wavefronts, lane refill, vote-scheduled phases -------------------------------------------------
// Profile of v1/v2 on the 1M-ray batch (profiles/): the SIMDs issue ~80 % of the time while only ~19 % of the lanes
// of an issued VALU instruction are live -- the kernel is instruction-issue bound and 4 of 5 lanes idle, because (a) a
// wave lives as long as its longest ray and (b) inside a cell step every lane waits for the lane with the longest
// reference list.  v3 attacks lane utilisation, the thing that costs twice as much on 64-wide waves as on the
// reference's 32-wide warps:
//   * wavefronts are persistent; a lane whose ray is finished takes the next ray from a global cursor (one atomic
//     per refill, issued when at least kRefillAt lanes are free);
//   * a ray is a small state machine -- it wants either a CELL step (voxel-map walk + cell load + exit plane) or ONE
//     TRIANGLE test -- and each iteration the wave votes (ballot + popcount, scalar) and runs the phase most lanes
//     are waiting for, so an issued instruction always has at least half of the ray-carrying lanes live;
//   * the next reference id is fetched one test ahead, next to the triangle loads.
// Every ray performs exactly the operation sequence of v1 / the oracle, so hits are identical.
//
// Why this is the LARGE-batch kernel only: a persistent wave is always full, so every lock-step iteration costs the
// slowest of 64 busy lanes for the whole life of a long ray, whereas a v2 wave thins out and lets its longest ray
// finish at its own pace.  A batch of a few rays per lane ends when its longest rays end and is 2.7x slower here
// (measured: 1M primary rays 0.40 ms with v2, 1.1 ms with any persistent variant -- also with v2's cell step inside
// the persistent loop, with deferred stores, with any refill threshold or chunk size; profiles/dev_r1_variant_sweep.txt).
constexpr int kBands = 8;           // one ray band + cursor per XCD (private L2 each)

template <bool SMALL>
__global__ void __launch_bounds__(64) traverse_kernel_v3(const TraverseArgs a, int* __restrict__ band_cursors, int chunk, int both_phases, int refill_at) {
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 csize(a.cs_x, a.cs_y, a.cs_z), ginv(a.inv_x, a.inv_y, a.inv_z);

    // wave-uniform ray supply: a local pool [pool_next, pool_end) refilled in chunks from the band cursors.
    // Workgroups are dispatched round-robin over the XCDs, so (blockIdx & 7) is the wave's home band: rays of one
    // band -- one slab of the image for primary rays -- stay on one XCD and its L2.  Empty bands are skipped.
    const int band_len = (a.num_rays + kBands - 1) / kBands;
    int band = blockIdx.x & (kBands - 1), bands_left = kBands;
    int pool_next = 0, pool_end = 0;
    bool exhausted = false;

    int ray_id = -1;                // -1: the lane carries no ray
    bool done = false;              // the lane's ray is finished, its result waits in registers for the next refill
    vec3 org(0.0f), dir(0.0f), inv_dir(0.0f);
    float tmin = 0.0f, hit_t = 0.0f, texit = 0.0f;
    int hit_id = -1;
    int vx = 0, vy = 0, vz = 0;
    int ref = -1, cur = 0, end = 0; // pending reference (prefetched id), its index, list end (Cell variant)
    bool outside = false;

    for (;;) {
        const bool has_ray = ray_id >= 0 && !done;
        const bool want_tri = has_ray && ref >= 0;
        const unsigned long long m_free = __ballot(!has_ray);
        const int n_free = __popcll(m_free);
        if (n_free == 64 && exhausted) break;

        // ---- refill (finished lanes first write their results: stores count against vmcnt on gfx9-family
        // hardware, so a store inside the stepping loop would make every following load-wait pay its latency) ---------------------------------------------------------------------------------------------
        if (!exhausted && (n_free >= refill_at)) {
            while (pool_next >= pool_end && !exhausted) {          // fetch a chunk: one atomic per `chunk` rays
                int base = 0;
                if (threadIdx.x == 0) base = atomicAdd(band_cursors + band, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                const int band_end = min((band + 1) * band_len, a.num_rays);
                const int first = band * band_len + base;
                if (first < band_end) { pool_next = first; pool_end = min(first + chunk, band_end); }
                else { band = (band + 1) & (kBands - 1); if (--bands_left == 0) exhausted = true; }
            }
            const int take = min(n_free, pool_end - pool_next);
            if (done) { nt_store4(a.hits + ray_id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f); done = false; ray_id = -1; }
            if (!has_ray) {
                const int rank = __builtin_amdgcn_mbcnt_hi(unsigned(m_free >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m_free), 0));
                if (rank < take) {
                    const int id = pool_next + rank;
                    const float4 r0 = nt_load4(a.rays + 2 * size_t(id)), r1 = nt_load4(a.rays + 2 * size_t(id) + 1);
                    org = vec3(r0.x, r0.y, r0.z); dir = vec3(r1.x, r1.y, r1.z);
                    tmin = r0.w;
                    const float tmax = r1.w;
                    inv_dir = vec3(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
                    const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
                    const vec3 t0 = min(ta, tb), t1 = max(ta, tb);
                    const float tstart = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), tmin);
                    const float tend = detail::fmin2(detail::fmin2(t1.x, detail::fmin2(t1.y, t1.z)), tmax);
                    if (tstart > tend) {
                        nt_store4(a.hits + id, __int_as_float(-1), tmax, 0.0f, 0.0f);     // misses the grid
                    } else {
                        const vec3 fv = (tstart * dir + org - gmin) * ginv;
                        vx = min(max(int(fv.x), 0), a.dims_x - 1);
                        vy = min(max(int(fv.y), 0), a.dims_y - 1);
                        vz = min(max(int(fv.z), 0), a.dims_z - 1);
                        hit_t = tmax; hit_id = -1; ref = -1;
                        ray_id = id;
                    }
                }
            }
            pool_next += take;
            continue;
        }

        // ---- phase vote ---------------------------------------------------------------------------------------------
        // Throughput mode (rays still available): run only the phase most lanes wait for, so issued instructions are
        // well filled.  Tail mode (no rays left to take): every lane advances every iteration -- the batch now ends
        // when its longest ray ends, and that ray must not wait for votes.
        const int n_tri = __popcll(__ballot(want_tri));
        const int n_cell = 64 - n_free - n_tri;
        const bool all = exhausted || both_phases;
        const bool run_cell = n_cell > 0 && (all || n_cell > n_tri);
        const bool run_tri = n_tri > 0 && (all || !run_cell);
        bool finished = false;

        if (run_cell) {
            if (has_ray && !want_tri) {
                uint32_t w = a.entries[(vx >> a.shift) + a.top_x * ((vy >> a.shift) + a.top_y * (vz >> a.shift))];
                int depth = 0;
                while (w & 3u) {
                    const int k = int(w & 3u);
                    depth += k;
                    const int s = a.shift - depth, m = (1 << k) - 1;
                    w = a.entries[(w >> 2) + ((vx >> s) & m) + ((((vy >> s) & m) + (((vz >> s) & m) << k)) << k)];
                }
                const CellBox c = load_cell_box<SMALL>(a.cells, w >> 2);
                const bool px = dir.x >= 0.0f, py = dir.y >= 0.0f, pz = dir.z >= 0.0f;
                const int cx = px ? c.hx : c.lx, cy = py ? c.hy : c.ly, cz = pz ? c.hz : c.lz;
                const vec3 tcell = (vec3(float(cx), float(cy), float(cz)) * csize + gmin - org) * inv_dir;
                texit = detail::fmin2(tcell.x, detail::fmin2(tcell.y, tcell.z));
                const vec3 ev = (texit * dir + org - gmin) * ginv;
                const int nx = texit == tcell.x ? cx + (px ? 0 : -1) : int(ev.x);
                const int ny = texit == tcell.y ? cy + (py ? 0 : -1) : int(ev.y);
                const int nz = texit == tcell.z ? cz + (pz ? 0 : -1) : int(ev.z);
                vx = px ? max(nx, vx) : min(nx, vx);
                vy = py ? max(ny, vy) : min(ny, vy);
                vz = pz ? max(nz, vz) : min(nz, vz);
                outside = (vx < 0) | (vx >= a.dims_x) | (vy < 0) | (vy >= a.dims_y) | (vz < 0) | (vz >= a.dims_z);
                cur = c.begin; end = c.end;
                const bool nonempty = SMALL ? c.begin >= 0 : c.begin < c.end;
                ref = nonempty ? a.refs[c.begin] : -1;
                finished = ref < 0 && (hit_t <= texit || outside);
            }
        }
        if (run_tri) {
            // one test per lane that was waiting for one at the vote (a lane that just finished its cell step and
            // found references starts testing in the next iteration)
            if (want_tri) {
                int next;
                if (SMALL) next = a.refs[cur + 1];
                else next = cur + 1 < end ? a.refs[cur + 1] : -1;
                Hit h(hit_id, hit_t, 0.0f, 0.0f);
                intersect_prim_ray(load_tri(a.tris, ref), Ray(org, tmin, dir, hit_t), ref, h);
                hit_id = h.id; hit_t = h.t;
                cur++;
                ref = next;
                finished = ref < 0 && (hit_t <= texit || outside);
            }
        }
        if (finished) { done = true; ref = -1; }
    }
    if (done) nt_store4(a.hits + ray_id, __int_as_float(hit_id), hit_t, 0.0f, 0.0f);
}



// ---- ray binning (extension; north_star: "ray packets sorted ... to tame divergence") ----------------------------------
// A batch without spatial order (random origins and directions) makes every load of a wavefront touch 64 unrelated cache
// lines.  Measured on MI355X (tools/dev_sort_potential.py): ordering such a batch by a coarse Morton key of the ray's
// position -- 8 x 8 x 8 bins are enough, the direction octant does not matter -- lifts traversal from 1.0 to 2.3-2.6
// Grays/s.  So the device does a counting sort on 512 bins, not a general sort:
//   ray_bin_count   : key = Morton3(entry point of the ray into the grid box, 3 bits per axis); per-workgroup histogram in
//                     LDS, written to table[bin][workgroup]
//   device_scan     : exclusive scan of the table in (bin, workgroup) order = first slot of every (bin, workgroup) run
//   ray_bin_scatter : slot = run start + rank inside the run (LDS atomic), perm[slot] = ray index
// The order inside a bin is irrelevant.  No global atomics; one extra 4-byte word per ray.
constexpr int kBinBits = 3;
constexpr int kBins = 1 << (3 * kBinBits);
constexpr int kBinItems = 16;                       // rays per thread
constexpr int kBinTile = kBlock * kBinItems;        // rays per workgroup

__device__ __forceinline__ uint32_t spread3(uint32_t x) {   // 3 bits -> every third bit
    return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4);
}

__device__ __forceinline__ int ray_bin_key(const TraverseArgs& a, int id) {
    const float4 r0 = a.rays[2 * size_t(id)], r1 = a.rays[2 * size_t(id) + 1];
    const vec3 org(r0.x, r0.y, r0.z), dir(r1.x, r1.y, r1.z);
    const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
    const vec3 t0 = min(ta, tb);
    float ts = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), r0.w);
    if (!(ts == ts) || ts > 3.0e38f || ts < -3.0e38f) ts = 0.0f;
    const vec3 p = (ts * dir + org - gmin) / (gmax - gmin) * float(1 << kBinBits);
    const int m = (1 << kBinBits) - 1;
    const int x = min(max(int(detail::fmin2(detail::fmax2(p.x, 0.0f), float(m))), 0), m);
    const int y = min(max(int(detail::fmin2(detail::fmax2(p.y, 0.0f), float(m))), 0), m);
    const int z = min(max(int(detail::fmin2(detail::fmax2(p.z, 0.0f), float(m))), 0), m);
    return int(spread3(uint32_t(x)) | (spread3(uint32_t(y)) << 1) | (spread3(uint32_t(z)) << 2));
}

// auto mode: `skip_if` (the row length found by detect_ray_rows) > 0 means the batch is image-ordered and is left alone;
// `diff` (64 words) receives the number of neighbouring rays (i, i + 1) whose keys differ -- the coherence estimate
__global__ void __launch_bounds__(kBlock) ray_bin_count(const TraverseArgs a, unsigned short* __restrict__ keys, int* __restrict__ table,
                                                        const int* __restrict__ skip_if, int* __restrict__ diff) {
    __shared__ int hist[kBins];
    __shared__ unsigned short tile_keys[kBinTile];
    __shared__ int lds[kWaves];
    if (skip_if && *skip_if > 0) return;
    for (int i = threadIdx.x; i < kBins; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * kBinTile;
    for (int j = 0; j < kBinItems; j++) {
        const int id = base + j * kBlock + threadIdx.x;
        if (id < a.num_rays) {
            const int k = ray_bin_key(a, id);
            keys[id] = (unsigned short)k;
            if (diff) tile_keys[j * kBlock + threadIdx.x] = (unsigned short)k;
            atomicAdd(&hist[k], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kBlock) table[size_t(i) * gridDim.x + blockIdx.x] = hist[i];
    if (diff) {
        int d = 0;
        for (int j = 0; j < kBinItems; j++) {
            const int i = j * kBlock + threadIdx.x;
            if (base + i + 1 < a.num_rays && i + 1 < kBinTile) d += tile_keys[i] != tile_keys[i + 1];
        }
        d = block_sum(d, lds);
        if (threadIdx.x == 0 && d) atomicAdd(diff + (blockIdx.x & 63), d);
    }
}

// auto mode: bin the batch iff it is not image-ordered and more than half of its neighbouring rays fall into different bins
__global__ void __launch_bounds__(64) ray_bin_decide(const int* __restrict__ row_len, int* __restrict__ diff, int num_rays, int* __restrict__ flag) {
    int d = diff[threadIdx.x];
    diff[threadIdx.x] = 0;                       // ready for the next batch
    d = wave_sum(d);
    if (threadIdx.x == 0) flag[0] = (*row_len == 0 && 2ll * d > num_rays) ? 1 : 0;
}

// The rays of a tile are first put in bin order inside LDS (local histogram -> local scan -> local rank), then written out: lanes
// that are neighbours in LDS write neighbouring words of `perm`, so a store instruction touches the runs of a few bins instead of 64
// unrelated lines (the lane-by-lane form moved 512 MB in 2.3 ms for 128M rays: bound by write transactions, not by bytes).
__global__ void __launch_bounds__(kBlock) ray_bin_scatter(const unsigned short* __restrict__ keys, const int* __restrict__ table_scan,
                                                          int num_rays, int* __restrict__ perm, const int* __restrict__ only_if) {
    static_assert(kBins == 2 * kBlock, "two bins per thread in the local scan");
    __shared__ int count[kBins];               // rays of the tile per bin, then the cursor of the local ranks
    __shared__ int lstart[kBins + 1];          // first LDS slot of every bin
    __shared__ int gstart[kBins];              // first slot of the (bin, workgroup) run in perm
    __shared__ int sorted_id[kBinTile];
    __shared__ unsigned short sorted_key[kBinTile];
    __shared__ int wsum[kWaves];
    if (only_if && *only_if == 0) return;
    for (int i = threadIdx.x; i < kBins; i += kBlock) { count[i] = 0; gstart[i] = table_scan[size_t(i) * gridDim.x + blockIdx.x]; }
    __syncthreads();
    const int base = blockIdx.x * kBinTile;
    int key[kBinItems], rank[kBinItems];
#pragma unroll
    for (int j = 0; j < kBinItems; j++) {
        const int id = base + j * kBlock + threadIdx.x;
        key[j] = id < num_rays ? int(keys[id]) : -1;
    }
#pragma unroll
    for (int j = 0; j < kBinItems; j++) rank[j] = key[j] >= 0 ? atomicAdd(&count[key[j]], 1) : 0;
    __syncthreads();
    {   // exclusive scan of the 512 counts: two per thread, wavefront scan, wavefront sums through LDS
        const int c0 = count[2 * threadIdx.x], c1 = count[2 * threadIdx.x + 1];
        const int incl = wave_inclusive_scan(c0 + c1);
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave_id(); w++) before += wsum[w];
        const int ex = before + incl - (c0 + c1);
        lstart[2 * threadIdx.x] = ex; lstart[2 * threadIdx.x + 1] = ex + c0;
        if (threadIdx.x == kBlock - 1) lstart[kBins] = ex + c0 + c1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBinItems; j++)
        if (key[j] >= 0) {
            const int at = lstart[key[j]] + rank[j];
            sorted_id[at] = base + j * kBlock + threadIdx.x;
            sorted_key[at] = (unsigned short)key[j];
        }
    __syncthreads();
    const int total = lstart[kBins];
    for (int i = threadIdx.x; i < total; i += kBlock) {
        const int k = sorted_key[i];
        perm[gstart[k] + (i - lstart[k])] = sorted_id[i];
    }
}

struct TableIn { const int* t; __device__ int operator()(int i) const { return t[i]; } };
struct TableOut { int* t; __device__ void operator()(int i, int s) const { t[i] = s; } };

template <bool SMALL, bool NARROW>
void launch_v2_mode(hipStream_t st, int blocks, unsigned mode, const TraverseArgs& a) {
    switch (mode & 3u) {
        case 0: traverse_kernel_v2<SMALL, 64, NARROW, 0><<<blocks, 64, 0, st>>>(a); break;
        case 1: traverse_kernel_v2<SMALL, 64, NARROW, 1><<<blocks, 64, 0, st>>>(a); break;
        case 2: traverse_kernel_v2<SMALL, 64, NARROW, 2><<<blocks, 64, 0, st>>>(a); break;
        default: traverse_kernel_v2<SMALL, 64, NARROW, 3><<<blocks, 64, 0, st>>>(a); break;
    }
}
void launch_v2(hipStream_t st, int blocks, bool small, bool narrow, unsigned mode, const TraverseArgs& a) {
    if (small) { if (narrow) launch_v2_mode<true, true>(st, blocks, mode, a); else launch_v2_mode<true, false>(st, blocks, mode, a); }
    else       { if (narrow) launch_v2_mode<false, true>(st, blocks, mode, a); else launch_v2_mode<false, false>(st, blocks, mode, a); }
}

// bytes from p to the end of the device allocation that holds it (all bits set if the runtime does not know the pointer)
size_t buffer_bytes_from(const void* p) {
    hipDeviceptr_t base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) == hipSuccess) return size - size_t(static_cast<const char*>(p) - static_cast<const char*>(base));
    (void)hipGetLastError();
    return ~size_t(0);
}

// the image kernel: plain traversal for every layout, the any-hit / barycentric variants for the flat narrow layouts
template <unsigned MODE>
bool launch_img_mode(hipStream_t st, int blocks, bool flat, bool narrow, bool uniform, int slim, bool tail, const TraverseArgs& a) {
    if (slim && !(flat && narrow && (slim == 20 || slim == 26))) return false;          // slim records are read by the flat narrow kernels only
    if (tail && MODE == 0 && slim && !a.wave_times && !uniform) {
        if (slim == 20) traverse_kernel_tail<20, false, false><<<blocks, 64, a.lds_pad, st>>>(a);
        else            traverse_kernel_tail<26, false, false><<<blocks, 64, a.lds_pad, st>>>(a);
    }
    else if (tail && MODE == 0 && uniform && slim && !(a.wave_times && slim != 20)) {
        if (slim == 20 && a.wave_times) traverse_kernel_tail<20, true><<<blocks, 64, 0, st>>>(a);
        else if (slim == 20) traverse_kernel_tail<20><<<blocks, 64, a.lds_pad, st>>>(a);
        else                 traverse_kernel_tail<26><<<blocks, 64, a.lds_pad, st>>>(a);
    }
    else if (slim == 20 && uniform) {
        if (MODE == 0 && a.wave_times) traverse_kernel_img<64, true, true, true, 0, true, 20><<<blocks, 64, 0, st>>>(a);
        else traverse_kernel_img<64, true, true, true, MODE, false, 20><<<blocks, 64, 0, st>>>(a);
    }
    else if (slim == 26 && uniform) traverse_kernel_img<64, true, true, true, MODE, false, 26><<<blocks, 64, 0, st>>>(a);
    else if (slim == 20)            traverse_kernel_img<64, true, true, false, MODE, false, 20><<<blocks, 64, 0, st>>>(a);
    else if (slim == 26)            traverse_kernel_img<64, true, true, false, MODE, false, 26><<<blocks, 64, 0, st>>>(a);
    else if (flat && narrow && uniform && MODE == 0 && a.wave_times) traverse_kernel_img<64, true, true, true, 0, true><<<blocks, 64, 0, st>>>(a);
    else if (flat && narrow && uniform) traverse_kernel_img<64, true, true, true, MODE><<<blocks, 64, 0, st>>>(a);
    else if (flat && narrow)       traverse_kernel_img<64, true, true, false, MODE><<<blocks, 64, 0, st>>>(a);
    else if (MODE != 0)            return false;
    else if (flat)                 traverse_kernel_img<64, true, false, false, 0><<<blocks, 64, 0, st>>>(a);
    else if (narrow)               traverse_kernel_img<64, false, true, false, 0><<<blocks, 64, 0, st>>>(a);
    else                           traverse_kernel_img<64, false, false, false, 0><<<blocks, 64, 0, st>>>(a);
    return true;
}
bool launch_img(hipStream_t st, int blocks, bool flat, bool narrow, bool uniform, int slim, bool tail, unsigned mode, const TraverseArgs& a) {
    switch (mode & 3u) {
        case 0: return launch_img_mode<0>(st, blocks, flat, narrow, uniform, slim, tail, a);
        case 1: return launch_img_mode<1>(st, blocks, flat, narrow, uniform, slim, tail, a);
        case 2: return launch_img_mode<2>(st, blocks, flat, narrow, uniform, slim, tail, a);
        default: return launch_img_mode<3>(st, blocks, flat, narrow, uniform, slim, tail, a);
    }
}

// row length of an image-ordered batch -> row_len[0] on the device.  The origin criterion costs ~17 us (2049 candidates x 256
// sampled pairs) and only pays where tile packets pay for bounce rays: it runs as a second launch for batches of at least
// kOriginMinRays rays and returns at once when the first criterion has already answered.
constexpr int kOriginMinRays = 1 << 22;
void launch_detect(hagrid_ctx* ctx, const TraverseArgs& a, int num_rays, int* row_len, int origin_min_rays = kOriginMinRays) {
    detect_ray_rows<<<1, kDetectBlock, 0, ctx->stream>>>(a.rays, num_rays, row_len, nullptr, 0.0f, 0); HG_DBG(ctx);
    const vec3 ext(a.max_x - a.min_x, a.max_y - a.min_y, a.max_z - a.min_z);
    const float tau = length(ext) / 64.0f;
    if (num_rays < origin_min_rays || !(tau > 0.0f) || !(tau < 3.0e18f)) return;
    if (!ctx->row_scores) {
        if (hipMalloc((void**)&ctx->row_scores, (kRowCandidates + 8) * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); ctx->row_scores = nullptr; return; }
        (void)hipMemsetAsync(ctx->row_scores, 0, (kRowCandidates + 8) * sizeof(int), ctx->stream);
    }
    detect_ray_rows<<<1 + (kRowCandidates + 1 + 3) / 4, kDetectBlock, 0, ctx->stream>>>(a.rays, num_rays, row_len, ctx->row_scores, 1.0f / (tau * tau), 1); HG_DBG(ctx);
}

int make_args(hagrid_ctx* ctx, const hagrid_grid* g, const void* tris, const void* rays, void* hits, int num_rays, TraverseArgs& a) {
    const bool released = g && ctx->image.detached && trav_image_matches(ctx, g);     // hagrid_grid_release_for_traversal: the image stands for entries and cells
    if (!g || !g->ref_ids || (!released && (!g->entries || (!g->cells && !g->small_cells)))) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: incomplete grid");
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
    a.steps = nullptr; a.stats = nullptr; a.perm = nullptr; a.perm_flag = nullptr; a.wave_times = nullptr; a.tile_order = nullptr;
    a.row_len = nullptr; a.row_len_hint = 0; a.super_log2 = ctx->opt_super_log2; a.xcd_chunk_log2 = ctx->opt_xcd_chunk_log2;
    a.img_table = nullptr; a.img_blocks = nullptr;
    a.num_rays = num_rays; a.shift = g->shift; a.id_is_steps = 0; a.quad_first_block = 0x7fffffff; a.lds_pad = ctx->opt_lds_pad;
    a.dims_x = dims.x; a.dims_y = dims.y; a.dims_z = dims.z;
    a.top_x = g->dims[0]; a.top_y = g->dims[1];
    a.top_xy = (long long)g->dims[0] * g->dims[1] < (1 << 23) ? g->dims[0] * g->dims[1] : 0;
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
    HG_TRY(make_args(ctx, grid, nullptr, nullptr, nullptr, 0, a));
    if (ctx->image.detached && trav_image_matches(ctx, grid)) return HAGRID_OK;      // a released grid: its image is all there is
    return trav_image_build(ctx, grid);     // "traverse.image" = 0: drops the image, traversal reads the construction format
}

extern "C" int hagrid_traverse_grid(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                                    const void* rays, void* hits, int num_rays) {
    return hagrid_traverse_grid_ex(ctx, grid, tris, rays, hits, num_rays, 0u);
}

extern "C" int hagrid_traverse_grid_ex(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                                       const void* rays, void* hits, int num_rays, uint32_t flags) {
    if (!ctx) return HAGRID_EINVAL;
    if (flags & ~uint32_t(HAGRID_TRAVERSE_ANY_HIT | HAGRID_TRAVERSE_UVS)) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid_ex: unknown flag");
    TraverseArgs a;
    if (trav_image_stale(ctx))
        HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: the traversal image this context borrows (hagrid_share_traversal) was dropped by its owner; renew the share or call hagrid_setup_traversal here");
    HG_TRY(make_args(ctx, grid, tris, rays, hits, num_rays, a));
    if (num_rays == 0) return HAGRID_OK;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->opt_id_is_steps && flags == 0) {
        // literal compatibility with the reference BINARY: its kernel overwrites Hit.id with the traversal step counter
        // (traverse.cu:80,93) and its viewer colours by it (main.cpp:100-107).  Served by the reference-shaped kernel.
        if (!grid->entries) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: traverse.id_is_steps needs the construction format (grid released for traversal)");
        a.id_is_steps = 1;
        const int blocks = grid_blocks(num_rays, 256);
        if (grid->small_cells) traverse_kernel<true, true><<<blocks, 256, 0, ctx->stream>>>(a);
        else                   traverse_kernel<false, true><<<blocks, 256, 0, ctx->stream>>>(a);
        HG_DBG(ctx);
        HG_HIP(ctx, hipGetLastError());
        return HAGRID_OK;
    }
    // binning buffers: released on every exit path.  The pool hands them to nobody else before the kernels below are done: in
    // keep mode free() only marks the slot (work of one context is stream-ordered), otherwise free() synchronises the stream first
    PoolTemps tmp(ctx);
    bool publish_row_len = false;
    int* perm = nullptr;
    if (ctx->ray_binning && num_rays > kBinTile) {
        const int tiles = grid_blocks(num_rays, kBinTile);
        const int table_n = kBins * tiles;
        perm = tmp.get<int>(size_t(num_rays));
        unsigned short* bin_keys = tmp.get<unsigned short>(size_t(num_rays));
        int* bin_table = tmp.get<int>(size_t(table_n));
        int* bin_partials = tmp.get<int>(size_t(scan_num_tiles(table_n)) + 1);
        if (!perm || !bin_keys || !bin_table || !bin_partials) return HAGRID_ENOMEM;
        if (ctx->ray_binning == 2) {
            // automatic: everything is decided on the device, nobody waits.  row length (image-ordered batches are left to the
            // tile packets) -> keys + coherence estimate -> scan -> decision -> scatter; the traversal kernel reads the decision.
            int* row_len = ctx->dscratch + 232;
            int* flag = ctx->dscratch + 233;
            if (!ctx->bin_diff) {
                HG_HIP(ctx, hipMalloc((void**)&ctx->bin_diff, 64 * sizeof(int)));
                HG_HIP(ctx, hipMemsetAsync(ctx->bin_diff, 0, 64 * sizeof(int), ctx->stream));
            }
            launch_detect(ctx, a, ctx->opt_image_width >= 0 ? num_rays : 0, row_len);
            ray_bin_count<<<tiles, kBlock, 0, ctx->stream>>>(a, bin_keys, bin_table, row_len, ctx->bin_diff); HG_DBG(ctx);
            (void)ctx_scan<int>(ctx, TableIn{bin_table}, TableOut{bin_table}, table_n, bin_partials, (const int*)nullptr, (int*)nullptr);
            ray_bin_decide<<<1, 64, 0, ctx->stream>>>(row_len, ctx->bin_diff, num_rays, flag); HG_DBG(ctx);
            ray_bin_scatter<<<tiles, kBlock, 0, ctx->stream>>>(bin_keys, bin_table, num_rays, perm, flag); HG_DBG(ctx);
            a.perm_flag = flag;
            if (ctx->opt_image_width == 0) a.row_len = row_len;
            else if (ctx->opt_image_width > 0) a.row_len_hint = ctx->opt_image_width;
        } else {
            ray_bin_count<<<tiles, kBlock, 0, ctx->stream>>>(a, bin_keys, bin_table, nullptr, nullptr); HG_DBG(ctx);
            (void)ctx_scan<int>(ctx, TableIn{bin_table}, TableOut{bin_table}, table_n, bin_partials, (const int*)nullptr, (int*)nullptr);
            ray_bin_scatter<<<tiles, kBlock, 0, ctx->stream>>>(bin_keys, bin_table, num_rays, perm, nullptr); HG_DBG(ctx);
        }
        a.perm = perm;
    }
    // Kernel choice.  With a traversal image (hagrid_setup_traversal built one for this very grid) its kernel is used for
    // every batch.  Without one: small batches (a few rays per resident lane) end when their longest rays end and the
    // latency-oriented v2 wins; large batches are throughput-bound and the persistent, vote-scheduled v3 wins (measured
    // crossover on MI355X between 8M and 16M primary rays, i.e. ~24 rays per lane of a full machine).
    // hagrid_set_option("traverse.variant", 1|2|3|4) forces a kernel (tests, experiments).
    const bool have_image = (ctx->opt_image || ctx->image.detached) && trav_image_matches(ctx, grid);
    if (ctx->image.detached && have_image && ((ctx->opt_variant && ctx->opt_variant != 4) || (flags && !ctx->image.flat)))
        HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: this grid was released for traversal, only the traversal-image kernel can serve it");
    if (ctx->opt_variant == 4 && !have_image) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: no traversal image for this grid (hagrid_setup_traversal)");
    const long long lanes = (long long)ctx->num_cus * 32 * 64;
    const bool large = num_rays >= 24 * lanes;
    int variant = ctx->opt_variant ? ctx->opt_variant : (have_image ? 4 : (large ? 3 : 2));
    if (perm && variant != 4) variant = 2;            // binned batches: the latency-oriented kernel wins at every size measured
    const bool img_narrow = have_image && ctx->opt_narrow && a.top_xy > 0 && grid->dims[2] < (1 << 23) && buffer_bytes_from(tris) < (size_t(1) << 32) &&
                            ctx->image.block_bytes < (size_t(1) << 32) && size_t(grid->num_cells) * 32 < (size_t(1) << 32) &&
                            size_t(grid->num_entries) * 4 < (size_t(1) << 32) && size_t(grid->num_refs) * 4 < (size_t(1) << 32);
    // any-hit / barycentrics: the flat narrow image kernels and v2 have these variants
    if (flags && !(variant == 4 && ctx->image.flat && img_narrow)) {
        if (ctx->image.detached && have_image) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: any-hit / barycentrics on a released grid need the narrow image kernel (arrays below 4 GB)");
        variant = 2;
    }
    if (variant == 4 && ctx->image.slim && !img_narrow) {
        // slim records are read by the narrow kernels only (arrays of 4 GB and more, "traverse.narrow" = 0): construction format
        if (ctx->image.detached) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: this grid was released for traversal and its slim traversal image needs the narrow kernel (arrays below 4 GB)");
        variant = 2;
    }
    if (variant == 4) {
        a.img_table = static_cast<const uint2*>(ctx->image.table);
        a.img_blocks = static_cast<const unsigned char*>(ctx->image.blocks);
    }
    // Tile packets (v2 and the image kernel, not for binned batches): "traverse.image_width" > 0 gives the row length, 0
    // (default) looks for one on the device, -1 switches the feature off.  The kernel reads the answer from device memory,
    // nobody waits for it -- except a large batch without a traversal image: image order + tiles + v2 beats v3
    // (4096^2 rays: 1.96 vs 2.55 ms), so there the kernel is chosen on the host after reading the row length back.
    if (!perm && ctx->opt_image_width >= 0 && (variant == 2 || variant == 4 || (!ctx->opt_variant && large))) {
        if (ctx->opt_image_width > 0) {
            a.row_len_hint = ctx->opt_image_width;
            if (variant == 3 && (a.row_len_hint & 7) == 0 && num_rays / a.row_len_hint >= 8) variant = 2;
        } else {
            // The row length only steers the lane <-> ray assignment (any value gives the same hits), so a row length FOUND for a ray
            // buffer is kept: calls with the same buffer and count reuse it and look again every 16th call ("traverse.row_cache" = 0:
            // at every call).  The host learns the answer without waiting (a 4-byte copy behind the traversal launch + an event it only
            // polls) and never keeps "not image-ordered" once it has seen it, so a buffer that alternates between unordered rays and an
            // image looks every time.  A buffer refilled with rows of another length runs on the stale length for at most 15 calls --
            // slower, never wrong.
            int* row_len = ctx->dscratch + 236;
            const bool same = ctx->opt_row_cache && variant != 3 && ctx->rowlen_rays == rays && ctx->rowlen_n == num_rays;
            if (same && ctx->rowlen_pending) {
                if (hipEventQuery(ctx->rowlen_evt) == hipSuccess) { ctx->rowlen_known = ctx->mailbox[300]; ctx->rowlen_pending = false; }
                else (void)hipGetLastError();                             // not ready yet: not an error
            }
            if (same && ctx->rowlen_known != 0 && ctx->rowlen_age < 15) ctx->rowlen_age++;
            else {
                launch_detect(ctx, a, num_rays, row_len);
                ctx->rowlen_rays = rays; ctx->rowlen_n = num_rays; ctx->rowlen_age = 0; ctx->rowlen_known = -1;
                publish_row_len = ctx->opt_row_cache && variant != 3;
            }
            a.row_len = row_len;
            if (variant == 3) {
                int w = 0;
                HG_TRY(read_back(ctx, row_len, &w, sizeof(int)));
                if (w > 0) variant = 2;
            }
        }
    }
    if (variant == 4) {
        int blocks = grid_blocks(num_rays, 64);
        const bool narrow = img_narrow;
        // Tail kernel: "traverse.quad_tail" per cent of the tiles, the last in dispatch order, start with four lanes per ray.  -1 (default):
        // a quarter of the tiles when the launch is between one and two rounds of the resident wavefronts (1024^2 rays on 256 CUs) --
        // there the last wavefronts to start are what the launch waits for (1024^2: 0.179 -> 0.175 ms; 6 / 12 / 37 / 50 %: 0.180 /
        // 0.181 / 0.189 / 0.204 ms).  Larger launches are throughput-bound and lose (2048^2 at 12 %: +3 %, 4096^2: +10 %), binned
        // batches too; an unordered 1M batch gains up to 20 % at 100 (profiles/dev_r3_quad_tail.txt).  Hits do not depend on it.
        int quad_pct = ctx->opt_quad_tail;
        if (quad_pct < 0) {
            const long long slots = (long long)ctx->num_cus * 32;
            quad_pct = (!perm && blocks > slots && 4ll * blocks <= 9 * slots) ? 25 : 0;
        }
        if (quad_pct > 0 && ctx->opt_tail && !flags && ctx->image.slim && ctx->image.flat && narrow) {
            const int chunk = 8 << (a.xcd_chunk_log2 >= 0 ? a.xcd_chunk_log2 : 4);
            const int full = std::min(blocks, int((long long)blocks * (100 - quad_pct) / 100 + chunk - 1) / chunk * chunk);
            if (full < blocks) { a.quad_first_block = full; blocks = full + 4 * (blocks - full); }
        }
        a.wave_times = ctx->kat_wave_times; a.tile_order = ctx->kat_tile_order;
        if (!launch_img(ctx->stream, blocks, ctx->image.flat, narrow, ctx->image.flat && ctx->image.uniform && narrow, ctx->image.slim, ctx->opt_tail != 0, flags, a))
            HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: the traversal image of this grid has no kernel for this call (slim records need arrays below 4 GB)");
    } else if (variant == 1) {
        const int blocks = grid_blocks(num_rays, 256);
        if (grid->small_cells) traverse_kernel<true, false><<<blocks, 256, 0, ctx->stream>>>(a);
        else                   traverse_kernel<false, false><<<blocks, 256, 0, ctx->stream>>>(a);
    } else if (variant == 2) {
        const int blocks = grid_blocks(num_rays, 64);
        // 32-bit offsets are enough when every gathered array is smaller than 4 GB
        const size_t tri_bytes = buffer_bytes_from(tris);
        const bool narrow = ctx->opt_narrow && tri_bytes < (size_t(1) << 32) && size_t(grid->num_cells) * 32 < (size_t(1) << 32) &&
                            a.top_xy > 0 && grid->dims[2] < (1 << 23) &&
                            size_t(grid->num_entries) * 4 < (size_t(1) << 32) && size_t(grid->num_refs) * 4 < (size_t(1) << 32);
        launch_v2(ctx->stream, blocks, grid->small_cells != nullptr, narrow, flags, a);
    } else {
        const int blocks = std::min(grid_blocks(num_rays, 64), ctx->num_cus * ctx->opt_waves_per_cu);
        // rays per cursor atomic: a few chunks per wave for balance, at least one wave-load, at most 1024
        int chunk = ctx->opt_chunk ? ctx->opt_chunk : (num_rays / (blocks * 2));
        chunk = std::max(64, std::min(1024, (chunk + 63) & ~63));
        const int both = ctx->opt_both_phases, refill_at = ctx->opt_refill_at;
        int* cursors = ctx->dscratch + 240;                  // 8 band cursors
        HG_HIP(ctx, hipMemsetAsync(cursors, 0, 8 * sizeof(int), ctx->stream));
        if (grid->small_cells) traverse_kernel_v3<true><<<blocks, 64, 0, ctx->stream>>>(a, cursors, chunk, both, refill_at);
        else                   traverse_kernel_v3<false><<<blocks, 64, 0, ctx->stream>>>(a, cursors, chunk, both, refill_at);
    }
    HG_DBG(ctx);                                   // the traversal kernel launched by one of the helpers above
    HG_HIP(ctx, hipGetLastError());
    if (publish_row_len) {                         // behind the traversal launch: nobody waits for it
        if (!ctx->rowlen_evt) HG_HIP(ctx, hipEventCreateWithFlags(&ctx->rowlen_evt, hipEventDisableTiming));
        HG_HIP(ctx, hipMemcpyAsync(ctx->mailbox + 300, ctx->dscratch + 236, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HG_HIP(ctx, hipEventRecord(ctx->rowlen_evt, ctx->stream));
        ctx->rowlen_pending = true;
    }
    return HAGRID_OK;
}

extern "C" int hagrid_set_option(hagrid_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return HAGRID_EINVAL;
    struct { const char* name; int* dst; int lo, hi; } table[] = {
        {"traverse.variant", &ctx->opt_variant, 0, 4},        {"traverse.waves_per_cu", &ctx->opt_waves_per_cu, 1, 32},
        {"traverse.chunk", &ctx->opt_chunk, 0, 1 << 20},      {"traverse.both_phases", &ctx->opt_both_phases, 0, 1},
        {"traverse.refill_at", &ctx->opt_refill_at, 1, 64},   {"expand.subset_only", &ctx->opt_expand_subset_only, 0, 1},
        {"traverse.image_width", &ctx->opt_image_width, -1, 1 << 24}, {"traverse.super_tile", &ctx->opt_super_log2, 0, 8},
        {"traverse.xcd_chunk", &ctx->opt_xcd_chunk_log2, -1, 16}, {"traverse.image", &ctx->opt_image, 0, 2},             {"traverse.image_uniform", &ctx->opt_image_uniform, 0, 2}, {"traverse.image_slim", &ctx->opt_image_slim, 0, 2}, {"traverse.tail", &ctx->opt_tail, 0, 1}, {"traverse.quad_tail", &ctx->opt_quad_tail, -1, 100}, {"traverse.lds_pad", &ctx->opt_lds_pad, 0, 65536}, {"traverse.row_cache", &ctx->opt_row_cache, 0, 1},
        {"traverse.image_max_mb", &ctx->opt_image_max_mb, 0, 1 << 20},
        {"traverse.narrow", &ctx->opt_narrow, 0, 1},
        {"traverse.id_is_steps", &ctx->opt_id_is_steps, 0, 1},
        {"merge.narrow_cells", &ctx->opt_merge_narrow, 0, 1},

    };
    for (auto& t : table)
        if (!strcmp(key, t.name)) {
            if (value < t.lo || value > t.hi) HG_FAIL(ctx, HAGRID_EINVAL, "set_option: value out of range");
            *t.dst = value;
            return HAGRID_OK;
        }
    HG_FAIL(ctx, HAGRID_EINVAL, "set_option: unknown key");
}

extern "C" int hagrid_set_ray_binning(hagrid_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return HAGRID_EINVAL;
    ctx->ray_binning = mode;
    return HAGRID_OK;
}

extern "C" int hagrid_traverse_grid_stats(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                                          const void* rays, void* hits, int num_rays,
                                          void* steps, hagrid_traversal_stats* stats) {
    if (!ctx) return HAGRID_EINVAL;
    if (grid && !grid->entries) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid_stats: the statistics walk the construction format (grid released for traversal)");
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, tris, rays, hits, num_rays, a));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (num_rays == 0) return HAGRID_OK;
    HG_HIP(ctx, hipSetDevice(ctx->device));
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
        stats->hits = (int64_t)h[6]; stats->long_list_refs = (int64_t)h[7];
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

__global__ void kat_image_records(TraverseArgs a, const int* vox, int n, uint32_t* out, int flat, int slim, int slim_uniform) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int vx = vox[3 * i], vy = vox[3 * i + 1], vz = vox[3 * i + 2];
    const uint2 tab = a.img_table[(vx >> a.shift) + a.top_x * ((vy >> a.shift) + a.top_y * (vz >> a.shift))];
    if (slim) {     // a slim record, brought into the form of the 32-byte record
        const int d = int(tab.y & 3u), sh = a.shift - d, m = (1 << d) - 1;             // (uniform layout: d == shift)
        const uint4 r = reinterpret_cast<const uint4*>(a.img_blocks)[size_t(tab.x) + size_t(((vx >> sh) & m) + ((((vy >> sh) & m) + (((vz >> sh) & m) << d)) << d))];
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
        } else {
            const int om = ~((1 << a.shift) - 1);
            o[0] = uint32_t((vx & om) + int(field(0, 8)) - 128) | uint32_t((vx & om) + int(field(8, 8)) - 128) << 16;
            o[1] = uint32_t((vy & om) + int(field(16, 8)) - 128) | uint32_t((vy & om) + int(field(24, 8)) - 128) << 16;
            o[2] = uint32_t((vz & om) + int(field(32, 8)) - 128) | uint32_t((vz & om) + int(field(40, 8)) - 128) << 16;
        }
        if (field(48 + (ni - 1) * slim, slim) == none - 1u) {
            const uint32_t cnt = field(80, 20);
            // lists of at most four ids are inline in the 32-byte record: read them through the index
            o[3] = cnt | (cnt > 4 ? 0x80000000u : 0u);
            if (cnt > 4) { o[4] = field(48, 32); o[5] = o[6] = o[7] = 0u; }
            else for (uint32_t j = 0; j < 4; j++) o[4 + j] = j < cnt ? uint32_t(a.refs[field(48, 32) + j]) : ~0u;
        } else {
            uint32_t cnt = 0;
            for (int j = 0; j < 4; j++) {
                const uint32_t id = j < ni ? field(48 + j * slim, slim) : none;
                o[4 + j] = id == none ? ~0u : id;
                if (id != none) cnt++;
            }
            o[3] = cnt;
        }
        return;
    }
    const uint4* rec = flat ? image_record<true>(a, tab, vx, vy, vz) : image_record<false>(a, tab, vx, vy, vz);
    uint4 ra = rec[0], rb = rec[1];
    if (ra.w >= 0xfffffffeu) { uint32_t off, meta; image_resolve_links(a, vx, vy, vz, ra, rb, off, meta); ra.w |= 0x40000000u; }     // bit 30: came through a nested block or a deep link
    uint32_t* o = out + 8 * size_t(i);
    o[0] = ra.x; o[1] = ra.y; o[2] = ra.z; o[3] = ra.w; o[4] = rb.x; o[5] = rb.y; o[6] = rb.z; o[7] = rb.w;
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
    return read_back(ctx, d, row_len, sizeof(int));
}

extern "C" int hagrid_kat_tile_slots(hagrid_ctx* ctx, int num_rays, int row_len, int super_log2, int xcd_chunk_log2, int32_t* slots) {
    if (!ctx || num_rays <= 0 || !slots || super_log2 < 0 || super_log2 > 8) return HAGRID_EINVAL;
    const int blocks = grid_blocks(num_rays, 64);
    TraverseArgs a;
    memset(&a, 0, sizeof(a));
    a.num_rays = num_rays; a.row_len_hint = row_len; a.super_log2 = super_log2; a.xcd_chunk_log2 = xcd_chunk_log2;
    Staged o(ctx, nullptr, size_t(blocks) * 64 * 4);
    if (!o.d) return HAGRID_ENOMEM;
    kat_tile_slots<<<blocks, 64, 0, ctx->stream>>>(a, (int*)o.d); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(slots);
}

extern "C" int hagrid_kat_wave_times(hagrid_ctx* ctx, unsigned long long* times_dev, const int* tile_order_dev) {
    if (!ctx) return HAGRID_EINVAL;
    ctx->kat_wave_times = times_dev;
    ctx->kat_tile_order = tile_order_dev;
    return HAGRID_OK;
}

extern "C" int hagrid_kat_image_format(hagrid_ctx* ctx, const hagrid_grid* grid, int32_t* format4) {
    if (!ctx || !grid || !format4) return HAGRID_EINVAL;
    if (!trav_image_matches(ctx, grid)) HG_FAIL(ctx, HAGRID_EINVAL, "no traversal image for this grid");
    format4[0] = ctx->image.flat ? 1 : 0; format4[1] = ctx->image.uniform ? 1 : 0; format4[2] = ctx->image.slim; format4[3] = ctx->image.slim ? 16 : 32;
    return HAGRID_OK;
}

extern "C" int hagrid_kat_image_records(hagrid_ctx* ctx, const hagrid_grid* grid, const int32_t* voxels3, int n, uint32_t* records8, int64_t* image_bytes) {
    if (!ctx || !grid || n < 0) return HAGRID_EINVAL;
    if (!trav_image_matches(ctx, grid)) HG_FAIL(ctx, HAGRID_EINVAL, "no traversal image for this grid");
    if (image_bytes) *image_bytes = (int64_t)ctx->image.block_bytes + 8ll * grid->dims[0] * grid->dims[1] * grid->dims[2];
    if (n == 0) return HAGRID_OK;
    TraverseArgs a;
    HG_TRY(make_args(ctx, grid, nullptr, nullptr, nullptr, 0, a));
    a.img_table = static_cast<const uint2*>(ctx->image.table);
    a.img_blocks = static_cast<const unsigned char*>(ctx->image.blocks);
    // staging must not disturb the image: these buffers are not grid arrays
    Staged v(ctx, voxels3, size_t(n) * 12), o(ctx, nullptr, size_t(n) * 32);
    if (!v.d || !o.d) return HAGRID_ENOMEM;
    kat_image_records<<<grid_blocks(n, 64), 64, 0, ctx->stream>>>(a, (const int*)v.d, n, (uint32_t*)o.d, ctx->image.flat ? 1 : 0, ctx->image.slim, ctx->image.uniform ? 1 : 0); HG_DBG(ctx);
    HG_HIP(ctx, hipGetLastError());
    return o.fetch(records8);
}

// trav_common.h -- what the traversal translation units share: the kernel argument block, record / triangle loads, the tile
// packets (lane <-> ray assignment), small device helpers, and the host entry points each unit offers the others.
//   traverse.hip    the C ABI (setup_traversal, traverse_grid[_ex|_stats], options) and the traversal-image kernels (trav_kernels.h)
//   trav_plain.hip  the kernels that walk the construction format: the reference-shaped one (statistics, Hit.id = steps) and v2
//   ray_order.hip   row-length detection of image-ordered batches and the counting sort of unordered ones (ray binning)
//   kat/kat.hip     known-answer hooks and timed diagnostic instantiations: a separate library, libhagrid_amd_kat.so (tests, dev tools)
#pragma once

#include "ctx.h"

#include "hagrid/grid.h"
#include "hagrid/prims.h"
#include "hagrid/ray.h"

namespace hagrid_trav {

using namespace hagrid;
using namespace hagrid_impl;

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
    int band_rows;                           // tile packets: rows of super-tiles per band (tile_packet_slot)
    int xcd_chunk_log2;                      // tile packets: blocks per XCD chunk, log2 (< 0: one eighth of the range per XCD)
    const int* __restrict__ tile_order;      // tail kernel (and the TIMES instantiations of kat/kat.hip): packet b processes tile tile_order[b]; nullptr: tile b
    int* tile_cost;                          // tail kernel: a wavefront leaves the iterations it ran at its tile's index (atomicMax); nullptr: nothing
    const float4* __restrict__ order_samples; // tail kernel: copy of the sample ray the tile order was learned on (2 float4); nullptr: the order is used unseen
    int* order_report;                       // pinned host word: receives order_epoch when the samples no longer fit the buffer
    int order_epoch;
    unsigned long long* __restrict__ wave_times; // TIMES instantiations only: start / end of every wavefront, 100 MHz wall clock
    const uint2* __restrict__ img_table;     // traversal image (trav_image.hip) or null
    const unsigned char* __restrict__ img_blocks;
    int img_wide;                 // host side only: the image holds wide records (which instantiation of the table-layout kernel is launched)
    // general layout: where a look-up that left its block starts again -- the image's VIRTUAL top level, `shift - gen_shift` (0 or 1) levels below the voxel map's
    // (trav_image.hip image_general_vtop): record gen_base + x' + gen_x * y' + gen_xy * z' with x' = x >> gen_shift
    int gen_shift, gen_x, gen_xy; uint32_t gen_base;
    size_t bin_working_set;       // host only: bytes a batch gathers from (traversal image or cells + entries, references, triangles): ray binning picks its bin count by it
    int num_rays;
    int lds_pad;                  // host only: dynamic LDS bytes per block of the tail kernel (experiments: fewer resident wavefronts)
    int tail_dual;                // host side only: which instantiation of the tail kernel is launched
    int mailbox;                  // host side only: the instantiation with a mailbox of the last four triangles per ray
    int quad_head;                // tail kernel: this many tiles at the HEAD of a learned tile order start with four lanes per ray (four blocks each, the first of the grid); 0: none
    int quad_first_block;         // tail kernel: blocks from this index on start with four lanes per ray (16 rays each, four blocks per tile); INT_MAX: none
    unsigned mode;                // v2 and the image kernel: HAGRID_TRAVERSE_ANY_HIT | HAGRID_TRAVERSE_UVS of this call (read at run time)
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

// the row-length word of detect_ray_rows: bit 30 = found by the second criterion (neighbouring ORIGINS: bounce rays in the image order of their primary hits --
// an image without coherent directions), the length in the bits below
constexpr int kRowsFromOrigins = 1 << 30, kRowLenMask = kRowsFromOrigins - 1;
__device__ __forceinline__ int tile_packet_row_len(const TraverseArgs& a) {      // 0: buffer order
    const int w = a.row_len_hint > 0 ? a.row_len_hint : (a.row_len ? (__builtin_amdgcn_readfirstlane(*a.row_len) & kRowLenMask) : 0);
    return (w < 8 || (w & 7) || a.num_rays / w < 8) ? 0 : w;
}

// Order of the tiles: BANDS of `band_rows` rows of super-tiles; inside a band the super-tile columns from left to right, inside a column its
// super-tiles from top to bottom, inside a super-tile the Z curve.  With one row per band (rounds 1-3) the tiles the machine holds at a time -- 8192
// wavefronts -- are a stripe across the whole image, 64 pixels high at 8192 pixels per row: primary rays of a stripe stay inside its thin frustum,
// but rays that start there in arbitrary directions (bounce rays in the image order of their primary hits) leave it at once and share nothing.
// Bands about as high as the in-flight set is wide make that set a compact block of the image.
__device__ __forceinline__ int tile_packet_slot(const TraverseArgs& a, int w, int b, int lane) {
    const int identity = b * 64 + lane;
    if (!w) return identity;
    const int tiles_x = w >> 3, tiles_y = (a.num_rays / w) >> 3;
    if (b >= tiles_x * tiles_y) return identity;             // ragged rows at the bottom, rays past the last full row
    const int S = 1 << a.super_log2;
    if (a.band_rows <= 1) {                                  // one row of super-tiles per band (launches of a few rounds): one division fewer on every wavefront's way to its rays
        const int band = b / (tiles_x * S), in_band = b - band * tiles_x * S;
        const int hb = min(S, tiles_y - band * S);
        const int col = in_band / (S * hb), in_super = in_band - col * S * hb;
        const int wc = min(S, tiles_x - col * S);
        int tx, ty;
        if (wc == S && hb == S) { tx = int(compact1by1(uint32_t(in_super))); ty = int(compact1by1(uint32_t(in_super) >> 1)); }
        else                    { ty = in_super / wc; tx = in_super - ty * wc; }
        return (((band * S + ty) << 3) + (lane >> 3)) * w + ((col * S + tx) << 3) + (lane & 7);
    }
    const int band_h = S * a.band_rows;                      // tile rows per band
    const int band = b / (tiles_x * band_h), in_band = b - band * tiles_x * band_h;
    const int hb = min(band_h, tiles_y - band * band_h);     // tile rows in this band
    const int col = in_band / (S * hb), in_col = in_band - col * S * hb;
    const int wc = min(S, tiles_x - col * S);                // tile columns in this column of super-tiles
    const int st = in_col / (wc * S), in_super = in_col - st * wc * S;       // (the super-tiles above this one in the column are S rows high)
    const int hs = min(S, hb - st * S);                      // tile rows in this super-tile
    int tx, ty;
    if (wc == S && hs == S) { tx = int(compact1by1(uint32_t(in_super))); ty = int(compact1by1(uint32_t(in_super) >> 1)); }
    else                    { ty = in_super / wc; tx = in_super - ty * wc; }
    const int px = ((col * S + tx) << 3) + (lane & 7), py = ((band * band_h + st * S + ty) << 3) + (lane >> 3);
    return py * w + px;
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

// ---- the general layout of slim records (trav_image.hip): one record per voxel-map entry, at the entry's index -------------------------------------
// The walk to the record of a voxel is the reference's lookup_entry (grid.h:103-116) over 16-byte records: the top-level record of the voxel's top-level cell,
// then -- while the record is a LINK -- the child the voxel selects in the block the link names.  A ray keeps the innermost block its last look-up ended in
// (`blk`: its first record, `bks`: k | s << 2 with (2^k)^3 records of 2^s finest-level voxels each); while the next voxel lies inside that block's region the
// look-up is one gather.  Per record form: LAST id field = NONE - 1 by index, NONE - 2 link, NONE - 3 wide (bounds in a.img_table as 16-byte wide records).
// A look-up that left its block starts at the image's VIRTUAL TOP LEVEL (round 5): a dense array of records one level below the voxel map's top level, each
// the record the walk from the top would have reached there -- on the oracle's traces of the clustered scene 74 % of a primary ray's look-ups start again and
// 58 % of those met a top-level link first (1.43 dependent gathers per step; with the virtual level 1.007, tests/analysis/general_walk_model.py).  A link
// names its block's first record and k (bits 80..81) and how many levels the block's region lies ABOVE the region the link was found in (`up`, bits 82..83:
// 1 for a top-level block of more than 2^3 entries seen from the virtual level, else 0).
template <int SLIM>
struct GenWalk {
    static constexpr int NI = 80 / SLIM, LAST = 48 + (NI - 1) * SLIM;
    static constexpr uint32_t NONE = (1u << SLIM) - 1u;
    uint32_t blk, bks;
    __device__ __forceinline__ static uint32_t last_field(const uint4& r) {
        constexpr int o = LAST - 96;                       // LAST >= 100 for both id widths: the field lies in the record's fourth word
        return (r.w >> o) & NONE;
    }
    __device__ __forceinline__ static bool is_link(const uint4& r) { return last_field(r) == NONE - 2u; }
    __device__ __forceinline__ static bool is_wide(const uint4& r) { return last_field(r) == NONE - 3u; }
    __device__ __forceinline__ static uint32_t word48(const uint4& r) { return (r.y >> 16) | (r.z << 16); }          // bits 48..79
    __device__ __forceinline__ static uint32_t count80(const uint4& r) { return (r.z >> 16) | ((r.w & 0xfu) << 16); }  // bits 80..99
    __device__ __forceinline__ int region_shift() const { return int(bks >> 2); }
    __device__ __forceinline__ static uint4 rec_at(const TraverseArgs& a, uint32_t index) { return *reinterpret_cast<const uint4*>(a.img_blocks + (index << 4)); }
    __device__ __forceinline__ static uint32_t child(int x, int y, int z, uint32_t k, uint32_t s) {
        const uint32_t m = (1u << k) - 1u;
        return ((uint32_t(x) >> s) & m) + (((((uint32_t(y) >> s) & m)) + (((uint32_t(z) >> s) & m) << k)) << k);
    }
    // the record of voxel (x, y, z) -- possibly a link: descend() before use; moved: the bits in which the voxel differs from the voxel of the previous
    // look-up, or-ed over the axes (nothing above the block's region changed: the voxel is still inside the block)
    // (Measured, round 5: resolving the link INSIDE the look-up -- the wavefront waits for the top-level record, the tests overlap the child's gather, as the
    // table of round 4's block layout had it -- is slower everywhere: configuration 3's grid at 4096^2 1.571 -> 1.615 ms, clustered scene 0.235 -> 0.246 ms.)
    __device__ __forceinline__ void restart(const TraverseArgs& a) { blk = ~0u; bks = uint32_t(a.gen_shift) << 2; }
    __device__ __forceinline__ static uint32_t link_region(const uint4& rec, uint32_t bks) { return (bks >> 2) + ((rec.z >> 18) & 3u) - ((rec.z >> 16) & 3u); }   // s of the link's block
    __device__ __forceinline__ uint4 lookup(const TraverseArgs& a, int x, int y, int z, uint32_t moved) {
        const uint32_t k = bks & 3u, s = bks >> 2;
        if (blk != ~0u && (moved >> (s + k)) == 0u) return rec_at(a, blk + child(x, y, z, k, s));
        restart(a);
        return rec_at(a, a.gen_base + uint32_t(x >> a.gen_shift) + __umul24(uint32_t(a.gen_x), uint32_t(y >> a.gen_shift)) + __umul24(uint32_t(a.gen_xy), uint32_t(z >> a.gen_shift)));
    }
    __device__ __forceinline__ void descend(const TraverseArgs& a, uint4& rec, int x, int y, int z) {
        while (is_link(rec)) {
            const uint32_t k = (rec.z >> 16) & 3u, s = link_region(rec, bks);
            blk = word48(rec); bks = k | s << 2;
            rec = rec_at(a, blk + child(x, y, z, k, s));
        }
    }
    __device__ __forceinline__ static uint4 wide_at(const TraverseArgs& a, const uint4& rec) { return reinterpret_cast<const uint4*>(a.img_table)[word48(rec)]; }
};

// the traversal image of the context as kernel arguments (traverse.hip, kat/kat.hip)
inline void image_args(const hagrid_ctx* ctx, TraverseArgs& a) {
    const TravImageCache& im = ctx->image;
    a.img_table = static_cast<const uint2*>(im.table);
    a.img_blocks = static_cast<const unsigned char*>(im.blocks);
    a.img_wide = im.wide_records > 0;
    a.gen_shift = a.shift - im.vtop_k; a.gen_x = a.top_x << im.vtop_k; a.gen_xy = (a.top_x << im.vtop_k) * (a.top_y << im.vtop_k); a.gen_base = im.vtop_base;
}

// ---- host entry points of the translation units ------------------------------------------------------------------------------
// traverse.hip: the argument block of a grid (setup_traversal's constants, traverse.cu:97-109); tris / rays / hits may be null when num_rays == 0
int make_args(hagrid_ctx* ctx, const hagrid_grid* g, const void* tris, const void* rays, void* hits, int num_rays, TraverseArgs& a);
// bytes from p to the end of the device allocation that holds it (all bits set if the runtime does not know the pointer)
size_t buffer_bytes_from(const void* p);
// trav_plain.hip: 256 threads per block (reference-shaped kernel), one wavefront per block (v2)
void launch_plain(hipStream_t st, int num_rays, bool small, const TraverseArgs& a);
void launch_v2(hipStream_t st, int blocks, bool small, bool narrow, unsigned mode, const TraverseArgs& a);
// ray_order.hip: row length of an image-ordered batch -> row_len[0] on the device (0: none); nobody waits for it
#ifndef HG_ORIGIN_MIN_RAYS
#define HG_ORIGIN_MIN_RAYS (1 << 18)
#endif
constexpr int kOriginMinRays = HG_ORIGIN_MIN_RAYS;
void launch_detect(hagrid_ctx* ctx, const TraverseArgs& a, int num_rays, int* row_len, int origin_min_rays = kOriginMinRays);
// ray_order.hip: the tail kernel's tile order (longest tile first): buffers of the context for `tiles` tiles; order <- the costs the
// launches since the last call left, costs cleared
constexpr int kMaxOrderTiles = 1 << 18;          // launches of more tiles keep the default order whatever the options say (the sort is one workgroup)
bool tile_order_buffers(hagrid_ctx* ctx, hagrid_ctx::RayHints& h, int tiles);
void launch_tile_order(hagrid_ctx* ctx, hagrid_ctx::RayHints& h, int tiles, const TraverseArgs& a, int rot, int* suggest);
inline float4* tile_order_samples(const hagrid_ctx::RayHints& h) { return reinterpret_cast<float4*>(h.lpt_buf + 2 * size_t(h.lpt_cap)); }

// Does the tile order still describe the rays in the buffer?  One sample ray (three eighths into the buffer) is compared BIT FOR BIT with the copy the sort
// left behind the order: lanes 0 .. 7 of the launch's FIRST wavefront load one dword of each, one compare, one ballot; when they differ the pinned word
// `order_report` receives the order's epoch and the host learns the order again before its next launch (traverse.hip).  The launch that finds the
// difference still follows the order it was given -- a bijection over the tiles whatever the rays are; a stale one costs that single launch up to a
// tenth against the default order.  (Checked by EVERY wavefront in front of its choice of tile -- so that a stale order is not even followed once -- the
// same dozen instructions cost the 1024^2 launch 1.6 %, a form with four samples and a drift tolerance 2 %: same-box A/B against round 3's library.  A
// tolerance buys nothing: a tile's cost is that of its longest ray, which half a pixel of camera motion carries into the next tile.)
__device__ __forceinline__ size_t order_sample_index(int n) { return size_t(uint32_t(n) >> 3) * 3u; }
__device__ __forceinline__ void order_check(const TraverseArgs& a, int lane) {
    uint32_t now = 0, then = 0;
    if (lane < 8) {
        now = reinterpret_cast<const uint32_t*>(a.rays + 2 * order_sample_index(a.num_rays))[lane];
        then = reinterpret_cast<const uint32_t*>(a.order_samples)[lane];
    }
    if (__ballot(now != then) != 0ull && lane == 0 && a.order_report) *a.order_report = a.order_epoch;
}

// ray_order.hip: ray binning as the context has it switched (hagrid_set_ray_binning); fills a.perm (+ a.perm_flag, a.row_len in the
// automatic mode) from buffers of `tmp`, or leaves a.perm null for batches too small to bin
int bin_rays(hagrid_ctx* ctx, TraverseArgs& a, int num_rays, PoolTemps& tmp);

} // namespace hagrid_trav

// trav_image.hip -- the traversal image: the finished grid re-laid-out for the vector L1 of gfx950.
//
// No reference counterpart.  The reference traverses the construction format directly (traverse.cu:27-95): per cell step a top-level entry, a
// second-level entry, a 32-byte cell, then per reference a 4-byte id and a 48-byte triangle, each in a different region of memory.  On MI355X a
// divergent gather costs one 128-byte line fill of the vector L1 (64 B/clk/CU) however few bytes are used (tools/micro/l1_gather.hip: 2.4
// CU-cycles per record for 4, 16 or 32 bytes), so what bounds the traversal is the number of distinct lines a ray touches and the number of
// dependent round trips, not its bytes.  The image is ONE 16-byte "slim" record per cell step that carries the cell's bounds and -- for lists of
// up to four -- the reference ids themselves:
//   bits   0..47   lo.x hi.x lo.y hi.y lo.z hi.z as byte offsets (lo and hi of an axis in neighbouring bytes; from where: see the layouts)
//   bits  48..127  80 / IDB reference ids of IDB bits (IDB = 20: four, IDB = 26: three); unused = all ones
//   by index       the LAST id field = all ones - 1; bits 48..79 first reference index, bits 80..99 list length
//   (table and general layouts) last id field = all ones - 3: WIDE, the cell's bounds do not fit a byte -- bits 48..79 index of a 16-byte wide record
//                  (lo | hi << 16 per axis in absolute finest-level voxels, first reference index), bits 80..99 list length;
//   (general layout) last id field = all ones - 2: LINK to a block of child records, bits 48..79 first child, bits 80..81 log2 of its edge
// in one of three layouts:
//   uniform   grids of at most three levels in which (nearly) every top-level cell has the full depth: (2^shift)^3 records per top-level cell, block T at
//             T * (2^shift)^3 -- the record of a voxel is arithmetic on the voxel; bounds count from the record's own voxel (voxel - lo, hi - voxel)
//   table     grids of at most three levels otherwise: per top-level cell a block of (2^d)^3 records, d its own depth, found through a table
//             (uint2 per top-level cell: offset in records, d); bounds count from the top-level cell's origin, biased by 128
//   general   any depth: the voxel map itself with every 4-byte entry widened to a record at the SAME index (below)
// Cell bounds, reference order and every value the traversal arithmetic uses are copied unchanged: hits are identical to the traversal of the
// construction format (tests/test_traverse_gpu.py).  Built by hagrid_setup_traversal (traverse.cu:97-109 is where the reference prepares its traversal
// state), owned by the context, dropped when the source arrays are freed, overwritten or rebuilt.  Grids no layout describes -- a virtual resolution of
// 65536 and more per axis, ids beyond 26 bits, a list of 2^20 ids, 2^28 records -- are traversed in the construction format (trav_plain.hip).
// (Rounds 1-4 also had 32-byte records -- blocks, nested blocks three levels deeper, links back into the construction format -- and a compact form with
// de-duplicated records behind slot bytes; the general layout serves every grid those served, at half the bytes and one gather per step at any depth.)
#include "ctx.h"

#include <algorithm>
#include "wave_prims.h"

#include "hagrid/grid.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct ImgK {
    const uint32_t* __restrict__ entries;
    const int4* __restrict__ cells;        // two int4 per cell; null for a compressed grid
    const uint4* __restrict__ small_cells; // compressed grid: one uint4 per cell (grid.h:36-45), lists end with a negative id
    const int* __restrict__ refs;
    int top_x, top_y, num_top;
    int shift, num_entries, num_cells;
    long long source_bytes;                // entries + cells of the construction format
};

// The deepest subdivision inside every top-level cell (at most D = shift <= 3 levels): the edge of its block in the table layout, and what decides
// between the table and the uniform layout.  One wavefront per top-level cell, a lane per finest-level voxel (strided).
// (Set-up kernels take the depth D and the id width IDB as ARGUMENTS: they run once per grid and stream; an instantiation per value bought nothing.)
__global__ void __launch_bounds__(64) image_depths(const ImgK k, const int D, uint32_t* __restrict__ metas) {
    const int V = 1 << (3 * D), M = (1 << D) - 1;
    const int T = blockIdx.x;
    const uint32_t topw = k.entries[T];
    int depth_max = 0;
    for (int f = threadIdx.x; f < V; f += 64) {
        const int rx = f & M, ry = (f >> D) & M, rz = f >> (2 * D);
        uint32_t w = topw;
        int depth = 0;
        while (w & 3u) {
            const int kk = int(w & 3u);
            depth += kk;
            if (depth > D) break;                 // (a map deeper than its shift says: not a grid; the fill finds out)
            const int s = D - depth, m = (1 << kk) - 1;
            w = k.entries[int(w >> 2) + ((rx >> s) & m) + ((((ry >> s) & m) + (((rz >> s) & m) << kk)) << kk)];
        }
        depth_max = max(depth_max, min(depth, D));
    }
    depth_max = wave_max(depth_max);
    if (threadIdx.x == 0) metas[T] = uint32_t(depth_max);
}

// ---- slim records -------------------------------------------------------------------------------------------------------
// bits [pos, pos + n) of the 128-bit record {lo, hi} := v   (n <= 32)
__device__ __forceinline__ void put_bits(unsigned long long& lo, unsigned long long& hi, int pos, int n, uint32_t v) {
    const unsigned long long m = n == 32 ? 0xffffffffull : ((1ull << n) - 1ull), x = v & m;
    if (pos < 64) {
        lo = (lo & ~(m << pos)) | (x << pos);
        if (pos + n > 64) hi = (hi & ~(m >> (64 - pos))) | (x >> (64 - pos));
    } else hi = (hi & ~(m << (pos - 64))) | (x << (pos - 64));
}

// ---- general layout: one slim record per voxel-map ENTRY ---------------------------------------------------------------------------
// Grids whose top-level cells differ in depth, and grids deeper than three levels (non-uniform scenes; the reference advertises N-level maps, README.md:12).
// The image is the voxel map itself with every 4-byte entry widened to a 16-byte slim record at the SAME index: a leaf entry becomes its cell (bounds, ids
// inline or by index: the slim record of the uniform layout), an inner entry a LINK to its block of (2^k)^3 children (which is where the map has it), so
// the walk from the top level to a cell is the reference's lookup_entry (grid.h:103-116) with the cell at its end for free, and the kernel keeps the
// innermost block it is in: while a ray stays inside that block's region a cell step is ONE 16-byte gather at any depth.
//   bound bytes    offsets from the ORIGIN OF THE ENTRY'S REGION (origin - lo, hi - origin; the region is 2^s finest-level voxels wide, s = shift - depth)
//   last id field  all ones - 1: list by index (as in the uniform layout) | - 2: link, bits 48..79 first child record, bits 80..81 k
//                  | - 3: WIDE, the cell's bounds do not fit a byte (the large cells of empty space): bits 48..79 index of a wide record, bits 80..99 list length;
//                  wide record (16 bytes, its own array): lo | hi << 16 per axis in absolute finest-level voxels, first reference index
// Built top down, one launch per level of the map: the top-level entries by one thread each, then one wavefront per inner entry writes the records of its
// children (it knows their origins) and lists the inner ones for the next launch.  Wide records are per CELL (a large cell is named by many entries).
struct GenItem { int entry; uint32_t oxy, ozs; int pad; };           // inner entry, origin of its region (x | y << 16, z | s << 16)

__device__ __forceinline__ bool general_record(const ImgK& k, const int IDB, uint32_t word, int ox, int oy, int oz, uint4* __restrict__ rec, int* __restrict__ claim, int* __restrict__ status,
                                               uint32_t up = 0u) {
    const int NI = 80 / IDB;
    const uint32_t NONE = (1u << IDB) - 1u;
    unsigned long long rl = ~0ull << 48, rh = ~0ull;               // no bounds, every id field "unused"
    if (word & 3u) {
        rl &= ~(0xffffffffull << 48);
        put_bits(rl, rh, 48, 32, word >> 2);
        put_bits(rl, rh, 80, 2, word & 3u);
        put_bits(rl, rh, 82, 2, up);                                  // levels between the region the link is found in and the region of its block (virtual top level: 1)
        put_bits(rl, rh, 48 + (NI - 1) * IDB, IDB, NONE - 2u);
        *rec = make_uint4(uint32_t(rl), uint32_t(rl >> 32), uint32_t(rh), uint32_t(rh >> 32));
        return true;
    }
    const int c = int(word >> 2);
    int lo[3], hi[3], begin, n;
    if (k.small_cells) {
        const uint4 sc = k.small_cells[c];
        lo[0] = int(sc.x & 0xffffu); lo[1] = int(sc.x >> 16); lo[2] = int(sc.y & 0xffffu);
        hi[0] = int(sc.y >> 16); hi[1] = int(sc.z & 0xffffu); hi[2] = int(sc.z >> 16);
        begin = int(sc.w); n = 0;
        if (begin >= 0) while (k.refs[begin + n] >= 0) n++;
        else begin = 0;
    } else {
        const int4 a = k.cells[2 * size_t(c)], b = k.cells[2 * size_t(c) + 1];
        lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
        begin = a.w; n = b.w - a.w;
    }
    const int o[3] = {ox, oy, oz};
    bool fits = true;
    int bad = 0;
    for (int ax = 0; ax < 3; ax++) {
        const int dl = o[ax] - lo[ax], dh = hi[ax] - o[ax];
        if (dl < 0 || dl > 255 || dh < 0 || dh > 255) fits = false;
        rl |= (unsigned long long)((uint32_t(dl) & 255u) | (uint32_t(dh) & 255u) << 8) << (16 * ax);
    }
    if (n >= (1 << 20)) bad |= 2;
    // an id that the field cannot tell from its markers -- in ANY list form: the kernels end a list by index at id == NONE as well -- : the whole image needs
    // the wider field (lists by index and lists of wide cells were not looked at)
    bool wide_ids = false;
    for (int i = 0; i < n; i++) wide_ids = wide_ids || uint32_t(k.refs[begin + i]) >= NONE - 3u;
    if (wide_ids) atomicAdd(status + 1, 1);
    if (!fits) {
        // the cell's wide record: the first entry that names the cell draws its index; the record holds the CELL until image_general_patch
        // replaces it by that index (the winner's store may not be visible to the other entries of the cell during this launch)
        if (atomicCAS(claim + c, -1, -2) == -1) claim[c] = atomicAdd(status + 2, 1);
        put_bits(rl, rh, 48, 32, uint32_t(c));
        put_bits(rl, rh, 80, 20, uint32_t(n));
        put_bits(rl, rh, 48 + (NI - 1) * IDB, IDB, NONE - 3u);
    } else {
        if (n <= NI && !wide_ids) {
            for (int i = 0; i < n; i++) put_bits(rl, rh, 48 + i * IDB, IDB, uint32_t(k.refs[begin + i]));
        } else {
            put_bits(rl, rh, 48, 32, uint32_t(begin));
            put_bits(rl, rh, 80, 20, uint32_t(n));
            put_bits(rl, rh, 48 + (NI - 1) * IDB, IDB, NONE - 1u);
        }
    }
    if (bad) atomicOr(status, bad);
    *rec = make_uint4(uint32_t(rl), uint32_t(rl >> 32), uint32_t(rh), uint32_t(rh >> 32));
    return false;
}

// the top level: one thread per entry
__global__ void __launch_bounds__(kBlock) image_general_top(const ImgK k, const int IDB, uint4* __restrict__ recs, GenItem* __restrict__ items, int* __restrict__ num_items,
                                                            int* __restrict__ claim, int* __restrict__ status) {
    const int T = blockIdx.x * kBlock + threadIdx.x;
    bool inner = false;
    int ox = 0, oy = 0, oz = 0;
    if (T < k.num_top) {
        ox = (T % k.top_x) << k.shift; oy = ((T / k.top_x) % k.top_y) << k.shift; oz = (T / (k.top_x * k.top_y)) << k.shift;
        const uint32_t w = k.entries[T];
        inner = general_record(k, IDB, w, ox, oy, oz, recs + T, claim, status);
        if (inner && k.shift < int(w & 3u)) { atomicOr(status, 4); inner = false; }
    }
    const int at = wave_append(inner ? 1 : 0, num_items);
    if (inner) items[at] = GenItem{T, uint32_t(ox) | uint32_t(oy) << 16, uint32_t(oz) | uint32_t(k.shift) << 16, 0};
}
// one level down: one wavefront per inner entry of the level above
__global__ void __launch_bounds__(kBlock) image_general_level(const ImgK k, const int IDB, uint4* __restrict__ recs, const GenItem* __restrict__ in, const int* __restrict__ num_in,
                                                              GenItem* __restrict__ items, int* __restrict__ num_items, int* __restrict__ claim, int* __restrict__ status) {
    const int i = blockIdx.x * kWaves + wave_id();
    if (i >= *num_in) return;
    const GenItem it = in[i];
    const uint32_t word = k.entries[it.entry];
    const int kk = int(word & 3u), first = int(word >> 2), s = int(it.ozs >> 16) - kk, m = (1 << kk) - 1;
    const int ox = int(it.oxy & 0xffffu), oy = int(it.oxy >> 16), oz = int(it.ozs & 0xffffu);
    for (int base = 0; base < (1 << (3 * kk)); base += 64) {               // (1, 8 or 64 children per round of the wavefront; 512 in eight rounds)
        const int c = base + lane_id();
        bool inner = false;
        int cx = 0, cy = 0, cz = 0;
        if (c < (1 << (3 * kk))) {
            cx = ox + ((c & m) << s); cy = oy + (((c >> kk) & m) << s); cz = oz + ((c >> (2 * kk)) << s);
            const uint32_t w = k.entries[first + c];
            inner = general_record(k, IDB, w, cx, cy, cz, recs + first + c, claim, status);
            if (inner && s < int(w & 3u)) { atomicOr(status, 4); inner = false; }       // a map deeper than its shift says: not a grid
        }
        const int at = wave_append(inner ? 1 : 0, num_items);
        if (inner) items[at] = GenItem{first + c, uint32_t(cx) | uint32_t(cy) << 16, uint32_t(cz) | uint32_t(s) << 16, 0};
    }
}
// The VIRTUAL TOP LEVEL, one level below the voxel map's: record v of a dense (2 top_x) x (2 top_y) x (2 top_z) array is what the walk from the top level
// reaches for the half-size region v of its top-level cell -- the cell (bounds from the region's origin), the child's link, or, where the top-level block has
// more than 2^3 entries, that block's link seen from one level further down (up = 1).  A look-up that left its block starts here (trav_common.h GenWalk):
// a top-level cell with eight leaf children costs it one gather instead of two.  One thread per record.
__global__ void __launch_bounds__(kBlock) image_general_vtop(const ImgK k, const int IDB, uint4* __restrict__ recs, int* __restrict__ claim, int* __restrict__ status) {
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= 8 * k.num_top) return;
    const int wx = 2 * k.top_x, wy = 2 * k.top_y;
    const int x = v % wx, y = (v / wx) % wy, z = v / (wx * wy);
    const int T = (x >> 1) + k.top_x * ((y >> 1) + k.top_y * (z >> 1)), oct = (x & 1) | (y & 1) << 1 | (z & 1) << 2;
    const int s = k.shift - 1, ox = x << s, oy = y << s, oz = z << s;
    const uint32_t e = k.entries[T];
    uint4* rec = recs + size_t(k.num_entries) + v;
    if ((e & 3u) == 1u) general_record(k, IDB, k.entries[(e >> 2) + oct], ox, oy, oz, rec, claim, status);      // the child: its cell or its link
    else general_record(k, IDB, e, ox, oy, oz, rec, claim, status, 1u);                                           // a leaf (its cell from here), or a larger block from one level down
}
// records that name a wide cell get the index of its wide record; the wide records themselves
__global__ void __launch_bounds__(kBlock) image_general_patch(const int IDB, uint4* __restrict__ recs, int num_entries, const int* __restrict__ claim, int first_wide) {
    const int NI = 80 / IDB, LAST = 48 + (NI - 1) * IDB;
    const uint32_t NONE = (1u << IDB) - 1u;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= num_entries) return;
    uint4 r = recs[i];
    unsigned long long rl = r.x | (unsigned long long)r.y << 32, rh = r.z | (unsigned long long)r.w << 32;
    const uint32_t last = uint32_t(rh >> (LAST - 64)) & NONE;
    if (last != NONE - 3u) return;
    const uint32_t c = uint32_t(rl >> 48) | uint32_t(rh & 0xffffu) << 16;
    put_bits(rl, rh, 48, 32, uint32_t(first_wide + claim[c]));
    recs[i] = make_uint4(uint32_t(rl), uint32_t(rl >> 32), uint32_t(rh), uint32_t(rh >> 32));
}
__global__ void __launch_bounds__(kBlock) image_general_wide(const ImgK k, int num_cells, const int* __restrict__ claim, uint4* __restrict__ wide) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= num_cells || claim[c] < 0) return;
    uint4 w;
    if (k.small_cells) {
        const uint4 sc = k.small_cells[c];               // min.x min.y | min.z max.x | max.y max.z  ->  lo | hi << 16 per axis
        w.x = (sc.x & 0xffffu) | (sc.y & 0xffff0000u); w.y = (sc.x >> 16) | (sc.z << 16); w.z = (sc.y & 0xffffu) | (sc.z & 0xffff0000u);
        w.w = int(sc.w) < 0 ? 0u : sc.w;
    } else {
        const int4 lo = k.cells[2 * size_t(c)], hi = k.cells[2 * size_t(c) + 1];
        w.x = uint32_t(lo.x) | uint32_t(hi.x) << 16; w.y = uint32_t(lo.y) | uint32_t(hi.y) << 16; w.z = uint32_t(lo.z) | uint32_t(hi.z) << 16;
        w.w = uint32_t(lo.w);
    }
    wide[claim[c]] = w;
}

// Returns 1 when the grid does not fit the general layout (ids beyond 26 bits, a list of 2^20 ids, 2^28 records, the size limit): no image then.
int build_general(hagrid_ctx* ctx, const ImgK& k, TravImageCache& img) {
    const int num_cells = k.num_cells;
    hipStream_t st = ctx->stream;
    if (k.num_entries <= 0 || k.num_entries >= (1 << 28) || k.shift > 15) return 1;
    // the virtual top level (image_general_vtop): where the map has a level below its top level and the record index stays within the kernels' 24-bit products
    const int top_z = k.num_top / std::max(k.top_x * k.top_y, 1);
    // the same size limit as the block layouts (build_blocks): "traverse.image_max_mb", else 8x the arrays the image replaces and at least 1 GB; a virtual top
    // level that does not fit it is left out before the image is
    const long long limit = ctx->opt_image_max_mb > 0 ? (long long)ctx->opt_image_max_mb << 20 : std::max(1ll << 30, 8 * k.source_bytes);
    bool vtop = ctx->opt_image_vtop && k.shift >= 1 && 4ll * k.top_x * k.top_y < (1 << 23) && 2ll * top_z < (1 << 23) && (long long)k.num_entries + 8ll * k.num_top < (1ll << 28);
    if (vtop && ((long long)k.num_entries + 8ll * k.num_top) * 16 > limit) vtop = false;
    const size_t records = size_t(k.num_entries) + (vtop ? 8u * size_t(k.num_top) : 0u);
    if ((long long)records * 16 > limit) return 1;
    const size_t cap = size_t(std::max(k.num_top, k.num_entries / 8)) + 1;
    uint4* recs = static_cast<uint4*>(hagrid_mem_alloc(ctx, records * 16u));
    GenItem* items[2] = {pool_alloc<GenItem>(ctx, cap), pool_alloc<GenItem>(ctx, cap)};
    int* claim = pool_alloc<int>(ctx, size_t(num_cells));
    auto release = [&]() { hagrid_mem_free(ctx, items[0]); hagrid_mem_free(ctx, items[1]); hagrid_mem_free(ctx, claim); };
    if (!recs || !items[0] || !items[1] || !claim) { release(); hagrid_mem_free(ctx, recs); return HAGRID_ENOMEM; }
    int* status = ctx->dscratch + 160;           // [0] does not fit | [1] lists with an id beyond the field | [2] wide records
    int* counts = ctx->dscratch + 164;           // inner entries listed per level
    int rc = 1;
    for (int idb : {20, 26}) {
        if (idb == 20 && ctx->opt_image_slim == 2) continue;           // "traverse.image_slim" = 2: the 26-bit form whatever the ids (tests)
        (void)hipMemsetAsync(status, 0, 24 * sizeof(int), st);
        (void)hipMemsetAsync(claim, 0xFF, size_t(num_cells) * sizeof(int), st);
        image_general_top<<<grid_blocks(k.num_top, kBlock), kBlock, 0, st>>>(k, idb, recs, items[0], counts, claim, status);
        HG_DBG(ctx);
        int level = 0, n = 0;
        rc = read_back(ctx, counts, &n, sizeof(int));
        while (rc == HAGRID_OK && n > 0 && level < 16) {
            GenItem* in = items[level & 1]; GenItem* out = items[(level + 1) & 1];
            image_general_level<<<grid_blocks(n, kWaves), kBlock, 0, st>>>(k, idb, recs, in, counts + level, out, counts + level + 1, claim, status);
            HG_DBG(ctx);
            level++;
            rc = read_back(ctx, counts + level, &n, sizeof(int));
            if (size_t(n) > cap) { rc = 1; break; }                       // (cannot happen for a voxel map whose blocks are disjoint)
        }
        if (rc != HAGRID_OK) break;
        if (vtop) { image_general_vtop<<<grid_blocks(8ll * k.num_top, kBlock), kBlock, 0, st>>>(k, idb, recs, claim, status); HG_DBG(ctx); }
        int h[3] = {0, 0, 0};
        rc = read_back(ctx, status, h, sizeof(h));
        if (rc != HAGRID_OK) break;
        if (h[0] || n > 0) { rc = 1; break; }       // a list of 2^20 ids, a map deeper than 16 levels
        if (h[1]) { rc = 1; if (idb == 20) continue; break; }       // ids of more than 20 bits: three ids of 26 bits per record; of more than 26: no image
        uint4* wide = static_cast<uint4*>(hagrid_mem_alloc(ctx, size_t(std::max(h[2], 1)) * 16u));
        if (!wide) { rc = HAGRID_ENOMEM; break; }
        if (h[2] > 0) {
            image_general_patch<<<grid_blocks((long long)records, kBlock), kBlock, 0, st>>>(idb, recs, int(records), claim, 0);
            HG_DBG(ctx);
            image_general_wide<<<grid_blocks(num_cells, kBlock), kBlock, 0, st>>>(k, num_cells, claim, wide); HG_DBG(ctx);
        }
        img.vtop_k = vtop ? 1 : 0; img.vtop_base = vtop ? uint32_t(k.num_entries) : 0u;
        img.blocks = recs; img.block_bytes = records * 16u; img.table = wide; img.table_bytes = size_t(std::max(h[2], 1)) * 16u;
        img.slim = idb; img.general = true; img.uniform = false; img.wide_records = h[2];
        rc = HAGRID_OK;
        break;
    }
    release();
    if (rc != HAGRID_OK) hagrid_mem_free(ctx, recs);
    return rc;
}

// One wavefront per top-level cell of a grid whose top-level cells all resolve `shift` = D levels.  status: bit 0 = a bound does
// not fit a byte, bit 1 = a by-index list is too long; word 1 counts the lists that hold an id of more than IDB bits.
// TABLE: the block of top-level cell T has depth metas[T] & 3 and starts at record offsets[T]; its bound bytes count from the origin
// of the top-level cell, biased by 128.
// TABLE only: a cell whose bounds do not fit the bytes (the large cells of empty space) gets a WIDE record, as in the general layout below: the record names the
// cell (image_general_patch replaces it by the index of the cell's wide record), status word 2 counts the wide records, claim[c] holds their indices.
template <bool TABLE>
__global__ void __launch_bounds__(64) image_slim_fill(const ImgK k, const int D, const int IDB, uint4* __restrict__ recs, uint2* __restrict__ table, int* __restrict__ status,
                                                      const uint32_t* __restrict__ metas, const int* __restrict__ offsets, int* __restrict__ claim) {
    const int NI = 80 / IDB;
    const uint32_t NONE = (1u << IDB) - 1u;
    const int T = blockIdx.x, lane = threadIdx.x;
    const int tx = T % k.top_x, ty = (T / k.top_x) % k.top_y, tz = T / (k.top_x * k.top_y);
    const uint32_t topw = k.entries[T];
    const int d = TABLE ? int(metas[T] & 3u) : D, sd = D - d, V = 1 << (3 * d);
    const size_t first = TABLE ? size_t(offsets[T]) : size_t(T) << (3 * D);
    if (lane == 0) table[T] = make_uint2(uint32_t(first), uint32_t(d) | 8u | 16u | (uint32_t(V) << 8));   // offset in records; bit 4: slim
    for (int f = lane; f < V; f += 64) {
        // block voxel f at depth d -> its lowest finest-level voxel inside the top-level cell
        const int rx = (f & ((1 << d) - 1)) << sd, ry = ((f >> d) & ((1 << d) - 1)) << sd, rz = (f >> (2 * d)) << sd;
        uint32_t w = topw;
        int depth = 0;
        while (w & 3u) {
            const int kk = int(w & 3u);
            depth += kk;
            const int s = D - depth, m = (1 << kk) - 1;
            w = k.entries[int(w >> 2) + ((rx >> s) & m) + ((((ry >> s) & m) + (((rz >> s) & m) << kk)) << kk)];
        }
        const int c = int(w >> 2);
        int lo[3], hi[3], begin, n;
        if (k.small_cells) {
            const uint4 sc = k.small_cells[c];
            lo[0] = int(sc.x & 0xffffu); lo[1] = int(sc.x >> 16); lo[2] = int(sc.y & 0xffffu);
            hi[0] = int(sc.y >> 16); hi[1] = int(sc.z & 0xffffu); hi[2] = int(sc.z >> 16);
            begin = int(sc.w); n = 0;
            if (begin >= 0) while (k.refs[begin + n] >= 0) n++;
            else begin = 0;
        } else {
            const int4 a = k.cells[2 * size_t(c)], b = k.cells[2 * size_t(c) + 1];
            lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
            begin = a.w; n = b.w - a.w;
        }
        const int v[3] = {(tx << D) + (TABLE ? 0 : rx), (ty << D) + (TABLE ? 0 : ry), (tz << D) + (TABLE ? 0 : rz)};
        unsigned long long rl = ~0ull << 48, rh = ~0ull;               // every id field "unused"
        int bad = 0;
        for (int ax = 0; ax < 3; ax++) {
            const int dl = TABLE ? lo[ax] - v[ax] + 128 : v[ax] - lo[ax], dh = TABLE ? hi[ax] - v[ax] + 128 : hi[ax] - v[ax];
            if (dl < 0 || dl > 255 || dh < 0 || dh > 255) bad |= 1;
            rl |= (unsigned long long)((uint32_t(dl) & 255u) | (uint32_t(dh) & 255u) << 8) << (16 * ax);
        }
        // an id that does not fit the field (the kernel takes NONE for the end of a list, wherever the id came from): the whole image
        // needs the wider field
        bool wide = false;
        for (int i = 0; i < n; i++) wide = wide || uint32_t(k.refs[begin + i]) >= NONE - (TABLE ? 3u : 1u);
        if (wide) atomicAdd(status + 1, 1);
        if (TABLE && (bad & 1)) {
            bad &= ~1;
            if (atomicCAS(claim + c, -1, -2) == -1) claim[c] = atomicAdd(status + 2, 1);
            if (n >= (1 << 20)) bad |= 2;
            put_bits(rl, rh, 48, 32, uint32_t(c));
            put_bits(rl, rh, 80, 20, uint32_t(n));
            put_bits(rl, rh, 48 + (NI - 1) * IDB, IDB, NONE - 3u);
        } else if (n <= NI && !wide) {
            for (int i = 0; i < n; i++) put_bits(rl, rh, 48 + i * IDB, IDB, uint32_t(k.refs[begin + i]));
        } else {
            if (n >= (1 << 20)) bad |= 2;
            put_bits(rl, rh, 48, 32, uint32_t(begin));
            put_bits(rl, rh, 80, 20, uint32_t(n));
            put_bits(rl, rh, 48 + (NI - 1) * IDB, IDB, NONE - 1u);
        }
        if (bad) atomicOr(status, bad);
        recs[first + f] = make_uint4(uint32_t(rl), uint32_t(rl >> 32), uint32_t(rh), uint32_t(rh >> 32));
    }
}

struct SlimSizeIn { const uint32_t* m; __device__ int operator()(int i) const { return 1 << (3 * int(m[i] & 3u)); } };
struct SlimSizeOut { int* v; __device__ void operator()(int i, int s) const { v[i] = s; } };

// uniform: every block has (2^D)^3 records, block T starts at T * (2^D)^3; otherwise `metas` holds the depth of every block and the
// offsets come from a scan over the block sizes.  Returns 1 when some cell does not fit a slim record.
template <int D>
int build_slim(hagrid_ctx* ctx, const ImgK& k, TravImageCache& img, uint2* table, bool uniform, const uint32_t* metas, int* offsets, int* partials) {
    long long records = (long long)k.num_top << (3 * D);
    if (!uniform) {              // (the caller scanned the block sizes into `offsets`; the total stands in its scratch word)
        int h = 0;
        const int rc = read_back(ctx, ctx->dscratch + 224, &h, sizeof(h));
        if (rc != HAGRID_OK) return rc;
        records = h;
    }
    if (records <= 0 || records >= (1ll << 28)) return 1;               // record offsets of the narrow kernels: 32-bit byte offsets
    const size_t bytes = size_t(records) * 16u;
    uint4* recs = static_cast<uint4*>(hagrid_mem_alloc(ctx, bytes));
    if (!recs) return HAGRID_ENOMEM;
    int* claim = uniform ? nullptr : pool_alloc<int>(ctx, size_t(k.num_cells));       // table layout: the wide record of every cell that needs one
    if (!uniform && !claim) { hagrid_mem_free(ctx, recs); return HAGRID_ENOMEM; }
    int* status = ctx->dscratch + 228;
    int result = 1;
    for (int idb : {20, 26}) {
        if (idb == 20 && ctx->opt_image_slim == 2) continue;           // "traverse.image_slim" = 2: the 26-bit form whatever the ids (tests)
        (void)hipMemsetAsync(status, 0, 3 * sizeof(int), ctx->stream);
        if (claim) (void)hipMemsetAsync(claim, 0xFF, size_t(k.num_cells) * sizeof(int), ctx->stream);
        if (uniform) image_slim_fill<false><<<k.num_top, 64, 0, ctx->stream>>>(k, D, idb, recs, table, status, nullptr, nullptr, nullptr);
        else         image_slim_fill<true><<<k.num_top, 64, 0, ctx->stream>>>(k, D, idb, recs, table, status, metas, offsets, claim);
        HG_DBG(ctx);
        int h[3] = {0, 0, 0};
        const int rc = read_back(ctx, status, h, sizeof(h));
        if (rc != HAGRID_OK) { result = rc; break; }
        if (h[0]) break;                            // uniform layout: some cell does not fit a slim record (the table layout has wide records for those); a list of 2^20 ids
        if (h[1] && idb == 20) continue;            // ids of more than 20 bits: three ids of 26 bits per record
        if (h[1]) break;                            // ... of more than 26 bits: no image
        if (h[2] > 0) {
            // wide records behind the table, in ONE buffer (the kernels know one more pointer than the records): 16-byte units from the table's start
            const size_t table16 = (size_t(k.num_top) * 8u + 15u) / 16u;
            uint4* both = static_cast<uint4*>(hagrid_mem_alloc(ctx, (table16 + size_t(h[2])) * 16u));
            if (!both) { result = HAGRID_ENOMEM; break; }
            (void)hipMemcpyAsync(both, table, size_t(k.num_top) * 8u, hipMemcpyDeviceToDevice, ctx->stream);
            image_general_patch<<<grid_blocks(records, kBlock), kBlock, 0, ctx->stream>>>(idb, recs, int(records), claim, int(table16));
            HG_DBG(ctx);
            image_general_wide<<<grid_blocks(k.num_cells, kBlock), kBlock, 0, ctx->stream>>>(k, k.num_cells, claim, both + table16); HG_DBG(ctx);
            img.table = both; img.table_bytes = (table16 + size_t(h[2])) * 16u; img.wide_records = h[2];      // (the caller's table is released with its other temporaries)
        } else { img.table = nullptr; img.table_bytes = size_t(k.num_top) * 8u; img.wide_records = 0; }
        img.blocks = recs; img.block_bytes = bytes; img.slim = idb;
        result = HAGRID_OK;
        break;
    }
    hagrid_mem_free(ctx, claim);
    if (result != HAGRID_OK) hagrid_mem_free(ctx, recs);
    return result;
}

// Grids of at most three levels: the uniform layout, else the table layout.  Returns 1 when neither describes the grid (the general layout is tried next).
template <int D>
int build_blocks(hagrid_ctx* ctx, const ImgK& k, TravImageCache& img) {
    uint32_t* metas = pool_alloc<uint32_t>(ctx, size_t(k.num_top) + 1);
    int* offs = pool_alloc<int>(ctx, size_t(k.num_top) + 1);
    int* partials = pool_alloc<int>(ctx, size_t(scan_num_tiles(k.num_top)) + 1);
    uint2* table = pool_alloc<uint2>(ctx, size_t(k.num_top));
    auto release = [&]() { hagrid_mem_free(ctx, metas); hagrid_mem_free(ctx, offs); hagrid_mem_free(ctx, partials); };
    if (!metas || !offs || !partials || !table) { release(); hagrid_mem_free(ctx, table); return HAGRID_ENOMEM; }
    image_depths<<<k.num_top, 64, 0, ctx->stream>>>(k, D, metas); HG_DBG(ctx);
    int* total = ctx->dscratch + 224;
    if (!ctx_scan<int>(ctx, SlimSizeIn{metas}, SlimSizeOut{offs}, k.num_top, partials, (const int*)nullptr, total)) { release(); hagrid_mem_free(ctx, table); return HAGRID_ENOMEM; }
    int table_records = 0;
    int rc = read_back(ctx, total, &table_records, sizeof(int));
    if (rc != HAGRID_OK) { release(); hagrid_mem_free(ctx, table); return rc; }
    // An image may not cost more than 8x the arrays it replaces (and at least 1 GB is always allowed; "traverse.image_max_mb" sets the limit).  The uniform layout
    // (every top-level cell subdivided to the full depth: the record of a voxel is found by arithmetic alone) whenever it fits that limit; where it costs more than a
    // quarter more records than the table layout, the table layout is built NEXT TO it for the batches without coherence (TravImageCache::alt_blocks; round 6, same
    // box, gpurun_out/r6geo: the soup at --snd-density 3 / 4 / 5 and configuration 3's grid, three levels, 0.98 - 1.29 GB uniform against 120 - 450 MB: primary rays
    // -7 ... -13 % with the uniform layout at 1024^2 and 4096^2, 4M binned incoherent rays +26 / +17 / -2 % -- rounds 4 - 5 took the table layout for all of them).
    // "traverse.image_uniform" = 2: the uniform layout alone (tests), 0: never.
    const long long limit = ctx->opt_image_max_mb > 0 ? (long long)ctx->opt_image_max_mb << 20 : std::max(1ll << 30, 8 * k.source_bytes);
    const long long uniform_records = (long long)k.num_top << (3 * D);
    const bool much_bigger = uniform_records * 4 > (long long)table_records * 5;
    const bool uniform = ctx->opt_image_uniform && uniform_records * 16 <= limit;
    int rs = 1;
    uint2* own_table = nullptr;                  // (the table layout with wide records brings a table of its own: table + wide records in one buffer)
    if (uniform) rs = build_slim<D>(ctx, k, img, table, true, metas, nullptr, partials);
    const bool is_uniform = rs == HAGRID_OK;
    if (rs == 1 && (long long)table_records * 16 <= limit) {       // top-level cells of different depth, or a cell the uniform layout's bytes cannot hold
        rs = build_slim<D>(ctx, k, img, table, false, metas, offs, partials);
        own_table = static_cast<uint2*>(img.table);
    }
    if (is_uniform && much_bigger && ctx->opt_image_uniform == 1 && (long long)table_records * 16 <= limit) {          // ... and the compact layout next to it
        uint2* table2 = pool_alloc<uint2>(ctx, size_t(k.num_top));
        TravImageCache second;
        const int ra = table2 ? build_slim<D>(ctx, k, second, table2, false, metas, offs, partials) : HAGRID_ENOMEM;
        if (ra == HAGRID_OK) {
            if (second.table) hagrid_mem_free(ctx, table2); else second.table = table2;          // (wide records: build_slim made one buffer of table + wide records)
            img.alt_blocks = second.blocks; img.alt_block_bytes = second.block_bytes; img.alt_table = second.table; img.alt_table_bytes = second.table_bytes;
            img.alt_wide_records = second.wide_records; img.alt_slim = second.slim;
        } else {                                // (no room, or a cell the table layout cannot say either: the uniform layout serves every batch)
            if (table2) hagrid_mem_free(ctx, table2);
            if (ra < 0) { ctx->err.clear(); (void)hipGetLastError(); }
        }
    }
    release();
    if (rs != HAGRID_OK) { hagrid_mem_free(ctx, table); return rs; }
    img.uniform = is_uniform;
    if (own_table) hagrid_mem_free(ctx, table); else img.table = table;
    return HAGRID_OK;
}

} // namespace

void hagrid_impl::trav_image_drop(hagrid_ctx* ctx) {
    TravImageCache& img = ctx->image;
    void* t = img.borrowed ? nullptr : img.table; void* b = img.borrowed ? nullptr : img.blocks;
    void* t2 = img.borrowed ? nullptr : img.alt_table; void* b2 = img.borrowed ? nullptr : img.alt_blocks;
    if (!img.borrowed && img.alive) img.alive->store(false);          // every borrower sees it before the memory is handed on
    img = TravImageCache();
    if (t) hagrid_mem_free(ctx, t);
    if (b) hagrid_mem_free(ctx, b);
    if (t2) hagrid_mem_free(ctx, t2);
    if (b2) hagrid_mem_free(ctx, b2);
}

bool hagrid_impl::trav_image_stale(const hagrid_ctx* ctx) {
    const TravImageCache& img = ctx->image;
    return img.valid && img.borrowed && !(img.alive && img.alive->load());
}

bool hagrid_impl::trav_image_matches(const hagrid_ctx* ctx, const hagrid_grid* g) {
    const TravImageCache& img = ctx->image;
    if (trav_image_stale(ctx)) return false;
    if (img.valid && img.detached)      // the descriptor of a released grid: no entries, no cells, everything else as at setup time
        return !g->entries && !g->cells && !g->small_cells && g->ref_ids == img.refs && g->num_cells == img.num_cells && g->num_entries == img.num_entries &&
               g->num_refs == img.num_refs && g->shift == img.shift && g->dims[0] == img.dims[0] && g->dims[1] == img.dims[1] && g->dims[2] == img.dims[2];
    return img.valid && g->entries == img.entries && (g->small_cells ? g->small_cells : g->cells) == img.cells && g->ref_ids == img.refs &&
           g->num_cells == img.num_cells && g->num_entries == img.num_entries && g->num_refs == img.num_refs && g->shift == img.shift &&
           g->dims[0] == img.dims[0] && g->dims[1] == img.dims[1] && g->dims[2] == img.dims[2];
}

void hagrid_impl::trav_image_source_touched(hagrid_ctx* ctx, const void* ptr, size_t bytes) {
    const TravImageCache& img = ctx->image;
    if (!img.valid || !ptr) return;
    const char* lo = static_cast<const char*>(ptr);
    const char* hi = lo + (bytes ? bytes : 1);
    auto overlaps = [&](const void* p, size_t n) { const char* q = static_cast<const char*>(p); return q < hi && lo < q + n; };
    if ((img.entries && overlaps(img.entries, size_t(img.num_entries) * 4)) || (img.cells && overlaps(img.cells, size_t(img.num_cells) * size_t(img.cell_bytes))) ||
        overlaps(img.refs, size_t(img.num_refs) * 4))
        trav_image_drop(ctx);
}

namespace {
// the largest primitive id the grid refers to (sentinels of a compressed grid are negative): traverse_grid sizes its padded triangle copy by it
__global__ void __launch_bounds__(kBlock) max_ref_kernel(const int* __restrict__ refs, int n, int* __restrict__ out) {
    int m = -1;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) m = max(m, refs[i]);
    m = wave_max(m);
    if (lane_id() == 0 && m >= 0) atomicMax(out, m);
}
} // namespace

int hagrid_impl::trav_image_build(hagrid_ctx* ctx, const hagrid_grid* g) {
    trav_image_drop(ctx);
    if (!ctx->opt_image || !g->entries || (!g->cells && !g->small_cells) || !g->ref_ids || g->num_cells <= 0) return HAGRID_OK;
    if (g->shift < 0 || g->shift > 15) return HAGRID_OK;
    for (int i = 0; i < 3; i++)
        if (g->dims[i] <= 0 || (long long)g->dims[i] << g->shift > 65535) return HAGRID_OK;
    const long long num_top = (long long)g->dims[0] * g->dims[1] * g->dims[2];
    if (num_top > (1ll << 30)) return HAGRID_OK;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    ImgK k;
    k.entries = static_cast<const uint32_t*>(g->entries);
    k.cells = g->small_cells ? nullptr : static_cast<const int4*>(g->cells);
    k.small_cells = static_cast<const uint4*>(g->small_cells);
    k.refs = static_cast<const int*>(g->ref_ids);
    k.top_x = g->dims[0]; k.top_y = g->dims[1]; k.num_top = int(num_top); k.shift = g->shift;
    k.source_bytes = 4ll * g->num_entries + (g->small_cells ? 16ll : 32ll) * g->num_cells;
    k.num_entries = g->num_entries; k.num_cells = g->num_cells;
    TravImageCache img;
    // Grids of at most three levels: a block of records per top-level cell (uniform or table layout); deeper grids, and grids those layouts cannot hold: the
    // general layout.  ("traverse.image_general" = 2 of the test library: the general layout for every grid; 0: never.)
    int rc = 1;
    if (g->shift >= 1 && g->shift <= 3 && ctx->opt_image_general != 2) {
        switch (g->shift) {
            case 1: rc = build_blocks<1>(ctx, k, img); break;
            case 2: rc = build_blocks<2>(ctx, k, img); break;
            default: rc = build_blocks<3>(ctx, k, img); break;
        }
    }
    if (rc == 1 && ctx->opt_image_general) rc = build_general(ctx, k, img);
    if (rc == 1) return HAGRID_OK;                 // no layout describes this grid (or fits the size limit): traversal reads the construction format
    if (rc != HAGRID_OK) return rc;
    img.valid = img.blocks != nullptr;
    img.entries = g->entries; img.cells = g->small_cells ? g->small_cells : g->cells; img.refs = g->ref_ids;
    img.cell_bytes = g->small_cells ? 16 : 32;
    img.num_cells = g->num_cells; img.num_entries = g->num_entries; img.num_refs = g->num_refs; img.shift = g->shift;
    img.dims[0] = g->dims[0]; img.dims[1] = g->dims[1]; img.dims[2] = g->dims[2];
    if (img.valid) img.alive = std::make_shared<std::atomic<bool>>(true);
    if (img.valid && g->num_refs > 0) {
        int* word = ctx->dscratch + kScrMaxRef;
        HG_HIP(ctx, hipMemsetAsync(word, 0xff, sizeof(int), ctx->stream));
        max_ref_kernel<<<std::min(grid_blocks(g->num_refs, kBlock), 2048), kBlock, 0, ctx->stream>>>(static_cast<const int*>(g->ref_ids), g->num_refs, word); HG_DBG(ctx);
        const int rb = read_back(ctx, word, &img.max_ref, sizeof(int));
        if (rb != HAGRID_OK) {                     // img is not the context's yet: its buffers go back here
            hagrid_mem_free(ctx, img.blocks); hagrid_mem_free(ctx, img.table);
            return rb;
        }
    }
    ctx->image = img;
    ctx->image_serial++;
    return HAGRID_OK;
}

extern "C" int hagrid_grid_release_for_traversal(hagrid_ctx* ctx, hagrid_grid* grid) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    TravImageCache& img = ctx->image;
    if (!trav_image_matches(ctx, grid) || img.detached) HG_FAIL(ctx, HAGRID_EINVAL, "release_for_traversal: call hagrid_setup_traversal for this grid first");
    if (img.borrowed) HG_FAIL(ctx, HAGRID_EINVAL, "release_for_traversal: this context only borrows the traversal image (hagrid_share_traversal); release the grid in the context that built it");
    void* entries = grid->entries;
    void* cells = grid->cells ? grid->cells : grid->small_cells;
    img.entries = nullptr; img.cells = nullptr; img.detached = true;      // the frees below must not take the image with them
    HG_TRY(hagrid_mem_free(ctx, entries));
    HG_TRY(hagrid_mem_free(ctx, cells));
    grid->entries = nullptr; grid->cells = nullptr; grid->small_cells = nullptr;
    return HAGRID_OK;
}

extern "C" int hagrid_share_traversal(hagrid_ctx* dst, hagrid_ctx* src) {
    if (!dst || !src || dst == src) return HAGRID_EINVAL;
    if (dst->device != src->device) HG_FAIL(dst, HAGRID_EINVAL, "share_traversal: the two contexts are on different devices");
    if (!src->image.valid || trav_image_stale(src)) HG_FAIL(dst, HAGRID_EINVAL, "share_traversal: the source context has no traversal image (hagrid_setup_traversal)");
    HG_HIP(dst, hipSetDevice(src->device));
    HG_HIP(dst, hipStreamSynchronize(src->stream));          // the image may still be under construction on the owner's stream
    trav_image_drop(dst);
    dst->image = src->image;
    dst->image.borrowed = true;
    return HAGRID_OK;
}

extern "C" int hagrid_traversal_image_info(hagrid_ctx* ctx, const hagrid_grid* grid, int32_t* format4, int64_t* image_bytes) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    if (!trav_image_matches(ctx, grid)) HG_FAIL(ctx, HAGRID_EINVAL, "no traversal image for this grid");
    const TravImageCache& img = ctx->image;
    if (format4) { format4[0] = img.general ? 2 : 1; format4[1] = (img.uniform ? 1 : 0) | (img.alt_blocks ? 2 : 0); format4[2] = img.slim; format4[3] = 16; }
    if (image_bytes) *image_bytes = (int64_t)img.block_bytes + (int64_t)img.table_bytes + (int64_t)img.alt_block_bytes + (int64_t)img.alt_table_bytes;
    return HAGRID_OK;
}

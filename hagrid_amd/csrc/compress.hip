// compress.hip -- hagrid_compress_grid: 16-bit cells and sentinel-terminated reference lists, on gfx950.
//
// Replaces the reference's compress.cu: compress_grid (:38-63), count_sentinel_refs (:6-13),
// emit_small_cells (:15-36).  Bit-identical to the CPU oracle.  The count lives in the scan's input functor.
#include "ctx.h"
#include "wave_prims.h"

#include "hagrid/grid.h"

using namespace hagrid;
using namespace hagrid_impl;

namespace {

struct SentinelIn {       // count_sentinel_refs, compress.cu:6-13
    const Cell* cells;
    __device__ int operator()(int i) const {
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
        const int n = p[1].w - p[0].w;
        return n > 0 ? n + 1 : 0;
    }
};

// emit_small_cells, compress.cu:15-36 (fused into the scan's output functor)
struct SmallCellOut {
    const Cell* cells; const int* refs; uint4* small_cells; int* out_refs;
    __device__ void operator()(int i, int first) const {
        const int4* p = reinterpret_cast<const int4*>(cells) + 2 * size_t(i);
        const int4 a = p[0], b = p[1];
        const int n = b.w - a.w;
        small_cells[i] = make_uint4((uint32_t(a.x) & 0xffffu) | (uint32_t(a.y) << 16),
                                    (uint32_t(a.z) & 0xffffu) | (uint32_t(b.x) << 16),
                                    (uint32_t(b.y) & 0xffffu) | (uint32_t(b.z) << 16),
                                    uint32_t(n > 0 ? first : -1));
        if (n > 0) {
            for (int j = 0; j < n; j++) out_refs[first + j] = refs[a.w + j];
            out_refs[first + n] = -1;
        }
    }
};
struct NullOut { __device__ void operator()(int, int) const {} };

} // namespace

extern "C" int hagrid_compress_grid(hagrid_ctx* ctx, hagrid_grid* grid) {
    if (!ctx || !grid) return HAGRID_EINVAL;
    if (!grid->cells || !grid->ref_ids) HG_FAIL(ctx, HAGRID_EINVAL, "compress_grid: incomplete (or already compressed) grid");
    trav_image_drop(ctx);            // the traversal image of this context describes a grid that is about to change
    for (int c = 0; c < 3; c++)
        if ((grid->dims[c] << grid->shift) >= (1 << 16)) return 0;          // compress.cu:41-44
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n = grid->num_cells;
    const Cell* cells = static_cast<const Cell*>(grid->cells);
    int* partials = pool_alloc<int>(ctx, size_t(scan_num_tiles(n)) + 1);
    uint4* small = static_cast<uint4*>(hagrid_mem_alloc(ctx, size_t(n) * 16));
    if (!partials || !small) { hagrid_mem_free(ctx, partials); hagrid_mem_free(ctx, small); return HAGRID_ENOMEM; }
    // the list lengths are known from the cells alone: total first (one small reduction + read-back), then
    // a single scan whose output functor writes the small cells and copies the lists
    int* total = ctx->dscratch;
    scan_partials<int, SentinelIn><<<std::max(scan_num_tiles(n), 1), kBlock, 0, st>>>(SentinelIn{cells}, n, partials); HG_DBG(ctx);
    scan_spine<int><<<1, kBlock, 0, st>>>(partials, scan_num_tiles(n), nullptr, total); HG_DBG(ctx);
    int h = 0;
    int rc = read_back(ctx, total, &h, sizeof(int));
    if (rc != HAGRID_OK || h < 0) { hagrid_mem_free(ctx, partials); hagrid_mem_free(ctx, small); return rc != HAGRID_OK ? rc : HAGRID_ERANGE; }
    int* srefs = pool_alloc<int>(ctx, size_t(h));
    if (!srefs) { hagrid_mem_free(ctx, partials); hagrid_mem_free(ctx, small); return HAGRID_ENOMEM; }
    scan_apply<int, SentinelIn, SmallCellOut><<<std::max(scan_num_tiles(n), 1), kBlock, 0, st>>>(
        SentinelIn{cells}, SmallCellOut{cells, static_cast<const int*>(grid->ref_ids), small, srefs}, n, partials); HG_DBG(ctx);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hagrid_mem_free(ctx, partials);
    ctx->counts.compressed = 1; ctx->counts.compress_cells = n; ctx->counts.compress_refs_out = h;
    if (e != hipSuccess) { hagrid_mem_free(ctx, small); hagrid_mem_free(ctx, srefs); HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e)); }
    hagrid_mem_free(ctx, grid->cells);          // compress.cu:55-60
    hagrid_mem_free(ctx, grid->ref_ids);
    grid->cells = nullptr;
    grid->small_cells = small;
    grid->ref_ids = srefs;
    grid->num_refs = h;
    return 1;
}

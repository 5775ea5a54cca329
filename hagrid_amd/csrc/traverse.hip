// traverse.hip -- gfx950 ray traversal of the irregular grid: the C ABI of the traversal side and the launches of the
// traversal-image kernels.
//
// Replaces the reference's traverse.cu: setup_traversal (:97-109), traverse_grid (:111-117) and the
// traverse<CellT, Tri> kernel (:27-95) with intersect_ray_box (:14-21) and compute_voxel (:23-25).
// Results per ray are identical to the CPU oracle's (same IEEE operation sequence, contraction off):
// the primitive id of the nearest hit (-1 on a miss) and its distance t.
//
// The grid constants travel as kernel arguments (scalar registers), not as __constant__ symbols, so several grids / contexts
// traverse concurrently.  Kernels: trav_kernels.h (traversal image: every BASELINE configuration), trav_plain.hip (construction
// format); how rays meet lanes: trav_common.h (tile packets), ray_order.hip (row-length detection, ray binning).  DESIGN.md 4.2.
#include "trav_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace hagrid;
using namespace hagrid_impl;
using namespace hagrid_trav;

namespace {

// the kernels of the traversal image: the tail kernel for the nearest hit, the image kernel for any-hit / barycentrics (and with "traverse.tail" = 0)
template <bool UVS>
bool launch_img_mode(hipStream_t st, int blocks, bool narrow, bool uniform, bool general, int slim, bool tail, const TraverseArgs& a) {
    if (!narrow || (slim != 20 && slim != 26)) return false;          // (32-bit offsets: the caller sends larger grids to the construction-format kernels)
    if (tail && a.mode == 0u && slim && general) {          // a record per voxel-map entry: grids deeper than three levels, cells too long for the block layouts' bound bytes
        if (a.tile_cost) {
            if (slim == 20) traverse_kernel_tail<20, false, false, false, true, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false, false, true, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
        } else {
            if (slim == 20) traverse_kernel_tail<20, false, false, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
        }
    }
    else if (tail && a.mode == 0u && slim && !uniform && a.img_wide) {          // table layout whose image holds wide records (cells its bound bytes cannot say)
        if (a.tile_cost) {
            if (slim == 20) traverse_kernel_tail<20, false, false, false, true, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false, false, true, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
        } else {
            if (slim == 20) traverse_kernel_tail<20, false, false, false, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false, false, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
        }
    }
    else if (tail && a.mode == 0u && slim && !uniform) {
        // (the table layout has no registers to spare for the second request of "traverse.tail_dual", and keeps costs for the tile order only where asked to)
        if (a.tile_cost) {
            if (slim == 20) traverse_kernel_tail<20, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false, false, true><<<blocks, 64, a.lds_pad, st>>>(a);
        } else {
            if (slim == 20) traverse_kernel_tail<20, false, false><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, false><<<blocks, 64, a.lds_pad, st>>>(a);
        }
    }
    else if (tail && a.mode == 0u && uniform && slim && a.mailbox) {          // (one id per round trip: the mailbox sits in front of every round; the caller's triangles)
        if (slim == 20) traverse_kernel_tail<20, false, true, false, true, true><<<blocks, 64, a.lds_pad, st>>>(a);
        else            traverse_kernel_tail<26, false, true, false, true, true><<<blocks, 64, a.lds_pad, st>>>(a);
    }
    else if (tail && a.mode == 0u && uniform && slim) {
        if (a.tail_dual) {
            if (slim == 20) traverse_kernel_tail<20, false, true, true><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26, false, true, true><<<blocks, 64, a.lds_pad, st>>>(a);
        } else {
            if (slim == 20) traverse_kernel_tail<20><<<blocks, 64, a.lds_pad, st>>>(a);
            else            traverse_kernel_tail<26><<<blocks, 64, a.lds_pad, st>>>(a);
        }
    }
    else if (slim == 20) {
        if (uniform)      traverse_kernel_img<UVS, 20, 0><<<blocks, 64, 0, st>>>(a);
        else if (general) traverse_kernel_img<UVS, 20, 2><<<blocks, 64, 0, st>>>(a);
        else              traverse_kernel_img<UVS, 20, 1><<<blocks, 64, 0, st>>>(a);
    } else {
        if (uniform)      traverse_kernel_img<UVS, 26, 0><<<blocks, 64, 0, st>>>(a);
        else if (general) traverse_kernel_img<UVS, 26, 2><<<blocks, 64, 0, st>>>(a);
        else              traverse_kernel_img<UVS, 26, 1><<<blocks, 64, 0, st>>>(a);
    }
    return true;
}
bool launch_img(hipStream_t st, int blocks, bool narrow, bool uniform, bool general, int slim, bool tail, unsigned mode, const TraverseArgs& a0) {
    TraverseArgs a = a0;
    a.mode = mode & 3u;             // (the any-hit rule is read at run time, the barycentrics are an instantiation: two more registers)
    return (mode & HAGRID_TRAVERSE_UVS) ? launch_img_mode<true>(st, blocks, narrow, uniform, general, slim, tail, a)
                                        : launch_img_mode<false>(st, blocks, narrow, uniform, general, slim, tail, a);
}

} // namespace

size_t hagrid_trav::buffer_bytes_from(const void* p) {
    hipDeviceptr_t base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) == hipSuccess) return size - size_t(static_cast<const char*>(p) - static_cast<const char*>(base));
    (void)hipGetLastError();
    return ~size_t(0);
}


int hagrid_trav::make_args(hagrid_ctx* ctx, const hagrid_grid* g, const void* tris, const void* rays, void* hits, int num_rays, TraverseArgs& a) {
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
    a.steps = nullptr; a.stats = nullptr; a.perm = nullptr; a.perm_flag = nullptr; a.wave_times = nullptr; a.tile_order = nullptr; a.tile_cost = nullptr; a.order_samples = nullptr; a.order_report = nullptr; a.order_epoch = 0;
    a.row_len = nullptr; a.row_len_hint = 0; a.super_log2 = ctx->opt_super_log2;
    // (the XCDs take chunks of 8 blocks in turn, of 32 in launches of at least twelve rounds of the resident wavefronts.  Round 6, same box, gpurun_out/r6geo: at 32 rounds 32
    // blocks per chunk gain 0.5 - 1.3 % -- the soup, configuration 3's grid -- and 0.6 % at 16 -- configuration 5's share; at 8 rounds, where round 4's rule took them as well,
    // they cost the stadium mesh 4.7 % and gain nothing anywhere; up to 4 rounds 4 ... 64 blocks are the same within a per cent on three scene families)
    a.xcd_chunk_log2 = ctx->opt_xcd_chunk_log2 != -2 ? ctx->opt_xcd_chunk_log2 : (grid_blocks(num_rays, 64) < 12ll * ctx->num_cus * 32 ? 3 : 5);
    // Bands of four rows of super-tiles for launches of at least eight rounds of the resident wavefronts, one row below ("traverse.band_rows" > 0
    // forces it).  Measured (profiles/NOTES.md "Round 4"): the bounce rays of configuration 5 (16 rounds at its per-GPU share) +3.7 % with four rows,
    // +1.9 % with eleven (a square in-flight block), -2 % with 22; primary batches of 8 and 32 rounds +-0 with four, -1 ... -2 % with eleven; a 1024^2
    // launch (two rounds) in the DEFAULT tile order 0.164 -> 0.180 ms with four: its second round then starts in a corner of the image.
    a.band_rows = ctx->opt_band_rows > 0 ? ctx->opt_band_rows : (grid_blocks(num_rays, 64) >= 8ll * std::max(ctx->num_cus, 1) * 32 ? 4 : 1);
    a.img_table = nullptr; a.img_blocks = nullptr; a.img_wide = 0; a.gen_shift = g->shift; a.gen_x = g->dims[0]; a.gen_xy = 0; a.gen_base = 0u;
    a.bin_working_set = 0; a.num_rays = num_rays; a.shift = g->shift; a.id_is_steps = 0; a.mode = 0u; a.quad_first_block = 0x7fffffff; a.quad_head = 0; a.lds_pad = ctx->opt_lds_pad; a.tail_dual = 0; a.mailbox = 0;
    a.dims_x = dims.x; a.dims_y = dims.y; a.dims_z = dims.z;
    a.top_x = g->dims[0]; a.top_y = g->dims[1];
    a.top_xy = (long long)g->dims[0] * g->dims[1] < (1 << 23) ? g->dims[0] * g->dims[1] : 0;
    a.min_x = lo.x; a.min_y = lo.y; a.min_z = lo.z;
    a.max_x = hi.x; a.max_y = hi.y; a.max_z = hi.z;
    a.cs_x = cs.x; a.cs_y = cs.y; a.cs_z = cs.z;
    a.inv_x = ginv.x; a.inv_y = ginv.y; a.inv_z = ginv.z;
    return HAGRID_OK;
}

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
        launch_plain(ctx->stream, num_rays, grid->small_cells != nullptr, a);
        HG_DBG(ctx);
        HG_HIP(ctx, hipGetLastError());
        return HAGRID_OK;
    }
    // binning buffers: released on every exit path.  The pool hands them to nobody else before the kernels below are done: in
    // keep mode free() only marks the slot (work of one context is stream-ordered), otherwise free() synchronises the stream first
    PoolTemps tmp(ctx);
    bool publish_row_len = false;
    {   // what the batch gathers from: the traversal image (or cells and entries), the references, the triangles
        const bool img = (ctx->opt_image || ctx->image.detached) && trav_image_matches(ctx, grid);
        a.bin_working_set = (img ? ((ctx->image.alt_blocks && ctx->ray_binning) ? ctx->image.alt_block_bytes : ctx->image.block_bytes) : size_t(grid->num_cells) * (grid->small_cells ? 16 : 32) + size_t(grid->num_entries) * 4)
                            + size_t(grid->num_refs) * 4 + size_t(std::max<int64_t>(ctx->counts.num_tris, 0)) * 48;
    }
    HG_TRY(bin_rays(ctx, a, num_rays, tmp));
    const int* perm = a.perm;
    // the hints kept for this ray buffer (row length, tile order): its slot, or the least recently used one (which then forgets its buffer)
    int hint_slot = 0;
    {
        int lru = 0;
        bool found = false;
        for (int i = 0; i < hagrid_ctx::kRayHints && !found; i++) {
            if (ctx->hints[i].key_rays == rays && ctx->hints[i].key_n == num_rays) { hint_slot = i; found = true; }
            if (ctx->hints[i].used < ctx->hints[lru].used) lru = i;
        }
        if (!found) {
            hint_slot = lru;
            hagrid_ctx::RayHints& N = ctx->hints[lru];
            N.key_rays = rays; N.key_n = num_rays;
            N.rowlen_rays = nullptr; N.rowlen_n = 0; N.rowlen_age = 0; N.rowlen_known = -1; N.rowlen_seen = 0;      // (a read-back still under way is overtaken by the next look)
            N.lpt_rays = nullptr; N.lpt_valid = false; N.lpt_rot = 0; N.rot_adopted = false; N.head_disabled = false; N.t_base = N.t_head = N.t_all = 0.0f; N.n_base = N.n_head = N.n_all = N.n_conf = 0; N.learned_all = false; N.all_stage = 0; N.cmp_pending = N.cmp_done = false; N.trial_pending = false; N.relearn_streak = 0; N.cooldown = 0; N.cooldown_len = 64; N.last_report = -1;
            // (the slot's epochs go on counting -- a launch over the forgotten buffer may still report one -- and the report word says "nothing": epochs are >= 1)
            N.lpt_epoch++; __atomic_store_n(ctx->mailbox + kMbxOrderStale + lru, -1, __ATOMIC_RELAXED); __atomic_store_n(ctx->mailbox + kMbxHeadSuggest + lru, 0, __ATOMIC_RELAXED);
            // A buffer of the same shape the context knows (a renderer's next frame in a new allocation) stands in until this one's own answers are there: its
            // row length counts as seen (the kernel reads the one found for THIS buffer either way), its tile order is the first order (below).
            const hagrid_ctx::RayHints* donor = nullptr;
            for (const auto& d : ctx->hints)
                if (&d != &N && d.key_n == num_rays && d.rowlen_seen > 0 && (!donor || d.used > donor->used)) donor = &d;
            N.share_choice = -1; N.share_last = -1; N.share_issued = N.share_done = 0; N.share_launches = 0; N.share_serial = ctx->image_serial; N.share_ncand = 0; N.share_shape_nc = 0; N.order_loses = false; N.learned_once = false;
            N.rows_from_origins = false;
            if (donor) { N.rowlen_seen = donor->rowlen_seen; N.rows_from_origins = donor->rows_from_origins; }
            if (donor && donor->share_serial == ctx->image_serial && donor->share_ncand > 0) {       // (the donor's answer, its candidates and times with it: same launch shape)
                N.share_ncand = donor->share_ncand; N.share_shape_nc = donor->share_shape_nc; for (int i = 0; i < 4; i++) { N.share_cands[i] = donor->share_cands[i]; N.share_t[i] = donor->share_t[i]; }
                N.share_choice = donor->share_choice; N.share_last = donor->share_last; N.share_launches = donor->share_launches; N.order_loses = donor->order_loses;
                if (N.share_choice < 0) N.share_ncand = N.share_shape_nc = 0;          // (a donor still measuring: measured here from nothing, with its last answer meanwhile)
            }
        }
        ctx->hints[hint_slot].used = ++ctx->hint_clock;
    }
    hagrid_ctx::RayHints& H = ctx->hints[hint_slot];
    if (H.order_serial != ctx->image_serial) {        // another traversal image since the slot's tile order was learned (another grid, another scene): learned from nothing
        H.order_serial = ctx->image_serial; H.lpt_rays = nullptr; H.lpt_valid = false; H.cooldown = 0; H.cooldown_len = 64;
    }
    // Kernel choice.  With a traversal image (hagrid_setup_traversal built one for this very grid) its kernel is used for every
    // batch; without one the latency-oriented v2 walks the construction format (trav_plain.hip).
    // hagrid_set_option("traverse.variant", 1|2|4) forces the reference-shaped kernel, v2 or the image kernel (tests, experiments).
    const bool have_image = (ctx->opt_image || ctx->image.detached) && trav_image_matches(ctx, grid);
    if (ctx->image.detached && have_image && ctx->opt_variant && ctx->opt_variant != 4)
        HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: this grid was released for traversal, only the traversal-image kernel can serve it");
    if (ctx->opt_variant == 4 && !have_image) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: no traversal image for this grid (hagrid_setup_traversal)");
    int variant = ctx->opt_variant ? ctx->opt_variant : (have_image ? 4 : 2);
    if (perm && variant != 4) variant = 2;            // (the reference-shaped kernel knows no permutation)
    const bool img_narrow = have_image && ctx->opt_narrow && a.top_xy > 0 && grid->dims[2] < (1 << 23) && buffer_bytes_from(tris) < (size_t(1) << 32) &&
                            ctx->image.block_bytes < (size_t(1) << 32) && size_t(grid->num_cells) * 32 < (size_t(1) << 32) &&
                            size_t(grid->num_entries) * 4 < (size_t(1) << 32) && size_t(grid->num_refs) * 4 < (size_t(1) << 32);
    // any-hit / barycentrics: the flat narrow image kernels and v2 have these variants
    if (flags && !(variant == 4 && img_narrow)) {
        if (ctx->image.detached && have_image) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: any-hit / barycentrics on a released grid need the narrow image kernel (arrays below 4 GB)");
        variant = 2;
    }
    if (variant == 4 && ctx->image.slim && !img_narrow) {
        // slim records are read by the narrow kernels only (arrays of 4 GB and more, "traverse.narrow" = 0): construction format
        if (ctx->image.detached) HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: this grid was released for traversal and its slim traversal image needs the narrow kernel (arrays below 4 GB)");
        variant = 2;
    }
    // (an image of two layouts -- trav_image.hip build_blocks: a uniform one much bigger than the table layout next to it -- serves the batches WITHOUT coherent directions
    // from the compact one: binned batches, and rays in image order whose row length came from neighbouring origins alone -- bounce rays -- once the host has seen that:
    // configuration 3's grid, bounce rays at 1024^2 / 2048^2 / 4096^2: +2 / +17 / +8 % with the table layout, primary rays -8 ... -13 %, gpurun_out/r6two)
    const bool compact = variant == 4 && ctx->image.alt_blocks && (perm || H.rows_from_origins);
    if (variant == 4) {
        image_args(ctx, a);
        if (compact) { a.img_table = static_cast<const uint2*>(ctx->image.alt_table); a.img_blocks = static_cast<const unsigned char*>(ctx->image.alt_blocks); a.img_wide = ctx->image.alt_wide_records > 0; }
    }
    const bool img_uniform = ctx->image.uniform && !compact;
    const int img_slim = compact ? ctx->image.alt_slim : ctx->image.slim;
    // Tile packets (v2 and the image kernel, not for binned batches): "traverse.image_width" > 0 gives the row length, 0
    // (default) looks for one on the device, -1 switches the feature off.  The kernel reads the answer from device memory,
    // nobody waits for it.
    if (!perm && ctx->opt_image_width >= 0 && (variant == 2 || variant == 4)) {
        if (ctx->opt_image_width > 0) {
            a.row_len_hint = ctx->opt_image_width;
        } else {
            // The row length only steers the lane <-> ray assignment (any value gives the same hits), so a row length FOUND for a ray
            // buffer is kept: calls with the same buffer and count reuse it and look again every 16th call ("traverse.row_cache" = 0:
            // at every call).  The host learns the answer without waiting (a 4-byte copy behind the traversal launch + an event it only
            // polls) and never keeps "not image-ordered" once it has seen it, so a buffer that alternates between unordered rays and an
            // image looks every time.  A buffer refilled with rows of another length runs on the stale length for at most 15 calls --
            // slower, never wrong.
            int* row_len = ctx->dscratch + kScrRowLen + hint_slot;
            const bool same = ctx->opt_row_cache && H.rowlen_rays == rays && H.rowlen_n == num_rays;
            if (same && H.rowlen_pending) {
                if (hipEventQuery(H.rowlen_evt) == hipSuccess) { const int word = ctx->mailbox[kMbxRowLen + hint_slot]; H.rowlen_known = word & kRowLenMask; H.rows_from_origins = (word & kRowsFromOrigins) != 0; H.rowlen_pending = false; H.rowlen_seen = H.rowlen_known; }
                else (void)hipGetLastError();                             // not ready yet: not an error
            }
            if (same && H.rowlen_known != 0 && H.rowlen_age < 15) H.rowlen_age++;
            else {
                launch_detect(ctx, a, num_rays, row_len);
                H.rowlen_rays = rays; H.rowlen_n = num_rays; H.rowlen_age = 0; H.rowlen_known = -1;
                publish_row_len = ctx->opt_row_cache != 0;
            }
            a.row_len = row_len;
        }
    }
    if (variant == 4) {
        int blocks = grid_blocks(num_rays, 64);
        const bool narrow = img_narrow;
        // Tail kernel: "traverse.quad_tail" per cent of the tiles, the last in dispatch order, start with four lanes per ray.  -1 (default):
        // by the size of the launch in rounds of the resident wavefronts (256 CUs x 32).  A launch of a few rounds is as long as its last
        // wavefronts' chains, and wavefront slots are free while it drains: the fewer rounds, the larger the share that pays
        // (profiles/dev_r3_quad_tail.txt, 14 image sizes x 8 shares on the 1M-triangle scene: up to 0.4 rounds all tiles (-11 ... -25 %),
        // up to 0.65 half of them (-14 % at 640 x 480), up to 1.1 rounds 37 % (-7 ... -16 %), up to 3.2 rounds a quarter (-3 ... -6 %;
        // 1024^2: 0.176 -> 0.171 ms), beyond that none (throughput-bound: +2 % at 1920 x 1080, +10 % at 4096^2).  Not for binned batches
        // (they lose), not while the image is shared between contexts: several batches in flight fill each other's drain and the
        // extra wavefronts only cost issue slots (two in flight: 0.117 -> 0.144 ms per batch).  Hits do not depend on it.
        // Tile order ("traverse.tile_order", tail kernel, rays in tile-packet order).  A launch of a few rounds of resident wavefronts ends when
        // the wavefronts that hold its longest rays end, and those start whenever the dispatch order reaches their tiles -- in the default
        // order half of them in the second round.  Every wavefront therefore leaves the number of iterations it ran (= cells of its longest
        // ray) at its tile's index, and the NEXT launches over the same ray buffer and count dispatch the tiles longest first (a stable sort of
        // the costs, one small kernel behind the launch that learns and behind every 32nd one after it; equal costs keep the Z order).  Like the
        // row length this only steers which wavefront takes which rays: hits never depend on it, a buffer refilled with other rays runs on a
        // stale order until the next refresh (slower at worst), a new buffer or count starts in the default order.
        // Measured with the reference step counts as the key (tools/dev_wave_timeline.py): 1024^2 0.181 -> 0.152 ms, mean occupancy 0.61 -> 0.79.
        // The kernel's own key counts an iteration with one ray per lane twice (its lists' rounds run one after the other): 1024^2 0.1455 ms with
        // plain iterations, 0.1380 / 0.1389 with that phase counted twice / three times, 0.142 with 2.5 on a key of half the resolution.
        const int tiles = blocks;
        bool learn_order = false;
        const long long rounds100 = 100ll * blocks / std::max((long long)ctx->num_cus * 32, 1ll);        // size of the launch in rounds of the resident wavefronts, per cent
        // ---- "traverse.share_trial" (round 6): what a launch in the DEFAULT order does with its tiles is measured, and so is whether a learned order beats it ----------
        // The share of tiles that start with four lanes per ray (the last in dispatch order) follows a rule fitted on the uniform soup (below: all tiles up to 0.4 rounds
        // ... none beyond 3.2, none for binned batches).  Scenes with a few expensive regions want HALF of their tiles at every size up to ten rounds (six blobs in a sparse
        // soup, back to back: 1024^2 0.259 -> 0.180 ms, 1280 x 720 0.183 -> 0.141 and 0.125 with all of them, 1920 x 1080 0.277 -> 0.211; the stadium mesh 0.300 -> 0.186;
        // a sphere shell 0.177 -> 0.165), the soup and a density gradient do not (+17 %, +10 %), bounce rays want none where the rule says a quarter (+12 ... +14 %), a
        // binned incoherent batch of two rounds wants all where the rule says none (+21 % on the soup) -- and nothing the host knows about a grid tells them apart
        // (gpurun_out/r6c, r6e, r6m: tools/dev_policy_regret.py).  An event pair around the kernel does: the first launches over a buffer run in the default order with the
        // candidates in turn -- the rule's share, a half, none, all (up to five rounds) -- three samples each, every sample with an event pair of its own (polled by later
        // calls, nobody waits; a caller that never synchronises has all of them in flight at once); the smallest time wins, the rule's share unless another is 3 % faster.
        // The answer is about the scene and the launch shape, not about the rays: it survives a camera that moves, a new buffer of the same shape starts with it, it is
        // measured again every 1024 launches.  A LEARNED tile order (below) is then held against it: an order whose steady launches (three timed ones, with the head share
        // if that was adopted) are not 3 % faster than the best default-order launch is not followed (clustered 1280 x 720: learned 0.181 ms, default order with all tiles
        // four lanes per ray 0.125; stadium 1920 x 1080 0.366 against 0.186; bounce rays over the soup at 2048^2 1.17 against 0.85 -- round 5's rule ordered all three).
        bool default_sample = false; int share_pct = -1;
        {
            const bool shared_image = ctx->image.alive && ctx->image.alive.use_count() > 1;
            const bool rows = a.row_len_hint > 0 || a.row_len != nullptr;
            const bool elig = ctx->opt_share_trial && ctx->opt_quad_tail < 0 && ctx->opt_tail && !flags && narrow && !shared_image && (perm || rows) && tiles >= 64 && tiles <= kMaxOrderTiles;
            if (elig) {
                const int rule = (perm != nullptr) ? 0 : (rounds100 <= 40 ? 100 : (rounds100 <= 65 ? 50 : (rounds100 <= 110 ? 37 : (rounds100 <= 320 ? 25 : 0))));
                int cands[4]; int nc = 0;
                cands[nc++] = rule;
                if (rounds100 <= 1000) {                     // (beyond ten rounds four lanes per ray lose everywhere measured: the rule's share -- none -- is sampled alone, for the order's comparison)
                    if (rule != 50) cands[nc++] = 50;
                    if (rule != 0) cands[nc++] = 0;
                    if (rule != 100 && rounds100 <= 500) cands[nc++] = 100;
                }
                // ... and, where the bands rule of make_args takes four rows of super-tiles (launches of eight rounds and more; fitted on configuration 5's bounce rays in round
                // 4), ONE row with the rule's share (bit 8 of a candidate): the stadium at 2048^2 0.386 -> 0.361 ms, bounce rays there 0.911 -> 0.878 (the last two cells of
                // tools/dev_policy_regret.py above 3 %).  The band decides which tile a block index means, so it is chosen here, before an order is learned, and then holds for
                // the order as well.
                if (!perm && ctx->opt_band_rows <= 0 && a.band_rows > 1 && nc < 4) cands[nc++] = rule | 256;
                if (H.share_serial != ctx->image_serial || H.share_shape_nc != nc || H.share_cands[0] != rule) {          // another grid, another launch shape: measured from nothing
                    H.share_serial = ctx->image_serial; H.share_shape_nc = H.share_ncand = nc; for (int i = 0; i < nc; i++) H.share_cands[i] = cands[i];
                    H.share_choice = -1; H.share_issued = H.share_done = 0; H.order_loses = false; H.learned_once = false;
                }
                nc = H.share_ncand;                           // (a later trial samples fewer candidates: below)
                while (H.share_done < H.share_issued) {
                    const int k = H.share_done;
                    if (hipEventQuery(H.share_evt[k][1]) != hipSuccess) { (void)hipGetLastError(); break; }          // not ready yet: not an error (samples finish in stream order)
                    float ms = 0.0f;
                    if (hipEventElapsedTime(&ms, H.share_evt[k][0], H.share_evt[k][1]) != hipSuccess || !(ms > 0.0f)) ms = 1.0e30f;
                    H.share_t[k % nc] = k < nc ? ms : std::min(H.share_t[k % nc], ms);
                    H.share_done++;
                }
                if (H.share_choice < 0 && H.share_done >= 3 * nc) {
                    int best = 0;
                    for (int i = 1; i < nc; i++) if (H.share_t[i] < 0.97f * H.share_t[0] && H.share_t[i] < H.share_t[best]) best = i;
                    H.share_choice = best; H.share_last = H.share_cands[best]; H.share_launches = 0;
                    // the learned order starts from nothing behind the samples (their costs come from launches in which a share of the tiles ran with four lanes per
                    // ray and counted differently: a sort over those suggests no head share where a clean one suggests a seventh of the tiles): sorted behind the next
                    // launch and behind the one after it, then timed -- without its head share and with it -- and held against the winner of the samples
                    // (the FIRST trial of a launch shape; a later one -- every 1024 launches -- only samples the default order again and holds the order's known times
                    // against the new winner: learning again would mean dozens of launches in the order alone, which on the stadium mesh is twice as slow as what
                    // the trials then settle on)
                    if (!H.learned_once) {
                        H.learned_once = true;
                        H.lpt_valid = false; H.lpt_age = 0; H.rot_adopted = false; H.head_disabled = false; H.t_base = H.t_head = H.t_all = 0.0f; H.n_base = H.n_head = H.n_all = H.n_conf = 0; H.learned_all = false; H.all_stage = 0; H.cmp_pending = H.cmp_done = false; H.lpt_rot = 0;
                    }
                }
                if (H.share_choice >= 0 && ++H.share_launches >= 1024) {         // (the scene in view may have changed: measured again, and the learned order held against it again)
                    // ... the candidates that came within 15 % of the winner only (the rule's share always): on the soup all tiles with four lanes per ray cost a launch
                    // twice its time -- a dozen such frames every 1024 are a hiccup a viewer sees
                    int keep = 1;
                    const float bar = 1.15f * H.share_t[H.share_choice];
                    for (int i = 1; i < nc; i++) if (H.share_t[i] <= bar) { H.share_cands[keep] = H.share_cands[i]; keep++; }
                    H.share_ncand = nc = keep;
                    H.share_choice = -1; H.share_issued = H.share_done = 0; H.order_loses = false; H.cmp_done = false; H.cmp_pending = false; H.n_conf = 0;
                }
                if (H.share_choice >= 0) share_pct = H.share_cands[H.share_choice];
                else if (H.share_issued < 3 * nc) { share_pct = H.share_cands[H.share_issued % nc]; default_sample = true; }
                else share_pct = H.share_last >= 0 ? H.share_last : rule;        // (the samples are still in flight: the last answer, else the rule)
                if (share_pct & 256) { a.band_rows = 1; share_pct &= 255; }      // (one row of super-tiles per band: in the learned order as well)
            }
        }
        {
            const bool tail_kernel = ctx->opt_tail && !flags && narrow;
            // by default for launches of up to 25 rounds (2048^2, eight rounds: -8 %; 2560^2: -4.9 %, 3072^2, 18 rounds: -1.3 %, 4096^2, 32 rounds: +-0 -- the tiles
            // of a class of equal cost are scattered over the image, and a throughput-bound launch pays for that in its caches) and not while the image is shared
            // between contexts (batches in flight fill each other's drain: two in flight 0.118 -> 0.119 ms per batch).
            const bool shared_image = ctx->image.alive && ctx->image.alive.use_count() > 1;
            // (with the share trial the order is held against the default order by measurement -- below -- and may be TRIED on launches of any size the sort covers; the
            // limit fitted in rounds 3 - 5 -- 25 rounds -- stands where nothing is measured: "traverse.share_trial" = 0 of the test library)
            const bool measured = share_pct >= 0;
            const int want = ctx->opt_tile_order < 0 ? (((measured || rounds100 <= 2500) && !shared_image) ? 1 : 0) : ctx->opt_tile_order;
            // (a row length the host has seen: a batch without one gets no tile packets and keeps the plain rules; while the length is looked for
            // again -- every 16th call -- the last answer counts)
            const bool rows_known = a.row_len_hint > 0 || (a.row_len && (H.rowlen_known > 0 || (H.rowlen_known < 0 && H.rowlen_seen > 0)));
            if (H.cooldown > 0) H.cooldown--;              // orders did not last on this buffer (a camera that moves fast): not learned for a while
            else if (want && tail_kernel && !perm && rows_known && tiles >= 64 && tiles <= kMaxOrderTiles && tile_order_buffers(ctx, H, tiles)) {
                int* report = ctx->mailbox + kMbxOrderStale + hint_slot;
                // What the launches since the last look reported (their first wavefront compares the sample ray the order was sorted with against the buffer, bit for bit,
                // and leaves the order's epoch when the buffer holds other rays): a report that names the current order.
                const int rep = __atomic_load_n(report, __ATOMIC_ACQUIRE);
                const bool new_report = rep != H.last_report && rep > 0;
                H.last_report = rep;
                const bool same_buffer = H.lpt_valid && H.lpt_rays == rays && H.lpt_n == num_rays && H.lpt_blocks == tiles;
                if (same_buffer && new_report && rep == H.lpt_epoch) {
                    // A launch since the last sort found other rays in the buffer than the order was learned on: learn again, from costs of the new rays only (the
                    // cost words hold the maximum over the launches since the last sort).  Rays that changed once (a buffer refilled now and then) get an order again; rays
                    // that change again within four launches of the last time are a camera that moves: no order for 64 launches, for twice as many every time that happens
                    // again, up to 1024 -- the default order with its measured share is what serves them (sorting again behind every launch from the previous frame's costs
                    // was built and measured: it loses 10 - 19 % at the reference viewer's speed, tools/proto/order_moving.patch).
                    const bool short_lived = ctx->hint_clock - H.relearn_clock < 4;
                    H.relearn_clock = ctx->hint_clock;
                    // (what the head share's trial found -- four lanes per ray for the longest tiles pay on this scene at this launch shape, or do not -- is about the scene,
                    // not about these rays: a concluded trial stands, the next order is stored with the same share at its head; an unfinished one starts again)
                    H.lpt_valid = false; H.lpt_age = 0;
                    if (!H.cmp_done && !(H.n_base >= 3 && (H.n_head >= 3 || H.head_disabled))) { H.rot_adopted = false; H.head_disabled = false; H.t_base = H.t_head = H.t_all = 0.0f; H.n_base = H.n_head = H.n_all = H.n_conf = 0; H.learned_all = false; H.all_stage = 0; H.cmp_pending = H.cmp_done = false; }
                    (void)hipMemsetAsync(H.lpt_buf, 0, size_t(tiles) * sizeof(int), ctx->stream);
                    if (short_lived) { H.cooldown = H.cooldown_len; H.cooldown_len = std::min(2 * H.cooldown_len, 1024); }
                }
                if (H.cooldown > 0) { /* this launch and the next ones: default order, no costs */ }
                else {
                if (H.lpt_rays != rays || H.lpt_n != num_rays || H.lpt_blocks != tiles) {
                    H.lpt_rays = rays; H.lpt_n = num_rays; H.lpt_blocks = tiles; H.lpt_age = 0;
                    (void)hipMemsetAsync(H.lpt_buf, 0, size_t(tiles) * sizeof(int), ctx->stream);   // (costs some launch over another buffer may have left)
                    H.lpt_valid = false;
                    // the order of the most recently used buffer of the same shape, if there is one
                    const hagrid_ctx::RayHints* donor = nullptr;
                    for (const auto& d : ctx->hints)
                        if (&d != &H && d.lpt_valid && d.lpt_buf && d.lpt_n == num_rays && d.lpt_blocks == tiles && (!donor || d.used > donor->used)) donor = &d;
                    if (donor && hipMemcpyAsync(H.lpt_buf + H.lpt_cap, donor->lpt_buf + donor->lpt_cap, size_t(tiles) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                        hipMemcpyAsync(tile_order_samples(H), tile_order_samples(*donor), 8 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess) {
                        // (with the donor's sample rays: the order is followed only if THIS buffer's rays are near them)
                        // (the stand-in's refresh count goes on: buffers that come and go -- a new allocation per frame -- still re-sort every 32nd launch,
                        // from the costs of that one launch: every wavefront of a launch leaves its cost)
                        H.lpt_valid = true; H.lpt_period = 32; H.lpt_age = donor->lpt_age; H.lpt_rot = donor->lpt_rot;
                        H.lpt_epoch++;                 // an order of its own epoch: no report written so far can name it
                    }
                }
                a.tile_cost = H.lpt_buf;
                if (H.lpt_valid) { a.tile_order = H.lpt_buf + H.lpt_cap; a.order_samples = ctx->opt_order_gate ? tile_order_samples(H) : nullptr; a.order_report = report; a.order_epoch = H.lpt_epoch; }
                // (sorted behind the launch that learns, behind the next one -- the first costs come from a launch in which a share of the tiles
                // started with four lanes per ray and counted differently -- and behind every 32nd after that)
                learn_order = !H.lpt_valid || ++H.lpt_age >= H.lpt_period;
                if (H.lpt_valid && learn_order && H.lpt_period >= 32) H.cooldown_len = 64;       // an order that lasted through a refresh period
                }
            }
        }
        const bool order_lost = H.order_loses && share_pct >= 0 && ctx->opt_tile_order < 0;           // ("traverse.tile_order" = 1 of the test library: followed whatever it costs)
        // (the samples keep no costs: shares of their tiles run with four lanes per ray and count differently, the order is learned from nothing behind them anyway -- and on
        // the table and general layouts cost bookkeeping is another instantiation, a few per cent slower on launches of many rounds: the default order has to be timed as it
        // would run for good, without it.  Configuration 3 at 4096^2: 13.3 Grays/s so, 12.4 when an order that only beat cost-keeping samples was followed, gpurun_out/r6z2)
        if (default_sample) a.tile_cost = nullptr;
        // Before an order is accepted or dropped for good the default order is timed ONCE MORE, next to the order's own samples: the first samples of a launch shape are the
        // first launches of a process as often as not (clocks still rising, first touches: 1.31 ms where the steady default order takes 1.13 at 4096^2 on the soup) and an
        // order measured half a second later would beat them whatever it is worth (configuration 3 at 4096^2: 12.3 instead of 13.3 Grays/s, gpurun_out/r6w).
        const bool conf_sample = a.tile_order && H.cmp_pending && H.n_conf < 3 && !H.trial_pending && !learn_order && !default_sample;
        if (conf_sample) a.tile_cost = nullptr;
        if (a.tile_order && (default_sample || order_lost || conf_sample)) { a.tile_order = nullptr; a.order_samples = nullptr; }
        if (order_lost) { learn_order = false; a.tile_cost = nullptr; }          // (an order that lost is neither followed nor refreshed: launches without the cost bookkeeping)
        // "traverse.tail_dual": phase 1 of the tail kernel tests two ids of an inline list per round trip (the second triangle comes through
        // LDS, trav_kernels.h test_list).  -1 (default): for rays in tile-packet order (1024^2: -2.6 %, 640 x 480: -4.4 %, 2048^2 and
        // beyond -0.2 ... -0.4 %), not for binned batches (+2.2 %: their wavefronts hold few rays per cell, the second request is mostly
        // issued for one or two lanes).  Hits do not depend on it.
        a.tail_dual = ctx->opt_tail_dual < 0 ? (perm ? 0 : 1) : ctx->opt_tail_dual;
        // "traverse.mailbox": every ray skips a triangle it was tested against among its last four tests (trav_kernels.h, MAILBOX; seven instead of eight
        // wavefronts per SIMD, one id per round trip).  What it saves are triangle fetches; what it costs is an LDS round trip in front of every
        // triangle round and a resident wavefront.  -1 (default): for BINNED batches of at least eight rounds of wavefronts whose rays are fewer than the 64-byte
        // sectors of a working set beyond 512 MB -- nearly every fetch is a first touch there: 16M binned incoherent rays over the 8M-triangle soup 6.54 -> 5.98 ms
        // (-9 %, round 6, gpurun_out/r6pad).  Not for rays in image order (the 8.4M bounce rays of configuration 5's per-GPU share: 5263 Mrays/s without, 4851 with;
        // round 4 had measured +3.6 % there, before bands and measured orders), not for cache-resident scenes (-1 % ... -13 %).  The instantiation whose lanes took new
        // rays as they finished (REFILL, round 4: +6.5 % on that share then) lost to the plain kernel by 2.5 - 6.7 % on the same share in round 6 and was removed
        // (tools/proto/pruned_r6.patch).
        a.mailbox = ctx->opt_mailbox < 0 ? (perm != nullptr && a.bin_working_set > (size_t(512) << 20) && grid_blocks(num_rays, 64) >= 8ll * std::max(ctx->num_cus, 1) * 32 &&
                                            size_t(num_rays) * 64 < a.bin_working_set)
                                         : ctx->opt_mailbox;
        if (a.mailbox) a.tail_dual = 0;
        int quad_pct = ctx->opt_quad_tail;
        const bool share_timed = default_sample;
        if (quad_pct < 0) {
            const long long slots = (long long)ctx->num_cus * 32;
            const bool shared = ctx->image.alive && ctx->image.alive.use_count() > 1;
            const long long r100 = 100ll * blocks / std::max(slots, 1ll);           // rounds, in per cent
            quad_pct = (perm || shared) ? 0 : (r100 <= 40 ? 100 : (r100 <= 65 ? 50 : (r100 <= 110 ? 37 : (r100 <= 320 ? 25 : 0))));
            // in a learned tile order the tiles with the longest rays come first: all tiles of a launch of up to one round start with four lanes
            // per ray (256^2 ... 960 x 540: -8 ... -23 % against the shares above in the default order), none of a larger one
            if (a.tile_order && !shared) quad_pct = r100 <= 100 ? 100 : 0;
            else if (share_pct >= 0) quad_pct = share_pct;               // the default order: the share the trial above chose (or is sampling)
        }
        // "traverse.quad_head": in a learned order of a launch of MORE than one round the tiles that cost several times the median tile -- the chains the launch is as
        // long as, where a scene has a few dense objects -- start with four lanes per ray, and first.  The sort counts them (a pinned word the host polls), the NEXT sort
        // stores the order rotated by that many positions (its last lpt_rot positions are the longest tiles) and the kernel dispatches the blocks of those positions
        // first (a.quad_head).  The share follows the suggestion at the periodic sorts; the first suggestion gets a sort of its own.
        int want_rot = 0;
        int* suggest = ctx->mailbox + kMbxHeadSuggest + hint_slot;
        const bool head_ok = ctx->opt_quad_head > 0 && ctx->opt_quad_tail < 0 && rounds100 > 100 && rounds100 <= 500 && !perm && ctx->opt_tail && !flags && narrow &&
                             !(ctx->image.alive && ctx->image.alive.use_count() > 1);
        // The share measures itself (the rule above is fitted on two scene families; on a soup with a density gradient it takes tiles whose lists are short and loses
        // 7 - 19 %): timed launches in the learned order without it first (three samples, the smallest counts), then with it; a share that is not 3 % faster is dropped
        // until the order is learned again from nothing.
        if (H.trial_opt != ctx->opt_quad_head || H.head_serial != ctx->image_serial) {            // (the test library changed the threshold, or another grid: the trial starts again)
            H.trial_opt = ctx->opt_quad_head; H.head_serial = ctx->image_serial; H.head_disabled = false; H.rot_adopted = false; H.t_base = H.t_head = H.t_all = 0.0f; H.n_base = H.n_head = H.n_all = H.n_conf = 0; H.learned_all = false; H.all_stage = 0; H.cmp_pending = H.cmp_done = false;
        }
        if (H.trial_pending && hipEventQuery(H.trial_evt[1]) == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, H.trial_evt[0], H.trial_evt[1]) == hipSuccess && ms > 0.0f) {
                if (H.trial_kind == 1) { H.t_head = H.n_head ? std::min(H.t_head, ms) : ms; H.n_head++; }
                else if (H.trial_kind == 2) { H.t_all = H.n_all ? std::min(H.t_all, ms) : ms; H.n_all++; }
                else if (H.trial_kind == 3) { H.t_conf = H.n_conf ? std::min(H.t_conf, ms) : ms; H.n_conf++; }
                else { H.t_base = H.n_base ? std::min(H.t_base, ms) : ms; H.n_base++; }
            }
            H.trial_pending = false;
            if (H.n_head >= 3 && H.n_base >= 3 && !H.head_disabled && H.t_head > 0.97f * H.t_base) H.head_disabled = true;
        } else if (H.trial_pending) (void)hipGetLastError();          // not ready yet: not an error
        // A third candidate of the learned order, for launches of one to five rounds in which the head share is not in play (not suggested, or measured and dropped): ALL
        // tiles with four lanes per ray, in an order learned from launches that run that way -- an order learned from one-ray-per-lane costs serves it badly (clustered
        // 1280 x 720: the order alone 0.189 ms, all tiles in that order 0.170, in an order of their own 0.124, the best default-order launch 0.142; stadium 1920 x 1080
        // 0.376 / 0.248 / 0.188 / 0.204: gpurun_out/r6s, r6t).  Stages: 1, 2 = a launch that way with a sort behind it each (the second sort sees only such costs),
        // 3 = three timed launches; kept if 3 % faster than the order alone, else the order is learned again from one-ray-per-lane launches.
        const bool head_wanted = head_ok && !H.head_disabled && (H.lpt_rot > 0 || __atomic_load_n(suggest, __ATOMIC_RELAXED) > 0);
        const bool head_done = !head_wanted || H.n_head >= 3;
        const bool all_ok = share_pct >= 0 && rounds100 > 100 && rounds100 <= 500 && !perm && ctx->opt_tile_order < 0;
        bool all_sample = false;
        if (a.tile_order && all_ok && H.lpt_rot == 0 && H.n_base >= 3 && quad_pct == 0 && (H.all_stage > 0 || (head_done && !(H.n_head >= 3 && !H.head_disabled)))) {
            if (H.all_stage == 0 && !learn_order && !H.trial_pending) H.all_stage = 1;
            if (H.all_stage == 1 || H.all_stage == 2) { quad_pct = 100; learn_order = true; H.all_stage++; }                 // (the sort below: behind this launch)
            else if (H.all_stage == 3) {
                quad_pct = 100;
                if (H.n_all >= 3) {
                    H.learned_all = H.t_all < 0.97f * H.t_base; H.all_stage = 4;
                    if (!H.learned_all) { H.lpt_valid = false; quad_pct = 0; a.tile_order = nullptr; a.order_samples = nullptr; learn_order = true; }      // back: learned again from one-ray-per-lane launches
                } else all_sample = !learn_order && !H.trial_pending && H.lpt_age >= 2;
            } else if (H.learned_all) quad_pct = 100;
        }
        const bool all_done = !all_ok || H.all_stage >= 4 || (H.n_head >= 3 && !H.head_disabled);
        // the learned order against the best default-order launch (the share trial's): all known -> an order that is not 3 % faster is not followed
        if (share_pct >= 0 && H.share_choice >= 0 && !H.order_loses && !H.cmp_done && H.n_base >= 3 && head_done && all_done) {
            if (H.n_conf < 3) H.cmp_pending = true;              // (three launches in the default order, timed now: above)
            else {
                float learned = H.t_base;
                if (H.n_head >= 3 && !H.head_disabled) learned = std::min(learned, H.t_head);
                if (H.all_stage >= 4 && H.learned_all) learned = std::min(learned, H.t_all);
                if (learned > 0.97f * std::min(H.share_t[H.share_choice], H.t_conf)) H.order_loses = true;
                H.cmp_pending = false; H.cmp_done = true;
            }
        }
        if (head_ok && H.n_base >= 3 && !H.head_disabled && H.all_stage == 0) {
            const int chunk = 8 << (a.xcd_chunk_log2 >= 0 ? a.xcd_chunk_log2 : 4);
            const int s = std::min(std::max(__atomic_load_n(suggest, __ATOMIC_RELAXED), 0), tiles / 8);
            const int full = std::min(blocks, (blocks - s + chunk / 2) / chunk * chunk);            // (whole XCD chunks of ordinary blocks)
            // a share once taken up stays: the costs the head's tiles leave while they run with four lanes per ray are lower, a sort over those would suggest none
            // (and the share would come and go every other refresh)
            want_rot = H.lpt_rot > 0 ? H.lpt_rot : blocks - full;
        }
        if (a.tile_order && H.lpt_rot > 0 && head_ok && !H.head_disabled) {
            a.quad_first_block = tiles - H.lpt_rot; a.quad_head = H.lpt_rot; blocks = tiles + 3 * H.lpt_rot;
        } else if (a.tile_order && H.lpt_rot > 0) {
            a.tile_order = nullptr; a.order_samples = nullptr; learn_order = true;              // (rotated for a launch this one is not: default order, sorted again behind it)
        } else if (quad_pct > 0 && ctx->opt_tail && !flags && narrow) {
            const int chunk = 8 << (a.xcd_chunk_log2 >= 0 ? a.xcd_chunk_log2 : 4);
            const int full = std::min(blocks, int((long long)blocks * (100 - quad_pct) / 100 + chunk - 1) / chunk * chunk);
            if (full < blocks) { a.quad_first_block = full; blocks = full + 4 * (blocks - full); }
        }
        // (a timed launch of the trial: in the learned order, in its steady state -- not the launch that learns or follows a sort)
        const bool timed = conf_sample || (a.tile_order && !learn_order && !H.trial_pending && H.lpt_age >= 2 &&
                                           (a.quad_head ? (head_ok && !H.head_disabled && H.n_head < 3 && ctx->opt_quad_head > 0) : (all_sample || H.n_base < 3)));
        if (timed) {
            for (auto& e : H.trial_evt) if (!e) HG_HIP(ctx, hipEventCreate(&e));
            HG_HIP(ctx, hipEventRecord(H.trial_evt[0], ctx->stream));
        }
        if (share_timed) {
            for (auto& e : H.share_evt[H.share_issued]) if (!e) HG_HIP(ctx, hipEventCreate(&e));
            HG_HIP(ctx, hipEventRecord(H.share_evt[H.share_issued][0], ctx->stream));
        }
        if (!launch_img(ctx->stream, blocks, narrow, img_uniform && narrow, ctx->image.general, img_slim, ctx->opt_tail != 0, flags, a))
            HG_FAIL(ctx, HAGRID_EINVAL, "traverse_grid: the traversal image of this grid has no kernel for this call (slim records need arrays below 4 GB)");
        if (timed) { HG_HIP(ctx, hipEventRecord(H.trial_evt[1], ctx->stream)); H.trial_pending = true; H.trial_with_head = a.quad_head > 0; H.trial_kind = conf_sample ? 3 : (a.quad_head > 0 ? 1 : (all_sample ? 2 : 0)); }
        if (share_timed) { HG_HIP(ctx, hipEventRecord(H.share_evt[H.share_issued][1], ctx->stream)); H.share_issued++; }
        if (a.tile_order && H.lpt_valid && !H.rot_adopted && want_rot != H.lpt_rot) { learn_order = true; H.rot_adopted = true; }
#ifdef HAGRID_DEBUG_TRACE                      // (development builds only: the decisions of the head share, tools/build_variant.sh -DHAGRID_DEBUG_TRACE)
        if (learn_order && getenv("HAGRID_TRACE_HEAD"))
            fprintf(stderr, "[head] call %llu: sort rot %d (was %d) suggestion %d quad_head %d base %.4f x%d head %.4f x%d disabled %d\n", ctx->hint_clock, want_rot, H.lpt_rot,
                    __atomic_load_n(suggest, __ATOMIC_RELAXED), a.quad_head, H.t_base, H.n_base, H.t_head, H.n_head, int(H.head_disabled));
#endif
        if (learn_order) { launch_tile_order(ctx, H, tiles, a, want_rot, suggest); H.lpt_period = H.lpt_valid ? 32 : 1; H.lpt_valid = true; H.lpt_age = 0; }
    } else if (variant == 1) {
        launch_plain(ctx->stream, num_rays, grid->small_cells != nullptr, a);
    } else {
        const int blocks = grid_blocks(num_rays, 64);
        // 32-bit offsets are enough when every gathered array is smaller than 4 GB
        const size_t tri_bytes = buffer_bytes_from(tris);
        const bool narrow = ctx->opt_narrow && tri_bytes < (size_t(1) << 32) && size_t(grid->num_cells) * 32 < (size_t(1) << 32) &&
                            a.top_xy > 0 && grid->dims[2] < (1 << 23) &&
                            size_t(grid->num_entries) * 4 < (size_t(1) << 32) && size_t(grid->num_refs) * 4 < (size_t(1) << 32);
        launch_v2(ctx->stream, blocks, grid->small_cells != nullptr, narrow, flags, a);
    }
    HG_DBG(ctx);                                   // the traversal kernel launched by one of the helpers above
    HG_HIP(ctx, hipGetLastError());
    if (publish_row_len) {                         // behind the traversal launch: nobody waits for it
        if (!H.rowlen_evt) HG_HIP(ctx, hipEventCreateWithFlags(&H.rowlen_evt, hipEventDisableTiming));
        HG_HIP(ctx, hipMemcpyAsync(ctx->mailbox + kMbxRowLen + hint_slot, ctx->dscratch + kScrRowLen + hint_slot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HG_HIP(ctx, hipEventRecord(H.rowlen_evt, ctx->stream));
        H.rowlen_pending = true;
    }
    return HAGRID_OK;
}

extern "C" int hagrid_set_option(hagrid_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return HAGRID_EINVAL;
    // The product's options: behaviour a caller may want.  Which of several equivalent code paths runs (kernel variants, record forms, dispatch
    // geometry) is not an option of the product: the parity tests and the sweep tools force those through hagrid_kat_set_option of the test
    // library (csrc/kat/kat.hip), which is not installed.
    struct { const char* name; int* dst; int lo, hi; } table[] = {
        {"expand.subset_only", &ctx->opt_expand_subset_only, 0, 1}, {"traverse.id_is_steps", &ctx->opt_id_is_steps, 0, 1},
        {"traverse.image", &ctx->opt_image, 0, 2},                  {"traverse.image_max_mb", &ctx->opt_image_max_mb, 0, 1 << 20},
        {"traverse.image_width", &ctx->opt_image_width, -1, 1 << 24}, {"traverse.tile_order", &ctx->opt_tile_order, -1, 1},
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
    launch_plain(ctx->stream, num_rays, grid->small_cells != nullptr, a);
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

/*
 * hagrid_amd.h -- C ABI of libhagrid_amd.so, the MI355X (gfx950) implementation of Hagrid's
 * irregular-grid construction and ray-traversal hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no torch types.  Every entry
 * point names the reference interface it stands behind (paths relative to the reference's src/).  The
 * C++ API of the reference (build.h / traverse.h / mem_manager.h / profile) is provided on top of this
 * file as header-only shims in include/hagrid/ -- see INTEGRATION.md.
 *
 * Conventions
 *   - Every function returning int returns HAGRID_OK (0) or a negative HAGRID_E* code;
 *     hagrid_last_error(ctx) then holds "file(line): message" (the text the reference prints before
 *     abort(), common.h:103-108).  Nothing aborts behind the ABI; the C++ shims abort like the reference.
 *   - All Tri / Ray / Hit / grid array pointers are DEVICE pointers.  Grid arrays produced by the build
 *     passes come from the context's buffer pool and are released with hagrid_mem_free (main.cpp:496-498).
 *   - All work is enqueued on the context's HIP stream (default: the null stream, like the reference's
 *     <<<...>>> launches).  Build passes synchronise with the host where they need sizes; traversal is
 *     asynchronous.
 *   - One host thread per context.  Contexts are independent (the reference keeps per-TU __constant__
 *     state and therefore allows one grid per process: traverse.cu:7-12; here the traversal constants
 *     live in the context / the grid descriptor).
 */
#ifndef HAGRID_AMD_H
#define HAGRID_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAGRID_ABI_VERSION 3   /* 3: code-path selectors left hagrid_set_option (test library); 2: hagrid_traversal_stats grew by long_list_refs (64 bytes), hagrid_grid_broadcast checks the communicator */
#define HAGRID_MAX_LEVELS 32

enum {
    HAGRID_OK = 0,
    HAGRID_EINVAL = -1,   /* bad argument */
    HAGRID_EHIP = -2,     /* a HIP runtime call failed */
    HAGRID_ENOMEM = -3,   /* device allocation failed */
    HAGRID_ERANGE = -4,   /* a count does not fit the 30-bit entry / 31-bit index space */
    HAGRID_ENODEV = -5    /* no usable gfx950 device */
};

typedef struct hagrid_ctx hagrid_ctx;

/* Grid descriptor: the reference's `struct Grid` (grid.h:48-62) as a POD.  std::vector<int> offsets
 * becomes a fixed array + count. */
typedef struct hagrid_grid {
    void* entries;        /* uint32 words, log_dim | begin << 2     (grid.h:12-20) */
    void* ref_ids;        /* int32                                  (grid.h:50)    */
    void* cells;          /* 32-byte Cell records, NULL if compressed (grid.h:23-33) */
    void* small_cells;    /* 16-byte SmallCell records or NULL      (grid.h:36-45) */
    float bbox_min[3];
    float bbox_max[3];
    int32_t dims[3];      /* top-level resolution */
    int32_t num_cells;
    int32_t num_entries;
    int32_t num_refs;
    int32_t shift;
    int32_t num_offsets;
    int32_t offsets[HAGRID_MAX_LEVELS];
} hagrid_grid;

/* Per-batch traversal counters (exact integers; the algorithmic-bytes formula of DESIGN.md). */
typedef struct hagrid_traversal_stats {
    int64_t rays, rays_hit_grid, cells, entry_words, refs, sentinels, hits;
    int64_t long_list_refs;   /* of `refs`: tested in lists of more than four ids (shorter lists are inline in the traversal image) */
} hagrid_traversal_stats;

/* Sizes the construction passes of a context went through since its last hagrid_build_grid (diagnostics: the inputs of the
 * compulsory-traffic formula of SURVEY.md 8(d) "algorithmic bytes -- build"; bench.py: roofline_build). */
#define HAGRID_MAX_MERGE_PASSES 96
typedef struct hagrid_build_counts {
    int64_t num_tris, top_cells, top_refs;                 /* N, T, R0 (references emitted at the top level, before the SAT filter) */
    int32_t num_levels, merge_passes, expand_passes, compressed;
    int64_t level_refs[HAGRID_MAX_LEVELS];                 /* R_l: references entering level l */
    int64_t level_cells[HAGRID_MAX_LEVELS];                /* C_l: cells of level l */
    int64_t level_kept[HAGRID_MAX_LEVELS];                 /* references of level l that stay in a leaf (R_l - kept = split_l) */
    int64_t build_cells, build_refs, build_entries;        /* C, R, E after build_grid */
    int64_t merge_cells[HAGRID_MAX_MERGE_PASSES];          /* cells / references entering each merge axis pass */
    int64_t merge_refs[HAGRID_MAX_MERGE_PASSES];
    int64_t merged_cells, merged_refs;                     /* after merge_grid */
    int64_t flatten_entries_in, flatten_entries_out;       /* E, E' */
    int64_t expand_cells;                                  /* C of every expand axis pass */
    int64_t compress_cells, compress_refs_out;             /* compress_grid: C and R + sentinels */
} hagrid_build_counts;

/* ---- context ----------------------------------------------------------------------------------- */

int hagrid_abi_version(void);
/* 1 when the library was built with HAGRID_DEBUG_SYNC (every kernel launch of a pass is followed by a stream synchronisation and an
 * error check that aborts with "file(line): message", the reference's DEBUG_SYNC of common.h:95-108), else 0. */
int hagrid_debug_sync_enabled(void);

/* Creates a context on HIP device `device`.  keep != 0 is MemManager's keep mode (mem_manager.h:40-42):
 * freed buffers stay allocated for reuse by later builds. */
int hagrid_ctx_create(hagrid_ctx** out, int device, int keep);
void hagrid_ctx_destroy(hagrid_ctx* ctx);
/* Launch all further work on `stream` (a hipStream_t; NULL = null stream). */
int hagrid_ctx_set_stream(hagrid_ctx* ctx, void* stream);
/* Waits until all work queued on the context's stream is done (hipStreamSynchronize): for callers that keep several contexts in
 * flight and have no HIP of their own to wait with.  The reference synchronises with cudaDeviceSynchronize / event waits in
 * its front-end (main.cpp:414-425, profile.cu:5-18). */
int hagrid_ctx_synchronize(hagrid_ctx* ctx);
const char* hagrid_last_error(const hagrid_ctx* ctx);
/* Name / compute-unit count / memory of the context's device (diagnostics, bench records). */
int hagrid_device_info(const hagrid_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* total_mem);

/* ---- MemManager backend (mem_manager.h:34-119, mem_manager.cu:6-75) -------------------------------- */
/* alloc<T>(n) -> hagrid_mem_alloc(n * sizeof(T)): best-fit reuse of a free slot, else hipMalloc. */
void* hagrid_mem_alloc(hagrid_ctx* ctx, size_t bytes);
/* free(ptr): NULL is a no-op; an untracked pointer is an error (the reference asserts). */
int hagrid_mem_free(hagrid_ctx* ctx, void* ptr);
/* copy<HST_TO_DEV | DEV_TO_HST | DEV_TO_DEV>; blocking like cudaMemcpy. */
int hagrid_mem_copy_h2d(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes);
int hagrid_mem_copy_d2h(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes);
int hagrid_mem_copy_d2d(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes);
/* zero() / one() (memset 0x00 / 0xFF). */
int hagrid_mem_zero(hagrid_ctx* ctx, void* ptr, size_t bytes);
int hagrid_mem_one(hagrid_ctx* ctx, void* ptr, size_t bytes);
size_t hagrid_mem_usage(const hagrid_ctx* ctx);
size_t hagrid_mem_max_usage(const hagrid_ctx* ctx);
/* debug_slots(): prints the slot table to stdout. */
void hagrid_mem_debug_slots(const hagrid_ctx* ctx);

/* Diagnostics for the bench record (SURVEY.md 8(d) "BW_peak": a measured device copy / triad figure from the same run):
 * streams `bytes` per array through a float4 copy (c = a) and a triad (c = a + 3 b) kernel `iters` times and returns the best
 * rate of each in GB/s (copy: 2 * bytes, triad: 3 * bytes per pass). */
int hagrid_bandwidth_probe(hagrid_ctx* ctx, size_t bytes, int iters, float* copy_gbps, float* triad_gbps);
/* The sizes recorded by the construction passes of this context (see hagrid_build_counts). */
int hagrid_get_build_counts(const hagrid_ctx* ctx, hagrid_build_counts* out);

/* ---- profile (common.h:15, profile.cu:5-18) ------------------------------------------------------- */
/* Event pair on the context's stream around arbitrary host code; end returns the elapsed ms. */
int hagrid_profile_begin(hagrid_ctx* ctx);
float hagrid_profile_end(hagrid_ctx* ctx);

/* ---- construction (build.h:17-31) ------------------------------------------------------------------ */
/* build_grid (build.cu:718-760).  tris: 48-byte Tri records on the device.  grid is overwritten. */
int hagrid_build_grid(hagrid_ctx* ctx, const void* tris, int num_tris, hagrid_grid* grid,
                      float top_density, float snd_density);
/* merge_grid (merge.cu:331-377).  On an error the grid is gone: cells and ref_ids are released and NULL in the descriptor
 * (the voxel map may already name the new cells); entries stay the caller's to free. */
int hagrid_merge_grid(hagrid_ctx* ctx, hagrid_grid* grid, float alpha);
/* flatten_grid (flatten.cu:109-175) */
int hagrid_flatten_grid(hagrid_ctx* ctx, hagrid_grid* grid);
/* expand_grid (expand.cu:199-225) */
int hagrid_expand_grid(hagrid_ctx* ctx, hagrid_grid* grid, const void* tris, int iters);
/* compress_grid (compress.cu:38-63): returns 1 when compressed, 0 when the virtual resolution does not
 * fit 16 bits (grid untouched), negative on error. */
int hagrid_compress_grid(hagrid_ctx* ctx, hagrid_grid* grid);

/* ---- the grid as one buffer: multi-GPU broadcast and save / load (no reference counterpart) -------------- */
/* Blob = header (256 bytes) + entries + cells | small_cells + ref_ids + triangles, every section 128-byte aligned.  The same bytes
 * are the file form.  Little-endian, offsets in bytes from the start of the blob. */
typedef struct hagrid_blob_header {
    uint32_t magic, version;                       /* "HGRB", 1 */
    int32_t dims[3], shift;
    int32_t num_cells, num_entries, num_refs, num_tris;
    int32_t compressed, num_offsets;
    int32_t offsets[HAGRID_MAX_LEVELS];
    float bbox_min[3], bbox_max[3];
    uint64_t off_entries, off_cells, off_refs, off_tris, total_bytes;
    uint8_t reserved[16];
} hagrid_blob_header;
/* Size of the blob of `grid` with `num_tris` triangles (0 for an incomplete grid). */
size_t hagrid_grid_blob_bytes(const hagrid_grid* grid, int num_tris);
/* Copies grid + triangles into one new pool buffer (device-to-device); release it with hagrid_mem_free. */
int hagrid_grid_pack(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, int num_tris, void** blob, size_t* bytes);
/* Turns a blob held in a pool buffer of this context into a grid IN PLACE: the buffer is split into the four arrays, which from then
 * on are ordinary pool buffers (grid->entries, grid->cells | small_cells, grid->ref_ids, *tris: each released with hagrid_mem_free, in
 * any order, like the arrays of a built grid); `blob` itself must not be freed afterwards.  Nothing is copied. */
int hagrid_grid_unpack(hagrid_ctx* ctx, void* blob, size_t bytes, hagrid_grid* grid, void** tris, int* num_tris);
/* The blob as a file. */
int hagrid_grid_save(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, int num_tris, const char* path);
int hagrid_grid_load(hagrid_ctx* ctx, const char* path, hagrid_grid* grid, void** tris, int* num_tris);
/* One process per GPU: rank `root` passes its grid and triangles (they stay untouched), every other rank receives them (grid, *tris,
 * *num_tris are outputs there, arrays owned as after hagrid_grid_unpack).  `comm` is an ncclComm_t of RCCL spanning the ranks; two
 * ncclBroadcast calls on the context's stream: the 256-byte header, then the blob, straight from / into pool memory.  RCCL is
 * looked up in the running process (the copy PyTorch or the host program loaded, else /opt/rocm/lib/librccl.so).  Blocking. */
int hagrid_grid_broadcast(hagrid_ctx* ctx, void* comm, int rank, int root, hagrid_grid* grid, void** tris, int* num_tris);

/* ---- traversal (traverse.h:11-14) ------------------------------------------------------------------- */
/* setup_traversal (traverse.cu:97-109): prepares the traversal state of `grid`.  The reference uploads constants; here
 * the constants travel with every launch and this call builds the TRAVERSAL IMAGE of the grid in the context (one per
 * context: the grid of the last call): ONE 16-byte record per cell step -- the cell's bounds as byte offsets, the reference ids of
 * lists of up to four (20-bit ids) or three (26-bit ids) inline -- so that a cell step is one dependent gather instead of
 * entry -> entry -> cell, and the reference-id gather disappears for short lists.  Three layouts (hagrid_amd/csrc/trav_image.hip):
 * grids of at most three levels get a block of records per top-level cell, indexed by the voxel -- found by arithmetic where
 * (nearly) every top-level cell has the full depth (uniform layout), through a table otherwise (table layout); every other grid
 * gets a record per voxel-map entry at the entry's index (general layout: inner entries are links to their child blocks, and the
 * kernel keeps the innermost block a ray is in -- one gather per step at any depth).  Cells whose bounds a byte cannot hold (the
 * large cells of empty space) get a wide record in the table and general layouts.  "traverse.image" = 0 builds nothing (traversal
 * reads the construction format); 1 and 2 build the image (1 was round 1-4's compact form and is kept as a value).  An image that
 * would exceed max(1 GB, 8x the entries + cells it replaces) ("traverse.image_max_mb") is not built, nor is one for a grid no layout
 * describes (a virtual resolution of 65536 per axis, reference ids beyond 26 bits, a list of 2^20 ids).
 * hagrid_traverse_grid uses the image when it is called with the same grid (same arrays, same counts); the image is
 * dropped when a construction pass runs in this context or when one of the grid's arrays is freed or overwritten through
 * this API; without an image traversal reads the construction format.  Hits are identical either way.  Synchronous (size / fit
 * read-backs); 0.3 ms and 129 MB for the 1M-triangle scene of BASELINE.md. */
int hagrid_setup_traversal(hagrid_ctx* ctx, const hagrid_grid* grid);
/* Extension: after hagrid_setup_traversal built the image of `grid` (no layout refers to the voxel map), the
 * caller may give the construction format up: entries and cells | small_cells are released to the pool and set to NULL in the
 * descriptor, the image answers for them (1M-triangle scene: 166 MB of 365 MB).  hagrid_traverse_grid[_ex] keep working with that
 * descriptor; what reads the construction format (construction passes, hagrid_traverse_grid_stats, hagrid_grid_pack, forced kernel
 * variants) is refused.  ref_ids and the triangles stay with the caller as before. */
int hagrid_grid_release_for_traversal(hagrid_ctx* ctx, hagrid_grid* grid);
/* Extension: independent batches in flight.  A launch over a SMALL batch (1M rays) keeps the machine full for the first half of its
 * time only; the second half is the drain of its last wavefronts.  A caller with independent batches (tiles of a frame, samples,
 * frames) fills that drain by giving every batch in flight its own context = its own stream (hagrid_ctx_set_stream): contexts are
 * independent.  This call lets `dst` traverse with the traversal image hagrid_setup_traversal built in `src` (same device) instead
 * of building a copy of its own -- one image in the caches, however many streams.  The image stays the property of `src`: it
 * must outlive its use in `dst`, and after the next hagrid_setup_traversal / construction pass / free of the grid in `src` the
 * share must be renewed (hagrid_setup_traversal(dst, ...) or a construction pass in `dst` ends it too).  Waits for `src`'s stream.
 * Measured, 1M-triangle scene, 1024 x 1024 primary rays: 0.177 ms per batch with one in flight, 0.118 ms with two
 * (profiles/dev_r2_inflight.txt). */
int hagrid_share_traversal(hagrid_ctx* dst, hagrid_ctx* src);
/* traverse_grid (traverse.cu:111-117): rays 32-byte Ray records, hits 16-byte Hit records.
 * hits[i].id = primitive id or -1, hits[i].t = distance (tmax on a miss), u = v = 0.  Asynchronous. */
int hagrid_traverse_grid(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                         const void* rays, void* hits, int num_rays);
/* Variants of the same walk (no separate entry point in the reference).  flags:
 *   HAGRID_TRAVERSE_ANY_HIT  a ray is finished at its FIRST accepted intersection in traversal order (cells along the ray,
 *                            references in list order): shadow / occlusion rays.  hits[i].id >= 0 exactly when the nearest-hit
 *                            traversal finds a hit; id and t are those of that first intersection.
 *   HAGRID_TRAVERSE_UVS      hits[i].u, hits[i].v = barycentrics of the hit, as the reference stores them when it is compiled
 *                            with COMPUTE_UVS (prims.h:285-288).
 * flags = 0 is hagrid_traverse_grid. */
#define HAGRID_TRAVERSE_ANY_HIT 1u
#define HAGRID_TRAVERSE_UVS 2u
int hagrid_traverse_grid_ex(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                            const void* rays, void* hits, int num_rays, uint32_t flags);
/* Same traversal, additionally: steps[i] (device int32, may be NULL) = the reference's per-ray step
 * count (traverse.cu:80,93), and *stats (host, may be NULL) = batch totals.  Synchronous. */
int hagrid_traverse_grid_stats(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris,
                               const void* rays, void* hits, int num_rays,
                               void* steps, hagrid_traversal_stats* stats);

/* Extension (no reference counterpart): spatial binning of the ray batch before traversal.  mode 0 (default): rays
 * are traversed in buffer order, as the reference does.  mode 1: each hagrid_traverse_grid call first bins the rays by
 * the position where they enter the grid (512 Morton-ordered bins, counting sort on the device) and traverses them in
 * bin order; hits are written to the rays' original slots, results are identical.  Pays for batches without spatial
 * order (random origins / directions: ~2.3x); costs a few percent on batches that are already coherent.  mode 2: automatic --
 * the binning passes run, but image-ordered batches (see "traverse.image_width") skip them, and a batch whose neighbouring
 * rays mostly share a bin (bounce rays in image order, ...) is traversed in buffer order; the decision is taken on the device,
 * the call stays asynchronous. */
int hagrid_set_ray_binning(hagrid_ctx* ctx, int mode);

/* Options: behaviour a caller may want to change; the defaults are the reference's behaviour at the tuned speed.  Keys:
 * "traverse.image": what hagrid_setup_traversal builds -- 2 (default) and 1 = the traversal image, 0 = nothing (traversal walks the construction format);
 * "traverse.image_max_mb": size limit of the image in MB (0, default = max(1 GB, 8x the arrays it replaces)); an image beyond it is not built;
 * "traverse.image_width": tile packets -- a batch in image order (ray y * w + x, as gen_rays of main.cpp:55-66 writes it) is traversed with
 *   one 8 x 8 pixel tile per wavefront instead of a 64 x 1 strip; 0 (default) = the row length w is looked for on the device (constant
 *   (origin, direction) step along a row; for batches of 4M rays or more also from the origins alone -- bounce rays in the image order of
 *   their primary hits), > 0 = w given by the caller, -1 = off.  It only steers the lane <-> ray assignment: hits never depend on it;
 * "traverse.tile_order": -1 (default) / 1 = launches over a ray buffer the context has traversed before dispatch their 8 x 8 tiles longest
 *   first, by the costs the previous launches left (the order is dropped on the device when the buffer holds other rays than the ones it was
 *   learned on; with -1 it is also held against the default order by measurement -- event pairs around launches, polled, nobody waits -- and not
 *   followed where it loses); 0 = every launch in the default order -- like the row length it only steers which wavefront takes which rays,
 *   hits never depend on it.  In either order the context MEASURES, over the first dozen launches of a launch shape and again every 1024
 *   launches, which share of the tiles starts with four lanes per ray (a scheduling choice: same hits); what it finds belongs to the scene and
 *   the shape of the launch, not to the rays, so a camera that moves keeps it;
 * "traverse.id_is_steps": 1 = hagrid_traverse_grid stores the traversal step count in Hit.id, exactly what the reference's kernel leaves
 *   there (traverse.cu:80,93) for its viewer's heat-map display (main.cpp:100-107); 0 (default) = the primitive id or -1 that ray.h:22
 *   documents; t is the same either way;
 * "expand.subset_only": 1 (default) = the reference's compiled setting; 0 = the precise expansion of expand.cu:39-57,96-127 -- this one
 *   changes the grid, not the hits.
 * Returns HAGRID_EINVAL for an unknown key or a value out of range.  (Which of several equivalent kernels / record forms / dispatch
 * geometries runs is not an option of the product: the parity tests force each of them through the test library, csrc/kat/hagrid_amd_kat.h.
 * ABI version 3: the keys "traverse.variant|narrow|image_uniform|image_slim|tail|quad_tail|tail_dual|tile_order_rounds|super_tile|
 * xcd_chunk|row_cache|lds_pad" and "merge.narrow_cells" of version 2 moved there; version 2 had moved the known-answer test hooks out of this library.) */
int hagrid_set_option(hagrid_ctx* ctx, const char* key, int value);

/* The traversal image this context holds for `grid`: format4 = { flat (1: a record per voxel; 2: the general layout, a slim record per voxel-map entry), bit 0: uniform (table-free) | bit 1: the compact table layout is held next to a much bigger uniform one (binned batches gather from it), bits per packed
 * reference id of slim 16-byte records (0: 32-byte records), bytes per record }, *image_bytes = its size (table + blocks, both layouts); either
 * pointer may be NULL.  HAGRID_EINVAL when the context holds no image of this grid. */
int hagrid_traversal_image_info(hagrid_ctx* ctx, const hagrid_grid* grid, int32_t* format4, int64_t* image_bytes);


#ifdef __cplusplus
}
#endif
#endif /* HAGRID_AMD_H */

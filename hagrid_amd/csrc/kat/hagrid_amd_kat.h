/* hagrid_amd_kat.h -- C ABI of libhagrid_amd_kat.so: known-answer hooks for the device code (tests/) and diagnostic instantiations
 * (tools/dev_*.py).  TEST INFRASTRUCTURE -- not part of the drop-in boundary (include/hagrid_amd.h) and not in the product library.
 * The hooks work on contexts of the product library (the two libraries share one process and one HIP runtime). */
#ifndef HAGRID_AMD_KAT_H
#define HAGRID_AMD_KAT_H

#include "hagrid_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- known-answer hooks for the L0 device functions (tests only; tiny launches) ---------------------- */
/* Each evaluates the named device function for n inputs (host arrays in, host arrays out). */
int hagrid_kat_intersect_prim_ray(hagrid_ctx* ctx, const void* tris, const void* rays, const int32_t* tri_index,
                                  int n, int32_t* ret, int32_t* hit_id, float* hit_t);
/* the same with COMPUTE_UVS: additionally the barycentrics */
int hagrid_kat_intersect_prim_ray_uvs(hagrid_ctx* ctx, const void* tris, const void* rays, const int32_t* tri_index,
                                      int n, int32_t* ret, int32_t* hit_id, float* hit_t, float* hit_u, float* hit_v);
int hagrid_kat_intersect_prim_cell(hagrid_ctx* ctx, const void* tris, const void* boxes, const int32_t* tri_index,
                                   int n, int32_t* ret);
int hagrid_kat_compute_range(hagrid_ctx* ctx, const int32_t* dims3, const void* grid_bb, const void* obj_bb, int n, int32_t* out6);
int hagrid_kat_compute_grid_dims(hagrid_ctx* ctx, const void* bb, const int32_t* num_prims, const float* density, int n, int32_t* out3);
int hagrid_kat_lookup_entry(hagrid_ctx* ctx, const uint32_t* entries, int num_entries, int shift, const int32_t* top_dims3,
                            const int32_t* voxels3, int n, uint32_t* out);
/* The device-wide ordered exclusive scan the construction passes use in place of cub::DeviceScan::ExclusiveSum (parallel.cuh:31-42):
 * out[i] = carry + sum of values[0..i), total = carry + sum of all; words = 1 (int) or 2 (pairs of ints, interleaved); carry_in =
 * `words` ints or NULL; lookback: 0 the three-kernel reduce-then-scan form, 1 the single-pass decoupled look-back form, 2 the same with the
 * helping path taken at the first miss; + 4: the scan runs IN PLACE (output over the input, as several passes call it). */
int hagrid_kat_scan(hagrid_ctx* ctx, const int32_t* values, int n, int words, const int32_t* carry_in, int lookback,
                    int32_t* out, int32_t* total);
/* Tile packets (see "traverse.image_width"): the row length the device finds for a ray buffer in device memory (0 = not
 * image-ordered), and the ray slot every lane of every 64-lane block gets for a batch of num_rays rays with rows of
 * row_len rays, in block dispatch order (slots: 64 * ceil(num_rays / 64) ints; values >= num_rays mark idle lanes). */
int hagrid_kat_detect_ray_rows(hagrid_ctx* ctx, const void* rays_dev, int num_rays, float bbox_diag, int32_t* row_len);
int hagrid_kat_tile_slots(hagrid_ctx* ctx, int num_rays, int row_len, int super_log2, int xcd_chunk_log2, int32_t* slots);
/* One launch of the headline kernel (table-free image with 20-bit slim records, nearest hit) in its timed instantiation: per block b
 * (64 rays) the 100 MHz wall clock at its start in times_dev[2b] and when its last lane left in times_dev[2b + 1] (the caller zeroes
 * the buffer).  row_len = image width of the batch; tail = 0 times the plain slim kernel instead; tile_order_dev (or null): block b
 * traverses the 8x8 tile tile_order_dev[b] of the default order -- an experiment on dispatch order.  tools/dev_wave_timeline.py. */
int hagrid_kat_traverse_timed(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, const void* rays, void* hits, int num_rays,
                              int row_len, int tail, unsigned long long* times_dev, const int* tile_order_dev);
/* Traversal image (hagrid_setup_traversal): the 8 words of the record that each voxel (finest-level coordinates)
 * resolves to -- u16 lo.x hi.x | lo.y hi.y | lo.z hi.z | n, bit 31 = list given by index, bit 30 = reached through a link of the general layout | the
 * reference ids (n <= 4, bit 31 clear) or the first reference index -- and the size of the image in bytes.
 * HAGRID_EINVAL when the context holds no image of this grid. */
int hagrid_kat_image_records(hagrid_ctx* ctx, const hagrid_grid* grid, const int32_t* voxels3, int n, uint32_t* records8, int64_t* image_bytes);

/* Code-path selectors for the parity tests and the sweep tools (NOT options of the product; hits never depend on them; keys of the product's
 * hagrid_set_option are passed through):
 *   "traverse.variant"   0 = choose (the traversal-image kernel when the grid has an image, else v2), 1 = the reference-shaped kernel, 2 = v2 on the
 *                        construction format, 4 = the traversal-image kernel (an error without an image); 3 no longer exists
 *   "traverse.narrow"    1 (default) = 32-bit offsets / 24-bit multiplies when every gathered array is below 4 GB, 0 = 64-bit addressing
 *   "traverse.image_uniform" / "traverse.image_slim" / "traverse.image_general"   traversal image: table-free layout when it is not much bigger (1), whatever it
 *                        costs (2), never (0) / reference ids of 20 bits where they fit (1), the 26-bit form even where 20 bits would do (2) / the general
 *                        layout where the block layouts do not fit (1), for every grid (2), never (0)
 *   "traverse.tail"      1 (default) = slim-record images are traversed by the kernel with the tail mode, 0 = one ray per lane throughout
 *   "traverse.quad_tail" per cent of the tiles, the last in dispatch order, that start with four lanes per ray; -1 (default) = by launch size
 *   "traverse.tail_dual" 1 = phase 1 of the tail kernel tests two ids of an inline list per round trip; -1 (default) = 1 unless the batch is binned
 *   "traverse.super_tile", "traverse.xcd_chunk"   tile packets: log2 of the tiles per super-tile edge (3); the XCDs take chunks of 2^k wavefronts in
 *                        turn (k >= 0), one eighth of the range each (-1), by launch size (-2, default)
 *   "traverse.row_cache" 1 (default) = a row length found for a ray buffer is kept for the next 15 calls, 0 = looked for at every call
 *   "traverse.lds_pad"   bytes of dynamic LDS per workgroup of the tail kernel (limits the resident wavefronts: profiles/dev_r3_quad_tail.txt)
 *   "merge.narrow_cells" 1 (default) = 16-byte working cell records between the merge passes when the virtual resolution is below 65536
 *   "scan.lookback"      construction scans: 1 (default) = single-pass decoupled look-back, 2 = the same helping at the first miss (the three-kernel form exists in hagrid_kat_scan only)
 *   "traverse.image_vtop" 1 (default) = the general layout of the image has a virtual top level one level below the voxel map's (eight records per top-level cell,
 *                        where look-ups that left their block start again), 0 = look-ups start at the map's top level
 *   "traverse.quad_head" 20 (default): in a learned tile order of a launch of one to five rounds the tiles that cost at least 2.0 times the median working tile -- if they
 *                        are more than a twelfth of the tiles -- start with four lanes per ray and are dispatched first; tenths of the median; 0 = never
 *   "ctx.fast_readback"  1 (default) = scalar read-backs through a publishing wavefront and a spinning host, 0 = hipMemcpyAsync + hipStreamSynchronize */
int hagrid_kat_set_option(hagrid_ctx* ctx, const char* key, int value);

/* What the context remembers about the ray buffer `rays` (traverse.hip, RayHints): out12 = { slot or -1, order valid, no order for a while (rays that keep changing), positions the order is rotated by (the
 * four-lanes-per-ray head), head share dropped by its trial, timed samples without / with the head, the share trial's choice (-1 measuring, 0 rule, 1 half), its samples (rule + 100 x half), cooldown, epoch, launches since the choice };
 * ms4 = the head trial's best launch times without / with the head, the share trial's with the rule's share / with a half.  Dev tools and tests only: nothing in the product reads it. */
/* Makes the context forget every ray buffer it has traversed (row lengths, tile orders, the trials' answers): the next call over any buffer starts from nothing.
 * Sweep tools call it between settings, so that what was measured under one setting does not decide under the next. */
int hagrid_kat_forget_hints(hagrid_ctx* ctx);
int hagrid_kat_order_state(hagrid_ctx* ctx, const void* rays, int32_t* out12, float* ms4);

#ifdef __cplusplus
}
#endif
#endif /* HAGRID_AMD_KAT_H */

/* A plain C99 translation unit using the whole C ABI the way a cgo / JNI / ctypes binding would see it:
 * proves include/hagrid_amd.h is a C header (no C++ types, extern "C" linkage). */
#include <stdio.h>
#include <string.h>
#include "hagrid_amd.h"

int run(const void* host_tris, int n, const void* host_rays, int nrays, void* host_hits) {
    hagrid_ctx* ctx = NULL;
    hagrid_grid grid;
    hagrid_traversal_stats st;
    int rc;
    memset(&grid, 0, sizeof grid);
    if (hagrid_abi_version() != HAGRID_ABI_VERSION) return -100;
    if ((rc = hagrid_ctx_create(&ctx, 0, 1)) != HAGRID_OK) return rc;
    void* tris = hagrid_mem_alloc(ctx, (size_t)n * 48);
    void* rays = hagrid_mem_alloc(ctx, (size_t)nrays * 32);
    void* hits = hagrid_mem_alloc(ctx, (size_t)nrays * 16);
    if (!tris || !rays || !hits) { fprintf(stderr, "%s\n", hagrid_last_error(ctx)); return HAGRID_ENOMEM; }
    hagrid_mem_copy_h2d(ctx, tris, host_tris, (size_t)n * 48);
    hagrid_mem_copy_h2d(ctx, rays, host_rays, (size_t)nrays * 32);
    hagrid_profile_begin(ctx);
    rc = hagrid_build_grid(ctx, tris, n, &grid, 0.12f, 2.4f);
    if (rc == HAGRID_OK) rc = hagrid_merge_grid(ctx, &grid, 0.995f);
    if (rc == HAGRID_OK) rc = hagrid_flatten_grid(ctx, &grid);
    if (rc == HAGRID_OK) rc = hagrid_expand_grid(ctx, &grid, tris, 3);
    printf("build %.3f ms, %d cells, %d refs\n", hagrid_profile_end(ctx), grid.num_cells, grid.num_refs);
    if (rc == HAGRID_OK && hagrid_compress_grid(ctx, &grid) < 0) rc = HAGRID_EHIP;
    if (rc == HAGRID_OK) rc = hagrid_setup_traversal(ctx, &grid);
    if (rc == HAGRID_OK) rc = hagrid_set_ray_binning(ctx, 1);
    if (rc == HAGRID_OK) rc = hagrid_set_option(ctx, "traverse.tile_order", 0);
    if (rc == HAGRID_OK) rc = hagrid_traverse_grid(ctx, &grid, tris, rays, hits, nrays);
    if (rc == HAGRID_OK) rc = hagrid_traverse_grid_stats(ctx, &grid, tris, rays, hits, nrays, NULL, &st);
    if (rc == HAGRID_OK) {      /* a second context traversing with the first one's traversal image: same hits, written last */
        hagrid_ctx* ctx2 = NULL;
        rc = hagrid_ctx_create(&ctx2, 0, 0);
        if (rc == HAGRID_OK) rc = hagrid_share_traversal(ctx2, ctx);
        if (rc == HAGRID_OK) rc = hagrid_mem_zero(ctx, hits, (size_t)nrays * 16);
        if (rc == HAGRID_OK) rc = hagrid_ctx_synchronize(ctx);
        if (rc == HAGRID_OK) rc = hagrid_traverse_grid(ctx2, &grid, tris, rays, hits, nrays);
        if (rc == HAGRID_OK) rc = hagrid_ctx_synchronize(ctx2);
        if (rc != HAGRID_OK && ctx2) fprintf(stderr, "%s\n", hagrid_last_error(ctx2));
        if (ctx2) hagrid_ctx_destroy(ctx2);
    }
    if (rc == HAGRID_OK) rc = hagrid_mem_copy_d2h(ctx, host_hits, hits, (size_t)nrays * 16);
    if (rc != HAGRID_OK) fprintf(stderr, "%s\n", hagrid_last_error(ctx));
    hagrid_mem_free(ctx, grid.entries); hagrid_mem_free(ctx, grid.cells); hagrid_mem_free(ctx, grid.small_cells); hagrid_mem_free(ctx, grid.ref_ids);
    hagrid_mem_free(ctx, tris); hagrid_mem_free(ctx, rays); hagrid_mem_free(ctx, hits);
    printf("usage after free %zu, peak %zu, hits %lld\n", hagrid_mem_usage(ctx), hagrid_mem_max_usage(ctx), (long long)st.hits);
    hagrid_ctx_destroy(ctx);
    return rc;
}

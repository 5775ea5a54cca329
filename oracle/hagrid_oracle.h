/*
 * hagrid_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never measured
 * as the product).  Plain-C restatement of the irregular-grid build + traversal hot path of
 * cg-saarland/hagrid.  Every function cites the reference file:line it follows (paths relative
 * to the reference checkout's src/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Pinning: the L0 arithmetic (intersect_prim_ray, tri/box SAT, compute_range,
 * compute_grid_dims, lookup_entry, ...) is checked bit-for-bit against golden vectors produced
 * by the reference's own headers (oracle/_ref, tests/golden/).  The reference holds no tests or
 * fixtures for the device passes (build/merge/flatten/expand/compress/traverse kernels) and
 * those .cu files cannot be compiled here (nvcc and the un-vendored NVlabs/cub are absent), so
 * for the multi-kernel passes the oracle is pinned indirectly: traversal of the oracle-built grid
 * must reproduce a brute-force nearest hit computed with the REFERENCE's intersect_prim_ray.
 * Grid *structure* parity with a real CUDA run is unpinned ("parity unpinned").
 *
 * Floating point: compile with -ffp-contract=off and without fast-math.  All float->int casts
 * are C truncations of in-range values.
 */
#ifndef HAGRID_ORACLE_H
#define HAGRID_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ovec3;
typedef struct { int x, y, z; } oivec3;

/* prims.h:13-25  (48 B) */
typedef struct { ovec3 v0; float nx; ovec3 e1; float ny; ovec3 e2; float nz; } OTri;
/* ray.h:9-20 (32 B), ray.h:22-33 (16 B) */
typedef struct { ovec3 org; float tmin; ovec3 dir; float tmax; } ORay;
typedef struct { int id; float t, u, v; } OHit;
/* bbox.h:10-14 (32 B) */
typedef struct { ovec3 min; int pad0; ovec3 max; int pad1; } OBBox;
/* grid.h:12-20: raw word = log_dim | begin << 2 */
typedef uint32_t OEntry;
/* grid.h:23-33 (32 B) */
typedef struct { oivec3 min; int begin; oivec3 max; int end; } OCell;
/* grid.h:36-45 (16 B) */
typedef struct { uint16_t min[3]; uint16_t max[3]; int begin; } OSmallCell;
/* grid.h:64-75 */
typedef struct { int lx, ly, lz, hx, hy, hz; } ORange;

#define ORC_MAX_LEVELS 32

/* grid.h:48-62 flattened to a POD (std::vector offsets -> fixed array) */
typedef struct {
    OEntry*     entries;
    int*        ref_ids;
    OCell*      cells;
    OSmallCell* small_cells;
    OBBox       bbox;
    oivec3      dims;
    int         num_cells;
    int         num_entries;
    int         num_refs;
    int         shift;
    int         num_offsets;
    int         offsets[ORC_MAX_LEVELS];
} OGrid;

/* exact integer counters for the algorithmic-bytes formula */
typedef struct {
    int64_t rays;
    int64_t rays_hit_grid;   /* rays that pass the grid-box test */
    int64_t cells;           /* visited cells (one lookup + one cell load each) */
    int64_t entry_words;     /* voxel-map words dereferenced, sum of L_c */
    int64_t refs;            /* tested references (one ref id + one Tri each) */
    int64_t sentinels;       /* sentinel words read (compressed grids only) */
    int64_t hits;            /* rays with id >= 0 */
    int64_t long_list_refs;  /* of `refs`: those tested in lists of more than four ids (the traversal image keeps shorter lists inline) */
} OStats;

/* ---- L0 -------------------------------------------------------------------------------- */
float    orc_safe_rcp(float x);                       /* common.h:40-42 */
float    orc_prodsign(float x, float y);              /* common.h:45-47 */
int      orc_ilog2_i32(int t);                        /* common.h:81-93 */
OEntry   orc_make_entry(uint32_t log_dim, uint32_t begin); /* grid.h:78-81 */
float    orc_cbrtf(float x);                          /* deterministic cbrt shared with the HIP side */
void     orc_tri_bbox(const OTri* tri, OBBox* out);   /* prims.h:27-31 */
void     orc_compute_range(const oivec3* dims, const OBBox* grid_bb, const OBBox* obj_bb, ORange* out); /* grid.h:84-93 */
void     orc_compute_grid_dims(const OBBox* bb, int num_prims, float density, oivec3* out);            /* grid.h:96-101 */
uint32_t orc_lookup_entry(const OEntry* entries, int shift, const oivec3* top_dims, const oivec3* voxel, int* words); /* grid.h:103-116 */
int      orc_intersect_prim_cell(const OTri* tri, const OBBox* box);            /* prims.h:161-264 */
int      orc_intersect_prim_ray(const OTri* tri, const ORay* ray, int id, OHit* hit); /* prims.h:266-295 */
int      orc_intersect_prim_ray_uv(const OTri* tri, const ORay* ray, int id, OHit* hit); /* same with COMPUTE_UVS (prims.h:285-288) */

/* ---- passes (build.h:17-31, traverse.h:11-14) -------------------------------------------- */
void orc_grid_init(OGrid* g);
void orc_grid_free(OGrid* g);
int  orc_build_grid(const OTri* tris, int num_tris, OGrid* grid, float top_density, float snd_density);
int  orc_merge_grid(OGrid* grid, float alpha);
int  orc_flatten_grid(OGrid* grid);
int  orc_expand_grid(OGrid* grid, const OTri* tris, int iters);              /* subset_only = true, expand.cu:159 */
int  orc_expand_grid_ex(OGrid* grid, const OTri* tris, int iters, int subset_only);   /* false: expand.cu:39-57,96-127 */
int  orc_compress_grid(OGrid* grid);   /* 1 on success, 0 if dims do not fit 16 bits */

/* traversal: hits[i].id = primitive id or -1 (decision SURVEY.md 8(b)); steps (optional) receives
 * the reference's step counter (traverse.cu:80,93) */
void orc_traverse_grid(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits,
                       int64_t num_rays, int* steps, OStats* stats);
/* same over nthreads pthreads that take chunks of 4096 consecutive rays from a shared counter (CPU baseline) */
void orc_traverse_grid_mt(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits,
                          int64_t num_rays, int nthreads, OStats* stats);
/* same, thread t pinned to hardware thread cpus[t % num_cpus] (oracle.py: one per physical core; NULL: not pinned) */
void orc_traverse_grid_pinned(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits,
                              int64_t num_rays, int nthreads, const int* cpus, int num_cpus, OStats* stats);
/* traversal variants: ORC_ANY_HIT = the walk stops at the first accepted intersection (shadow rays:
 * id/t of that intersection), ORC_UVS = hits carry the barycentrics of prims.h:285-288 */
#define ORC_ANY_HIT 1u
#define ORC_UVS 2u
void orc_traverse_grid_ex(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits, int64_t num_rays, int nthreads, unsigned flags);
/* dev analysis: lens[i*cap + s] = list length of the s-th cell ray i visits (clamped to 255), num_cells[i] = cells visited */
void orc_traverse_trace(const OGrid* grid, const OTri* tris, const ORay* rays, int64_t num_rays, int cap, unsigned char* lens, int* num_cells,
                        int ids_cap, int* ids /* may be NULL: tested reference ids in order */, int* num_ids);
/* dev analysis: as orc_traverse_trace, plus vox[(i * cap + s) * 3 ..] = the voxel of the s-th look-up of ray i */
void orc_traverse_trace_voxels(const OGrid* grid, const OTri* tris, const ORay* rays, int64_t num_rays, int cap, unsigned char* lens, int* num_cells, short* vox);
/* nearest hit over all triangles with orc_intersect_prim_ray, ascending id order */
void orc_brute_force(const OTri* tris, int num_tris, const ORay* rays, OHit* hits, int64_t num_rays, int nthreads);

/* structural invariants of a finished grid; returns 0 if ok, else a negative code; msg gets text */
int  orc_check_grid(const OGrid* grid, const OTri* tris, int num_tris, int check_coverage, char* msg, int msg_len);

/* "as-CUDA" structure mode (test / analysis only): bit 0 = the rejected half of the partition in reverse order (CUB), bit 1 = expand
 * leaves unprocessed cells stale.  0 (default) = the documented intent, which the product implements. */
void orc_set_cuda_quirks(int mask);
int orc_get_cuda_quirks(void);

#ifdef __cplusplus
}
#endif
#endif

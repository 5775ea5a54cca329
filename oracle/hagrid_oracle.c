/*
 * hagrid_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see hagrid_oracle.h).
 *
 * Plain-C, scalar restatement of the reference algorithm.  Kernels are restated as loops over
 * the thread id; warp-cooperative sections (build.cu:106-135, merge.cu:245-270) are restated by
 * their sequential meaning.  Citations are reference src/ file:line.
 *
 * Deliberate, documented differences from a literal reading of the CUDA sources:
 *  D1  partition keeps BOTH halves in input order (the reference's CUB version is not vendored;
 *      CUB's DevicePartition reverses the rejected half, which would give descending ref lists
 *      on odd levels although merge.cu:57 / expand.cu:20 assume ascending lists).
 *  D2  expand copies unprocessed cells through to the new buffer (expand.cu:154-155,181 leaves
 *      them stale).
 *  D3  grid.shift is the cell-coordinate shift max(log_dims) (build.cu:508), padded offsets,
 *      instead of levels.size()-1 (build.cu:706); the two agree whenever the deepest level is
 *      reached.
 *  D4  Hit.id carries the primitive id (-1 on a miss), steps are returned separately
 *      (traverse.cu:93 overwrites id with the step count).
 *  D5  cbrtf is replaced by a deterministic double-precision Newton cbrt so that the CPU and the
 *      GPU agree on integer grid dimensions.
 *
 * orc_set_cuda_quirks(mask) switches D1 (bit 0) and D2 (bit 1) to what a literal CUDA run with CUB would do, as far as the
 * sources and CUB's documented behaviour say -- "as-CUDA" structure mode, only to QUANTIFY what D1 / D2 change (cells, references,
 * traversal steps; tests/test_oracle_golden.py, tools/dev_cuda_quirks.py); hits are the same either way.  The product implements
 * the documented intent (mask 0).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE          /* pthread_setaffinity_np, cpu_set_t (the all-cores CPU baseline pins its threads) */
#endif
#include "hagrid_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_cuda_quirks = 0;          /* bit 0: CUB's reversed rear partition (D1), bit 1: expand's stale double buffer (D2) */
void orc_set_cuda_quirks(int mask) { g_cuda_quirks = mask; }
int orc_get_cuda_quirks(void) { return g_cuda_quirks; }

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers: hagrid::min/max are `a < b ? a : b` / `a > b ? a : b` (common.h:23-25) */

static inline float fmin_t(float a, float b) { return a < b ? a : b; }
static inline float fmax_t(float a, float b) { return a > b ? a : b; }
static inline int   imin(int a, int b) { return a < b ? a : b; }
static inline int   imax(int a, int b) { return a > b ? a : b; }

static inline ovec3 v3(float x, float y, float z) { ovec3 r = { x, y, z }; return r; }
static inline ovec3 v3_add(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ovec3 v3_sub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ovec3 v3_mul(ovec3 a, ovec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ovec3 v3_div(ovec3 a, ovec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline ovec3 v3_scale(ovec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline ovec3 v3_min(ovec3 a, ovec3 b) { return v3(fmin_t(a.x, b.x), fmin_t(a.y, b.y), fmin_t(a.z, b.z)); }
static inline ovec3 v3_max(ovec3 a, ovec3 b) { return v3(fmax_t(a.x, b.x), fmax_t(a.y, b.y), fmax_t(a.z, b.z)); }
/* vec.h:100: a.x*b.x + a.y*b.y + a.z*b.z, left associative */
static inline float v3_dot(ovec3 a, ovec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec.h:104-109 */
static inline ovec3 v3_cross(ovec3 a, ovec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline ovec3 v3_from_i(oivec3 a) { return v3((float)a.x, (float)a.y, (float)a.z); }
static inline oivec3 iv3(int x, int y, int z) { oivec3 r = { x, y, z }; return r; }
static inline int iget(oivec3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
static inline void iset(oivec3* v, int axis, int val) { if (axis == 0) v->x = val; else if (axis == 1) v->y = val; else v->z = val; }
static inline float fget(ovec3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define ENTRY_LOG_DIM(e) ((e) & 3u)
#define ENTRY_BEGIN(e)   ((e) >> 2)

/* ------------------------------------------------------------------------------------------ */
/* L0 */

/* common.h:40-42 */
float orc_safe_rcp(float x) { return x != 0 ? 1.0f / x : copysignf(u2f(0x7f800000u), x); }

/* common.h:45-47 */
float orc_prodsign(float x, float y) { return u2f(f2u(x) ^ (f2u(y) & 0x80000000u)); }

/* common.h:81-93 with T = int: 5 bisection steps over [0, 32] */
int orc_ilog2_i32(int t) {
    unsigned a = 0, b = 32;
    uint32_t all = 0xFFFFFFFFu, ut = (uint32_t)t;
    for (int i = 0; i < 5; i++) {
        unsigned m = (a + b) / 2;
        uint32_t mask = all << m;
        if (ut & mask) a = m + 1; else b = m;
    }
    return (int)a;
}

/* grid.h:78-81 */
OEntry orc_make_entry(uint32_t log_dim, uint32_t begin) { return (log_dim & 3u) | (begin << 2); }

/* D5: deterministic cube root.  Only +,-,*,/ on IEEE doubles and integer ops, so the HIP device
 * code (same operation sequence, contraction off) produces the same bits. */
float orc_cbrtf(float v) {
    if (v == 0.0f || v != v) return v;
    double x = fabs((double)v);
    if (x > 1.7e308) return v; /* inf */
    uint64_t i; memcpy(&i, &x, 8);
    i = i / 3 + 0x2A9F7893782DA1CEull;
    double y; memcpy(&y, &i, 8);
    for (int k = 0; k < 6; k++) {
        double y2 = y * y;
        y = y - (y2 * y - x) / (3.0 * y2);
    }
    float r = (float)y;
    return v < 0 ? -r : r;
}

/* prims.h:27-31 */
void orc_tri_bbox(const OTri* tri, OBBox* out) {
    ovec3 v1 = v3_sub(tri->v0, tri->e1);
    ovec3 v2 = v3_add(tri->v0, tri->e2);
    out->min = v3_min(tri->v0, v3_min(v1, v2));
    out->max = v3_max(tri->v0, v3_max(v1, v2));
    out->pad0 = 0; out->pad1 = 0;
}

/* grid.h:84-93 */
void orc_compute_range(const oivec3* dims, const OBBox* g, const OBBox* o, ORange* r) {
    ovec3 inv = v3_div(v3_from_i(*dims), v3_sub(g->max, g->min));
    r->lx = imax((int)((o->min.x - g->min.x) * inv.x), 0);
    r->ly = imax((int)((o->min.y - g->min.y) * inv.y), 0);
    r->lz = imax((int)((o->min.z - g->min.z) * inv.z), 0);
    r->hx = imin((int)((o->max.x - g->min.x) * inv.x), dims->x - 1);
    r->hy = imin((int)((o->max.y - g->min.y) * inv.y), dims->y - 1);
    r->hz = imin((int)((o->max.z - g->min.z) * inv.z), dims->z - 1);
}
static inline int range_size(const ORange* r) { /* grid.h:74 */
    return (r->hx - r->lx + 1) * (r->hy - r->ly + 1) * (r->hz - r->lz + 1);
}

/* grid.h:96-101 (cbrtf -> orc_cbrtf, D5) */
void orc_compute_grid_dims(const OBBox* bb, int num_prims, float density, oivec3* out) {
    ovec3 e = v3_sub(bb->max, bb->min);
    float volume = e.x * e.y * e.z;
    float ratio = orc_cbrtf(density * num_prims / volume);
    out->x = imax(1, (int)(e.x * ratio));
    out->y = imax(1, (int)(e.y * ratio));
    out->z = imax(1, (int)(e.z * ratio));
}

/* grid.h:103-116; *words (optional) receives the number of voxel-map words dereferenced */
uint32_t orc_lookup_entry(const OEntry* entries, int shift, const oivec3* dims, const oivec3* voxel, int* words) {
    OEntry entry = entries[(voxel->x >> shift) + dims->x * ((voxel->y >> shift) + dims->y * (voxel->z >> shift))];
    uint32_t log_dim = ENTRY_LOG_DIM(entry), d = log_dim;
    int w = 1;
    while (log_dim) {
        uint32_t begin = ENTRY_BEGIN(entry);
        int mask = (1 << log_dim) - 1;
        int s = (int)(shift - d);
        int kx = (voxel->x >> s) & mask, ky = (voxel->y >> s) & mask, kz = (voxel->z >> s) & mask;
        entry = entries[begin + kx + ((ky + (kz << log_dim)) << log_dim)];
        log_dim = ENTRY_LOG_DIM(entry);
        d += log_dim;
        w++;
    }
    if (words) *words = w;
    return ENTRY_BEGIN(entry);
}

/* prims.h:161-181 (non-CUDA-7 branch) */
static int plane_overlap_box(ovec3 n, float d, ovec3 mn, ovec3 mx) {
    ovec3 first = v3(n.x > 0 ? mn.x : mx.x, n.y > 0 ? mn.y : mx.y, n.z > 0 ? mn.z : mx.z);
    ovec3 last  = v3(n.x <= 0 ? mn.x : mx.x, n.y <= 0 ? mn.y : mx.y, n.z <= 0 ? mn.z : mx.z);
    float d0 = v3_dot(n, first) - d;
    float d1 = v3_dot(n, last) - d;
    return d1 * d0 <= 0.0f;
}
/* prims.h:183-190 */
static int axis_test_x(ovec3 h, ovec3 e, ovec3 f, ovec3 a, ovec3 b) {
    float p0 = e.y * a.z - e.z * a.y, p1 = e.y * b.z - e.z * b.y;
    float rad = f.z * h.y + f.y * h.z;
    return (fminf(p0, p1) > rad) | (fmaxf(p0, p1) < -rad);
}
/* prims.h:192-199 */
static int axis_test_y(ovec3 h, ovec3 e, ovec3 f, ovec3 a, ovec3 b) {
    float p0 = e.z * a.x - e.x * a.z, p1 = e.z * b.x - e.x * b.z;
    float rad = f.z * h.x + f.x * h.z;
    return (fminf(p0, p1) > rad) | (fmaxf(p0, p1) < -rad);
}
/* prims.h:201-208 */
static int axis_test_z(ovec3 h, ovec3 e, ovec3 f, ovec3 a, ovec3 b) {
    float p0 = e.x * a.y - e.y * a.x, p1 = e.x * b.y - e.y * b.x;
    float rad = f.y * h.x + f.x * h.y;
    return (fminf(p0, p1) > rad) | (fmaxf(p0, p1) < -rad);
}
/* prims.h:210-260 with bounds_check=false, cross_axes=true (prims.h:262-264) */
static int tri_box(ovec3 v0, ovec3 e1, ovec3 e2, ovec3 n, ovec3 mn, ovec3 mx) {
    if (!plane_overlap_box(n, v3_dot(v0, n), mn, mx)) return 0;
    ovec3 v1 = v3_sub(v0, e1), v2 = v3_add(v0, e2);
    ovec3 center = v3_scale(v3_add(mx, mn), 0.5f);
    ovec3 half   = v3_scale(v3_sub(mx, mn), 0.5f);
    ovec3 w0 = v3_sub(v0, center), w1 = v3_sub(v1, center), w2 = v3_sub(v2, center);
    ovec3 f1 = v3(fabsf(e1.x), fabsf(e1.y), fabsf(e1.z));
    if (axis_test_x(half, e1, f1, w0, w2) || axis_test_y(half, e1, f1, w0, w2) || axis_test_z(half, e1, f1, w1, w2)) return 0;
    ovec3 f2 = v3(fabsf(e2.x), fabsf(e2.y), fabsf(e2.z));
    if (axis_test_x(half, e2, f2, w0, w1) || axis_test_y(half, e2, f2, w0, w1) || axis_test_z(half, e2, f2, w1, w2)) return 0;
    ovec3 e3 = v3_add(e1, e2);
    ovec3 f3 = v3(fabsf(e3.x), fabsf(e3.y), fabsf(e3.z));
    if (axis_test_x(half, e3, f3, w0, w2) || axis_test_y(half, e3, f3, w0, w2) || axis_test_z(half, e3, f3, w0, w1)) return 0;
    return 1;
}
int orc_intersect_prim_cell(const OTri* t, const OBBox* b) {
    return tri_box(t->v0, t->e1, t->e2, v3(t->nx, t->ny, t->nz), b->min, b->max);
}

/* prims.h:266-295 (COMPUTE_UVS undefined) */
int orc_intersect_prim_ray(const OTri* tri, const ORay* ray, int id, OHit* hit) {
    ovec3 n = v3(tri->nx, tri->ny, tri->nz);
    ovec3 c = v3_sub(tri->v0, ray->org);
    ovec3 r = v3_cross(ray->dir, c);
    float det = v3_dot(n, ray->dir);
    float abs_det = fabsf(det);
    float u = orc_prodsign(v3_dot(r, tri->e2), det);
    float v = orc_prodsign(v3_dot(r, tri->e1), det);
    float w = abs_det - u - v;
    float eps = 1e-9f;
    if (u >= -eps && v >= -eps && w >= -eps) {
        float t = orc_prodsign(v3_dot(n, c), det);
        if (t >= abs_det * ray->tmin && abs_det * ray->tmax > t) {
            float inv_det = 1.0f / abs_det;
            hit->t = t * inv_det;
            hit->id = id;
            return 1;
        }
    }
    return 0;
}

/* prims.h:266-295 with COMPUTE_UVS defined (:285-288): the barycentrics of the accepted hit are stored as well */
int orc_intersect_prim_ray_uv(const OTri* tri, const ORay* ray, int id, OHit* hit) {
    ovec3 n = v3(tri->nx, tri->ny, tri->nz);
    ovec3 c = v3_sub(tri->v0, ray->org);
    ovec3 r = v3_cross(ray->dir, c);
    float det = v3_dot(n, ray->dir);
    float abs_det = fabsf(det);
    float u = orc_prodsign(v3_dot(r, tri->e2), det);
    float v = orc_prodsign(v3_dot(r, tri->e1), det);
    float w = abs_det - u - v;
    float eps = 1e-9f;
    if (u >= -eps && v >= -eps && w >= -eps) {
        float t = orc_prodsign(v3_dot(n, c), det);
        if (t >= abs_det * ray->tmin && abs_det * ray->tmax > t) {
            float inv_det = 1.0f / abs_det;
            hit->t = t * inv_det;
            hit->u = u * inv_det;
            hit->v = v * inv_det;
            hit->id = id;
            return 1;
        }
    }
    return 0;
}

/* traversal mode of the calling thread (orc_traverse_grid_ex): ORC_ANY_HIT stops at the first accepted intersection
 * (shadow rays), ORC_UVS stores the barycentrics (prims.h:285-288) */
static __thread unsigned g_mode = 0;

/* ------------------------------------------------------------------------------------------ */
/* grid lifetime */

void orc_grid_init(OGrid* g) { memset(g, 0, sizeof(*g)); }
void orc_grid_free(OGrid* g) {
    free(g->entries); free(g->ref_ids); free(g->cells); free(g->small_cells);
    g->entries = NULL; g->ref_ids = NULL; g->cells = NULL; g->small_cells = NULL;
}

static void* xmalloc(size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "hagrid_oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void* xcalloc(size_t n, size_t s) {
    void* p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "hagrid_oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* build_grid: build.cu:470-760 */

typedef struct {              /* build.cu:14-35 */
    int* ref_ids; int* cell_ids;
    int num_refs; int num_kept;
    OCell* cells; OEntry* entries; int num_cells;
} Level;

typedef struct {              /* build.cu:37-40 __constant__ state */
    oivec3 dims; OBBox bbox; ovec3 cell_size; int shift;
} BuildConsts;

static OBBox cell_world_box(const BuildConsts* k, const OCell* c) { /* build.cu:150-151 */
    OBBox b;
    b.min = v3_add(k->bbox.min, v3_mul(v3_from_i(c->min), k->cell_size));
    b.max = v3_add(k->bbox.min, v3_mul(v3_from_i(c->max), k->cell_size));
    b.pad0 = b.pad1 = 0;
    return b;
}

/* build.cu:160-216 */
static int split_mask(const BuildConsts* k, const OCell* cell, const OTri* prim) {
    ovec3 cmin = v3_add(k->bbox.min, v3_mul(k->cell_size, v3_from_i(cell->min)));
    ovec3 cmax = v3_add(k->bbox.min, v3_mul(k->cell_size, v3_from_i(cell->max)));
    ovec3 mid  = v3_scale(v3_add(cmin, cmax), 0.5f);
    int mask = 0xFF;
    OBBox rb; orc_tri_bbox(prim, &rb);
    if (rb.min.x > cmax.x || rb.max.x < cmin.x) mask = 0;
    if (rb.min.x > mid.x) mask &= 0xAA;
    if (rb.max.x < mid.x) mask &= 0x55;
    if (rb.min.y > cmax.y || rb.max.y < cmin.y) mask = 0;
    if (rb.min.y > mid.y) mask &= 0xCC;
    if (rb.max.y < mid.y) mask &= 0x33;
    if (rb.min.z > cmax.z || rb.max.z < cmin.z) mask = 0;
    if (rb.min.z > mid.z) mask &= 0xF0;
    if (rb.max.z < mid.z) mask &= 0x0F;
    /* build.cu:200-213: visit set bits in ascending order (with mask==0 the reference starts at
     * i=-1, whose test result is discarded because `mask &= ~(1 << -1)` cannot set a bit) */
    for (int i = 0; i < 8; i++) {
        if (!(mask & (1 << i))) continue;
        OBBox b;
        b.min = v3(i & 1 ? mid.x : cmin.x, i & 2 ? mid.y : cmin.y, i & 4 ? mid.z : cmin.z);
        b.max = v3(i & 1 ? cmax.x : mid.x, i & 2 ? cmax.y : mid.y, i & 4 ? cmax.z : mid.z);
        if (!orc_intersect_prim_cell(prim, &b)) mask &= ~(1 << i);
    }
    return mask;
}

int orc_build_grid(const OTri* tris, int num_tris, OGrid* grid, float top_density, float snd_density) {
    BuildConsts k;
    /* build.cu:723-727: bboxes + reduction from BBox::empty() */
    OBBox* bboxes = (OBBox*)xmalloc(sizeof(OBBox) * (size_t)(num_tris + 1));
    OBBox gb; gb.min = v3(FLT_MAX, FLT_MAX, FLT_MAX); gb.max = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX); gb.pad0 = gb.pad1 = 0;
    for (int i = 0; i < num_tris; i++) {
        orc_tri_bbox(&tris[i], &bboxes[i]);
        gb.min = v3_min(gb.min, bboxes[i].min);
        gb.max = v3_max(gb.max, bboxes[i].max);
    }
    /* build.cu:728-737 */
    oivec3 dims; orc_compute_grid_dims(&gb, num_tris, top_density, &dims);
    dims.x = dims.x % 2 ? dims.x + 1 : dims.x;
    dims.y = dims.y % 2 ? dims.y + 1 : dims.y;
    dims.z = dims.z % 2 ? dims.z + 1 : dims.z;
    ovec3 ext = v3_sub(gb.max, gb.min);
    gb.min = v3_sub(gb.min, v3_scale(ext, 0.001f));
    gb.max = v3_add(gb.max, v3_scale(ext, 0.001f));
    k.dims = dims; k.bbox = gb;

    int num_top = dims.x * dims.y * dims.z;

    /* ---- first_build_iter, build.cu:470-525 ---- */
    /* count_new_refs build.cu:57-66 + scan build.cu:487 */
    int64_t* start_emit = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)(num_tris + 1));
    int64_t total = 0;
    for (int i = 0; i < num_tris; i++) {
        ORange r; orc_compute_range(&dims, &gb, &bboxes[i], &r);
        start_emit[i] = total;
        total += imax(0, range_size(&r));
    }
    start_emit[num_tris] = total;
    if (total > 0x3fffffff) { free(bboxes); free(start_emit); return -1; }
    int R0 = (int)total;
    int* ref_ids  = (int*)xmalloc(sizeof(int) * (size_t)R0);
    int* cell_ids = (int*)xmalloc(sizeof(int) * (size_t)R0);
    /* emit_new_refs build.cu:69-136: x fastest, then y, then z, primitive-major */
    for (int i = 0; i < num_tris; i++) {
        int64_t s = start_emit[i], e = start_emit[i + 1];
        if (s >= e) continue;
        ORange r; orc_compute_range(&dims, &gb, &bboxes[i], &r);
        int x = r.lx, y = r.ly, z = r.lz;
        for (int64_t cur = s; cur < e; cur++) {
            ref_ids[cur] = i;
            cell_ids[cur] = x + dims.x * (y + dims.y * z);
            x++;
            if (x > r.hx) { x = r.lx; y++; }
            if (y > r.hy) { y = r.ly; z++; }
        }
    }
    free(start_emit);
    free(bboxes);
    /* count_refs_per_cell build.cu:246-253 (before filtering) */
    int* refs_per_cell = (int*)xcalloc((size_t)num_top, sizeof(int));
    for (int i = 0; i < R0; i++) if (cell_ids[i] >= 0) refs_per_cell[cell_ids[i]]++;
    /* compute_log_dims build.cu:256-270, reduce max build.cu:508 */
    int* log_dims = (int*)xmalloc(sizeof(int) * (size_t)(num_top + 1));
    int shift = 0;
    {
        ovec3 cext = v3_div(v3_sub(gb.max, gb.min), v3_from_i(dims));
        OBBox cb; cb.min = v3(0, 0, 0); cb.max = cext; cb.pad0 = cb.pad1 = 0;
        for (int i = 0; i < num_top; i++) {
            oivec3 d; orc_compute_grid_dims(&cb, refs_per_cell[i], snd_density, &d);
            int max_dim = imax(d.x, imax(d.y, d.z));
            int log_dim = 31 - __builtin_clz((unsigned)max_dim);
            log_dim = (1 << log_dim) < max_dim ? log_dim + 1 : log_dim;
            log_dims[i] = log_dim;
            shift = imax(shift, log_dim);
        }
    }
    free(refs_per_cell);
    k.shift = shift;
    {   /* build.cu:509 */
        oivec3 vd = iv3(dims.x << shift, dims.y << shift, dims.z << shift);
        k.cell_size = v3_div(v3_sub(gb.max, gb.min), v3_from_i(vd));
    }
    /* emit_top_cells build.cu:332-351 */
    OCell* cells = (OCell*)xmalloc(sizeof(OCell) * (size_t)num_top);
    for (int id = 0; id < num_top; id++) {
        int x = id % dims.x, y = (id / dims.x) % dims.y, z = id / (dims.x * dims.y);
        int inc = 1 << shift;
        x <<= shift; y <<= shift; z <<= shift;
        cells[id].min = iv3(x, y, z); cells[id].max = iv3(x + inc, y + inc, z + inc);
        cells[id].begin = 0; cells[id].end = 0;
    }
    OEntry* entries = (OEntry*)xcalloc((size_t)num_top + 1, sizeof(OEntry));
    /* filter_refs build.cu:139-157 */
    for (int i = 0; i < R0; i++) {
        OBBox b = cell_world_box(&k, &cells[cell_ids[i]]);
        if (!orc_intersect_prim_cell(&tris[ref_ids[i]], &b)) { cell_ids[i] = -1; ref_ids[i] = -1; }
    }

    Level levels[ORC_MAX_LEVELS];
    int num_levels = 0;
    levels[num_levels++] = (Level){ ref_ids, cell_ids, R0, R0, cells, entries, num_top };

    /* ---- build_iter, build.cu:527-619 ---- */
    for (;;) {
        Level* L = &levels[num_levels - 1];
        int num_refs = L->num_refs, num_cells = L->num_cells;
        /* compute_dims build.cu:286-302 */
        for (int i = 0; i < num_refs; i++) {
            int c = L->cell_ids[i];
            if (c < 0) continue;
            oivec3 m = L->cells[c].min;
            int top = (m.x >> shift) + dims.x * ((m.y >> shift) + dims.y * (m.z >> shift));
            L->entries[c] = orc_make_entry((uint32_t)imin(log_dims[top], 1), 0);
        }
        /* update_log_dims build.cu:273-278 */
        for (int i = 0; i < num_top; i++) log_dims[i] = imax(0, log_dims[i] - 1);
        /* scan build.cu:557-559 + update_entries build.cu:317-329 */
        int num_new_cells = 0;
        for (int i = 0; i < num_cells; i++) {
            uint32_t ld = ENTRY_LOG_DIM(L->entries[i]);
            L->entries[i] = orc_make_entry(ld, ld ? (uint32_t)num_new_cells : (uint32_t)i);
            num_new_cells += ld ? 8 : 0;
        }
        /* mark_kept_refs build.cu:305-314 + partition build.cu:568-569 (D1: both halves stable) */
        int* nref = (int*)xmalloc(sizeof(int) * (size_t)num_refs);
        int* ncel = (int*)xmalloc(sizeof(int) * (size_t)num_refs);
        int num_kept = 0;
        for (int i = 0; i < num_refs; i++) {
            int c = L->cell_ids[i];
            if (c >= 0 && ENTRY_LOG_DIM(L->entries[c]) == 0) { nref[num_kept] = L->ref_ids[i]; ncel[num_kept] = c; num_kept++; }
        }
        int rear = num_kept;
        if (g_cuda_quirks & 1) {
            /* cub::DevicePartition::Flagged (parallel.cuh:59-71): "rejected items are written to the rear in REVERSE order" */
            for (int i = num_refs - 1; i >= 0; i--) {
                int c = L->cell_ids[i];
                if (!(c >= 0 && ENTRY_LOG_DIM(L->entries[c]) == 0)) { nref[rear] = L->ref_ids[i]; ncel[rear] = c; rear++; }
            }
        } else
        for (int i = 0; i < num_refs; i++) {
            int c = L->cell_ids[i];
            if (!(c >= 0 && ENTRY_LOG_DIM(L->entries[c]) == 0)) { nref[rear] = L->ref_ids[i]; ncel[rear] = c; rear++; }
        }
        free(L->ref_ids); free(L->cell_ids);
        L->ref_ids = nref; L->cell_ids = ncel; L->num_kept = num_kept;
        if (getenv("ORC_VERBOSE"))
            fprintf(stderr, "[oracle] level %d: cells %d refs %d kept %d new_cells %d\n", num_levels - 1, num_cells, num_refs, num_kept, num_new_cells);
        if (num_new_cells == 0) break;      /* build.cu:583-587 */
        if (num_levels >= ORC_MAX_LEVELS) return -2;

        int num_split = num_refs - num_kept;
        /* compute_split_masks build.cu:594 + scan build.cu:597 + split_refs build.cu:219-243 */
        unsigned char* masks = (unsigned char*)xmalloc((size_t)num_split + 1);
        int64_t nn = 0;
        for (int i = 0; i < num_split; i++) {
            int c = ncel[num_kept + i];
            int m = 0;
            if (c >= 0) m = split_mask(&k, &L->cells[c], &tris[nref[num_kept + i]]);
            masks[i] = (unsigned char)m;
            nn += __builtin_popcount((unsigned)m);
        }
        if (nn > 0x3fffffff) return -1;
        int num_new_refs = (int)nn;
        int* cref = (int*)xmalloc(sizeof(int) * (size_t)num_new_refs);
        int* ccel = (int*)xmalloc(sizeof(int) * (size_t)num_new_refs);
        int pos = 0;
        for (int i = 0; i < num_split; i++) {
            int m = masks[i];
            if (!m) continue;
            int c = ncel[num_kept + i];
            uint32_t begin = ENTRY_BEGIN(L->entries[c]);
            for (int child = 0; child < 8; child++) if (m & (1 << child)) {
                cref[pos] = nref[num_kept + i];
                ccel[pos] = (int)begin + child;
                pos++;
            }
        }
        free(masks);
        /* emit_new_cells build.cu:354-383 */
        OCell* ncells = (OCell*)xmalloc(sizeof(OCell) * (size_t)num_new_cells);
        for (int id = 0; id < num_cells; id++) {
            OEntry e = L->entries[id];
            if (ENTRY_LOG_DIM(e) == 0) continue;
            int start = (int)ENTRY_BEGIN(e);
            OCell c = L->cells[id];
            int inc = (c.max.x - c.min.x) >> 1;
            for (int i = 0; i < 8; i++) {
                int x = c.min.x + (i & 1) * inc, y = c.min.y + ((i >> 1) & 1) * inc, z = c.min.z + (i >> 2) * inc;
                ncells[start + i].min = iv3(x, y, z);
                ncells[start + i].max = iv3(x + inc, y + inc, z + inc);
                ncells[start + i].begin = 0; ncells[start + i].end = 0;
            }
        }
        OEntry* nentries = (OEntry*)xcalloc((size_t)num_new_cells + 1, sizeof(OEntry));
        levels[num_levels++] = (Level){ cref, ccel, num_new_refs, num_new_refs, ncells, nentries, num_new_cells };
    }
    free(log_dims);

    /* ---- concat_levels, build.cu:621-716 ---- */
    int64_t total_refs64 = 0, total_cells64 = 0;
    for (int i = 0; i < num_levels; i++) { total_refs64 += levels[i].num_kept; total_cells64 += levels[i].num_cells; }
    if (total_refs64 > 0x3fffffff || total_cells64 > 0x3fffffff) return -1;
    int total_refs = (int)total_refs64, total_cells = (int)total_cells64;
    /* start_cell = exclusive scan of leaf flags over the concatenated cells (build.cu:650-659) */
    int* start_cell = (int*)xmalloc(sizeof(int) * ((size_t)total_cells + 1));
    int new_total_cells = 0;
    {
        int o = 0;
        for (int l = 0; l < num_levels; l++)
            for (int i = 0; i < levels[l].num_cells; i++) {
                start_cell[o++] = new_total_cells;
                new_total_cells += ENTRY_LOG_DIM(levels[l].entries[i]) == 0;
            }
        start_cell[o] = new_total_cells;
    }
    /* copy_cells build.cu:407-419, copy_entries build.cu:422-440 */
    OCell* out_cells = (OCell*)xmalloc(sizeof(OCell) * (size_t)new_total_cells);
    OEntry* out_entries = (OEntry*)xmalloc(sizeof(OEntry) * (size_t)total_cells);
    for (int l = 0, off = 0; l < num_levels; off += levels[l].num_cells, l++) {
        int nc = levels[l].num_cells, next_level_off = off + nc;
        for (int i = 0; i < nc; i++) {
            int s = start_cell[off + i], e = start_cell[off + i + 1];
            if (s < e) out_cells[s] = levels[l].cells[i];
            OEntry en = levels[l].entries[i];
            if (ENTRY_LOG_DIM(en) == 0) en = orc_make_entry(0, (uint32_t)start_cell[off + (int)ENTRY_BEGIN(en)]);
            else en = orc_make_entry(ENTRY_LOG_DIM(en), ENTRY_BEGIN(en) + (uint32_t)next_level_off);
            out_entries[off + i] = en;
        }
    }
    /* copy refs + copy_refs + remap_refs + stable sort by cell (build.cu:634-647,681,691) as a
     * counting sort, which is what a stable LSD radix sort over all key bits produces */
    int* out_refs = (int*)xmalloc(sizeof(int) * (size_t)total_refs);
    int* counts = (int*)xcalloc((size_t)new_total_cells + 1, sizeof(int));
    for (int l = 0, cell_off = 0; l < num_levels; cell_off += levels[l].num_cells, l++)
        for (int i = 0; i < levels[l].num_kept; i++)
            counts[start_cell[levels[l].cell_ids[i] + cell_off] + 1]++;
    for (int i = 0; i < new_total_cells; i++) counts[i + 1] += counts[i];
    /* compute_cell_ranges build.cu:453-468: cells without refs keep begin = end = 0 */
    for (int i = 0; i < new_total_cells; i++) {
        if (counts[i + 1] > counts[i]) { out_cells[i].begin = counts[i]; out_cells[i].end = counts[i + 1]; }
        else { out_cells[i].begin = 0; out_cells[i].end = 0; }
    }
    {
        int* cursor = (int*)xmalloc(sizeof(int) * (size_t)(new_total_cells + 1));
        memcpy(cursor, counts, sizeof(int) * (size_t)(new_total_cells + 1));
        for (int l = 0, cell_off = 0; l < num_levels; cell_off += levels[l].num_cells, l++)
            for (int i = 0; i < levels[l].num_kept; i++) {
                int c = start_cell[levels[l].cell_ids[i] + cell_off];
                out_refs[cursor[c]++] = levels[l].ref_ids[i];
            }
        free(cursor);
    }
    free(counts);
    free(start_cell);

    grid->entries = out_entries; grid->ref_ids = out_refs; grid->cells = out_cells; grid->small_cells = NULL;
    grid->shift = shift;                              /* D3 */
    grid->num_cells = new_total_cells; grid->num_entries = total_cells; grid->num_refs = total_refs;
    grid->num_offsets = shift + 1;
    {
        int off = 0;
        for (int i = 0; i <= shift; i++) {          /* build.cu:711-715 (+ D3 padding) */
            if (i < num_levels) off += levels[i].num_cells;
            grid->offsets[i] = off;
        }
    }
    grid->dims = dims; grid->bbox = gb;
    for (int l = 0; l < num_levels; l++) { free(levels[l].ref_ids); free(levels[l].cell_ids); free(levels[l].cells); free(levels[l].entries); }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* merge_grid: merge.cu:21-377 */

typedef struct { oivec3 dims; ovec3 cell_size; int shift; } MergeConsts;

static int aligned_cells(int axis, const OCell* c1, const OCell* c2) { /* merge.cu:21-31 */
    int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    return iget(c1->max, axis) == iget(c2->min, axis) &&
           iget(c1->min, a1) == iget(c2->min, a1) && iget(c1->min, a2) == iget(c2->min, a2) &&
           iget(c1->max, a1) == iget(c2->max, a1) && iget(c1->max, a2) == iget(c2->max, a2);
}
static int merge_allowed(const MergeConsts* k, int empty_mask, int pos) { /* merge.cu:34-39 */
    int top_level_mask = (1 << k->shift) - 1;
    int is_shifted = (pos >> k->shift) & empty_mask;
    int is_top_level = !(pos & top_level_mask);
    return !is_shifted || !is_top_level;
}
static oivec3 next_cell_pos(int axis, oivec3 mn, oivec3 mx) { /* merge.cu:42-47 */
    return iv3(axis == 0 ? mx.x : mn.x, axis == 1 ? mx.y : mn.y, axis == 2 ? mx.z : mn.z);
}
static int count_union(const int* p0, int c0, const int* p1, int c1) { /* merge.cu:58-69 */
    int i = 0, j = 0, c = 0;
    while ((i < c0) & (j < c1)) {
        int a = p0[i], b = p1[j];
        i += (a <= b); j += (a >= b); c++;
    }
    return c + (c1 - j) + (c0 - i);
}
static void merge_refs(const int* p0, int c0, const int* p1, int c1, int* q) { /* merge.cu:72-88 */
    int i = 0, j = 0;
    while (i < c0 && j < c1) {
        int a = p0[i], b = p1[j];
        *(q++) = (a < b) ? a : b;
        i += (a <= b); j += (a >= b);
    }
    int kk = i < c0 ? i : j, c = i < c0 ? c0 : c1;
    const int* p = i < c0 ? p0 : p1;
    while (kk < c) *(q++) = p[kk++];
}

static void merge_iteration(int axis, const MergeConsts* k, OGrid* grid, int empty_mask) { /* merge.cu:292-329 */
    int num_cells = grid->num_cells, num_entries = grid->num_entries;
    const OCell* cells = grid->cells; const int* refs = grid->ref_ids; OEntry* entries = grid->entries;
    oivec3 top = iv3(k->dims.x >> k->shift, k->dims.y >> k->shift, k->dims.z >> k->shift);
    int* merge_counts = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    int* nexts = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    int* prevs = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    int* flags = (int*)xcalloc((size_t)num_cells + 1, sizeof(int));
    for (int i = 0; i <= num_cells; i++) prevs[i] = -1;                         /* merge.cu:302 */
    const float unit_cost = 1.0f;
    /* compute_merge_counts merge.cu:91-142 */
    for (int id = 0; id < num_cells; id++) {
        OCell c1 = cells[id];
        oivec3 np = next_cell_pos(axis, c1.min, c1.max);
        int count = -(c1.end - c1.begin + 1);
        int next_id = -1;
        if (merge_allowed(k, empty_mask, iget(c1.min, axis)) && iget(np, axis) < iget(k->dims, axis)) {
            next_id = (int)orc_lookup_entry(entries, k->shift, &top, &np, NULL);
            OCell c2 = cells[next_id];
            if (aligned_cells(axis, &c1, &c2)) {
                ovec3 e1 = v3_mul(v3_from_i(iv3(c1.max.x - c1.min.x, c1.max.y - c1.min.y, c1.max.z - c1.min.z)), k->cell_size);
                ovec3 e2 = v3_mul(v3_from_i(iv3(c2.max.x - c2.min.x, c2.max.y - c2.min.y, c2.max.z - c2.min.z)), k->cell_size);
                float a1 = e1.x * (e1.y + e1.z) + e1.y * e1.z;
                float a2 = e2.x * (e2.y + e2.z) + e2.y * e2.z;
                float a = a1 + a2 - fget(e1, (axis + 1) % 3) * fget(e1, (axis + 2) % 3);
                int n1 = c1.end - c1.begin, n2 = c2.end - c2.begin;
                float cc1 = a1 * (n1 + unit_cost), cc2 = a2 * (n2 + unit_cost);
                if (a * (imax(n1, n2) + unit_cost) <= cc1 + cc2) {
                    int n = count_union(refs + c1.begin, n1, refs + c2.begin, n2);
                    float c = a * (n + unit_cost);
                    if (c <= cc1 + cc2) count = n;
                }
            }
        }
        merge_counts[id] = count;
        next_id = count >= 0 ? next_id : -1;
        nexts[id] = next_id;
        if (next_id >= 0) prevs[next_id] = id;
    }
    /* compute_cell_flags merge.cu:145-170 */
    for (int id = 0; id < num_cells; id++) {
        if (prevs[id] < 0) {
            int next_id = nexts[id];
            flags[id] = 1;
            if (next_id >= 0) {
                int count = 1;
                do { flags[next_id] = count % 2 ? 0 : 1; next_id = nexts[next_id]; count++; } while (next_id >= 0);
            }
        }
    }
    /* compute_ref_counts merge.cu:173-186 + scans merge.cu:310-311 */
    int* cell_scan = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    int* ref_scan  = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    int num_new_cells = 0, num_new_refs = 0;
    for (int id = 0; id < num_cells; id++) {
        int count = 0;
        if (flags[id]) { int m = merge_counts[id]; count = m >= 0 ? m : -(m + 1); }
        cell_scan[id] = num_new_cells; ref_scan[id] = num_new_refs;
        num_new_cells += flags[id]; num_new_refs += count;
    }
    cell_scan[num_cells] = num_new_cells; ref_scan[num_cells] = num_new_refs;
    /* merge merge.cu:189-278 */
    OCell* new_cells = (OCell*)xmalloc(sizeof(OCell) * (size_t)imax(num_new_cells, 1));
    int* new_refs = (int*)xmalloc(sizeof(int) * (size_t)imax(num_new_refs, 1));
    int* new_cell_ids = (int*)xmalloc(sizeof(int) * (size_t)(num_cells + 1));
    for (int id = 0; id < num_cells; id++) {
        int new_id = cell_scan[id];
        if (!(cell_scan[id + 1] > new_id)) continue;
        OCell cell = cells[id];
        int mc = merge_counts[id];
        int nb = ref_scan[id];
        new_cell_ids[id] = new_id;
        OCell out;
        if (mc >= 0) {
            oivec3 np = next_cell_pos(axis, cell.min, cell.max);
            int next_id = (int)orc_lookup_entry(entries, k->shift, &top, &np, NULL);
            OCell nc = cells[next_id];
            new_cell_ids[next_id] = new_id;
            out.min = iv3(imin(nc.min.x, cell.min.x), imin(nc.min.y, cell.min.y), imin(nc.min.z, cell.min.z));
            out.max = iv3(imax(nc.max.x, cell.max.x), imax(nc.max.y, cell.max.y), imax(nc.max.z, cell.max.z));
            out.begin = nb; out.end = nb + mc;
            if (nc.begin < nc.end)
                merge_refs(refs + cell.begin, cell.end - cell.begin, refs + nc.begin, nc.end - nc.begin, new_refs + nb);
            else
                memcpy(new_refs + nb, refs + cell.begin, sizeof(int) * (size_t)(cell.end - cell.begin));
        } else {
            out.min = cell.min; out.max = cell.max; out.begin = nb; out.end = nb + (cell.end - cell.begin);
            memcpy(new_refs + nb, refs + cell.begin, sizeof(int) * (size_t)(cell.end - cell.begin));
        }
        new_cells[new_id] = out;
    }
    /* remap_entries merge.cu:281-290 */
    for (int i = 0; i < num_entries; i++) {
        OEntry e = entries[i];
        if (ENTRY_LOG_DIM(e) == 0) entries[i] = orc_make_entry(0, (uint32_t)new_cell_ids[ENTRY_BEGIN(e)]);
    }
    free(merge_counts); free(nexts); free(prevs); free(flags); free(cell_scan); free(ref_scan); free(new_cell_ids);
    free(grid->cells); free(grid->ref_ids);
    grid->cells = new_cells; grid->ref_ids = new_refs;
    grid->num_cells = num_new_cells; grid->num_refs = num_new_refs;
}

int orc_merge_grid(OGrid* grid, float alpha) { /* merge.cu:331-377 */
    MergeConsts k;
    k.dims = iv3(grid->dims.x << grid->shift, grid->dims.y << grid->shift, grid->dims.z << grid->shift);
    k.cell_size = v3_div(v3_sub(grid->bbox.max, grid->bbox.min), v3_from_i(k.dims));
    k.shift = grid->shift;
    if (alpha > 0) {
        int prev_num_cells = 0, iter = 0;
        do {
            prev_num_cells = grid->num_cells;
            int mask = iter > 3 ? 0 : (1 << (iter + 1)) - 1;
            merge_iteration(0, &k, grid, mask);
            merge_iteration(1, &k, grid, mask);
            merge_iteration(2, &k, grid, mask);
            iter++;
        } while (grid->num_cells < alpha * prev_num_cells);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* flatten_grid: flatten.cu:6-175 */

#define FLAT_LEVELS 3 /* flatten.cu:6 */

int orc_flatten_grid(OGrid* grid) {
    int num_entries = grid->num_entries, shift = grid->shift;
    OEntry* entries = grid->entries;
    int* depths = (int*)xcalloc((size_t)num_entries + 1, sizeof(int));
    /* collapse_entries flatten.cu:9-29 + compute_depths flatten.cu:32-46, deepest level first */
    for (int i = shift; i >= 0; i--) {
        int first = i > 0 ? grid->offsets[i - 1] : 0, last = grid->offsets[i];
        for (int id = first; id < last; id++) {
            OEntry e = entries[id];
            if (ENTRY_LOG_DIM(e)) {
                const OEntry* p = entries + ENTRY_BEGIN(e);
                if (p[0] == p[1] && p[0] == p[2] && p[0] == p[3] && p[0] == p[4] && p[4] == p[5] && p[4] == p[6] && p[4] == p[7])
                    entries[id] = p[0];
            }
        }
        for (int id = first; id < last; id++) {
            OEntry e = entries[id];
            int d = 0;
            if (ENTRY_LOG_DIM(e)) {
                const int* p = depths + ENTRY_BEGIN(e);
                int m = p[0];
                for (int c = 1; c < 8; c++) m = imax(m, p[c]);
                d = 1 + m;
            }
            depths[id] = d;
        }
    }
    /* flatten.cu:127-141 */
    int* start_entries = (int*)xcalloc((size_t)num_entries + 1, sizeof(int));
    int level_offsets[ORC_MAX_LEVELS + FLAT_LEVELS];
    int total_entries = grid->offsets[0];
    for (int i = 0; i < shift; i += FLAT_LEVELS) {
        int first = i > 0 ? grid->offsets[i - 1] : 0, last = grid->offsets[i];
        int sum = 0;
        for (int id = first; id < last; id++) {
            start_entries[id] = sum;
            int d = depths[id];
            sum += d > 0 ? 1 << (imin(d, FLAT_LEVELS) * 3) : 0;
        }
        level_offsets[i] = total_entries;
        total_entries += sum;
    }
    OEntry* ne = (OEntry*)xmalloc(sizeof(OEntry) * (size_t)total_entries);
    int new_offsets[ORC_MAX_LEVELS], num_new_offsets = 0;
    /* copy_top_level flatten.cu:49-62 */
    for (int id = 0; id < grid->offsets[0]; id++) {
        OEntry e = entries[id];
        if (ENTRY_LOG_DIM(e)) e = orc_make_entry((uint32_t)imin(depths[id], FLAT_LEVELS), (uint32_t)(grid->offsets[0] + start_entries[id]));
        ne[id] = e;
    }
    /* flatten_level flatten.cu:65-107 */
    for (int i = 0; i < shift; i += FLAT_LEVELS) {
        int first = i > 0 ? grid->offsets[i - 1] : 0, last = grid->offsets[i];
        int next_offset = i + FLAT_LEVELS < shift ? level_offsets[i + FLAT_LEVELS] : 0;
        for (int id = 0; id < last - first; id++) {
            int d = imin(depths[id + first], FLAT_LEVELS);
            int nsub = d == 0 ? 0 : 1 << (3 * d);
            if (nsub <= 0) continue;
            int start = level_offsets[i] + start_entries[id + first];
            OEntry root = entries[id + first];
            for (int m = 0; m < nsub; m++) {
                int cur_d = d, x = 0, y = 0, z = 0, next_id = id;
                OEntry e = root;
                while (cur_d > 0) {
                    cur_d--;
                    int pos = m >> (cur_d * 3);
                    x += (pos & 1) ? (1 << cur_d) : 0;
                    y += (pos & 2) ? (1 << cur_d) : 0;
                    z += (pos & 4) ? (1 << cur_d) : 0;
                    if (ENTRY_LOG_DIM(e)) { next_id = (int)ENTRY_BEGIN(e) + (pos & 7); e = entries[next_id]; }
                }
                if (ENTRY_LOG_DIM(e))
                    e = orc_make_entry((uint32_t)imin(depths[next_id], FLAT_LEVELS), (uint32_t)(next_offset + start_entries[next_id]));
                ne[start + x + ((y + (z << d)) << d)] = e;
            }
        }
        new_offsets[num_new_offsets++] = level_offsets[i];
    }
    new_offsets[num_new_offsets++] = total_entries;
    free(grid->entries); free(depths); free(start_entries);
    grid->entries = ne; grid->num_entries = total_entries;
    grid->num_offsets = num_new_offsets;
    memcpy(grid->offsets, new_offsets, sizeof(int) * (size_t)num_new_offsets);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* expand_grid: expand.cu:11-225 with subset_only = true (expand.cu:159) */

typedef struct { oivec3 dims; int shift; oivec3 top; ovec3 gmin, cell_size, grid_inv; int subset_only; const OTri* tris; } ExpandConsts;

static int is_subset(const int* p0, int c0, const int* p1, int c1) { /* expand.cu:21-36 */
    if (c1 > c0) return 0;
    if (c1 == 0) return 1;
    int i = 0, j = 0;
    do {
        int a = p0[i], b = p1[j];
        if (b < a) return 0;
        j += (a == b);
        i++;
    } while ((i < c0) & (j < c1));
    return j == c1;
}

static long long g_walk_iters, g_walk_max;   /* ORC_VERBOSE statistics */

/* expand.cu:39-57: how far may `cell` grow along axis before it would have to reference `prim` */
static int compute_overlap(int axis, int dir, const ExpandConsts* k, const OTri* prim, const OCell* cell, const OBBox* cb, int d) {
    int axis1 = (axis + 1) % 3, axis2 = (axis + 2) % 3;
    OBBox pb; orc_tri_bbox(prim, &pb);
    if (fget(pb.min, axis1) <= fget(cb->max, axis1) && fget(pb.max, axis1) >= fget(cb->min, axis1) &&
        fget(pb.min, axis2) <= fget(cb->max, axis2) && fget(pb.max, axis2) >= fget(cb->min, axis2)) {
        int prim_d = (int)(((dir ? fget(pb.min, axis) : fget(pb.max, axis)) - fget(k->gmin, axis)) * fget(k->grid_inv, axis));
        d = dir ? imin(d, prim_d - iget(cell->max, axis)) : imax(d, prim_d - iget(cell->min, axis) + 1);
        d = dir ? imax(d, 0) : imin(d, 0);
    }
    return d;
}

/* expand.cu:60-143 */
static int find_overlap(int axis, int dir, const ExpandConsts* k, const OEntry* entries, const int* refs,
                        const OCell* cells, const OCell* cell, int* continue_overlap) {
    int axis1 = (axis + 1) % 3, axis2 = (axis + 2) % 3;
    /* overlap_possible expand.cu:12-18 */
    if (dir) { if (!(iget(cell->max, axis) < iget(k->dims, axis))) return 0; }
    else     { if (!(iget(cell->min, axis) > 0)) return 0; }
    int d = dir ? iget(k->dims, axis) : -iget(k->dims, axis);
    int k1, k2 = iget(k->dims, axis2);
    int i = iget(cell->min, axis1), j = iget(cell->min, axis2);
    int max_d = d;
    long long iters = 0;
    for (;;) {
        iters++;
        oivec3 np;
        int a = dir ? iget(cell->max, axis) : iget(cell->min, axis) - 1;
        if (axis == 0) np = iv3(a, i, j);
        else if (axis == 1) np = iv3(j, a, i);
        else np = iv3(i, j, a);
        uint32_t en = orc_lookup_entry(entries, k->shift, &k->top, &np, NULL);
        OCell next = cells[en];
        max_d = dir ? imin(max_d, iget(next.max, axis) - iget(cell->max, axis))
                    : imax(max_d, iget(next.min, axis) - iget(cell->min, axis));
        d = dir ? imin(d, max_d) : imax(d, max_d);
        if (k->subset_only) {
            if (!is_subset(refs + cell->begin, cell->end - cell->begin, refs + next.begin, next.end - next.begin)) { d = 0; break; }
        } else {
            /* expand.cu:96-127: references of the neighbour that the cell does not hold limit the growth */
            if (next.begin < next.end) {
                OBBox cb;
                cb.min = v3_add(k->gmin, v3_mul(k->cell_size, v3_from_i(cell->min)));
                cb.max = v3_add(k->gmin, v3_mul(k->cell_size, v3_from_i(cell->max)));
                int p1 = cell->begin, p2 = next.begin;
                int ref2 = refs[p2];
                for (;;) {
                    while (p1 < cell->end) {
                        int ref1 = refs[p1];
                        if (ref1 > ref2) break;
                        if (ref1 == ref2) {
                            if (++p2 >= next.end) break;
                            ref2 = refs[p2];
                        }
                        p1++;
                    }
                    if (p2 >= next.end) break;
                    d = compute_overlap(axis, dir, k, &k->tris[ref2], cell, &cb, d);
                    if (d == 0 || ++p2 >= next.end) break;
                    ref2 = refs[p2];
                }
            }
            if (d == 0) break;
        }
        k1 = iget(next.max, axis1) - i;
        k2 = imin(k2, iget(next.max, axis2) - j);
        i += k1;
        if (i >= iget(cell->max, axis1)) {
            i = iget(cell->min, axis1);
            j += k2;
            k2 = iget(k->dims, axis2);
            if (j >= iget(cell->max, axis2)) break;
        }
    }
    g_walk_iters += iters; if (iters > g_walk_max) g_walk_max = iters;
    *continue_overlap |= d == max_d;
    return d;
}

int orc_expand_grid(OGrid* grid, const OTri* tris, int iters) { return orc_expand_grid_ex(grid, tris, iters, 1); }

int orc_expand_grid_ex(OGrid* grid, const OTri* tris, int iters, int subset_only) { /* expand.cu:199-225 */
    if (iters == 0) return 0;
    ExpandConsts k;
    k.dims = iv3(grid->dims.x << grid->shift, grid->dims.y << grid->shift, grid->dims.z << grid->shift);
    k.shift = grid->shift; k.top = grid->dims;
    k.subset_only = subset_only; k.tris = tris;
    {   /* expand.cu:208-216 */
        ovec3 ext = v3_sub(grid->bbox.max, grid->bbox.min);
        k.gmin = grid->bbox.min;
        k.cell_size = v3_div(ext, v3_from_i(k.dims));
        k.grid_inv = v3_div(v3_from_i(k.dims), ext);
    }
    int n = grid->num_cells;
    OCell* new_cells = (OCell*)xmalloc(sizeof(OCell) * (size_t)imax(n, 1));
    int* flags = (int*)xmalloc(sizeof(int) * (size_t)imax(n, 1));
    for (int i = 0; i < n; i++) flags[i] = -1;                                   /* expand.cu:206 */
    for (int it = 0; it < iters; it++) {
        for (int axis = 0; axis < 3; axis++) {                                   /* expand.cu:184-197 */
            const OCell* cells = grid->cells;
            if (getenv("ORC_VERBOSE")) {
                int flagged = 0;
                for (int id = 0; id < n; id++) flagged += (flags[id] & (1 << axis)) != 0;
                fprintf(stderr, "[oracle] expand iter %d axis %d: %d of %d cells processed; previous step: %lld face-walk iterations, longest %lld\n", it, axis, flagged, n, g_walk_iters, g_walk_max);
                g_walk_iters = 0; g_walk_max = 0;
            }
            for (int id = 0; id < n; id++) {                                      /* overlap_step expand.cu:145-182 */
                if ((flags[id] & (1 << axis)) == 0) {
                    /* D2: copied through.  As-CUDA (expand.cu:154-155: `return` without a store): the slot keeps what the pass two passes
                     * ago left there.  (The first three passes process every cell, so no slot is ever uninitialised.) */
                    if (!(g_cuda_quirks & 2)) new_cells[id] = cells[id];
                    continue;
                }
                OCell cell = cells[id];
                int flag = 0;
                int ov1 = find_overlap(axis, 0, &k, grid->entries, grid->ref_ids, cells, &cell, &flag);
                int ov2 = find_overlap(axis, 1, &k, grid->entries, grid->ref_ids, cells, &cell, &flag);
                iset(&cell.min, axis, iget(cell.min, axis) + ov1);
                iset(&cell.max, axis, iget(cell.max, axis) + ov2);
                flags[id] = (flag ? 1 << axis : 0) | (flags[id] & ~(1 << axis));
                new_cells[id] = cell;
            }
            OCell* t = grid->cells; grid->cells = new_cells; new_cells = t;
        }
    }
    free(new_cells); free(flags);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* compress_grid: compress.cu:6-63 */

int orc_compress_grid(OGrid* grid) {
    oivec3 dims = iv3(grid->dims.x << grid->shift, grid->dims.y << grid->shift, grid->dims.z << grid->shift);
    if (dims.x >= (1 << 16) || dims.y >= (1 << 16) || dims.z >= (1 << 16)) return 0;
    int n = grid->num_cells;
    OSmallCell* sc = (OSmallCell*)xmalloc(sizeof(OSmallCell) * (size_t)imax(n, 1));
    int64_t total = 0;
    for (int i = 0; i < n; i++) { int c = grid->cells[i].end - grid->cells[i].begin; total += c > 0 ? c + 1 : 0; }
    int* srefs = (int*)xmalloc(sizeof(int) * (size_t)(total ? total : 1));
    int first = 0;
    for (int i = 0; i < n; i++) {
        OCell c = grid->cells[i];
        int count = c.end - c.begin;
        sc[i].min[0] = (uint16_t)c.min.x; sc[i].min[1] = (uint16_t)c.min.y; sc[i].min[2] = (uint16_t)c.min.z;
        sc[i].max[0] = (uint16_t)c.max.x; sc[i].max[1] = (uint16_t)c.max.y; sc[i].max[2] = (uint16_t)c.max.z;
        sc[i].begin = count > 0 ? first : -1;
        if (count > 0) {
            memcpy(srefs + first, grid->ref_ids + c.begin, sizeof(int) * (size_t)count);
            srefs[first + count] = -1;
            first += count + 1;
        }
    }
    free(grid->cells); free(grid->ref_ids);
    grid->cells = NULL; grid->small_cells = sc; grid->ref_ids = srefs; grid->num_refs = (int)total;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* traverse: traverse.cu:14-117 */

typedef struct { /* traverse.cu:7-12, setup_traversal traverse.cu:97-109 */
    oivec3 dims; ovec3 gmin, gmax, cell_size, grid_inv; int shift; oivec3 top;
} TravConsts;

static void setup_consts(const OGrid* g, TravConsts* k) {
    ovec3 ext = v3_sub(g->bbox.max, g->bbox.min);
    k->dims = iv3(g->dims.x << g->shift, g->dims.y << g->shift, g->dims.z << g->shift);
    k->grid_inv = v3_div(v3_from_i(k->dims), ext);
    k->cell_size = v3_div(ext, v3_from_i(k->dims));
    k->gmin = g->bbox.min; k->gmax = g->bbox.max; k->shift = g->shift;
    k->top = iv3(k->dims.x >> k->shift, k->dims.y >> k->shift, k->dims.z >> k->shift);
}

static inline ovec3 compute_voxel(const TravConsts* k, ovec3 org, ovec3 dir, float t) { /* traverse.cu:23-25 */
    return v3_mul(v3_sub(v3_add(v3_scale(dir, t), org), k->gmin), k->grid_inv);
}

static inline int clampi(int a, int b, int c) { return imin(c, imax(b, a)); } /* common.h:27 */

/* dev analysis hook (tests/analysis/wave_model.py): list length of every visited cell of the ray being traversed */
static __thread unsigned char* g_trace = NULL;
static __thread int g_trace_cap = 0, g_trace_len = 0;
static __thread int* g_trace_ids = NULL;
static __thread short* g_trace_vox = NULL;        /* dev analysis: the voxel of every look-up, 3 shorts per step (same capacity as g_trace) */
static __thread int g_trace_ids_cap = 0, g_trace_ids_len = 0;

static void traverse_one(const TravConsts* k, const OGrid* g, const OTri* tris, const ORay* rp, OHit* out, int* steps_out, OStats* st) {
    ORay ray = *rp;
    ovec3 inv_dir = v3(orc_safe_rcp(ray.dir.x), orc_safe_rcp(ray.dir.y), orc_safe_rcp(ray.dir.z));
    /* intersect_ray_box traverse.cu:14-21 */
    ovec3 tmn = v3_mul(v3_sub(k->gmin, ray.org), inv_dir);
    ovec3 tmx = v3_mul(v3_sub(k->gmax, ray.org), inv_dir);
    ovec3 t0 = v3_min(tmn, tmx), t1 = v3_max(tmn, tmx);
    float tbx = fmaxf(t0.x, fmaxf(t0.y, t0.z));
    float tby = fminf(t1.x, fminf(t1.y, t1.z));
    float tstart = fmaxf(tbx, ray.tmin);
    float tend   = fminf(tby, ray.tmax);
    OHit hit = { -1, ray.tmax, 0, 0 };
    int steps = 0;
    if (st) st->rays++;
    if (!(tstart > tend)) {
        if (st) st->rays_hit_grid++;
        ovec3 fv = compute_voxel(k, ray.org, ray.dir, tstart);
        oivec3 voxel = iv3(clampi((int)fv.x, 0, k->dims.x - 1), clampi((int)fv.y, 0, k->dims.y - 1), clampi((int)fv.z, 0, k->dims.z - 1));
        for (;;) {
            int words = 0;
            uint32_t entry = orc_lookup_entry(g->entries, k->shift, &k->top, &voxel, &words);
            const oivec3 look = voxel;                 /* (dev analysis: the voxel of this look-up) */
            oivec3 cmin, cmax; int cbegin, cend = 0;
            if (g->small_cells) {
                OSmallCell sc = g->small_cells[entry];
                cmin = iv3(sc.min[0], sc.min[1], sc.min[2]); cmax = iv3(sc.max[0], sc.max[1], sc.max[2]); cbegin = sc.begin;
            } else {
                OCell c = g->cells[entry];
                cmin = c.min; cmax = c.max; cbegin = c.begin; cend = c.end;
            }
            /* traverse.cu:63-68 */
            oivec3 cp = iv3(ray.dir.x >= 0.0f ? cmax.x : cmin.x, ray.dir.y >= 0.0f ? cmax.y : cmin.y, ray.dir.z >= 0.0f ? cmax.z : cmin.z);
            ovec3 tcell = v3_mul(v3_sub(v3_add(v3_mul(v3_from_i(cp), k->cell_size), k->gmin), ray.org), inv_dir);
            float texit = fminf(tcell.x, fminf(tcell.y, tcell.z));
            /* traverse.cu:70-77 */
            ovec3 ev = compute_voxel(k, ray.org, ray.dir, texit);
            oivec3 ep = iv3((int)ev.x, (int)ev.y, (int)ev.z);
            oivec3 nv = iv3(texit == tcell.x ? cp.x + (ray.dir.x >= 0.0f ? 0 : -1) : ep.x,
                            texit == tcell.y ? cp.y + (ray.dir.y >= 0.0f ? 0 : -1) : ep.y,
                            texit == tcell.z ? cp.z + (ray.dir.z >= 0.0f ? 0 : -1) : ep.z);
            voxel.x = ray.dir.x >= 0.0f ? imax(nv.x, voxel.x) : imin(nv.x, voxel.x);
            voxel.y = ray.dir.y >= 0.0f ? imax(nv.y, voxel.y) : imin(nv.y, voxel.y);
            voxel.z = ray.dir.z >= 0.0f ? imax(nv.z, voxel.z) : imin(nv.z, voxel.z);
            /* foreach_ref grid.h:118-140 */
            int nrefs = 0, found_any = 0;
            if (g->small_cells) {
                if (cbegin >= 0) {
                    int cur = cbegin;
                    for (;;) {
                        int ref = g->ref_ids[cur++];
                        if (ref < 0) break;
                        ORay r2 = { ray.org, ray.tmin, ray.dir, hit.t };
                        int got = (g_mode & ORC_UVS) ? orc_intersect_prim_ray_uv(&tris[ref], &r2, ref, &hit) : orc_intersect_prim_ray(&tris[ref], &r2, ref, &hit);
                        if (got && (g_mode & ORC_ANY_HIT)) { found_any = 1; break; }
                    }
                    nrefs = cur - cbegin;
                    if (st) { st->refs += nrefs - 1; st->sentinels += 1; if (nrefs - 1 > 4) st->long_list_refs += nrefs - 1; }
                }
            } else {
                for (int cur = cbegin; cur < cend; cur++) {
                    int ref = g->ref_ids[cur];
                    if (ref < 0) break;
                    if (g_trace_ids) { if (g_trace_ids_len < g_trace_ids_cap) g_trace_ids[g_trace_ids_len] = ref; g_trace_ids_len++; }
                    ORay r2 = { ray.org, ray.tmin, ray.dir, hit.t };
                    int got = (g_mode & ORC_UVS) ? orc_intersect_prim_ray_uv(&tris[ref], &r2, ref, &hit) : orc_intersect_prim_ray(&tris[ref], &r2, ref, &hit);
                    if (got && (g_mode & ORC_ANY_HIT)) { found_any = 1; break; }
                }
                nrefs = cend - cbegin;
                if (st) { st->refs += nrefs; if (nrefs > 4) st->long_list_refs += nrefs; }
            }
            steps += 1 + nrefs;
            if (g_trace) { if (g_trace_len < g_trace_cap) { g_trace[g_trace_len] = (unsigned char)(nrefs > 255 ? 255 : nrefs); if (g_trace_vox) { g_trace_vox[3 * g_trace_len] = (short)look.x; g_trace_vox[3 * g_trace_len + 1] = (short)look.y; g_trace_vox[3 * g_trace_len + 2] = (short)look.z; } } g_trace_len++; }
            if (st) { st->cells++; st->entry_words += words; }
            /* traverse.cu:85-89 */
            if (found_any || hit.t <= texit ||
                ((voxel.x < 0) | (voxel.x >= k->dims.x) | (voxel.y < 0) | (voxel.y >= k->dims.y) | (voxel.z < 0) | (voxel.z >= k->dims.z)))
                break;
        }
    }
    if (st && hit.id >= 0) st->hits++;
    *out = hit;                 /* D4: id stays the primitive id */
    if (steps_out) *steps_out = steps;
}

void orc_traverse_grid(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits, int64_t n, int* steps, OStats* stats) {
    TravConsts k; setup_consts(grid, &k);
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int64_t i = 0; i < n; i++) traverse_one(&k, grid, tris, &rays[i], &hits[i], steps ? &steps[i] : NULL, stats);
}

/* dev analysis: as orc_traverse_trace, plus vox[(i * cap + s) * 3 ..] = the voxel of the s-th look-up of ray i */
void orc_traverse_trace_voxels(const OGrid* grid, const OTri* tris, const ORay* rays, int64_t n, int cap, unsigned char* lens, int* num_cells, short* vox) {
    TravConsts k; setup_consts(grid, &k);
    for (int64_t i = 0; i < n; i++) {
        OHit hit;
        g_trace = lens + i * cap; g_trace_cap = cap; g_trace_len = 0; g_trace_vox = vox + i * cap * 3;
        traverse_one(&k, grid, tris, &rays[i], &hit, NULL, NULL);
        num_cells[i] = g_trace_len;
    }
    g_trace = NULL; g_trace_vox = NULL;
}

void orc_traverse_trace(const OGrid* grid, const OTri* tris, const ORay* rays, int64_t n, int cap, unsigned char* lens, int* num_cells,
                        int ids_cap, int* ids, int* num_ids) {
    TravConsts k; setup_consts(grid, &k);
    for (int64_t i = 0; i < n; i++) {
        OHit hit;
        g_trace = lens + i * cap; g_trace_cap = cap; g_trace_len = 0;
        g_trace_ids = ids ? ids + i * ids_cap : NULL; g_trace_ids_cap = ids_cap; g_trace_ids_len = 0;
        traverse_one(&k, grid, tris, &rays[i], &hit, NULL, NULL);
        num_cells[i] = g_trace_len;
        if (num_ids) num_ids[i] = g_trace_ids_len;
    }
    g_trace = NULL; g_trace_ids = NULL;
}

/* Threads take chunks of kChunk consecutive rays from one shared counter (dynamic scheduling: ray costs differ by two orders of
 * magnitude and a static split leaves threads idle), count into a LOCAL statistics block (the per-thread blocks used to sit side
 * by side in one array -- every cell of every ray was a write to a cache line shared with the neighbour thread) and may be pinned,
 * one per entry of `cpus` (oracle.py: one hardware thread per physical core).  Test infrastructure: the CPU baseline of bench.py. */
enum { kChunk = 4096 };
typedef struct {
    const OGrid* grid; const OTri* tris; const ORay* rays; OHit* hits; int64_t n; int64_t* next; int cpu;
    int num_tris; int brute; unsigned mode;
    char pad0[64];
    OStats stats;
    char pad1[64];
} Job;

static void* job_main(void* p) {
    Job* j = (Job*)p;
    if (j->cpu >= 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);      /* best effort: a refused mask leaves the thread where it is */
    }
    OStats local; memset(&local, 0, sizeof(local));
    TravConsts k;
    if (!j->brute) { setup_consts(j->grid, &k); g_mode = j->mode; }
    for (;;) {
        const int64_t begin = __atomic_fetch_add(j->next, (int64_t)kChunk, __ATOMIC_RELAXED);
        if (begin >= j->n) break;
        const int64_t end = begin + kChunk < j->n ? begin + kChunk : j->n;
        if (j->brute) {
            for (int64_t i = begin; i < end; i++) {
                OHit hit = { -1, j->rays[i].tmax, 0, 0 };
                for (int t = 0; t < j->num_tris; t++) {
                    ORay r2 = { j->rays[i].org, j->rays[i].tmin, j->rays[i].dir, hit.t };
                    orc_intersect_prim_ray(&j->tris[t], &r2, t, &hit);
                }
                j->hits[i] = hit;
            }
        } else {
            for (int64_t i = begin; i < end; i++) traverse_one(&k, j->grid, j->tris, &j->rays[i], &j->hits[i], NULL, &local);
        }
    }
    g_mode = 0;
    j->stats = local;
    return NULL;
}

static void run_jobs_on(Job* proto, int64_t n, int nthreads, const int* cpus, int num_cpus, OStats* stats) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    Job* jobs = (Job*)xmalloc(sizeof(Job) * (size_t)nthreads);
    pthread_t* th = (pthread_t*)xmalloc(sizeof(pthread_t) * (size_t)nthreads);
    int64_t* next = (int64_t*)xmalloc(256);                    /* the chunk counter on a cache line of its own */
    next[16] = 0;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = *proto;
        jobs[t].n = n; jobs[t].next = next + 16;
        jobs[t].cpu = (cpus && num_cpus > 0) ? cpus[t % num_cpus] : -1;
        memset(&jobs[t].stats, 0, sizeof(OStats));
        if (nthreads == 1) job_main(&jobs[t]); else pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        if (stats) {
            stats->rays += jobs[t].stats.rays; stats->rays_hit_grid += jobs[t].stats.rays_hit_grid;
            stats->cells += jobs[t].stats.cells; stats->entry_words += jobs[t].stats.entry_words;
            stats->refs += jobs[t].stats.refs; stats->sentinels += jobs[t].stats.sentinels; stats->hits += jobs[t].stats.hits;
            stats->long_list_refs += jobs[t].stats.long_list_refs;
        }
    }
    free(jobs); free(th); free(next);
}
static void run_jobs(Job* proto, int64_t n, int nthreads, OStats* stats) { run_jobs_on(proto, n, nthreads, NULL, 0, stats); }

/* as orc_traverse_grid_mt, thread t pinned to cpus[t % num_cpus] (NULL: not pinned) */
void orc_traverse_grid_pinned(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits, int64_t n, int nthreads,
                              const int* cpus, int num_cpus, OStats* stats) {
    Job p; memset(&p, 0, sizeof(p));
    p.grid = grid; p.tris = tris; p.rays = rays; p.hits = hits; p.brute = 0;
    run_jobs_on(&p, n, nthreads, cpus, num_cpus, stats);
}

void orc_traverse_grid_mt(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits, int64_t n, int nthreads, OStats* stats) {
    Job p; memset(&p, 0, sizeof(p));
    p.grid = grid; p.tris = tris; p.rays = rays; p.hits = hits; p.brute = 0;
    run_jobs(&p, n, nthreads, stats);
}

void orc_traverse_grid_ex(const OGrid* grid, const OTri* tris, const ORay* rays, OHit* hits, int64_t n, int nthreads, unsigned flags) {
    Job p; memset(&p, 0, sizeof(p));
    p.grid = grid; p.tris = tris; p.rays = rays; p.hits = hits; p.brute = 0; p.mode = flags;
    run_jobs(&p, n, nthreads, NULL);
}

void orc_brute_force(const OTri* tris, int num_tris, const ORay* rays, OHit* hits, int64_t n, int nthreads) {
    Job p; memset(&p, 0, sizeof(p));
    p.tris = tris; p.num_tris = num_tris; p.rays = rays; p.hits = hits; p.brute = 1;
    run_jobs(&p, n, nthreads, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* structural invariants */

#define FAIL(code, ...) do { if (msg) snprintf(msg, (size_t)msg_len, __VA_ARGS__); return code; } while (0)

int orc_check_grid(const OGrid* g, const OTri* tris, int num_tris, int check_coverage, char* msg, int msg_len) {
    oivec3 vd = iv3(g->dims.x << g->shift, g->dims.y << g->shift, g->dims.z << g->shift);
    if (msg && msg_len) msg[0] = 0;
    if (g->num_offsets < 1 || g->offsets[g->num_offsets - 1] != g->num_entries) FAIL(-1, "offsets do not end at num_entries");
    int compressed = g->small_cells != NULL;
    for (int i = 0; i < g->num_cells; i++) {
        oivec3 mn, mx; int b, e;
        if (compressed) {
            OSmallCell s = g->small_cells[i];
            mn = iv3(s.min[0], s.min[1], s.min[2]); mx = iv3(s.max[0], s.max[1], s.max[2]); b = s.begin;
            e = b;
            if (b >= 0) { while (e < g->num_refs && g->ref_ids[e] >= 0) e++; if (e >= g->num_refs) FAIL(-2, "cell %d: missing sentinel", i); }
            else { b = e = 0; }
        } else { OCell c = g->cells[i]; mn = c.min; mx = c.max; b = c.begin; e = c.end; }
        if (mn.x < 0 || mn.y < 0 || mn.z < 0 || mx.x > vd.x || mx.y > vd.y || mx.z > vd.z || mn.x >= mx.x || mn.y >= mx.y || mn.z >= mx.z)
            FAIL(-3, "cell %d: bad box", i);
        if (b < 0 || e < b || e > g->num_refs) FAIL(-4, "cell %d: bad ref range [%d,%d)", i, b, e);
        for (int r = b; r < e; r++) {
            if (g->ref_ids[r] < 0 || g->ref_ids[r] >= num_tris) FAIL(-5, "cell %d: ref out of range", i);
            if (r > b && g->ref_ids[r - 1] >= g->ref_ids[r]) FAIL(-6, "cell %d: refs not strictly ascending", i);
        }
    }
    if (check_coverage) {
        /* every voxel maps to a cell whose box contains it; every triangle whose SAT test passes on
         * the voxel's own box is referenced by that cell */
        ovec3 ext = v3_sub(g->bbox.max, g->bbox.min);
        ovec3 cs = v3_div(ext, v3_from_i(vd));
        oivec3 top = g->dims;
        for (int z = 0; z < vd.z; z++) for (int y = 0; y < vd.y; y++) for (int x = 0; x < vd.x; x++) {
            oivec3 v = iv3(x, y, z);
            uint32_t ci = orc_lookup_entry(g->entries, g->shift, &top, &v, NULL);
            if ((int)ci >= g->num_cells) FAIL(-7, "voxel (%d,%d,%d): cell index out of range", x, y, z);
            oivec3 mn, mx; int b, e;
            if (compressed) {
                OSmallCell s = g->small_cells[ci];
                mn = iv3(s.min[0], s.min[1], s.min[2]); mx = iv3(s.max[0], s.max[1], s.max[2]); b = s.begin; e = b;
                if (b >= 0) while (g->ref_ids[e] >= 0) e++; else b = e = 0;
            } else { OCell c = g->cells[ci]; mn = c.min; mx = c.max; b = c.begin; e = c.end; }
            if (x < mn.x || y < mn.y || z < mn.z || x >= mx.x || y >= mx.y || z >= mx.z) FAIL(-8, "voxel (%d,%d,%d) outside its cell %u", x, y, z, ci);
            if (check_coverage > 1) {
                OBBox vb;
                vb.min = v3_add(g->bbox.min, v3_mul(v3_from_i(v), cs));
                vb.max = v3_add(g->bbox.min, v3_mul(v3_from_i(iv3(x + 1, y + 1, z + 1)), cs));
                /* shrink a little: SAT on the exact voxel box may differ in the last ulp from the SAT on
                 * the enclosing build cell */
                ovec3 eps = v3_scale(cs, 1e-3f);
                vb.min = v3_add(vb.min, eps); vb.max = v3_sub(vb.max, eps);
                for (int t = 0; t < num_tris; t++) {
                    OBBox tb; orc_tri_bbox(&tris[t], &tb);
                    if (tb.min.x > vb.max.x || tb.max.x < vb.min.x || tb.min.y > vb.max.y || tb.max.y < vb.min.y || tb.min.z > vb.max.z || tb.max.z < vb.min.z) continue;
                    if (!orc_intersect_prim_cell(&tris[t], &vb)) continue;
                    int found = 0;
                    for (int r = b; r < e && !found; r++) found = g->ref_ids[r] == t;
                    if (!found) FAIL(-9, "voxel (%d,%d,%d): triangle %d missing from cell %u", x, y, z, t, ci);
                }
            }
        }
    }
    return 0;
}

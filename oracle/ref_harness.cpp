// ref_harness.cpp -- thin extern "C" shim around the REFERENCE's own inline L0 functions.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the reference
// headers from where they lie (-I$(REF)/src, see oracle/Makefile) and is only compiled when the
// reference checkout is present.  The output (oracle/_ref/libhagrid_ref.so) is git-ignored.
//
// What is buildable: the header-only HOST DEVICE layer (common.h, vec.h, bbox.h, ray.h, prims.h,
// grid.h) under g++ with -DHOST= -DDEVICE=, exactly how the reference compiles main.cpp
// (src/CMakeLists.txt:42).  What is NOT buildable here: every .cu file (needs nvcc, the CUDA
// runtime and the un-vendored NVlabs/cub submodule) -- so there is no reference build of the
// kernels; see DESIGN.md.
//
// Uses: (1) golden known-answer vectors (tests/golden/make_golden.py), (2) brute-force nearest
// hits computed with the reference's intersect_prim_ray, the grid-independent ground truth.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <thread>
#include <vector>

// Overload environment of the device compile.  The reference's hot path is compiled by nvcc, whose
// math API declares float overloads of fabs/fmin/fmax/copysign in the GLOBAL namespace, so the
// unqualified `fabs(det)` of prims.h:273 is float arithmetic on the GPU.  Under plain g++ the same
// call would bind to C's ::fabs(double) and silently promote abs_det, w, inv_det and t to double
// (observed: hit.t differs by 1 ulp on ~25 % of hits).  Making libstdc++'s float overloads visible
// globally reproduces the device overload set; nothing of the reference is replaced.
using std::fabs;
using std::fmin;
using std::fmax;
using std::copysign;

#include "grid.h"
#include "prims.h"
#include "ray.h"
#include "bbox.h"
#include "vec.h"
#include "common.h"

using namespace hagrid;

static_assert(sizeof(Tri) == 48 && sizeof(Ray) == 32 && sizeof(Hit) == 16 && sizeof(BBox) == 32, "layout");
static_assert(sizeof(Cell) == 32 && sizeof(SmallCell) == 16 && sizeof(Entry) == 4, "layout");

extern "C" {

float ref_safe_rcp(float x) { return safe_rcp(x); }
float ref_prodsign(float x, float y) { return prodsign(x, y); }
int   ref_ilog2_i32(int t) { return ilog2(t); }
uint32_t ref_make_entry(uint32_t log_dim, uint32_t begin) { return as<uint32_t>(make_entry(log_dim, begin)); }

void ref_tri_bbox(const Tri* tri, BBox* out) { *out = tri->bbox(); out->pad0 = 0; out->pad1 = 0; }

void ref_compute_range(const int* dims, const BBox* grid_bb, const BBox* obj_bb, int* out6) {
    Range r = compute_range(ivec3(dims[0], dims[1], dims[2]), *grid_bb, *obj_bb);
    out6[0] = r.lx; out6[1] = r.ly; out6[2] = r.lz; out6[3] = r.hx; out6[4] = r.hy; out6[5] = r.hz;
}

void ref_compute_grid_dims(const BBox* bb, int num_prims, float density, int* out3) {
    ivec3 d = compute_grid_dims(*bb, num_prims, density);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

uint32_t ref_lookup_entry(const uint32_t* entries, int shift, const int* top_dims, const int* voxel) {
    return lookup_entry(reinterpret_cast<const Entry*>(entries), shift,
                        ivec3(top_dims[0], top_dims[1], top_dims[2]), ivec3(voxel[0], voxel[1], voxel[2]));
}

int ref_intersect_prim_cell(const Tri* tri, const BBox* box) { return intersect_prim_cell(*tri, *box) ? 1 : 0; }

int ref_intersect_prim_ray(const Tri* tri, const Ray* ray, int id, Hit* hit) {
    return intersect_prim_ray(*tri, *ray, id, *hit) ? 1 : 0;
}

// foreach_ref (grid.h:118-140): returns the count and writes the visited refs
int ref_foreach_ref_cell(const Cell* cell, const int* ref_ids, int* visited) {
    int n = 0;
    int r = foreach_ref(*cell, ref_ids, [&](int ref) { visited[n++] = ref; });
    return r;
}
int ref_foreach_ref_small(const SmallCell* cell, const int* ref_ids, int* visited) {
    int n = 0;
    int r = foreach_ref(*cell, ref_ids, [&](int ref) { visited[n++] = ref; });
    return r;
}

// Brute-force nearest hit with the reference's Moeller-Trumbore, triangles in ascending id order,
// tmax tightened to the current hit exactly as traverse.cu:80-83 does.  Hit.id = primitive id.
void ref_brute_force(const Tri* tris, int num_tris, const Ray* rays, Hit* hits, int64_t num_rays, int nthreads) {
    auto work = [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) {
            Hit hit(-1, rays[i].tmax, 0, 0);
            for (int t = 0; t < num_tris; t++)
                intersect_prim_ray(tris[t], Ray(rays[i].org, rays[i].tmin, rays[i].dir, hit.t), t, hit);
            hits[i] = hit;
        }
    };
    if (nthreads <= 1) { work(0, num_rays); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, num_rays * t / nthreads, num_rays * (t + 1) / nthreads);
    for (auto& t : th) t.join();
}

} // extern "C"

// hagrid/prims.h -- the triangle primitive and its two intersection routines
// (API mirror of the reference's src/prims.h: Tri :13-25, bbox :27-31, triangle/box SAT :161-264,
// Moeller-Trumbore :266-295).  The arithmetic follows the reference expression by expression -- the
// results feed integer decisions (which cells reference which triangles, which triangle is hit), so
// association order is part of the contract; see tests/golden/l0_kat.npz.
//
// Not mirrored: Tri::clipped_bounds (prims.h:33-156) -- dead code in the reference, no caller.
#ifndef HAGRID_PRIMS_H
#define HAGRID_PRIMS_H

#include <cfloat>
#include "vec.h"
#include "bbox.h"
#include "ray.h"

namespace hagrid {

/// 48 bytes: vertex 0, the two edges e1 = v0 - v1 and e2 = v2 - v0, and the (unnormalised) normal
/// cross(e1, e2) spread over the three w slots.
struct Tri {
    vec3 v0; float nx;
    vec3 e1; float ny;
    vec3 e2; float nz;

    HOST DEVICE Tri() {}
    HOST DEVICE Tri(const vec3& v0_, float nx_, const vec3& e1_, float ny_, const vec3& e2_, float nz_)
        : v0(v0_), nx(nx_), e1(e1_), ny(ny_), e2(e2_), nz(nz_) {}

    HOST DEVICE vec3 normal() const { return vec3(nx, ny, nz); }

    HOST DEVICE BBox bbox() const {
        const vec3 v1 = v0 - e1, v2 = v0 + e2;
        return BBox(min(v0, min(v1, v2)), max(v0, max(v1, v2)));
    }
};

static_assert(sizeof(Tri) == 48, "Tri must be 48 bytes");

namespace detail {

// IEEE minNum/maxNum (what fminf/fmaxf and v_min_f32/v_max_f32 compute)
HOST DEVICE inline float fmin2(float a, float b) { return __builtin_fminf(a, b); }
HOST DEVICE inline float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
HOST DEVICE inline float fabs1(float a) { return __builtin_fabsf(a); }

/// Does the plane dot(n, x) = d cut the box?  The two box corners extremal along n lie on opposite
/// sides (or on the plane).
HOST DEVICE inline bool plane_cuts_box(const vec3& n, float d, const vec3& lo, const vec3& hi) {
    const vec3 near_c(n.x > 0 ? lo.x : hi.x, n.y > 0 ? lo.y : hi.y, n.z > 0 ? lo.z : hi.z);
    const vec3 far_c(n.x <= 0 ? lo.x : hi.x, n.y <= 0 ? lo.y : hi.y, n.z <= 0 ? lo.z : hi.z);
    const float s0 = dot(n, near_c) - d;
    const float s1 = dot(n, far_c) - d;
    return s1 * s0 <= 0.0f;
}

/// Separating-axis test for the axis cross(unit(A), e): true when the projections are disjoint.
/// (B, C) are the two coordinates other than A in the cyclic order that fixes the signs.
template <int A>
HOST DEVICE inline bool edge_axis_separates(const vec3& half, const vec3& e, const vec3& f, const vec3& a, const vec3& b) {
    float p0, p1, rad;
    if (A == 0) {
        p0 = e.y * a.z - e.z * a.y; p1 = e.y * b.z - e.z * b.y; rad = f.z * half.y + f.y * half.z;
    } else if (A == 1) {
        p0 = e.z * a.x - e.x * a.z; p1 = e.z * b.x - e.x * b.z; rad = f.z * half.x + f.x * half.z;
    } else {
        p0 = e.x * a.y - e.y * a.x; p1 = e.x * b.y - e.y * b.x; rad = f.y * half.x + f.x * half.y;
    }
    return fmin2(p0, p1) > rad || fmax2(p0, p1) < -rad;
}

} // namespace detail

HOST DEVICE inline bool plane_overlap_box(const vec3& n, float d, const vec3& min, const vec3& max) {
    return detail::plane_cuts_box(n, d, min, max);
}

/// Exact triangle / box overlap: the triangle's plane, optionally the three box axes
/// (bounds_check), optionally the nine edge cross axes (cross_axes).
template <bool bounds_check, bool cross_axes>
HOST DEVICE inline bool intersect_tri_box(const vec3& v0, const vec3& e1, const vec3& e2, const vec3& n, const vec3& min, const vec3& max) {
    using namespace detail;
    if (!plane_cuts_box(n, dot(v0, n), min, max)) return false;
    const vec3 v1 = v0 - e1, v2 = v0 + e2;
    if (bounds_check) {
        if (fmin2(v0.x, fmin2(v1.x, v2.x)) > max.x || fmax2(v0.x, fmax2(v1.x, v2.x)) < min.x) return false;
        if (fmin2(v0.y, fmin2(v1.y, v2.y)) > max.y || fmax2(v0.y, fmax2(v1.y, v2.y)) < min.y) return false;
        if (fmin2(v0.z, fmin2(v1.z, v2.z)) > max.z || fmax2(v0.z, fmax2(v1.z, v2.z)) < min.z) return false;
    }
    if (cross_axes) {
        const vec3 c = (max + min) * 0.5f, half = (max - min) * 0.5f;
        const vec3 w0 = v0 - c, w1 = v1 - c, w2 = v2 - c;
        // per edge: which two vertices give distinct projections on each of the three axes
        const vec3 f1(fabs1(e1.x), fabs1(e1.y), fabs1(e1.z));
        if (edge_axis_separates<0>(half, e1, f1, w0, w2) || edge_axis_separates<1>(half, e1, f1, w0, w2) ||
            edge_axis_separates<2>(half, e1, f1, w1, w2)) return false;
        const vec3 f2(fabs1(e2.x), fabs1(e2.y), fabs1(e2.z));
        if (edge_axis_separates<0>(half, e2, f2, w0, w1) || edge_axis_separates<1>(half, e2, f2, w0, w1) ||
            edge_axis_separates<2>(half, e2, f2, w1, w2)) return false;
        const vec3 e3 = e1 + e2;
        const vec3 f3(fabs1(e3.x), fabs1(e3.y), fabs1(e3.z));
        if (edge_axis_separates<0>(half, e3, f3, w0, w2) || edge_axis_separates<1>(half, e3, f3, w0, w2) ||
            edge_axis_separates<2>(half, e3, f3, w0, w1)) return false;
    }
    return true;
}

HOST DEVICE inline bool intersect_prim_cell(const Tri& tri, const BBox& bbox) {
    return intersect_tri_box<false, true>(tri.v0, tri.e1, tri.e2, tri.normal(), bbox.min, bbox.max);
}

/// Moeller-Trumbore with the stored normal; all barycentrics are scaled by |det| and compared with
/// their signs folded in (prodsign), a hit is accepted for t in [tmin, tmax).
HOST DEVICE inline bool intersect_prim_ray(const Tri& tri, const Ray& ray, int id, Hit& hit) {
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
#ifdef COMPUTE_UVS
            hit.u = u * inv_det;
            hit.v = v * inv_det;
#endif
            hit.id = id;
            return true;
        }
    }
    return false;
}

} // namespace hagrid

#endif // HAGRID_PRIMS_H

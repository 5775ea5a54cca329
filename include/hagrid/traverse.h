// hagrid/traverse.h -- ray traversal with the reference's signatures (src/traverse.h:11-14), as
// header-only shims over the C ABI.  Work goes to the current MemManager's context (the reference keeps
// the equivalent state in per-process __constant__ symbols, traverse.cu:7-12).
#ifndef HAGRID_TRAVERSE_H
#define HAGRID_TRAVERSE_H

#include "build.h"
#include "grid.h"
#include "vec.h"
#include "prims.h"

namespace hagrid {

/// Prepares the traversal of `grid` (traverse.cu:97-109): validates it and builds the context's traversal image
/// (hagrid_amd.h: hagrid_setup_traversal).  Call it again whenever the grid was rebuilt, as the reference's front-end does.
inline void setup_traversal(const Grid& grid) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(detail::current_ctx(), hagrid_setup_traversal(detail::current_ctx(), &p));
}

/// Nearest hit per ray: hits[i].id = primitive id or -1, hits[i].t = distance.  Asynchronous on the
/// context's stream, like a kernel launch.
inline void traverse_grid(const Grid& grid, const Tri* tris, const Ray* rays, Hit* hits, int num_rays) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(detail::current_ctx(), hagrid_traverse_grid(detail::current_ctx(), &p, tris, rays, hits, num_rays));
}

/// Extensions over the same walk (no counterpart in src/traverse.h): occlusion rays stop at their first accepted
/// intersection (hits[i].id >= 0 <=> occluded); `with_uvs` stores the barycentrics like a COMPUTE_UVS build (prims.h:285-288).
inline void traverse_grid_any_hit(const Grid& grid, const Tri* tris, const Ray* rays, Hit* hits, int num_rays) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(detail::current_ctx(), hagrid_traverse_grid_ex(detail::current_ctx(), &p, tris, rays, hits, num_rays, HAGRID_TRAVERSE_ANY_HIT));
}
inline void traverse_grid_with_uvs(const Grid& grid, const Tri* tris, const Ray* rays, Hit* hits, int num_rays) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(detail::current_ctx(), hagrid_traverse_grid_ex(detail::current_ctx(), &p, tris, rays, hits, num_rays, HAGRID_TRAVERSE_UVS));
}

/// Extension: independent batches in flight.  Every MemManager is a context with a stream of its own (hagrid_ctx_set_stream on
/// mem.context()); `share_traversal(dst, src)` lets `dst` traverse with the traversal image setup_traversal built in `src`, and the
/// overload below traverses on a named manager instead of the current one.  Two 1M-ray batches in flight take 0.118 ms each
/// instead of 0.177 ms (hagrid_amd.h: hagrid_share_traversal).
inline void share_traversal(MemManager& dst, MemManager& src) {
    detail::check(dst.context(), hagrid_share_traversal(dst.context(), src.context()));
}
inline void traverse_grid(MemManager& on, const Grid& grid, const Tri* tris, const Ray* rays, Hit* hits, int num_rays) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(on.context(), hagrid_traverse_grid(on.context(), &p, tris, rays, hits, num_rays));
}

} // namespace hagrid

#endif // HAGRID_TRAVERSE_H

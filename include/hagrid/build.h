// hagrid/build.h -- grid construction with the reference's signatures (src/build.h:17-31), as
// header-only shims over the C ABI (include/hagrid_amd.h).  Grid <-> hagrid_grid is a field copy.
#ifndef HAGRID_BUILD_H
#define HAGRID_BUILD_H

#include "mem_manager.h"
#include "prims.h"
#include "grid.h"

namespace hagrid {

namespace detail {
inline hagrid_grid to_pod(const Grid& g) {
    hagrid_grid p = hagrid_grid();
    p.entries = g.entries; p.ref_ids = g.ref_ids; p.cells = g.cells; p.small_cells = g.small_cells;
    p.bbox_min[0] = g.bbox.min.x; p.bbox_min[1] = g.bbox.min.y; p.bbox_min[2] = g.bbox.min.z;
    p.bbox_max[0] = g.bbox.max.x; p.bbox_max[1] = g.bbox.max.y; p.bbox_max[2] = g.bbox.max.z;
    p.dims[0] = g.dims.x; p.dims[1] = g.dims.y; p.dims[2] = g.dims.z;
    p.num_cells = g.num_cells; p.num_entries = g.num_entries; p.num_refs = g.num_refs; p.shift = g.shift;
    p.num_offsets = int(g.offsets.size()) < HAGRID_MAX_LEVELS ? int(g.offsets.size()) : HAGRID_MAX_LEVELS;
    for (int i = 0; i < p.num_offsets; i++) p.offsets[i] = g.offsets[i];
    return p;
}
inline void from_pod(Grid& g, const hagrid_grid& p) {
    g.entries = static_cast<Entry*>(p.entries); g.ref_ids = static_cast<int*>(p.ref_ids);
    g.cells = static_cast<Cell*>(p.cells); g.small_cells = static_cast<SmallCell*>(p.small_cells);
    g.bbox = BBox(vec3(p.bbox_min[0], p.bbox_min[1], p.bbox_min[2]), vec3(p.bbox_max[0], p.bbox_max[1], p.bbox_max[2]));
    g.dims = ivec3(p.dims[0], p.dims[1], p.dims[2]);
    g.num_cells = p.num_cells; g.num_entries = p.num_entries; g.num_refs = p.num_refs; g.shift = p.shift;
    g.offsets.assign(p.offsets, p.offsets + p.num_offsets);
}
} // namespace detail

/// Builds the initial irregular grid: a uniform top level of density top_density, an independent
/// octree depth per top-level cell from snd_density, references split down to each cell's depth.
inline void build_grid(MemManager& mem, const Tri* tris, int num_tris, Grid& grid, float top_density, float snd_density) {
    hagrid_grid p = hagrid_grid();
    detail::check(mem.context(), hagrid_build_grid(mem.context(), tris, num_tris, &p, top_density, snd_density));
    detail::from_pod(grid, p);
}

/// Neighbour merging guided by the surface area heuristic.
inline void merge_grid(MemManager& mem, Grid& grid, float alpha) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(mem.context(), hagrid_merge_grid(mem.context(), &p, alpha));
    detail::from_pod(grid, p);
}

/// Fuses up to three octree levels of the voxel map per node.
inline void flatten_grid(MemManager& mem, Grid& grid) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(mem.context(), hagrid_flatten_grid(mem.context(), &p));
    detail::from_pod(grid, p);
}

/// Grows cells over neighbours whose references are a subset of their own.
inline void expand_grid(MemManager& mem, Grid& grid, const Tri* tris, int iters) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(mem.context(), hagrid_expand_grid(mem.context(), &p, tris, iters));
    detail::from_pod(grid, p);
}

/// 16-bit cells + sentinel-terminated reference lists; false when the grid is too fine for 16 bits.
inline bool compress_grid(MemManager& mem, Grid& grid) {
    hagrid_grid p = detail::to_pod(grid);
    const int rc = hagrid_compress_grid(mem.context(), &p);
    detail::check(mem.context(), rc);
    if (rc == 1) detail::from_pod(grid, p);
    return rc == 1;
}

} // namespace hagrid

#endif // HAGRID_BUILD_H

// hagrid/grid.h -- the irregular grid data layout (API mirror of the reference's src/grid.h).
//
//   voxel --lookup_entry--> Entry chain (voxel map) --> Cell (integer AABB on the virtual grid
//   dims << shift, plus a [begin, end) range into ref_ids) --> primitive ids.
//
// Layout contract (bytes, bit positions) is identical to the reference: Entry 4 B with log_dim in
// bits 0-1 and begin in bits 2-31 (grid.h:12-20), Cell 32 B (:23-33), SmallCell 16 B (:36-45).
#ifndef HAGRID_GRID_H
#define HAGRID_GRID_H

#include <vector>
#include "vec.h"
#include "bbox.h"

namespace hagrid {

/// Voxel map word.  log_dim == 0: leaf, begin = cell index.  log_dim = k > 0: inner node whose
/// (2^k)^3 children start at entry index begin, x fastest.
struct Entry {
    enum { LOG_DIM_BITS = 2, BEGIN_BITS = 32 - LOG_DIM_BITS };
    uint32_t log_dim : LOG_DIM_BITS;
    uint32_t begin : BEGIN_BITS;
};

struct Cell {
    ivec3 min; int begin;   ///< lower corner (virtual grid units), first reference
    ivec3 max; int end;     ///< upper corner (exclusive), past-the-end reference
    HOST DEVICE Cell() {}
    HOST DEVICE Cell(const ivec3& lo, int b, const ivec3& hi, int e) : min(lo), begin(b), max(hi), end(e) {}
};

/// Compressed cell: 16-bit corners; its reference list ends with a -1 sentinel, begin = -1 if empty.
struct SmallCell {
    usvec3 min; usvec3 max; int begin;
    HOST DEVICE SmallCell() {}
    HOST DEVICE SmallCell(const usvec3& lo, const usvec3& hi, int b) : min(lo), max(hi), begin(b) {}
};

static_assert(sizeof(Entry) == 4 && sizeof(Cell) == 32 && sizeof(SmallCell) == 16, "grid record layout");

struct Grid {
    Entry* entries;           ///< voxel map (device)
    int* ref_ids;             ///< primitive references (device)
    Cell* cells;              ///< cells (device), nullptr once compressed
    SmallCell* small_cells;   ///< compressed cells (device), nullptr unless compressed
    BBox bbox;                ///< grid bounding box (scene box enlarged by 0.1 %)
    ivec3 dims;               ///< top-level resolution
    int num_cells, num_entries, num_refs;
    int shift;                ///< log2 of the finest subdivision: virtual resolution = dims << shift
    std::vector<int> offsets; ///< cumulative entry count per voxel-map level
};

struct Range {
    int lx, ly, lz, hx, hy, hz;
    HOST DEVICE Range() {}
    HOST DEVICE Range(int lx_, int ly_, int lz_, int hx_, int hy_, int hz_) : lx(lx_), ly(ly_), lz(lz_), hx(hx_), hy(hy_), hz(hz_) {}
    HOST DEVICE int size() const { return (hx - lx + 1) * (hy - ly + 1) * (hz - lz + 1); }
};

HOST DEVICE inline Entry make_entry(uint32_t log_dim, uint32_t begin) {
    Entry e; e.log_dim = log_dim; e.begin = begin; return e;
}

/// Inclusive range of grid cells touched by obj_bb, clamped to the grid (truncating casts).
HOST DEVICE inline Range compute_range(const ivec3& dims, const BBox& grid_bb, const BBox& obj_bb) {
    const vec3 inv = vec3(dims) / grid_bb.extents();
    return Range(max(int((obj_bb.min.x - grid_bb.min.x) * inv.x), 0),
                 max(int((obj_bb.min.y - grid_bb.min.y) * inv.y), 0),
                 max(int((obj_bb.min.z - grid_bb.min.z) * inv.z), 0),
                 min(int((obj_bb.max.x - grid_bb.min.x) * inv.x), dims.x - 1),
                 min(int((obj_bb.max.y - grid_bb.min.y) * inv.y), dims.y - 1),
                 min(int((obj_bb.max.z - grid_bb.min.z) * inv.z), dims.z - 1));
}

/// Resolution for num_prims primitives in bb at the given density (Cleary's formula);
/// the cube root is the deterministic det_cbrtf so host and device agree.
HOST DEVICE inline ivec3 compute_grid_dims(const BBox& bb, int num_prims, float density) {
    const vec3 e = bb.extents();
    const float volume = e.x * e.y * e.z;
    const float ratio = det_cbrtf(density * num_prims / volume);
    return max(ivec3(1), ivec3(int(e.x * ratio), int(e.y * ratio), int(e.z * ratio)));
}

/// Walks the voxel map from the top-level entry of `voxel` (virtual grid coordinates) down to a leaf
/// and returns the cell index.  dims = top-level resolution.
HOST DEVICE inline uint32_t lookup_entry(const Entry* entries, int shift, const ivec3& dims, const ivec3& voxel) {
    const uint32_t* words = reinterpret_cast<const uint32_t*>(entries);
    uint32_t w = words[(voxel.x >> shift) + dims.x * ((voxel.y >> shift) + dims.y * (voxel.z >> shift))];
    int depth = 0;
    while (w & 3u) {
        const int k = int(w & 3u);
        depth += k;
        const int s = shift - depth, m = (1 << k) - 1;
        const int cx = (voxel.x >> s) & m, cy = (voxel.y >> s) & m, cz = (voxel.z >> s) & m;
        w = words[(w >> 2) + cx + ((cy + (cz << k)) << k)];
    }
    return w >> 2;
}

/// Calls f(ref) for every reference of the cell; returns how many words of ref_ids were consumed.
template <typename F>
HOST DEVICE inline int foreach_ref(Cell cell, const int* ref_ids, F f) {
    for (int i = cell.begin; i < cell.end; i++) {
        const int ref = ref_ids[i];
        if (ref < 0) break;
        f(ref);
    }
    return cell.end - cell.begin;
}

template <typename F>
HOST DEVICE inline int foreach_ref(SmallCell cell, const int* ref_ids, F f) {
    if (cell.begin < 0) return 0;
    int i = cell.begin;
    for (;;) {
        const int ref = ref_ids[i++];
        if (ref < 0) break;
        f(ref);
    }
    return i - cell.begin;
}

} // namespace hagrid

#endif // HAGRID_GRID_H

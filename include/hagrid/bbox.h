// hagrid/bbox.h -- axis-aligned box, 32 bytes so that it moves as two 16-byte words.
// API mirror of the reference's BBox (src/bbox.h:10-85): same members, same method names and results.
// The set predicates are written once, over component-wise helpers.
#ifndef HAGRID_BBOX_H
#define HAGRID_BBOX_H

#include <cfloat>
#include "vec.h"

namespace hagrid {

namespace detail {
/// a <= b in every component (false as soon as a NaN is involved)
HOST DEVICE inline bool all_le(const vec3& a, const vec3& b) { return a.x <= b.x && a.y <= b.y && a.z <= b.z; }
/// a < b in at least one component
HOST DEVICE inline bool any_lt(const vec3& a, const vec3& b) { return a.x < b.x || a.y < b.y || a.z < b.z; }
HOST DEVICE inline float clamp0(float v) { return v > 0.0f ? v : 0.0f; }
} // namespace detail

struct BBox {
    vec3 min;
    int pad0;
    vec3 max;
    int pad1;

    HOST DEVICE BBox() {}
    HOST DEVICE BBox(const vec3& point) : min(point), max(point) {}
    HOST DEVICE BBox(const vec3& lower, const vec3& upper) : min(lower), max(upper) {}

    // ---- growing / intersecting (all return *this for chaining) ----
    HOST DEVICE BBox& extend(const vec3& point) { return grow(point, point); }
    HOST DEVICE BBox& extend(const BBox& other) { return grow(other.min, other.max); }
    HOST DEVICE BBox& overlap(const BBox& other) {
        min = hagrid::max(min, other.min);
        max = hagrid::min(max, other.max);
        return *this;
    }

    // ---- measures ----
    HOST DEVICE vec3 extents() const { return max - min; }
    HOST DEVICE vec3 center() const { return 0.5f * (max + min); }
    /// half of the surface area, negative extents counted as zero
    HOST DEVICE float half_area() const {
        const vec3 e = extents();
        const float ex = detail::clamp0(e.x), ey = detail::clamp0(e.y), ez = detail::clamp0(e.z);
        return ex * (ey + ez) + ey * ez;
    }

    // ---- predicates ----
    HOST DEVICE bool is_empty() const { return detail::any_lt(max, min); }
    HOST DEVICE bool is_inside(const vec3& point) const { return detail::all_le(min, point) && detail::all_le(point, max); }
    HOST DEVICE bool is_overlapping(const BBox& other) const { return detail::all_le(min, other.max) && detail::all_le(other.min, max); }
    HOST DEVICE bool is_included(const BBox& outer) const { return detail::all_le(outer.min, min) && detail::all_le(max, outer.max); }
    HOST DEVICE bool is_strictly_included(const BBox& outer) const {
        return is_included(outer) && (detail::any_lt(outer.min, min) || detail::any_lt(max, outer.max));
    }

    /// the neutral element of extend() and of overlap()
    HOST DEVICE static BBox empty() { return BBox(vec3(FLT_MAX), vec3(-FLT_MAX)); }
    HOST DEVICE static BBox full() { return BBox(vec3(-FLT_MAX), vec3(FLT_MAX)); }

private:
    HOST DEVICE BBox& grow(const vec3& lower, const vec3& upper) {
        min = hagrid::min(min, lower);
        max = hagrid::max(max, upper);
        return *this;
    }
};

static_assert(sizeof(BBox) == 32, "BBox must stay 32 bytes");

} // namespace hagrid

#endif // HAGRID_BBOX_H

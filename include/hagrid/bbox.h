// hagrid/bbox.h -- axis-aligned bounding box, 32 bytes so it moves as two 16-byte words
// (API mirror of the reference's src/bbox.h:10-85).
#ifndef HAGRID_BBOX_H
#define HAGRID_BBOX_H

#include <cfloat>
#include "vec.h"

namespace hagrid {

struct BBox {
    vec3 min; int pad0;
    vec3 max; int pad1;

    HOST DEVICE BBox() {}
    HOST DEVICE BBox(const vec3& p) : min(p), max(p) {}
    HOST DEVICE BBox(const vec3& lo, const vec3& hi) : min(lo), max(hi) {}

    HOST DEVICE BBox& extend(const vec3& p) { min = hagrid::min(min, p); max = hagrid::max(max, p); return *this; }
    HOST DEVICE BBox& extend(const BBox& o) { min = hagrid::min(min, o.min); max = hagrid::max(max, o.max); return *this; }
    HOST DEVICE BBox& overlap(const BBox& o) { min = hagrid::max(min, o.min); max = hagrid::min(max, o.max); return *this; }

    HOST DEVICE vec3 extents() const { return max - min; }
    HOST DEVICE vec3 center() const { return 0.5f * (max + min); }
    HOST DEVICE float half_area() const {
        const vec3 d = max - min;
        const float a = hagrid::max(d.x, 0.0f), b = hagrid::max(d.y, 0.0f), c = hagrid::max(d.z, 0.0f);
        return a * (b + c) + b * c;
    }

    HOST DEVICE bool is_empty() const { return min.x > max.x || min.y > max.y || min.z > max.z; }
    HOST DEVICE bool is_inside(const vec3& p) const {
        return p.x >= min.x && p.y >= min.y && p.z >= min.z && p.x <= max.x && p.y <= max.y && p.z <= max.z;
    }
    HOST DEVICE bool is_overlapping(const BBox& o) const {
        return min.x <= o.max.x && max.x >= o.min.x && min.y <= o.max.y && max.y >= o.min.y && min.z <= o.max.z && max.z >= o.min.z;
    }
    HOST DEVICE bool is_included(const BBox& o) const {
        return min.x >= o.min.x && max.x <= o.max.x && min.y >= o.min.y && max.y <= o.max.y && min.z >= o.min.z && max.z <= o.max.z;
    }
    HOST DEVICE bool is_strictly_included(const BBox& o) const {
        return is_included(o) && (min.x > o.min.x || max.x < o.max.x || min.y > o.min.y || max.y < o.max.y || min.z > o.min.z || max.z < o.max.z);
    }

    HOST DEVICE static BBox empty() { return BBox(vec3(FLT_MAX), vec3(-FLT_MAX)); }
    HOST DEVICE static BBox full() { return BBox(vec3(-FLT_MAX), vec3(FLT_MAX)); }
};

static_assert(sizeof(BBox) == 32, "BBox must be 32 bytes");

} // namespace hagrid

#endif // HAGRID_BBOX_H

// hagrid/ray.h -- Ray (32 B) and Hit (16 B) records (API mirror of the reference's src/ray.h:9-33).
//
// Deliberate difference from the reference binary: traverse_grid leaves the PRIMITIVE id in Hit::id
// (-1 when nothing was hit), which is what this struct documents and what intersect_prim_ray stores;
// the reference kernel overwrites it with its step counter (traverse.cu:93).  The step counter is
// available through hagrid_traverse_grid_stats (include/hagrid_amd.h).
#ifndef HAGRID_RAY_H
#define HAGRID_RAY_H

#include "vec.h"

namespace hagrid {

/// org + t * dir, t in [tmin, tmax]
struct Ray {
    vec3 org; float tmin;
    vec3 dir; float tmax;
    HOST DEVICE Ray() {}
    HOST DEVICE Ray(const vec3& o, float t0, const vec3& d, float t1) : org(o), tmin(t0), dir(d), tmax(t1) {}
};

/// id is -1 if there is no hit
struct Hit {
    int id; float t, u, v;
    HOST DEVICE Hit() {}
    HOST DEVICE Hit(int id_, float t_, float u_, float v_) : id(id_), t(t_), u(u_), v(v_) {}
};

static_assert(sizeof(Ray) == 32 && sizeof(Hit) == 16, "Ray/Hit layout");

} // namespace hagrid

#endif // HAGRID_RAY_H

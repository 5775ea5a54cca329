// hagrid/vec.h -- small fixed-size vectors (API mirror of the reference's src/vec.h).
//
// Same type names (tvec2/tvec3, vec2, vec3, ivec2, ivec3, usvec2, usvec3), same component-wise
// operator set (+ - * / << >> & | between vectors and between a vector and a scalar on either side),
// same free functions (min, max, clamp, dot, length, normalize, cross, rotate, get<axis>).  Floating
// point expressions keep the reference's association order: dot is (x*x' + y*y') + z*z'.
#ifndef HAGRID_VEC_H
#define HAGRID_VEC_H

#include "common.h"

namespace hagrid {

template <typename T>
struct tvec2 {
    T x, y;
    HOST DEVICE tvec2() {}
    HOST DEVICE tvec2(T s) : x(s), y(s) {}
    HOST DEVICE tvec2(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> HOST DEVICE explicit tvec2(const tvec2<U>& o) : x(T(o.x)), y(T(o.y)) {}
};

template <typename T>
struct tvec3 {
    union { T x; T r; };
    union { T y; T g; };
    union { T z; T b; };
    HOST DEVICE tvec3() {}
    HOST DEVICE tvec3(T s) : x(s), y(s), z(s) {}
    HOST DEVICE tvec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    template <typename U> HOST DEVICE explicit tvec3(const tvec3<U>& o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
};

// One macro instantiates an operator for both arities; E2/E3 expand the per-component expression.
#define HAGRID_VEC_BINOP(OP)                                                                                          \
    template <typename T> HOST DEVICE inline tvec2<T> operator OP(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x OP b.x, a.y OP b.y); } \
    template <typename T> HOST DEVICE inline tvec2<T> operator OP(const tvec2<T>& a, T s) { return tvec2<T>(a.x OP s, a.y OP s); }                  \
    template <typename T> HOST DEVICE inline tvec2<T> operator OP(T s, const tvec2<T>& b) { return tvec2<T>(s OP b.x, s OP b.y); }                  \
    template <typename T> HOST DEVICE inline tvec3<T> operator OP(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
    template <typename T> HOST DEVICE inline tvec3<T> operator OP(const tvec3<T>& a, T s) { return tvec3<T>(a.x OP s, a.y OP s, a.z OP s); }       \
    template <typename T> HOST DEVICE inline tvec3<T> operator OP(T s, const tvec3<T>& b) { return tvec3<T>(s OP b.x, s OP b.y, s OP b.z); }
HAGRID_VEC_BINOP(+)
HAGRID_VEC_BINOP(-)
HAGRID_VEC_BINOP(*)
HAGRID_VEC_BINOP(/)
HAGRID_VEC_BINOP(<<)
HAGRID_VEC_BINOP(>>)
HAGRID_VEC_BINOP(&)
HAGRID_VEC_BINOP(|)
#undef HAGRID_VEC_BINOP

#define HAGRID_VEC_ASSIGN(OP)                                                                                        \
    template <typename T> HOST DEVICE inline tvec2<T>& operator OP##=(tvec2<T>& a, const tvec2<T>& b) { a = a OP b; return a; } \
    template <typename T> HOST DEVICE inline tvec3<T>& operator OP##=(tvec3<T>& a, const tvec3<T>& b) { a = a OP b; return a; }
HAGRID_VEC_ASSIGN(+)
HAGRID_VEC_ASSIGN(-)
HAGRID_VEC_ASSIGN(*)
HAGRID_VEC_ASSIGN(/)
#undef HAGRID_VEC_ASSIGN
template <typename T> HOST DEVICE inline tvec2<T>& operator*=(tvec2<T>& a, T s) { a = a * s; return a; }
template <typename T> HOST DEVICE inline tvec2<T>& operator/=(tvec2<T>& a, T s) { a = a / s; return a; }
template <typename T> HOST DEVICE inline tvec3<T>& operator*=(tvec3<T>& a, T s) { a = a * s; return a; }
template <typename T> HOST DEVICE inline tvec3<T>& operator/=(tvec3<T>& a, T s) { a = a / s; return a; }

template <typename T> HOST DEVICE inline tvec2<T> min(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(min(a.x, b.x), min(a.y, b.y)); }
template <typename T> HOST DEVICE inline tvec2<T> max(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(max(a.x, b.x), max(a.y, b.y)); }
template <typename T> HOST DEVICE inline tvec3<T> min(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
template <typename T> HOST DEVICE inline tvec3<T> max(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
template <typename T> HOST DEVICE inline tvec2<T> clamp(const tvec2<T>& v, T lo, T hi) { return tvec2<T>(min(max(v.x, lo), hi), min(max(v.y, lo), hi)); }
template <typename T> HOST DEVICE inline tvec3<T> clamp(const tvec3<T>& v, T lo, T hi) { return tvec3<T>(min(max(v.x, lo), hi), min(max(v.y, lo), hi), min(max(v.z, lo), hi)); }

template <typename T> HOST DEVICE inline T dot(const tvec2<T>& a, const tvec2<T>& b) { return a.x * b.x + a.y * b.y; }
template <typename T> HOST DEVICE inline T dot(const tvec3<T>& a, const tvec3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> HOST DEVICE inline T length(const tvec2<T>& a) { return std::sqrt(dot(a, a)); }
template <typename T> HOST DEVICE inline T length(const tvec3<T>& a) { return std::sqrt(dot(a, a)); }
template <typename T> HOST DEVICE inline tvec2<T> normalize(const tvec2<T>& a) { return a * (1.0f / length(a)); }
template <typename T> HOST DEVICE inline tvec3<T> normalize(const tvec3<T>& a) { return a * (1.0f / length(a)); }

template <typename T>
HOST DEVICE inline tvec3<T> cross(const tvec3<T>& a, const tvec3<T>& b) {
    return tvec3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/// Rotation of v about a unit axis by angle (radians), via the quaternion sandwich q v q*.
template <typename T>
HOST DEVICE inline tvec3<T> rotate(const tvec3<T>& v, const tvec3<T>& axis, T angle) {
    const T h = angle / 2;
    const T s = std::sin(h), w = std::cos(h);
    const tvec3<T> u(axis.x * s, axis.y * s, axis.z * s);
    // t = q * (0, v)
    const tvec3<T> tv(w * v.x + u.y * v.z - u.z * v.y,
                      w * v.y - u.x * v.z + u.z * v.x,
                      w * v.z + u.x * v.y - u.y * v.x);
    const T tw = -(u.x * v.x + u.y * v.y + u.z * v.z);
    // t * conj(q)
    return tvec3<T>(tw * -u.x + tv.x * w + tv.y * -u.z - tv.z * -u.y,
                    tw * -u.y - tv.x * -u.z + tv.y * w + tv.z * -u.x,
                    tw * -u.z + tv.x * -u.y - tv.y * -u.x + tv.z * w);
}

template <int axis, typename T> HOST DEVICE inline T get(const tvec2<T>& v) { return axis == 0 ? v.x : v.y; }
template <int axis, typename T> HOST DEVICE inline T get(const tvec3<T>& v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

typedef tvec2<float> vec2;
typedef tvec2<int> ivec2;
typedef tvec2<unsigned short> usvec2;
typedef tvec3<float> vec3;
typedef tvec3<int> ivec3;
typedef tvec3<unsigned short> usvec3;

} // namespace hagrid

#endif // HAGRID_VEC_H

// vec3.h — fp64 3-vector helpers for the host side of the hip way.
//
// The host only needs vectors for scene construction, the camera constructor and the
// per-primitive precompute; the per-sample arithmetic lives in csrc/ptw_device.h.  Operation
// order matters because these values feed the device path and must round like the reference
// (src/math/Vec3.h: dot = x*x' + y*y' + z*z', division = multiply by reciprocal).  All host
// translation units are compiled with -ffp-contract=off.
#pragma once

#include <cmath>

namespace ptw {

struct Vec3d {
  double x{}, y{}, z{};
  constexpr Vec3d() = default;
  constexpr Vec3d(double xx, double yy, double zz) : x(xx), y(yy), z(zz) {}
  explicit Vec3d(const double *p) : x(p[0]), y(p[1]), z(p[2]) {}
  void store(double *p) const { p[0] = x, p[1] = y, p[2] = z; }
};

inline Vec3d operator+(Vec3d a, Vec3d b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3d operator-(Vec3d a, Vec3d b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3d operator-(Vec3d a) { return {-a.x, -a.y, -a.z}; }
inline Vec3d operator*(Vec3d a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3d operator*(double s, Vec3d a) { return {s * a.x, s * a.y, s * a.z}; }
inline bool operator==(Vec3d a, Vec3d b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline double dot(Vec3d a, Vec3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3d cross(Vec3d a, Vec3d b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(Vec3d a) { return std::sqrt(dot(a, a)); }
// v * (1 / |v|): the reference divides a Vec3 by a scalar via the reciprocal (Vec3.h:51-54).
inline Vec3d normalised(Vec3d a) {
  const double reciprocal = 1.0 / length(a);
  return {a.x * reciprocal, a.y * reciprocal, a.z * reciprocal};
}

} // namespace ptw

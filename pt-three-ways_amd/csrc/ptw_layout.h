// ptw_layout.h — HBM data layout of the hip way (plain structs, no HIP headers) shared by the
// host-side precompute (host/precompute.cpp, strict fp64) and the kernels.
#pragma once

#include <cstdint>

namespace ptw {

// Per-triangle shading record, fetched once per hit with wave-uniform (scalar) loads.
// dod::Scene stores faceNormal() three times per triangle (src/dod/Scene.cpp:181-187), so the
// interpolated normal of intersectTriangles (Scene.cpp:100-106) is a per-triangle constant;
// so is OrthoNormalBasis::fromZ(normal).  Both are precomputed on the host in strict fp64
// (csrc/capi_render.hip: precomputeTriangle) together with a copy of the material.
// A back-facing hit uses (-normal, -basisX, basisY): negation commutes exactly with every
// operation of fromZ.
struct alignas(64) TriShade {
  double normal[3];
  double basisX[3];
  double basisY[3];
  double emission[3];
  double diffuse[3];
  double ior;
  double invIor; // 1.0 / ior
  double reflectivity;
  double coneAngle;
  double pad[5];
}; // 24 doubles = 192 B

struct alignas(64) SphereRec {
  double centre[3];
  double radiusSquared; // radius * radius, src/dod/Sphere.h:11-12
  double emission[3];
  double diffuse[3];
  double ior;
  double invIor;
  double reflectivity;
  double coneAngle;
  double pad[2];
}; // 16 doubles = 128 B

// Compact per-triangle record for the LDS-resident shading table of the SEQUENTIAL kernel:
// normal, basisX, basisY (as in TriShade), the material index as a double, and the material's
// "lobe threshold" (kTriLobeThreshold) so the common diffuse bounce needs no second, dependent
// fetch of the material:
//   reflectivity >= 0            -> the reflectivity itself: diffuse lobe iff !(p < threshold)
//   reflectivity < 0, ior == 1   -> -1: Norm3::reflectance is below the spacing of the draws
//                                   unless the ray grazes (see lobeIsReflective), so the lobe is
//                                   diffuse iff cos(theta_i) >= 1e-3 and p > 0 - else evaluate
//   reflectivity < 0, ior != 1   -> 2: always evaluate Norm3::reflectance
// One pad double keeps records 16-byte aligned (96 B) for ds_read_b128.  Materials sit in their
// own table of kMatDoubles doubles each: emission, diffuse, ior, 1/ior, reflectivity, cone angle.
constexpr int kTriCompactDoubles = 12;
constexpr int kTriMaterialIndex = 9;
constexpr int kTriLobeThreshold = 10;
constexpr int kMatDoubles = 10;

constexpr int kMtWords = 624;
constexpr int kMtDoubles = 312; // canonical doubles per regeneration (2 words each)
constexpr int kMaxDepth = 64;
// traceSequentialSpec parks, per pass: the canonical doubles of both ring slots (2 x 316), the
// slot and position of the stream frontier.
constexpr int kSpecStateDoubles = 2 * 316 + 2;   // radiance stack capacity (levels kept in LDS)

} // namespace ptw

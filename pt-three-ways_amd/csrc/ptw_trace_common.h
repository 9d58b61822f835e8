// ptw_trace_common.h — device code shared by the radiance kernels of csrc/ (ptw_launch.h lists the kernel
// files): nearest-hit primitives (Scene::intersect*, src/dod/Scene.cpp:13-113), the camera
// ray (src/math/Camera.h:20-60) and the std::mt19937 stream ring of the speculative kernels.
#pragma once

#include "ptw_device.h"
#include "ptw_kernels.h"

namespace ptw {
using namespace ptwd;

namespace {

constexpr uint32_t kMiss = 0xffffffffu;

struct HitKey {
  double t;     // distance along the ray (+inf on a miss)
  uint32_t idx; // combined primitive index: spheres [0, nsph), triangles nsph + k; kMiss
  double det;   // Moller-Trumbore determinant of the winning triangle (backface test)
};

// What radiance() needs to know about the surface at a hit (Scene.cpp:135-152).
struct Surface {
  d3 pos;
  d3 normal;
  Basis basis;
  double reflectivity;    // resolved value (eager paths)
  d3 emission;
  d3 diffuse;
  double coneAngle;
  // inputs of Norm3::reflectance for paths that resolve the lobe lazily
  double matReflectivity; // MaterialSpec::reflectivity (< 0 => Fresnel-ish)
  double iorFrom, iorTo, iorRatio;
};

__device__ __forceinline__ double resolveReflectivity(const Surface &s, d3 dirIn) {
  return s.matReflectivity < 0 ? reflectance(s.normal, dirIn, s.iorFrom, s.iorTo, s.iorRatio)
                               : s.matReflectivity;
}

__device__ __forceinline__ d3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

// One Moller-Trumbore test, Scene.cpp:62-98, against a triangle given as v0, e1, e2.
// Updates (bestT, bestIdx, bestDet) when this triangle is a strictly nearer acceptable hit.
__device__ __forceinline__ void testTriangle(d3 o, d3 d, d3 v0, d3 e1, d3 e2, uint32_t idx,
                                             double &bestT, uint32_t &bestIdx, double &bestDet) {
  const d3 pVec = cross(d, e2);
  const double det = dot(e1, pVec);
  if (__builtin_fabs(det) < kEpsilon) return;
  const double invDet = rcp(det);
  const d3 tVec = o - v0;
  const double u = dot(tVec, pVec) * invDet;
  const d3 qVec = cross(tVec, e1);
  const double v = dot(d, qVec) * invDet;
  if ((u < 0.0) | (u > 1.0) | (v < 0.0) | (u + v > 1)) return;
  const double t = dot(e2, qVec) * invDet;
  if (t > kEpsilon && t < bestT) {
    bestT = t;
    bestIdx = idx;
    bestDet = det;
  }
}

// The same test for kernels in which all lanes of a wave test the SAME triangle (PERPIXEL policy):
// u is decided first.  The reference rejects on (u < 0 | u > 1 | v < 0 | u + v > 1) as one fused test
// (Scene.cpp:89); a triangle rejected on u is rejected whatever v is, so when no lane passes the u
// test the wave skips qVec, v and t - same decisions, same values.
// -DPTW_U_FIRST=0: the fused test everywhere (A/B builds: make alt ALT_FLAGS=-DPTW_U_FIRST=0)
#ifndef PTW_U_FIRST
#define PTW_U_FIRST 1
#endif
__device__ __forceinline__ void testTriangleUFirst(d3 o, d3 d, d3 v0, d3 e1, d3 e2, uint32_t idx,
                                                   double &bestT, uint32_t &bestIdx, double &bestDet) {
#if !PTW_U_FIRST
  testTriangle(o, d, v0, e1, e2, idx, bestT, bestIdx, bestDet);
  return;
#endif
  const d3 pVec = cross(d, e2);
  const double det = dot(e1, pVec);
  if (__builtin_fabs(det) < kEpsilon) return;
  const double invDet = rcp(det);
  const d3 tVec = o - v0;
  const double u = dot(tVec, pVec) * invDet;
  if ((u < 0.0) | (u > 1.0)) return;
  const d3 qVec = cross(tVec, e1);
  const double v = dot(d, qVec) * invDet;
  if ((v < 0.0) | (u + v > 1)) return;
  const double t = dot(e2, qVec) * invDet;
  if (t > kEpsilon && t < bestT) {
    bestT = t;
    bestIdx = idx;
    bestDet = det;
  }
}

// The same test for the worker waves of the SEQUENTIAL kernels, in which the 64 lanes of a wave test 64 CONSECUTIVE
// triangles (one unit) against one ray.  The faces of a mesh follow each other in space, so for most rays NO triangle
// of a unit passes the u test (scripts/sim/unit_skip_stats.py: ce 67-70 % of the units, suzanne 17-30 %; 2.7 % of the
// triangles pass it) - and then the wave skips qVec, v and t for the unit: one ballot and one branch.  Same decisions,
// same values as testTriangle (the argument above); a lane whose determinant is too small computes a garbage u that
// `ok` masks.  Returns nothing: updates (bestT, bestIdx, bestDet) like testTriangle.
// Taken when TraceParams::seqUnitUFirst says the scene is of that kind (host/precompute.h unitUSkipFraction >= 0.4:
// ce two masters +10 %, one master +4 %; suzanne, where 17-30 % of the units skip, -1 %; random soups -1 ... -4 %:
// profiles/r06aa_*); the other scenes run the fused test.
// V_BALLOT: a second ballot after v - on ce 97 % of the (ray, unit) pairs have no lane past u AND v - skips t for the
// unit as well (the resident slots: ce two masters +4.5 %, one master +5 %; not the streamed tail, where it measured
// +1 / -3 %: profiles/r06ag_*).
template <bool V_BALLOT>
__device__ __forceinline__ void testTriangleUnit(d3 o, d3 d, d3 v0, d3 e1, d3 e2, uint32_t idx,
                                                 double &bestT, uint32_t &bestIdx, double &bestDet) {
  const d3 pVec = cross(d, e2);
  const double det = dot(e1, pVec);
  const bool ok = !(__builtin_fabs(det) < kEpsilon);
  const double invDet = rcp(det);
  const d3 tVec = o - v0;
  const double u = dot(tVec, pVec) * invDet;
  const bool pu = ok & !((u < 0.0) | (u > 1.0));
  if (__builtin_amdgcn_ballot_w64(pu) == 0) return; // wave-uniform: nobody in this unit gets past u
  if constexpr (V_BALLOT) {
    d3 qVec = mk(0, 0, 0);
    bool puv = false;
    if (pu) {
      qVec = cross(tVec, e1);
      const double v = dot(d, qVec) * invDet;
      puv = !((v < 0.0) | (u + v > 1));
    }
    if (__builtin_amdgcn_ballot_w64(puv) == 0) return; // ... nor past v
    if (puv) {
      const double t = dot(e2, qVec) * invDet;
      if (t > kEpsilon && t < bestT) {
        bestT = t;
        bestIdx = idx;
        bestDet = det;
      }
    }
  } else if (pu) {
    const d3 qVec = cross(tVec, e1);
    const double v = dot(d, qVec) * invDet;
    if ((v < 0.0) | (u + v > 1)) return;
    const double t = dot(e2, qVec) * invDet;
    if (t > kEpsilon && t < bestT) {
      bestT = t;
      bestIdx = idx;
      bestDet = det;
    }
  }
}

// Two fp32 values in one 64-bit register pair: the operand type of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// (the conservative fp32 prefilter of PTW_ACCEL_PREFILTER: host/prefilter.h).
typedef float Float2 __attribute__((ext_vector_type(2)));

// The prefilter's look at TWO triangles (the halves of every operand) against one ray: r < 0 in a half = that
// triangle is rejected for certain.  U = tVec . pVec, V = d . qVec, D = e1 . pVec, W = D - U - V in fp32;
// r = max(min(U, V, W) + E, E - max(U, V, W)) with E = ea + |o|_inf * eb: negative exactly when two of U, V, W
// certainly have opposite signs - then one of them certainly has the opposite sign of D and the reference's test
// (u < 0 | v < 0 | u + v > 1, src/dod/Scene.cpp:89) rejects whatever D's sign.  A NaN r (a ray with a NaN in it)
// compares false: kept.
struct PrefilterRay {
  Float2 ox, oy, oz, dx, dy, dz, oInf;
};
__device__ __forceinline__ PrefilterRay prefilterRay(d3 o, d3 d) {
  auto splat = [](double x) { const float f = static_cast<float>(x); return (Float2){f, f}; };
  PrefilterRay r;
  r.ox = splat(o.x), r.oy = splat(o.y), r.oz = splat(o.z);
  r.dx = splat(d.x), r.dy = splat(d.y), r.dz = splat(d.z);
  // |o|_inf, not below the true value after the conversion (the bound's coefficient is scaled by it)
  const float oMax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.ox.x), __builtin_fabsf(r.oy.x)), __builtin_fabsf(r.oz.x)) *
                     (1.0f + 0x1p-22f);
  r.oInf = (Float2){oMax, oMax};
  return r;
}
__device__ __forceinline__ Float2 prefilterPair(const PrefilterRay &r, Float2 v0x, Float2 v0y, Float2 v0z, Float2 e1x, Float2 e1y,
                                                Float2 e1z, Float2 e2x, Float2 e2y, Float2 e2z, Float2 ea, Float2 eb) {
  auto fma2 = [](Float2 a, Float2 b, Float2 c) { return __builtin_elementwise_fma(a, b, c); };
  const Float2 tx = r.ox - v0x, ty = r.oy - v0y, tz = r.oz - v0z;
  const Float2 px = fma2(r.dy, e2z, -(r.dz * e2y)); // pVec = d x e2
  const Float2 py = fma2(r.dz, e2x, -(r.dx * e2z));
  const Float2 pz = fma2(r.dx, e2y, -(r.dy * e2x));
  const Float2 det = fma2(e1z, pz, fma2(e1y, py, e1x * px));
  const Float2 uN = fma2(tz, pz, fma2(ty, py, tx * px));
  const Float2 qx = fma2(ty, e1z, -(tz * e1y)); // qVec = tVec x e1
  const Float2 qy = fma2(tz, e1x, -(tx * e1z));
  const Float2 qz = fma2(tx, e1y, -(ty * e1x));
  const Float2 vN = fma2(r.dz, qz, fma2(r.dy, qy, r.dx * qx));
  const Float2 wN = det - uN - vN;
  const Float2 E = fma2(r.oInf, eb, ea);
  const Float2 mn = (Float2){__builtin_fminf(__builtin_fminf(uN.x, vN.x), wN.x), __builtin_fminf(__builtin_fminf(uN.y, vN.y), wN.y)};
  const Float2 mx = (Float2){__builtin_fmaxf(__builtin_fmaxf(uN.x, vN.x), wN.x), __builtin_fmaxf(__builtin_fmaxf(uN.y, vN.y), wN.y)};
  const Float2 lo = mn + E, hi = E - mx;
  return (Float2){__builtin_fmaxf(lo.x, hi.x), __builtin_fmaxf(lo.y, hi.y)};
}

// One sphere test, Scene.cpp:17-35.
__device__ __forceinline__ void testSphere(d3 o, d3 d, d3 centre, double radiusSquared,
                                           uint32_t idx, double &bestT, uint32_t &bestIdx) {
  const d3 op = centre - o;
  const double b = dot(op, d);
  double determinant = b * b - dot(op, op) + radiusSquared;
  if (determinant < 0) return;
  determinant = sqrtPos(determinant);
  const double minusT = b - determinant;
  const double plusT = b + determinant;
  if (minusT < kEpsilon && plusT < kEpsilon) return;
  const double t = minusT > kEpsilon ? minusT : plusT;
  if (t < bestT) {
    bestT = t;
    bestIdx = idx;
  }
}


// Camera::rayFromUnit / randomRay, src/math/Camera.h:20-37,54-60.  r0..r3 are canonical
// draws in stream order (r2, r3 unused for a pinhole camera).
template <bool SC = false> // (SC: polynomial constants in scalar registers, ptw_device.h sconst())
__device__ __forceinline__ void cameraRay(const ptw_camera &c, int px, int py, double r0,
                                          double r1, double r2, double r3, d3 &o, d3 &d) {
  const double x0 = (px + r0) * c.reciprocal_width;
  const double y0 = (py + r1) * c.reciprocal_height;
  const double x = 2 * x0 - 1, y = 2 * y0 - 1;
  const d3 ax = ld3(c.axis_x), ay = ld3(c.axis_y), az = ld3(c.axis_z), centre = ld3(c.centre);
  const d3 xContrib = (ax * -x) * c.aspect_ratio;
  const d3 yContrib = ay * -y;
  const d3 zContrib = az * c.camera_plane_dist;
  const d3 direction = normalised((xContrib + yContrib) + zContrib);
  if (c.aperture_radius == 0) {
    o = centre;
    d = direction;
    return;
  }
  const d3 focalPoint = centre + direction * c.focal_distance;
  const double angle = r2 * (2 * kPi - 0) + 0;      // uniform_real_distribution(0, 2*pi)
  const double radius = r3 * (c.aperture_radius - 0) + 0;
  double sn, cs;
  sinCos<true, SC>(angle, sn, cs); // r2 in [0, 1)
  const d3 origin = (centre + (ax * cs) * radius) + (ay * sn) * radius;
  o = origin;
  d = normalised(focalPoint - origin); // Ray::fromTwoPoints, Ray.h:12-15
}


// SPEC (traceSequentialSpec): the generator output lives in a two-block ring in LDS, each block
// with a few entries of overlap copied from its successor, so a group of consecutive draws never
// straddles a regeneration and this context never regenerates: it only reads at (ringOff, pos).
constexpr unsigned kRingStride = 16384;     // bytes between the two ring slots (XOR toggles)
constexpr unsigned kRingHemiOff = 2560;     // hemi table inside a slot, after 316 canon doubles
constexpr int kRingCanonDoubles = 316;      // 312 + 4 entries of the next block

// The twist of all 624 state words by one wave (see the comment above).
__device__ __forceinline__ void mtTwistWave(uint32_t *x, int lane) {
  waveSync();
  for (int base = 0; base < 227; base += 64) { // k in [0, 227): far = old x[k + 397]
    const int k = base + lane;
    uint32_t nv = 0;
    if (k < 227) nv = mtTwist(x[k], x[k + 1], x[k + 397]);
    waveSync();
    if (k < 227) x[k] = nv;
    waveSync();
  }
  for (int base = 227; base < 623; base += 64) { // k in [227, 623): far = new x[k - 227]
    const int k = base + lane;
    uint32_t nv = 0;
    if (k < 623) nv = mtTwist(x[k], x[k + 1], x[k - 227]);
    waveSync();
    if (k < 623) x[k] = nv;
    waveSync();
  }
  if (lane == 0) x[623] = mtTwist(x[623], x[0], x[396]);
  waveSync();
}


// One entry of the draw-derived hemisphere table (see SeqShared::hemi).
__device__ __forceinline__ void hemiEntry(double u, double v, double *out) {
  const double theta = (2 * kPi) * u;
  const double radius = sqrtPos(v);
  double sn, cs;
  sinCos<true>(theta, sn, cs);
  out[0] = cs * radius;
  out[1] = sn * radius;
  out[2] = sqrtPos(1 - v);
}

// SPEC ring: generates the next block of the stream into the slot at `slotOff` (one wave).  The
// first entries of the new block are also the overlap of the block in the other slot, whose last
// hemi entry becomes computable with them.
__device__ __noinline__ void specGenerateBlock(uint32_t *x, char *ring, unsigned slotOff, int lane) {
  mtTwistWave(x, lane);
  double *canon = reinterpret_cast<double *>(ring + slotOff);
  double *hemi = reinterpret_cast<double *>(ring + slotOff + kRingHemiOff);
  double *otherCanon = reinterpret_cast<double *>(ring + (slotOff ^ kRingStride));
  double *otherHemi = reinterpret_cast<double *>(ring + (slotOff ^ kRingStride) + kRingHemiOff);
  for (int i = lane; i < kMtDoubles; i += 64)
    canon[i] = canonicalFromWords(mtTemper(x[2 * i]), mtTemper(x[2 * i + 1]));
  waveSync();
  if (lane < kRingCanonDoubles - kMtDoubles) otherCanon[kMtDoubles + lane] = canon[lane];
  waveSync();
  for (int q = lane; q + 1 < kMtDoubles; q += 64) hemiEntry(canon[q], canon[q + 1], hemi + 3 * q);
  if (lane == 0)
    hemiEntry(otherCanon[kMtDoubles - 1], otherCanon[kMtDoubles], otherHemi + 3 * (kMtDoubles - 1));
  waveSync();
}


// Commands of the tracing waves to the generator wave (one word per barrier parity).
constexpr uint32_t kGenNone = 0, kGenSlot0 = 1, kGenSlot1 = 2, kGenExit = 3;

} // namespace
} // namespace ptw

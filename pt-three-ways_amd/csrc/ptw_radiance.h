// ptw_radiance.h - the radiance recursion of src/dod/Scene.cpp:124-179 as an iteration, generic over the
// execution context (SeqCtx: one workgroup per pass; PixCtxT: one lane per sample), and the build switches
// every kernel translation unit shares.  Internal to csrc/ (device code; included by the kernel files only).
#pragma once
#include "ptw_trace_common.h"

// -DPTW_PROFILE_PHASES=1: debug build that times the phases of the sequential kernel with
// s_memtime and printf()s the per-ray averages of pass 0 (never enabled in the shipped library).
#ifndef PTW_PROFILE_PHASES
#define PTW_PROFILE_PHASES 0
#endif
// two-master kernels, large scenes: share of the younger wave of a worker pair in percent of an older
// wave's (seqUnitSplitByPlace; 100 = equal shares, round 3's form)
#ifndef PTW_SEQ_YOUNG_PERCENT
#define PTW_SEQ_YOUNG_PERCENT 70
#endif
#if PTW_PROFILE_PHASES
#define PTW_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PTW_ACC(slot, a, b) prof[slot] += (b) - (a)
#else
#define PTW_T(var)
#define PTW_ACC(slot, a, b)
#endif


namespace ptw {
using namespace ptwd;
namespace {

// Builds the Surface for a hit.  `uniform` callers pass a wave-uniform key so the record
// loads become scalar loads.
__device__ __forceinline__ Surface makeSurface(const TraceParams &p, const TriShade *triShade,
                                               const SphereRec *spheres, const HitKey &k, d3 o,
                                               d3 d) {
  Surface s;
  s.pos = o + d * k.t; // Ray::positionAlong, Ray.h:25-27
  double ior, invIor, reflectivity;
  bool inside;
  if (k.idx >= p.nsph) {
    const TriShade &r = triShade[k.idx - p.nsph];
    const bool backfacing = k.det < kEpsilon; // Scene.cpp:107
    const d3 n = ld3(r.normal), bx = ld3(r.basisX);
    s.normal = backfacing ? -n : n;
    s.basis.x = backfacing ? -bx : bx;
    s.basis.y = ld3(r.basisY);
    s.basis.z = s.normal;
    s.emission = ld3(r.emission);
    s.diffuse = ld3(r.diffuse);
    s.coneAngle = r.coneAngle;
    ior = r.ior, invIor = r.invIor, reflectivity = r.reflectivity;
    inside = backfacing;
  } else {
    const SphereRec &r = spheres[k.idx];
    d3 n = normalised(s.pos - ld3(r.centre)); // Scene.cpp:40-44
    inside = dot(n, d) > 0;
    if (inside) n = -n;
    s.normal = n;
    s.basis = basisFromZ(n);
    s.emission = ld3(r.emission);
    s.diffuse = ld3(r.diffuse);
    s.coneAngle = r.coneAngle;
    ior = r.ior, invIor = r.invIor, reflectivity = r.reflectivity;
  }
  // Scene.cpp:140-146
  s.iorFrom = inside ? ior : 1.0;
  s.iorTo = inside ? 1.0 : ior;
  s.iorRatio = inside ? ior : invIor; // ior / 1.0 == ior ; 1.0 / ior
  s.matReflectivity = reflectivity;
  s.reflectivity = resolveReflectivity(s, d);
  return s;
}

// -----------------------------------------------------------------------------------------
// The radiance recursion of Scene.cpp:124-179 as an iteration, generic over the execution
// context CTX, which supplies:
//   double draw()                          next canonical double of this sample's stream
//   HitKey intersect(d3 o, d3 d)           Scene::intersect (nearest hit, reference tie-break)
//   bool   branch(bool)                    the condition (made wave-uniform where it is)
//   void   push(int level, E, D, refl) / Level top(int level)   the per-depth (E, T) stack
// The recursion L_d = E_d + T_d * L_{d+1} is folded innermost-first, as the reference
// evaluates it, so the rounding sequence is the same.
// -----------------------------------------------------------------------------------------
struct Level {
  d3 emission;
  d3 diffuse;
  bool reflective;
};

template <typename CTX>
__device__ __forceinline__ bool scatter(CTX &ctx, const Surface &s, d3 dirIn, double u, double v,
                                        double pDraw, d3 &dirOut) {
  if (ctx.branch(pDraw < s.reflectivity)) { // Scene.cpp:163-168
    dirOut = coneSample(reflect(s.normal, dirIn), s.coneAngle, u, v);
    return true;
  }
  dirOut = hemisphereSample<CTX::kScalarConsts>(s.basis, u, v); // Scene.cpp:169-175
  return false;
}

// radiance(rng, ray, depth >= 1, ...) for the single-sample levels.
template <typename CTX>
__device__ __forceinline__ d3 radianceChain(CTX &ctx, const TraceParams &p,
                                            const TriShade *triShade, const SphereRec *spheres,
                                            d3 o, d3 d) {
  int nlev = 0;
  d3 L;
  for (int depth = 1;; ++depth) {
    if (depth >= p.maxDepth) { // Scene.cpp:128
      L = mk(0, 0, 0);
      break;
    }
    const HitKey k = ctx.intersect(o, d);
    if (ctx.branch(k.idx == kMiss)) { // Scene.cpp:131-133
      L = ld3(p.env);
      break;
    }
    if (depth + 1 >= p.maxDepth) {
      // Last level: the child is radiance(depth + 1 >= maxDepth) = 0 (Scene.cpp:128), so this
      // level returns E + 0 or E + D * 0 = E whatever the lobe; the new direction is never
      // used.  Only the three draws it consumes matter to the stream.
      const unsigned long long tE0 = ctx.now();
      ctx.skip3();
      L = ctx.emissionAt(k);
      ctx.acc(8, tE0, L.x);
      break;
    }
    const Surface s = ctx.surfaceAt(k, o, d, false);
    // numUSamples == numVSamples == 1: (0 + xi) / 1.0 == xi exactly
    const unsigned long long tS0 = ctx.now();
    d3 nd;
    const bool refl = ctx.scatterChain(s, d, nd);
    ctx.acc(3, tS0, nd.x);
    ctx.push(nlev++, s.emission, s.diffuse, refl, k.idx);
    o = s.pos;
    d = nd;
  }
  // fold: result = 0 + (E + T * child); result / 1 (both exact no-ops on the value)
  const unsigned long long tF0 = ctx.now();
  for (int i = nlev - 1; i >= 0; --i) L = ctx.fold(i, L);
  ctx.acc(7, tF0, L.x);
  return L;
}

// Stratified (u, v) of sub-sample (uS, vS): (double(uSample) + unit(rng)) / double(numUSamples)
// (Scene.cpp:152-159).  A power-of-two divisor is an exact scaling, so multiply by its reciprocal
// (one decision for the usual 4x4 / 2x2 / 1x1 fan-outs); otherwise divide.
__device__ __forceinline__ void stratify(const TraceParams &p, int uS, int vS, double xu, double xv,
                                         double invU, double invV, double &u, double &v) {
  const double ur = static_cast<double>(uS) + xu;
  const double vr = static_cast<double>(vS) + xv;
  if ((p.uPow2 & p.vPow2) != 0) {
    u = ur * invU;
    v = vr * invV;
  } else {
    u = p.uPow2 ? ur * invU : ur / static_cast<double>(p.fbU);
    v = p.vPow2 ? vr * invV : vr / static_cast<double>(p.fbV);
  }
}

// radiance(rng, ray, 0, renderParams): the depth-0 level with its fbU x fbV fan-out.
template <typename CTX>
__device__ __forceinline__ d3 radiance0(CTX &ctx, const TraceParams &p, const TriShade *triShade,
                                        const SphereRec *spheres, d3 o, d3 d) {
  if (p.maxDepth <= 0) return mk(0, 0, 0);
  ctx.markRay(0);
  const HitKey k = ctx.intersect(o, d);
  if (ctx.branch(k.idx == kMiss)) return ld3(p.env);
  const Surface s = ctx.surfaceAt(k, o, d);
  if (p.preview) return s.diffuse; // Scene.cpp:137-138
  d3 result = mk(0, 0, 0);
  // in vector registers for the fan-out loop (as scalars they would be re-read from the spill
  // lanes of the kernel-argument tuple for every sub-sample)
  double invU = p.invU, invV = p.invV;
  asm volatile("" : "+v"(invU), "+v"(invV));
  if constexpr (CTX::kLookAhead) ctx.setLookAheadFrame(s, d, invU, invV);
  for (int uS = 0; uS < p.fbU; ++uS) {
    for (int vS = 0; vS < p.fbV; ++vS) {
      const unsigned long long tB0 = ctx.now();
      d3 nd;
      bool refl;
      bool have = false;
      // (worker-wave kernels: this scatter may have been evaluated already, while the workers
      // were searching the previous sub-sample's last ray - see SeqCtx::lookAhead)
      if constexpr (CTX::kLookAhead) have = ctx.takeLookAhead(nd, refl);
      if (!have) {
        double xu, xv, pd;
        ctx.draw3(xu, xv, pd);
        double u, v;
        stratify(p, uS, vS, xu, xv, invU, invV, u, v);
        refl = scatter(ctx, s, d, u, v, pd, nd);
      }
      ctx.acc(6, tB0, nd.x);
      const unsigned long long tA0 = ctx.now();
      if constexpr (CTX::kLookAhead) {
        int nu = uS, nv = vS + 1;
        if (nv == p.fbV) nv = 0, ++nu;
        ctx.armLookAhead(nu < p.fbU, nu, nv);
      }
      ctx.acc(3, tA0, nd.x);
      d3 child;
      if constexpr (CTX::kMasterChain) {
        // (worker-wave masters: the sub-sample whose first ray leaves the scene - most of them in an
        // open scene - costs the pick of the workers' answers, one branch and this sum)
        if (p.maxDepth <= 1) {
          child = mk(0, 0, 0);
        } else {
          ctx.markRay(1);
          const HitKey k1 = ctx.intersect(s.pos, nd);
          child = uniformBool(k1.idx == kMiss) ? ctx.envColour : ctx.chainMasterFrom(p, s.pos, nd, k1);
        }
      } else {
        child = ctx.runChain(p, triShade, spheres, s.pos, nd);
      }
      const unsigned long long tR0 = ctx.now();
      result = result + (refl ? s.emission + child : s.emission + s.diffuse * child);
      ctx.acc(7, tR0, result.x);
    }
  }
  return result * p.invFirstBounce; // Vec3::operator/(double): multiply by 1.0 / (nU * nV)
}

} // namespace
} // namespace ptw

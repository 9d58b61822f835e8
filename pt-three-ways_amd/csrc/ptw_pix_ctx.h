// ptw_pix_ctx.h - PERPIXEL policy: the per-lane execution context and the lock-step sample loop shared by
// tracePerPixel (perpixel.hip) and the accelerated modes (accel.hip).  Internal to csrc/.
#pragma once
#include "ptw_launch.h"
#include "ptw_radiance.h"

namespace ptw {
using namespace ptwd;
namespace {

// -----------------------------------------------------------------------------------------
// PERPIXEL policy: one lane per (pass, pixel) sample; primitives streamed from memory with
// wave-uniform addresses (every lane of a wave tests the same triangle).
// -----------------------------------------------------------------------------------------
struct TriRegs {
  double v[9]; // v0, e1, e2
};
// A triangle's nine doubles through the constant address space: the compiler may then use scalar loads (s_load into
// SGPRs, which every instruction of the test can take as its one scalar operand) even where it
// cannot prove that the kernel never writes the buffer.
typedef const double __attribute__((address_space(4))) ConstDouble;
__device__ __forceinline__ TriRegs loadTriScalar(const double *triGeom, uint32_t k) {
  TriRegs t;
  ConstDouble *g = (ConstDouble *)(triGeom) + 9 * static_cast<size_t>(k);
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = g[i];
  return t;
}

// Device view of the BVH of the accelerated mode (host/bvh.h).
struct BvhNodeDev {
  double lo[2][3], hi[2][3];
  int32_t child[2];
  int32_t count[2];
};
constexpr int kBvhStack = 32;

// How PixCtxT::intersect walks the triangles: every one in fp64 (the reference's brute force); the BVH-culled
// form; every one in fp32 first, two per instruction, and in fp64 only where fp32 cannot prove a rejection.
constexpr int kPixBrute = 0, kPixBvh = 1, kPixPrefilter = 2;

// The fp32 data of two triangles for the prefilter (host/prefilter.h: component-interleaved, then the
// error-bound coefficients), through scalar loads like the fp64 triangles.
typedef const float __attribute__((address_space(4))) ConstFloat;
struct TriPairRegs {
  Float2 v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, ea, eb;
};
__device__ __forceinline__ TriPairRegs loadTriPairScalar(const float *triPacked, uint32_t pair) {
  ConstFloat *g = (ConstFloat *)(triPacked) + 22 * static_cast<size_t>(pair);
  TriPairRegs t;
  t.v0x = (Float2){g[0], g[1]}, t.v0y = (Float2){g[2], g[3]}, t.v0z = (Float2){g[4], g[5]};
  t.e1x = (Float2){g[6], g[7]}, t.e1y = (Float2){g[8], g[9]}, t.e1z = (Float2){g[10], g[11]};
  t.e2x = (Float2){g[12], g[13]}, t.e2y = (Float2){g[14], g[15]}, t.e2z = (Float2){g[16], g[17]};
  t.ea = (Float2){g[18], g[19]}, t.eb = (Float2){g[20], g[21]};
  return t;
}

// Scene::intersectTriangles with the fp32 PREFILTER (host/prefilter.h has the argument): two triangles per
// packed instruction; with U = tVec . pVec, V = d . qVec, D = e1 . pVec and W = D - U - V, a triangle is
// skipped only if min(U, V, W) < -E and max(U, V, W) > E - two of the three certainly have opposite signs, so
// one of them certainly has the opposite sign of D and the fp64 test (u < 0 | v < 0 | u + v > 1, Scene.cpp:89)
// rejects whatever D's sign.  All lanes of a wave look at the same pair; the pair goes to the reference's fp64
// test (all lanes, in index order: the tie-break of the brute-force loop) when ANY lane could not prove its
// rejection (a comparison with a NaN counts as "could not").
__device__ __forceinline__ void prefilteredTriangles(d3 o, d3 d, const float *triPacked, const double *triGeom,
                                                   uint32_t nsph, uint32_t ntri, HitKey &key) {
  const PrefilterRay ray = prefilterRay(o, d);
  const uint32_t npairs = (ntri + 1) >> 1;
  TriPairRegs cur = loadTriPairScalar(triPacked, 0);
  for (uint32_t k = 0; k < npairs; ++k) {
    const TriPairRegs nxt = loadTriPairScalar(triPacked, k + 1 < npairs ? k + 1 : k);
    // (a ray with a NaN or an infinity in it makes r NaN - kept - or leaves the fp64 test without a hit as well;
    // nothing else can produce one: the scene's coordinates and every ray origin are bounded, host/prefilter.h)
    const Float2 r = prefilterPair(ray, cur.v0x, cur.v0y, cur.v0z, cur.e1x, cur.e1y, cur.e1z, cur.e2x, cur.e2y, cur.e2z, cur.ea, cur.eb);
    const bool keepA = !(r.x < 0.0f), keepB = !(r.y < 0.0f);
    if (__builtin_amdgcn_ballot_w64(keepA | keepB) != 0) {
      const uint32_t ia = 2 * k, ib = 2 * k + 1;
      if (__builtin_amdgcn_ballot_w64(keepA) != 0) {
        const TriRegs t = loadTriScalar(triGeom, ia);
        testTriangleUFirst(o, d, mk(t.v[0], t.v[1], t.v[2]), mk(t.v[3], t.v[4], t.v[5]), mk(t.v[6], t.v[7], t.v[8]),
                           nsph + ia, key.t, key.idx, key.det);
      }
      if (ib < ntri && __builtin_amdgcn_ballot_w64(keepB) != 0) {
        const TriRegs t = loadTriScalar(triGeom, ib);
        testTriangleUFirst(o, d, mk(t.v[0], t.v[1], t.v[2]), mk(t.v[3], t.v[4], t.v[5]), mk(t.v[6], t.v[7], t.v[8]),
                           nsph + ib, key.t, key.idx, key.det);
      }
    }
    cur = nxt;
  }
}

template <int MODE>
struct PixCtxT {
  static constexpr bool BVH = MODE == kPixBvh;
  static constexpr bool kLookAhead = false; // (radiance0: a lane never waits for anybody here)
  static constexpr bool kMasterChain = false;
  static constexpr bool kScalarConsts = true; // (four waves per SIMD at 128 registers: ptw_device.h, sconst())
  const TraceParams *p;
  const double *triGeom;
  const TriShade *triShade;
  const SphereRec *spheres;
  // BVH mode only
  const BvhNodeDev *bvhNodes;
  const double *bvhLeafGeom;
  const uint32_t *bvhLeafIndex;
  int32_t *bvhStack; // this lane's slice of the block's LDS traversal stack, stride = blockDim.x
  const float *triPacked; // prefilter mode only: [(ntri + 1) / 2][22] floats (host/prefilter.h)
  __device__ __forceinline__ Surface surfaceAt(const HitKey &k, d3 o, d3 d, bool = true) const {
    return makeSurface(*p, triShade, spheres, k, o, d);
  }
  __device__ __forceinline__ d3 emissionAt(const HitKey &k) const {
    return k.idx >= p->nsph ? ld3(triShade[k.idx - p->nsph].emission) : ld3(spheres[k.idx].emission);
  }
  __device__ __forceinline__ void skip3() {
    for (int i = 0; i < 6; ++i) (void)rng.next();
    words += 6;
  }
  Sfc32 rng;
  unsigned words;
  unsigned long long rays;
  uint32_t *stack; // this lane's slice of the block's LDS stack, stride = blockDim.x

  __device__ __forceinline__ bool branch(bool b) const { return b; }
  __device__ __forceinline__ double draw() {
    const uint32_t w0 = rng.next();
    const uint32_t w1 = rng.next();
    words += 2;
    return canonicalFromWords(w0, w1);
  }
  __device__ __forceinline__ bool scatterChain(const Surface &s, d3 dirIn, d3 &dirOut) {
    double u, v, pd;
    draw3(u, v, pd);
    return scatter(*this, s, dirIn, u, v, pd, dirOut);
  }
  __device__ __forceinline__ void markRay(int) {}
  __device__ __forceinline__ unsigned long long now() const { return 0; }
  __device__ __forceinline__ void acc(int, unsigned long long, double &) {}
  __device__ __forceinline__ void draw3(double &a, double &b, double &c) {
    a = draw();
    b = draw();
    c = draw();
  }
  // The (E, T) stack holds one word per level: the combined primitive index of the hit and the
  // lobe flag; emission and diffuse are re-read from the (cache-resident) records at fold time.
  __device__ __forceinline__ void push(int level, d3, d3, bool refl, uint32_t idx) {
    stack[level * blockDim.x] = idx | (refl ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ Level top(int level) const {
    const uint32_t w = stack[level * blockDim.x];
    const uint32_t idx = w & 0x7fffffffu;
    Level lv;
    lv.reflective = (w >> 31) != 0;
    if (idx >= p->nsph) {
      const TriShade &r = triShade[idx - p->nsph];
      lv.emission = ld3(r.emission);
      lv.diffuse = ld3(r.diffuse);
    } else {
      lv.emission = ld3(spheres[idx].emission);
      lv.diffuse = ld3(spheres[idx].diffuse);
    }
    return lv;
  }
  // one step of the innermost-first fold: L_level = E + T * L_child (Scene.cpp:163-175)
  __device__ __forceinline__ d3 fold(int level, d3 L) const {
    const Level lv = top(level);
    return lv.reflective ? lv.emission + L : lv.emission + lv.diffuse * L;
  }
  __device__ __forceinline__ d3 runChain(const TraceParams &tp, const TriShade *ts, const SphereRec *sp,
                                         d3 o, d3 d) {
    return radianceChain(*this, tp, ts, sp, o, d);
  }

  // One Moller-Trumbore test that keeps the lexicographic minimum of (t, combined index): the
  // BVH visits triangles in its own order, and the reference's scan (strict `<` in insertion
  // order, spheres first) resolves exact ties towards the lowest index.
  __device__ __forceinline__ static void testTriangleLex(d3 o, d3 d, d3 v0, d3 e1, d3 e2, uint32_t idx,
                                                         HitKey &key) {
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
    if (t > kEpsilon && (t < key.t || (t == key.t && idx < key.idx))) {
      key.t = t;
      key.idx = idx;
      key.det = det;
    }
  }

  // Scene::intersect with triangles culled by the BVH: the same tests on fewer triangles, the same
  // nearest hit (see host/bvh.h for why nothing that could win is skipped).
  __device__ __forceinline__ void intersectBvh(d3 o, d3 d, HitKey &key) {
    const uint32_t nsph = p->nsph;
    const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z; // IEEE: +-inf for a zero component
    const int stride = blockDim.x;
    int sp = 0;
    bvhStack[0] = 0;
    sp = 1;
    while (sp > 0) {
      const BvhNodeDev &n = bvhNodes[bvhStack[--sp * stride]];
      double entry[2];
      bool hit[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // slab test; fmin / fmax drop the NaN of 0 * inf (origin on a slab plane of a flat axis)
        const double ax = (n.lo[c][0] - o.x) * ix, bx = (n.hi[c][0] - o.x) * ix;
        const double ay = (n.lo[c][1] - o.y) * iy, by = (n.hi[c][1] - o.y) * iy;
        const double az = (n.lo[c][2] - o.z) * iz, bz = (n.hi[c][2] - o.z) * iz;
        const double tmin = __builtin_fmax(__builtin_fmax(__builtin_fmin(ax, bx), __builtin_fmin(ay, by)),
                                           __builtin_fmin(az, bz));
        const double tmax = __builtin_fmin(__builtin_fmin(__builtin_fmax(ax, bx), __builtin_fmax(ay, by)),
                                           __builtin_fmax(az, bz));
        entry[c] = tmin;
        // `<=`: a box whose entry distance equals the best hit may hold an exact tie
        hit[c] = n.count[c] >= 0 && tmin <= tmax && tmax >= 0.0 && tmin <= key.t;
      }
      // leaves are tested at once, inner children go on the stack (the nearer one on top)
      int push[2], npush = 0;
      const int first = (hit[0] && hit[1] && entry[1] < entry[0]) ? 1 : 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = k == 0 ? first : 1 - first;
        if (!hit[c]) continue;
        if (n.count[c] > 0) {
          if (!(entry[c] <= key.t)) continue; // the other leaf may have shortened the ray
          for (int i = 0; i < n.count[c]; ++i) {
            const uint32_t e = static_cast<uint32_t>(n.child[c] + i);
            const double *g = bvhLeafGeom + 9 * static_cast<size_t>(e);
            testTriangleLex(o, d, ld3(g), ld3(g + 3), ld3(g + 6), nsph + bvhLeafIndex[e], key);
          }
        } else {
          push[npush++] = n.child[c];
        }
      }
      // nearer child last, so that it is popped first
      for (int k = npush - 1; k >= 0; --k)
        if (sp < kBvhStack) bvhStack[sp++ * stride] = push[k];
    }
  }

  __device__ __forceinline__ HitKey intersect(d3 o, d3 d) {
#if PTW_PROFILE_PHASES
    // PTW_PIX_COUNT_SLOTS=1 (prof build): count lane SLOTS instead of rays - 64 per call of this
    // function by a wave, whatever the number of lanes that still hold a ray: rays / slots is the
    // lane occupancy of the lock-step kernel
    if (p->padA) {
      const unsigned long long exec = __builtin_amdgcn_ballot_w64(true);
      if (static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(exec)) rays += 64;
    } else
#endif
    rays++;
    HitKey key;
    key.t = kInf, key.idx = kMiss, key.det = 0;
    const uint32_t nsph = p->nsph, ntri = p->ntri;
    for (uint32_t i = 0; i < nsph; ++i) {
      const SphereRec &r = spheres[i];
      testSphere(o, d, ld3(r.centre), r.radiusSquared, i, key.t, key.idx);
    }
    if (BVH) {
      if (ntri) intersectBvh(o, d, key);
      return key;
    }
    if (MODE == kPixPrefilter) {
      if (ntri) prefilteredTriangles(o, d, triPacked, triGeom, nsph, ntri, key);
      return key;
    }
    // Scalar loads, one triangle ahead: the nine doubles of a triangle arrive in SGPRs, and every
    // instruction of the test takes at most one of them - no vector loads, no copies, nothing for
    // the other waves of the SIMD to hide (measured against vector loads issued per iteration:
    // +15 % on Cornell, profiles/r02q_perpixel_scalar_loads_probe.txt).
    if (!ntri) return key;
    TriRegs cur = loadTriScalar(triGeom, 0);
    for (uint32_t k = 0; k < ntri; ++k) {
      const TriRegs nxt = loadTriScalar(triGeom, k + 1 < ntri ? k + 1 : k);
      testTriangleUFirst(o, d, mk(cur.v[0], cur.v[1], cur.v[2]), mk(cur.v[3], cur.v[4], cur.v[5]),
                         mk(cur.v[6], cur.v[7], cur.v[8]), nsph + k, key.t, key.idx, key.det);
      cur = nxt;
    }
    return key;
  }
};

using PixCtx = PixCtxT<kPixBrute>;

constexpr int kPixBlock = 256;
// resident waves per SIMD the lock-step PERPIXEL kernel is compiled for (A/B: -DPTW_PIX_WAVES=n)
#ifndef PTW_PIX_WAVES
#define PTW_PIX_WAVES 4
#endif

// Lock-step kernel.  With the first-bounce surface carried through the fan-out it needs 174 VGPRs
// (two waves per SIMD); measured in that form with the grid-stride loop on Cornell 1024 x 1024 @ 256
// (profiles/r03i_lockstep_waves_per_simd.txt): 2 waves 180, 3 waves (168 VGPRs, 5 spilled) 224,
// **4 waves (128 VGPRs, 80 spilled) 243**, 5 waves 244, 6 waves (80 VGPRs, 152 spilled) 246
// Msamples/s - occupancy buys more than spills cost, and flattens out at four.  The shipped form
// rebuilds the surface per sub-sample (radiance0Pix below) and spills 10 registers at four waves.
// (Round 2 measured 3 = 4 on the one-sample-per-lane form of this kernel.)
// radiance0() for the lock-step PERPIXEL kernel with the first-bounce surface REBUILT for every
// sub-sample instead of carried through the fan-out: a Surface is 27 doubles, live across sixteen
// chains, and the kernel runs at four waves per SIMD (128 registers).  What survives a chain is the
// hit (distance, index, determinant) and the primary ray; the surface is re-derived from the tables
// before the scatter and its two colours are re-read after the chain - the same loads and the same
// arithmetic on the same inputs, so the same values.  (The empty asm statements keep the compiler
// from hoisting the rebuild out of the loop, which would bring the 54 registers back.)
// Measured against the carried surface (round 3, profiles/r03j_lockstep_rebuild_surface_ab.txt; that
// form left the tree in round 5, last revision 916a1dc): 10 spilled registers instead of 80, 67 instead of
// 548 B of HBM traffic per sample (24 are the algorithmic ones), Cornell 240.5 against 243.0,
// suzanne 21.0 against 21.6, single-sphere 313 against 304 Msamples/s.
template <int MODE>
__device__ __forceinline__ d3 radiance0Pix(PixCtxT<MODE> &ctx, const TraceParams &p, const TriShade *triShade,
                                           const SphereRec *spheres, d3 o, d3 d) {
  if (p.maxDepth <= 0) return mk(0, 0, 0);
  HitKey k = ctx.intersect(o, d);
  if (k.idx == kMiss) return ld3(p.env);
  if (p.preview) return makeSurface(p, triShade, spheres, k, o, d).diffuse; // Scene.cpp:137-138
  d3 result = mk(0, 0, 0);
  for (int uS = 0; uS < p.fbU; ++uS) {
    for (int vS = 0; vS < p.fbV; ++vS) {
      d3 nd, from;
      bool refl;
      {
        asm volatile("" : "+v"(k.t));
        const Surface s = makeSurface(p, triShade, spheres, k, o, d);
        double xu, xv, pd;
        ctx.draw3(xu, xv, pd);
        double u, v;
        stratify(p, uS, vS, xu, xv, p.invU, p.invV, u, v);
        refl = scatter(ctx, s, d, u, v, pd, nd);
        from = s.pos;
      }
      const d3 child = ctx.runChain(p, triShade, spheres, from, nd);
      asm volatile("" : "+v"(k.idx));
      const double *m = k.idx >= p.nsph ? triShade[k.idx - p.nsph].emission : spheres[k.idx].emission;
      const double *df = k.idx >= p.nsph ? triShade[k.idx - p.nsph].diffuse : spheres[k.idx].diffuse;
      const d3 emission = ld3(m), diffuse = ld3(df);
      result = result + (refl ? emission + child : emission + diffuse * child);
    }
  }
  return result * p.invFirstBounce; // Vec3::operator/(double): multiply by 1.0 / (nU * nV)
}

template <int MODE>
__device__ __forceinline__ void perPixelSample(const TraceParams &p, const TraceBuffers &b, uint32_t *ldsWords) {
  const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
  // intersect() calls of all of this lane's samples: ONE atomic per wave at the end (the address is
  // wave-uniform, so the compiler reduces the 64 lanes first).  Round 3 added each sample's count to
  // its pass's counter - a 32-byte memory request per sample, 2.8x the path's algorithmic HBM
  // traffic; only the sum over the passes is ever read (ptw_context_get_stats).
  unsigned long long laneRays = 0;
  // grid-stride: a lane traces sample gid, gid + grid, ... one after another (launchTracePerPixel
  // sizes the grid for kPixSamplesPerLane samples per lane; 1 = one sample per lane per launch)
  for (uint64_t gid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; gid < total;
       gid += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
  // consecutive lanes = consecutive pixels of one pass (coalesced stage writes)
  const uint32_t pass = static_cast<uint32_t>(gid / p.pixCount);
  const uint32_t i = static_cast<uint32_t>(gid % p.pixCount);
  const uint32_t pix = globalPixel(p, p.pixBegin + i);

  PixCtxT<MODE> ctx;
  ctx.p = &p;
  ctx.triGeom = b.triGeom;
  ctx.triShade = b.triShade;
  ctx.spheres = b.spheres;
  ctx.bvhNodes = reinterpret_cast<const BvhNodeDev *>(b.bvhNodes);
  ctx.bvhLeafGeom = b.bvhLeafGeom;
  ctx.bvhLeafIndex = b.bvhLeafIndex;
  ctx.triPacked = b.triPacked;
  const int levels = p.maxDepth > 1 ? p.maxDepth - 1 : 1;
  ctx.bvhStack = reinterpret_cast<int32_t *>(ldsWords + static_cast<size_t>(levels) * blockDim.x) + threadIdx.x;
  ctx.words = 0;
  ctx.rays = 0;
  ctx.stack = ldsWords + threadIdx.x;
  ctx.rng.seed(p.passSeedBase + pass, pix);

  const int px = static_cast<int>(pix % static_cast<uint32_t>(p.width));
  const int py = static_cast<int>(pix / static_cast<uint32_t>(p.width));
  const double r0 = ctx.draw();
  const double r1 = ctx.draw();
  double r2 = 0, r3 = 0;
  if (p.cam.aperture_radius != 0) {
    r2 = ctx.draw();
    r3 = ctx.draw();
  }
  d3 o, d;
  cameraRay<true>(p.cam, px, py, r0, r1, r2, r3, o, d);
  const d3 L = radiance0Pix(ctx, p, b.triShade, b.spheres, o, d);
  double *out = b.stage + (static_cast<size_t>(pass) * p.pixCount + i) * 3;
  out[0] = L.x, out[1] = L.y, out[2] = L.z;
  if (b.words) b.words[static_cast<size_t>(pass) * p.npix + pix] = ctx.words;
  laneRays += ctx.rays;
  }
  if (b.rays) atomicAdd(&b.rays[0], laneRays);
}

} // namespace
} // namespace ptw

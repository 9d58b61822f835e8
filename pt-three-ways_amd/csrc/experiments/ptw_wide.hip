// ptw_wide.hip — traceSequentialWide: the SEQUENTIAL RNG policy (bit-compatible with the
// reference's per-pass std::mt19937 stream, src/dod/Scene.cpp:208-217) for scenes of at most
// G x SLOTS triangles, with the first-bounce fan-out traced speculatively by MANY candidates per
// round, G lanes per candidate.
//
// Why.  Within a pass everything is serial: sub-sample j+1 of a pixel's 4x4 first-bounce fan-out
// starts in the stream where sub-sample j stopped, and j consumes 3 draws per level it reaches -
// known only when it is done.  traceSequentialSpec (ptw_kernels.hip) speculates with four waves,
// one sub-path each: every wave spends 64 lanes on one ray (a lane per primitive for the
// nearest-hit search, then 64 lanes computing the same shading), and commits 2.0 sub-samples per
// round on Cornell.  The chip is then issue-bound on redundant work: at 256 passes every SIMD
// carries one wave that issues ~65 % of the time for 1/64 of a wave's worth of shading.
//
// Here a wave carries 64 / G candidates (G = 8: eight), a workgroup of eight waves - two per SIMD,
// so that one wave's scalar work, branches and LDS waits overlap the other's arithmetic - 64.  A
// candidate
// is a pair (m, D): "sub-sample j + m of the current pixel, starting D draws after the stream
// frontier".  The G lanes of a candidate split the scene between them for the nearest-hit search
// (lane s owns primitives s, s + G, s + 2G, ... in registers), combine their partial results
// with a DPP butterfly inside the group (quad_perm, row_half_mirror: no LDS, no readlane) and then
// all run the same shading for their candidate - so the shading of eight sub-paths costs what one
// cost before.  The candidate set is the prefix-closed set of (m, D) nodes that maximises the
// expected number of sub-samples a round commits for the distribution of per-sub-sample draw
// counts (built on the device, per band, from the histogram the previous band measured: see
// wideBuildCandidates): with 32 candidates a Cornell round commits ~4.1 sub-samples, with 64 ~5.7,
// instead of 2.0 (scripts/sim/spec_sim2.py, fed with the oracle's real count sequences).
//
// After a round every wave walks the chain of committed candidates ((0, 0) -> (1, c0) ->
// (2, c0 + c1) ...) through a successor table (two readlanes per step), wave 0 adds the committed
// contributions
// in sub-sample order: the value is the one the serial evaluation defines, bit for bit (the GPU
// tests compare the .raw bytes of all sequential kernel variants); wrong guesses cost energy, not
// correctness, and the ray counter counts committed sub-samples only.
//
// The stream ring (two 312-draw blocks in LDS, each with four entries of overlap and the
// draw-derived hemisphere table) and its parking format between bands are those of
// traceSequentialSpec, so a render may switch between the two kernels from band to band.  There is
// no generator wave here: when the frontier crosses into the other block all waves regenerate the
// block left behind together (the mt19937 twist in three data-parallel phases, then one entry of
// the tables per lane).
#include "ptw_trace_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

// -DPTW_PROFILE_PHASES=1: debug build that times the phases of a round with s_memtime and
// printf()s the per-sample averages of pass 0 (never enabled in the shipped library).
#ifndef PTW_PROFILE_PHASES
#define PTW_PROFILE_PHASES 0
#endif

namespace ptw {

extern __shared__ __attribute__((aligned(64))) unsigned char wideLds[];

namespace {

constexpr int kWideWaves = 8;       // tracing waves: two per SIMD of the CU
constexpr int kWideBlock = 64 * kWideWaves;
constexpr int kWideSphereSlots = 2; // spheres per lane: up to 2 G spheres

// LDS triangle record of this kernel (doubles): what a hit and a fold need, one fetch each.
constexpr int kWtNormal = 0, kWtBasisX = 3, kWtBasisY = 6, kWtThreshold = 9, kWtDiffuse = 10,
              kWtEmission = 13, kWtMaterial = 16, kWtDoubles = 18;

struct alignas(16) WideResult { // one per candidate and round parity, in LDS
  double L[3]; // radiance of the sub-path below the first-bounce surface
  int meta;    // canonical doubles consumed | lobe at the first-bounce surface << 8 | rays << 16
  int pad;
};

// ---- LDS layout (byte offsets into wideLds) ----
constexpr unsigned kOffRing = 0;
constexpr unsigned kOffMt = 2 * kRingStride;
constexpr unsigned kOffResults = kOffMt + kMtWords * sizeof(uint32_t);
// the first-bounce surface of the current pixel: every tracing wave keeps its own copy (it computes
// the same values) so that no synchronisation is needed; rounds re-read what they need
constexpr unsigned kFirstDoubles = 24, kOffFirst = (kOffResults + 2 * kWideMaxCand * sizeof(WideResult) + 63) & ~63u;
constexpr unsigned kOffTables = kOffFirst + kWideWaves * kFirstDoubles * 8;
// layout of that record (doubles)
constexpr unsigned kFsPos = 0, kFsNormal = 3, kFsBasisX = 6, kFsBasisY = 9, kFsEmission = 12, kFsDiffuse = 15,
                   kFsReflectivity = 18, kFsCone = 19, kFsDir = 20;

__host__ __device__ inline size_t wideLdsBytes(uint32_t ntri, uint32_t nmat, uint32_t nsph) {
  size_t n = kOffTables;
  n += static_cast<size_t>(nsph) * sizeof(SphereRec);
  n += static_cast<size_t>(ntri) * kWtDoubles * sizeof(double);
  n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  // more than half of a CU's 160 KB: one workgroup per CU, its eight waves two to a SIMD
  const size_t floor = 84 * 1024;
  return n < floor ? floor : n;
}

__device__ __forceinline__ double ldsD(unsigned byteOff) {
  return *reinterpret_cast<const double *>(wideLds + byteOff);
}
__device__ __forceinline__ d3 ldsD3(unsigned byteOff) {
  return mk(ldsD(byteOff), ldsD(byteOff + 8), ldsD(byteOff + 16));
}

// Minimum of an unsigned word over the G lanes of a group, delivered to all of them: a DPP
// butterfly (each step one v_min_u32 with the partner lane as DPP operand).
template <int G>
__device__ __forceinline__ unsigned groupMinU(unsigned x) {
  static_assert(G == 4 || G == 8 || G == 16, "group sizes with a DPP butterfly");
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(x));
  if (G >= 8)
    asm volatile("v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x));
  if (G >= 16)
    asm volatile("v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x));
  return x;
}

// The nearest hit of a group's ray as every lane of the group sees it.
struct GroupHit {
  double t;      // +inf on a miss
  unsigned code; // combined primitive index << 1 | (determinant < epsilon); 0xffffffff on a miss
};
constexpr unsigned kCodeMiss = 0xffffffffu;

// Moller-Trumbore (Scene.cpp:62-98) without branches: the same operations as testTriangle() on
// every path that can win, rejection by select.  A wave that is alone on its issue port pays
// 25-45 cycles per branch (profiles/README.md); five triangles per lane made that fifteen.
// A degenerate determinant makes invDet inf/NaN: every comparison with a NaN is false, and the
// determinant test itself rejects, as in the reference.
__device__ __forceinline__ void testTriangleSelect(d3 o, d3 d, d3 v0, d3 e1, d3 e2, unsigned code0,
                                                   double &bestT, unsigned &bestCode) {
  const d3 pVec = cross(d, e2);
  const double det = dot(e1, pVec);
  const double invDet = rcp(det);
  const d3 tVec = o - v0;
  const double u = dot(tVec, pVec) * invDet;
  const d3 qVec = cross(tVec, e1);
  const double v = dot(d, qVec) * invDet;
  const double t = dot(e2, qVec) * invDet;
  const bool reject = (__builtin_fabs(det) < kEpsilon) | (u < 0.0) | (u > 1.0) | (v < 0.0) | (u + v > 1);
  const bool take = !reject & (t > kEpsilon) & (t < bestT);
  bestT = take ? t : bestT;
  bestCode = take ? (code0 | (det < kEpsilon ? 1u : 0u)) : bestCode;
}

template <int G, int SLOTS>
struct WideGeom {
  // this lane's share of the scene (lane s of a group owns primitives s, s + G, s + 2 G, ...)
  double v0x[SLOTS], v0y[SLOTS], v0z[SLOTS];
  double e1x[SLOTS], e1y[SLOTS], e1z[SLOTS];
  double e2x[SLOTS], e2y[SLOTS], e2z[SLOTS];
  double scx[kWideSphereSlots], scy[kWideSphereSlots], scz[kWideSphereSlots], sr2[kWideSphereSlots];

  __device__ __forceinline__ void load(const TraceParams &p, const double *triGeom, const SphereRec *spheres,
                                       int sub) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      // branch-free (see SeqCtx::loadPrimitives): an unused slot holds a degenerate triangle
      const uint32_t idx = static_cast<uint32_t>(k * G + sub);
      const bool valid = idx < p.ntri;
      const double *g = triGeom + 9 * static_cast<size_t>(valid ? idx : 0u);
      v0x[k] = valid ? g[0] : 0.0, v0y[k] = valid ? g[1] : 0.0, v0z[k] = valid ? g[2] : 0.0;
      e1x[k] = valid ? g[3] : 0.0, e1y[k] = valid ? g[4] : 0.0, e1z[k] = valid ? g[5] : 0.0;
      e2x[k] = valid ? g[6] : 0.0, e2y[k] = valid ? g[7] : 0.0, e2z[k] = valid ? g[8] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kWideSphereSlots; ++k) {
      const uint32_t idx = static_cast<uint32_t>(k * G + sub);
      const bool valid = idx < p.nsph;
      const SphereRec &r = spheres[valid ? idx : 0u];
      // an unused slot holds a sphere no ray can hit: determinant = b^2 - |op|^2 + r2 < 0 always
      scx[k] = valid ? r.centre[0] : 0.0, scy[k] = valid ? r.centre[1] : 0.0, scz[k] = valid ? r.centre[2] : 0.0;
      sr2[k] = valid ? r.radiusSquared : -1e300;
    }
  }

  // Scene::intersect (Scene.cpp:115-122) for the group's ray: this lane's primitives, then the
  // lexicographic minimum of (t, combined index) over the group - the reference's tie-break
  // (strict `<` while scanning spheres, then triangles, in insertion order).
  __device__ __forceinline__ GroupHit intersect(const TraceParams &p, d3 o, d3 d, int sub) const {
    double bestT = kInf;
    unsigned bestCode = kCodeMiss;
    if (p.nsph != 0) {
      uint32_t idx = kMiss;
      testSphere(o, d, mk(scx[0], scy[0], scz[0]), sr2[0], static_cast<uint32_t>(sub), bestT, idx);
      if (p.nsph > static_cast<uint32_t>(G))
        testSphere(o, d, mk(scx[1], scy[1], scz[1]), sr2[1], static_cast<uint32_t>(G + sub), bestT, idx);
      bestCode = idx == kMiss ? kCodeMiss : idx << 1;
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k)
      testTriangleSelect(o, d, mk(v0x[k], v0y[k], v0z[k]), mk(e1x[k], e1y[k], e1z[k]), mk(e2x[k], e2y[k], e2z[k]),
                         (p.nsph + static_cast<uint32_t>(k * G + sub)) << 1, bestT, bestCode);
    // non-negative doubles order like their bit patterns: high words, then low words among the
    // lanes that hold the minimal high word, then the index among the lanes that tie exactly
    const unsigned hi = static_cast<unsigned>(hi32(bestT)), lo = static_cast<unsigned>(lo32(bestT));
    const unsigned mhi = groupMinU<G>(hi);
    const unsigned mlo = groupMinU<G>(hi == mhi ? lo : 0xffffffffu);
    const unsigned mcode = groupMinU<G>(((hi == mhi) & (lo == mlo)) ? bestCode : kCodeMiss);
    GroupHit h;
    h.t = mk64(static_cast<int>(mlo), static_cast<int>(mhi));
    h.code = mcode;
    return h;
  }
};

// Per-lane stream cursor into the two-block ring (see SeqCtx SPEC mode).
struct Cursor {
  unsigned off; // 0 or kRingStride
  int pos;      // canonical double inside the slot
  __device__ __forceinline__ void set(unsigned fOff, int fQ, int delta) { // delta < kMtDoubles
    const int np = fQ + delta;
    const bool wrap = np >= kMtDoubles;
    off = wrap ? fOff ^ kRingStride : fOff;
    pos = wrap ? np - kMtDoubles : np;
  }
  __device__ __forceinline__ void advance(int n) {
    const int np = pos + n;
    const bool wrap = np >= kMtDoubles;
    pos = wrap ? np - kMtDoubles : np;
    off = wrap ? off ^ kRingStride : off;
  }
  __device__ __forceinline__ unsigned canonAt() const { return kOffRing + off + 8u * static_cast<unsigned>(pos); }
  __device__ __forceinline__ unsigned hemiAt() const {
    return kOffRing + off + kRingHemiOff + 24u * static_cast<unsigned>(pos);
  }
};

// Tables in LDS.
struct WideTables {
  unsigned sph, tri, mat; // byte offsets into wideLds
  __device__ __forceinline__ unsigned triRec(uint32_t k) const { return tri + k * (kWtDoubles * 8u); }
  __device__ __forceinline__ unsigned matRec(uint32_t k) const { return mat + k * (kMatDoubles * 8u); }
  __device__ __forceinline__ unsigned sphRec(uint32_t k) const { return sph + k * static_cast<unsigned>(sizeof(SphereRec)); }
  // where emission (then diffuse, 24 bytes on) of the primitive with combined index idx sits
  __device__ __forceinline__ unsigned emissionOf(uint32_t idx, uint32_t nsph) const {
    return idx >= nsph ? triRec(idx - nsph) + 8 * kWtEmission : sphRec(idx) + kSphEmissionOff;
  }
  __device__ __forceinline__ unsigned diffuseOf(uint32_t idx, uint32_t nsph) const {
    return idx >= nsph ? triRec(idx - nsph) + 8 * kWtDiffuse : sphRec(idx) + kSphDiffuseOff;
  }
  static constexpr unsigned kSphEmissionOff = offsetof(SphereRec, emission), kSphDiffuseOff = offsetof(SphereRec, diffuse);
};
constexpr unsigned kSphEmission = offsetof(SphereRec, emission), kSphDiffuse = offsetof(SphereRec, diffuse),
                   kSphCentre = offsetof(SphereRec, centre), kSphIor = offsetof(SphereRec, ior),
                   kSphInvIor = offsetof(SphereRec, invIor), kSphReflectivity = offsetof(SphereRec, reflectivity),
                   kSphCone = offsetof(SphereRec, coneAngle);

// `p < reflectivity` with the exact shortcuts of SeqCtx::lobeIsReflective.
__device__ __forceinline__ bool wideLobeIsReflective(double matReflectivity, double iorFrom, double iorTo,
                                                     double iorRatio, d3 normal, d3 dirIn, double pd) {
  if (matReflectivity >= 0) return pd < matReflectivity;
  if (iorFrom == 1.0 && iorTo == 1.0) {
    const double cosThetaI = -dot(normal, dirIn);
    if (cosThetaI >= 1e-3 && pd > 0.0) return false;
  }
  return pd < reflectance(normal, dirIn, iorFrom, iorTo, iorRatio);
}

// The scatter of a single-sample level (depth >= 1) at a surface that is not the common
// "triangle, diffuse lobe" case: spheres, reflective lobes, Fresnel evaluation.  Same operations,
// in the same order, as SeqCtx::surfaceAt + scatterChain.
__device__ __noinline__ d3 wideScatterGeneral(const WideTables tab, uint32_t nsph, uint32_t idx, bool backfacing,
                                              d3 pos, d3 dirIn, double u, double v, double pd, d3 local,
                                              bool &reflOut) {
  d3 normal;
  Basis basis;
  double ior, invIor, matReflectivity, coneAngle;
  bool inside;
  if (idx >= nsph) {
    const unsigned r = tab.triRec(idx - nsph);
    const d3 n = ldsD3(r + 8 * kWtNormal), bx = ldsD3(r + 8 * kWtBasisX);
    normal = backfacing ? -n : n;
    basis.x = backfacing ? -bx : bx;
    basis.y = ldsD3(r + 8 * kWtBasisY);
    basis.z = normal;
    const unsigned m = tab.matRec(static_cast<uint32_t>(ldsD(r + 8 * kWtMaterial)));
    ior = ldsD(m + 48), invIor = ldsD(m + 56), matReflectivity = ldsD(m + 64), coneAngle = ldsD(m + 72);
    inside = backfacing;
  } else {
    const unsigned r = tab.sphRec(idx);
    d3 n = normalised(pos - ldsD3(r + kSphCentre)); // Scene.cpp:40-44
    inside = dot(n, dirIn) > 0;
    if (inside) n = -n;
    normal = n;
    basis = basisFromZ(n);
    ior = ldsD(r + kSphIor), invIor = ldsD(r + kSphInvIor), matReflectivity = ldsD(r + kSphReflectivity);
    coneAngle = ldsD(r + kSphCone);
  }
  const double iorFrom = inside ? ior : 1.0, iorTo = inside ? 1.0 : ior, iorRatio = inside ? ior : invIor;
  if (wideLobeIsReflective(matReflectivity, iorFrom, iorTo, iorRatio, normal, dirIn, pd)) { // Scene.cpp:163-168
    reflOut = true;
    return coneSample(reflect(normal, dirIn), coneAngle, u, v);
  }
  reflOut = false;
  return normalisedNearUnit(transform(basis, local)); // Scene.cpp:169-175
}

// Surface at the first-bounce hit (uniform over the workgroup), as SeqCtx::surfaceAt builds it.
struct FirstSurface {
  d3 pos, normal;
  Basis basis;
  d3 emission, diffuse;
  double reflectivity, coneAngle;
};
__device__ __forceinline__ FirstSurface wideFirstSurface(const WideTables tab, uint32_t nsph, const GroupHit &k,
                                                         d3 o, d3 d) {
  FirstSurface s;
  s.pos = o + d * k.t;
  const uint32_t idx = k.code >> 1;
  double ior, invIor, matReflectivity;
  bool inside;
  if (idx >= nsph) {
    const unsigned r = tab.triRec(idx - nsph);
    const bool backfacing = (k.code & 1u) != 0;
    const d3 n = ldsD3(r + 8 * kWtNormal), bx = ldsD3(r + 8 * kWtBasisX);
    s.normal = backfacing ? -n : n;
    s.basis.x = backfacing ? -bx : bx;
    s.basis.y = ldsD3(r + 8 * kWtBasisY);
    s.basis.z = s.normal;
    s.emission = ldsD3(r + 8 * kWtEmission);
    s.diffuse = ldsD3(r + 8 * kWtDiffuse);
    const unsigned m = tab.matRec(static_cast<uint32_t>(ldsD(r + 8 * kWtMaterial)));
    ior = ldsD(m + 48), invIor = ldsD(m + 56), matReflectivity = ldsD(m + 64), s.coneAngle = ldsD(m + 72);
    inside = backfacing;
  } else {
    const unsigned r = tab.sphRec(idx);
    d3 n = normalised(s.pos - ldsD3(r + kSphCentre));
    inside = dot(n, d) > 0;
    if (inside) n = -n;
    s.normal = n;
    s.basis = basisFromZ(n);
    s.emission = ldsD3(r + kSphEmission);
    s.diffuse = ldsD3(r + kSphDiffuse);
    ior = ldsD(r + kSphIor), invIor = ldsD(r + kSphInvIor), matReflectivity = ldsD(r + kSphReflectivity);
    s.coneAngle = ldsD(r + kSphCone);
  }
  const double iorFrom = inside ? ior : 1.0, iorTo = inside ? 1.0 : ior, iorRatio = inside ? ior : invIor;
  s.reflectivity = matReflectivity < 0 ? reflectance(s.normal, d, iorFrom, iorTo, iorRatio) : matReflectivity;
  return s;
}

// The next block of the stream into the ring slot at `slotOff`, by all kWideBlock lanes of the
// workgroup (uniform call).  std::mt19937's twist x[k] = f(x[k], x[k+1], x[k+397 mod 624]) reads
// words at most 227 behind its own position that this regeneration has already rewritten, so
// k in [0,227), [227,454), [454,623) are three data-parallel phases (all reads, barrier, all
// writes, barrier), then x[623].  Then one entry per lane: tempering + generate_canonical, the
// overlap entries of the block in the other slot, the draw-derived hemisphere table.
__device__ __noinline__ void wideGenerateBlock(unsigned slotOff) {
  uint32_t *x = reinterpret_cast<uint32_t *>(wideLds + kOffMt);
  const int tid = threadIdx.x;
  for (int base = 0; base < 623; base += 227) {
    const int k = base + tid;
    const bool mine = tid < 227 && k < 623;
    uint32_t nv = 0;
    if (mine) nv = mtTwist(x[k], x[k + 1], base == 0 ? x[k + 397] : x[k - 227]);
    ldsBarrier();
    if (mine) x[k] = nv;
    ldsBarrier();
  }
  if (tid == 0) x[623] = mtTwist(x[623], x[0], x[396]);
  ldsBarrier();
  double *canon = reinterpret_cast<double *>(wideLds + kOffRing + slotOff);
  double *hemi = reinterpret_cast<double *>(wideLds + kOffRing + slotOff + kRingHemiOff);
  double *otherCanon = reinterpret_cast<double *>(wideLds + kOffRing + (slotOff ^ kRingStride));
  double *otherHemi = reinterpret_cast<double *>(wideLds + kOffRing + (slotOff ^ kRingStride) + kRingHemiOff);
  if (tid < kMtDoubles) {
    const double c = canonicalFromWords(mtTemper(x[2 * tid]), mtTemper(x[2 * tid + 1]));
    canon[tid] = c;
    if (tid < kRingCanonDoubles - kMtDoubles) otherCanon[kMtDoubles + tid] = c;
  }
  ldsBarrier();
  if (tid + 1 < kMtDoubles) hemiEntry(canon[tid], canon[tid + 1], hemi + 3 * tid);
  if (tid == kMtDoubles - 1)
    hemiEntry(otherCanon[kMtDoubles - 1], otherCanon[kMtDoubles], otherHemi + 3 * (kMtDoubles - 1));
  ldsBarrier();
}

template <int G, int SLOTS>
__global__ __launch_bounds__(kWideBlock) void traceSequentialWide(
    const TraceParams p, const WideCandidates *__restrict__ candSet, const double *__restrict__ triGeom,
    const SphereRec *__restrict__ spheres, const double *__restrict__ triCompact,
    const double *__restrict__ matTable, uint32_t *__restrict__ mtState, double *__restrict__ specState,
    double *__restrict__ stage, uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters,
    unsigned long long *__restrict__ countHist) {
  constexpr int kBlock = kWideBlock;
  constexpr int kGroups = 64 / G; // candidates per wave
  char *ring = reinterpret_cast<char *>(wideLds + kOffRing);
  uint32_t *mt = reinterpret_cast<uint32_t *>(wideLds + kOffMt);

  const int pass = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int sub = lane % G;

  // ---- shading tables into LDS: spheres and materials as they are, triangles re-packed ----
  WideTables tab;
  tab.sph = kOffTables;
  tab.tri = tab.sph + p.nsph * static_cast<unsigned>(sizeof(SphereRec));
  tab.mat = tab.tri + p.ntri * (kWtDoubles * 8u);
  {
    double *ls = reinterpret_cast<double *>(wideLds + tab.sph);
    const double *gs = reinterpret_cast<const double *>(spheres);
    for (uint32_t i = threadIdx.x; i < p.nsph * (sizeof(SphereRec) / 8); i += kBlock) ls[i] = gs[i];
    double *lm = reinterpret_cast<double *>(wideLds + tab.mat);
    for (uint32_t i = threadIdx.x; i < p.nmat * kMatDoubles; i += kBlock) lm[i] = matTable[i];
    double *lt = reinterpret_cast<double *>(wideLds + tab.tri);
    for (uint32_t k = threadIdx.x; k < p.ntri; k += kBlock) {
      const double *c = triCompact + static_cast<size_t>(k) * kTriCompactDoubles;
      const double *m = matTable + static_cast<size_t>(static_cast<uint32_t>(c[kTriMaterialIndex])) * kMatDoubles;
      double *r = lt + static_cast<size_t>(k) * kWtDoubles;
      for (int i = 0; i < 9; ++i) r[i] = c[i];
      r[kWtThreshold] = c[kTriLobeThreshold];
      for (int i = 0; i < 3; ++i) r[kWtDiffuse + i] = m[3 + i], r[kWtEmission + i] = m[i];
      r[kWtMaterial] = c[kTriMaterialIndex];
      r[kWtMaterial + 1] = 0;
    }
  }
  WideGeom<G, SLOTS> geom;
  geom.load(p, triGeom, spheres, sub);

  // ---- the stream: resume (or start) this pass's generator ring (format of traceSequentialSpec) ----
  uint32_t *myState = mtState + static_cast<size_t>(pass) * kMtWords;
  double *myPark = specState + static_cast<size_t>(pass) * kSpecStateDoubles;
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) mt[i] = myState[i];
  unsigned fOff = 0;
  int fQ = 0;
  __syncthreads();
  if (p.firstBand) {
    wideGenerateBlock(0);
    wideGenerateBlock(kRingStride); // (completes block 0's overlap and its last hemisphere entry)
  } else {
    for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
      const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
      reinterpret_cast<double *>(ring + slot * kRingStride)[k] = myPark[i];
    }
    fOff = __builtin_amdgcn_readfirstlane(static_cast<int>(myPark[2 * kRingCanonDoubles])) ? kRingStride : 0u;
    fQ = __builtin_amdgcn_readfirstlane(static_cast<int>(myPark[2 * kRingCanonDoubles + 1]));
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kMtDoubles; i += kBlock) {
      const int slot = i / kMtDoubles, q = i - slot * kMtDoubles;
      const double *cn = reinterpret_cast<const double *>(ring + slot * kRingStride);
      hemiEntry(cn[q], cn[q + 1], reinterpret_cast<double *>(ring + slot * kRingStride + kRingHemiOff) + 3 * q);
    }
  }
  __syncthreads();

  // The block after the next is generated when the frontier has crossed into the other slot: the
  // slot it left is free then (every wave is past the round's barrier and reads the ring again
  // only in the next round).
  auto advanceFrontier = [&](int n) { // n < kMtDoubles; uniform
    const int np = fQ + n;
    if (np >= kMtDoubles) {
      const unsigned left = fOff;
      fQ = np - kMtDoubles;
      fOff ^= kRingStride;
      wideGenerateBlock(left);
    } else {
      fQ = np;
    }
  };

  // this lane's candidate (the group it belongs to) and, for the walk after a round, what lane c
  // knows about candidate c
  const int nCand = __builtin_amdgcn_readfirstlane(candSet->count);
  const int myCand = wave * kGroups + lane / G;
  const unsigned myNode = myCand < nCand ? candSet->node[myCand] : 0xffffu;
  const int myM = static_cast<int>(myNode >> 8), myD = static_cast<int>(myNode & 0xff);
  const unsigned succV = lane < nCand ? candSet->succ[lane] : 0xffffffffu;

  const int maxDepth = p.maxDepth;
  const int w = p.width;
  const bool lens = p.cam.aperture_radius != 0;
  const int nSub = p.fbU * p.fbV;
  const int vShift = p.fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(p.fbV)) : 0;
  const bool fastFan = (p.uPow2 & p.vPow2) != 0;
  const int vMask = p.fbV - 1;
  const uint32_t nsph = p.nsph;
  const d3 envColour = ld3(p.env);
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
  unsigned long long raysTotal = 0;
  // committed sub-samples by levels reached (1..4, 5 and more): what the candidate set is built
  // from (static indices only - a register array indexed at run time would live in scratch)
  unsigned h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
  int parity = 0;
#if PTW_PROFILE_PHASES
  unsigned long long stRounds = 0, stCommits = 0, stPrimary = 0, stFirst = 0, stChain = 0, stFold = 0, stWait = 0,
                     stCommit = 0, stLevels = 0, stIdle = 0;
  const unsigned long long stT0 = __builtin_amdgcn_s_memtime();
#define WT(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define WT(var)
#endif

  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    const int px = static_cast<int>(pix % static_cast<uint32_t>(w));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(w));
    // ---- every group: camera ray and first hit at the frontier (redundant, in parallel) ----
    WT(tP0);
    Cursor cur;
    cur.set(fOff, fQ, 0);
    double r0, r1, r2 = 0, r3 = 0;
    {
      const unsigned a = cur.canonAt();
      r0 = ldsD(a), r1 = ldsD(a + 8);
      if (lens) r2 = ldsD(a + 16), r3 = ldsD(a + 24);
    }
    const int camDraws = lens ? 4 : 2;
    d3 o, d;
    cameraRay(p.cam, px, py, r0, r1, r2, r3, o, d);
    int sampleDraws = camDraws;
    d3 L = mk(0, 0, 0);
    bool traced = false;
    GroupHit k0;
    k0.t = kInf, k0.code = kCodeMiss;
    if (maxDepth > 0) {
      k0 = geom.intersect(p, o, d, sub);
      raysTotal++;
      if (uniformBool(k0.code == kCodeMiss)) {
        L = envColour;
      } else {
        traced = true;
      }
    }
    const unsigned fs = kOffFirst + static_cast<unsigned>(wave) * (kFirstDoubles * 8);
    if (traced) {
      const FirstSurface first = wideFirstSurface(tab, nsph, k0, o, d);
      if (lane == 0) {
        double *f = reinterpret_cast<double *>(wideLds + fs);
        f[kFsPos] = first.pos.x, f[kFsPos + 1] = first.pos.y, f[kFsPos + 2] = first.pos.z;
        f[kFsNormal] = first.normal.x, f[kFsNormal + 1] = first.normal.y, f[kFsNormal + 2] = first.normal.z;
        f[kFsBasisX] = first.basis.x.x, f[kFsBasisX + 1] = first.basis.x.y, f[kFsBasisX + 2] = first.basis.x.z;
        f[kFsBasisY] = first.basis.y.x, f[kFsBasisY + 1] = first.basis.y.y, f[kFsBasisY + 2] = first.basis.y.z;
        f[kFsEmission] = first.emission.x, f[kFsEmission + 1] = first.emission.y, f[kFsEmission + 2] = first.emission.z;
        f[kFsDiffuse] = first.diffuse.x, f[kFsDiffuse + 1] = first.diffuse.y, f[kFsDiffuse + 2] = first.diffuse.z;
        f[kFsReflectivity] = first.reflectivity, f[kFsCone] = first.coneAngle;
        f[kFsDir] = d.x, f[kFsDir + 1] = d.y, f[kFsDir + 2] = d.z;
      }
      waveSync();
    }
    advanceFrontier(camDraws); // (may regenerate a block: uniform)
#if PTW_PROFILE_PHASES
    stPrimary += __builtin_amdgcn_s_memtime() - tP0;
#endif
    if (traced && p.preview) {
      L = ldsD3(fs + 8 * kFsDiffuse); // Scene.cpp:137-138
    } else if (traced) {
      d3 result = mk(0, 0, 0);
      int j = 0;
      unsigned pixHist = 0; // 6-bit fields
      while (j < nSub) {
        WT(tR0);
        // ---- this group's candidate: sub-sample j + myM, stream position frontier + myD ----
        const int myIdx = j + myM;
        bool alive = myIdx < nSub; // (an unused group carries node 0xffff: myM = 255)
        cur.set(fOff, fQ, alive ? myD : 0);
        int draws = 0;
        unsigned rays = 0;
        bool reflFirst = false;
        unsigned long long stackBits = 0; // level i in bits [8i, 8i+8): combined index | lobe << 7
        int nlev = 0;
        d3 ro = ldsD3(fs + 8 * kFsPos), rd = mk(0, 0, 1);
        d3 child = mk(0, 0, 0);
        const bool candidate = alive;
        if (alive) {
          // the first-bounce scatter (Scene.cpp:157-175 at depth 0) of sub-sample myIdx
          const unsigned a = cur.canonAt();
          const double xu = ldsD(a), xv = ldsD(a + 8), pd = ldsD(a + 16);
          cur.advance(3);
          draws = 3;
          double u, v;
          if (fastFan) {
            const int uS = myIdx >> vShift, vS = myIdx & vMask;
            u = (static_cast<double>(uS) + xu) * p.invU;
            v = (static_cast<double>(vS) + xv) * p.invV;
          } else {
            const int uS = myIdx / p.fbV, vS = myIdx - uS * p.fbV;
            const double ur = static_cast<double>(uS) + xu, vr = static_cast<double>(vS) + xv;
            u = p.uPow2 ? ur * p.invU : ur / static_cast<double>(p.fbU);
            v = p.vPow2 ? vr * p.invV : vr / static_cast<double>(p.fbV);
          }
          const d3 fn = ldsD3(fs + 8 * kFsNormal);
          if (pd < ldsD(fs + 8 * kFsReflectivity)) { // Scene.cpp:163-168
            rd = coneSample(reflect(fn, ldsD3(fs + 8 * kFsDir)), ldsD(fs + 8 * kFsCone), u, v);
            reflFirst = true;
          } else {
            Basis fb;
            fb.x = ldsD3(fs + 8 * kFsBasisX), fb.y = ldsD3(fs + 8 * kFsBasisY), fb.z = fn;
            rd = hemisphereSample(fb, u, v); // Scene.cpp:169-175
          }
          if (maxDepth <= 1) alive = false; // radiance(depth 1 >= maxDepth) = 0 (Scene.cpp:128)
        }
        // ---- the chain below the first bounce, level-synchronous over the wave's groups ----
#if PTW_PROFILE_PHASES
        asm volatile("" : "+v"(rd.x));
        WT(tR1);
        stIdle += !candidate;
#endif
        for (int level = 1; level < maxDepth; ++level) {
          if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
#if PTW_PROFILE_PHASES
          stLevels++;
#endif
          const GroupHit k = geom.intersect(p, ro, rd, sub);
          if (alive) {
            rays++;
            if (k.code == kCodeMiss) { // Scene.cpp:131-133
              child = envColour;
              alive = false;
            } else if (level + 1 >= maxDepth) {
              // last level: the child is radiance(depth >= maxDepth) = 0, so this level is its
              // emission whatever the lobe; only the three draws it consumes matter
              cur.advance(3);
              draws += 3;
              child = ldsD3(tab.emissionOf(k.code >> 1, nsph));
              alive = false;
            } else {
              const uint32_t idx = k.code >> 1;
              const bool backfacing = (k.code & 1u) != 0;
              const unsigned ca = cur.canonAt(), ha = cur.hemiAt();
              // everything the common case needs, requested together and waited for once
              const bool isTri = idx >= nsph;
              const unsigned r = tab.triRec(isTri ? idx - nsph : 0u);
              double pd = ldsD(ca + 16), thr = ldsD(r + 8 * kWtThreshold);
              d3 local = ldsD3(ha), n = ldsD3(r + 8 * kWtNormal);
              Basis b;
              b.x = ldsD3(r + 8 * kWtBasisX), b.y = ldsD3(r + 8 * kWtBasisY);
              asm volatile("" : "+v"(pd), "+v"(thr), "+v"(local.x), "+v"(n.x), "+v"(b.x.x), "+v"(b.y.x));
              b.z = n;
              const d3 pos = ro + rd * k.t;
              const double ndotd = dot(n, rd);
              const double cosThetaI = backfacing ? ndotd : -ndotd;
              // the lobe threshold of ptw_layout.h: diffuse unless p < threshold, and for Fresnel
              // surfaces with ior = 1 unless the ray grazes or p == 0
              const bool common = isTri & !(pd < thr) & ((thr >= 0.0) | ((cosThetaI >= 1e-3) & (pd > 0.0)));
              const double sgn = backfacing ? -1.0 : 1.0;
              d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
              bool refl = false;
              if (!common)
                nd = wideScatterGeneral(tab, nsph, idx, backfacing, pos, rd, ldsD(ca), ldsD(ca + 8), pd, local, refl);
              cur.advance(3);
              draws += 3;
              stackBits |= static_cast<unsigned long long>((idx & 0x7fu) | (refl ? 0x80u : 0u)) << (8 * nlev);
              ++nlev;
              ro = pos;
              rd = nd;
            }
          }
        }
        // fold innermost-first: L_level = E + T * L_child (Scene.cpp:163-175)
#if PTW_PROFILE_PHASES
        asm volatile("" : "+v"(child.x));
        WT(tR2);
#endif
        for (int lv = maxDepth - 3; lv >= 0; --lv) {
          if (__builtin_amdgcn_ballot_w64(lv < nlev) == 0) continue;
          if (lv < nlev) {
            const unsigned wd = static_cast<unsigned>(stackBits >> (8 * lv)) & 0xffu;
            const uint32_t idx = wd & 0x7fu;
            const d3 e = ldsD3(tab.emissionOf(idx, nsph)), df = ldsD3(tab.diffuseOf(idx, nsph));
            child = (wd & 0x80u) ? e + child : e + df * child;
          }
        }
        // ---- publish ----
        const unsigned resBase = kOffResults + static_cast<unsigned>(parity * kWideMaxCand) * sizeof(WideResult);
        if (sub == 0 && myCand < kWideMaxCand) {
          WideResult *slot = reinterpret_cast<WideResult *>(wideLds + resBase) + myCand;
          slot->L[0] = child.x, slot->L[1] = child.y, slot->L[2] = child.z;
          slot->meta = candidate ? (draws | (reflFirst ? 0x100 : 0) | static_cast<int>(rays << 16)) : 0;
        }
        WT(tR3);
        ldsBarrier();
        WT(tR4);
        // ---- commit: walk the chain of candidates that started where their predecessor stopped
        //      (identical in every wave): candidate 0 is the frontier itself; candidate c's
        //      successor for the count it consumed comes from the table of the candidate set ----
        // Lane c prepares, in parallel, what the walk needs to know about candidate c: its draw
        // count, its rays, and the candidate that continues it (63: none in the set).
        const WideResult *res = reinterpret_cast<const WideResult *>(wideLds + resBase);
        const int metaV = lane < nCand ? res[lane].meta : 0;
        int packV;
        {
          const int cnt = metaV & 0xff;
          const int levels = (cnt * 11) >> 5; // cnt / 3 for cnt <= 27
          unsigned next = levels >= 1 && levels <= 5 ? (succV >> (6 * (levels - 1))) & 63u : 63u;
          packV = static_cast<int>(next | (static_cast<unsigned>(metaV & 0xff) << 8) |
                                   (static_cast<unsigned>(metaV >> 16) << 16));
        }
        int m = 0, D = 0;
        unsigned raysRound = 0;
        unsigned long long chain = 0; // committed candidate indices, 6 bits each (at most 10 are kept)
        for (int c = 0; c != 63 && j + m < nSub;) {
          const unsigned wd = static_cast<unsigned>(__builtin_amdgcn_readlane(packV, c));
          chain |= static_cast<unsigned long long>(c) << (6 * m);
          D += static_cast<int>((wd >> 8) & 0xffu);
          raysRound += wd >> 16;
          ++m;
          c = m < 10 ? static_cast<int>(wd & 63u) : 63;
        }
        raysTotal += raysRound;
        if (wave == 0) { // only the wave that stores the sample needs the radiance (and the statistics)
          const d3 myL = lane < nCand ? mk(res[lane].L[0], res[lane].L[1], res[lane].L[2]) : mk(0, 0, 0);
          const d3 fe = ldsD3(fs + 8 * kFsEmission), fd = ldsD3(fs + 8 * kFsDiffuse);
          for (int q = 0; q < m; ++q) {
            const int src = static_cast<int>((chain >> (6 * q)) & 63u);
            const int meta = __builtin_amdgcn_readlane(metaV, src);
            const d3 ch = mk(readLane(myL.x, src), readLane(myL.y, src), readLane(myL.z, src));
            result = result + ((meta & 0x100) ? fe + ch : fe + fd * ch);
            const int levels = ((meta & 0xff) * 11) >> 5;
            pixHist += 1u << (6 * ((levels < 5 ? levels : 5) - 1));
          }
        }
        j += m;
        sampleDraws += D;
        parity ^= 1;
        advanceFrontier(D); // (may regenerate a block: uniform)
#if PTW_PROFILE_PHASES
        stRounds++, stCommits += m;
        stFirst += tR1 - tR0, stChain += tR2 - tR1, stFold += tR3 - tR2, stWait += tR4 - tR3;
        stCommit += __builtin_amdgcn_s_memtime() - tR4;
#endif
      }
      L = result * p.invFirstBounce;
      h1 += pixHist & 63u, h2 += (pixHist >> 6) & 63u, h3 += (pixHist >> 12) & 63u;
      h4 += (pixHist >> 18) & 63u, h5 += (pixHist >> 24) & 63u;
    }
    if (threadIdx.x == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = 2u * static_cast<unsigned>(sampleDraws);
    }
  }

#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const double n = static_cast<double>(p.pixCount);
    printf("WIDE<%d,%d> wave %d: cycles/sample=%.0f rounds/sample=%.2f commits/round=%.2f levels/round=%.2f "
           "idle=%.2f | per sample: primary=%.0f first=%.0f chain=%.0f fold+publish=%.0f wait=%.0f commit=%.0f\n",
           G, SLOTS, wave, (__builtin_amdgcn_s_memtime() - stT0) / n, stRounds / n,
           static_cast<double>(stCommits) / stRounds, static_cast<double>(stLevels) / stRounds,
           static_cast<double>(stIdle) / stRounds, stPrimary / n, stFirst / n, stChain / n, stFold / n,
           stWait / n, stCommit / n);
  }
#endif
  if (threadIdx.x == 0) {
    myPark[2 * kRingCanonDoubles] = fOff ? 1.0 : 0.0;
    myPark[2 * kRingCanonDoubles + 1] = static_cast<double>(fQ);
    if (rayCounters) rayCounters[pass] += raysTotal;
    if (countHist) {
      atomicAdd(&countHist[0], static_cast<unsigned long long>(h1));
      atomicAdd(&countHist[1], static_cast<unsigned long long>(h2));
      atomicAdd(&countHist[2], static_cast<unsigned long long>(h3));
      atomicAdd(&countHist[3], static_cast<unsigned long long>(h4));
      atomicAdd(&countHist[4], static_cast<unsigned long long>(h5));
    }
  }
  // ---- park the stream for the next band ----
  __syncthreads();
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = mt[i];
  for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
    const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
    myPark[i] = reinterpret_cast<const double *>(ring + slot * kRingStride)[k];
  }
}

template <int G, int SLOTS>
hipError_t launchWide(const TraceParams &p, const TraceBuffers &b, hipStream_t stream) {
  auto kernel = traceSequentialWide<G, SLOTS>;
  const size_t lds = wideLdsBytes(p.ntri, p.nmat, p.nsph);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(p.npass), dim3(kWideBlock), lds, stream, p,
                     reinterpret_cast<const WideCandidates *>(b.wideCands), b.triGeom, b.spheres, b.triCompact,
                     b.matTable, b.mtState, b.specState, b.stage, b.words, b.rays, b.countHist);
  return hipGetLastError();
}

} // namespace

bool wideKernelApplies(const TraceParams &p) {
  return p.ntri <= 64u && p.nsph <= static_cast<uint32_t>(8 * kWideSphereSlots) &&
         p.nsph + p.ntri <= 127 && p.maxDepth <= 9 && p.fbU * p.fbV >= 1 &&
         wideLdsBytes(p.ntri, p.nmat, p.nsph) <= 150 * 1024;
}


hipError_t launchTraceSequentialWide(const TraceParams &p, const TraceBuffers &b, hipStream_t stream,
                                     const char **variant) {
  // G lanes per candidate: 8 (64 candidates per round) while the lane's share of the triangles fits
  // the register file, 16 (32 candidates) beyond.  PTW_WIDE_G overrides for A/B runs.
  static const char *gEnv = std::getenv("PTW_WIDE_G");
  static const char *nEnv = std::getenv("PTW_WIDE_CANDIDATES");
  int G = p.ntri <= 40 ? 8 : 16;
  if (gEnv && (std::atoi(gEnv) == 8 || std::atoi(gEnv) == 16)) G = std::atoi(gEnv);
  if (G == 8 && p.ntri > 40) G = 16;
  int n = kWideWaves * 64 / G;
  if (nEnv) n = std::max(1, std::min(n, std::atoi(nEnv)));
  // the candidate set for this band, from what the previous band measured
  hipError_t e = launchBuildCandidates(p, b, n, stream);
  if (e != hipSuccess) return e;
  auto pick = [&](const char *name) {
    if (variant) *variant = name;
  };
  if (G == 8) {
    if (p.ntri <= 16) return pick("traceSequentialWide<8,2>"), launchWide<8, 2>(p, b, stream);
    return pick("traceSequentialWide<8,5>"), launchWide<8, 5>(p, b, stream);
  }
  if (p.ntri <= 16) return pick("traceSequentialWide<16,1>"), launchWide<16, 1>(p, b, stream);
  if (p.ntri <= 48) return pick("traceSequentialWide<16,3>"), launchWide<16, 3>(p, b, stream);
  return pick("traceSequentialWide<16,4>"), launchWide<16, 4>(p, b, stream);
}

} // namespace ptw

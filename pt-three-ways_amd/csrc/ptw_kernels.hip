// ptw_kernels.hip — hand-written HIP kernels (gfx950, wave64) for pt-three-ways' DoD radiance
// path: dod::Scene::render -> radiance -> intersect* (src/dod/Scene.cpp).
//
// Kernels (launchTraceSequential / launchTracePerPixel pick the variant):
//
//  traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS>   SEQUENTIAL RNG policy (bit-compatible with
//      the reference's per-pass std::mt19937 stream).  Within a pass the reference consumes one
//      RNG stream across pixels in row-major order with a data-dependent number of draws per
//      pixel (Scene.cpp:211-217), so pixels of one pass are serially dependent.  Parallelism is
//      therefore (a) across passes: ONE WORKGROUP PER PASS, and (b) inside a ray's brute-force
//      nearest-hit search: every lane owns SLOTS triangles, resident in VGPRs for the whole
//      launch (v0, e1, e2 as 9 doubles each; zero memory traffic in the hot loop), tests them
//      against the wave-uniform ray, and the nearest hit is picked with the reference's
//      tie-break (lowest insertion index; spheres before triangles).  WAVES == 1: one wave does
//      everything (scenes up to 128 triangles; REG: the issue-slot-lean variant for up to 64).
//      WAVES == 7: worker waves hold the primitives, a master wave without primitives runs the
//      path logic and exchanges ray / nearest hit with them through LDS.  WAVES == 6, MASTERS == 2
//      (more passes than CUs): two passes per workgroup - two masters, one barrier apart, over six
//      shared workers that search one master's ray while the other master shades.  While its ray
//      is searched a master evaluates the next sub-sample's first-bounce scatter ahead of time
//      (SeqCtx::lookAhead).  Everything after the
//      pick (shading, sampling, RNG) is wave-uniform.  mt19937 lives in LDS: 624 raw words plus
//      the 312 canonical doubles and the draw-derived hemisphere table they yield, regenerated
//      64 lanes at a time, so a draw is one LDS broadcast read.
//
//  traceSequentialSpec          the same policy for scenes up to 64 triangles with the
//      first-bounce fan-out traced speculatively by four waves against a two-block stream ring
//      that a fifth wave keeps filled (the headline kernel; see the comment at the kernel).
//
//  (csrc/experiments/: traceSequentialGang (several CUs per pass) and the decoupled protocol of the
//      two-master kernels - built, bit-identical, measured no faster and
//      therefore NOT in the shipped library; `make experiments` builds them into
//      experiments/libptw_hip.so.  Round 2's traceSequentialWide / traceSequentialSpec8 (DESIGN.md
//      3.1c) were retired from the tree in round 4; their last revision is commit 9d6656f.)
//
//  tracePerPixel, tracePerPixelPersistent   PERPIXEL policy: one lane per (pass, pixel) sample,
//      sfc32 stream per sample, triangles streamed wave-uniformly (scalar loads, SGPR operands);
//      the persistent form (scenes from 128 triangles) hands samples out through a device-wide
//      queue so lanes whose paths end early do not idle.
//
//  tracePerPixelBvh             the SEPARATE accelerated mode (ptw_render_params.accel): the same
//      sample with Scene::intersect culled by a BVH (host/bvh.cpp) - bit-identical results.
//
//  resolveKernel                adds the staged per-pass radiance into the fp64 running sums in
//      pass order (ArrayOutput::operator+=, src/util/ArrayOutput.cpp:48-56) so the accumulation
//      order - and therefore the rounding - is that of `--max-cpus 1`.
//
//  intersectBatchKernel, rngKatKernel   known-answer entry points (ptw_context_intersect /
//      ptw_context_rng_doubles).
#include "ptw_trace_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

// -DPTW_PROFILE_PHASES=1: debug build that times the phases of the sequential kernel with
// s_memtime and printf()s the per-ray averages of pass 0 (never enabled in the shipped library).
#ifndef PTW_PROFILE_PHASES
#define PTW_PROFILE_PHASES 0
#endif
// -DPTW_EXPERIMENTS=1 (make experiments): also build the measured-slower opt-in kernel of
// csrc/experiments/ (traceSequentialGang) and its dispatch.
#ifndef PTW_EXPERIMENTS
#define PTW_EXPERIMENTS 0
#endif
// two-master kernels, large scenes: share of the younger wave of a worker pair in percent of an older
// wave's (seqUnitSplitByPlace; 100 = equal shares, round 3's form)
#ifndef PTW_SEQ_YOUNG_PERCENT
#define PTW_SEQ_YOUNG_PERCENT 70
#endif
// (Round 5 removed the A/B paths whose verdict is recorded in DESIGN.md 3.1 - rejection by select,
// the lane-per-worker pick, radianceChain for the two-master path, the lexicographic pick, the DPP
// reduction in the worker waves, the balance ratios by side, the decoupled two-master protocol and
// its polling variants: last revision with all of them is commit 916a1dc.  Also measured in round 5 and
// not kept: both master waves on ONE SIMD - hardware waves 0 and 4 - with the six workers two to a SIMD
// on the other three: the masters' work per answer did not change (2.62 k against 2.65 k cycles in the
// instrumented build: it is not the worker beside them that makes them slow), suzanne 11.66 against
// 12.04 and ce 2.01 against 2.15 Msamples/s; profiles/r05e_*.)

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

// -----------------------------------------------------------------------------------------
// SEQUENTIAL policy context: one workgroup (WAVES x 64 lanes) per pass.
// -----------------------------------------------------------------------------------------
struct SeqShared {
  uint32_t mt[kMtWords];
  double canon[kMtDoubles];
  // For every position q of the block: the local cosine-hemisphere direction that
  // hemisphereSample() builds from (u, v) = (canon[q], canon[q + 1]) before the basis transform:
  // (cos(2 pi u) sqrt(v), sin(2 pi u) sqrt(v), sqrt(1 - v)).  It depends only on the draws, so all
  // positions are evaluated 64 lanes at a time when the block is generated, instead of one
  // sincos + two square roots on the serial path of every bounce.
  double hemi[kMtDoubles][3];
};

// A worker wave's answer: its nearest hit.  16 bytes, one ds_read_b128 for the master: the distance
// and the combined primitive index with the only fact ever used of the determinant - the sign test
// `det < epsilon` of Scene.cpp:107 - in bit 0 (kMiss, all ones, with t = +inf for "nothing hit").
struct alignas(16) PartialHit {
  double t;
  uint32_t idxSign; // combined index << 1 | (det < epsilon); kMiss (all ones) with t = +inf for "nothing"
  uint32_t pad;
};
__device__ __forceinline__ uint32_t packAnswer(const HitKey &k) {
  return k.idx == kMiss ? kMiss : ((k.idx << 1) | (k.det < kEpsilon ? 1u : 0u));
}
// The nearest of n answers with the reference's tie-break (strictly nearer wins, an exact tie goes to
// the lower combined index: Scene.cpp:31,95,118), computed by every lane alike - no cross-lane traffic:
// the minimum distance, then the lowest packed index among the answers that have it (the packing
// keeps the order of the indices; a miss is +inf / all ones and loses against everything).
template <int N>
__device__ __forceinline__ HitKey pickOfAnswers(const PartialHit (&ph)[N]) {
  HitKey key;
  double bt = ph[0].t;
#pragma unroll
  for (int w = 1; w < N; ++w) bt = vmin64(bt, ph[w].t);
  uint32_t bw = kMiss;
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const uint32_t c = ph[w].t == bt ? ph[w].idxSign : kMiss;
    bw = c < bw ? c : bw;
  }
  key.t = bt;
  key.idx = bw == kMiss ? kMiss : (bw >> 1);
  key.det = (bw & 1u) ? -1.0 : 1.0; // (only its sign test is ever used)
  return key;
}
// Behind the answers: the masters' commands (128 bytes each: two rays + the request word), then 8 bytes
// per worker wave and ray (pickNearest's LDS atomic).
constexpr size_t kSeqCmdBytes = 384;
constexpr size_t kSeqMinSlotOffset = 256; // into the command area; [8 waves][2 rays] x 8 bytes
// PAIR: per master, the first-bounce scatter directions of the next sub-samples at the stream positions
// they may start at - 64 entries, one per lane of the master (SeqCtx::fanBuild) - behind the commands
struct alignas(32) FanEntry {
  double dir[3];
  uint32_t ok; // the entry is usable: sub-sample and draws exist, and the draws choose the diffuse lobe
  uint32_t pad;
};
constexpr int kSeqPixRecDoubles = 24; // ... then the pixel's first-bounce surface (SeqCtx::pixRec) ...
// ... and a copy of the camera: as a kernel argument its 36 dwords sit in scalar registers the master's
// loop has no room for (they were spilled to vector-register lanes and read back per pixel)
constexpr size_t kSeqFanBytes = 64 * sizeof(FanEntry) + kSeqPixRecDoubles * sizeof(double);
// Worker-wave kernels: a copy of the camera in LDS.  As part of the kernel argument its 36 dwords sit in
// scalar registers the master's loop has no room for: they were spilled to vector-register lanes and read
// back for every pixel; from LDS the camera ray reads them with one wait.
constexpr size_t kSeqCamBytes = (sizeof(ptw_camera) + 63) & ~static_cast<size_t>(63);

// Master -> worker request of the multi-wave sequential kernels: one ray, or - two-master kernels
// with pairing - two (the second one belongs to the master's speculated chain, see PairMaster).
constexpr uint32_t kCmdTrace = 1, kCmdExit = 2;
constexpr uint32_t kCmdLive = 0xffffffffu; // two masters: this master still has rays
struct alignas(16) SeqCommand {
  double o[3], d[3];   // ray A
  double o2[3], d2[3]; // ray B (nrays == 2)
  uint32_t op;         // one master: kCmdTrace / kCmdExit; two masters: see workerLoop
  uint32_t nrays;      // (PAIR, experiments build: mask of the slots that hold a ray)
  uint32_t pad[6];     // 128 bytes
};
static_assert(sizeof(SeqCommand) == 128, "SeqCommand layout");

// std::mt19937 regeneration (the "twist") + tempering + generate_canonical for all 312
// doubles, by the 64 lanes of one wave.  Chunks of 64 consecutive k are processed in order;
// inside a chunk every lane reads its inputs, waveSync(), then writes, waveSync() - the
// fences keep the compiler from reordering one lane's loads across another lane's stores.
// Kept out of line: it runs once per 312 draws and would otherwise be cloned into every
// draw() site.
__device__ __forceinline__ void fillHemiTable(SeqShared *sh, int lane) {
  for (int q = lane; q + 1 < kMtDoubles; q += 64) {
    const double u = sh->canon[q], v = sh->canon[q + 1];
    const double theta = (2 * kPi) * u;
    const double radius = sqrtPos(v);
    double sn, cs;
    sinCos<true>(theta, sn, cs);
    sh->hemi[q][0] = cs * radius;
    sh->hemi[q][1] = sn * radius;
    sh->hemi[q][2] = sqrtPos(1 - v);
  }
}

__device__ __noinline__ void mtRegenerateWave(SeqShared *sh, int lane) {
  uint32_t *x = sh->mt;
  mtTwistWave(x, lane);
  for (int i = lane; i < kMtDoubles; i += 64)
    sh->canon[i] = canonicalFromWords(mtTemper(x[2 * i]), mtTemper(x[2 * i + 1]));
  waveSync();
  fillHemiTable(sh, lane);
  waveSync();
}

// Shading tables in LDS (filled once per launch): compact triangle records, the material
// table and the sphere records.  A hit costs one LDS round trip instead of a scalar + vector
// global fetch on the critical path of every ray.
struct SeqTables {
  const double *tri;     // [ntri][kTriCompactDoubles]   (LDS or global)
  const double *mat;     // [nmat][kMatDoubles]           (LDS or global)
  const SphereRec *sph;  // [nsph]                        (LDS or global)
};

// Layout of the per-lane shading record of the REG path (doubles).
constexpr int kRecEmission = 0, kRecDiffuse = 3, kRecDoubles = 6;

template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, bool SPEC = false, int MASTERS = 1, bool PAIR = false,
          bool PICKS = true>
struct SeqCtx {
  // REG (single wave, one triangle per lane, at most 127 primitives, maxDepth <= 9): every lane
  // also keeps the emission and diffuse colour of its triangle in registers, and the (E, T)
  // stack is one byte per level (combined primitive index + lobe flag) in a scalar register
  // pair.  Pushing a level is three scalar instructions; folding one fetches the colours from
  // the owner lane with v_readlane (only the diffuse colour when the emission is zero): no LDS
  // traffic, and with one wave per SIMD nothing would hide an LDS wait.  (Fetching the whole
  // surface record that way was measured and is slower: one ds_read_b128 moves what four
  // v_readlane do.)
  static_assert(!REG || (SLOTS == 1 && WAVES == 1), "REG needs one wave and one triangle per lane");
  double rec[kRecDoubles]; // REG only (never touched otherwise, so it costs nothing there)
  unsigned long long stackBits; // REG: level i in bits [8i, 8i+8): combined index | lobe << 7
  unsigned long long emissiveMask; // REG: lanes whose triangle has a non-zero emission

  // WAVES == 1: one wave does everything.  WAVES > 1: WAVES worker waves hold the primitives
  // and one extra master wave (wave 0, no resident primitives) runs the path logic.
  static constexpr int kThreads = 64 * WAVES;                        // lanes that hold primitives
  // WAVES > 1: while the workers search a ray the master has nothing to do.  Most sub-samples of
  // an open scene end with a ray that leaves it, and then the next sub-sample's first-bounce
  // scatter (one sincos and two square roots on the serial path) starts at the stream position
  // the master is looking at right now: it is evaluated in that idle time, and taken if the
  // position still matches when the next sub-sample starts (lookAhead / takeLookAhead).
  static constexpr bool kLookAhead = WAVES > 1;
  // sincos constants in scalar registers (ptw_device.h, sconst()): the two-master kernels, whose
  // master path is short of vector registers
  static constexpr bool kScalarConsts = MASTERS == 2;
  // two masters per workgroup: radiance0 and the chain use chainMaster / chainMasterFrom
  static constexpr bool kMasterChain = WAVES > 1 && MASTERS == 2;
  // PAIR (two masters): every request carries up to two rays - the next ray of the sub-sample at the
  // stream frontier and the next ray of the sub-sample AFTER it, started at a guessed stream position
  // (pairPixel); the workers test their resident triangles against both.
  static_assert(!PAIR || MASTERS == 2, "paired requests are a form of the two-master kernels");
  // worker waves that share their SIMD with a master wave (the others sit two to a SIMD among themselves)
  static constexpr int kSideB = MASTERS;
  Surface laSurf;  // look-ahead inputs: the first-bounce surface, the incoming direction (set once per
  d3 laDir;        // pixel, before the fan-out: inside its loop they are the caller's own values), ...
  double laInvU, laInvV;
  int laU, laV;    // ... the stratum of the next sub-sample
  bool laArmed;
  int laMisses;    // look-aheads in a row that were not taken (closed scenes: nearly all) ...
  unsigned laTick; // ... after two of them only every eighth sub-sample tries again
  int laPos;       // stream position the result was evaluated for (-1: none)
  d3 laOut;
  bool laRefl;
  // WAVES > 1: the (E, T) stack entry of the level the master has just left is written while the
  // workers search the next ray (flushPending, called between the two barriers of intersect()):
  // its material fetch - two dependent LDS round trips for a triangle - is off the serial path.
  int pendKind;      // 0 none, 1 a triangle hit that took the diffuse lobe (colours still to be fetched)
  int pendLevel;
  uint32_t pendIdx;
  // workgroup size.  MASTERS == 2 (traceSequentialMM): two passes share the worker waves - wave
  // m < 2 runs pass 2 * blockIdx.x + m, and the workers alternate between the two masters' rays,
  // so that one master shades while the other one's ray is being searched.
  static constexpr int kBlock = WAVES == 1 ? 64 : 64 * (WAVES + MASTERS);
  static_assert(MASTERS == 1 || (MASTERS == 2 && WAVES > 1 && !REG && !SPEC), "two masters need worker waves");

  // per-lane resident triangles (SoA in registers)
  double v0x[SLOTS], v0y[SLOTS], v0z[SLOTS];
  double e1x[SLOTS], e1y[SLOTS], e1z[SLOTS];
  double e2x[SLOTS], e2y[SLOTS], e2z[SLOTS];
  // per-lane resident sphere (lane tid owns sphere tid when tid < nsph)
  double scx, scy, scz, sr2;
  bool hasSphere;

  const TraceParams *p;
  const ptw_camera *cam; // worker-wave kernels: the LDS copy of p->cam
  const double *triGeom;
  const SphereRec *spheresGlobal;
  const double *triCompactGlobal; // REG: source of the per-lane shading records
  const double *matTableGlobal;
  SeqTables tab;
  SeqShared *sh;
  Level *stack;          // this wave's private radiance stack in LDS
  PartialHit *partials;  // [WAVES] cross-wave exchange (WAVES > 1)
  SeqCommand *cmd;       // master -> workers (WAVES > 1); MASTERS == 2: this master's of allCmds[2]
  SeqCommand *allCmds;   // MASTERS == 2: both masters' commands
  unsigned tick;         // MASTERS == 2, lock step: workgroup barriers this wave has executed
  unsigned long long *minSlot; // worker waves: this wave's 2 x 8 bytes of LDS for pickNearest's atomic form
  int masterIndex;
  // Pick checksum (ptw_debug_options.d_picks): sum over the sample's intersect() calls r = 0, 1, ... of
  // (r + 1) * (combined index + 1), misses 0.  pickS1 / pickS2 accumulate sum (idx + 1) and
  // sum (i + 1) (idx + 1) over the calls since pickReset(); a segment that starts at call ordinal b
  // contributes b * S1 + S2 (the speculative kernels commit whole sub-samples at once).
  // PICKS = false compiles it out: the kernels whose wave has its SIMD to itself (WAVES == 1, the
  // speculative kernel) pay an issue slot for every instruction, so their shipped instantiation carries
  // none of this and a second one (launched when d_picks is set) does.
  bool picksOn;
  uint32_t pickS1, pickS2, pickN;
  int tid;               // index among the primitive-holding lanes (workers); master: lane id
  int pos;               // next canonical double in sh->canon (wave-uniform)
  d3 envColour;          // chainHot: the environment colour, kept in vector registers
  char *ringBase;        // SPEC: LDS address of ring slot 0
  unsigned ringOff;      // SPEC: 0 or kRingStride - the slot `pos` indexes
  unsigned words;        // RNG words consumed by the current sample
  unsigned long long rays;
  unsigned parity;
#if PTW_PROFILE_PHASES
  unsigned long long prof[12];
  unsigned long long mprof[6]; // master, inside intersect(): publish, wait B1, shadow work, wait B2, pick
  // master, OUTSIDE intersect(): cycles from one answer to the next published ray, by what the answer
  // was: [ray kind 0 primary / 1 first ray of a sub-sample / 2 deeper][0 hit / 1 miss]
  // (six scalars, not an array: a run-time index would put it in scratch memory and ruin the timing)
  unsigned long long g00, g01, g10, g11, g20, g21, n00, n01, n10, n11, n20, n21, lastExit;
  int rayKind, lastKind, lastMiss;
  unsigned long long pairReq, pairReq2, fanBuilds, fanHits, fanMisses;
#endif

  // Triangle held in slot s of this lane.  WAVES == 1: slot-major (slot s of all lanes covers
  // triangles [64 s, 64 s + 64)).  WAVES > 1: a worker wave holds `myUnits` consecutive units of 64
  // triangles starting at unit `unitBase` - fewer for the waves that share their SIMD with another
  // worker, more for those beside a master (seqUnitSplit) - one unit per slot; slots from myUnits on
  // are empty and skipped with one scalar branch.
  int unitBase, myUnits;
  __device__ __forceinline__ uint32_t slotTriangle(int s) const {
    if (WAVES == 1) return static_cast<uint32_t>(s) * kThreads + static_cast<uint32_t>(tid);
    return static_cast<uint32_t>(unitBase + s) * 64u + static_cast<uint32_t>(tid & 63);
  }
  // triangles resident in the workers' registers (the rest is streamed: localNearest)
  __device__ __forceinline__ uint32_t residentTriangles() const {
    if (WAVES == 1) return static_cast<uint32_t>(kThreads) * SLOTS;
    constexpr int nB = kSideB, nA = WAVES - kSideB;
    return static_cast<uint32_t>((nA / 2) * (p->seqUnitsA + p->seqUnitsY) + nB * p->seqUnitsB) * 64u;
  }

  __device__ __forceinline__ void loadPrimitives() {
    const uint32_t ntri = p->ntri;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      // Branch-free on purpose: with an if/else the compiler sinks the two stores into one with a
      // runtime slot index, which sends the slot arrays to scratch memory.  An unused slot gets a
      // degenerate triangle (det == 0 -> always skipped); triGeom holds at least one record.
      const uint32_t k = slotTriangle(s);
      const bool valid = k < ntri && (WAVES == 1 || s < myUnits);
      const double *g = triGeom + 9 * static_cast<size_t>(valid ? k : 0u);
      v0x[s] = valid ? g[0] : 0.0, v0y[s] = valid ? g[1] : 0.0, v0z[s] = valid ? g[2] : 0.0;
      e1x[s] = valid ? g[3] : 0.0, e1y[s] = valid ? g[4] : 0.0, e1z[s] = valid ? g[5] : 0.0;
      e2x[s] = valid ? g[6] : 0.0, e2y[s] = valid ? g[7] : 0.0, e2z[s] = valid ? g[8] : 0.0;
    }
    if (REG) {
      // lanes without a triangle never win a hit, their record is never read
#pragma unroll
      for (int i = 0; i < kRecDoubles; ++i) rec[i] = 0.0;
      if (static_cast<uint32_t>(tid) < ntri) {
        const double *r = triCompactGlobal + static_cast<size_t>(tid) * kTriCompactDoubles;
        const double *m = matTableGlobal + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
#pragma unroll
        for (int i = 0; i < 6; ++i) rec[i] = m[i]; // emission, diffuse
      }
      const bool emissive = rec[kRecEmission] != 0.0 || rec[kRecEmission + 1] != 0.0 ||
                            rec[kRecEmission + 2] != 0.0;
      emissiveMask = __builtin_amdgcn_ballot_w64(emissive);
      stackBits = 0;
    }
    hasSphere = static_cast<uint32_t>(tid) < p->nsph;
    if (hasSphere) {
      const SphereRec &r = spheresGlobal[tid];
      scx = r.centre[0], scy = r.centre[1], scz = r.centre[2], sr2 = r.radiusSquared;
    } else {
      scx = scy = scz = sr2 = 0;
    }
  }

  // Only the master wave draws random numbers, so it regenerates on its own.
  __device__ __forceinline__ void regenerate() {
#if PTW_PROFILE_PHASES
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    mtRegenerateWave(sh, threadIdx.x & 63);
    laPos = -1; // positions of the old block mean nothing in the new one
#if PTW_EXPERIMENTS
    fanOk = 0; // (PAIR: the tabulated scatters belong to the old block)
#endif
#if PTW_PROFILE_PHASES
    prof[9] += __builtin_amdgcn_s_memtime() - t0;
#endif
  }

  // The same by the calling wave alone, without a workgroup barrier (several masters).
  __device__ __forceinline__ void rebuildCanonWave() {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < kMtDoubles; i += 64)
      sh->canon[i] = canonicalFromWords(mtTemper(sh->mt[2 * i]), mtTemper(sh->mt[2 * i + 1]));
    waveSync();
    fillHemiTable(sh, lane);
    waveSync();
  }

  // Rebuild canon[] from the current raw state without twisting (state resumed mid-block).
  __device__ __forceinline__ void rebuildCanon() {
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      for (int i = lane; i < kMtDoubles; i += 64)
        sh->canon[i] = canonicalFromWords(mtTemper(sh->mt[2 * i]), mtTemper(sh->mt[2 * i + 1]));
      waveSync();
      fillHemiTable(sh, lane);
    }
    __syncthreads();
  }

  // SPEC accessors: canon / hemi of the slot `pos` indexes
  __device__ __forceinline__ const double *ringCanon() const {
    return reinterpret_cast<const double *>(ringBase + ringOff);
  }
  __device__ __forceinline__ const double *ringHemi(int q) const {
    return reinterpret_cast<const double *>(ringBase + ringOff + kRingHemiOff) + 3 * q;
  }
  // SPEC: consume n draws (branch-free wrap into the other slot)
  __device__ __forceinline__ void advance(int n) {
    const int np = pos + n;
    const bool wrap = np >= kMtDoubles;
    pos = wrap ? np - kMtDoubles : np;
    ringOff = wrap ? ringOff ^ kRingStride : ringOff;
    words += 2 * n;
  }
  __device__ __forceinline__ void setStream(unsigned off, int q) {
    ringOff = off;
    pos = q;
  }

  __device__ __forceinline__ double draw() {
    if (SPEC) {
      const double v = ringCanon()[pos];
      advance(1);
      return v;
    }
    if (pos == kMtDoubles) {
      regenerate();
      pos = 0;
    }
    words += 2;
    return sh->canon[pos++];
  }
  // consecutive draws with one LDS round trip when they do not straddle a regeneration
  __device__ __forceinline__ void draw3(double &a, double &b, double &c) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      a = cn[0], b = cn[1], c = cn[2];
      advance(3);
      return;
    }
    if (pos + 3 <= kMtDoubles) {
      a = sh->canon[pos];
      b = sh->canon[pos + 1];
      c = sh->canon[pos + 2];
      pos += 3;
      words += 6;
    } else {
      a = draw();
      b = draw();
      c = draw();
    }
  }
  __device__ __forceinline__ void draw4(double &a, double &b, double &c, double &d) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      a = cn[0], b = cn[1], c = cn[2], d = cn[3];
      advance(4);
      return;
    }
    if (pos + 4 <= kMtDoubles) {
      a = sh->canon[pos];
      b = sh->canon[pos + 1];
      c = sh->canon[pos + 2];
      d = sh->canon[pos + 3];
      pos += 4;
      words += 8;
    } else {
      a = draw();
      b = draw();
      c = draw();
      d = draw();
    }
  }

  // The nearest of the lanes' candidates (t, combined index, determinant) with the reference's
  // tie-break, as a wave-uniform result.
  // `slot`: the calling wave's 8 bytes of LDS for the atomic form of the many-candidates case (worker
  // waves), nullptr for the DPP form.
  __device__ __forceinline__ static HitKey pickNearest(double bestT, uint32_t bestIdx, double bestDet,
                                                        unsigned long long *slot = nullptr) {
    HitKey key;
    // Most rays leave at most two lanes with a candidate (the line through a closed scene crosses
    // few primitives on its positive side): pick the nearer of them with scalar code instead of
    // a 64-lane reduction.
    const unsigned long long cands = __builtin_amdgcn_ballot_w64(bestIdx != kMiss);
    const int ncand = __builtin_popcountll(cands);
    // Only the sign test `det < epsilon` of the winner's determinant is ever used: inside this function
    // it travels as bit 31 of the index word (the answers exchanged BETWEEN waves carry it in bit 0:
    // packAnswer), which saves the two cross-lane reads of the determinant (a v_readlane with a computed
    // lane costs a lone wave four issue slots).
    const uint32_t packed = bestIdx | (bestDet < kEpsilon ? 0x80000000u : 0u);
    uint32_t pw;
    if (ncand == 0) {
      key.t = kInf, key.idx = kMiss, key.det = 0;
      return key;
    } else if (ncand == 1) {
      const int la = __builtin_ctzll(cands);
      key.t = readLane(bestT, la);
      pw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), la));
    } else if (ncand == 2) {
      const int la = __builtin_ctzll(cands);
      const int lb = __builtin_ctzll(cands & (cands - 1));
      const double ta = readLane(bestT, la), tb = readLane(bestT, lb);
      const uint32_t pa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), la));
      const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), lb));
      // strictly nearer wins; an exact tie goes to the lower combined index (Scene.cpp:31,95,118)
      const bool pickB = uniformBool((tb < ta) | ((tb == ta) & ((pb & 0x7fffffffu) < (pa & 0x7fffffffu))));
      key.t = pickB ? tb : ta;
      pw = pickB ? pb : pa;
    } else {
      // (walking three to six candidates with scalar code instead - three v_readlane and a few
      // scalar compares each - measured slower: Cornell 7.28 against 7.69 Msamples/s,
      // profiles/r02r_pick_loop_probe.txt)
      unsigned tHi, tLo;
      double tmin;
      if (WAVES > 1 && slot) {
        // Distances are positive doubles: their bit patterns order like unsigned 64-bit integers.  The
        // first candidate lane resets the slot, every candidate lane folds its distance in with one
        // ds_min_u64, everybody reads the result - three LDS instructions of one wave to one address,
        // served in order - instead of twelve DPP steps.
        typedef unsigned long long __attribute__((address_space(3))) LdsU64;
        LdsU64 *ls = (LdsU64 *)(slot);
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(bestT));
        if (static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(cands)) *(volatile LdsU64 *)ls = ~0ull;
        asm volatile("" ::: "memory");
        if (bestIdx != kMiss) (void)__hip_atomic_fetch_min(ls, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        asm volatile("" ::: "memory");
        const unsigned long long tb = *(volatile LdsU64 *)ls;
        tHi = static_cast<unsigned>(tb >> 32), tLo = static_cast<unsigned>(tb);
        tmin = __longlong_as_double(static_cast<long long>(tb));
      } else {
        tmin = waveMinPositive(bestT, tHi, tLo);
      }
      unsigned long long owner = __builtin_amdgcn_ballot_w64(
          static_cast<unsigned>(hi32(bestT)) == tHi && static_cast<unsigned>(lo32(bestT)) == tLo);
      if (__builtin_popcountll(owner) != 1) { // exact tie between lanes: lowest combined index wins
        const uint32_t imin = waveMinUFused(bestT == tmin ? bestIdx : kMiss);
        owner = __builtin_amdgcn_ballot_w64(bestIdx == imin);
      }
      key.t = tmin;
      pw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), __builtin_ctzll(owner)));
    }
    key.idx = pw & 0x7fffffffu;
    key.det = (pw >> 31) ? -1.0 : 1.0;
    return key;
  }

  // This wave's part of Scene::intersect (Scene.cpp:115-122): its lanes' resident primitives
  // against the ray, then the wave-level nearest hit with the reference's tie-break.
  __device__ __forceinline__ HitKey localNearest(d3 o, d3 d) {
    PTW_T(tA);
    double bestT = kInf, bestDet = 0;
    uint32_t bestIdx = kMiss;
    const uint32_t nsph = p->nsph;
    // spheres first (lower combined index)
    if (hasSphere) testSphere(o, d, mk(scx, scy, scz), sr2, static_cast<uint32_t>(tid), bestT, bestIdx);
    if (!REG && nsph > static_cast<uint32_t>(kThreads)) // rare: more spheres than lanes
      for (uint32_t i = tid + kThreads; i < nsph; i += kThreads) {
        const SphereRec &r = spheresGlobal[i];
        testSphere(o, d, ld3(r.centre), r.radiusSquared, i, bestT, bestIdx);
      }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      // (wave-uniform: this wave's share of the scene ends at myUnits.  A guard, not a `break`: with a
      // second loop exit the compiler stops unrolling from nine slots on, indexes the slot arrays
      // at run time and moves them to scratch memory)
      if (WAVES > 1 && s >= myUnits) continue;
      testTriangle(o, d, mk(v0x[s], v0y[s], v0z[s]), mk(e1x[s], e1y[s], e1z[s]),
                   mk(e2x[s], e2y[s], e2z[s]), nsph + slotTriangle(s),
                   bestT, bestIdx, bestDet);
    }
    // rare: more triangles than resident slots -> stream the remainder from memory
    if (!REG && p->ntri > residentTriangles())
      for (uint32_t k = residentTriangles() + tid; k < p->ntri; k += kThreads) {
        const double *g = triGeom + 9 * static_cast<size_t>(k);
        testTriangle(o, d, ld3(g), ld3(g + 3), ld3(g + 6), nsph + k, bestT, bestIdx, bestDet);
      }

    // wave reduction: lexicographic min of (t, idx)
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(bestT));
#endif
    PTW_T(tB);
    PTW_ACC(0, tA, tB);
    HitKey key = pickNearest(bestT, bestIdx, bestDet, WAVES > 1 ? minSlot : nullptr);
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(key.t));
#endif
    PTW_T(tC);
    PTW_ACC(1, tB, tC);
    return key;
  }

  __device__ __forceinline__ void setLookAheadFrame(const Surface &s, d3 dirIn, double invU, double invV) {
    laSurf = s, laDir = dirIn, laInvU = invU, laInvV = invV;
  }
  __device__ __forceinline__ void armLookAhead(bool on, int nu, int nv) {
    laArmed = on && (laMisses < 2 || (++laTick & 7u) == 0);
    laU = nu, laV = nv;
  }
  __device__ __forceinline__ void lookAhead() {
    laPos = -1;
    if (pos + 3 <= kMtDoubles) { // (a sub-sample whose draws straddle a regeneration takes the plain path)
      const double xu = sh->canon[pos], xv = sh->canon[pos + 1], pd = sh->canon[pos + 2];
      double u, v;
      stratify(*p, laU, laV, xu, xv, laInvU, laInvV, u, v);
      laRefl = scatter(*this, laSurf, laDir, u, v, pd, laOut);
      laPos = pos;
    }
  }
  __device__ __forceinline__ bool takeLookAhead(d3 &dirOut, bool &refl) {
    const bool hit = laPos == pos; // wave-uniform (pos >= 0)
    if (laArmed) laMisses = hit ? 0 : (laMisses < 2 ? laMisses + 1 : 2);
    laPos = -1;
    laArmed = false;
    if (!hit) return false;
    dirOut = laOut;
    refl = laRefl;
    pos += 3;
    words += 6;
    return true;
  }

  __device__ __forceinline__ void flushPending() {
    if (pendKind == 0) return;
    const double *r = tab.tri + static_cast<size_t>(pendIdx - p->nsph) * kTriCompactDoubles;
    const double *m = tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
    push(pendLevel, ld3(m), ld3(m + 3), false, pendIdx);
    pendKind = 0;
  }
  __device__ __forceinline__ void setPending(int level, uint32_t idx) {
    pendKind = 1, pendLevel = level, pendIdx = idx;
  }

  // The nearest of the WAVES workers' answers with the reference's tie-break (strictly nearer wins,
  // an exact tie goes to the lower combined index: Scene.cpp:31,95,118).  Every lane of the master
  // reads all answers (broadcast LDS reads, one wait) and runs the same pick: no cross-lane traffic at
  // all, where the general pick spends 0.5-0.9 k cycles on ballots, readlanes and - from three
  // candidates on - a 64-lane reduction.  PAIR: the answers of command slot `slot` (0 or 1).
  __device__ __forceinline__ HitKey pickPartials(int slot = 0) const {
    PartialHit ph[WAVES];
    // One ds_read_b128 per answer, all issued before the first is used (copied member by member the
    // compiler reads 8 + 4 bytes each - twelve LDS instructions for six answers): the
    // empty asm statement makes all four dwords of every answer count as used.
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    const U4 *src = reinterpret_cast<const U4 *>(partials + slot * WAVES);
    static_assert(WAVES == 1 || WAVES == 6 || WAVES == 7, "six or seven worker waves");
    U4 raw[WAVES];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) raw[w] = src[w];
    if constexpr (WAVES == 6)
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]));
    if constexpr (WAVES == 7)
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]));
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      ph[w].t = mk64(static_cast<int>(raw[w].x), static_cast<int>(raw[w].y));
      ph[w].idxSign = raw[w].z;
      ph[w].pad = 0;
    }
    return pickOfAnswers(ph);
  }

  // Pick checksum bookkeeping (see picksOn).
  __device__ __forceinline__ void pickReset() { pickS1 = 0, pickS2 = 0, pickN = 0; }
  __device__ __forceinline__ void pickNote(const HitKey &k) {
    const uint32_t v = k.idx == kMiss ? 0u : k.idx + 1u;
    pickN += 1u;
    pickS1 += v;
    pickS2 += pickN * v;
  }

  // Scene::intersect for the whole workgroup.  WAVES == 1: the wave's own result.  WAVES > 1:
  // wave 0 (the master, the only wave that runs the path logic) publishes the ray, every wave
  // searches its resident primitives, the partial results meet in LDS.  The worker waves sit in
  // workerLoop() and do nothing but this - while the master shades, their SIMDs are free for
  // the waves of other passes.
  __device__ __forceinline__ HitKey intersect(d3 o, d3 d) {
    rays++;
    if (WAVES == 1) {
      const HitKey key = localNearest(o, d);
      if constexpr (PICKS) if (picksOn) pickNote(key);
      return key;
    }
    PTW_T(tM0);
#if PTW_PROFILE_PHASES
    if (lastExit) {
      const unsigned long long gap = tM0 - lastExit;
      const int which = lastKind * 2 + lastMiss;
      if (which == 0) g00 += gap, n00++;
      if (which == 1) g01 += gap, n01++;
      if (which == 2) g10 += gap, n10++;
      if (which == 3) g11 += gap, n11++;
      if (which == 4) g20 += gap, n20++;
      if (which == 5) g21 += gap, n21++;
    }
#endif
    if ((threadIdx.x & 63) == 0) { // (the master wave's first lane; cmd / partials are this master's)
      cmd->o[0] = o.x, cmd->o[1] = o.y, cmd->o[2] = o.z;
      cmd->d[0] = d.x, cmd->d[1] = d.y, cmd->d[2] = d.z;
      if (MASTERS == 1) cmd->op = kCmdTrace;
    }
#if PTW_PROFILE_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMa);
    ldsBarrier(); // B1: ray visible to the workers
    PTW_T(tMb);
    // the search takes a thousand cycles and more: the stack entry of the level just left ...
    flushPending();
    if (laArmed) lookAhead(); // ... and the next sub-sample's first-bounce scatter
#if PTW_PROFILE_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMc);
    ldsBarrier(); // B2: partial results visible
    PTW_T(tMd);
#if PTW_PROFILE_PHASES
    mprof[0] += tMa - tM0, mprof[1] += tMb - tMa, mprof[2] += tMc - tMb, mprof[3] += tMd - tMc;
#endif
    // (MASTERS == 2: the same two barriers - the workers search this ray between them, and the
    // other master's ray between B2 and this master's next B1, i.e. while this one shades)
    if (MASTERS == 2) tick += 2;
    const HitKey key = pickPartials();
    if constexpr (PICKS) if (picksOn) pickNote(key);
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(const_cast<HitKey &>(key).t));
#endif
    PTW_T(tM1);
    PTW_ACC(5, tM0, tM1);
#if PTW_PROFILE_PHASES
    mprof[4] += tM1 - tMd;
    lastKind = rayKind, lastMiss = key.idx == kMiss ? 1 : 0, lastExit = __builtin_amdgcn_s_memtime();
    rayKind = 2; // (whoever traces a primary ray or a sub-sample's first ray says so before the call)
#endif
    return key;
  }

  // PAIR: one request for the rays that sit in this master's command slots (`mask`: bit c = slot c
  // holds a ray; written by startChain / advance).  The answers are picked per slot by the caller.
  // `shadow` runs between the two barriers, i.e. while the workers search.
  template <typename Shadow>
  __device__ __forceinline__ void searchSlots(uint32_t mask, Shadow &&shadow) {
    if ((threadIdx.x & 63) == 0) cmd->nrays = mask;
#if PTW_PROFILE_PHASES
    PTW_T(tM0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMa);
    ldsBarrier(); // B1
    PTW_T(tMb);
    shadow();
#if PTW_PROFILE_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMc);
    ldsBarrier(); // B2
    PTW_T(tMd);
#if PTW_PROFILE_PHASES
    mprof[0] += tMa - tM0, mprof[1] += tMb - tMa, mprof[2] += tMc - tMb, mprof[3] += tMd - tMc;
    prof[5] += tMd - tM0;
    mprof[5] += mask == 3u ? 1 : 0;
#endif
    tick += 2;
  }

  // Worker waves (WAVES > 1, wave != 0): serve nearest-hit requests until told to stop.
  __device__ __forceinline__ void workerLoop() {
#if PTW_PROFILE_PHASES
    for (int i = 0; i < 12; ++i) prof[i] = 0;
    unsigned long long nreq = 0, nreq2 = 0;
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (PAIR) {
#if PTW_EXPERIMENTS
#if PTW_PROFILE_PHASES
      workerLoopPair(nreq, nreq2);
#else
      unsigned long long unusedA = 0, unusedB = 0;
      workerLoopPair(unusedA, unusedB);
#endif
#endif
    } else if (MASTERS == 2) {
      // Barrier n is followed by the search of master (n & 1)'s ray, which that master published
      // before it.  A command's `op` holds the barrier index from which its master has no more
      // rays (kCmdLive while it has): a value that reads the same whenever it is looked at, so all
      // waves leave after the same barrier.
      for (unsigned n = 0;; ++n) {
        ldsBarrier();
        const SeqCommand &c = allCmds[n & 1];
        const uint32_t mine = c.op, other = allCmds[(n & 1) ^ 1].op;
        if (mine <= n) {
          if (other <= n) break;
          continue;
        }
        const d3 o = mk(c.o[0], c.o[1], c.o[2]);
        const d3 d = mk(c.d[0], c.d[1], c.d[2]);
        const HitKey found = localNearest(o, d);
        if ((tid & 63) == 0) {
          PartialHit ph;
          ph.t = found.t, ph.pad = 0;
          ph.idxSign = packAnswer(found);
          partials[(n & 1) * WAVES + (tid >> 6)] = ph;
        }
        // (Round 5 also had the worker wave that answers LAST - an LDS counter per master - pick the
        // nearest of the six answers, so that the master reads one: the 0.4 k cycles moved from the
        // master's tick to the search's tail, and suzanne ran 10.0 against 11.7, ce 1.90 against 2.11
        // Msamples/s, profiles/r05h_*: the tick is the maximum of both, not the master's alone.)
#if PTW_PROFILE_PHASES
        nreq++;
#endif
      }
    } else
    for (;;) {
      ldsBarrier(); // B1
      if (cmd->op == kCmdExit) break;
      const d3 o = mk(cmd->o[0], cmd->o[1], cmd->o[2]);
      const d3 d = mk(cmd->d[0], cmd->d[1], cmd->d[2]);
      const HitKey mine = localNearest(o, d);
      if ((tid & 63) == 0) {
        PartialHit ph;
        ph.t = mine.t, ph.pad = 0;
        ph.idxSign = packAnswer(mine);
        partials[tid >> 6] = ph;
      }
#if PTW_PROFILE_PHASES
      nreq++;
#endif
      ldsBarrier(); // B2
    }
#if PTW_PROFILE_PHASES
    if (blockIdx.x == 0 && (tid & 63) == 0) { // every worker wave: which ones are the slow ones?
      const unsigned long long w1 = __builtin_amdgcn_s_memtime();
      printf("WORKER rank=%d (hardware wave %d, %d units) requests=%llu (two rays: %llu) total/req=%.0f tests=%.0f reduce=%.0f\n", tid >> 6,
             (int)(threadIdx.x >> 6), myUnits, nreq, nreq2, (double)(w1 - w0) / nreq, (double)prof[0] / nreq, (double)prof[1] / nreq);
    }
#endif
  }
  __device__ __forceinline__ void stopWorkers() {
    if (WAVES == 1) return;
    if (MASTERS == 2) {
      // no more rays from this master as of its next barrier; keep the cadence until the other
      // one is done too (see workerLoop)
      if ((threadIdx.x & 63) == 0) cmd->op = tick;
      for (unsigned n = tick;; ++n) {
        ldsBarrier();
        if (allCmds[0].op <= n && allCmds[1].op <= n) break;
      }
      return;
    }
    if (threadIdx.x == 0) cmd->op = kCmdExit;
    ldsBarrier(); // pairs with the workers' B1
  }

  __device__ __forceinline__ bool branch(bool b) const { return uniformBool(b); }

  __device__ __forceinline__ d3 recD3(int at, int lane) const {
    return mk(readLane(rec[at], lane), readLane(rec[at + 1], lane), readLane(rec[at + 2], lane));
  }
  __device__ __forceinline__ d3 emissionAt(const HitKey &k) const {
    if (REG && k.idx >= p->nsph) {
      const int lane = static_cast<int>(k.idx - p->nsph);
      if (!((emissiveMask >> lane) & 1ull)) return mk(0, 0, 0);
      return recD3(kRecEmission, lane);
    }
    if (k.idx >= p->nsph) {
      const double *r = tab.tri + static_cast<size_t>(k.idx - p->nsph) * kTriCompactDoubles;
      return ld3(tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles);
    }
    return ld3(tab.sph[k.idx].emission);
  }
  // consume three draws without looking at them
  __device__ __forceinline__ void skip3() {
    if (SPEC) {
      advance(3);
      return;
    }
    if (pos + 3 <= kMtDoubles) {
      pos += 3;
      words += 6;
    } else {
      (void)draw();
      (void)draw();
      (void)draw();
    }
  }

  // The same scatter with the three draws at block position q (q + 3 <= kMtDoubles), without touching
  // the stream position: the frontier chain calls it through scatterChain(), a speculated chain (PAIR)
  // with its own position.
  __device__ __forceinline__ bool scatterChainAt(int q, const Surface &s, d3 dirIn, d3 &dirOut) const {
    const double pd = sh->canon[q + 2];
    const d3 local = mk(sh->hemi[q][0], sh->hemi[q][1], sh->hemi[q][2]);
    if (uniformBool(lobeIsReflective(s, dirIn, pd))) { // Scene.cpp:163-168
      dirOut = coneSample(reflect(s.normal, dirIn), s.coneAngle, sh->canon[q], sh->canon[q + 1]);
      return true;
    }
    dirOut = normalisedNearUnit(transform(s.basis, local)); // Scene.cpp:169-175
    return false;
  }

  // The scatter of a single-sample level (depth >= 1): u = xi1, v = xi2, p = xi3 drawn in that
  // order (Scene.cpp:157-161).  When the three draws sit inside the current block, the diffuse
  // lobe takes its local direction from the precomputed table.
  __device__ __forceinline__ bool scatterChain(const Surface &s, d3 dirIn, d3 &dirOut) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      const double *hm = ringHemi(pos);
      const double u = cn[0], v = cn[1], pd = cn[2];
      const d3 local = mk(hm[0], hm[1], hm[2]);
      advance(3);
      if (uniformBool(lobeIsReflective(s, dirIn, pd))) { // Scene.cpp:163-168
        dirOut = coneSample(reflect(s.normal, dirIn), s.coneAngle, u, v);
        return true;
      }
      dirOut = normalisedNearUnit(transform(s.basis, local)); // Scene.cpp:169-175
      return false;
    }
    if (pos + 3 <= kMtDoubles) {
      const int q = pos;
      pos += 3;
      words += 6;
      return scatterChainAt(q, s, dirIn, dirOut);
    }
    double u, v, pd;
    draw3(u, v, pd); // straddles a regeneration
    Surface r = s;
    r.reflectivity = resolveReflectivity(s, dirIn);
    return scatter(*this, r, dirIn, u, v, pd, dirOut);
  }

  // `p < reflectivity` (Scene.cpp:143-146,163) without always evaluating Norm3::reflectance.
  // For ior == 1 on both sides the reflectance is ((c - c') / (c + c'))^2 with c' = sqrt(1 - (1 -
  // c^2)) differing from c = cos(theta_i) only by rounding: |c'^2 - c^2| <= 3e-16, so for
  // c >= 1e-3 the value is below 2^-64, the spacing of the canonical draws - `p < reflectivity`
  // can then only hold for p == 0.  Everything else takes the exact evaluation.
  __device__ __forceinline__ bool lobeIsReflective(const Surface &s, d3 dirIn, double pd) const {
    if (uniformBool(s.matReflectivity >= 0)) return pd < s.matReflectivity;
    if (uniformBool(s.iorFrom == 1.0 && s.iorTo == 1.0)) {
      const double cosThetaI = -dot(s.normal, dirIn);
      if (uniformBool(cosThetaI >= 1e-3 && pd > 0.0)) return false;
    }
    return pd < reflectance(s.normal, dirIn, s.iorFrom, s.iorTo, s.iorRatio);
  }
#if PTW_PROFILE_PHASES
  __device__ __forceinline__ void markRay(int kind) { rayKind = kind; }
  __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void acc(int slot, unsigned long long t0, double &keep) {
    asm volatile("" : "+v"(keep));
    prof[slot] += __builtin_amdgcn_s_memtime() - t0;
  }
#else
  __device__ __forceinline__ void markRay(int) {}
  __device__ __forceinline__ unsigned long long now() const { return 0; }
  __device__ __forceinline__ void acc(int, unsigned long long, double &) {}
#endif
  __device__ __forceinline__ void push(int level, d3 e, d3 dif, bool refl, uint32_t idx) {
    if (REG) {
      const unsigned sh8 = static_cast<unsigned>(level) * 8u;
      const unsigned long long w =
          static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(idx)) & 0x7f) |
          (refl ? 0x80ull : 0ull);
      stackBits = (stackBits & ~(0xffull << sh8)) | (w << sh8);
      return;
    }
    // One lane stores (64 lanes writing one address would serialise in the LDS); every lane
    // reads it back later.  The address is the same for the store and the loads, so the
    // compiler keeps them ordered.
    if ((tid & 63) == 0) {
      Level lv;
      lv.emission = e;
      lv.diffuse = dif;
      lv.reflective = refl;
      stack[level] = lv;
    }
  }
  __device__ __forceinline__ Level top(int level) const {
    if (REG) {
      const unsigned w = static_cast<unsigned>(stackBits >> (static_cast<unsigned>(level) * 8u)) & 0xffu;
      const uint32_t idx = w & 0x7fu;
      Level lv;
      lv.reflective = (w >> 7) != 0;
      if (idx >= p->nsph) {
        const int lane = static_cast<int>(idx - p->nsph);
        lv.emission = recD3(kRecEmission, lane);
        lv.diffuse = recD3(kRecDiffuse, lane);
      } else {
        lv.emission = ld3(tab.sph[idx].emission);
        lv.diffuse = ld3(tab.sph[idx].diffuse);
      }
      return lv;
    }
    return stack[level];
  }

  __device__ __forceinline__ d3 runChain(const TraceParams &tp, const TriShade *ts, const SphereRec *sp,
                                         d3 o, d3 d) {
    if (REG) return chainHot(tp, o, d);
    // (two masters: suzanne 512 passes +1 %, ce +6 %; with one master per workgroup it measured 5 %
    // slower than radianceChain - profiles/r03c_worker_wave_master_path.txt - and is not used there)
    if (WAVES > 1 && MASTERS == 2) return chainMaster(tp, o, d);
    return radianceChain(*this, tp, ts, sp, o, d);
  }

  // radianceChain() for the master wave of the worker-wave kernels (WAVES > 1), arranged like
  // chainHot() around what the master's serial path pays for.  With two masters per workgroup a
  // tick of the protocol lasts as long as the slower of "the workers search one master's ray" and
  // "the other master reads the answers, picks, shades and publishes" - on suzanne the latter
  // (1.2 k against 2.1 k cycles per ray).  The common level - a triangle hit, the
  // diffuse lobe, the draws inside the generator block - is decided by two branches with all its
  // LDS operands (triangle record with the lobe threshold, the draw, the draw-derived local
  // direction) waited for once; the level's colours are not needed before the fold, so their fetch
  // (triangle record -> material index -> material: dependent round trips) and the stack entry are
  // left to flushPending(), which runs while the workers search the NEXT ray.  Everything else
  // takes the general code, which is the sequence radianceChain() runs.
  __device__ __forceinline__ d3 chainMaster(const TraceParams &tp, d3 o, d3 d) {
    if (tp.maxDepth <= 1) return mk(0, 0, 0); // Scene.cpp:128 at depth 1
    const HitKey k = intersect(o, d);
    if (uniformBool(k.idx == kMiss)) return envColour; // Scene.cpp:131-133
    return chainMasterFrom(tp, o, d, k);
  }
  // ... from the first hit `k` of the ray (o, d) on (not a miss: the fan-out loop of radiance0 deals
  // with the sub-sample whose first ray leaves the scene - four of five on suzanne - itself, in a
  // handful of instructions).
  __device__ __forceinline__ d3 chainMasterFrom(const TraceParams &tp, d3 o, d3 d, HitKey k) {
    int nlev = 0;
    d3 L;
    const int maxDepth = tp.maxDepth;
    const uint32_t nsph = tp.nsph, ntri = tp.ntri;
    int depth = 1;
    for (;; k = intersect(o, d)) {
      const int notLast = depth + 1 - maxDepth;   // < 0
      const int inBlock = pos + 2 - kMtDoubles;   // < 0  <=>  pos + 3 <= kMtDoubles
      const bool isTri = (k.idx - nsph) < ntri;   // unsigned: kMiss and spheres fail
      if (isTri & ((notLast & inBlock) < 0)) {
        const unsigned long long tH0 = now();
        const double *r = tab.tri + static_cast<size_t>(k.idx - nsph) * kTriCompactDoubles;
        const int q = pos;
        d3 n = ld3(r), bx = ld3(r + 3), by = ld3(r + 6);
        double thr = r[kTriLobeThreshold];
        double pd = sh->canon[q + 2];
        const double *hm = sh->hemi[q];
        d3 local = mk(hm[0], hm[1], hm[2]);
        asm volatile("" : "+v"(n.x), "+v"(bx.x), "+v"(by.x), "+v"(thr), "+v"(pd), "+v"(local.x)); // one wait
        const bool backfacing = k.det < kEpsilon; // Scene.cpp:107
        const double ndotd = dot(n, d);
        const double cosThetaI = backfacing ? ndotd : -ndotd; // -dot(+-n, d)
        const unsigned long long mNotRefl = __builtin_amdgcn_ballot_w64(!(pd < thr));
        const unsigned long long mPlain = __builtin_amdgcn_ballot_w64(thr >= 0.0);
        const unsigned long long mCos = __builtin_amdgcn_ballot_w64(cosThetaI >= 1e-3);
        const unsigned long long mPos = __builtin_amdgcn_ballot_w64(pd > 0.0);
        if ((mNotRefl & (mPlain | (mCos & mPos))) != 0) { // the diffuse lobe (see chainHot / ptw_layout.h)
          pos += 3;
          words += 6;
          Basis b;
          b.x = bx, b.y = by, b.z = n;
          const double sgn = backfacing ? -1.0 : 1.0;
          const d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
          o = o + d * k.t;
          d = nd;
          setPending(nlev++, k.idx);
          ++depth;
          acc(11, tH0, d.x);
          continue;
        }
      }
      // ---- general path: miss, last level, sphere, reflective lobe, straddling draws ----
      if (uniformBool(k.idx == kMiss)) { // Scene.cpp:131-133
        L = envColour;
        break;
      }
      if (depth + 1 >= maxDepth) { // last level: see radianceChain()
        skip3();
        L = emissionAt(k);
        break;
      }
      const Surface s = surfaceAt(k, o, d, false);
      d3 nd;
      const bool refl = scatterChain(s, d, nd);
      push(nlev++, s.emission, s.diffuse, refl, k.idx); // (rare: written at once)
      o = s.pos;
      d = nd;
      ++depth;
    }
    flushPending();
    for (int i = nlev - 1; i >= 0; --i) L = fold(i, L);
    return L;
  }

#if PTW_EXPERIMENTS
  // (the PAIR form of the two-master kernels: built and measured slower in round 5 - DESIGN.md 3.1e)
#include "experiments/ptw_pair.h"
#endif

  // radianceChain() for the REG variant, arranged around what a single wave per SIMD pays for:
  // every instruction is one issue slot, and a branch - even an untaken one - costs five to ten
  // of them (scripts/microbench/issue_costs.hip).  The common case, a chain level that hits a
  // triangle and takes the diffuse lobe with its three draws inside the current generator block,
  // is decided by TWO branches: one on scalar facts about the hit, one on the lobe predicate
  // (evaluated as lane-mask logic from the per-triangle lobe threshold, see ptw_layout.h).  All
  // its LDS operands (triangle record, draws, draw-derived local direction) are requested
  // together and waited for once.  Everything else - miss, last level, sphere, reflective lobe,
  // Fresnel evaluation, draws straddling a regeneration - takes the general code below, which
  // is the same sequence radianceChain() runs.
  //   backfacing hits: surfaceAt() negates the normal and basis.x; negation commutes exactly
  //   with the products of transform(), so the signs go onto the local direction instead.
  __device__ __forceinline__ d3 chainHot(const TraceParams &tp, d3 o, d3 d) {
    int nlev = 0;
    d3 L;
    const int maxDepth = tp.maxDepth;
    const uint32_t nsph = tp.nsph, ntri = tp.ntri;
    if (maxDepth <= 1) return mk(0, 0, 0); // Scene.cpp:128 at depth 1
    int depth = 1;
    for (;;) {
      HitKey k;
      // ---- hot loop: stays inside while every level is a diffuse triangle bounce ----
      for (;;) {
        k = intersect(o, d);
        // hit a triangle (not a miss, not a sphere), not the last level, draws inside the block:
        // three differences that are all negative exactly then
        const int notLast = depth + 1 - maxDepth;       // < 0
        const int inBlock = SPEC ? -1 : pos + 2 - kMtDoubles; // < 0  <=>  pos + 3 <= kMtDoubles
        const bool isTri = (k.idx - nsph) < ntri;       // unsigned: kMiss and spheres fail
        if (!(isTri & ((notLast & inBlock) < 0))) break;
        const double *r = tab.tri + static_cast<size_t>(k.idx - nsph) * kTriCompactDoubles;
        const int q = pos;
        d3 n = ld3(r), bx = ld3(r + 3), by = ld3(r + 6);
        double thr = r[kTriLobeThreshold];
        double pd = SPEC ? ringCanon()[q + 2] : sh->canon[q + 2];
        const double *hm = SPEC ? ringHemi(q) : sh->hemi[q];
        d3 local = mk(hm[0], hm[1], hm[2]);
        // one wait for everything: without this the loads the lobe test does not need sink below
        // its branch and are waited for a second time
        asm volatile("" : "+v"(n.x), "+v"(bx.x), "+v"(by.x), "+v"(thr), "+v"(pd), "+v"(local.x));
        const bool backfacing = k.det < kEpsilon; // Scene.cpp:107
        const double ndotd = dot(n, d);
        const double cosThetaI = backfacing ? ndotd : -ndotd; // -dot(+-n, d)
        // lobe predicate as lane-mask logic (all lanes agree)
        const unsigned long long mNotRefl = __builtin_amdgcn_ballot_w64(!(pd < thr));
        const unsigned long long mPlain = __builtin_amdgcn_ballot_w64(thr >= 0.0);
        const unsigned long long mCos = __builtin_amdgcn_ballot_w64(cosThetaI >= 1e-3);
        const unsigned long long mPos = __builtin_amdgcn_ballot_w64(pd > 0.0);
        if ((mNotRefl & (mPlain | (mCos & mPos))) == 0) break;
        if (SPEC) {
          advance(3);
        } else {
          pos += 3;
          words += 6;
        }
        Basis b;
        b.x = bx, b.y = by, b.z = n;
        const double sgn = backfacing ? -1.0 : 1.0;
        const d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
        o = o + d * k.t;
        d = nd;
        push(nlev++, mk(0, 0, 0), mk(0, 0, 0), false, k.idx);
        ++depth;
      }
      // ---- general path: miss, last level, sphere, reflective lobe, straddling draws ----
      if (uniformBool(k.idx == kMiss)) { // Scene.cpp:131-133
        L = envColour;
        break;
      }
      if (depth + 1 >= maxDepth) { // last level: see radianceChain()
        skip3();
        L = emissionAt(k);
        break;
      }
      const Surface s = surfaceAt(k, o, d, false);
      d3 nd;
      const bool refl = scatterChain(s, d, nd);
      push(nlev++, s.emission, s.diffuse, refl, k.idx);
      o = s.pos;
      d = nd;
      ++depth;
    }
    for (int i = nlev - 1; i >= 0; --i) L = fold(i, L);
    return L;
  }

  // One step of the innermost-first fold: L_level = E + T * L_child (Scene.cpp:163-175).
  __device__ __forceinline__ d3 fold(int level, d3 L) const {
    if (REG) {
      const unsigned w = static_cast<unsigned>(stackBits >> (static_cast<unsigned>(level) * 8u)) & 0xffu;
      const uint32_t idx = w & 0x7fu;
      const bool refl = (w >> 7) != 0;
      if (idx >= p->nsph) {
        const int lane = static_cast<int>(idx - p->nsph);
        if (!((emissiveMask >> lane) & 1ull)) {
          // E == +0 and the child radiance is never negative: E + x == x exactly
          return refl ? L : recD3(kRecDiffuse, lane) * L;
        }
        const d3 e = recD3(kRecEmission, lane);
        return refl ? e + L : e + recD3(kRecDiffuse, lane) * L;
      }
    }
    const Level lv = top(level);
    return uniformBool(lv.reflective) ? lv.emission + L : lv.emission + lv.diffuse * L;
  }

  // Surface at a hit from the shading tables (same values as makeSurface()).
  __device__ __forceinline__ Surface surfaceAt(const HitKey &k, d3 o, d3 d, bool eager = true) {
    PTW_T(tA);
    Surface s;
    s.pos = o + d * k.t;
    double ior, invIor, reflectivity;
    bool inside;
    if (k.idx >= p->nsph) {
      const double *r = tab.tri + static_cast<size_t>(k.idx - p->nsph) * kTriCompactDoubles;
      const bool backfacing = k.det < kEpsilon;
#if PTW_PROFILE_PHASES
      PTW_T(tL0);
#endif
      d3 n = ld3(r), bx = ld3(r + 3);
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n.x), "+v"(bx.x)::"memory");
      PTW_T(tL1);
      PTW_ACC(4, tL0, tL1);
#endif
      s.normal = backfacing ? -n : n;
      s.basis.x = backfacing ? -bx : bx;
      s.basis.y = ld3(r + 6);
      s.basis.z = s.normal;
      const double *m = tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
      s.emission = ld3(m);
      s.diffuse = ld3(m + 3);
      ior = m[6], invIor = m[7], reflectivity = m[8];
      s.coneAngle = m[9];
      inside = backfacing;
    } else {
      const SphereRec &r = tab.sph[k.idx];
      d3 n = normalised(s.pos - ld3(r.centre));
      inside = dot(n, d) > 0;
      if (inside) n = -n;
      s.normal = n;
      s.basis = basisFromZ(n);
      s.emission = ld3(r.emission);
      s.diffuse = ld3(r.diffuse);
      s.coneAngle = r.coneAngle;
      ior = r.ior, invIor = r.invIor, reflectivity = r.reflectivity;
    }
    s.iorFrom = inside ? ior : 1.0;
    s.iorTo = inside ? 1.0 : ior;
    s.iorRatio = inside ? ior : invIor;
    s.matReflectivity = reflectivity;
    s.reflectivity = eager ? resolveReflectivity(s, d) : 0.0;
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(s.reflectivity), "+v"(s.pos.x), "+v"(s.basis.y.z), "+v"(s.diffuse.x));
#endif
    PTW_T(tB);
    PTW_ACC(2, tA, tB);
    return s;
  }
};

// Bytes of dynamic LDS traceSequential needs (also computed on the host for the launch).
__host__ __device__ inline size_t seqLdsBytes(int waves, int maxDepth, bool ldsTables, uint32_t ntri,
                                              uint32_t nmat, uint32_t nsph, int masters = 1, bool pair = false) {
  size_t n = masters * sizeof(SeqShared);
  n += static_cast<size_t>(waves) * (maxDepth > 0 ? maxDepth : 1) * sizeof(Level);
  n = (n + 15) & ~static_cast<size_t>(15); // the answers: 16-byte aligned (ds_read_b128)
  // (two answer sets per master: the paired form's second command slot; 96 bytes more for the plain one)
  n += 2 * masters * static_cast<size_t>(waves) * sizeof(PartialHit) + kSeqCmdBytes;
  n = (n + 63) & ~static_cast<size_t>(63);
  if (waves > 1) n += kSeqCamBytes; // the camera (worker-wave kernels: see traceSequential)
  if (pair) n += masters * kSeqFanBytes;
  if (ldsTables) {
    n += static_cast<size_t>(nsph) * sizeof(SphereRec);
    n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
    n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  }
  return n;
}

template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, int MASTERS = 1, bool PAIR = false, bool PICKS = true>
__global__ __launch_bounds__(WAVES == 1 ? 64 : 64 * (WAVES + MASTERS)) void traceSequential(
    const TraceParams p, const double *__restrict__ triGeom,
    const TriShade *__restrict__ triShade, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, uint32_t *__restrict__ mtPos, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters, uint32_t *__restrict__ picks) {
  using Ctx = SeqCtx<SLOTS, WAVES, LDS_TABLES, REG, false, MASTERS, PAIR, PICKS>;
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  (void)triShade;
  const int depthSlots = p.maxDepth > 0 ? p.maxDepth : 1;
  // the master wave(s): wave m < MASTERS runs pass MASTERS * blockIdx.x + m on its own generator
  const int hwWave = static_cast<int>(threadIdx.x >> 6);
  const int master = MASTERS == 1 ? 0 : (hwWave < MASTERS ? hwWave : 0);
  SeqShared &sh = reinterpret_cast<SeqShared *>(ldsRaw)[master];
  Level *stacks = reinterpret_cast<Level *>(ldsRaw + MASTERS * sizeof(SeqShared));
  PartialHit *partials = reinterpret_cast<PartialHit *>(
      ldsRaw + ((MASTERS * sizeof(SeqShared) + static_cast<size_t>(WAVES) * (p.maxDepth > 0 ? p.maxDepth : 1) * sizeof(Level) + 15) &
                ~static_cast<size_t>(15)));
  // (the answers are read with ds_read_b128: their offset is rounded up to 16 bytes - with seven
  // worker waves and an odd maxDepth the stacks end on 8 mod 16)
  const size_t partialsOff = (MASTERS * sizeof(SeqShared) + static_cast<size_t>(WAVES) * depthSlots * sizeof(Level) + 15) &
                             ~static_cast<size_t>(15);
  size_t off = partialsOff + 2 * MASTERS * static_cast<size_t>(WAVES) * sizeof(PartialHit) + kSeqCmdBytes;
  off = (off + 63) & ~static_cast<size_t>(63);
  ptw_camera *camLds = reinterpret_cast<ptw_camera *>(ldsRaw + off);
  if (WAVES > 1) off += kSeqCamBytes;
  unsigned char *fanArea = ldsRaw + off;
  if (PAIR) off += MASTERS * kSeqFanBytes;
  (void)fanArea;

  const int pass = blockIdx.x * MASTERS + master;
  const bool hasPass = MASTERS == 1 || static_cast<uint32_t>(pass) < p.npass; // (odd pass count)
  Ctx ctx;
  ctx.triCompactGlobal = triCompact;
  ctx.matTableGlobal = matTable;
  ctx.p = &p;
  ctx.envColour = ld3(p.env);
  asm volatile("" : "+v"(ctx.envColour.x), "+v"(ctx.envColour.y), "+v"(ctx.envColour.z));
  ctx.triGeom = triGeom;
  ctx.spheresGlobal = spheres;
  ctx.sh = &sh;
  constexpr int kBlock = Ctx::kBlock;
  const bool isWorker = WAVES > 1 && hwWave >= MASTERS;
  const int lane = threadIdx.x & 63;
  // Worker waves in `tid` order: those on a SIMD of their own pair first, those that share a SIMD
  // with a master wave (waves go to the four SIMDs round robin: wave 4 sits with wave 0, wave 5
  // with wave 1) last - they get the scene's empty slots (SeqCtx::slotTriangle), because the master
  // beside them uses the search time for its look-ahead.
  int workerRank = hwWave - MASTERS;
  if (WAVES > 1 && isWorker) {
    const int firstShared = 4 - MASTERS, nShared = MASTERS; // worker indices of waves 4 .. 3 + MASTERS
    if (workerRank >= firstShared + nShared) workerRank -= nShared;
    else if (workerRank >= firstShared) workerRank += WAVES - firstShared - nShared;
  }
  ctx.tid = WAVES == 1 ? threadIdx.x : (isWorker ? workerRank * 64 + lane : lane);
  ctx.unitBase = 0, ctx.myUnits = 0;
  if (WAVES > 1 && isWorker) {
    // ranks [0, nA): the workers that share a SIMD with another worker; [nA, WAVES): beside a master
    constexpr int nA = WAVES - Ctx::kSideB;
    const bool sideB = workerRank >= nA;
    // ranks [0, nA / 2): the OLDER wave of each worker pair (lower hardware wave index), [nA / 2, nA):
    // the younger one
    static_assert(nA % 2 == 0, "the worker-only SIMDs carry two workers each");
    const bool young = !sideB && workerRank >= nA / 2;
    ctx.myUnits = __builtin_amdgcn_readfirstlane(sideB ? p.seqUnitsB : (young ? p.seqUnitsY : p.seqUnitsA));
    ctx.unitBase = __builtin_amdgcn_readfirstlane(
        sideB   ? (nA / 2) * (p.seqUnitsA + p.seqUnitsY) + (workerRank - nA) * p.seqUnitsB
        : young ? (nA / 2) * p.seqUnitsA + (workerRank - nA / 2) * p.seqUnitsY
                : workerRank * p.seqUnitsA);
  }
  // only the master waves use a radiance stack: one each - two with PAIR, one per chain (the stacks
  // of the worker waves' indices serve: WAVES >= 2 MASTERS)
  static_assert(!PAIR || WAVES >= 2 * MASTERS, "a stack per chain");
  ctx.stack = stacks + master * (PAIR ? 2 : 1) * depthSlots;
#if PTW_EXPERIMENTS
  if constexpr (PAIR) ctx.pairInit(p, fanArea + master * kSeqFanBytes, depthSlots, !isWorker);
#endif
  // [MASTERS][2 slots][WAVES] partial results (the plain form uses [MASTERS][WAVES] of them), then the
  // commands (128 B each) and the worker waves' atomic slots
  ctx.allCmds = reinterpret_cast<SeqCommand *>(partials + 2 * MASTERS * WAVES);
  ctx.partials = isWorker ? partials : partials + master * (PAIR ? 2 : 1) * WAVES;
  ctx.cmd = ctx.allCmds + master;
  ctx.minSlot = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(ctx.allCmds) + kSeqMinSlotOffset) +
                2 * (isWorker ? workerRank : 0);
  ctx.masterIndex = master;
  ctx.picksOn = PICKS && picks != nullptr;
  ctx.pickReset();
  static_assert(2 * sizeof(SeqCommand) <= kSeqMinSlotOffset && WAVES <= 8 &&
                    kSeqMinSlotOffset + 8 * 2 * sizeof(unsigned long long) <= kSeqCmdBytes, "commands and slots fit");
  ctx.tick = 0;
  ctx.laArmed = false;
  ctx.laPos = -1;
  ctx.laMisses = 0;
  ctx.laTick = 0;
  ctx.pendKind = 0;
  ctx.words = 0;
  ctx.rays = 0;
  ctx.parity = 0;
  if (LDS_TABLES) {
    SphereRec *ls = reinterpret_cast<SphereRec *>(ldsRaw + off);
    double *lt = reinterpret_cast<double *>(ls + p.nsph);
    double *lm = lt + static_cast<size_t>(p.ntri) * kTriCompactDoubles;
    const double *gs = reinterpret_cast<const double *>(spheres);
    double *lsd = reinterpret_cast<double *>(ls);
    for (uint32_t i = threadIdx.x; i < p.nsph * (sizeof(SphereRec) / 8); i += kBlock) lsd[i] = gs[i];
    for (uint32_t i = threadIdx.x; i < p.ntri * kTriCompactDoubles; i += kBlock) lt[i] = triCompact[i];
    for (uint32_t i = threadIdx.x; i < p.nmat * kMatDoubles; i += kBlock) lm[i] = matTable[i];
    ctx.tab.sph = ls;
    ctx.tab.tri = lt;
    ctx.tab.mat = lm;
  } else {
    ctx.tab.sph = spheres;
    ctx.tab.tri = triCompact;
    ctx.tab.mat = matTable;
  }
  if (WAVES == 1 || isWorker) {
    ctx.loadPrimitives();
  } else {
    ctx.hasSphere = false;
  }
  ctx.cam = &p.cam;
  if (WAVES > 1) {
    if (threadIdx.x < sizeof(ptw_camera) / sizeof(double))
      reinterpret_cast<double *>(camLds)[threadIdx.x] = reinterpret_cast<const double *>(&p.cam)[threadIdx.x];
    ctx.cam = camLds; // (visible after the barrier below)
  }

  // resume this pass's generator
  uint32_t *myState = mtState + static_cast<size_t>(hasPass ? pass : 0) * kMtWords;
  if (MASTERS == 1) {
    for (int i = threadIdx.x; i < kMtWords; i += kBlock) sh.mt[i] = myState[i];
    ctx.pos = __builtin_amdgcn_readfirstlane(static_cast<int>(mtPos[pass]));
    __syncthreads();
    if (ctx.pos < kMtDoubles) ctx.rebuildCanon();
  } else {
    ctx.pos = kMtDoubles;
    if (!isWorker) {
      if (hasPass) {
        for (int i = lane; i < kMtWords; i += 64) sh.mt[i] = myState[i];
        ctx.pos = __builtin_amdgcn_readfirstlane(static_cast<int>(mtPos[pass]));
        waveSync();
        if (ctx.pos < kMtDoubles) ctx.rebuildCanonWave();
      }
      // "live" / "no rays as of barrier 0"
      if (lane == 0) ctx.cmd->op = hasPass ? kCmdLive : 0u, ctx.cmd->nrays = 1u;
    }
    __syncthreads();
  }

  if (isWorker) {
    ctx.workerLoop();
  } else if (!hasPass) {
    ctx.stopWorkers();
  } else {
  if (MASTERS == 2 && !PAIR && master == 1) {
    ldsBarrier(); // the second master runs one barrier behind the first
    ctx.tick = 1;
  }
  // (raising the master waves' priority over the worker that shares their SIMD - s_setprio 1..3 -
  // measured no difference on suzanne and ce: profiles/r03e_master_priority_and_balance.txt)
  const int w = p.width;
  const bool lens = WAVES > 1 ? uniformBool(ctx.cam->aperture_radius != 0) : p.cam.aperture_radius != 0;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
#if PTW_PROFILE_PHASES
  for (int i = 0; i < 12; ++i) ctx.prof[i] = 0;
  for (int i = 0; i < 6; ++i) ctx.mprof[i] = 0;
  ctx.g00 = ctx.g01 = ctx.g10 = ctx.g11 = ctx.g20 = ctx.g21 = 0;
  ctx.n00 = ctx.n01 = ctx.n10 = ctx.n11 = ctx.n20 = ctx.n21 = 0;
  ctx.lastExit = 0, ctx.rayKind = 2, ctx.lastKind = 0, ctx.lastMiss = 0;
  ctx.pairReq = ctx.pairReq2 = ctx.fanBuilds = ctx.fanHits = ctx.fanMisses = 0;
  const unsigned long long tStart = __builtin_amdgcn_s_memtime();
#endif
#if PTW_EXPERIMENTS
  if constexpr (PAIR) ctx.pairRun(p, myStage, words, PICKS ? picks : nullptr, pass); // (the whole pass, and the cadence after it)
#endif
  if constexpr (!PAIR) {
  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    const int px = static_cast<int>(pix % static_cast<uint32_t>(w));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(w));
    ctx.words = 0;
    ctx.pickReset();
    const unsigned long long tC0 = ctx.now();
    double r0, r1, r2 = 0, r3 = 0;
    if (lens) {
      ctx.draw4(r0, r1, r2, r3);
    } else {
      r0 = ctx.draw();
      r1 = ctx.draw();
    }
    d3 o, d;
    cameraRay<MASTERS == 2>(*ctx.cam, px, py, r0, r1, r2, r3, o, d);
    ctx.acc(10, tC0, d.x);
    const d3 L = radiance0(ctx, p, triShade, spheres, o, d);
    if (lane == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = ctx.words;
      if (PICKS && picks) picks[static_cast<size_t>(pass) * p.npix + pix] = ctx.pickS2;
    }
  }

#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const unsigned long long tEnd = __builtin_amdgcn_s_memtime();
    const double r = static_cast<double>(ctx.rays);
    printf("PHASES rays=%llu total/ray=%.0f tests=%.0f reduce=%.0f surface=%.0f (lds1=%.0f) scatter|arm=%.0f "
           "xwave=%.0f first=%.0f fold|accum=%.0f lastE=%.0f regen=%.0f camera=%.0f hotlevel=%.0f other=%.0f\n",
           ctx.rays, (tEnd - tStart) / r, ctx.prof[0] / r, ctx.prof[1] / r, ctx.prof[2] / r,
           ctx.prof[4] / r, ctx.prof[3] / r, ctx.prof[5] / r, ctx.prof[6] / r, ctx.prof[7] / r,
           ctx.prof[8] / r, ctx.prof[9] / r, ctx.prof[10] / r, ctx.prof[11] / r,
           ((tEnd - tStart) - ctx.prof[0] - ctx.prof[1] - ctx.prof[2] - ctx.prof[3] - ctx.prof[5] -
            ctx.prof[6] - ctx.prof[7] - ctx.prof[8] - ctx.prof[10] - ctx.prof[11]) / r);
    if (WAVES > 1 && !PAIR)
      printf("MASTER per ray, inside intersect(): publish=%.0f waitB1=%.0f shadow(flush+lookahead)=%.0f waitB2=%.0f pick=%.0f; "
             "outside intersect()=%.0f\n",
             ctx.mprof[0] / r, ctx.mprof[1] / r, ctx.mprof[2] / r, ctx.mprof[3] / r, ctx.mprof[4] / r,
             ((tEnd - tStart) - ctx.prof[5]) / r);
    if (WAVES > 1 && !PAIR) {
      const double px = static_cast<double>(p.pixCount);
      auto avg = [](unsigned long long sum, unsigned long long n) { return n ? static_cast<double>(sum) / n : 0.0; };
      printf("MASTER cycles from an answer to the next published ray (and answers per sample): primary hit %.0f (%.2f) miss %.0f "
             "(%.2f) | first ray of a sub-sample hit %.0f (%.2f) miss %.0f (%.2f) | deeper hit %.0f (%.2f) miss %.0f (%.2f)\n",
             avg(ctx.g00, ctx.n00), ctx.n00 / px, avg(ctx.g01, ctx.n01), ctx.n01 / px, avg(ctx.g10, ctx.n10), ctx.n10 / px,
             avg(ctx.g11, ctx.n11), ctx.n11 / px, avg(ctx.g20, ctx.n20), ctx.n20 / px, avg(ctx.g21, ctx.n21), ctx.n21 / px);
    }
  }
#endif
  ctx.stopWorkers();
  } // !PAIR
  } // master
  // park the generator for the next band
  if (MASTERS == 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = sh.mt[i];
    if (threadIdx.x == 0) {
      mtPos[pass] = static_cast<uint32_t>(ctx.pos); // thread 0 belongs to the master wave
      if (rayCounters) rayCounters[pass] += ctx.rays;
    }
  } else if (!isWorker && hasPass) {
    waveSync();
    for (int i = lane; i < kMtWords; i += 64) myState[i] = sh.mt[i];
    if (lane == 0) {
      mtPos[pass] = static_cast<uint32_t>(ctx.pos);
      if (rayCounters) rayCounters[pass] += ctx.rays;
    }
  }
}

// -----------------------------------------------------------------------------------------
// traceSequentialSpec: SEQUENTIAL policy for scenes of at most 64 triangles, with the
// sub-samples of the first-bounce fan-out traced SPECULATIVELY in parallel.
//
// The stream makes everything serial: sub-sample j+1 starts where sub-sample j stopped, and how
// many draws j consumes (3 per level it reaches) is known only when it is done.  But that count
// takes few values, and the values repeat (a closed scene mostly runs every path to the depth cap,
// an open one mostly loses the first ray).  So a workgroup of kSpecWaves waves - one per SIMD of
// a CU, each holding the whole scene in registers like the single-wave kernel - works per round
// on:  wave 0: sub-sample j at the true stream position (the frontier);
//      wave 1: sub-sample j+1, assuming j consumes m1 (the most recent count);
//      wave 2: sub-sample j+1 assuming m2 (the most recent different count) - or, while no second
//              value has been seen, sub-sample j+3 assuming m1 three times;
//      wave 3: sub-sample j+2, assuming m1 twice.
// After a barrier every wave reads all results and commits, in order, as many sub-samples as the
// assumptions allow (always j; j+1 if a wave started where j really stopped; and so on).  Wrong
// guesses cost nothing but the energy: the result is the one the serial order defines, bit for bit
// - each wave accumulates the committed contributions itself, in sub-sample order.
//
// For that the stream must be readable ahead of the frontier: the generator output sits in a
// ring of two blocks (SeqCtx SPEC mode); while the frontier is in one block the next one is
// already there, and when the frontier crosses into it, wave 0 generates the block after it into
// the slot that just became free.
// -----------------------------------------------------------------------------------------
constexpr int kSpecWaves = 4;

struct alignas(16) SpecResult { // one per wave and round parity, in LDS
  double L[3]; // radiance of the sub-path below the first-bounce surface
  int meta;    // canonical doubles consumed | lobe at the first-bounce surface << 8 | rays << 16
  int pad;
};

__host__ __device__ inline size_t specLdsBytes(uint32_t ntri, uint32_t nmat, uint32_t nsph) {
  size_t n = 2 * kRingStride;                          // the ring
  n += kMtWords * sizeof(uint32_t);                    // raw generator state
  n += 2 * kSpecWaves * sizeof(SpecResult);            // results, double-buffered
  n += 64 + kSeqCamBytes;                              // generator commands, the camera
  n = (n + 63) & ~static_cast<size_t>(63);
  n += static_cast<size_t>(nsph) * sizeof(SphereRec);
  n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
  n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  // more than half of a CU's 160 KB: one workgroup per CU, so its four waves get a SIMD each
  const size_t floor = 84 * 1024;
  return n < floor ? floor : n;
}

template <bool PICKS>
__global__ __launch_bounds__(64 * (kSpecWaves + 1)) void traceSequentialSpec(
    const TraceParams p, const double *__restrict__ triGeom, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, double *__restrict__ specState, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters, uint32_t *__restrict__ picks) {
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  constexpr int kBlock = 64 * (kSpecWaves + 1);
  char *ring = reinterpret_cast<char *>(ldsRaw);
  uint32_t *mt = reinterpret_cast<uint32_t *>(ldsRaw + 2 * kRingStride);
  SpecResult *results = reinterpret_cast<SpecResult *>(mt + kMtWords);
  // (taken from ldsRaw inside the lambdas too: a captured pointer loses its LDS address space and
  // the stores turn into flat instructions with a vmcnt wait)
  constexpr size_t kGenCmdOffset = 2 * kRingStride + kMtWords * sizeof(uint32_t) + 2 * kSpecWaves * sizeof(SpecResult);
  uint32_t *genCmd = reinterpret_cast<uint32_t *>(ldsRaw + kGenCmdOffset);
  size_t off = 2 * kRingStride + kMtWords * sizeof(uint32_t) + 2 * kSpecWaves * sizeof(SpecResult) + 64;
  // (the camera in LDS: as part of the kernel argument its 36 dwords were spilled to vector-register lanes
  // and read back for every pixel - see kSeqCamBytes)
  ptw_camera *camLds = reinterpret_cast<ptw_camera *>(ldsRaw + off);
  off += kSeqCamBytes;
  off = (off + 63) & ~static_cast<size_t>(63);
  if (threadIdx.x < sizeof(ptw_camera) / sizeof(double))
    reinterpret_cast<double *>(camLds)[threadIdx.x] = reinterpret_cast<const double *>(&p.cam)[threadIdx.x];

  const int pass = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;

  using Ctx = SeqCtx<1, 1, true, true, true, 1, false, PICKS>;
  Ctx ctx;
  ctx.triCompactGlobal = triCompact;
  ctx.matTableGlobal = matTable;
  ctx.p = &p;
  ctx.envColour = ld3(p.env);
  asm volatile("" : "+v"(ctx.envColour.x), "+v"(ctx.envColour.y), "+v"(ctx.envColour.z));
  ctx.triGeom = triGeom;
  ctx.spheresGlobal = spheres;
  ctx.sh = nullptr;
  ctx.tid = lane; // every wave owns the whole scene: lane k holds triangle k
  ctx.stack = nullptr;
  ctx.partials = nullptr;
  ctx.cmd = nullptr;
  ctx.words = 0;
  ctx.rays = 0;
  ctx.parity = 0;
  ctx.picksOn = PICKS && picks != nullptr;
  ctx.pickReset();
  ctx.ringBase = ring;
  {
    SphereRec *ls = reinterpret_cast<SphereRec *>(ldsRaw + off);
    double *lt = reinterpret_cast<double *>(ls + p.nsph);
    double *lm = lt + static_cast<size_t>(p.ntri) * kTriCompactDoubles;
    const double *gs = reinterpret_cast<const double *>(spheres);
    double *lsd = reinterpret_cast<double *>(ls);
    for (uint32_t i = threadIdx.x; i < p.nsph * (sizeof(SphereRec) / 8); i += kBlock) lsd[i] = gs[i];
    for (uint32_t i = threadIdx.x; i < p.ntri * kTriCompactDoubles; i += kBlock) lt[i] = triCompact[i];
    for (uint32_t i = threadIdx.x; i < p.nmat * kMatDoubles; i += kBlock) lm[i] = matTable[i];
    ctx.tab.sph = ls;
    ctx.tab.tri = lt;
    ctx.tab.mat = lm;
  }
  const bool isGenerator = wave == kSpecWaves; // the fifth wave only produces the stream
  if (!isGenerator) ctx.loadPrimitives();

  // ---- the stream: resume (or start) this pass's generator ring ----
  uint32_t *myState = mtState + static_cast<size_t>(pass) * kMtWords;
  double *myPark = specState + static_cast<size_t>(pass) * kSpecStateDoubles;
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) mt[i] = myState[i];
  unsigned fOff = 0; // frontier: ring slot (0 or kRingStride) ...
  int fQ = 0;        // ... and position in it
  if (p.firstBand) {
    __syncthreads();
    if (isGenerator) {
      specGenerateBlock(mt, ring, 0, lane);           // block 0
      specGenerateBlock(mt, ring, kRingStride, lane); // block 1 (completes block 0's overlap)
    }
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

  // ---- the generator wave: serves one command per workgroup barrier until told to exit ----
  if (isGenerator) {
    for (unsigned k = 0;; ++k) {
      ldsBarrier();
      const uint32_t cmd = genCmd[k & 1];
      if (cmd == kGenExit) break;
      if (cmd == kGenSlot0) specGenerateBlock(mt, ring, 0, lane);
      if (cmd == kGenSlot1) specGenerateBlock(mt, ring, kRingStride, lane);
    }
  } else {
  // Stream bookkeeping of the tracing waves (identical in all of them).  When the frontier
  // enters the other slot, the slot it left is handed to the generator wave with the next
  // barrier (genState 1 -> 2); the block is complete once the barrier after that has been passed
  // (2 -> 0), because the generator arrives there only when it is done.  The tracing waves
  // read at most `ahead` draws beyond the frontier, so they only have to wait for an
  // outstanding block when the frontier comes that close to the end of its slot.
  unsigned barriers = 0;
  int genState = 0;
  unsigned genSlot = 0;
  const int ahead = 12 * (p.maxDepth > 0 ? p.maxDepth : 1) + 8;
  auto roundBarrier = [&](uint32_t exitCmd) {
    if (threadIdx.x == 0)
      reinterpret_cast<uint32_t *>(ldsRaw + kGenCmdOffset)[barriers & 1] =
          exitCmd ? exitCmd : (genState == 1 ? (genSlot ? kGenSlot1 : kGenSlot0) : kGenNone);
    ldsBarrier();
    ++barriers;
    genState = genState == 1 ? 2 : 0;
  };
  auto ensureAhead = [&]() {
    while (genState != 0 && fQ + ahead >= kMtDoubles) roundBarrier(0);
  };
  auto advanceFrontier = [&](int n) { // n < kMtDoubles
    const int np = fQ + n;
    if (np >= kMtDoubles) {
      genSlot = fOff; // the slot left behind takes the block after the next
      genState = 1;
      fQ = np - kMtDoubles;
      fOff ^= kRingStride;
    } else {
      fQ = np;
    }
  };

  const int w = p.width;
  const bool lens = uniformBool(camLds->aperture_radius != 0);
  const int nSub = p.fbU * p.fbV;
  const int vShift = p.fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(p.fbV)) : 0;
  // Per-round constants in vector registers: as kernel arguments they sit in a 16-register
  // scalar tuple that does not survive the rounds and would be re-read from its spill lanes
  // (eighteen v_readlane) for every sub-sample.
  double invU = p.invU, invV = p.invV;
  asm volatile("" : "+v"(invU), "+v"(invV));
  const bool fastFan = (p.uPow2 & p.vPow2) != 0;
  const int vMask = p.fbV - 1;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
  unsigned long long raysTotal = 0;
  // The guesses: m1 = the most frequent count of draws a sub-sample has consumed so far in this
  // pass (3 per level reached), m2 = the second most frequent (m2 == m1 while only one value has
  // been seen).  `hist` counts them in 6-bit fields, halved when a field passes 31.
  int m1 = 3 * (p.maxDepth > 0 ? p.maxDepth : 1), m2 = m1;
  unsigned long long hist = 0;
  int parity = 0;
#if PTW_PROFILE_PHASES
  unsigned long long stRounds = 0, stCommits = 0, stWork = 0, stWait = 0, stCommit = 0, stPrimary = 0;
  unsigned long long stOk1 = 0, stOk2a = 0, stOk3 = 0, stOk2b = 0, stIdle = 0;
  const unsigned long long stT0 = __builtin_amdgcn_s_memtime();
#endif

  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    const int px = static_cast<int>(pix % static_cast<uint32_t>(w));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(w));
    // ---- every wave: camera ray and first hit at the frontier (redundant, in parallel) ----
    PTW_T(tP0);
    ensureAhead();
    ctx.setStream(fOff, fQ);
    double r0, r1, r2 = 0, r3 = 0;
    if (lens) {
      ctx.draw4(r0, r1, r2, r3);
    } else {
      r0 = ctx.draw();
      r1 = ctx.draw();
    }
    const int camDraws = lens ? 4 : 2;
    d3 o, d;
    cameraRay(*camLds, px, py, r0, r1, r2, r3, o, d);
    int sampleDraws = camDraws;
    ctx.pickReset();
    uint32_t pickSum = 0, pickBase = 1; // the sample's pick checksum; intersect() calls committed so far
    d3 L = mk(0, 0, 0);
    bool traced = false;
    HitKey k0;
    k0.t = kInf, k0.idx = kMiss, k0.det = 0;
    if (p.maxDepth > 0) {
      k0 = ctx.intersect(o, d);
      raysTotal++;
      if (PICKS) pickSum = ctx.pickS2; // (the primary ray is call 0)
      if (uniformBool(k0.idx == kMiss)) {
        L = ld3(p.env);
      } else {
        traced = true;
      }
    }
    advanceFrontier(camDraws);
#if PTW_PROFILE_PHASES
    stPrimary += __builtin_amdgcn_s_memtime() - tP0;
#endif
    if (traced) {
      const Surface first = ctx.surfaceAt(k0, o, d);
      if (p.preview) {
        L = first.diffuse; // Scene.cpp:137-138
      } else {
        d3 result = mk(0, 0, 0);
        int j = 0;
        if (hist != 0) { // refresh the guesses once per sample
          if (hist & 0x0820820820820820ull) hist = (hist >> 1) & 0x07df7df7df7df7dfull;
          int best = 0, bestN = -1, second = 0, secondN = 0;
#pragma unroll
          for (int f = 1; f <= 9; ++f) {
            const int n = static_cast<int>(hist >> (6 * f)) & 63;
            const bool top = n > bestN;
            const bool sec = !top & (n > secondN);
            second = top ? best : (sec ? f : second);
            secondN = top ? bestN : (sec ? n : secondN);
            best = top ? f : best;
            bestN = top ? n : bestN;
          }
          m1 = 3 * best;
          m2 = secondN > 0 ? 3 * second : m1;
        }
        while (j < nSub) {
          ensureAhead();
          // ---- this wave's assignment: sub-sample j + ioff, stream position frontier + delta ----
          const bool oneMode = m2 == m1;
          int ioff = 0, delta = 0;
          if (wave == 1) ioff = 1, delta = m1;
          if (wave == 2) ioff = oneMode ? 3 : 1, delta = oneMode ? 3 * m1 : m2;
          if (wave == 3) ioff = 2, delta = 2 * m1;
          const int myIdx = j + ioff;
          PTW_T(tW0);
          SpecResult mine;
          mine.L[0] = mine.L[1] = mine.L[2] = 0;
          mine.meta = 0, mine.pad = 0;
          if (myIdx < nSub) {
            const int np = fQ + delta; // delta < kMtDoubles
            const bool wrap = np >= kMtDoubles;
            ctx.setStream(wrap ? fOff ^ kRingStride : fOff, wrap ? np - kMtDoubles : np);
            ctx.words = 0;
            ctx.rays = 0;
            ctx.pickReset();
            // sub-sample index -> stratum (uS, vS) -> stratified (u, v); ONE decision for the
            // usual power-of-two fan-outs (shift / mask / multiply), the general case apart
            double xu, xv, pd;
            ctx.draw3(xu, xv, pd);
            double u, v;
            if (fastFan) {
              const int uS = myIdx >> vShift, vS = myIdx & vMask;
              u = (static_cast<double>(uS) + xu) * invU;
              v = (static_cast<double>(vS) + xv) * invV;
            } else {
              const int uS = myIdx / p.fbV, vS = myIdx - uS * p.fbV;
              const double ur = static_cast<double>(uS) + xu, vr = static_cast<double>(vS) + xv;
              u = p.uPow2 ? ur * invU : ur / static_cast<double>(p.fbU);
              v = p.vPow2 ? vr * invV : vr / static_cast<double>(p.fbV);
            }
            d3 nd;
            const bool refl = scatter(ctx, first, d, u, v, pd, nd);
            const d3 child = ctx.chainHot(p, first.pos, nd);
            mine.L[0] = child.x, mine.L[1] = child.y, mine.L[2] = child.z;
            mine.meta = static_cast<int>(ctx.words >> 1) | (refl ? 0x100 : 0) |
                        (static_cast<int>(ctx.rays) << 16);
            if (PICKS) mine.pad = static_cast<int>(ctx.pickS1 | (ctx.pickS2 << 16)); // (<= 9 calls of <= 127 primitives)
          }
          SpecResult *slot = results + parity * kSpecWaves;
          if (lane == 0) slot[wave] = mine;
          PTW_T(tW1);
          roundBarrier(0);
          PTW_T(tW2);
          // ---- commit (the scalar part identical in every wave) ----
          const int metaV = slot[lane & 3].meta;
          const int meta0 = __builtin_amdgcn_readlane(metaV, 0), meta1 = __builtin_amdgcn_readlane(metaV, 1);
          const int meta2 = __builtin_amdgcn_readlane(metaV, 2), meta3 = __builtin_amdgcn_readlane(metaV, 3);
          const int c0 = meta0 & 0xff, c1 = meta1 & 0xff, c2 = meta2 & 0xff, c3 = meta3 & 0xff;
          // The assignments this round was made with, and which of them held - as all-ones /
          // zero integer masks in scalar registers (conditions kept as C++ bools become lane masks
          // that take a trip through a vector register per use).
          auto eq = [](int a, int b) { return ((a ^ b) - 1) >> 31; };  // a, b >= 0: -1 if equal
          auto lt = [](int a, int b) { return (a - b) >> 31; };        // -1 if a < b
          const int d1 = m1, d2 = oneMode ? 3 * m1 : m2, d3v = 2 * m1;
          const int w2Second = oneMode ? 0 : -1; // wave 2 ran sub-sample j+1 (else j+3)
          const int ok1 = lt(j + 1, nSub) & eq(d1, c0);
          const int ok2a = lt(j + 1, nSub) & ~ok1 & w2Second & eq(d2, c0);
          const int cur1 = c0 + (c1 & ok1) + (c2 & ok2a);
          const int two = ok1 | ok2a;
          const int ok3 = two & lt(j + 2, nSub) & eq(d3v, cur1);
          const int cur2 = cur1 + (c3 & ok3);
          const int ok2b = ok3 & lt(j + 3, nSub) & ~w2Second & eq(d2, cur2);
          const int cur = cur2 + (c2 & ok2b);
          const int nIdx = 1 - two - ok3 - ok2b;
          // histogram of the committed counts (6-bit fields indexed by count / 3)
          auto note = [&](int on, int c) {
            hist += (1ull << (6 * ((c * 11) >> 5))) & static_cast<unsigned long long>(static_cast<long long>(on));
          };
          note(-1, c0);
          note(ok1, c1);
          note(ok2a, c2);
          note(ok3, c3);
          note(ok2b, c2);
          raysTotal += static_cast<unsigned>(meta0 >> 16) + (static_cast<unsigned>(meta1 >> 16) & ok1) +
                       (static_cast<unsigned>(meta2 >> 16) & (ok2a | ok2b)) +
                       (static_cast<unsigned>(meta3 >> 16) & ok3);
          if (PICKS && picks && wave == 0) { // the committed sub-samples' picks, in sub-sample order
            auto addPicks = [&](int wv, int meta) {
              const uint32_t pw = static_cast<uint32_t>(slot[wv].pad);
              pickSum += pickBase * (pw & 0xffffu) + (pw >> 16);
              pickBase += static_cast<uint32_t>(meta) >> 16;
            };
            addPicks(0, meta0);
            if (ok1) addPicks(1, meta1);
            if (ok2a) addPicks(2, meta2);
            if (ok3) addPicks(3, meta3);
            if (ok2b) addPicks(2, meta2);
          }
          if (wave == 0) { // only the wave that stores the sample needs the radiance
            auto add = [&](int wv, int meta) {
              const SpecResult &r = slot[wv];
              const d3 child = mk(r.L[0], r.L[1], r.L[2]);
              result = result + ((meta & 0x100) ? first.emission + child
                                                : first.emission + first.diffuse * child);
            };
            add(0, meta0);
            if (ok1) add(1, meta1);
            if (ok2a) add(2, meta2);
            if (ok3) add(3, meta3);
            if (ok2b) add(2, meta2);
          }
          j += nIdx;
          sampleDraws += cur;
          parity ^= 1;
          advanceFrontier(cur);
#if PTW_PROFILE_PHASES
          stRounds++, stCommits += nIdx;
          stOk1 -= ok1, stOk2a -= ok2a, stOk3 -= ok3, stOk2b -= ok2b, stIdle += !(myIdx < nSub);
          stWork += tW1 - tW0, stWait += tW2 - tW1, stCommit += __builtin_amdgcn_s_memtime() - tW2;
#endif
        }
        L = result * p.invFirstBounce;
      }
    }
    if (threadIdx.x == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = 2u * static_cast<unsigned>(sampleDraws);
      if (PICKS && picks) picks[static_cast<size_t>(pass) * p.npix + pix] = pickSum;
    }
  }

  while (genState != 0) roundBarrier(0); // an outstanding block must be in the ring that gets parked
  roundBarrier(kGenExit);
#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const double n = static_cast<double>(p.pixCount);
    printf("SPEC wave %d: cycles/sample=%.0f rounds/sample=%.2f commits/round=%.2f primary=%.0f work=%.0f "
           "wait=%.0f commit+advance=%.0f (per sample)\n",
           wave, (__builtin_amdgcn_s_memtime() - stT0) / n, stRounds / n,
           static_cast<double>(stCommits) / stRounds, stPrimary / n, stWork / n, stWait / n, stCommit / n);
    printf("SPEC wave %d: per round ok1=%.3f ok2a=%.3f ok3=%.3f ok2b=%.3f idle=%.3f\n", wave,
           (double)stOk1 / stRounds, (double)stOk2a / stRounds, (double)stOk3 / stRounds,
           (double)stOk2b / stRounds, (double)stIdle / stRounds);
  }
#endif
  if (threadIdx.x == 0) {
    myPark[2 * kRingCanonDoubles] = fOff ? 1.0 : 0.0;
    myPark[2 * kRingCanonDoubles + 1] = static_cast<double>(fQ);
    if (rayCounters) rayCounters[pass] += raysTotal;
  }
  } // tracing waves
  // ---- park the stream for the next band ----
  __syncthreads();
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = mt[i];
  for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
    const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
    myPark[i] = reinterpret_cast<const double *>(ring + slot * kRingStride)[k];
  }
}

#if PTW_EXPERIMENTS
#include "experiments/ptw_gang.h"
#endif

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

template <bool BVH>
struct PixCtxT {
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

using PixCtx = PixCtxT<false>;

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
template <bool BVH>
__device__ __forceinline__ d3 radiance0Pix(PixCtxT<BVH> &ctx, const TraceParams &p, const TriShade *triShade,
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

template <bool BVH>
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

  PixCtxT<BVH> ctx;
  ctx.p = &p;
  ctx.triGeom = b.triGeom;
  ctx.triShade = b.triShade;
  ctx.spheres = b.spheres;
  ctx.bvhNodes = reinterpret_cast<const BvhNodeDev *>(b.bvhNodes);
  ctx.bvhLeafGeom = b.bvhLeafGeom;
  ctx.bvhLeafIndex = b.bvhLeafIndex;
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

__global__ __launch_bounds__(kPixBlock) __attribute__((amdgpu_waves_per_eu(PTW_PIX_WAVES, PTW_PIX_WAVES))) void tracePerPixel(
    const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x]
  perPixelSample<false>(p, b, pixStacks);
}

// ACCELERATED mode (ptw_render_params.accel == PTW_ACCEL_BVH): the same
// sample, with Scene::intersect culled by a BVH - bit-identical results, different work.  Reported
// separately, never in the headline numbers.
__global__ __launch_bounds__(kPixBlock) void tracePerPixelBvh(const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x] levels + [kBvhStack][blockDim.x] traversal
  perPixelSample<true>(p, b, pixStacks);
}

// -----------------------------------------------------------------------------------------
// PERPIXEL policy, persistent form: every lane owns a queue of (pass, pixel) samples and runs
// the path state machine; each trip of the outer loop traces ONE ray per lane against all
// primitives, then every lane advances its own path (shade, scatter, fold, start the next
// sample) until it again holds a ray to trace.  Lanes whose paths end early immediately pick
// up the next sample instead of idling until the slowest lane of the wave is done.
// Triangles are streamed with wave-uniform scalar loads, double-buffered one triangle ahead
// so the SMEM latency hides behind the ~45 VALU instructions of a Moller-Trumbore test.
// The kernel for open scenes and for small renders (launchTracePerPixel; capi_render.hip times it
// against the lock-step kernel once per scene); 128 VGPRs at 4 waves per SIMD.
// -----------------------------------------------------------------------------------------
constexpr int kPix2Block = 256;

template <int W, bool LDS_STATE>
__global__ __launch_bounds__(kPix2Block) __attribute__((amdgpu_waves_per_eu(W, W))) void tracePerPixelPersistent(
    const TraceParams p, const double *__restrict__ triGeom,
    const SphereRec *__restrict__ spheres, const double *__restrict__ triCompact,
    const double *__restrict__ matTable, double *__restrict__ stage, uint32_t *__restrict__ words,
    unsigned long long *__restrict__ rayCounters, unsigned long long *__restrict__ sampleQueue) {
  extern __shared__ uint32_t pixLevels[]; // [maxDepth][blockDim.x]: combined primitive index | reflective lobe << 31
  const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
  // Samples are handed out through one device-wide counter: a lane that finishes a sample takes
  // the next index (the compiler folds the lanes of a wave into one atomic).  A static
  // lane -> sample map would pin a lane to one pixel across passes, and pixels differ in cost
  // by 60x (background vs. many-bounce paths).
  uint64_t sample = atomicAdd(sampleQueue, 1ull);
  const uint32_t nsph = p.nsph, ntri = p.ntri;
  const int nSub = p.fbU * p.fbV;

  // ---- per-lane path state ----
  unsigned long long nrays = 0;
  d3 o = mk(0, 0, 0), d = mk(0, 0, 1);
  int depth = 0;      // depth of the ray currently held
  int sub = 0, nlev = 0;
  // What a lane keeps of the first-bounce surface while its fan-out runs, and the running sum of
  // the fan-out, are touched once per sub-sample.  LDS_STATE: they live in LDS ([9][blockDim.x]
  // doubles behind the level words) instead of taking 18 of the 128 registers for the whole kernel
  // - on Cornell the same speed with 70 instead of 550 B of HBM traffic per sample (the spills);
  // from 128 triangles on registers measured faster (ce 4.09 against 3.83 Msamples/s).
  double *fanState = reinterpret_cast<double *>(pixLevels + static_cast<size_t>(p.maxDepth > 1 ? p.maxDepth : 1) * kPix2Block) + threadIdx.x;
  double fanRegs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stD3 = [&](int k, d3 v) {
    if (LDS_STATE) {
      fanState[(k + 0) * kPix2Block] = v.x, fanState[(k + 1) * kPix2Block] = v.y, fanState[(k + 2) * kPix2Block] = v.z;
    } else {
      fanRegs[k] = v.x, fanRegs[k + 1] = v.y, fanRegs[k + 2] = v.z;
    }
  };
  auto ldD3 = [&](int k) {
    if (LDS_STATE)
      return mk(fanState[(k + 0) * kPix2Block], fanState[(k + 1) * kPix2Block], fanState[(k + 2) * kPix2Block]);
    return mk(fanRegs[k], fanRegs[k + 1], fanRegs[k + 2]);
  };
  constexpr int kFanPos = 0, kFanDir = 3, kFanSum = 6;
  uint32_t firstIdx = 0; // combined primitive index | back-facing << 31
  // The sample's generator (four words), its word count and its (pass, pixel) are touched when a
  // sample starts or ends and once per scatter: seven words that live next to the doubles.
  uint32_t *fanWords = reinterpret_cast<uint32_t *>(fanState - threadIdx.x + 9 * kPix2Block) + threadIdx.x;
  uint32_t wordRegs[7] = {0, 0, 0, 0, 0, 0, 0};
  constexpr int kWRng = 0, kWCount = 4, kWPass = 5, kWPix = 6;
  auto stW = [&](int k, uint32_t v) {
    if (LDS_STATE) fanWords[k * kPix2Block] = v;
    else wordRegs[k] = v;
  };
  auto ldW = [&](int k) -> uint32_t { return LDS_STATE ? fanWords[k * kPix2Block] : wordRegs[k]; };
  auto ldRng = [&]() {
    Sfc32 r;
    r.a = ldW(kWRng), r.b = ldW(kWRng + 1), r.c = ldW(kWRng + 2), r.counter = ldW(kWRng + 3);
    return r;
  };
  auto stRng = [&](const Sfc32 &r) { stW(kWRng, r.a), stW(kWRng + 1, r.b), stW(kWRng + 2, r.c), stW(kWRng + 3, r.counter); };
  bool active = sample < total;

  // Starts sample `sample`: seeds the stream, draws the camera ray.  Returns false when a
  // sample needs no tracing at all (maxDepth <= 0) - handled by the caller's loop.
  auto beginSample = [&]() {
    const uint32_t pass = static_cast<uint32_t>(sample / p.pixCount);
    const uint32_t pixIdx = static_cast<uint32_t>(sample % p.pixCount);
    const uint32_t pix = globalPixel(p, p.pixBegin + pixIdx);
    Sfc32 rng;
    rng.seed(p.passSeedBase + pass, pix);
    unsigned nwords = 0;
    auto draw = [&]() {
      const uint32_t w0 = rng.next();
      const uint32_t w1 = rng.next();
      nwords += 2;
      return canonicalFromWords(w0, w1);
    };
    const int px = static_cast<int>(pix % static_cast<uint32_t>(p.width));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(p.width));
    const double r0 = draw();
    const double r1 = draw();
    double r2 = 0, r3 = 0;
    if (p.cam.aperture_radius != 0) {
      r2 = draw();
      r3 = draw();
    }
    cameraRay<true>(p.cam, px, py, r0, r1, r2, r3, o, d);
    depth = 0;
    stRng(rng);
    stW(kWCount, nwords), stW(kWPass, pass), stW(kWPix, pixIdx);
  };
  auto finishSample = [&](d3 L) {
    const uint32_t pass = ldW(kWPass), pixIdx = ldW(kWPix), nwords = ldW(kWCount);
    double *out = stage + (static_cast<size_t>(pass) * p.pixCount + pixIdx) * 3;
    out[0] = L.x, out[1] = L.y, out[2] = L.z;
    if (words) words[static_cast<size_t>(pass) * p.npix + globalPixel(p, p.pixBegin + pixIdx)] = nwords;
    sample = atomicAdd(sampleQueue, 1ull);
    active = sample < total;
  };

  if (active) {
    beginSample();
    while (active && p.maxDepth <= 0) { // degenerate: radiance() returns 0 without tracing
      finishSample(mk(0, 0, 0));
      if (active) beginSample();
    }
  }

  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    // ------------------------------------------------------------------ trace one ray / lane
    HitKey key;
    key.t = kInf, key.idx = kMiss, key.det = 0;
    if (active) {
      nrays++;
      for (uint32_t i = 0; i < nsph; ++i) {
        const SphereRec &r = spheres[i];
        testSphere(o, d, ld3(r.centre), r.radiusSquared, i, key.t, key.idx);
      }
      if (ntri) {
        // One Moller-Trumbore test.  `prefetch` puts the next scalar loads in flight right after the
        // first use of this triangle's registers: SMEM returns out of order, so the only wait there
        // is is "all of them" - a load issued before that wait would be waited for at once.
        auto test = [&](const TriRegs &tr, uint32_t k, auto prefetch) {
          const d3 v0 = mk(tr.v[0], tr.v[1], tr.v[2]), e1 = mk(tr.v[3], tr.v[4], tr.v[5]), e2 = mk(tr.v[6], tr.v[7], tr.v[8]);
          const d3 pVec = cross(d, e2);
          const double det = dot(e1, pVec);
          __builtin_amdgcn_sched_barrier(0);
          prefetch();
          __builtin_amdgcn_sched_barrier(0);
          if (!(__builtin_fabs(det) < kEpsilon)) { // per-lane early-outs: coherent rays skip whole triangles
            const double invDet = rcp(det);
            const d3 tVec = o - v0;
            const double u = dot(tVec, pVec) * invDet;
            // u first: the reference rejects on (u < 0 | u > 1 | v < 0 | u + v > 1) as one fused test
            // (Scene.cpp:89); a triangle rejected on u is rejected whatever v is, so when no lane of
            // the wave passes the u test the wave skips qVec, v and t (18 of the test's 50 issue
            // slots) - same decisions, same values.
            if (!PTW_U_FIRST || !((u < 0.0) | (u > 1.0))) {
              const d3 qVec = cross(tVec, e1);
              const double v = dot(d, qVec) * invDet;
              if (PTW_U_FIRST ? !((v < 0.0) | (u + v > 1)) : !((u < 0.0) | (u > 1.0) | (v < 0.0) | (u + v > 1))) {
                const double t = dot(e2, qVec) * invDet;
                if (t > kEpsilon && t < key.t) {
                  key.t = t;
                  key.idx = nsph + k;
                  key.det = det;
                }
              }
            }
          }
        };
        // two triangles per trip, each in its own registers: no rotation copies (+3 % over one per
        // trip; rejection by select instead of branches measured -20 % on suzanne, +-0 on Cornell)
        TriRegs a = loadTriScalar(triGeom, 0), b;
        for (uint32_t k = 0; k < ntri; k += 2) {
          test(a, k, [&] { b = loadTriScalar(triGeom, k + 1 < ntri ? k + 1 : k); });
          // (odd count: the last triangle once more - it cannot beat itself, `<` is strict)
          test(b, k + 1, [&] { a = loadTriScalar(triGeom, k + 2 < ntri ? k + 2 : k); });
        }
      }
    }

    // ------------------------------------------------- advance this lane's path to its next ray
    // One copy of each piece of work, because the lanes of a wave are in different states and the
    // wave executes every piece some lane needs: ONE surface block and ONE scatter block serve the
    // first-bounce fan-out (stratified u, v) and the deeper levels, ONE finish + begin block every
    // way a sample can end.  Of the first-bounce surface a lane keeps only the primitive, the hit
    // point and the incoming direction: each sub-sample of the fan-out rebuilds the surface from
    // the tables (the block runs anyway, for the lanes that have just hit something), and its
    // colours are level 0 of the (E, T) stack, folded like every other level.
    if (active) {
      bool haveHit = true, capped = false, finished = false;
      d3 doneL = mk(0, 0, 0);
      uint32_t sIdx = 0;
      bool sBack = false;
      d3 sPos = o, sDin = d;
      for (;;) {
        d3 term = mk(0, 0, 0);
        bool terminated = capped; // the ray just spawned sits at the depth cap: radiance() = 0 (Scene.cpp:128)
        capped = false;
        if (haveHit) {
          haveHit = false;
          if (key.idx == kMiss) {
            term = ld3(p.env);
            terminated = true;
          } else {
            sIdx = key.idx;
            sBack = key.det < kEpsilon;
            sPos = o + d * key.t;
            sDin = d;
            if (depth == 0) { // the first-bounce surface: its fan-out starts
              firstIdx = sIdx | (sBack ? 0x80000000u : 0u);
              stD3(kFanPos, sPos);
              stD3(kFanDir, d);
              sub = 0;
              stD3(kFanSum, mk(0, 0, 0));
            }
          }
        }
        if (terminated) {
          if (depth == 0) { // primary ray missed
            doneL = term;
            finished = true;
            break;
          }
          // fold innermost-first; level 0 is the first-bounce surface
          d3 L = term;
          for (int i = nlev - 1; i >= 0; --i) {
            const uint32_t w = pixLevels[static_cast<size_t>(i) * blockDim.x + threadIdx.x];
            const uint32_t idx = w & 0x7fffffffu;
            const double *m =
                idx >= nsph
                    ? matTable + static_cast<size_t>(static_cast<uint32_t>(
                                     triCompact[static_cast<size_t>(idx - nsph) * kTriCompactDoubles + kTriMaterialIndex])) *
                                     kMatDoubles
                    : spheres[idx].emission; // SphereRec: emission[3] then diffuse[3]
            const d3 e = ld3(m), df = ld3(m + 3);
            L = (w >> 31) ? e + L : e + df * L;
          }
          const d3 result = ldD3(kFanSum) + L;
          if (++sub == nSub) {
            doneL = result * p.invFirstBounce;
            finished = true;
            break;
          }
          // next sub-sample of the fan-out: the first-bounce surface again
          sIdx = firstIdx & 0x7fffffffu;
          sBack = (firstIdx >> 31) != 0;
          stD3(kFanSum, result);
          sPos = ldD3(kFanPos);
          sDin = ldD3(kFanDir);
          depth = 0;
        }
        // the surface (per-lane gather of the compact record + material)
        d3 normal, diffuse;
        Basis basis;
        double coneAngle, ior, invIor, reflectivity;
        bool inside;
        if (sIdx >= nsph) {
          const double *r = triCompact + static_cast<size_t>(sIdx - nsph) * kTriCompactDoubles;
          const d3 n = ld3(r), bx = ld3(r + 3);
          normal = sBack ? -n : n;
          basis.x = sBack ? -bx : bx;
          basis.y = ld3(r + 6);
          basis.z = normal;
          const double *m = matTable + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
          diffuse = ld3(m + 3);
          ior = m[6], invIor = m[7], reflectivity = m[8];
          coneAngle = m[9];
          inside = sBack;
        } else {
          const SphereRec &r = spheres[sIdx];
          d3 n = normalised(sPos - ld3(r.centre));
          inside = dot(n, sDin) > 0;
          if (inside) n = -n;
          normal = n;
          basis = basisFromZ(n);
          diffuse = ld3(r.diffuse);
          coneAngle = r.coneAngle;
          ior = r.ior, invIor = r.invIor, reflectivity = r.reflectivity;
        }
        if (p.preview) { // radiance() in preview mode: the diffuse colour of the first hit
          doneL = diffuse;
          finished = true;
          break;
        }
        const double iorFrom = inside ? ior : 1.0;
        const double iorTo = inside ? 1.0 : ior;
        const double iorRatio = inside ? ior : invIor;
        if (reflectivity < 0) reflectivity = reflectance(normal, sDin, iorFrom, iorTo, iorRatio);
        // scatter; depth == 0: a sub-sample of the fan-out (stratified u, v)
        const bool fromFirst = depth == 0;
        Sfc32 rng = ldRng();
        const uint32_t w0 = rng.next(), w1 = rng.next(), w2 = rng.next(), w3 = rng.next(),
                       w4 = rng.next(), w5 = rng.next();
        stRng(rng);
        stW(kWCount, ldW(kWCount) + 6);
        const double xu = canonicalFromWords(w0, w1), xv = canonicalFromWords(w2, w3),
                     pd = canonicalFromWords(w4, w5);
        const int uS = sub / p.fbV, vS = sub - uS * p.fbV;
        const double ur = static_cast<double>(uS) + xu, vr = static_cast<double>(vS) + xv;
        const double us = p.uPow2 ? ur * p.invU : ur / static_cast<double>(p.fbU);
        const double vs = p.vPow2 ? vr * p.invV : vr / static_cast<double>(p.fbV);
        const double u = fromFirst ? us : xu, v = fromFirst ? vs : xv;
        d3 nd;
        bool refl;
        if (pd < reflectivity) {
          nd = coneSample(reflect(normal, sDin), coneAngle, u, v);
          refl = true;
        } else {
          nd = hemisphereSample<true>(basis, u, v);
          refl = false;
        }
        // push: one word per level (combined primitive index + lobe flag)
        if (fromFirst) nlev = 0;
        pixLevels[static_cast<size_t>(nlev) * blockDim.x + threadIdx.x] = sIdx | (refl ? 0x80000000u : 0u);
        nlev++;
        depth++;
        o = sPos;
        d = nd;
        if (depth < p.maxDepth) break; // a ray to trace
        capped = true;
      }
      if (finished) {
        finishSample(doneL);
        if (active) beginSample();
      }
    }
  }
  if (rayCounters && nrays) atomicAdd(&rayCounters[0], nrays);
}

// -----------------------------------------------------------------------------------------
// resolve: pass-ordered accumulation into the ArrayOutput-shaped running sums.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolveKernel(const TraceParams p,
                                                     const double *__restrict__ stage,
                                                     double *__restrict__ rgbSum,
                                                     uint32_t *__restrict__ counts) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; // element = local pixel * 3 + channel
  const uint32_t n = p.pixCount * 3;
  if (e >= n) return;
  const uint32_t l = e / 3, c = e - l * 3;
  const size_t base = static_cast<size_t>(globalPixel(p, p.pixBegin + l)) * 3 + c;
  double acc = rgbSum[base];
  for (uint32_t k = 0; k < p.npass; ++k) acc += stage[static_cast<size_t>(k) * n + e];
  rgbSum[base] = acc;
  if (e < p.pixCount) counts[globalPixel(p, p.pixBegin + e)] += p.npass;
}

// -----------------------------------------------------------------------------------------
// Batch Scene::intersect for known-answer tests: one lane per ray.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void intersectBatchKernel(
    const TraceParams p, const double *__restrict__ triGeom,
    const TriShade *__restrict__ triShade, const SphereRec *__restrict__ spheres,
    const double *__restrict__ rays, uint64_t n, double *__restrict__ hits) {
  const uint64_t gid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  PixCtx ctx;
  ctx.p = &p;
  ctx.triGeom = triGeom;
  ctx.triShade = triShade;
  ctx.spheres = spheres;
  ctx.rays = 0;
  const d3 o = ld3(rays + gid * 6), d = ld3(rays + gid * 6 + 3);
  const HitKey k = ctx.intersect(o, d);
  double *h = hits + gid * 9;
  if (k.idx == kMiss) {
    h[0] = -1;
    for (int i = 1; i < 9; ++i) h[i] = 0;
    return;
  }
  const d3 pos = o + d * k.t;
  d3 n3;
  bool inside;
  if (k.idx >= p.nsph) {
    const TriShade &r = triShade[k.idx - p.nsph];
    inside = k.det < kEpsilon;
    n3 = inside ? -ld3(r.normal) : ld3(r.normal);
  } else {
    n3 = normalised(pos - ld3(spheres[k.idx].centre));
    inside = dot(n3, d) > 0;
    if (inside) n3 = -n3;
  }
  h[0] = k.t;
  h[1] = inside ? 1.0 : 0.0;
  h[2] = pos.x, h[3] = pos.y, h[4] = pos.z;
  h[5] = n3.x, h[6] = n3.y, h[7] = n3.z;
  h[8] = static_cast<double>(k.idx); // combined primitive index; the host maps it to a material
}

// Device RNG known-answer kernel: one wave drives the same LDS generator the render uses.
__global__ __launch_bounds__(64) void rngKatKernel(int rngPolicy,
                                                   const uint32_t *__restrict__ seedState,
                                                   uint32_t seed, uint32_t pixel, uint32_t n,
                                                   double *__restrict__ out) {
  __shared__ SeqShared sh;
  if (rngPolicy == PTW_RNG_SEQUENTIAL) {
    for (int i = threadIdx.x; i < kMtWords; i += 64) sh.mt[i] = seedState[i];
    int pos = kMtDoubles;
    for (uint32_t i = 0; i < n; ++i) {
      if (pos == kMtDoubles) {
        mtRegenerateWave(&sh, threadIdx.x);
        pos = 0;
      }
      const double v = sh.canon[pos++];
      if (threadIdx.x == 0) out[i] = v;
    }
  } else if (threadIdx.x == 0) {
    Sfc32 rng;
    rng.seed(seed, pixel);
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t w0 = rng.next();
      const uint32_t w1 = rng.next();
      out[i] = canonicalFromWords(w0, w1);
    }
  }
}

constexpr size_t kLdsTableBudget = 150 * 1024; // bytes of LDS we are willing to spend on tables

// Name of the variant the last launch*() call of this thread picked (reported through
// ptw_kernel_stats so that callers do not have to re-derive the dispatch rules).
thread_local const char *tlsVariant = "";
thread_local char tlsVariantBuf[80];

// The units of 64 triangles per worker wave: older / younger wave of a worker pair, master-side wave.
// Equal shares (seqUnitSplit); two masters, scenes from 31 units on: shares by the wave's place
// (seqUnitSplitByPlace).  LaunchHints::seqUnits sets them outright (tests, A/B runs; what does not fit
// the waves' shares is streamed from memory).
void seqUnitsFor(uint32_t ntri, int nA, int nB, int cap, const LaunchHints &hints, int &uO, int &uY, int &uM) {
  int uA, uB;
  seqUnitSplit(ntri, nA, nB, 100, cap, uA, uB);
  uO = uY = uA, uM = uB;
  if (nA == 4 && nB == 2) (void)seqUnitSplitByPlace(ntri, PTW_SEQ_YOUNG_PERCENT, cap, uO, uY, uM);
  const int o = hints.seqUnits[0], y = hints.seqUnits[1], m = hints.seqUnits[2];
  if ((o | y | m) != 0 && o >= 0 && y >= 0 && m >= 0 && o <= cap && y <= cap && m <= cap) uO = o, uY = y, uM = m;
}

template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, int MASTERS = 1, bool PAIR = false>
hipError_t launchSeq(const TraceParams &pIn, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  TraceParams p = pIn;
  if (WAVES > 1) {
    int uO, uY, uM;
    seqUnitsFor(p.ntri, WAVES - MASTERS, MASTERS, SLOTS, hints, uO, uY, uM);
    p.seqUnitsA = uO, p.seqUnitsY = uY, p.seqUnitsB = uM;
  }
  // (one wave per pass: the pick checksum is its own instantiation, see SeqCtx::picksOn)
  auto kernel = WAVES == 1 && !b.picks ? traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS, PAIR, WAVES != 1>
                                       : traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS, PAIR, true>;
  std::snprintf(tlsVariantBuf, sizeof tlsVariantBuf, "traceSequential<%d,%d,%s,%s%s%s>", SLOTS, WAVES,
                LDS_TABLES ? "lds" : "global", REG ? "reg" : "stack", MASTERS == 2 ? ",2 masters" : "",
                PAIR ? ",paired" : "");
  tlsVariant = tlsVariantBuf;
  const size_t lds = seqLdsBytes(WAVES, p.maxDepth, LDS_TABLES, p.ntri, p.nmat, p.nsph, MASTERS, PAIR);
  if (lds > 48 * 1024) { // per launch: the attribute belongs to the current device's copy of the kernel
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3((p.npass + MASTERS - 1) / MASTERS),
                     dim3(WAVES == 1 ? 64 : 64 * (WAVES + MASTERS)), lds, stream, p, b.triGeom, b.triShade,
                     b.spheres, b.triCompact, b.matTable, b.mtState, b.mtPos, b.stage, b.words,
                     b.rays, b.picks);
  return hipGetLastError();
}

// ... with the shading tables in LDS when they fit (LaunchHints::seqLdsTables == 0: global memory
// whatever their size - the tests reach the global-table instantiations with small scenes that way)
template <int SLOTS, int WAVES, int MASTERS = 1, bool PAIR = false>
hipError_t launchSeqAuto(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  const size_t tables = seqLdsBytes(WAVES, p.maxDepth, true, p.ntri, p.nmat, p.nsph, MASTERS, PAIR);
  if (hints.seqLdsTables != 0 && tables <= kLdsTableBudget)
    return launchSeq<SLOTS, WAVES, true, false, MASTERS, PAIR>(p, b, hints, stream);
  return launchSeq<SLOTS, WAVES, false, false, MASTERS, PAIR>(p, b, hints, stream);
}

// The two-master kernels by the largest share of 64-triangle units any worker wave gets; shares are
// capped at what the register file holds without spilling inside the search loop (11 units = 198
// registers), the rest of a larger scene is streamed from memory.
template <bool PAIR>
hipError_t launchSeqTwoMasters(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  int uO, uY, uM;
  seqUnitsFor(p.ntri, 4, 2, 11, hints, uO, uY, uM);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 6, 2, PAIR>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 6, 2, PAIR>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 6, 2, PAIR>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 6, 2, PAIR>(p, b, hints, stream);
  if (need <= 6) return launchSeqAuto<6, 6, 2, PAIR>(p, b, hints, stream);
  if (need <= 9) return launchSeq<9, 6, false, false, 2, PAIR>(p, b, hints, stream);
  if (need <= 10) return launchSeq<10, 6, false, false, 2, PAIR>(p, b, hints, stream);
  return launchSeq<11, 6, false, false, 2, PAIR>(p, b, hints, stream);
}


hipError_t launchSeqSpec(const TraceParams &p, const TraceBuffers &b, hipStream_t stream) {
  tlsVariant = "traceSequentialSpec";
  const size_t lds = specLdsBytes(p.ntri, p.nmat, p.nsph);
  auto kernel = b.picks ? traceSequentialSpec<true> : traceSequentialSpec<false>;
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(p.npass), dim3(64 * (kSpecWaves + 1)), lds, stream, p,
                     b.triGeom, b.spheres, b.triCompact, b.matTable, b.mtState, b.specState, b.stage,
                     b.words, b.rays, b.picks);
  return hipGetLastError();
}

#if PTW_EXPERIMENTS
hipError_t launchSeqGang(const TraceParams &pIn, const TraceBuffers &b, hipStream_t stream, int G) {
  std::snprintf(tlsVariantBuf, sizeof tlsVariantBuf, "traceSequentialGang<%d CUs per pass>", G);
  tlsVariant = tlsVariantBuf;
  // the candidate set of this launch, from what the previous launch measured
  hipError_t e = launchBuildCandidates(pIn, b, kGangWaves * G, stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(b.gangRecords, 0, gangRecordBytes(pIn.npass), stream); // tags of earlier launches
  if (e != hipSuccess) return e;
  const size_t lds = gangLdsBytes(pIn.ntri, pIn.nmat, pIn.nsph);
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(traceSequentialGang),
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return e;
  TraceParams p = pIn;
  const double *triGeom = b.triGeom;
  const SphereRec *spheres = b.spheres;
  const double *triCompact = b.triCompact, *matTable = b.matTable;
  uint32_t *mtState = b.mtState, *words = b.words;
  double *specState = b.specState, *stage = b.stage;
  unsigned long long *rays = b.rays, *countHist = b.countHist;
  const WideCandidates *cands = reinterpret_cast<const WideCandidates *>(b.wideCands);
  GangRecord *records = reinterpret_cast<GangRecord *>(b.gangRecords);
  int groups = G;
  void *args[] = {&p, &triGeom, &spheres, &triCompact, &matTable, &mtState, &specState, &stage, &words, &rays,
                  &cands, &countHist, &records, &groups};
  // cooperative: the workgroups of a pass wait for each other, all of them must be resident
  const uint32_t grid = ((pIn.npass + 7u) / 8u) * 8u * static_cast<uint32_t>(G);
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(traceSequentialGang), dim3(grid),
                                    dim3(64 * kGangWaves), args, static_cast<unsigned>(lds), stream);
}

#endif

} // namespace

#if PTW_EXPERIMENTS
size_t wideCandidateBytes() { return sizeof(WideCandidates); }
size_t gangRecordBytes(uint32_t npass) { return static_cast<size_t>(npass) * 2 * kGangMaxCand * sizeof(GangRecord); }

hipError_t launchBuildCandidates(const TraceParams &p, const TraceBuffers &b, int n, hipStream_t stream) {
  // A round reads up to maxD + 3 maxDepth draws beyond the frontier and only the frontier's block
  // and the next one exist: candidates stay within 312 - 3 maxDepth - 8 draws of the frontier.
  const int ahead = kMtDoubles - 3 * (p.maxDepth > 0 ? p.maxDepth : 1) - 8;
  const int maxAhead = ahead < 255 ? ahead : 255;
  hipLaunchKernelGGL(wideBuildCandidates, dim3(1), dim3(64), 0, stream, b.countHist, n, p.fbU * p.fbV, maxAhead,
                     reinterpret_cast<WideCandidates *>(b.wideCands));
  return hipGetLastError();
}

#endif

namespace {
hipError_t dispatchSequential(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
int deviceCus();
// the scenes the register-resident speculative kernels handle (traceSequentialSpec / Gang)
bool specApplies(const TraceParams &p) {
  return p.ntri <= 64 && p.nsph <= 64 && p.nsph + p.ntri <= 127 && p.maxDepth <= 9 &&
         seqLdsBytes(1, p.maxDepth, true, p.ntri, p.nmat, p.nsph) <= kLdsTableBudget;
}
}

#if PTW_EXPERIMENTS
// Workgroups (CUs) per pass for traceSequentialGang (LaunchHints::gangGroups = 2, 4 or 8: that many
// when every workgroup of the launch can be resident; 0, the default: never - DESIGN.md 3.1d).
int seqGangGroups(const TraceParams &p, const LaunchHints &hints) {
  const int g = hints.gangGroups;
  if (!(g == 2 || g == 4 || g == 8) || hints.seqSmallKernel == 0 || hints.seqSmallKernel == 1) return 0;
  if (!specApplies(p) || p.npass == 0) return 0;
  const uint32_t cus = static_cast<uint32_t>(deviceCus());
  return ((p.npass + 7u) / 8u) * 8u * static_cast<uint32_t>(g) <= cus ? g : 0;
}
#else
int seqGangGroups(const TraceParams &, const LaunchHints &) { return 0; } // (the experiments build only)
#endif

hipError_t launchTraceSequential(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints,
                                 hipStream_t stream, const char **variant) {
  const hipError_t e = dispatchSequential(p, b, hints, stream);
  if (variant) *variant = tlsVariant;
  return e;
}

namespace {
int deviceCus() {
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

hipError_t dispatchSequential(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  const uint32_t n = p.ntri;
  // Smallest configuration that keeps every triangle resident in VGPRs.  Up to 128 triangles
  // one wave does everything.  Beyond that 7 worker waves + 1 master wave = 8 waves = 2 per SIMD
  // of one CU (256 registers per lane each): SLOTS triangles per worker lane.
  if (n <= 64) {
    // register-resident shading records + scalar (E, T) stack when the byte-per-level encoding
    // fits (LaunchHints::seqSmallKernel == 0 forces the LDS-table variant)
    const bool reg = hints.seqSmallKernel != 0 && specApplies(p);
    // ... and, by default, with the fan-out traced speculatively by four waves.  The speculative
    // kernel spends a whole CU on a pass.  That pays while there are at most as many passes as CUs;
    // with more, one wave per pass on every SIMD is the better use of the chip.
    const int cus = deviceCus();
    const bool forced = hints.seqSmallKernel == 2; // whatever the pass count
    if (reg && hints.seqSmallKernel != 1 && b.specState && (forced || p.npass <= static_cast<uint32_t>(cus))) {
#if PTW_EXPERIMENTS
      // several CUs per pass (experiments/ptw_gang.h)
      if (b.gangRecords && b.countHist && b.wideCands)
        if (const int G = seqGangGroups(p, hints)) return b.picks ? hipErrorNotSupported : launchSeqGang(p, b, stream, G);
#endif
      return launchSeqSpec(p, b, stream);
    }
    if (reg) return launchSeq<1, 1, true, true>(p, b, hints, stream);
    return launchSeqAuto<1, 1>(p, b, hints, stream);
  }
  if (n <= 128) return launchSeqAuto<2, 1>(p, b, hints, stream);
  // Beyond 128 triangles a pass occupies a whole CU (8 waves of up to 256 registers), and its
  // workers idle while the master shades.  With more passes than CUs, two passes share a
  // workgroup instead: two masters over six worker waves, the workers searching one master's request
  // while the other master shades (measured: suzanne 512 passes 7.2 -> 11.9 Msamples/s, ce 1024
  // passes 1.43 -> 2.02; with no more passes than CUs it would only leave CUs empty).
  // LaunchHints::seqTwoMasters 0 / 1: never / always.
  const bool mm = hints.seqTwoMasters == 0 || hints.seqTwoMasters == 1 ? hints.seqTwoMasters == 1
                                                                      : p.npass > static_cast<uint32_t>(deviceCus());
  if (mm) {
#if PTW_EXPERIMENTS
    // LaunchHints::seqPairing == 1: the PAIR form (experiments/ptw_pair.h: two sub-samples in flight per
    // master, two rays per request) where the fan-out has sub-samples to pair and the chains levels to trace
    const bool canPair = p.maxDepth >= 2 && p.maxDepth <= 9 && p.fbU * p.fbV >= 2 && !p.preview;
    if (canPair && hints.seqPairing == 1) return launchSeqTwoMasters<true>(p, b, hints, stream);
#endif
    return launchSeqTwoMasters<false>(p, b, hints, stream);
  }
  int uO, uY, uM;
  seqUnitsFor(n, 6, 1, 12, hints, uO, uY, uM);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 7>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 7>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 7>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 7>(p, b, hints, stream);
  if (need <= 6) return launchSeq<6, 7, false>(p, b, hints, stream);
  if (need <= 8) return launchSeq<8, 7, false>(p, b, hints, stream);
  return launchSeq<12, 7, false>(p, b, hints, stream); // beyond 5376 the tail is streamed from memory
}
} // namespace

hipError_t launchTracePerPixel(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints,
                               hipStream_t stream, const char **variant) {
  // Two kernels, each the better one somewhere (one box, one run: profiles/r03a_perpixel_*):
  //   lock-step (tracePerPixel, a lane traces a whole sample, 8 samples per lane through a
  //     grid-stride loop): Cornell 1024x1024 @ 256 spp 224 Msamples/s against 146 - in a closed scene
  //     nearly every path runs to the depth cap, the lanes of a wave stay together, and the
  //     shading code runs once per level for all of them;
  //   persistent (tracePerPixelPersistent, lanes that finish a path take the next sample): suzanne
  //     44 against 19, bbc-owl 395 against 139 - open scenes, where most paths of a wave end early.
  // p.pixKernel carries the caller's choice (ptw_render_params.pix_kernel, or what ptw_context_calibrate
  // measured for this scene and frame shape; the persistent kernel when neither); the accelerated mode
  // has its own kernel.
  if (p.accel == PTW_ACCEL_BVH) {
    if (variant) *variant = "tracePerPixelBvh";
    const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
    const uint32_t blocks = static_cast<uint32_t>((total + kPixBlock - 1) / kPixBlock);
    const int levels = p.maxDepth > 1 ? p.maxDepth - 1 : 1;
    const size_t lds = static_cast<size_t>(levels + kBvhStack) * kPixBlock * sizeof(uint32_t);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tracePerPixelBvh),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(tracePerPixelBvh, dim3(blocks), dim3(kPixBlock), lds, stream, p, b);
    return hipGetLastError();
  }
  const bool persistent = p.pixKernel != kPixKernelLockstep;
  if (variant) *variant = persistent ? "tracePerPixelPersistent" : "tracePerPixel";
  if (persistent) {
    const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // persistent grid: W blocks of 256 lanes per CU (W waves per SIMD), fewer for tiny jobs.
    // W = 4: 128 VGPRs with 128 B/lane of scratch outside the triangle loop; 3 waves (153 VGPRs,
    // nothing spilled) measured the same or up to 9 % slower (profiles/r02q_persistent_lean_probe.txt).
    // LaunchHints::pixWavesPerSimd = 2 | 3 | 4 for A/B runs.
    const int W = hints.pixWavesPerSimd >= 2 && hints.pixWavesPerSimd <= 4 ? hints.pixWavesPerSimd : 4;
    uint64_t blocks = static_cast<uint64_t>(cus) * W;
    const uint64_t needed = (total + kPix2Block - 1) / kPix2Block;
    if (blocks > needed) blocks = needed;
    const int levels = p.maxDepth > 1 ? p.maxDepth : 1; // level 0: the first-bounce surface
    // level words (+ nine doubles and seven words of per-lane state for the small scenes)
    const bool ldsState = p.ntri < 128;
    const size_t lds = ((static_cast<size_t>(levels) * kPix2Block * sizeof(uint32_t) + 7) & ~size_t(7)) +
                       (ldsState ? 9 * kPix2Block * sizeof(double) + 7 * kPix2Block * sizeof(uint32_t) : 0);
    hipError_t e = hipMemsetAsync(b.sampleQueue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    auto kernel = ldsState ? (W == 2 ? tracePerPixelPersistent<2, true> : W == 3 ? tracePerPixelPersistent<3, true>
                                                                               : tracePerPixelPersistent<4, true>)
                           : (W == 2 ? tracePerPixelPersistent<2, false> : W == 3 ? tracePerPixelPersistent<3, false>
                                                                                : tracePerPixelPersistent<4, false>);
    if (lds > 48 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(blocks)),
                       dim3(kPix2Block), lds, stream, p, b.triGeom, b.spheres, b.triCompact,
                       b.matTable, b.stage, b.words, b.rays, b.sampleQueue);
    return hipGetLastError();
  }
  const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
  // LaunchHints::pixSamplesPerLane = n: n samples per lane through the kernel's grid-stride loop (default
  // 8; 1 = a block per 256 samples)
  // (measured on Cornell 1024x1024 @ 256: 1 -> 194, 4 -> 223, 16 -> 224, 64 -> 219, 256 -> 201 Msamples/s:
  // a block per 256 samples is a million block dispatches per frame)
  const uint64_t spl = hints.pixSamplesPerLane > 0 ? static_cast<uint64_t>(hints.pixSamplesPerLane) : 8;
  const uint32_t blocks = static_cast<uint32_t>((total + kPixBlock * spl - 1) / (kPixBlock * spl));
  const int levels = p.maxDepth > 1 ? p.maxDepth - 1 : 1;
  const size_t lds = static_cast<size_t>(levels) * kPixBlock * sizeof(uint32_t);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tracePerPixel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(tracePerPixel, dim3(blocks), dim3(kPixBlock), lds, stream, p, b);
  return hipGetLastError();
}

hipError_t launchResolve(const TraceParams &p, const double *stage, double *rgbSum,
                         uint32_t *counts, hipStream_t stream) {
  const uint32_t n = p.pixCount * 3;
  hipLaunchKernelGGL(resolveKernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, stage, rgbSum,
                     counts);
  return hipGetLastError();
}

hipError_t launchRngKat(int rngPolicy, const uint32_t *mtSeedState, uint32_t seed, uint32_t pixel,
                        uint32_t n, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(rngKatKernel, dim3(1), dim3(64), 0, stream, rngPolicy, mtSeedState, seed,
                     pixel, n, out);
  return hipGetLastError();
}

hipError_t launchIntersectBatch(const TraceParams &p, const TraceBuffers &b, const double *rays,
                                uint64_t n, double *hitsOut, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(intersectBatchKernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256),
                     0, stream, p, b.triGeom, b.triShade, b.spheres, rays, n, hitsOut);
  return hipGetLastError();
}

} // namespace ptw

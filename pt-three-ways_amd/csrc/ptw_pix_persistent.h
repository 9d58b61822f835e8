// ptw_pix_persistent.h - tracePerPixelPersistent, the persistent form of the PERPIXEL policy: perpixel.hip
// instantiates the brute-force kernel, accel.hip the one with the fp32 prefilter.  Internal to csrc/.
#pragma once
#include "ptw_pix_ctx.h"

namespace ptw {
using namespace ptwd;
namespace {

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

// PREFILTER (accel.hip, PTW_ACCEL_PREFILTER): the triangle loop looks at two triangles per packed fp32
// instruction first and runs the fp64 test only where fp32 cannot prove a rejection (prefilteredTriangles,
// ptw_pix_ctx.h) - the same hits, bit for bit.
template <int W, bool LDS_STATE, bool PREFILTER = false>
__global__ __launch_bounds__(kPix2Block) __attribute__((amdgpu_waves_per_eu(W, W))) void tracePerPixelPersistent(
    const TraceParams p, const double *__restrict__ triGeom,
    const SphereRec *__restrict__ spheres, const double *__restrict__ triCompact,
    const double *__restrict__ matTable, double *__restrict__ stage, uint32_t *__restrict__ words,
    unsigned long long *__restrict__ rayCounters, unsigned long long *__restrict__ sampleQueue,
    const float *__restrict__ triPacked) {
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
      if (PREFILTER) {
        if (ntri) prefilteredTriangles(o, d, triPacked, triGeom, nsph, ntri, key);
      } else if (ntri) {
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


} // namespace
} // namespace ptw

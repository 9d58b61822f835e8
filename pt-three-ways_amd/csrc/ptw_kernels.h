// ptw_kernels.h — device data layout and kernel launchers of the hip way.
#pragma once

#include "../../include/ptw.h"
#include "ptw_layout.h"

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ptw {

struct TraceParams {
  ptw_camera cam;
  double env[3];
  double invFirstBounce; // 1.0 / (fbU * fbV)
  double invU, invV;     // 1.0 / fbU, 1.0 / fbV (used when they are powers of two)
  uint32_t ntri, nsph, nmat, padA;
  int32_t width, height;
  int32_t maxDepth, fbU, fbV, preview;
  int32_t uPow2, vPow2;
  int32_t rngPolicy;
  uint32_t passSeedBase; // uint32(seed + first_pass): pass k uses passSeedBase + k
  // A launch covers the band [pixBegin, pixBegin + pixCount) of the shard's LOCAL pixel index
  // l = localRow * width + x; local row r is image row rowFirst + r * rowStride (PERPIXEL
  // interleaved-row shards).  SEQUENTIAL renders whole frames: rowFirst = 0, rowStride = 1,
  // so there local == global.
  uint32_t pixBegin;
  uint32_t pixCount;     // pixels in the band
  uint32_t npix;         // width * height
  uint32_t npass;
  int32_t firstBand;     // this launch starts the passes' streams (first band of a render)
  int32_t rowFirst;      // image row of local row 0
  int32_t rowStride;     // image rows between consecutive local rows (>= 1)
  int32_t accel;         // ptw_accel: PERPIXEL only
  int32_t pixKernel;     // PERPIXEL: 0 = launcher's default, 1 = lock-step (grid-stride), 2 = persistent
  // SEQUENTIAL worker-wave kernels: resident triangles in units of 64 per worker wave - seqUnitsA
  // for the waves that share a SIMD with another worker, seqUnitsB for those beside a master wave
  // (set by the launcher, see seqUnitSplit)
  int32_t seqUnitsA, seqUnitsB;
  // ... and, among the waves that share a SIMD with another worker, the YOUNGER one of each pair (the
  // arbiter serves the older wave first: round 4 measured the younger ones 35-40 % slower per triangle,
  // profiles/r04k_*): seqUnitsA is then the older wave's share.  Equal to seqUnitsA by default.
  int32_t seqUnitsY;
  // ... and whether the worker waves try the unit-level u-first early-out (testTriangleUnit: scenes whose units of
  // 64 consecutive triangles mostly fail the u test as a whole - host/precompute.h unitUSkipFraction)
  int32_t seqUnitUFirst;
};
constexpr int kPixKernelAuto = 0, kPixKernelLockstep = 1, kPixKernelPersistent = 2;

// How the worker-wave kernels spread a scene's ceil(ntri / 64) units of 64 triangles over their
// worker waves.  A CU's four SIMDs carry eight waves: the master wave(s) and nB workers sit two to a
// SIMD with each other, the other nA workers two to a SIMD among themselves.  The fp64 pipe of a
// SIMD issues one wave instruction per four cycles whoever it comes from, so two workers on one
// SIMD search at half speed each, while a worker beside a master has the SIMD to itself whenever
// that master waits for this very search.  Equal shares therefore leave the search waiting for the
// doubly loaded SIMDs; `ratio` (percent) is the share of a master-side worker relative to the
// others.  Shares are capped at `maxUnits` (what the register file holds); what does not fit is
// streamed from memory by all workers.
__host__ __device__ inline void seqUnitSplit(uint32_t ntri, int nA, int nB, int ratio, int maxUnits, int &uA, int &uB) {
  const int U = static_cast<int>((ntri + 63u) / 64u);
  const int den = nA * 100 + nB * ratio;
  uB = nB > 0 ? (U * ratio + den / 2) / den : 0;
  if (uB > maxUnits) uB = maxUnits;
  if (nB > 0 && uB * nB > U) uB = U / nB;
  uA = nA > 0 ? (U - nB * uB + nA - 1) / nA : 0;
  if (uA > maxUnits) uA = maxUnits;
  if (uA < 0) uA = 0;
}

// Two-master kernels (four workers in two pairs + two master-side workers), scenes from 12 units on:
// shares by the wave's PLACE.  Of two workers on one
// SIMD the arbiter serves the older wave first; measured per triangle (profiles/r04k_*, ce) the younger
// one is 35-40 % slower and sets the tick, the master-side waves sit in between.  So the younger wave
// of each pair gets youngPercent (70) of an older / master-side wave's share: ce's 54 units go 10 / 7 /
// 10 instead of 9 / 9 / 9 (1.975 -> 2.150 Msamples/s, profiles/r04l_*).  Through round 5 the rule started at
// 31 units (scenes whose tick is the search); round 6 measured it below (profiles/r06w_*): suzanne's 16 units
// 3 / 2 / 3 instead of 3 / 3 / 3 +4.9 % (the masters set its tick, and the younger workers are slow there
// too), closed soups +1.4 % at 16 units, +8 % at 25, nothing at 8.  Returns false - leave the equal shares -
// when the scene is smaller, or when an older wave would need more than `cap` (the instantiation the equal
// shares chose: a bigger one costs more than the shares bring) or more than ten units (the largest
// instantiation that does not spill).
__host__ __device__ inline bool seqUnitSplitByPlace(uint32_t ntri, int youngPercent, int cap, int &uOld, int &uYoung, int &uMaster) {
  const int U = static_cast<int>((ntri + 63u) / 64u);
  if (U < 12 || youngPercent <= 0 || youngPercent >= 100) return false;
  const int half = (U + 1) / 2;                 // one pair + one master-side wave: uOld + uYoung + uMaster
  const int den = 200 + youngPercent;
  const int y = (half * youngPercent + den / 2) / den;
  const int o = (half - y + 1) / 2;
  if (o > cap || o > 10) return false;
  uOld = uMaster = o;
  uYoung = y;
  return true;
}

// One-master kernels (seven workers: three pairs on SIMD 1-3 + one wave beside the master), scenes from 8 units on:
// the same observation (profiles/r06y_*, 256 passes: suzanne 3 / 3 / 3 -> 3 / 2 / 1 +8.7 %, what the wave beside the
// master holds does not matter there; ce 8 / 8 / 6 -> 9 / 6 / 9 +6 % in a new <9,7> instantiation, 10 / 7 / 3 and
// 8 / 7 / 10 no better than equal shares; closed soups +2.6 % at 8 units ... +6 % at 25-57, never slower).  A unit
// costs a younger wave 100 / youngPercent of what it costs an older or the master-side wave; of the shares o >= y,
// o >= m with 3 o + 3 y + m >= U and o <= cap (and <= 10: twelve slots spill) the ones with the cheapest slowest wave,
// then the smallest o, y, m.  Returns false - equal shares - when nothing fits.
__host__ __device__ inline bool seqUnitSplitByPlaceOneMaster(uint32_t ntri, int youngPercent, int cap, int &uOld, int &uYoung, int &uMaster) {
  const int U = static_cast<int>((ntri + 63u) / 64u);
  if (U < 8 || youngPercent <= 0 || youngPercent >= 100) return false;
  long best = -1;
  for (int o = 1; o <= cap && o <= 10; ++o)
    for (int y = 1; y <= o; ++y) {
      int m = U - 3 * o - 3 * y;
      if (m > o) continue;
      if (m < 0) m = 0;
      const long costY = (static_cast<long>(y) * 10000 + youngPercent - 1) / youngPercent; // in 1 / 100 units
      long cost = static_cast<long>(o) * 100;
      if (costY > cost) cost = costY;
      const long key = ((cost * 16 + o) * 16 + y) * 16 + m;
      if (best < 0 || key < best) best = key, uOld = o, uYoung = y, uMaster = m;
    }
  return best >= 0;
}

// Global (row-major, full-frame) index of local pixel l.
__host__ __device__ inline uint32_t globalPixel(const TraceParams &p, uint32_t l) {
  if (p.rowStride == 1) return static_cast<uint32_t>(p.rowFirst) * static_cast<uint32_t>(p.width) + l;
  const uint32_t w = static_cast<uint32_t>(p.width);
  const uint32_t r = l / w, x = l - r * w;
  return (static_cast<uint32_t>(p.rowFirst) + r * static_cast<uint32_t>(p.rowStride)) * w + x;
}

struct TraceBuffers {
  const double *triGeom;     // [ntri][9]: v0, e1 = v1 - v0, e2 = v2 - v0
  const TriShade *triShade;  // [ntri]
  const SphereRec *spheres;  // [nsph]
  const double *triCompact;  // [ntri][kTriCompactDoubles]  (copied into LDS by traceSequential)
  const double *matTable;    // [nmat][kMatDoubles]
  uint32_t *mtState;         // [npass][624] raw mt19937 state (SEQUENTIAL)
  uint32_t *mtPos;           // [npass] next canonical double (0..312; 312 = regenerate)
  double *stage;             // [npass][pixCount][3] this band's per-pass radiance
  uint32_t *words;           // optional [npass][npix]
  uint32_t *picks;           // optional [npass][npix]: per-sample pick checksum (SEQUENTIAL kernels)
  unsigned long long *rays;  // optional [npass] intersect() call counters
  unsigned long long *sampleQueue; // one word: next sample index (tracePerPixelPersistent)
  double *specState;         // [npass][kSpecStateDoubles] parked stream ring (traceSequentialSpec)
  // accelerated mode (host/bvh.h)
  const void *bvhNodes;
  const double *bvhLeafGeom;
  const uint32_t *bvhLeafIndex;
  // prefilter mode (host/prefilter.h): [(ntri + 1) / 2][22] floats, two triangles per record
  const float *triPacked;
};

// What ptw_debug_options asks of the launchers (include/ptw.h; the defaults leave every decision to
// the dispatch rules).  Tests and A/B runs only.
struct LaunchHints {
  int seqTwoMasters = -1, seqLdsTables = -1, seqSmallKernel = -1;
  int seqUnits[3] = {0, 0, 0};
  int pixSamplesPerLane = 0, pixWavesPerSimd = 0;
  // ptw_dispatch_plan: name the kernel the dispatch rules pick without launching anything (no device needed),
  // for `cus` compute units (0 = the current device's count)
  bool dryRun = false;
  int cus = 0;
};

// SEQUENTIAL policy: one workgroup per pass walks the band's pixels in row-major order.
// `variant` (may be null) receives the name of the kernel variant that was launched.
hipError_t launchTraceSequential(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints,
                                 hipStream_t stream, const char **variant = nullptr);
// Scenes of at most 64 triangles with MORE passes than compute units: the four-wave speculative kernel (a CU
// per pass, the workgroups in turns) or one wave per pass (the SIMDs fill up as the pass count grows)?  Which one
// wins between one and about six passes per CU depends on how often the scene's speculation commits (closed
// scenes: four sub-samples per round; Cornell: two) - round 6's sweep measured 1.7x either way.  True when the
// dispatch rules leave that open for this launch (then capi_render.hip times both once: ptw_context_calibrate).
bool seqSmallKernelIsOpen(const TraceParams &p, const LaunchHints &hints);
// PERPIXEL policy: one lane per (pass, pixel) sample.
hipError_t launchTracePerPixel(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints,
                               hipStream_t stream, const char **variant = nullptr);
// rgbSum[pix] += sum over passes (in pass order) of stage[pass][l]; counts[pix] += npass, for the
// local pixels l of the band (pix = globalPixel(p, l)).
hipError_t launchResolve(const TraceParams &p, const double *stage, double *rgbSum,
                         uint32_t *counts, hipStream_t stream);
// Scene::intersect for a batch of rays (known-answer tests).
hipError_t launchIntersectBatch(const TraceParams &p, const TraceBuffers &b, const double *rays,
                                uint64_t n, double *hitsOut, hipStream_t stream);

// Device RNG known-answer test: n canonical doubles from mt19937(state) / the sfc32 stream.
hipError_t launchRngKat(int rngPolicy, const uint32_t *mtSeedState, uint32_t seed, uint32_t pixel,
                        uint32_t n, double *out, hipStream_t stream);

} // namespace ptw

// dispatch.hip - which kernel runs: the dispatch rules of the SEQUENTIAL policy (launchTraceSequential) and
// what the kernel translation units share on the host side (ptw_launch.h).  The kernels themselves:
//
//  traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS>   (ptw_seq_kernel.h; seq_single.hip, seq_worker.hip,
//      seq_worker2.hip)  SEQUENTIAL RNG policy, bit-compatible with the reference's per-pass std::mt19937
//      stream.  Within a pass the reference consumes one RNG stream across pixels in row-major order with a
//      data-dependent number of draws per pixel (Scene.cpp:211-217), so pixels of one pass are serially
//      dependent.  Parallelism is therefore (a) across passes: ONE WORKGROUP PER PASS, and (b) inside a ray's
//      brute-force nearest-hit search: every lane owns SLOTS triangles, resident in VGPRs for the whole
//      launch, tests them against the wave-uniform ray, and the nearest hit is picked with the reference's
//      tie-break (lowest insertion index; spheres before triangles).  WAVES == 1: one wave does everything
//      (scenes up to 128 triangles; REG: the issue-slot-lean variant for up to 64).  WAVES == 7: worker waves
//      hold the primitives, a master wave runs the path logic.  WAVES == 6, MASTERS == 2 (more passes than
//      CUs): two passes per workgroup, the workers searching one master's ray while the other master shades.
//  traceSequentialSpec   (seq_spec.hip)  the same policy for scenes up to 64 triangles with the first-bounce
//      fan-out traced speculatively by four waves against a two-block stream ring that a fifth wave keeps
//      filled (the headline kernel).
//  tracePerPixel, tracePerPixelPersistent   (perpixel.hip)  PERPIXEL policy: one lane per (pass, pixel) sample.
//  tracePerPixelBvh, tracePerPixelPrefilter   (accel.hip)  the SEPARATE accelerated modes - bit-identical results.
//  resolveKernel, intersectBatchKernel, rngKatKernel   (resolve_kat.hip)
//
// (Kernels that were built, measured slower and retired from the tree - several CUs per pass, paired requests of
// the two-master kernels, the decoupled two-master protocol, the many-candidate kernels - are described in
// LAB.md with the commits that hold their last revision.)
#include "ptw_launch.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>

namespace ptw {

namespace {
thread_local char tlsVariantBuf[96] = "";
} // namespace

void setVariant(const char *format, ...) {
  va_list args;
  va_start(args, format);
  std::vsnprintf(tlsVariantBuf, sizeof tlsVariantBuf, format, args);
  va_end(args);
}
const char *lastVariant() { return tlsVariantBuf; }

int deviceCus() {
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

bool seqSmallKernelIsOpen(const TraceParams &p, const LaunchHints &hints) {
  const uint32_t cus = static_cast<uint32_t>(cusFor(hints));
  return p.rngPolicy == PTW_RNG_SEQUENTIAL && p.ntri <= 64 && hints.seqSmallKernel == -1 && specApplies(p) && p.npass > cus &&
         p.npass <= 6 * cus;
}

namespace {
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
    const int cus = cusFor(hints);
    const bool forced = hints.seqSmallKernel >= 2 && hints.seqSmallKernel <= 4; // whatever the pass count
    if (reg && hints.seqSmallKernel != 1 && b.specState && (forced || p.npass <= static_cast<uint32_t>(cus)))
      return launchSeqSpec(p, b, hints, stream);
    return launchSeqSingle(p, b, hints, stream, 1, reg);
  }
  if (n <= 128) return launchSeqSingle(p, b, hints, stream, 2, false);
  // Beyond 128 triangles a pass occupies a whole CU (8 waves of up to 256 registers), and its
  // workers idle while the master shades.  With more passes than CUs, two passes share a
  // workgroup instead: two masters over six worker waves, the workers searching one master's request
  // while the other master shades (measured: suzanne 512 passes 7.2 -> 11.9 Msamples/s, ce 1024
  // passes 1.43 -> 2.02; with no more passes than CUs it would only leave CUs empty).
  // LaunchHints::seqTwoMasters 0 / 1: never / always.
  const bool mm = hints.seqTwoMasters == 0 || hints.seqTwoMasters == 1 ? hints.seqTwoMasters == 1
                                                                      : p.npass > static_cast<uint32_t>(cusFor(hints));
  if (p.accel == PTW_ACCEL_PREFILTER) // (the separate mode: the worker lanes look in fp32 first)
    return mm ? launchSeqTwoMastersPrefilter(p, b, hints, stream) : launchSeqOneMasterPrefilter(p, b, hints, stream);
  if (p.seqUnitUFirst) // (scenes whose units of 64 consecutive triangles mostly fail the u test as a whole)
    return mm ? launchSeqTwoMastersUnit(p, b, hints, stream) : launchSeqOneMasterUnit(p, b, hints, stream);
  return mm ? launchSeqTwoMasters(p, b, hints, stream) : launchSeqOneMaster(p, b, hints, stream);
}
} // namespace

hipError_t launchTraceSequential(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints,
                                 hipStream_t stream, const char **variant) {
  const hipError_t e = dispatchSequential(p, b, hints, stream);
  if (variant) *variant = lastVariant();
  return e;
}

} // namespace ptw

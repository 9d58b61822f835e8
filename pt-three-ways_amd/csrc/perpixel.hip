// perpixel.hip - PERPIXEL policy, brute force: tracePerPixel (lock step) and tracePerPixelPersistent.
#include "ptw_pix_persistent.h"

namespace ptw {
using namespace ptwd;
namespace {

__global__ __launch_bounds__(kPixBlock) __attribute__((amdgpu_waves_per_eu(PTW_PIX_WAVES, PTW_PIX_WAVES))) void tracePerPixel(
    const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x]
  perPixelSample<kPixBrute>(p, b, pixStacks);
}

} // namespace
} // namespace ptw

namespace ptw {
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
  if (p.accel != PTW_ACCEL_NONE) {
    const hipError_t e = launchTraceAccel(p, b, hints, stream);
    if (variant) *variant = lastVariant();
    return e;
  }
  const bool persistent = p.pixKernel != kPixKernelLockstep;
  if (variant) *variant = persistent ? "tracePerPixelPersistent" : "tracePerPixel";
  if (hints.dryRun) return hipSuccess;
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
                       b.matTable, b.stage, b.words, b.rays, b.sampleQueue, b.triPacked);
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

} // namespace ptw

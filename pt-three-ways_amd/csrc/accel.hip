// accel.hip - the SEPARATE accelerated modes of the PERPIXEL policy: bit-identical
// samples, different work; never in the headline numbers.
#include "ptw_pix_persistent.h"

namespace ptw {
using namespace ptwd;
namespace {

// ACCELERATED mode (ptw_render_params.accel == PTW_ACCEL_BVH): the same
// sample, with Scene::intersect culled by a BVH - bit-identical results, different work.  Reported
// separately, never in the headline numbers.
__global__ __launch_bounds__(kPixBlock) void tracePerPixelBvh(const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x] levels + [kBvhStack][blockDim.x] traversal
  perPixelSample<kPixBvh>(p, b, pixStacks);
}

// ACCELERATED mode (PTW_ACCEL_PREFILTER): the same sample, with every triangle looked at in fp32 first - two
// per packed instruction (v_pk_fma_f32), operands of both triangles in one scalar register pair - and the
// reference's fp64 test only where fp32 cannot prove a rejection (PixCtxT::intersectPrefiltered,
// host/prefilter.h).  Bit-identical results; four waves per SIMD like tracePerPixel.
__global__ __launch_bounds__(kPixBlock) __attribute__((amdgpu_waves_per_eu(PTW_PIX_WAVES, PTW_PIX_WAVES))) void tracePerPixelPrefilter(
    const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x]
  perPixelSample<kPixPrefilter>(p, b, pixStacks);
}

} // namespace

hipError_t launchTraceAccel(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  if (p.accel == PTW_ACCEL_BVH) {
    setVariant("tracePerPixelBvh");
    if (hints.dryRun) return hipSuccess;
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
  if (p.accel == PTW_ACCEL_PREFILTER && p.pixKernel != kPixKernelLockstep) {
    // the persistent form (the default, like the brute-force policy's: lanes whose paths end take the next sample)
    setVariant("tracePerPixelPersistentPrefilter");
    if (hints.dryRun) return hipSuccess;
    const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
    uint64_t blocks = static_cast<uint64_t>(cusFor(hints)) * 4; // four waves per SIMD
    const uint64_t needed = (total + kPix2Block - 1) / kPix2Block;
    if (blocks > needed) blocks = needed;
    const int levels = p.maxDepth > 1 ? p.maxDepth : 1;
    const bool ldsState = p.ntri < 128;
    const size_t lds = ((static_cast<size_t>(levels) * kPix2Block * sizeof(uint32_t) + 7) & ~size_t(7)) +
                       (ldsState ? 9 * kPix2Block * sizeof(double) + 7 * kPix2Block * sizeof(uint32_t) : 0);
    hipError_t e = hipMemsetAsync(b.sampleQueue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    auto kernel = ldsState ? tracePerPixelPersistent<4, true, true> : tracePerPixelPersistent<4, false, true>;
    if (lds > 48 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(blocks)), dim3(kPix2Block), lds, stream, p, b.triGeom, b.spheres,
                       b.triCompact, b.matTable, b.stage, b.words, b.rays, b.sampleQueue, b.triPacked);
    return hipGetLastError();
  }
  if (p.accel == PTW_ACCEL_PREFILTER) {
    setVariant("tracePerPixelPrefilter");
    if (hints.dryRun) return hipSuccess;
    const uint64_t total = static_cast<uint64_t>(p.npass) * p.pixCount;
    const uint64_t spl = hints.pixSamplesPerLane > 0 ? static_cast<uint64_t>(hints.pixSamplesPerLane) : 8; // (as tracePerPixel)
    const uint32_t blocks = static_cast<uint32_t>((total + kPixBlock * spl - 1) / (kPixBlock * spl));
    const int levels = p.maxDepth > 1 ? p.maxDepth - 1 : 1;
    const size_t lds = static_cast<size_t>(levels) * kPixBlock * sizeof(uint32_t);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tracePerPixelPrefilter),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(tracePerPixelPrefilter, dim3(blocks), dim3(kPixBlock), lds, stream, p, b);
    return hipGetLastError();
  }
  return hipErrorInvalidValue;
}

} // namespace ptw

// accel.hip - the SEPARATE accelerated modes of the PERPIXEL policy: bit-identical
// samples, different work; never in the headline numbers.
#include "ptw_pix_ctx.h"

namespace ptw {
using namespace ptwd;
namespace {

// ACCELERATED mode (ptw_render_params.accel == PTW_ACCEL_BVH): the same
// sample, with Scene::intersect culled by a BVH - bit-identical results, different work.  Reported
// separately, never in the headline numbers.
__global__ __launch_bounds__(kPixBlock) void tracePerPixelBvh(const TraceParams p, const TraceBuffers b) {
  extern __shared__ uint32_t pixStacks[]; // [maxDepth][blockDim.x] levels + [kBvhStack][blockDim.x] traversal
  perPixelSample<true>(p, b, pixStacks);
}

} // namespace

hipError_t launchTraceAccel(const TraceParams &p, const TraceBuffers &b, const LaunchHints &, hipStream_t stream) {
  if (p.accel == PTW_ACCEL_BVH) {
    setVariant("tracePerPixelBvh");
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
  return hipErrorInvalidValue;
}

} // namespace ptw

// resolve_kat.hip - resolveKernel (ArrayOutput::operator+= in pass order) and the known-answer kernels.
#include "ptw_launch.h"
#include "ptw_pix_ctx.h"
#include "ptw_seq_ctx.h"

namespace ptw {
using namespace ptwd;
namespace {

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
template <int MODE> // kPixBrute, or kPixPrefilter: the same search through the fp32 prefilter (tests)
__global__ __launch_bounds__(256) void intersectBatchKernel(
    const TraceParams p, const double *__restrict__ triGeom,
    const TriShade *__restrict__ triShade, const SphereRec *__restrict__ spheres,
    const double *__restrict__ rays, uint64_t n, double *__restrict__ hits, const float *__restrict__ triPacked) {
  const uint64_t gid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  PixCtxT<MODE> ctx;
  ctx.p = &p;
  ctx.triPacked = triPacked;
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

} // namespace
} // namespace ptw

namespace ptw {
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
  auto kernel = p.accel == PTW_ACCEL_PREFILTER ? intersectBatchKernel<kPixPrefilter> : intersectBatchKernel<kPixBrute>;
  hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, p, b.triGeom,
                     b.triShade, b.spheres, rays, n, hitsOut, b.triPacked);
  return hipGetLastError();
}

} // namespace ptw

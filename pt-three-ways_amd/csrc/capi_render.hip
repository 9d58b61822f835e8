// capi_render.hip — the device half of the C ABI declared in include/ptw.h: context, scene
// upload, and the render entry points that replace dod::Scene::render
// (src/dod/Scene.cpp:197-254).  Only HIP runtime calls here; the kernels are in
// ptw_kernels.hip and the strict-fp64 host precompute in host/precompute.cpp.
#include "capi_common.h"
#include "ptw_kernels.h"

#include "../host/precompute.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

namespace ptw {
namespace {

void check(hipError_t e, const char *what) {
  if (e == hipSuccess) return;
  const int status = (e == hipErrorNoDevice || e == hipErrorInvalidDevice ||
                      e == hipErrorInsufficientDriver)
                         ? PTW_ERR_NO_DEVICE
                         : PTW_ERR_HIP;
  throw DeviceError(status, std::string(what) + ": " + hipGetErrorString(e));
}

// Owning device allocation that can grow.
template <typename T>
struct DeviceArray {
  T *ptr = nullptr;
  size_t capacity = 0;
  ~DeviceArray() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    capacity = 0;
  }
  void reserve(size_t n) {
    if (n <= capacity) return;
    release();
    check(hipMalloc(reinterpret_cast<void **>(&ptr), std::max<size_t>(n, 1) * sizeof(T)),
          "hipMalloc");
    capacity = n;
  }
  void upload(const T *src, size_t n, hipStream_t stream) {
    reserve(n);
    if (n) check(hipMemcpyAsync(ptr, src, n * sizeof(T), hipMemcpyHostToDevice, stream), "H2D");
  }
};

bool isPowerOfTwo(int v) { return v > 0 && (v & (v - 1)) == 0; }

} // namespace
} // namespace ptw

using namespace ptw;

struct ptw_context {
  int device = 0;
  bool haveScene = false;
  uint32_t ntri = 0, nsph = 0;
  double env[3] = {0, 0, 0};
  std::vector<uint32_t> triMaterial, sphMaterial; // for ptw_context_intersect
  DeviceArray<double> triGeom;
  DeviceArray<TriShade> triShade;
  DeviceArray<SphereRec> spheres;
  DeviceArray<double> triCompact, matTable;
  DeviceArray<double> specState; // parked stream rings of traceSequentialSpec
  uint32_t nmat = 0;
  DeviceArray<uint32_t> mtState, mtPos;
  DeviceArray<double> stage;
  DeviceArray<unsigned long long> sampleQueue; // work counter of the persistent kernel
  DeviceArray<unsigned long long> rays; // per-pass intersect() counters, accumulated
  uint64_t rayCarry = 0;                // counts folded in when `rays` had to grow
  std::vector<uint32_t> hostSeedStates; // kept alive for the async upload

  bool statsEnabled = false;
  struct Timed {
    hipEvent_t begin, end;
    bool trace;
  };
  std::vector<Timed> timed;
  uint64_t statSamples = 0;

  size_t stageBudgetBytes = size_t(256) << 20; // per-band staging buffer budget

  void activate() const { check(hipSetDevice(device), "hipSetDevice"); }
  // Reads back and zeroes the per-pass ray counters (synchronous).
  uint64_t drainRays() {
    if (!rays.capacity) return 0;
    std::vector<unsigned long long> host(rays.capacity);
    check(hipMemcpy(host.data(), rays.ptr, host.size() * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost),
          "D2H rays");
    check(hipMemset(rays.ptr, 0, host.size() * sizeof(unsigned long long)), "memset");
    uint64_t total = 0;
    for (auto v : host) total += v;
    return total;
  }
  void clearEvents() {
    for (auto &t : timed) {
      (void)hipEventDestroy(t.begin);
      (void)hipEventDestroy(t.end);
    }
    timed.clear();
  }
  ~ptw_context() { clearEvents(); }
};

namespace {

#define PTW_GUARD_BEGIN try {
#define PTW_GUARD_END                                                                          \
  }                                                                                            \
  catch (...) {                                                                                \
    return translateException();                                                               \
  }

void validate(const ptw_render_params &p) {
  if (p.width <= 0 || p.height <= 0) throw std::invalid_argument("width/height must be positive");
  if (static_cast<uint64_t>(p.width) * p.height > 0x7fffffffull / 3)
    throw std::invalid_argument("frame too large");
  if (p.samples_per_pixel < 0) throw std::invalid_argument("samples_per_pixel must be >= 0");
  if (p.max_depth > kMaxDepth)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "max_depth above " + std::to_string(kMaxDepth));
  if (p.first_bounce_u < 0 || p.first_bounce_v < 0)
    throw std::invalid_argument("first_bounce_u/v must be >= 0");
  if (p.rng_policy != PTW_RNG_SEQUENTIAL && p.rng_policy != PTW_RNG_PERPIXEL)
    throw std::invalid_argument("unknown rng_policy");
  if (p.row_end < p.row_begin || p.row_begin < 0 || p.row_end > p.height)
    throw std::invalid_argument("bad row window");
}

TraceParams makeTraceParams(const ptw_context &ctx, const ptw_camera &cam,
                            const ptw_render_params &p) {
  TraceParams t;
  std::memset(&t, 0, sizeof t);
  t.cam = cam;
  std::memcpy(t.env, ctx.env, sizeof t.env);
  const int nU = p.first_bounce_u, nV = p.first_bounce_v;
  // Vec3::operator/(double b): reciprocal = 1.0 / b with b = double(nU * nV)
  t.invFirstBounce = 1.0 / static_cast<double>(nU * nV);
  t.invU = nU ? 1.0 / static_cast<double>(nU) : 0.0;
  t.invV = nV ? 1.0 / static_cast<double>(nV) : 0.0;
  t.uPow2 = isPowerOfTwo(nU);
  t.vPow2 = isPowerOfTwo(nV);
  t.ntri = ctx.ntri;
  t.nsph = ctx.nsph;
  t.nmat = ctx.nmat;
  t.width = p.width;
  t.height = p.height;
  t.maxDepth = p.max_depth;
  t.fbU = nU;
  t.fbV = nV;
  t.preview = p.preview;
  t.rngPolicy = p.rng_policy;
  t.passSeedBase = static_cast<uint32_t>(p.seed + p.first_pass);
  t.npix = static_cast<uint32_t>(p.width) * static_cast<uint32_t>(p.height);
  t.npass = static_cast<uint32_t>(p.samples_per_pixel);
  return t;
}

// Enqueues the whole render on `stream`.  `betweenBands`, when set, is called after each
// band's launches have been enqueued with the samples enqueued so far; returning true cancels.
template <typename BetweenBands>
void enqueueRender(ptw_context &ctx, const ptw_camera &cam, const ptw_render_params &p,
                   double *dRgb, uint32_t *dCounts, uint32_t *dWords, hipStream_t stream,
                   BetweenBands &&betweenBands) {
  validate(p);
  if (!ctx.haveScene) throw std::invalid_argument("no scene set on this context");
  ctx.activate();
  const uint32_t npass = static_cast<uint32_t>(p.samples_per_pixel);
  if (npass == 0) return;
  TraceParams t = makeTraceParams(ctx, cam, p);
  const bool sequential = p.rng_policy == PTW_RNG_SEQUENTIAL;

  uint32_t pixFirst = 0, pixLast = t.npix;
  if (!sequential && p.row_end > p.row_begin) {
    pixFirst = static_cast<uint32_t>(p.row_begin) * p.width;
    pixLast = static_cast<uint32_t>(p.row_end) * p.width;
  }
  const uint32_t pixTotal = pixLast - pixFirst;

  // Band size: the staging buffer holds npass x bandPix x 3 doubles.
  uint64_t bandPix = ctx.stageBudgetBytes / (static_cast<uint64_t>(npass) * 24);
  bandPix = std::max<uint64_t>(bandPix, 64);
  bandPix = std::min<uint64_t>(bandPix, pixTotal);
  ctx.stage.reserve(static_cast<size_t>(npass) * bandPix * 3);
  if (npass > ctx.rays.capacity) {
    ctx.rayCarry += ctx.drainRays();
    ctx.rays.reserve(npass);
    check(hipMemset(ctx.rays.ptr, 0, npass * sizeof(unsigned long long)), "memset");
  }

  if (sequential) {
    // std::mt19937 rng(seed + curSample++), Scene.cpp:211: seed the generators on the host
    ctx.hostSeedStates.resize(static_cast<size_t>(npass) * kMtWords);
    for (uint32_t k = 0; k < npass; ++k)
      seedMt19937(t.passSeedBase + k, &ctx.hostSeedStates[static_cast<size_t>(k) * kMtWords]);
    ctx.mtState.upload(ctx.hostSeedStates.data(), ctx.hostSeedStates.size(), stream);
    std::vector<uint32_t> pos(npass, kMtDoubles); // 312 = "regenerate before the first draw"
    ctx.mtPos.reserve(npass);
    check(hipMemcpyAsync(ctx.mtPos.ptr, pos.data(), npass * sizeof(uint32_t),
                         hipMemcpyHostToDevice, stream),
          "H2D mtPos");
    check(hipStreamSynchronize(stream), "sync after seeding"); // `pos` is a local
    ctx.specState.reserve(static_cast<size_t>(npass) * kSpecStateDoubles);
  }

  TraceBuffers b;
  std::memset(&b, 0, sizeof b);
  b.triGeom = ctx.triGeom.ptr;
  b.triShade = ctx.triShade.ptr;
  b.spheres = ctx.spheres.ptr;
  b.triCompact = ctx.triCompact.ptr;
  b.matTable = ctx.matTable.ptr;
  b.mtState = ctx.mtState.ptr;
  b.mtPos = ctx.mtPos.ptr;
  b.stage = ctx.stage.ptr;
  b.words = dWords;
  b.rays = ctx.rays.ptr;
  ctx.sampleQueue.reserve(1);
  b.sampleQueue = ctx.sampleQueue.ptr;
  b.specState = sequential ? ctx.specState.ptr : nullptr;

  auto timedLaunch = [&](bool trace, auto &&launch) {
    if (!ctx.statsEnabled) {
      check(launch(), "kernel launch");
      return;
    }
    ptw_context::Timed ev;
    ev.trace = trace;
    check(hipEventCreate(&ev.begin), "hipEventCreate");
    check(hipEventCreate(&ev.end), "hipEventCreate");
    check(hipEventRecord(ev.begin, stream), "hipEventRecord");
    check(launch(), "kernel launch");
    check(hipEventRecord(ev.end, stream), "hipEventRecord");
    ctx.timed.push_back(ev);
  };

  uint64_t done = 0;
  for (uint32_t begin = pixFirst; begin < pixLast; begin += static_cast<uint32_t>(bandPix)) {
    t.pixBegin = begin;
    t.pixCount = static_cast<uint32_t>(std::min<uint64_t>(bandPix, pixLast - begin));
    t.firstBand = begin == pixFirst;
    if (sequential)
      timedLaunch(true, [&] { return launchTraceSequential(t, b, stream); });
    else
      timedLaunch(true, [&] { return launchTracePerPixel(t, b, stream); });
    timedLaunch(false, [&] {
      return launchResolve(ctx.stage.ptr, npass, t.pixBegin, t.pixCount, dRgb, dCounts, stream);
    });
    done += static_cast<uint64_t>(t.pixCount) * npass;
    ctx.statSamples += static_cast<uint64_t>(t.pixCount) * npass;
    if (betweenBands(done, static_cast<uint64_t>(pixTotal) * npass)) break;
  }
}

} // namespace

extern "C" {

int ptw_context_create(int32_t device, ptw_context **out) {
  if (!out) return invalid("out");
  PTW_GUARD_BEGIN
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw DeviceError(PTW_ERR_NO_DEVICE,
                      std::string("no HIP device available (the hip way has no CPU fallback)") +
                          (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : ""));
  if (device < 0 || device >= count)
    throw DeviceError(PTW_ERR_NO_DEVICE, "HIP device ordinal out of range");
  auto ctx = std::make_unique<ptw_context>();
  ctx->device = device;
  // Staging budget override (MiB); small values force many bands - used by the tests to prove
  // the result does not depend on how the frame is cut into launches.
  if (const char *mb = std::getenv("PTW_STAGE_BUDGET_MB")) {
    const long v = std::strtol(mb, nullptr, 10);
    if (v > 0) ctx->stageBudgetBytes = static_cast<size_t>(v) << 20;
  }
  if (const char *kb = std::getenv("PTW_STAGE_BUDGET_KB")) {
    const long v = std::strtol(kb, nullptr, 10);
    if (v > 0) ctx->stageBudgetBytes = static_cast<size_t>(v) << 10;
  }
  ctx->activate();
  *out = ctx.release();
  return PTW_OK;
  PTW_GUARD_END
}

void ptw_context_destroy(ptw_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  delete ctx;
}

int ptw_context_set_scene(ptw_context *ctx, const ptw_scene_view *scene) {
  if (!ctx || !scene) return invalid("null pointer");
  PTW_GUARD_BEGIN
  ctx->activate();
  DeviceSceneData data = precomputeScene(*scene);
  ctx->triGeom.upload(data.triGeom.data(), data.triGeom.size(), nullptr);
  ctx->triShade.upload(data.triShade.data(), data.triShade.size(), nullptr);
  ctx->spheres.upload(data.spheres.data(), data.spheres.size(), nullptr);
  ctx->triCompact.upload(data.triCompact.data(), data.triCompact.size(), nullptr);
  ctx->matTable.upload(data.matTable.data(), data.matTable.size(), nullptr);
  ctx->nmat = scene->num_materials;
  check(hipStreamSynchronize(nullptr), "scene upload");
  ctx->ntri = scene->num_triangles;
  ctx->nsph = scene->num_spheres;
  std::memcpy(ctx->env, data.environment, sizeof ctx->env);
  ctx->triMaterial = std::move(data.triMaterial);
  ctx->sphMaterial = std::move(data.sphMaterial);
  ctx->haveScene = true;
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_render(ptw_context *ctx, const ptw_camera *camera, const ptw_render_params *params,
                       void *d_rgb_sum, void *d_counts, void *d_words, void *hip_stream) {
  if (!ctx || !camera || !params || !d_rgb_sum || !d_counts) return invalid("null pointer");
  PTW_GUARD_BEGIN
  enqueueRender(*ctx, *camera, *params, static_cast<double *>(d_rgb_sum),
                static_cast<uint32_t *>(d_counts), static_cast<uint32_t *>(d_words),
                static_cast<hipStream_t>(hip_stream), [](uint64_t, uint64_t) { return false; });
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_enable_stats(ptw_context *ctx, int32_t enable) {
  if (!ctx) return invalid("ctx");
  ctx->statsEnabled = enable != 0;
  return PTW_OK;
}

int ptw_context_get_stats(ptw_context *ctx, ptw_kernel_stats *out, int32_t reset) {
  if (!ctx || !out) return invalid("null pointer");
  PTW_GUARD_BEGIN
  ctx->activate();
  std::memset(out, 0, sizeof *out);
  for (auto &t : ctx->timed) {
    check(hipEventSynchronize(t.end), "hipEventSynchronize");
    float ms = 0;
    check(hipEventElapsedTime(&ms, t.begin, t.end), "hipEventElapsedTime");
    if (t.trace) {
      out->trace_launches++;
      out->trace_ms += ms;
    } else {
      out->resolve_launches++;
      out->resolve_ms += ms;
    }
  }
  out->samples = ctx->statSamples;
  ctx->rayCarry += ctx->drainRays();
  out->rays = ctx->rayCarry;
  if (reset) {
    ctx->clearEvents();
    ctx->statSamples = 0;
    ctx->rayCarry = 0;
  }
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_intersect(ptw_context *ctx, const double *rays, uint64_t n, double *hits_out) {
  if (!ctx || (!rays && n) || (!hits_out && n)) return invalid("null pointer");
  PTW_GUARD_BEGIN
  if (!ctx->haveScene) throw std::invalid_argument("no scene set on this context");
  ctx->activate();
  if (n == 0) return PTW_OK;
  DeviceArray<double> dRays, dHits;
  dRays.upload(rays, n * 6, nullptr);
  dHits.reserve(n * 9);
  ptw_camera cam;
  std::memset(&cam, 0, sizeof cam);
  ptw_render_params rp;
  std::memset(&rp, 0, sizeof rp);
  rp.width = rp.height = 1;
  rp.first_bounce_u = rp.first_bounce_v = 1;
  TraceParams t = makeTraceParams(*ctx, cam, rp);
  TraceBuffers b;
  std::memset(&b, 0, sizeof b);
  b.triGeom = ctx->triGeom.ptr;
  b.triShade = ctx->triShade.ptr;
  b.spheres = ctx->spheres.ptr;
  b.triCompact = ctx->triCompact.ptr;
  b.matTable = ctx->matTable.ptr;
  check(launchIntersectBatch(t, b, dRays.ptr, n, dHits.ptr, nullptr), "intersect launch");
  check(hipMemcpy(hits_out, dHits.ptr, n * 9 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
  // the kernel reports the combined primitive index; the ABI promises the material index
  for (uint64_t i = 0; i < n; ++i) {
    double *h = hits_out + i * 9;
    if (h[0] < 0) continue;
    const uint32_t idx = static_cast<uint32_t>(h[8]);
    h[8] = idx < ctx->nsph ? ctx->sphMaterial[idx] : ctx->triMaterial[idx - ctx->nsph];
  }
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_rng_doubles(ptw_context *ctx, int32_t rng_policy, uint32_t seed, uint32_t pixel,
                            uint32_t n, double *out) {
  if (!ctx || (!out && n)) return invalid("null pointer");
  PTW_GUARD_BEGIN
  ctx->activate();
  if (n == 0) return PTW_OK;
  uint32_t state[kMtWords];
  seedMt19937(seed, state);
  DeviceArray<uint32_t> dState;
  DeviceArray<double> dOut;
  dState.upload(state, kMtWords, nullptr);
  dOut.reserve(n);
  check(launchRngKat(rng_policy, dState.ptr, seed, pixel, n, dOut.ptr, nullptr), "rng kat launch");
  check(hipMemcpy(out, dOut.ptr, n * sizeof(double), hipMemcpyDeviceToHost), "D2H");
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_render(const ptw_scene_view *scene, const ptw_camera *camera,
               const ptw_render_params *params, double *rgb_sum, uint32_t *counts,
               ptw_progress_fn progress, void *user) {
  if (!scene || !camera || !params || !rgb_sum || !counts) return invalid("null pointer");
  ptw_context *raw = nullptr;
  int rc = ptw_context_create(params->device, &raw);
  if (rc != PTW_OK) return rc;
  std::unique_ptr<ptw_context, void (*)(ptw_context *)> ctx(raw, ptw_context_destroy);
  rc = ptw_context_set_scene(ctx.get(), scene);
  if (rc != PTW_OK) return rc;
  PTW_GUARD_BEGIN
  validate(*params);
  const size_t npix = static_cast<size_t>(params->width) * params->height;
  DeviceArray<double> dRgb;
  DeviceArray<uint32_t> dCounts;
  dRgb.upload(rgb_sum, npix * 3, nullptr);
  dCounts.upload(counts, npix, nullptr);
  bool cancelled = false;
  enqueueRender(*ctx, *camera, *params, dRgb.ptr, dCounts.ptr, nullptr, nullptr,
                [&](uint64_t done, uint64_t total) {
                  if (!progress) return false;
                  check(hipStreamSynchronize(nullptr), "band");
                  cancelled = progress(user, done, total) != 0;
                  return cancelled;
                });
  check(hipStreamSynchronize(nullptr), "render");
  check(hipMemcpy(rgb_sum, dRgb.ptr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
  check(hipMemcpy(counts, dCounts.ptr, npix * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H");
  if (cancelled) {
    setLastError("cancelled by the progress callback");
    return PTW_ERR_INVALID;
  }
  return PTW_OK;
  PTW_GUARD_END
}

} // extern "C"

// capi_render.hip — the device half of the C ABI declared in include/ptw.h: context, scene
// upload, and the render entry points that replace dod::Scene::render
// (src/dod/Scene.cpp:197-254).  Only HIP runtime calls here; the kernels are in the kernel
// files of this directory (ptw_kernels.h lists their launchers) and the strict-fp64 host precompute in
// host/precompute.cpp.
#include "capi_common.h"
#include "ptw_kernels.h"

#include "../host/bvh.h"
#include "../host/prefilter.h"
#include "../host/precompute.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
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
  // accelerated mode (PTW_ACCEL_BVH): built with the scene
  DeviceArray<BvhNode> bvhNodes;
  DeviceArray<double> bvhLeafGeom;
  DeviceArray<uint32_t> bvhLeafIndex;
  // accelerated mode (PTW_ACCEL_PREFILTER): the fp32 pair records, built with the scene
  DeviceArray<float> triPacked;
  bool prefilterUsable = false;
  bool unitsCoherent = false; // unitUSkipFraction(scene) >= kUnitUFirstThreshold: the worker waves' unit-level u-first early-out
  // SEQUENTIAL, scenes of at most 64 triangles, more passes than CUs: the small-scene kernel a timed trial chose
  // (ptw_debug_options.seq_small_kernel's values 1 / 2; 0 = none) and what it was measured for
  int seqSmallChoice = 0;
  uint64_t seqSmallChoiceKey = 0;
  DeviceArray<double> specState; // parked stream rings of traceSequentialSpec
  uint32_t nmat = 0;
  DeviceArray<uint32_t> mtState, mtPos;
  DeviceArray<double> stage;
  DeviceArray<unsigned long long> sampleQueue; // work counter of the persistent kernel
  DeviceArray<unsigned long long> rays; // per-pass intersect() counters, accumulated
  uint64_t rayCarry = 0;                // counts folded in when `rays` had to grow
  // Host sources of the asynchronous uploads of a render; they live in the context because
  // the upload may still be in flight when ptw_context_render returns (one render in flight per
  // context, see include/ptw.h).
  std::vector<uint32_t> hostSeedStates, hostPos;
  hipEvent_t uploadsDone = nullptr; // recorded after a render's uploads: the host vectors are free again
  // PERPIXEL: which of the two kernels this scene + frame shape runs faster on (timed once, see
  // ptw_context_calibrate / calibratePixKernel); 0 = not decided yet
  uint64_t pixChoiceKey = 0;
  int pixChoice = 0;
  uint64_t sceneGeneration = 0;
  char traceKernel[64] = ""; // variant name of the last trace launch (a copy: the launcher's
                             // string lives in thread-local storage of the launching thread)

  // tests / A-B runs only (ptw_context_set_debug): what the launchers are asked to do differently
  ptw_debug_options debug;

  bool statsEnabled = false;
  struct Timed {
    hipEvent_t begin, end;
    bool trace;
  };
  std::vector<Timed> timed;
  uint64_t statSamples = 0;

  // Per-band staging buffer budget: npass x bandPix x 24 B.  4 GiB keeps the headline frame
  // (1024 x 1024 @ 256 spp = 6.4 GB staged) to two launches; the part has 288 GB.
  size_t stageBudgetBytes = size_t(4) << 30;
  // contexts that share this device within one render (ptw_render_ex, share_device = 2): the clamp
  // of the budget to the free memory is divided among them
  int deviceShare = 1;

  ptw_context() { ptw_debug_defaults(&debug); }
  LaunchHints hints() const {
    LaunchHints h;
    h.seqTwoMasters = debug.seq_two_masters;
    h.seqLdsTables = debug.seq_lds_tables, h.seqSmallKernel = debug.seq_small_kernel;
    for (int i = 0; i < 3; ++i) h.seqUnits[i] = debug.seq_units[i];
    h.pixSamplesPerLane = debug.pix_samples_per_lane, h.pixWavesPerSimd = debug.pix_waves_per_simd;
    return h;
  }
  void activate() const { check(hipSetDevice(device), "hipSetDevice"); }
  // Reads back and zeroes the per-pass ray counters (synchronous).
  uint64_t drainRays() {
    if (!rays.capacity) return 0;
    check(hipDeviceSynchronize(), "hipDeviceSynchronize"); // counters of renders on any stream
    std::vector<unsigned long long> host(rays.capacity);
    check(hipMemcpy(host.data(), rays.ptr, host.size() * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost),
          "D2H rays");
    check(hipMemset(rays.ptr, 0, host.size() * sizeof(unsigned long long)), "memset");
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
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
  ~ptw_context() {
    clearEvents();
    if (uploadsDone) (void)hipEventDestroy(uploadsDone);
  }
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
  // A zero fan-out makes the reference divide 0 by 0 (Scene.cpp:178); it is not a render.
  if (p.first_bounce_u < 1 || p.first_bounce_v < 1)
    throw std::invalid_argument("first_bounce_u/v must be >= 1");
  if (static_cast<int64_t>(p.first_bounce_u) * p.first_bounce_v > (1 << 20))
    throw std::invalid_argument("first_bounce_u * first_bounce_v too large");
  if (p.rng_policy != PTW_RNG_SEQUENTIAL && p.rng_policy != PTW_RNG_PERPIXEL)
    throw std::invalid_argument("unknown rng_policy");
  if (p.row_end < p.row_begin || p.row_begin < 0 || p.row_end > p.height)
    throw std::invalid_argument("bad row window");
  if (p.row_stride < 0 || p.row_phase < 0 || (p.row_stride > 1 && p.row_phase >= p.row_stride) ||
      (p.row_stride <= 1 && p.row_phase != 0))
    throw std::invalid_argument("bad row_stride / row_phase");
  if (p.accel != PTW_ACCEL_NONE && p.accel != PTW_ACCEL_BVH && p.accel != PTW_ACCEL_PREFILTER)
    throw std::invalid_argument("unknown accel mode");
  if (p.pix_kernel != PTW_PIX_KERNEL_AUTO && p.pix_kernel != PTW_PIX_KERNEL_LOCKSTEP &&
      p.pix_kernel != PTW_PIX_KERNEL_PERSISTENT)
    throw std::invalid_argument("unknown pix_kernel");
  if (p.accel == PTW_ACCEL_BVH && p.rng_policy != PTW_RNG_PERPIXEL)
    throw DeviceError(PTW_ERR_UNSUPPORTED,
                      "PTW_ACCEL_BVH needs PTW_RNG_PERPIXEL (the SEQUENTIAL kernels search the scene cooperatively, a lane "
                      "per primitive: there is no ray to walk a hierarchy with)");

  // Under PTW_RNG_SEQUENTIAL the pixels of a pass share one stream, consumed in row-major order: the
  // only window that means anything is a PREFIX of the frame (rows [0, row_end): exactly what the
  // full render produces for those rows - used for timed sub-runs of very large frames).
  if (p.rng_policy == PTW_RNG_SEQUENTIAL && (p.row_begin != 0 || p.row_stride > 1))
    throw DeviceError(PTW_ERR_UNSUPPORTED,
                      "a row window needs PTW_RNG_PERPIXEL: under PTW_RNG_SEQUENTIAL the pixels of a "
                      "pass share one stream (shard by first_pass instead; only a prefix [0, row_end) "
                      "of the frame can be rendered on its own)");
}

// The image rows a render covers: first row, row stride, number of rows.
struct RowSet {
  int first, stride, count;
};
RowSet rowsOf(const ptw_render_params &p) {
  const bool all = p.row_begin == 0 && p.row_end == 0;
  const int begin = all ? 0 : p.row_begin, end = all ? p.height : p.row_end;
  const int stride = p.row_stride > 1 ? p.row_stride : 1;
  // first row >= begin that is congruent to row_phase
  int first = begin + ((p.row_phase - begin % stride) + stride) % stride;
  RowSet r;
  r.first = first;
  r.stride = stride;
  r.count = first < end ? (end - first + stride - 1) / stride : 0;
  return r;
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
  t.rowFirst = 0;
  t.rowStride = 1;
  t.accel = p.accel;
  t.seqUnitUFirst = ctx.debug.seq_unit_ufirst >= 0 ? (ctx.debug.seq_unit_ufirst != 0) : ctx.unitsCoherent;
#if defined(PTW_PROFILE_PHASES) && PTW_PROFILE_PHASES
  if (const char *v = std::getenv("PTW_PIX_COUNT_SLOTS")) t.padA = v[0] == '1'; // (the prof build's lane-slot counter)
#endif
  return t;
}

// PERPIXEL policy: lock-step or persistent kernel (perpixel.hip, launchTracePerPixel)?  Which
// one is faster depends on how uniformly long the scene's paths are, which no host-side number
// says.  ptw_context_calibrate() times a trial of both - about two million samples each, image rows
// spread over the whole frame, all passes' worth of lanes - and the context remembers the winner
// under a key of what the decision depends on; PTW_PIX_KERNEL_AUTO resolves to it.  The trial's
// outputs land in the staging buffer (overwritten by the next render) and nowhere else.
// ptw_context_render never calibrates (it is asynchronous); ptw_render / ptw_render_ex do, once, for
// renders of 16 M samples or more.
constexpr uint64_t kCalibrateFromSamples = 16ull << 20;

uint64_t pixChoiceKeyOf(const ptw_context &ctx, const TraceParams &t, const RowSet &rows) {
  uint64_t key = 1469598103934665603ull; // FNV-1a
  auto mix = [&](const void *data, size_t n) {
    const unsigned char *c = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) key = (key ^ c[i]) * 1099511628211ull;
  };
  mix(&ctx.sceneGeneration, sizeof ctx.sceneGeneration);
  mix(&t.cam, sizeof t.cam);
  const int32_t shape[7] = {t.width, t.height, t.maxDepth, t.fbU, t.fbV, t.preview, rows.stride};
  mix(shape, sizeof shape);
  return key;
}

// ... and of the SEQUENTIAL small-scene choice: the scene, the camera, the frame's shape AND the pass count
uint64_t seqSmallChoiceKeyOf(const ptw_context &ctx, const TraceParams &t) {
  RowSet none;
  none.first = 0, none.count = 0, none.stride = static_cast<int>(t.npass);
  return pixChoiceKeyOf(ctx, t, none) ^ 0x9e3779b97f4a7c15ull;
}

// The timed trial of the two small-scene SEQUENTIAL kernels (blocks until it has run): the first pixels of the
// frame from freshly seeded generators, each kernel once; the render that follows seeds again.
int calibrateSeqSmall(ptw_context &ctx, const TraceParams &t, const TraceBuffers &b, uint32_t pixTotal, size_t stageDoubles,
                      hipStream_t stream) {
  LaunchHints hints = ctx.hints();
  if (!seqSmallKernelIsOpen(t, hints) || !b.specState) return 0;
  TraceParams tt = t;
  tt.pixBegin = 0;
  tt.firstBand = 1;
  tt.pixCount = static_cast<uint32_t>(std::min<uint64_t>(std::min<uint32_t>(pixTotal, 192), stageDoubles / (3ull * t.npass)));
  if (tt.pixCount < 16) return 0; // (too small a frame to tell, or to matter)
  TraceBuffers bb = b;
  bb.rays = nullptr, bb.words = nullptr, bb.picks = nullptr;
  // the three candidates (ptw_debug_options.seq_small_kernel's values): one wave per pass, four speculating waves (a
  // CU per pass, in turns), two speculating waves (two workgroups per CU)
  constexpr int kKinds = 3;
  const int kind[kKinds] = {1, 2, 4};
  hipEvent_t ev[kKinds + 1];
  for (auto &e : ev) check(hipEventCreate(&e), "hipEventCreate");
  float ms[kKinds] = {0, 0, 0};
  try {
    for (int warm = 1; warm >= 0; --warm) { // first pass: code objects loaded, tiny; second: timed
      TraceParams run = tt;
      if (warm) run.pixCount = 4;
      check(hipEventRecord(ev[0], stream), "hipEventRecord");
      for (int k = 0; k < kKinds; ++k) {
        hints.seqSmallKernel = kind[k];
        check(launchTraceSequential(run, bb, hints, stream), "trial launch");
        check(hipEventRecord(ev[k + 1], stream), "hipEventRecord");
      }
    }
    check(hipEventSynchronize(ev[kKinds]), "hipEventSynchronize");
    for (int k = 0; k < kKinds; ++k) check(hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]), "hipEventElapsedTime");
  } catch (...) {
    for (auto &e : ev) (void)hipEventDestroy(e);
    throw;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  int best = 0;
  for (int k = 1; k < kKinds; ++k)
    if (ms[k] < ms[best]) best = k;
  ctx.seqSmallChoiceKey = seqSmallChoiceKeyOf(ctx, t);
  ctx.seqSmallChoice = kind[best];
  if (ctx.debug.trace)
    std::fprintf(stderr, "ptw: SEQUENTIAL small-scene trial (%u pixels x %u passes): one wave per pass %.3f ms, four speculating "
                         "waves %.3f ms, two speculating waves %.3f ms -> seq_small_kernel %d\n", tt.pixCount, tt.npass, ms[0], ms[1],
                 ms[2], ctx.seqSmallChoice);
  return ctx.seqSmallChoice;
}

// What PTW_PIX_KERNEL_AUTO means for this launch: the calibrated choice, else the persistent kernel.
int resolvePixKernel(const ptw_context &ctx, const ptw_render_params &p, const TraceParams &t, const RowSet &rows) {
  if (p.pix_kernel != PTW_PIX_KERNEL_AUTO) return p.pix_kernel;
  if (ctx.pixChoice != 0 && ctx.pixChoiceKey == pixChoiceKeyOf(ctx, t, rows)) return ctx.pixChoice;
  return kPixKernelPersistent;
}

// The timed trial (blocks until it has run).  `stageDoubles`: capacity of b.stage.
int calibratePixKernel(ptw_context &ctx, const TraceParams &t, const TraceBuffers &b, const RowSet &rows,
                       size_t stageDoubles, hipStream_t stream) {
  if (t.accel != PTW_ACCEL_NONE || t.rngPolicy != PTW_RNG_PERPIXEL || rows.count == 0 || t.npass == 0)
    return kPixKernelAuto;
  TraceParams tt = t;
  tt.npass = std::min<uint32_t>(t.npass, 32);
  uint32_t trialRows = static_cast<uint32_t>(std::max<uint64_t>(1, (2ull << 20) / (static_cast<uint64_t>(t.width) * tt.npass)));
  trialRows = std::min<uint32_t>(trialRows, static_cast<uint32_t>(rows.count));
  tt.rowFirst = rows.first;
  tt.rowStride = rows.stride * std::max<int>(1, rows.count / static_cast<int>(trialRows)); // spread over the frame
  tt.pixBegin = 0;
  // The trial writes tt.npass x tt.pixCount x 3 doubles of staged radiance: never more than the
  // staging buffer holds (a tiny staging budget can leave less than one image row per pass).
  const uint64_t fits = stageDoubles / (3ull * tt.npass);
  tt.pixCount = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(trialRows) * static_cast<uint32_t>(t.width), fits));
  if (tt.pixCount == 0) return kPixKernelAuto;
  TraceBuffers bb = b;
  bb.rays = nullptr;  // the trial is not part of the render's statistics
  bb.words = nullptr;
  const LaunchHints hints = ctx.hints();
  hipEvent_t ev[3];
  for (auto &e : ev) check(hipEventCreate(&e), "hipEventCreate");
  float ms[2] = {0, 0};
  try {
    for (int warm = 1; warm >= 0; --warm) { // first pass: code objects loaded, tiny; second: timed
      TraceParams run = tt;
      if (warm) run.pixCount = std::min<uint32_t>(run.pixCount, 256), run.npass = 1;
      check(hipEventRecord(ev[0], stream), "hipEventRecord");
      run.pixKernel = kPixKernelLockstep;
      check(launchTracePerPixel(run, bb, hints, stream), "trial launch");
      check(hipEventRecord(ev[1], stream), "hipEventRecord");
      run.pixKernel = kPixKernelPersistent;
      check(launchTracePerPixel(run, bb, hints, stream), "trial launch");
      check(hipEventRecord(ev[2], stream), "hipEventRecord");
    }
    check(hipEventSynchronize(ev[2]), "hipEventSynchronize");
    check(hipEventElapsedTime(&ms[0], ev[0], ev[1]), "hipEventElapsedTime");
    check(hipEventElapsedTime(&ms[1], ev[1], ev[2]), "hipEventElapsedTime");
  } catch (...) {
    for (auto &e : ev) (void)hipEventDestroy(e);
    throw;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  ctx.pixChoiceKey = pixChoiceKeyOf(ctx, t, rows);
  ctx.pixChoice = ms[0] < ms[1] ? kPixKernelLockstep : kPixKernelPersistent;
  if (ctx.debug.trace)
    std::fprintf(stderr, "ptw: PERPIXEL trial (%u pixels x %u passes): lock-step %.3f ms, persistent %.3f ms -> %s\n",
                 tt.pixCount, tt.npass, ms[0], ms[1], ctx.pixChoice == kPixKernelLockstep ? "lock-step" : "persistent");
  return ctx.pixChoice;
}

// Enqueues the whole render on `stream`.  `betweenBands`, when set, is called after each
// band's launches have been enqueued with the band's local pixel range and the samples enqueued so
// far; returning true cancels.  `minBands` > 1 cuts the frame into at least that many bands.
// `calibrate`: nullptr - render; otherwise nothing is rendered: the PERPIXEL kernel trial runs on the
// buffers the render would use and *calibrate receives the winner (ptw_context_calibrate).
template <typename BetweenBands>
void enqueueRender(ptw_context &ctx, const ptw_camera &cam, const ptw_render_params &p,
                   double *dRgb, uint32_t *dCounts, uint32_t *dWords, hipStream_t stream,
                   int minBands, BetweenBands &&betweenBands, int *calibrate = nullptr) {
  if (calibrate) *calibrate = kPixKernelAuto;
  validate(p);
  if (!ctx.haveScene) throw std::invalid_argument("no scene set on this context");
  if (p.accel == PTW_ACCEL_PREFILTER && p.rng_policy == PTW_RNG_SEQUENTIAL && ctx.ntri <= 128)
    throw DeviceError(PTW_ERR_UNSUPPORTED,
                      "PTW_ACCEL_PREFILTER under PTW_RNG_SEQUENTIAL is a form of the worker-wave kernels: scenes beyond 128 "
                      "triangles (a smaller scene's search is a handful of instructions per lane)");
  if (p.accel == PTW_ACCEL_PREFILTER && !(ctx.prefilterUsable && prefilterAcceptsOrigin(cam.centre, cam.aperture_radius)))
    throw DeviceError(PTW_ERR_UNSUPPORTED,
                      "PTW_ACCEL_PREFILTER: the scene (or the camera) has coordinates that are not finite or beyond 1e12 - "
                      "the fp32 prefilter's products could overflow (host/prefilter.h)");
  ctx.activate();
  const uint32_t npass = static_cast<uint32_t>(p.samples_per_pixel);
  if (npass == 0) return;
  TraceParams t = makeTraceParams(ctx, cam, p);
  const bool sequential = p.rng_policy == PTW_RNG_SEQUENTIAL;

  const RowSet rows = rowsOf(p);
  if (rows.count == 0) return; // an empty shard contributes nothing
  t.rowFirst = rows.first;
  t.rowStride = rows.stride;
  const uint32_t pixTotal = static_cast<uint32_t>(rows.count) * static_cast<uint32_t>(p.width);

  // Band size: the staging buffer holds npass x bandPix x 3 doubles.  The budget is clamped to
  // half of what the device has free right now (plus what this context already holds for staging):
  // a frame that does not fit is cut into more bands instead of failing in hipMalloc.
  size_t budget = ctx.stageBudgetBytes;
  {
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
      const size_t held = ctx.stage.capacity * sizeof(double);
      const size_t share = static_cast<size_t>(std::max(1, ctx.deviceShare));
      budget = std::min(budget, std::max<size_t>((freeB / share + held) / 2, size_t(1) << 20));
    }
  }
  uint64_t bandPix = budget / (static_cast<uint64_t>(npass) * 24);
  if (minBands > 1) bandPix = std::min<uint64_t>(bandPix, (pixTotal + minBands - 1) / minBands);
  bandPix = std::max<uint64_t>(bandPix, 64);
  bandPix = std::min<uint64_t>(bandPix, pixTotal);
  LaunchHints hints = ctx.hints();
  // Kernels that left the tree (LAB.md: the paired two-master form, several CUs per pass): asking for one is
  // an error, not a silent run of the default dispatch under the old label.
  if (ctx.debug.seq_pairing == 1 || ctx.debug.gang_groups > 0)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "ptw_debug_options.seq_pairing / gang_groups: those experimental kernels were retired "
                                           "in round 6 (LAB.md names the commit that holds them)");
  if (ctx.debug.d_picks && !sequential)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "the pick checksum (ptw_debug_options.d_picks) needs PTW_RNG_SEQUENTIAL");
  // equal bands (the last one is not a sliver)
  const uint64_t nBands = (pixTotal + bandPix - 1) / bandPix;
  bandPix = (pixTotal + nBands - 1) / nBands;
  ctx.stage.reserve(static_cast<size_t>(npass) * bandPix * 3);
  if (npass > ctx.rays.capacity) {
    ctx.rayCarry += ctx.drainRays();
    ctx.rays.reserve(npass);
    check(hipMemset(ctx.rays.ptr, 0, npass * sizeof(unsigned long long)), "memset");
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }

  if (sequential) {
    // std::mt19937 rng(seed + curSample++), Scene.cpp:211: seed the generators on the host
    if (ctx.uploadsDone) // an earlier render's upload may still be reading the host vectors
      check(hipEventSynchronize(ctx.uploadsDone), "hipEventSynchronize");
    else
      check(hipEventCreateWithFlags(&ctx.uploadsDone, hipEventDisableTiming), "hipEventCreate");
    ctx.hostSeedStates.resize(static_cast<size_t>(npass) * kMtWords);
    for (uint32_t k = 0; k < npass; ++k)
      seedMt19937(t.passSeedBase + k, &ctx.hostSeedStates[static_cast<size_t>(k) * kMtWords]);
    ctx.mtState.upload(ctx.hostSeedStates.data(), ctx.hostSeedStates.size(), stream);
    ctx.hostPos.assign(npass, kMtDoubles); // 312 = "regenerate before the first draw"
    ctx.mtPos.upload(ctx.hostPos.data(), npass, stream);
    check(hipEventRecord(ctx.uploadsDone, stream), "hipEventRecord");
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
  b.picks = sequential ? static_cast<uint32_t *>(ctx.debug.d_picks) : nullptr;
  b.rays = ctx.rays.ptr;
  ctx.sampleQueue.reserve(1);
  b.sampleQueue = ctx.sampleQueue.ptr;
  b.specState = sequential ? ctx.specState.ptr : nullptr;
  b.bvhNodes = ctx.bvhNodes.ptr;
  b.bvhLeafGeom = ctx.bvhLeafGeom.ptr;
  b.bvhLeafIndex = ctx.bvhLeafIndex.ptr;
  b.triPacked = ctx.triPacked.ptr;

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

  if (calibrate) {
    if (sequential)
      (void)calibrateSeqSmall(ctx, t, b, pixTotal, ctx.stage.capacity, stream); // (*calibrate stays AUTO: not a PERPIXEL choice)
    else
      *calibrate = calibratePixKernel(ctx, t, b, rows, ctx.stage.capacity, stream);
    return;
  }
  if (!sequential) t.pixKernel = resolvePixKernel(ctx, p, t, rows);
  // a measured small-scene choice for exactly this scene, camera, frame shape and pass count
  if (sequential && ctx.seqSmallChoice != 0 && seqSmallKernelIsOpen(t, hints) && ctx.seqSmallChoiceKey == seqSmallChoiceKeyOf(ctx, t))
    hints.seqSmallKernel = ctx.seqSmallChoice;

  uint64_t done = 0;
  for (uint32_t begin = 0; begin < pixTotal;) {
    t.pixBegin = begin;
    t.pixCount = static_cast<uint32_t>(std::min<uint64_t>(bandPix, pixTotal - begin));
    begin += t.pixCount;
    t.firstBand = t.pixBegin == 0;
    const char *variant = "";
    if (sequential)
      timedLaunch(true, [&] { return launchTraceSequential(t, b, hints, stream, &variant); });
    else
      timedLaunch(true, [&] { return launchTracePerPixel(t, b, hints, stream, &variant); });
    std::snprintf(ctx.traceKernel, sizeof ctx.traceKernel, "%s", variant ? variant : "");
    timedLaunch(false, [&] { return launchResolve(t, ctx.stage.ptr, dRgb, dCounts, stream); });
    done += static_cast<uint64_t>(t.pixCount) * npass;
    ctx.statSamples += static_cast<uint64_t>(t.pixCount) * npass;
    if (betweenBands(t, done, static_cast<uint64_t>(pixTotal) * npass)) break;
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
  const Bvh bvh = buildBvh(data.triGeom.data(), scene->num_triangles);
  ctx->bvhNodes.upload(bvh.nodes.data(), bvh.nodes.size(), nullptr);
  ctx->bvhLeafGeom.upload(bvh.leafGeom.data(), bvh.leafGeom.size(), nullptr);
  ctx->bvhLeafIndex.upload(bvh.leafIndex.data(), bvh.leafIndex.size(), nullptr);
  const PrefilterData pre = buildPrefilter(data.triGeom.data(), scene->num_triangles, scene->sph_centre_radius, scene->num_spheres);
  ctx->triPacked.upload(pre.pairs.data(), pre.pairs.size(), nullptr);
  ctx->prefilterUsable = pre.usable;
  ctx->unitsCoherent = unitUSkipFraction(data.triGeom.data(), scene->num_triangles) >= kUnitUFirstThreshold;
  check(hipStreamSynchronize(nullptr), "scene upload");
  ctx->ntri = scene->num_triangles;
  ctx->nsph = scene->num_spheres;
  std::memcpy(ctx->env, data.environment, sizeof ctx->env);
  ctx->triMaterial = std::move(data.triMaterial);
  ctx->sphMaterial = std::move(data.sphMaterial);
  ctx->haveScene = true;
  ctx->sceneGeneration++;
  ctx->pixChoice = 0;
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_render(ptw_context *ctx, const ptw_camera *camera, const ptw_render_params *params,
                       void *d_rgb_sum, void *d_counts, void *d_words, void *hip_stream) {
  if (!ctx || !camera || !params || !d_rgb_sum || !d_counts) return invalid("null pointer");
  PTW_GUARD_BEGIN
  enqueueRender(*ctx, *camera, *params, static_cast<double *>(d_rgb_sum),
                static_cast<uint32_t *>(d_counts), static_cast<uint32_t *>(d_words),
                static_cast<hipStream_t>(hip_stream), 0,
                [](const TraceParams &, uint64_t, uint64_t) { return false; });
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_calibrate(ptw_context *ctx, const ptw_camera *camera, const ptw_render_params *params,
                          void *hip_stream, int32_t *kernel_out) {
  if (!ctx || !camera || !params) return invalid("null pointer");
  PTW_GUARD_BEGIN
  int choice = kPixKernelAuto;
  enqueueRender(*ctx, *camera, *params, nullptr, nullptr, nullptr, static_cast<hipStream_t>(hip_stream), 0,
                [](const TraceParams &, uint64_t, uint64_t) { return false; }, &choice);
  if (kernel_out) *kernel_out = choice;
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_context_set_debug(ptw_context *ctx, const ptw_debug_options *options) {
  if (!ctx) return invalid("ctx");
  if (options)
    ctx->debug = *options;
  else
    ptw_debug_defaults(&ctx->debug);
  return PTW_OK;
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
  std::snprintf(out->trace_kernel, sizeof out->trace_kernel, "%s", ctx->traceKernel);
  if (reset) {
    ctx->clearEvents();
    ctx->statSamples = 0;
    ctx->rayCarry = 0;
  }
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_dispatch_plan(const ptw_dispatch_query *q, const ptw_debug_options *debug, char *out, size_t capacity) {
  if (!q || !out || capacity == 0) return invalid("null pointer");
  PTW_GUARD_BEGIN
  if (q->rng_policy != PTW_RNG_SEQUENTIAL && q->rng_policy != PTW_RNG_PERPIXEL) throw std::invalid_argument("unknown rng_policy");
  if (q->accel == PTW_ACCEL_BVH && q->rng_policy != PTW_RNG_PERPIXEL)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "PTW_ACCEL_BVH needs PTW_RNG_PERPIXEL");
  if (q->accel == PTW_ACCEL_PREFILTER && q->rng_policy == PTW_RNG_SEQUENTIAL && q->num_triangles <= 128)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "PTW_ACCEL_PREFILTER under PTW_RNG_SEQUENTIAL: scenes beyond 128 triangles");
  if (q->samples_per_pixel <= 0) throw std::invalid_argument("samples_per_pixel must be positive");
  ptw_context probe; // (never touches a device: only its debug -> hints translation is used)
  if (debug) probe.debug = *debug;
  LaunchHints hints = probe.hints();
  hints.dryRun = true;
  hints.cus = q->compute_units;
  TraceParams t;
  std::memset(&t, 0, sizeof t);
  t.ntri = q->num_triangles, t.nsph = q->num_spheres, t.nmat = q->num_materials;
  t.maxDepth = q->max_depth;
  t.fbU = t.fbV = 4;
  t.width = t.height = 1024;
  t.npix = 1024u * 1024u;
  t.pixCount = t.npix;
  t.npass = static_cast<uint32_t>(q->samples_per_pixel);
  t.rngPolicy = q->rng_policy;
  t.accel = q->accel;
  t.seqUnitUFirst = debug && debug->seq_unit_ufirst == 1; // (the rule itself needs the scene: ptw_scene_unit_coherence)
  t.pixKernel = q->pix_kernel == PTW_PIX_KERNEL_LOCKSTEP ? kPixKernelLockstep : kPixKernelPersistent;
  TraceBuffers b;
  std::memset(&b, 0, sizeof b);
  static double notNull;
  b.specState = &notNull; // (the dispatcher only asks whether the ring's parking space exists)
  const char *variant = "";
  const hipError_t e = q->rng_policy == PTW_RNG_SEQUENTIAL ? launchTraceSequential(t, b, hints, nullptr, &variant)
                                                           : launchTracePerPixel(t, b, hints, nullptr, &variant);
  if (e != hipSuccess) throw DeviceError(PTW_ERR_HIP, "dispatch plan");
  const size_t n = std::strlen(variant);
  if (n + 1 > capacity) throw std::invalid_argument("ptw_dispatch_plan: buffer too small");
  std::memcpy(out, variant, n + 1);
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
  // (tests: the same search through the fp32 prefilter of PTW_ACCEL_PREFILTER)
  if (ctx->debug.intersect_accel == PTW_ACCEL_PREFILTER) {
    bool ok = ctx->prefilterUsable;
    for (uint64_t i = 0; i < n && ok; ++i) ok = prefilterAcceptsOrigin(rays + 6 * i, 0.0);
    if (!ok) throw DeviceError(PTW_ERR_UNSUPPORTED, "PTW_ACCEL_PREFILTER: coordinates (scene or ray origins) beyond 1e12");
    t.accel = PTW_ACCEL_PREFILTER;
    b.triPacked = ctx->triPacked.ptr;
  } else if (ctx->debug.intersect_accel != PTW_ACCEL_NONE) {
    throw std::invalid_argument("ptw_debug_options.intersect_accel: PTW_ACCEL_NONE or PTW_ACCEL_PREFILTER");
  }
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
  ptw_render_options opt;
  std::memset(&opt, 0, sizeof opt);
  opt.progress = progress;
  opt.progress_user = user;
  return ptw_render_ex(scene, camera, params, rgb_sum, counts, &opt);
}

} // extern "C"

namespace {

// ptw_render / ptw_render_ex are synchronous calls: a PERPIXEL render of 16 M samples or more whose
// caller left the kernel choice open first times the trial (ptw_context_calibrate).
bool wantsCalibration(const ptw_render_params &p) {
  if (p.rng_policy != PTW_RNG_PERPIXEL || p.pix_kernel != PTW_PIX_KERNEL_AUTO || p.accel != PTW_ACCEL_NONE)
    return false;
  const uint64_t samples = static_cast<uint64_t>(rowsOf(p).count) * static_cast<uint64_t>(p.width) *
                           static_cast<uint64_t>(std::max(0, p.samples_per_pixel));
  return samples >= kCalibrateFromSamples;
}
// ... and a SEQUENTIAL render of that size whose pass count may leave the small-scene kernel open (whether it
// does depends on the scene: ptw_context_calibrate looks, and does nothing when it does not)
bool wantsSeqCalibration(const ptw_render_params &p) {
  if (p.rng_policy != PTW_RNG_SEQUENTIAL) return false;
  const uint64_t samples = static_cast<uint64_t>(rowsOf(p).count) * static_cast<uint64_t>(p.width) *
                           static_cast<uint64_t>(std::max(0, p.samples_per_pixel));
  return samples >= kCalibrateFromSamples;
}

// One device's share of a ptw_render_ex call: its own context (one scene upload), its own
// stream, device-resident framebuffer.
struct DeviceShard {
  int device = 0;
  ptw_render_params params;
  std::unique_ptr<ptw_context, void (*)(ptw_context *)> ctx{nullptr, ptw_context_destroy};
  DeviceArray<double> rgb;
  DeviceArray<uint32_t> counts;
  hipStream_t stream = nullptr;
  int status = PTW_OK;
  std::string error;
  ~DeviceShard() {
    if (stream) {
      (void)hipSetDevice(device);
      (void)hipStreamDestroy(stream);
    }
  }
};

// Runs `body` and records a failure in the shard instead of letting it escape the thread.
template <typename F>
void guarded(DeviceShard &sh, F &&body) {
  try {
    body();
  } catch (...) {
    sh.status = translateException();
    sh.error = ptw_last_error(); // thread-local: copy it out on the thread that failed
  }
}

void renderSingle(const ptw_scene_view &scene, const ptw_camera &camera,
                  const ptw_render_params &params, double *rgbSum, uint32_t *counts,
                  const ptw_render_options &opt) {
  ptw_context *raw = nullptr;
  if (ptw_context_create(params.device, &raw) != PTW_OK) throw DeviceError(PTW_ERR_NO_DEVICE, ptw_last_error());
  std::unique_ptr<ptw_context, void (*)(ptw_context *)> ctx(raw, ptw_context_destroy);
  if (int rc = ptw_context_set_scene(ctx.get(), &scene); rc != PTW_OK) throw DeviceError(rc, ptw_last_error());
  if (opt.debug) ctx->debug = *opt.debug;
  const size_t npix = static_cast<size_t>(params.width) * params.height;
  DeviceArray<double> dRgb;
  DeviceArray<uint32_t> dCounts;
  dRgb.upload(rgbSum, npix * 3, nullptr);
  dCounts.upload(counts, npix, nullptr);
  bool cancelled = false;
  const bool talk = opt.progress || opt.update;
  // With a callback the frame is cut into bands so that there is something to report: 20 for
  // the 5 % steps of the reference's Progressifier, `min_updates` for snapshots.
  const int minBands = opt.update ? (opt.min_updates > 0 ? opt.min_updates : 16) : (opt.progress ? 20 : 0);
  ptw_render_params rp = params;
  if (wantsSeqCalibration(rp))
    if (int rc = ptw_context_calibrate(ctx.get(), &camera, &rp, nullptr, nullptr); rc != PTW_OK) throw DeviceError(rc, ptw_last_error());
  if (wantsCalibration(rp)) {
    int32_t choice = PTW_PIX_KERNEL_AUTO;
    if (int rc = ptw_context_calibrate(ctx.get(), &camera, &rp, nullptr, &choice); rc != PTW_OK)
      throw DeviceError(rc, ptw_last_error());
    rp.pix_kernel = choice;
  }
  enqueueRender(*ctx, camera, rp, dRgb.ptr, dCounts.ptr, nullptr, nullptr, minBands,
                [&](const TraceParams &t, uint64_t done, uint64_t total) {
                  if (!talk) return false;
                  check(hipStreamSynchronize(nullptr), "band");
                  if (opt.update) {
                    // bring the caller's buffers up to date: the image rows this band touched
                    const size_t w = static_cast<size_t>(params.width);
                    const size_t r0 = t.pixBegin / w, r1 = (t.pixBegin + t.pixCount - 1) / w;
                    auto copyRows = [&](size_t row, size_t rows) {
                      check(hipMemcpy(rgbSum + row * w * 3, dRgb.ptr + row * w * 3,
                                      rows * w * 3 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
                      check(hipMemcpy(counts + row * w, dCounts.ptr + row * w,
                                      rows * w * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H");
                    };
                    if (t.rowStride == 1) {
                      copyRows(static_cast<size_t>(t.rowFirst) + r0, r1 - r0 + 1);
                    } else { // interleaved rows are not contiguous in the frame
                      for (size_t r = r0; r <= r1; ++r)
                        copyRows(static_cast<size_t>(t.rowFirst) + r * static_cast<size_t>(t.rowStride), 1);
                    }
                    if (opt.update(opt.update_user, done, total, rgbSum, counts) != 0) cancelled = true;
                  }
                  if (opt.progress && opt.progress(opt.progress_user, done, total) != 0) cancelled = true;
                  return cancelled;
                });
  check(hipStreamSynchronize(nullptr), "render");
  check(hipMemcpy(rgbSum, dRgb.ptr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
  check(hipMemcpy(counts, dCounts.ptr, npix * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H");
  if (cancelled) throw std::invalid_argument("cancelled by the callback");
}

void renderMulti(const ptw_scene_view &scene, const ptw_camera &camera,
                 const ptw_render_params &params, double *rgbSum, uint32_t *counts,
                 const ptw_render_options &opt) {
  if (opt.update)
    throw DeviceError(PTW_ERR_UNSUPPORTED, "update callbacks need a single-device render");
  validate(params);
  const int n = opt.num_devices;
  const bool sequential = params.rng_policy == PTW_RNG_SEQUENTIAL;
  if (!sequential && (params.row_begin != 0 || params.row_end != 0 || params.row_stride > 1))
    throw DeviceError(PTW_ERR_UNSUPPORTED, "multi-device renders shard the rows themselves");
  const size_t npix = static_cast<size_t>(params.width) * params.height;
  const int total = params.samples_per_pixel;

  std::vector<DeviceShard> shards(static_cast<size_t>(n));
  std::vector<int32_t> devices(static_cast<size_t>(n));
  int deviceCount = 0;
  if (hipGetDeviceCount(&deviceCount) != hipSuccess || deviceCount <= 0)
    throw DeviceError(PTW_ERR_NO_DEVICE, "no HIP device available (the hip way has no CPU fallback)");
  for (int g = 0; g < n; ++g) {
    devices[g] = opt.share_device ? params.device : (opt.devices ? opt.devices[g] : params.device + g);
    if (devices[g] < 0 || devices[g] >= deviceCount)
      throw DeviceError(PTW_ERR_NO_DEVICE, "HIP device ordinal " + std::to_string(devices[g]) +
                                               " out of range (" + std::to_string(deviceCount) +
                                               " device(s) visible)");
    DeviceShard &sh = shards[g];
    sh.device = devices[g];
    sh.params = params;
    sh.params.device = devices[g];
    if (sequential) { // the reference's decomposition: ranges of passes (Scene.cpp:208-246)
      const int base = total / n, extra = total % n;
      sh.params.first_pass = params.first_pass + g * base + std::min(g, extra);
      sh.params.samples_per_pixel = base + (g < extra ? 1 : 0);
    } else {          // interleaved image rows
      sh.params.row_stride = n;
      sh.params.row_phase = g;
    }
  }

  if (opt.share_device == 1) {
    // Hook for 1-GPU boxes: the same shards, one after another, accumulated on the device.
    DeviceShard &sh = shards[0];
    ptw_context *raw = nullptr;
    if (ptw_context_create(sh.device, &raw) != PTW_OK) throw DeviceError(PTW_ERR_NO_DEVICE, ptw_last_error());
    sh.ctx.reset(raw);
    if (int rc = ptw_context_set_scene(sh.ctx.get(), &scene); rc != PTW_OK) throw DeviceError(rc, ptw_last_error());
    if (opt.debug) sh.ctx->debug = *opt.debug;
    sh.rgb.upload(rgbSum, npix * 3, nullptr);
    sh.counts.upload(counts, npix, nullptr);
    // one PERPIXEL kernel for every shard, measured once (as the threaded path does)
    if (wantsCalibration(shards[0].params)) {
      int32_t choice = PTW_PIX_KERNEL_AUTO;
      if (int rc = ptw_context_calibrate(sh.ctx.get(), &camera, &shards[0].params, nullptr, &choice); rc != PTW_OK)
        throw DeviceError(rc, ptw_last_error());
      for (DeviceShard &each : shards) each.params.pix_kernel = choice;
    }
    for (int g = 0; g < n; ++g) {
      enqueueRender(*sh.ctx, camera, shards[g].params, sh.rgb.ptr, sh.counts.ptr, nullptr, nullptr, 0,
                    [](const TraceParams &, uint64_t, uint64_t) { return false; });
      check(hipStreamSynchronize(nullptr), "render");
      if (opt.progress && opt.progress(opt.progress_user, static_cast<uint64_t>(g + 1), static_cast<uint64_t>(n)) != 0)
        throw std::invalid_argument("cancelled by the callback");
    }
    check(hipMemcpy(rgbSum, sh.rgb.ptr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
    check(hipMemcpy(counts, sh.counts.ptr, npix * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H");
    return;
  }

  // One communicator per shard: RCCL over the devices, or - share_device == 2, every shard on
  // params->device - the in-process loopback transport (RCCL refuses two ranks on one GPU): the same
  // host threads, contexts, streams and collective entry points as the N-GPU render.
  std::vector<ptw_comm *> comms(static_cast<size_t>(n), nullptr);
  const int commRc = opt.share_device ? ptw_comm_create_loopback(n, params.device, comms.data())
                                      : ptw_comm_create_all(n, devices.data(), comms.data());
  if (commRc != PTW_OK) throw DeviceError(PTW_ERR_HIP, ptw_last_error());
  struct CommGuard {
    std::vector<ptw_comm *> &c;
    ~CommGuard() {
      for (auto *x : c) ptw_comm_destroy(x);
    }
  } commGuard{comms};
  auto abortAll = [&] {
    for (auto *c : comms) (void)ptw_comm_abort(c);
  };
  // Failure injection for the tests (a shard that fails must end the render with an error on every
  // path, never leave its peers waiting in a collective), ptw_debug_options: fail_shard = g fails shard
  // g's set-up, fail_collective = g its collective call; silent_shard = g reports success WITHOUT entering
  // the collective - the in-process picture of a peer that dies after the others have enqueued theirs;
  // the watchdog must end it.
  const int failSetup = opt.debug ? opt.debug->fail_shard : -1;
  const int failCollective = opt.debug ? opt.debug->fail_collective : -1;
  const int silentShard = opt.debug ? opt.debug->silent_shard : -1;

  // The PERPIXEL kernel choice is made ONCE, on the first shard's context, and handed to every
  // shard: all of them run the same kernel (two shards that each timed their own trial could pick
  // differently - same bytes, different speed, and the gather waits for the slower one).
  auto makeContext = [&](DeviceShard &sh) {
    ptw_context *raw = nullptr;
    if (ptw_context_create(sh.device, &raw) != PTW_OK) throw DeviceError(PTW_ERR_NO_DEVICE, ptw_last_error());
    sh.ctx.reset(raw);
    if (opt.share_device) raw->deviceShare = n;
    if (int rc = ptw_context_set_scene(sh.ctx.get(), &scene); rc != PTW_OK) throw DeviceError(rc, ptw_last_error());
    if (opt.debug) raw->debug = *opt.debug;
  };
  if (wantsCalibration(shards[0].params) && failSetup != 0) {
    makeContext(shards[0]);
    int32_t choice = PTW_PIX_KERNEL_AUTO;
    if (int rc = ptw_context_calibrate(shards[0].ctx.get(), &camera, &shards[0].params, nullptr, &choice); rc != PTW_OK)
      throw DeviceError(rc, ptw_last_error());
    for (DeviceShard &sh : shards) sh.params.pix_kernel = choice;
  }

  // ---- phase 1, one host thread per device: context + scene upload, the render, and its
  // completion.  No collective yet: a shard that fails here has no peer waiting for it. ----
  std::atomic<bool> cancelled{false};
  {
    std::vector<std::thread> threads;
    for (int g = 0; g < n; ++g)
      threads.emplace_back([&, g] {
        DeviceShard &sh = shards[g];
        guarded(sh, [&] {
          if (g == failSetup) throw DeviceError(PTW_ERR_HIP, "injected failure (ptw_debug_options.fail_shard)");
          if (!sh.ctx) makeContext(sh);
          sh.ctx->activate();
          check(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking), "hipStreamCreate");
          sh.rgb.reserve(npix * 3);
          sh.counts.reserve(npix);
          // The caller's running sums (ArrayOutput +=): under the pass sharding they live on the first
          // device and the reduce adds the others' passes; under the row sharding every device adds its
          // rows to its own copy and the gather brings those rows - old content included - to the root.
          if (g == 0 || !sequential) {
            sh.rgb.upload(rgbSum, npix * 3, sh.stream);
            sh.counts.upload(counts, npix, sh.stream);
          } else {
            check(hipMemsetAsync(sh.rgb.ptr, 0, npix * 3 * sizeof(double), sh.stream), "memset");
            check(hipMemsetAsync(sh.counts.ptr, 0, npix * sizeof(uint32_t), sh.stream), "memset");
          }
          // (SEQUENTIAL, a small scene, this shard's pass count between one and six per CU - cfg5's per-GPU share:
          // every shard times the small-scene kernels on its own device; same bytes whichever wins)
          if (wantsSeqCalibration(sh.params))
            if (int rc = ptw_context_calibrate(sh.ctx.get(), &camera, &sh.params, sh.stream, nullptr); rc != PTW_OK)
              throw DeviceError(rc, ptw_last_error());
          enqueueRender(*sh.ctx, camera, sh.params, sh.rgb.ptr, sh.counts.ptr, nullptr, sh.stream,
                        g == 0 && opt.progress ? 20 : 0,
                        [&](const TraceParams &, uint64_t done, uint64_t totalSamples) {
                          if (g == 0 && opt.progress) {
                            check(hipStreamSynchronize(sh.stream), "band");
                            if (opt.progress(opt.progress_user, done * n, totalSamples * n) != 0) cancelled = true;
                          }
                          return cancelled.load(); // every shard stops at its next band
                        });
          check(hipStreamSynchronize(sh.stream), "render");
        });
      });
    for (auto &t : threads) t.join();
  }
  auto firstFailure = [&]() -> const DeviceShard * {
    for (const DeviceShard &sh : shards)
      if (sh.status != PTW_OK) return &sh;
    return nullptr;
  };
  if (const DeviceShard *bad = firstFailure()) {
    abortAll();
    throw DeviceError(bad->status, "device " + std::to_string(bad->device) + ": " + bad->error);
  }
  if (cancelled) {
    abortAll();
    throw std::invalid_argument("cancelled by the callback");
  }

  // ---- phase 2: the one collective, entered by every shard (all of them are healthy), and its
  // completion under the watchdog (ptw_comm_wait).  A shard whose call fails, or whose wait ends with
  // an asynchronous RCCL error or the timeout - a peer that never arrived -, aborts its communicator,
  // which releases the peers (loopback: the host rendezvous; RCCL: ncclCommAbort ends the kernels
  // that wait for the missing peer on the device) - and every other communicator with it, from the
  // failing shard's own thread (ptw_comm_abort may be called while another thread waits on the
  // communicator: that is what it is for). ----
  {
    std::vector<std::thread> threads;
    for (int g = 0; g < n; ++g)
      threads.emplace_back([&, g] {
        DeviceShard &sh = shards[g];
        int rc;
        if (g == failCollective) {
          rc = PTW_ERR_HIP;
          sh.error = "injected failure (ptw_debug_options.fail_collective)";
        } else if (g == silentShard) {
          return; // (test hook: this shard never shows up)
        } else {
          if (sequential)
            rc = ptw_comm_reduce_framebuffer(comms[g], sh.rgb.ptr, sh.counts.ptr, npix, 0, sh.stream);
          else
            rc = ptw_comm_gather_rows(comms[g], sh.rgb.ptr, sh.counts.ptr, params.width, params.height, 0,
                                      sh.stream);
          if (rc == PTW_OK) rc = ptw_comm_wait(comms[g], sh.stream, 0);
          if (rc != PTW_OK) sh.error = ptw_last_error();
        }
        if (rc != PTW_OK) {
          // this shard's collective failed: give up on EVERY communicator at once, so that the healthy
          // shards' waits end now ("communicator aborted") and not at the watchdog's deadline - on the RCCL
          // path aborting one rank does not release its peers
          sh.status = rc;
          abortAll();
        }
      });
    for (auto &t : threads) t.join();
  }
  if (const DeviceShard *bad = firstFailure()) {
    // report the shard that failed on its own, not a peer that was released by the abort
    for (const DeviceShard &sh : shards)
      if (sh.status != PTW_OK && sh.error.find("communicator aborted") != 0) {
        bad = &sh;
        break;
      }
    const int status = bad->status;
    const std::string message = "device " + std::to_string(bad->device) + ": " + bad->error;
    abortAll();
    throw DeviceError(status, message);
  }
  check(hipSetDevice(shards[0].device), "hipSetDevice");
  check(hipMemcpy(rgbSum, shards[0].rgb.ptr, npix * 3 * sizeof(double), hipMemcpyDeviceToHost), "D2H");
  check(hipMemcpy(counts, shards[0].counts.ptr, npix * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H");
}

} // namespace

extern "C" int ptw_render_ex(const ptw_scene_view *scene, const ptw_camera *camera,
                             const ptw_render_params *params, double *rgb_sum, uint32_t *counts,
                             const ptw_render_options *options) {
  if (!scene || !camera || !params || !rgb_sum || !counts) return invalid("null pointer");
  ptw_render_options opt;
  std::memset(&opt, 0, sizeof opt);
  if (options) opt = *options;
  if (opt.num_devices < 0) return invalid("num_devices");
  PTW_GUARD_BEGIN
  validate(*params);
  if (opt.num_devices > 1)
    renderMulti(*scene, *camera, *params, rgb_sum, counts, opt);
  else
    renderSingle(*scene, *camera, *params, rgb_sum, counts, opt);
  return PTW_OK;
  PTW_GUARD_END
}

// main.cpp — `pt_three_ways_hip`: the reference's command line (src/main/main.cpp:370-474) with
// the hip way plugged in where `oo` / `fp` / `dod` are dispatched (main.cpp:350-366).
//
// Same flags, defaults and output as upstream:
//   -w/--width -h/--height --max-cpus --spp --first-bounce-u --first-bounce-v --max-depth
//   --seed --preview --save-every --way --scene --raw <output>
// plus what a GPU way needs: --device N, --gpus N, --rng sequential|perpixel, --scenes-dir DIR.
// Everything goes through the C ABI of include/ptw.h - the same boundary a cgo/JNI/ctypes host
// would bind - so this file is also the worked example for INTEGRATION.md.
#include "../../include/ptw.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <string>
#include <vector>

namespace {

struct Options {
  ptw_render_params params;
  int maxCpus = 1;
  int saveEvery = 30;
  int gpus = 1;
  bool raw = false;
  bool help = false;
  std::string way = "hip";
  std::string scene = "cornell";
  std::string scenesDir = "scenes";
  std::string output;
  // --debug name=value[,name=value...]: the library's dispatch forced from outside (ptw_debug_options:
  // tests and A/B runs only - the byte-equality tests run every kernel variant through this binary)
  bool haveDebug = false;
  ptw_debug_options debug;
  int shareDevice = 0; // --debug share_device=1|2: every shard of --gpus N on the one device
};

[[noreturn]] void usageError(const std::string &message) {
  std::cerr << "Error in command line: " << message << '\n';
  std::exit(1);
}

void printHelp() {
  std::cout << "usage: pt_three_ways_hip [options] <output>\n"
               "  -w, --width <width>        output image width (1920)\n"
               "  -h, --height <height>      output image height (1080)\n"
               "  --max-cpus <#cpus>         accepted for compatibility (the hip way runs on the GPU)\n"
               "  --spp <samples>            number of samples per pixel (40)\n"
               "  --first-bounce-u <samples> number of first bounce u samples (4)\n"
               "  --first-bounce-v <samples> number of first bounce v samples (4)\n"
               "  --max-depth <depth>        maximum recursion depth (5)\n"
               "  --seed <seed>              set rendering seed (0 to use random seed)\n"
               "  --preview                  super quick preview\n"
               "  --save-every <secs>        periodically save (every secs), 0 to disable (30)\n"
               "  --way <way>                which way: hip (oo, fp, dod are the reference's CPU ways)\n"
               "  --scene <scene>            cornell suzanne ce single-sphere multi-sphere example1 bbc-owl\n"
               "  --raw                      output in raw form\n"
               "  --device <n>               HIP device ordinal (0)\n"
               "  --gpus <n>                 shard the passes over devices <device> .. <device>+n-1 (1)\n"
               "  --rng <policy>             sequential (reference-exact, default) | perpixel\n"
               "  --accel <mode>             none (the reference's brute force, default) | bvh (perpixel only:\n"
               "                             same image, triangles culled by a bounding-volume hierarchy) |\n"
               "                             prefilter (perpixel only: same image, every triangle looked at in\n"
               "                             fp32 first and in fp64 only where fp32 cannot prove a miss)\n"
               "  --pix-kernel <kernel>      perpixel policy: auto (default) | lockstep | persistent\n"
               "  --scenes-dir <dir>         where the .obj/.mtl files live (scenes)\n"
               "  --debug <name=value,...>   tests / A-B runs only: ptw_debug_options fields (include/ptw.h),\n"
               "                             share_device=1|2 (--gpus N on one device)\n"
               "  -?, --help\n";
}

int toInt(const std::string &flag, const char *text);

void parseDebug(Options &o, const std::string &spec) {
  if (!o.haveDebug) {
    ptw_debug_defaults(&o.debug);
    o.haveDebug = true;
  }
  size_t at = 0;
  while (at < spec.size()) {
    size_t end = spec.find(',', at);
    if (end == std::string::npos) end = spec.size();
    const std::string item = spec.substr(at, end - at);
    at = end + 1;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) usageError("--debug wants name=value, got '" + item + "'");
    const std::string name = item.substr(0, eq), text = item.substr(eq + 1);
    if (name == "seq_units") { // seq_units=o:y:m
      if (std::sscanf(text.c_str(), "%d:%d:%d", &o.debug.seq_units[0], &o.debug.seq_units[1], &o.debug.seq_units[2]) != 3)
        usageError("--debug seq_units wants older:younger:master");
      continue;
    }
    const int v = toInt("--debug " + name, text.c_str());
    if (name == "seq_two_masters") o.debug.seq_two_masters = v;
    else if (name == "seq_pairing") o.debug.seq_pairing = v;
    else if (name == "seq_lds_tables") o.debug.seq_lds_tables = v;
    else if (name == "seq_small_kernel") o.debug.seq_small_kernel = v;
    else if (name == "pix_samples_per_lane") o.debug.pix_samples_per_lane = v;
    else if (name == "pix_waves_per_simd") o.debug.pix_waves_per_simd = v;
    else if (name == "gang_groups") o.debug.gang_groups = v;
    else if (name == "trace") o.debug.trace = v;
    else if (name == "seq_unit_ufirst") o.debug.seq_unit_ufirst = v;
    else if (name == "share_device") o.shareDevice = v;
    else usageError("Unknown --debug option " + name);
  }
}

int toInt(const std::string &flag, const char *text) {
  char *end = nullptr;
  const long v = std::strtol(text, &end, 10);
  if (end == text || *end != '\0') usageError("Unable to convert '" + std::string(text) + "' for " + flag);
  return static_cast<int>(v);
}

Options parse(int argc, const char *argv[]) {
  Options o;
  ptw_default_params(&o.params);
  auto value = [&](int &i, const std::string &flag) -> const char * {
    if (i + 1 >= argc) usageError("Expected argument following " + flag);
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-w" || a == "--width") o.params.width = toInt(a, value(i, a));
    else if (a == "-h" || a == "--height") o.params.height = toInt(a, value(i, a));
    else if (a == "--max-cpus") o.maxCpus = toInt(a, value(i, a));
    else if (a == "--spp") o.params.samples_per_pixel = toInt(a, value(i, a));
    else if (a == "--first-bounce-u") o.params.first_bounce_u = toInt(a, value(i, a));
    else if (a == "--first-bounce-v") o.params.first_bounce_v = toInt(a, value(i, a));
    else if (a == "--max-depth") o.params.max_depth = toInt(a, value(i, a));
    else if (a == "--seed") o.params.seed = toInt(a, value(i, a));
    else if (a == "--preview") o.params.preview = 1;
    else if (a == "--save-every") o.saveEvery = toInt(a, value(i, a));
    else if (a == "--way") o.way = value(i, a);
    else if (a == "--scene") o.scene = value(i, a);
    else if (a == "--raw") o.raw = true;
    else if (a == "--device") o.params.device = toInt(a, value(i, a));
    else if (a == "--gpus") o.gpus = toInt(a, value(i, a));
    else if (a == "--scenes-dir") o.scenesDir = value(i, a);
    else if (a == "--rng") {
      const std::string p = value(i, a);
      if (p == "sequential") o.params.rng_policy = PTW_RNG_SEQUENTIAL;
      else if (p == "perpixel") o.params.rng_policy = PTW_RNG_PERPIXEL;
      else usageError("Unknown rng policy " + p);
    } else if (a == "--accel") {
      const std::string m = value(i, a);
      if (m == "none") o.params.accel = PTW_ACCEL_NONE;
      else if (m == "bvh") o.params.accel = PTW_ACCEL_BVH;
      else if (m == "prefilter") o.params.accel = PTW_ACCEL_PREFILTER;
      else usageError("Unknown accel mode " + m);
    } else if (a == "--pix-kernel") {
      const std::string m = value(i, a);
      if (m == "auto") o.params.pix_kernel = PTW_PIX_KERNEL_AUTO;
      else if (m == "lockstep") o.params.pix_kernel = PTW_PIX_KERNEL_LOCKSTEP;
      else if (m == "persistent") o.params.pix_kernel = PTW_PIX_KERNEL_PERSISTENT;
      else usageError("Unknown pix kernel " + m);
    } else if (a == "--debug") parseDebug(o, value(i, a));
    else if (a == "-?" || a == "--help") o.help = true;
    else if (!a.empty() && a[0] == '-' && a.size() > 1) usageError("Unrecognised token: " + a);
    else o.output = a;
  }
  return o;
}

struct Output {
  const Options *options;
  std::vector<double> rgbSum;
  std::vector<uint32_t> counts;
};

bool save(const Output &out) {
  const Options &o = *out.options;
  const int w = o.params.width, h = o.params.height;
  if (o.raw) {
    if (ptw_raw_save(o.output.c_str(), w, h, out.rgbSum.data(), out.counts.data()) == PTW_OK) return true;
  } else {
    std::vector<uint8_t> rgb8(static_cast<size_t>(w) * h * 3);
    if (ptw_pixels_rgb8(w, h, out.rgbSum.data(), out.counts.data(), rgb8.data()) == PTW_OK &&
        ptw_png_save(o.output.c_str(), w, h, rgb8.data()) == PTW_OK)
      return true;
  }
  std::cerr << "Unable to save " << (o.raw ? "raw" : "PNG") << ": " << ptw_last_error() << "\n";
  return false;
}

// Progressifier (src/util/Progressifier.cpp:11-21): a line every >= 5 % of progress.
struct Progress {
  double last = 0.0;
  std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
};
int onProgress(void *user, uint64_t done, uint64_t total) {
  auto *p = static_cast<Progress *>(user);
  const double pct = total ? 100.0 * static_cast<double>(done) / static_cast<double>(total) : 100.0;
  if (pct >= p->last + 5.0) {
    const double secs =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - p->start).count();
    std::printf("%.1fs : %.2f%% (%llu / %llu)\n", secs, pct, static_cast<unsigned long long>(done),
                static_cast<unsigned long long>(total));
    std::fflush(stdout);
    p->last = pct;
  }
  return 0;
}

} // namespace

int main(int argc, const char *argv[]) {
  Options o = parse(argc, argv);
  if (o.help) {
    printHelp();
    return 0;
  }
  if (o.output.empty()) {
    std::cerr << "Missing output filename.\n";
    printHelp();
    return 1;
  }
  if (o.way != "hip") {
    // main.cpp:364-366 throws "Unknown way"; oo/fp/dod are the reference's CPU renderers and
    // are not part of this build.
    std::cerr << "Unknown way " << o.way << " (this build provides the 'hip' way)\n";
    return 1;
  }
  if (o.params.seed == 0) { // main.cpp:426-429
    std::random_device device;
    o.params.seed = static_cast<int32_t>(device());
  }

  ptw_scene *scene = nullptr;
  ptw_camera camera;
  if (ptw_scene_create(&scene) != PTW_OK ||
      ptw_scene_build_named(scene, o.scene.c_str(), o.scenesDir.c_str(), o.params.width,
                            o.params.height, &camera) != PTW_OK) {
    std::cerr << ptw_last_error() << "\n";
    return 1;
  }
  ptw_scene_view view;
  ptw_scene_view_of(scene, &view);
  std::cout << "Scene contains " << view.num_triangles << " triangles and " << view.num_spheres
            << " spheres.\n"; // StatsSceneBuilder::report, main.cpp:320-323

  Output out;
  out.options = &o;
  out.rgbSum.assign(static_cast<size_t>(o.params.width) * o.params.height * 3, 0.0);
  out.counts.assign(static_cast<size_t>(o.params.width) * o.params.height, 0u);

  const auto startTime = std::chrono::system_clock::now();
  Progress progress;
  // --save-every (main.cpp:331-343): the reference's updateFunc re-saves the running output when
  // the interval has elapsed.  Here the library hands the running framebuffer back between its
  // launches (ptw_update_fn) inside ONE render: one context, one scene upload, every pass of a
  // band in one launch.
  struct Saver {
    Output *out;
    std::chrono::system_clock::time_point next;
    int every;
  } saver{&out, startTime + std::chrono::seconds(o.saveEvery), o.saveEvery};
  auto onUpdate = [](void *user, uint64_t done, uint64_t total, const double *, const uint32_t *) -> int {
    auto *sv = static_cast<Saver *>(user);
    const auto now = std::chrono::system_clock::now();
    if (now > sv->next && done < total) { // the buffers handed in are the caller's own (out)
      save(*sv->out);
      sv->next = now + std::chrono::seconds(sv->every);
    }
    return 0;
  };
  ptw_render_options ro;
  std::memset(&ro, 0, sizeof ro);
  ro.progress = onProgress;
  ro.progress_user = &progress;
  if (o.gpus > 1) {
    // The passes (SEQUENTIAL) or the interleaved image rows (PERPIXEL) are spread over the
    // devices and merged with one RCCL collective on the devices (ptw_render_ex).
    ro.num_devices = o.gpus;
    ro.share_device = o.shareDevice; // (--debug share_device=1: tests on a 1-GPU box)
  } else if (o.saveEvery > 0 && o.params.samples_per_pixel > 1) {
    ro.update = onUpdate;
    ro.update_user = &saver;
  }
  if (o.haveDebug) ro.debug = &o.debug;
  const int rc = ptw_render_ex(&view, &camera, &o.params, out.rgbSum.data(), out.counts.data(), &ro);
  const auto endTime = std::chrono::system_clock::now();
  ptw_scene_destroy(scene);
  if (rc != PTW_OK) {
    std::cerr << "render failed: " << ptw_last_error() << "\n";
    return 1;
  }
  if (!save(out)) return 1;

  // main.cpp:462-473
  const auto taken = endTime - startTime;
  const uint64_t totalSamples = ptw_total_samples(o.params.width, o.params.height, out.counts.data());
  std::cout << "Took " << std::chrono::duration_cast<std::chrono::seconds>(taken).count() << "s\n";
  std::cout << "Total samples: " << totalSamples << "\n";
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(taken).count();
  std::cout << "Samples/ms: " << static_cast<double>(totalSamples) / static_cast<double>(ms) << "\n";
  return 0;
}

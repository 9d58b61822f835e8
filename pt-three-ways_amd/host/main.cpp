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
#include <thread>
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
               "  --scenes-dir <dir>         where the .obj/.mtl files live (scenes)\n"
               "  -?, --help\n";
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
    } else if (a == "-?" || a == "--help") o.help = true;
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
  int rc = PTW_OK;
  Progress progress;
  if (o.gpus > 1) {
    // One host thread per device, each with its own context (ptw_render creates one): device d
    // renders a contiguous range of the passes - the reference's own decomposition, one task per
    // pass merged with ArrayOutput::operator+= (src/dod/Scene.cpp:208-246) - into its own
    // buffers; the partial frames are added in device order.
    struct Shard {
      std::vector<double> rgbSum;
      std::vector<uint32_t> counts;
      int rc = PTW_OK;
      std::string error;
    };
    const int total = o.params.samples_per_pixel;
    const bool shareDevice = std::getenv("PTW_CLI_SHARE_DEVICE") != nullptr; // tests on a 1-GPU box
    std::vector<Shard> shards(static_cast<size_t>(o.gpus));
    std::vector<std::thread> threads;
    for (int g = 0; g < o.gpus; ++g) {
      const int base = total / o.gpus, extra = total % o.gpus;
      const int first = g * base + std::min(g, extra), count = base + (g < extra ? 1 : 0);
      threads.emplace_back([&, g, first, count] {
        Shard &sh = shards[static_cast<size_t>(g)];
        sh.rgbSum.assign(out.rgbSum.size(), 0.0);
        sh.counts.assign(out.counts.size(), 0u);
        if (count == 0) return;
        ptw_render_params part = o.params;
        part.device = o.params.device + (shareDevice ? 0 : g);
        part.first_pass = o.params.first_pass + first;
        part.samples_per_pixel = count;
        sh.rc = ptw_render(&view, &camera, &part, sh.rgbSum.data(), sh.counts.data(), nullptr, nullptr);
        if (sh.rc != PTW_OK) sh.error = ptw_last_error(); // thread-local: copy it out here
      });
    }
    for (auto &t : threads) t.join();
    std::string firstError;
    for (const Shard &sh : shards) {
      if (sh.rc != PTW_OK && rc == PTW_OK) rc = sh.rc, firstError = sh.error;
      for (size_t i = 0; i < out.rgbSum.size(); ++i) out.rgbSum[i] += sh.rgbSum[i];
      for (size_t i = 0; i < out.counts.size(); ++i) out.counts[i] += sh.counts[i];
    }
    if (rc != PTW_OK) {
      std::cerr << "render failed: " << firstError << "\n";
      ptw_scene_destroy(scene);
      return 1;
    }
  } else if (o.saveEvery > 0 && o.params.samples_per_pixel > 1) {
    // --save-every (main.cpp:331-343): render in pass chunks, re-saving the running sum when
    // the interval has elapsed.  Chunks continue the same pass sequence through first_pass.
    auto nextSave = startTime + std::chrono::seconds(o.saveEvery);
    const int total = o.params.samples_per_pixel;
    const int chunk = std::max(1, total / 8);
    for (int first = 0; first < total && rc == PTW_OK; first += chunk) {
      ptw_render_params part = o.params;
      part.first_pass = o.params.first_pass + first;
      part.samples_per_pixel = std::min(chunk, total - first);
      rc = ptw_render(&view, &camera, &part, out.rgbSum.data(), out.counts.data(), nullptr, nullptr);
      if (rc == PTW_OK)
        onProgress(&progress, static_cast<uint64_t>(first + part.samples_per_pixel), total);
      const auto now = std::chrono::system_clock::now();
      if (rc == PTW_OK && now > nextSave && first + chunk < total) {
        save(out);
        nextSave = now + std::chrono::seconds(o.saveEvery);
      }
    }
  } else {
    rc = ptw_render(&view, &camera, &o.params, out.rgbSum.data(), out.counts.data(), onProgress,
                    &progress);
  }
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

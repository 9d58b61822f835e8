// raw_to_png.cpp — `raw_to_png_hip <output.png> <input.raw>...`: sums any number of `.raw`
// partial renders (same format as the reference's ArrayOutput::save) and writes one PNG, like
// src/main/raw_to_png.cpp:39-79.  It is the offline half of multi-process / multi-node
// rendering: run several renders with different --seed (or first_pass) and --raw, then merge.
#include "../../include/ptw.h"

#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, const char *argv[]) {
  if (argc < 3) {
    std::cerr << (argc < 2 ? "Missing output filename.\n" : "Missing inputs.\n")
              << "usage: raw_to_png_hip <output> <input>...\n";
    return 1;
  }
  const std::string output = argv[1];
  int32_t width = 0, height = 0;
  std::vector<double> rgbSum;
  std::vector<uint32_t> counts;
  uint64_t totalSamples = 0;
  for (int i = 2; i < argc; ++i) {
    std::cout << "Loading " << argv[i] << "...\n";
    int32_t w = 0, h = 0;
    if (ptw_raw_read_header(argv[i], &w, &h) != PTW_OK) {
      std::cerr << ptw_last_error() << "\n";
      return 1;
    }
    if (rgbSum.empty()) {
      width = w, height = h;
      std::cout << "  width: " << w << " height: " << h << '\n';
      rgbSum.assign(static_cast<size_t>(w) * h * 3, 0.0);
      counts.assign(static_cast<size_t>(w) * h, 0u);
    }
    if (w != width || h != height) {
      std::cerr << "Mismatch in size, width " << w << " height " << h << '\n';
      return 1;
    }
    const uint64_t before = ptw_total_samples(width, height, counts.data());
    if (ptw_raw_load_accumulate(argv[i], width, height, rgbSum.data(), counts.data()) != PTW_OK) {
      std::cerr << ptw_last_error() << "\n";
      return 1;
    }
    const uint64_t samples = ptw_total_samples(width, height, counts.data()) - before;
    totalSamples += samples;
    std::cout << "  samples: " << samples << '\n';
  }
  std::printf("Saving %s with %llu samples (%.1f per pixel)...\n", output.c_str(),
              static_cast<unsigned long long>(totalSamples),
              static_cast<double>(totalSamples) / (static_cast<double>(width) * height));
  std::vector<uint8_t> rgb8(static_cast<size_t>(width) * height * 3);
  if (ptw_pixels_rgb8(width, height, rgbSum.data(), counts.data(), rgb8.data()) != PTW_OK ||
      ptw_png_save(output.c_str(), width, height, rgb8.data()) != PTW_OK) {
    std::cerr << "Unable to save PNG: " << ptw_last_error() << "\n";
    return 1;
  }
  return 0;
}

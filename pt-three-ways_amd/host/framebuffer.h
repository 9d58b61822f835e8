// framebuffer.h — the ArrayOutput surface of the hip way (src/util/ArrayOutput.{h,cpp},
// src/util/SampledPixel.{h,cpp}, src/main/PngWriter.cpp).
//
// The device accumulates into two flat arrays — fp64 RGB running sums and uint32 sample
// counts, pixel index x + y*width with y = 0 at the top — which is what ArrayOutput's
// vector<SampledPixel> holds.  This file provides the byte-compatible `.raw` reader/writer,
// the gamma-2.2 8-bit conversion and a dependency-free PNG encoder.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptw {

// std::logic_error("Two differently-sized arrays ...") of ArrayOutput::operator+=.
struct SizeMismatch : std::logic_error {
  using std::logic_error::logic_error;
};

struct Framebuffer {
  int width{}, height{};
  std::vector<double> rgbSum;    // width*height*3
  std::vector<uint32_t> counts;  // width*height
  Framebuffer() = default;
  Framebuffer(int w, int h)
      : width(w), height(h), rgbSum(static_cast<size_t>(w) * h * 3, 0.0),
        counts(static_cast<size_t>(w) * h, 0u) {}
  [[nodiscard]] uint64_t totalSamples() const;
};

// lround(pow(clamp(x, 0, 1), 1/2.2) * 255), ArrayOutput.cpp:9-12
uint8_t componentToInt(double x);
// SampledPixel::result then componentToInt per channel, for every pixel.
void toRgb8(int width, int height, const double *rgbSum, const uint32_t *counts, uint8_t *out);

// .raw: {u32 1, u32 1, u32 height, u32 width} + per pixel {3 x f64 sum, u32 count}.
void saveRaw(const std::string &path, int width, int height, const double *rgbSum,
             const uint32_t *counts);
void readRawHeader(const std::string &path, int &width, int &height);
// Adds the file into the buffers; throws SizeMismatch if the dimensions differ.
void loadRawAccumulate(const std::string &path, int width, int height, double *rgbSum,
                       uint32_t *counts);

// 8-bit RGB, non-interlaced PNG (stored deflate blocks; no libpng/zlib needed).
void savePng(const std::string &path, int width, int height, const uint8_t *rgb8);

} // namespace ptw

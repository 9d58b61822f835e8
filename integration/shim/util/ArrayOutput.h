// test stand-in (integration/shim/README.md): an ArrayOutput-shaped accumulator - running colour
// sum and sample count per pixel, pixel index x + y * width.
#pragma once
#include "math/Vec3.h"
#include <cstddef>
#include <vector>
class ArrayOutput {
  int width_, height_;
  std::vector<double> sum_;
  std::vector<size_t> n_;

public:
  ArrayOutput(int width, int height)
      : width_(width), height_(height), sum_(size_t(width) * size_t(height) * 3), n_(size_t(width) * size_t(height)) {}
  [[nodiscard]] int width() const { return width_; }
  [[nodiscard]] int height() const { return height_; }
  void addSamples(int x, int y, const Vec3 &colour, int numSamples) {
    const size_t i = size_t(x) + size_t(y) * size_t(width_);
    sum_[i * 3] += colour.x(), sum_[i * 3 + 1] += colour.y(), sum_[i * 3 + 2] += colour.z();
    n_[i] += size_t(numSamples);
  }
  [[nodiscard]] const double *sums() const { return sum_.data(); }
  [[nodiscard]] size_t samplesAt(int x, int y) const { return n_[size_t(x) + size_t(y) * size_t(width_)]; }
  [[nodiscard]] size_t totalSamples() const {
    size_t t = 0;
    for (size_t v : n_) t += v;
    return t;
  }
};

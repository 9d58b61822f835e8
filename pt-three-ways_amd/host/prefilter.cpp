#include "prefilter.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace ptw {
namespace {

// x >= 0 as a float that is not below it (the coefficients of the error bound are rounded UP)
float roundedUp(double x) {
  float f = static_cast<float>(x);
  if (static_cast<double>(f) < x) f = std::nextafter(f, std::numeric_limits<float>::infinity());
  return f;
}

} // namespace

bool prefilterAcceptsOrigin(const double centre[3], double apertureRadius) {
  for (int c = 0; c < 3; ++c)
    if (!std::isfinite(centre[c]) || std::fabs(centre[c]) + std::fabs(apertureRadius) > kPrefilterMaxCoordinate) return false;
  return std::isfinite(apertureRadius);
}

PrefilterData buildPrefilter(const double *triGeom, uint32_t ntri, const double *sphCentreRadius, uint32_t nsph) {
  PrefilterData out;
  for (uint32_t i = 0; i < nsph; ++i) // rays start on sphere surfaces too
    if (!prefilterAcceptsOrigin(sphCentreRadius + 4 * static_cast<size_t>(i), sphCentreRadius[4 * static_cast<size_t>(i) + 3]))
      out.usable = false;
  const uint32_t npairs = (ntri + 1) / 2;
  out.pairs.assign(static_cast<size_t>(std::max<uint32_t>(npairs, 1)) * kPrefilterFloatsPerPair, 0.0f);
  for (uint32_t k = 0; k < npairs; ++k) {
    float *rec = &out.pairs[static_cast<size_t>(k) * kPrefilterFloatsPerPair];
    for (int half = 0; half < 2; ++half) {
      const uint32_t i = std::min(2 * k + static_cast<uint32_t>(half), ntri - 1); // (odd count: B repeats A)
      const double *g = triGeom + 9 * static_cast<size_t>(i);
      double vmax = 0, a1 = 0, a2 = 0;
      for (int c = 0; c < 9; ++c) {
        if (!std::isfinite(g[c]) || std::fabs(g[c]) > kPrefilterMaxCoordinate) out.usable = false;
        rec[2 * c + half] = static_cast<float>(g[c]); // round to nearest: |error| <= 2^-24 |g[c]|
      }
      for (int c = 0; c < 3; ++c) {
        vmax = std::max(vmax, std::fabs(g[c]));
        a1 += std::fabs(g[3 + c]);
        a2 += std::fabs(g[6 + c]);
      }
      // E = margin * (a1 a2 + 2 (|o|_inf + |v0|_inf)(a1 + a2)) = EA + |o|_inf * EB; the floor covers what
      // fp32 underflow can add (an absolute 2^-126 per operation, scaled by factors below 1e12 twice)
      const double ea = kPrefilterMargin * (a1 * a2 + 2.0 * vmax * (a1 + a2)) + 1e-12;
      const double eb = kPrefilterMargin * 2.0 * (a1 + a2);
      rec[18 + half] = roundedUp(ea);
      rec[20 + half] = roundedUp(eb);
    }
  }
  return out;
}

} // namespace ptw

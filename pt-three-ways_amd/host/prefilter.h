// prefilter.h — the fp32 data of the conservative triangle PREFILTER (ptw_render_params.accel ==
// PTW_ACCEL_PREFILTER; SURVEY.md section 8 f4, second half).
//
// The reference tests every ray against every triangle in fp64 (src/dod/Scene.cpp:62-98).  This mode does the
// same - every triangle is looked at for every ray - but looks first in fp32, two triangles per instruction
// (v_pk_fma_f32), and runs the reference's fp64 test only where the fp32 look cannot PROVE that the fp64 test
// would reject.  Nothing that could be a hit is skipped, so every sample is bit-identical to the brute-force
// result; the work per test is not the reference's, which is why the mode is reported separately.
//
// The proof obligation (DESIGN.md 3.4 has the derivation).  With U = tVec . pVec, V = d . qVec, D = e1 . pVec
// (the numerators and the determinant of Scene.cpp:71-88) and W = D - U - V, the reference rejects when
// u = U / D < 0, v = V / D < 0, u + v > 1 (W / D < 0), u > 1, or |D| < epsilon.  If two of U, V, W have
// opposite signs, at least one of them has the opposite sign of D whatever D's sign is - one of the first three
// clauses holds (or |D| < epsilon does).  The fp32 evaluation of U, V, W (inputs rounded to fp32, the same
// expression trees, at most K = 11 roundings on any path) is within 11 * 2^-24 * M of the real value, where
// M = the expression evaluated on absolute values <= |e1|_1 |e2|_1 + 2 (|o|_inf + |v0|_inf)(|e1|_1 + |e2|_1) for
// a unit direction.  The device rejects a triangle only if min(U, V, W) < -E and max(U, V, W) > E with
// E = 1.0e-6 * that bound (1.5 x the fp32 error; the fp64 evaluation's own error is nine orders smaller), i.e.
// only if the fp64 test is certain to reject.  Everything else - including every NaN - goes to the fp64 test.
#pragma once

#include <cstdint>
#include <vector>

namespace ptw {

// One record = TWO triangles (2k, 2k + 1), component-interleaved so that one 64-bit scalar register pair
// feeds both halves of a packed instruction: v0x(A,B) v0y v0z e1x e1y e1z e2x e2y e2z, then the error-bound
// coefficients EA(A,B), EB(A,B): E = EA + |o|_inf * EB.
constexpr int kPrefilterFloatsPerPair = 22;
constexpr double kPrefilterMargin = 1.0e-6;      // >= 1.5 * 11 * 2^-24
constexpr double kPrefilterMaxCoordinate = 1e12; // beyond this fp32 products could overflow: the mode is refused

struct PrefilterData {
  std::vector<float> pairs; // [(ntri + 1) / 2][kPrefilterFloatsPerPair]; an odd scene's last B half repeats A
  // false: a triangle coordinate, or a point a ray can start from - a sphere's surface (|centre| + radius) -, is
  // not finite or beyond kPrefilterMaxCoordinate.  (The third place rays start from, the camera, is checked per
  // render: prefilterAcceptsOrigin.)  With everything bounded no fp32 product of the prefilter overflows, so its
  // U, V, W are NaN only for a ray that has a NaN in it - which the fp64 test does not hit anything with either.
  bool usable = true;
};

// triGeom: [ntri][9] = v0, e1, e2 (precomputeScene's layout); sphCentreRadius: [nsph][4]
PrefilterData buildPrefilter(const double *triGeom, uint32_t ntri, const double *sphCentreRadius, uint32_t nsph);
// a camera centre (+ its aperture radius) the mode can take
bool prefilterAcceptsOrigin(const double centre[3], double apertureRadius);

} // namespace ptw

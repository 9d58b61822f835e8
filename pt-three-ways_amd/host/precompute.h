// precompute.h — per-primitive data the device wants, derived on the host in strict fp64.
//
// This is the hip way's counterpart of dod::Scene::addTriangle / addSphere
// (src/dod/Scene.cpp:181-195): it turns the SceneBuilder's flat arrays into the HBM layout of
// csrc/ptw_layout.h.  Everything here is a per-primitive constant of the reference algorithm
// (edge vectors, the re-normalised face normal that intersectTriangles returns, the
// OrthoNormalBasis::fromZ of that normal, 1/ior), computed with the reference's operation
// order; this translation unit is built with -ffp-contract=off.
#pragma once

#include "../../include/ptw.h"
#include "../csrc/ptw_layout.h"

#include <vector>

namespace ptw {

struct DeviceSceneData {
  std::vector<double> triGeom;      // [ntri][9]: v0, e1, e2
  std::vector<TriShade> triShade;   // [ntri]
  std::vector<SphereRec> spheres;   // [nsph]
  std::vector<double> triCompact;   // [ntri][kTriCompactDoubles]
  std::vector<double> matTable;     // [nmat][kMatDoubles]
  std::vector<uint32_t> triMaterial; // [ntri]  (for the intersect KAT entry point)
  std::vector<uint32_t> sphMaterial; // [nsph]
  double environment[3];
};

DeviceSceneData precomputeScene(const ptw_scene_view &scene);

// How often a worker wave's UNIT of 64 consecutive triangles holds no triangle that passes the u test of
// Moller-Trumbore (src/dod/Scene.cpp:79-89) - estimated on a fixed pseudo-random sample of rays that start on the
// scene's own triangles (up to 64 rays x 32 units; deterministic, a few milliseconds).  The faces of a mesh follow each
// other in space, so for such scenes most units fail as a whole (ce 0.7) and the worker waves of the SEQUENTIAL
// kernels skip the rest of the test for them (testTriangleUnit, csrc/ptw_trace_common.h); a random soup gives 0.
// `triGeom` = [ntri][9] (v0, e1, e2).  0 for scenes of fewer than two units.
double unitUSkipFraction(const double *triGeom, uint32_t ntri);
// ... and the fraction from which the early-out is switched on (measured: profiles/r06aa_*)
constexpr double kUnitUFirstThreshold = 0.4;

// Seeds std::mt19937(seed): x[0] = seed, x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i.
void seedMt19937(uint32_t seed, uint32_t state[624]);

} // namespace ptw

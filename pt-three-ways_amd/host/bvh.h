// bvh.h — bounding-volume hierarchy for the ACCELERATED mode (ptw_render_params.accel == PTW_ACCEL_BVH).
//
// The reference is deliberately brute force (README.md:5-6: every ray tests every primitive,
// src/dod/Scene.cpp:51-113), and so is everything the headline numbers are measured on.  This mode
// is the separate, separately reported one of SURVEY.md section 8(f4): it culls triangles a ray
// cannot hit and runs the SAME Moller-Trumbore arithmetic on the rest, so the nearest hit - and
// with it every sample - is bit-identical to the brute-force result, while the work the metric
// counts (tests per ray) is not the reference's.
//
// Conservative by construction: a triangle is skipped only when the ray misses a box that
// contains it with a margin far above the rounding error of the slab test and of the hit
// distance, and only when the box's entry distance is STRICTLY beyond the best hit so far, so
// exact ties still reach the reference's tie-break (lowest insertion index).
#pragma once

#include <cstdint>
#include <vector>

namespace ptw {

// One node = its two children's boxes (tested together, nearer child first).
struct BvhNode {
  double lo[2][3], hi[2][3];
  int32_t child[2]; // count == 0: index of the child node; count > 0: first leaf entry
  int32_t count[2]; // triangles in the leaf (0 for an inner child)
};
static_assert(sizeof(BvhNode) == 112, "device layout");

struct Bvh {
  std::vector<BvhNode> nodes;       // nodes[0] is the root pair (empty when there are no triangles)
  std::vector<double> leafGeom;     // [entries][9]: v0, e1, e2 of the triangle (copied: no indirection)
  std::vector<uint32_t> leafIndex;  // [entries]: the triangle's insertion index
  int depth = 0;
};

constexpr int kBvhMaxDepth = 30; // the device traversal stack holds 32 entries
constexpr int kBvhLeafSize = 4;

// triGeom: [ntri][9] = v0, e1 = v1 - v0, e2 = v2 - v0 (as precomputeScene lays it out)
Bvh buildBvh(const double *triGeom, uint32_t ntri);

} // namespace ptw

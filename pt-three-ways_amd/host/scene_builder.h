// scene_builder.h — the hip way's SceneBuilder: what `dod::Scene` is to the reference's
// createScene<SB>() / loadObjFile<SB>() duck-typed concept (src/dod/Scene.h:37-42).
//
// It keeps the reference's three calls (addTriangle / addSphere / setEnvironmentColour, same
// argument meaning, insertion order preserved because it is the nearest-hit tie-break order)
// but stores what the device wants: flat fp64 arrays plus a de-duplicated material table
// (dod::Scene copies one MaterialSpec per primitive, Scene.cpp:181-195; an index is enough).
#pragma once

#include "../../include/ptw.h"
#include "vec3.h"

#include <cstdint>
#include <vector>

namespace ptw {

// MaterialSpec factories, src/util/MaterialSpec.h:13-32.
namespace material {
ptw_material defaults();
double toRadians(double degrees); // angle / 360 * 2 * pi, evaluated left to right
ptw_material makeDiffuse(Vec3d colour);
ptw_material makeSpecular(Vec3d colour, double index);
ptw_material makeLight(Vec3d colour);
ptw_material makeGlossy(Vec3d colour, double index, double coneDegrees);
ptw_material makeReflective(Vec3d colour, double reflectivity, double coneDegrees);
bool equal(const ptw_material &a, const ptw_material &b);
} // namespace material

class SceneBuilder {
public:
  void addTriangle(const Vec3d &v0, const Vec3d &v1, const Vec3d &v2, const ptw_material &mat);
  void addSphere(const Vec3d &centre, double radius, const ptw_material &mat);
  void setEnvironmentColour(const Vec3d &colour);

  [[nodiscard]] uint32_t numTriangles() const { return static_cast<uint32_t>(triMat_.size()); }
  [[nodiscard]] uint32_t numSpheres() const { return static_cast<uint32_t>(sphMat_.size()); }
  // Borrowed pointers into this builder; invalidated by the next add*/set* call.
  [[nodiscard]] ptw_scene_view view() const;

private:
  uint32_t internMaterial(const ptw_material &mat);

  std::vector<double> triVerts_;   // [n][3][3]
  std::vector<uint32_t> triMat_;   // [n]
  std::vector<double> sphCentreR_; // [m][4]
  std::vector<uint32_t> sphMat_;   // [m]
  std::vector<ptw_material> materials_;
  Vec3d environment_{};
};

} // namespace ptw

// The opaque handle of the C ABI is exactly a SceneBuilder.
struct ptw_scene {
  ptw::SceneBuilder builder;
};

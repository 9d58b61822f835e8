#include "scene_builder.h"

#include <cstring>

namespace ptw {
namespace material {

ptw_material defaults() {
  ptw_material m;
  std::memset(&m, 0, sizeof m);
  m.index_of_refraction = 1.0;
  m.reflectivity = -1;
  m.reflection_cone_angle_rad = 0.0;
  return m;
}

double toRadians(double degrees) { return degrees / 360 * 2 * M_PI; }

ptw_material makeDiffuse(Vec3d colour) {
  ptw_material m = defaults();
  colour.store(m.diffuse);
  return m;
}

ptw_material makeSpecular(Vec3d colour, double index) {
  ptw_material m = makeDiffuse(colour);
  m.index_of_refraction = index;
  return m;
}

ptw_material makeLight(Vec3d colour) {
  ptw_material m = defaults();
  colour.store(m.emission);
  return m;
}

ptw_material makeGlossy(Vec3d colour, double index, double coneDegrees) {
  ptw_material m = makeSpecular(colour, index);
  m.reflectivity = -1;
  m.reflection_cone_angle_rad = toRadians(coneDegrees);
  return m;
}

ptw_material makeReflective(Vec3d colour, double reflectivity, double coneDegrees) {
  ptw_material m = makeDiffuse(colour);
  m.index_of_refraction = 1.0;
  m.reflectivity = reflectivity;
  m.reflection_cone_angle_rad = toRadians(coneDegrees);
  return m;
}

bool equal(const ptw_material &a, const ptw_material &b) {
  // Bitwise equality so that +0/-0 or NaN payloads never merge two distinct materials.
  return std::memcmp(&a, &b, sizeof a) == 0;
}

} // namespace material

uint32_t SceneBuilder::internMaterial(const ptw_material &mat) {
  // Scenes have a handful of materials; scan newest-first (consecutive faces share one).
  for (size_t i = materials_.size(); i-- > 0;)
    if (material::equal(materials_[i], mat)) return static_cast<uint32_t>(i);
  materials_.push_back(mat);
  return static_cast<uint32_t>(materials_.size() - 1);
}

void SceneBuilder::addTriangle(const Vec3d &v0, const Vec3d &v1, const Vec3d &v2,
                               const ptw_material &mat) {
  const double v[9] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z};
  triVerts_.insert(triVerts_.end(), v, v + 9);
  triMat_.push_back(internMaterial(mat));
}

void SceneBuilder::addSphere(const Vec3d &centre, double radius, const ptw_material &mat) {
  const double s[4] = {centre.x, centre.y, centre.z, radius};
  sphCentreR_.insert(sphCentreR_.end(), s, s + 4);
  sphMat_.push_back(internMaterial(mat));
}

void SceneBuilder::setEnvironmentColour(const Vec3d &colour) { environment_ = colour; }

ptw_scene_view SceneBuilder::view() const {
  ptw_scene_view v;
  std::memset(&v, 0, sizeof v);
  v.num_triangles = numTriangles();
  v.num_spheres = numSpheres();
  v.num_materials = static_cast<uint32_t>(materials_.size());
  v.tri_vertices = triVerts_.data();
  v.tri_material = triMat_.data();
  v.sph_centre_radius = sphCentreR_.data();
  v.sph_material = sphMat_.data();
  v.materials = materials_.data();
  environment_.store(v.environment);
  return v;
}

} // namespace ptw

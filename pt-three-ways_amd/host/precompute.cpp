#include "precompute.h"
#include "vec3.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ptw {
namespace {

void copyMaterial(const ptw_material &m, double emission[3], double diffuse[3], double &ior,
                  double &invIor, double &reflectivity, double &cone) {
  std::memcpy(emission, m.emission, sizeof m.emission);
  std::memcpy(diffuse, m.diffuse, sizeof m.diffuse);
  ior = m.index_of_refraction;
  invIor = 1.0 / m.index_of_refraction; // iorFrom / iorTo for an outside hit, Norm3.cpp:11
  reflectivity = m.reflectivity;
  cone = m.reflection_cone_angle_rad;
}

} // namespace

DeviceSceneData precomputeScene(const ptw_scene_view &scene) {
  DeviceSceneData out;
  std::memcpy(out.environment, scene.environment, sizeof out.environment);
  // at least one (degenerate) record so that device code may always read slot 0
  out.triGeom.assign(std::max<size_t>(static_cast<size_t>(scene.num_triangles), 1) * 9, 0.0);
  out.triShade.resize(scene.num_triangles);
  out.spheres.resize(scene.num_spheres);
  out.triMaterial.assign(scene.tri_material, scene.tri_material + scene.num_triangles);
  out.sphMaterial.assign(scene.sph_material, scene.sph_material + scene.num_spheres);

  out.triCompact.resize(static_cast<size_t>(scene.num_triangles) * kTriCompactDoubles);
  out.matTable.resize(static_cast<size_t>(scene.num_materials) * kMatDoubles);
  for (uint32_t m = 0; m < scene.num_materials; ++m) {
    double *t = &out.matTable[static_cast<size_t>(m) * kMatDoubles];
    copyMaterial(scene.materials[m], t, t + 3, t[6], t[7], t[8], t[9]);
  }

  for (uint32_t i = 0; i < scene.num_triangles; ++i) {
    const double *tv = scene.tri_vertices + 9 * static_cast<size_t>(i);
    const Vec3d v0(tv), v1(tv + 3), v2(tv + 6);
    const Vec3d e1 = v1 - v0; // TriangleVertices::uVector, TriangleVertices.h:25-27
    const Vec3d e2 = v2 - v0; // TriangleVertices::vVector, :29-31
    double *g = &out.triGeom[9 * static_cast<size_t>(i)];
    v0.store(g), e1.store(g + 3), e2.store(g + 6);

    // faceNormal() stored three times (Scene.cpp:183-185); intersectTriangles then forms
    // (u * (n1 - n0) + v * (n2 - n0) + n0).normalised() (Scene.cpp:100-106).  With n0 == n1 ==
    // n2 the deltas are exactly zero and u, v are finite, so the sum is n0 for every hit.
    const Vec3d face = normalised(cross(e1, e2));
    const Vec3d zero = face - face;
    const Vec3d blended = (0.5 * zero + 0.5 * zero) + face;
    const Vec3d normal = normalised(blended);

    // OrthoNormalBasis::fromZ(normal), OrthoNormalBasis.cpp:44-51
    const double zDotX = normal.x * 1.0 + normal.y * 0.0 + normal.z * 0.0;
    const Vec3d a = std::fabs(zDotX) > 0.9999 ? Vec3d(0, 1, 0) : Vec3d(1, 0, 0);
    const Vec3d bx = normalised(cross(a, normal));
    const Vec3d by = normalised(cross(normal, bx));

    TriShade &r = out.triShade[i];
    std::memset(&r, 0, sizeof r);
    normal.store(r.normal);
    bx.store(r.basisX);
    by.store(r.basisY);
    if (scene.tri_material[i] >= scene.num_materials)
      throw std::runtime_error("triangle material index out of range");
    copyMaterial(scene.materials[scene.tri_material[i]], r.emission, r.diffuse, r.ior, r.invIor,
                 r.reflectivity, r.coneAngle);
    double *c = &out.triCompact[static_cast<size_t>(i) * kTriCompactDoubles];
    normal.store(c), bx.store(c + 3), by.store(c + 6);
    c[kTriMaterialIndex] = static_cast<double>(scene.tri_material[i]);
    c[kTriLobeThreshold] = r.reflectivity >= 0 ? r.reflectivity : (r.ior == 1.0 ? -1.0 : 2.0);
    c[11] = 0.0;
  }

  for (uint32_t i = 0; i < scene.num_spheres; ++i) {
    const double *sp = scene.sph_centre_radius + 4 * static_cast<size_t>(i);
    SphereRec &r = out.spheres[i];
    std::memset(&r, 0, sizeof r);
    r.centre[0] = sp[0], r.centre[1] = sp[1], r.centre[2] = sp[2];
    r.radiusSquared = sp[3] * sp[3]; // dod::Sphere ctor, Sphere.h:11-12
    if (scene.sph_material[i] >= scene.num_materials)
      throw std::runtime_error("sphere material index out of range");
    copyMaterial(scene.materials[scene.sph_material[i]], r.emission, r.diffuse, r.ior, r.invIor,
                 r.reflectivity, r.coneAngle);
  }
  return out;
}

void seedMt19937(uint32_t seed, uint32_t state[624]) {
  state[0] = seed;
  for (uint32_t i = 1; i < 624; ++i) {
    const uint32_t prev = state[i - 1];
    state[i] = 1812433253u * (prev ^ (prev >> 30)) + i;
  }
}

double unitUSkipFraction(const double *g, uint32_t ntri) {
  const uint32_t units = (ntri + 63u) / 64u;
  if (units < 2) return 0.0;
  uint64_t state = 0x9e3779b97f4a7c15ull; // splitmix64: a fixed sample, the same on every host
  auto next = [&state]() {
    uint64_t z = (state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  auto uniform = [&next]() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); };
  const uint32_t perRay = units < 32 ? units : 32;
  uint64_t looked = 0, skipped = 0;
  for (int r = 0; r < 64; ++r) {
    // origin: a point inside a random triangle; direction: uniform on the sphere
    const double *a = g + 9 * static_cast<size_t>(next() % ntri);
    const double o[3] = {a[0] + (a[3] + a[6]) / 3.0, a[1] + (a[4] + a[7]) / 3.0, a[2] + (a[5] + a[8]) / 3.0};
    double d[3], n2;
    do {
      for (double &c : d) c = 2.0 * uniform() - 1.0;
      n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    } while (n2 > 1.0 || n2 < 1e-4);
    const uint32_t first = static_cast<uint32_t>(next() % units);
    for (uint32_t j = 0; j < perRay; ++j) {
      const uint32_t unit = (first + j * (units / perRay)) % units;
      bool any = false;
      for (uint32_t k = unit * 64u; k < unit * 64u + 64u && k < ntri && !any; ++k) {
        const double *t = g + 9 * static_cast<size_t>(k);
        const double *e1 = t + 3, *e2 = t + 6;
        const double p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
        const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if (std::fabs(det) < 1e-7) continue;
        const double u = ((o[0] - t[0]) * p[0] + (o[1] - t[1]) * p[1] + (o[2] - t[2]) * p[2]) / det;
        any = !(u < 0.0 || u > 1.0);
      }
      ++looked;
      skipped += any ? 0 : 1;
    }
  }
  return static_cast<double>(skipped) / static_cast<double>(looked);
}

} // namespace ptw

// integration_check.cpp - EXECUTES the reference-side binding integration/hip/Scene.h on a GPU.
//
// tests/test_integration_stub.py compiles that header against the reference's own headers (CPU
// container only: /root/reference does not travel).  Here it is compiled against the
// interface-shaped stand-ins of integration/shim/ and driven the way src/main/main.cpp drives
// dod::Scene (main.cpp:326-366): a SceneBuilder fed triangle by triangle, `render(camera,
// renderParams, updateFunc)` with an update function that sees the running ArrayOutput
// (src/dod/Scene.cpp:245), and the returned ArrayOutput compared - every sum, every count - with
// a plain ptw_render of the same scene.
//
//   integration_check <scene> <width> <height> <spp> <scenes_dir>
#include "hip/Scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {
[[noreturn]] void die(const std::string &what) {
  std::fprintf(stderr, "integration_check: %s\n", what.c_str());
  std::exit(1);
}
void ok(int rc, const char *what) {
  if (rc != PTW_OK) die(std::string(what) + ": " + ptw_last_error());
}
Vec3 v3(const double *p) { return Vec3(p[0], p[1], p[2]); }
Norm3 n3(const double *p) { return Norm3(p[0], p[1], p[2]); }
} // namespace

int main(int argc, char **argv) {
  if (argc != 6) die("usage: integration_check <scene> <width> <height> <spp> <scenes_dir>");
  const std::string name = argv[1];
  const int width = std::atoi(argv[2]), height = std::atoi(argv[3]), spp = std::atoi(argv[4]);

  // the scene catalogue of this repository stands in for createScene() (main.cpp:291-309) ...
  ptw_scene *catalogue = nullptr;
  ok(ptw_scene_create(&catalogue), "ptw_scene_create");
  ptw_camera cam;
  ok(ptw_scene_build_named(catalogue, name.c_str(), argv[5], width, height, &cam), "ptw_scene_build_named");
  ptw_scene_view view;
  ok(ptw_scene_view_of(catalogue, &view), "ptw_scene_view_of");

  // ... and feeds the reference-side SceneBuilder primitive by primitive, in insertion order
  hip::Scene scene;
  auto spec = [&](uint32_t m) {
    const ptw_material &pm = view.materials[m];
    MaterialSpec s;
    s.emission = v3(pm.emission), s.diffuse = v3(pm.diffuse);
    s.indexOfRefraction = pm.index_of_refraction, s.reflectivity = pm.reflectivity;
    s.reflectionConeAngleRadians = pm.reflection_cone_angle_rad;
    return s;
  };
  // dod::Scene tests spheres before triangles whatever the order of the calls; keep the catalogue's
  for (uint32_t i = 0; i < view.num_spheres; ++i)
    scene.addSphere(v3(view.sph_centre_radius + 4 * i), view.sph_centre_radius[4 * i + 3], spec(view.sph_material[i]));
  for (uint32_t i = 0; i < view.num_triangles; ++i) {
    const double *t = view.tri_vertices + 9 * size_t(i);
    scene.addTriangle(v3(t), v3(t + 3), v3(t + 6), spec(view.tri_material[i]));
  }
  scene.setEnvironmentColour(v3(view.environment));

  const Camera camera(v3(cam.centre), OrthoNormalBasis(n3(cam.axis_x), n3(cam.axis_y), n3(cam.axis_z)),
                      cam.aspect_ratio, cam.camera_plane_dist, cam.reciprocal_height, cam.reciprocal_width,
                      cam.aperture_radius, cam.focal_distance);
  RenderParams rp{};
  rp.width = width, rp.height = height, rp.samplesPerPixel = spp, rp.seed = 1;
  rp.maxDepth = 5, rp.firstBounceUSamples = 4, rp.firstBounceVSamples = 4, rp.preview = false;

  int updates = 0;
  size_t lastTotal = 0;
  bool monotonic = true;
  const ArrayOutput result = scene.render(camera, rp, [&](ArrayOutput &running) {
    ++updates;
    const size_t total = running.totalSamples();
    monotonic = monotonic && total > lastTotal && running.width() == width && running.height() == height;
    lastTotal = total;
  });

  // the same render through the plain entry point
  ptw_render_params p;
  ptw_default_params(&p);
  p.width = width, p.height = height, p.samples_per_pixel = spp, p.seed = 1;
  std::vector<double> sum(size_t(width) * size_t(height) * 3);
  std::vector<uint32_t> n(size_t(width) * size_t(height));
  ok(ptw_render(&view, &cam, &p, sum.data(), n.data(), nullptr, nullptr), "ptw_render");

  if (std::memcmp(result.sums(), sum.data(), sum.size() * sizeof(double)) != 0) die("radiance sums differ from ptw_render");
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      if (result.samplesAt(x, y) != n[size_t(x) + size_t(y) * size_t(width)]) die("sample counts differ from ptw_render");
  if (result.totalSamples() != size_t(width) * size_t(height) * size_t(spp)) die("total samples");
  if (updates < 2 || !monotonic || lastTotal != result.totalSamples())
    die("updateFunc: " + std::to_string(updates) + " calls, last total " + std::to_string(lastTotal));
  // errors surface as exceptions on the reference side (main.cpp lets them terminate)
  bool threw = false;
  try {
    RenderParams bad = rp;
    bad.firstBounceUSamples = 0;
    (void)scene.render(camera, bad, [](ArrayOutput &) {});
  } catch (const std::runtime_error &) {
    threw = true;
  }
  if (!threw) die("an invalid request did not throw");
  ptw_scene_destroy(catalogue);
  std::printf("INTEGRATION_OK scene=%s %dx%d spp=%d updates=%d samples=%zu\n", name.c_str(), width, height, spp,
              updates, result.totalSamples());
  return 0;
}

#include "scenes.h"
#include "obj_loader.h"

#include <cmath>
#include <cstring>

namespace ptw {

ptw_camera makeCamera(const Vec3d &eye, const Vec3d &lookAt, const Vec3d &up, int width,
                      int height, double verticalFovDegrees) {
  // OrthoNormalBasis::fromZY(z, y): x = normalise(y cross z), y' = z cross x
  // (src/math/OrthoNormalBasis.cpp:34-38)
  const Vec3d z = normalised(lookAt - eye);
  const Vec3d x = normalised(cross(normalised(up), z));
  const Vec3d y = cross(z, x);
  ptw_camera c;
  std::memset(&c, 0, sizeof c);
  eye.store(c.centre);
  x.store(c.axis_x);
  y.store(c.axis_y);
  z.store(c.axis_z);
  c.aspect_ratio = static_cast<double>(width) / height;
  c.camera_plane_dist = 1.0 / std::tan(verticalFovDegrees * M_PI / 360.0);
  c.reciprocal_height = 1.0 / height;
  c.reciprocal_width = 1.0 / width;
  return c;
}

void setFocus(ptw_camera &camera, const Vec3d &focalPoint, double apertureRadius) {
  camera.focal_distance = length(focalPoint - Vec3d(camera.centre));
  camera.aperture_radius = apertureRadius;
}

Vec3d hexColour(uint32_t hex) {
  auto channel = [](uint32_t v) { return std::pow((v & 0xffu) / 255.0, 2.2); };
  return {channel(hex >> 16u), channel(hex >> 8u), channel(hex)};
}

void addCube(SceneBuilder &sb, const Vec3d &low, const Vec3d &high, const ptw_material &mat) {
  // corner k: bit 4 selects low.x, bit 2 low.y, bit 1 low.z (else the high coordinate)
  auto corner = [&](unsigned k) {
    return Vec3d((k & 4u) ? low.x : high.x, (k & 2u) ? low.y : high.y, (k & 1u) ? low.z : high.z);
  };
  static const unsigned char faces[12][3] = {{0, 4, 6}, {0, 6, 2}, {1, 5, 7}, {1, 7, 3},
                                             {0, 4, 5}, {0, 5, 1}, {2, 6, 7}, {2, 7, 3},
                                             {0, 2, 3}, {0, 3, 1}, {4, 6, 7}, {4, 7, 5}};
  for (const auto &f : faces) sb.addTriangle(corner(f[0]), corner(f[1]), corner(f[2]), mat);
}

namespace {

struct CameraSpec {
  Vec3d eye, lookAt, up;
  double vfov;
};

ptw_camera cameraFor(const CameraSpec &s, int w, int h) {
  return makeCamera(s.eye, s.lookAt, s.up, w, h, s.vfov);
}

// main.cpp:69-86
ptw_camera cornell(SceneBuilder &sb, const std::string &dir, int w, int h) {
  loadObjFile(dir, "CornellBox-Original.obj", sb);
  sb.addSphere({-0.38, 0.281, 0.38}, 0.28, material::makeReflective({0.999, 0.999, 0.999}, 0.95, 5));
  sb.setEnvironmentColour(Vec3d(0.725, 0.71, 0.68) * 0.1);
  ptw_camera cam = cameraFor({{0, 1, 3}, {0, 1, 0}, {0, 1, 0}, 50.0}, w, h);
  setFocus(cam, {0, 0, 0}, 0.01);
  return cam;
}

// main.cpp:88-114
ptw_camera suzanne(SceneBuilder &sb, const std::string &dir, int w, int h) {
  loadObjFile(dir, "suzanne.obj", sb);
  const ptw_material light = material::makeLight({4, 4, 4});
  sb.addSphere({0.5, 1, 3}, 1, light);
  sb.addSphere({1, 1, 3}, 1, light);
  const ptw_material backdrop = material::makeDiffuse({0.20, 0.30, 0.36});
  const Vec3d tl(-5, -5, -1), tr(5, -5, -1), bl(-5, 5, -1), br(5, 5, -1);
  sb.addTriangle(tl, tr, bl, backdrop);
  sb.addTriangle(tr, bl, br, backdrop);
  const Vec3d lookAt(1, -0.6, 0.4);
  ptw_camera cam = cameraFor({{1, -0.45, 4}, lookAt, {0, 1, 0}, 40.0}, w, h);
  setFocus(cam, lookAt, 0.01);
  return cam;
}

// main.cpp:116-137
ptw_camera ce(SceneBuilder &sb, const std::string &dir, int w, int h) {
  loadObjFile(dir, "ce.obj", sb);
  sb.addSphere({0, 1.6, 0}, 1.0, material::makeLight(Vec3d(1, 1, 1) * 10));
  sb.addSphere({-0.2, 5.9, -0.3}, 5.0, material::makeLight(Vec3d(2.27, 3, 2.97) * 0.25));
  sb.addSphere({0, 0, 0}, 10, material::makeDiffuse({0.2, 0.2, 0.2}));
  const Vec3d lookAt(0, 0, 0);
  ptw_camera cam = cameraFor({{0.27, 1.15, 0.36}, lookAt, {0, 0, -1}, 40.0}, w, h);
  setFocus(cam, lookAt, 0.01);
  return cam;
}

// Shared by single-sphere and multi-sphere: camera and the big area light (main.cpp:141-153).
ptw_camera sphereStage(SceneBuilder &sb, int w, int h) {
  const Vec3d eye(0, 0, -3.2);
  const double lightRadius = 3.0;
  const Vec3d lightOffset(6, 6, 0);
  sb.addSphere(eye + lightOffset - Vec3d(0, 0, lightRadius), lightRadius,
               material::makeLight(Vec3d(1, 1, 1) * 8));
  return cameraFor({eye, {0, 0, 0}, {0, 1, 0}, 40.0}, w, h);
}

// main.cpp:139-165
ptw_camera singleSphere(SceneBuilder &sb, int w, int h) {
  ptw_camera cam = sphereStage(sb, w, h);
  ptw_material ball = material::makeDiffuse({0.2, 0.2, 0.2});
  ball.index_of_refraction = 1.3;
  ball.reflection_cone_angle_rad = 0.05;
  sb.addSphere({0, 0, 0}, 1, ball);
  sb.addSphere({0, 0, 0}, 10, material::makeDiffuse({0.2, 0.2, 0.5}));
  return cam;
}

// main.cpp:167-201
ptw_camera multiSphere(SceneBuilder &sb, int w, int h) {
  ptw_camera cam = sphereStage(sb, w, h);
  const double radius = 1.0 / 5.0;
  const double gap = radius * 2.15;
  for (int y = -2; y <= 2; ++y)
    for (int x = -4; x <= 4; ++x) {
      ptw_material m = material::makeDiffuse({0.90, 0.91, 0.92});
      m.reflection_cone_angle_rad = 0.075 * (x + 4);
      m.index_of_refraction = 1.0 + 0.15 * (y + 2);
      sb.addSphere({x * gap, y * gap, 0}, radius, m);
    }
  sb.addSphere({0, 0, 0}, 10, material::makeDiffuse({0.2, 0.2, 0.5}));
  return cam;
}

// main.cpp:203-230 (after @fogleman's pt example1.go)
ptw_camera example1(SceneBuilder &sb, int w, int h) {
  struct Ball {
    Vec3d centre;
    double radius;
    uint32_t colour;
  };
  static const Ball balls[] = {{{1.5, 1.25, 0}, 1.25, 0x004358},
                               {{-1, 1, 2}, 1.0, 0xffe11a},
                               {{-2.5, 0.75, 0}, 0.75, 0xfd7400},
                               {{-0.75, 0.5, -1}, 0.5, 0x000000}};
  for (const Ball &b : balls)
    sb.addSphere(b.centre, b.radius, material::makeSpecular(hexColour(b.colour), 1.3));
  addCube(sb, {-10, -1, -10}, {10, 0, 10}, material::makeGlossy({1, 1, 1}, 1.1, 10.0));
  sb.addSphere({-1.5, 4, 0}, 0.5, material::makeLight(Vec3d(1, 1, 1) * 30));
  ptw_camera cam = cameraFor({{0, 2, -5}, {0, 0.25, 3}, {0, 1, 0}, 45.0}, w, h);
  setFocus(cam, {-0.75, 1, -1}, 0.1);
  return cam;
}

// main.cpp:232-289.  The owl is a 17 x 21 dot matrix; bit i of a row = column i lit.
ptw_camera bbcOwl(SceneBuilder &sb, int w, int h) {
  static const uint32_t rows[21] = {0x15555, 0x08282, 0x11111, 0x02828, 0x11011, 0x08282, 0x14105,
                                    0x0200a, 0x11555, 0x000aa, 0x10155, 0x000aa, 0x10154, 0x002a8,
                                    0x10550, 0x00aa0, 0x11540, 0x02a80, 0x14440, 0x08aaa, 0x10000};
  constexpr int owlHeight = 21;
  constexpr size_t owlWidth = 17;
  const double spacing = 0.1;
  const double size = spacing * 0.7;
  double y = owlHeight * spacing - spacing / 2;
  for (uint32_t row : rows) {
    double x = owlWidth * spacing / 2;
    for (size_t col = 0; col < owlWidth; ++col) {
      if (row & (1u << col))
        sb.addSphere({x, y, 0}, size, material::makeSpecular(hexColour(0xfeffd5), 1.3));
      x -= spacing;
    }
    y -= spacing;
  }
  ptw_material plane = material::makeReflective({0.2, 0.2, 0.2}, 0.75, 3.0);
  plane.index_of_refraction = 1.5;
  addCube(sb, {-10, -1, -10}, {10, 0, 10}, plane);
  sb.addSphere({-1.5, 4.0, -1}, 0.75, material::makeLight(Vec3d(1, 1, 1) * 30));
  sb.setEnvironmentColour(Vec3d(0.2, 0.2, 0.5) * 0.05);
  ptw_camera cam = cameraFor({{4, 2.0, -5}, {0, 0.5, 0}, {0, 1, 0}, 33.0}, w, h);
  setFocus(cam, {0, 0.5, 0}, 0.1);
  return cam;
}

} // namespace

ptw_camera buildNamedScene(SceneBuilder &sb, const std::string &name,
                           const std::string &scenesDir, int width, int height) {
  if (name == "cornell") return cornell(sb, scenesDir, width, height);
  if (name == "suzanne") return suzanne(sb, scenesDir, width, height);
  if (name == "ce") return ce(sb, scenesDir, width, height);
  if (name == "single-sphere") return singleSphere(sb, width, height);
  if (name == "multi-sphere") return multiSphere(sb, width, height);
  if (name == "example1") return example1(sb, width, height);
  if (name == "bbc-owl") return bbcOwl(sb, width, height);
  throw UnknownScene("Unknown scene " + name);
}

} // namespace ptw

// integration/hip/Scene.h — the reference-side binding of the hip way: the file a pt-three-ways
// maintainer would add as src/hip/Scene.h.  It is written against the REFERENCE's headers
// (math/Camera.h, util/ArrayOutput.h, util/MaterialSpec.h, util/RenderParams.h) and the C ABI of
// include/ptw.h, and links with -lptw_hip.  tests/test_integration_stub.py compiles it against
// /root/reference/src where that tree is present.
//
// hip::Scene satisfies the duck-typed SceneBuilder concept used by createScene<SB>() and
// loadObjFile<SB>() (src/dod/Scene.h:37-42) and has dod::Scene::render's signature
// (src/dod/Scene.h:44-46), so the patch to src/main/main.cpp is one more branch in doRender()
// (main.cpp:360-363), a copy of the "dod" one:
//
//     } else if (way == "hip") {
//       hip::Scene scene;
//       auto camera = createScene(scene, sceneName, renderParams);
//       return scene.render(camera, renderParams, throttledSave);
//     }
//
// and ONE line in src/math/Camera.h, because Camera keeps its state private (Camera.h:11-18) and
// has no accessors:
//
//     class Camera {
//       friend class hip::Scene;        // <- the one-line patch (plus `namespace hip { class Scene; }`
//       Vec3 centre_;                   //    above the class)
//
// With that, render() takes the caller's Camera as it is - any camera, not only those of the
// built-in scenes.
#pragma once

#include "math/Camera.h"
#include "util/ArrayOutput.h"
#include "util/MaterialSpec.h"
#include "util/RenderParams.h"

#include <ptw.h>

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <vector>

namespace hip {

class Scene {
  ptw_scene *scene_{};

  static ptw_material toPod(const MaterialSpec &m) {
    return ptw_material{{m.emission.x(), m.emission.y(), m.emission.z()},
                        {m.diffuse.x(), m.diffuse.y(), m.diffuse.z()},
                        m.indexOfRefraction,
                        m.reflectivity,
                        m.reflectionConeAngleRadians};
  }
  static void check(int rc) {
    if (rc != PTW_OK) throw std::runtime_error(ptw_last_error());
  }
  static void put(double out[3], const Vec3 &v) { out[0] = v.x(), out[1] = v.y(), out[2] = v.z(); }
  static void put(double out[3], const Norm3 &v) { out[0] = v.x(), out[1] = v.y(), out[2] = v.z(); }

  struct UpdateState {
    const std::function<void(ArrayOutput &)> *updateFunc;
    int width, height;
  };
  static ArrayOutput toArrayOutput(int width, int height, const double *sum, const uint32_t *n) {
    ArrayOutput out(width, height);
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const size_t i = size_t(x) + size_t(y) * size_t(width);
        out.addSamples(x, y, Vec3(sum[i * 3], sum[i * 3 + 1], sum[i * 3 + 2]), int(n[i]));
      }
    return out;
  }
  static int onUpdate(void *user, uint64_t, uint64_t, const double *sum, const uint32_t *n) {
    auto *st = static_cast<UpdateState *>(user);
    ArrayOutput running = toArrayOutput(st->width, st->height, sum, n);
    (*st->updateFunc)(running); // updateFunc(output), src/dod/Scene.cpp:245
    return 0;
  }

public:
  Scene() { check(ptw_scene_create(&scene_)); }
  ~Scene() { ptw_scene_destroy(scene_); }
  Scene(const Scene &) = delete;
  Scene &operator=(const Scene &) = delete;

  // --- the SceneBuilder concept (src/dod/Scene.h:37-42) ---
  void addTriangle(const Vec3 &v0, const Vec3 &v1, const Vec3 &v2, const MaterialSpec &m) {
    double a[3], b[3], c[3];
    put(a, v0), put(b, v1), put(c, v2);
    const ptw_material pod = toPod(m);
    check(ptw_scene_add_triangle(scene_, a, b, c, &pod));
  }
  void addSphere(const Vec3 &centre, double radius, const MaterialSpec &m) {
    double c[3];
    put(c, centre);
    const ptw_material pod = toPod(m);
    check(ptw_scene_add_sphere(scene_, c, radius, &pod));
  }
  void setEnvironmentColour(const Vec3 &colour) {
    double c[3];
    put(c, colour);
    check(ptw_scene_set_environment(scene_, c));
  }

  // Camera's private state (src/math/Camera.h:11-18) as the POD the C ABI takes - the reason for
  // the `friend class hip::Scene;` line.
  static ptw_camera toPod(const Camera &camera) {
    ptw_camera c;
    put(c.centre, camera.centre_);
    put(c.axis_x, camera.axis_.x());
    put(c.axis_y, camera.axis_.y());
    put(c.axis_z, camera.axis_.z());
    c.aspect_ratio = camera.aspectRatio_;
    c.camera_plane_dist = camera.cameraPlaneDist_;
    c.reciprocal_height = camera.reciprocalHeight_;
    c.reciprocal_width = camera.reciprocalWidth_;
    c.aperture_radius = camera.apertureRadius_;
    c.focal_distance = camera.focalDistance_;
    return c;
  }
  static ptw_render_params toPod(const RenderParams &rp) {
    ptw_render_params p;
    ptw_default_params(&p);
    p.width = rp.width, p.height = rp.height, p.preview = rp.preview;
    p.samples_per_pixel = rp.samplesPerPixel, p.max_depth = rp.maxDepth;
    p.first_bounce_u = rp.firstBounceUSamples, p.first_bounce_v = rp.firstBounceVSamples;
    p.seed = rp.seed;
    return p;
  }
  [[nodiscard]] ptw_scene_view view() const {
    ptw_scene_view v;
    check(ptw_scene_view_of(scene_, &v));
    return v;
  }

  // --- dod::Scene::render (src/dod/Scene.h:44-46) ---
  ArrayOutput render(const Camera &camera, const RenderParams &renderParams,
                     const std::function<void(ArrayOutput &output)> &updateFunc) {
    const ptw_scene_view v = view();
    const ptw_camera cam = toPod(camera);
    const ptw_render_params p = toPod(renderParams);
    std::vector<double> sum(size_t(p.width) * size_t(p.height) * 3);
    std::vector<uint32_t> n(size_t(p.width) * size_t(p.height));
    UpdateState state{&updateFunc, p.width, p.height};
    ptw_render_options options{};
    options.update = onUpdate;
    options.update_user = &state;
    check(ptw_render_ex(&v, &cam, &p, sum.data(), n.data(), &options));
    return toArrayOutput(p.width, p.height, sum.data(), n.data());
  }
};

} // namespace hip

// ref_driver.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" driver around the REFERENCE's own sources, compiled where they lie under
// /root/reference/src by oracle/Makefile into oracle/_ref/libptw_ref*.so (git-ignored).  It
// contains no reference code: it #includes the reference's headers by path at build time and
// calls the reference's public functions (dod::Scene::addTriangle/addSphere/radiance/
// intersect*, Camera::randomRay, ArrayOutput, std::mt19937).  It exists to (1) validate the C
// restatement in ptw_oracle.c, (2) generate the golden vectors under tests/golden/
// (oracle/make_golden.py) and (3) optionally serve as bench.py's cpu_baseline ("reference").
//
// What of the reference is NOT buildable here (no stand-ins are written for missing deps):
//  - src/util/ObjLoader* needs <ctre.hpp> (CTRE, un-vendored)        -> scenes are fed in
//    through addTriangle/addSphere from the build's own loader;
//  - src/util/Progressifier.cpp needs <date/date.h>, so dod::Scene::render (the scheduler,
//    Scene.cpp:197-254) cannot link; this driver runs the worker lambda's pass loop
//    (Scene.cpp:209-219) itself.  -fvisibility=hidden + --gc-sections drop render().
//  - src/main/main.cpp needs clara + date + libpng dev headers.
#include "dod/Scene.h"
#include "math/Camera.h"
#include "util/ArrayOutput.h"
#include "util/MaterialSpec.h"
#include "util/RenderParams.h"

#include "../include/ptw.h"

#include <atomic>
#include <cstring>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
Vec3 V(const double *p) { return Vec3(p[0], p[1], p[2]); }

MaterialSpec toSpec(const ptw_material &m) {
  MaterialSpec s;
  s.emission = V(m.emission);
  s.diffuse = V(m.diffuse);
  s.indexOfRefraction = m.index_of_refraction;
  s.reflectivity = m.reflectivity;
  s.reflectionConeAngleRadians = m.reflection_cone_angle_rad;
  return s;
}

struct RefScene {
  dod::Scene scene;
};

// Camera description in the terms of the reference's ctor + setFocus (Camera.h:40-51).
struct CamDesc {
  double eye[3], look_at[3], up[3];
  double vfov_degrees;
  double focus[3];
  double aperture;
  int32_t has_focus;
};

Camera makeCamera(const CamDesc &d, int width, int height) {
  Camera camera(V(d.eye), V(d.look_at), V(d.up).normalised(), width, height, d.vfov_degrees);
  if (d.has_focus) camera.setFocus(V(d.focus), d.aperture);
  return camera;
}

void putHit(const std::optional<dod::IntersectionRecord> &ir, double *out) {
  if (!ir) {
    out[0] = -1;
    for (int i = 1; i < 17; ++i) out[i] = 0;
    return;
  }
  out[0] = ir->hit.distance;
  out[1] = ir->hit.inside ? 1.0 : 0.0;
  out[2] = ir->hit.position.x(), out[3] = ir->hit.position.y(), out[4] = ir->hit.position.z();
  out[5] = ir->hit.normal.x(), out[6] = ir->hit.normal.y(), out[7] = ir->hit.normal.z();
  const MaterialSpec &m = ir->material;
  out[8] = m.emission.x(), out[9] = m.emission.y(), out[10] = m.emission.z();
  out[11] = m.diffuse.x(), out[12] = m.diffuse.y(), out[13] = m.diffuse.z();
  out[14] = m.indexOfRefraction, out[15] = m.reflectivity, out[16] = m.reflectionConeAngleRadians;
}

RenderParams toParams(const ptw_render_params &p) {
  RenderParams rp;
  rp.width = p.width;
  rp.height = p.height;
  rp.preview = p.preview != 0;
  rp.samplesPerPixel = p.samples_per_pixel;
  rp.maxDepth = p.max_depth;
  rp.firstBounceUSamples = p.first_bounce_u;
  rp.firstBounceVSamples = p.first_bounce_v;
  rp.seed = p.seed;
  return rp;
}

// One pass exactly as the worker lambda does it (src/dod/Scene.cpp:209-219).
// `rowEnd` < height: only the PREFIX [0, rowEnd) of the frame's rows (a pass walks the pixels in
// row-major order, so the prefix is exactly what the full pass produces for those rows; bench.py's
// bounded parity legs on the large frames).
void renderPass(const dod::Scene &scene, const Camera &camera, const RenderParams &rp, int pass,
                int firstPass, double *radiance_out, uint32_t *words_out, int rowEnd) {
  std::mt19937 rng(rp.seed + firstPass + pass);
  std::mt19937 shadow = rng;
  for (auto y = 0; y < rowEnd; ++y) {
    for (auto x = 0; x < rp.width; ++x) {
      auto ray = camera.randomRay(x, y, rng);
      Vec3 c = scene.radiance(rng, ray, 0, rp);
      size_t pix = static_cast<size_t>(x) + static_cast<size_t>(y) * rp.width;
      radiance_out[pix * 3 + 0] = c.x();
      radiance_out[pix * 3 + 1] = c.y();
      radiance_out[pix * 3 + 2] = c.z();
      if (words_out) {
        uint32_t n = 0;
        while (!(shadow == rng)) {
          shadow();
          ++n;
        }
        words_out[pix] = n;
      }
    }
  }
}
} // namespace

REF_API void *ref_scene_create() { return new RefScene(); }
REF_API void ref_scene_destroy(void *s) { delete static_cast<RefScene *>(s); }
REF_API void ref_scene_add_triangle(void *s, const double *v0, const double *v1, const double *v2,
                                    const ptw_material *m) {
  static_cast<RefScene *>(s)->scene.addTriangle(V(v0), V(v1), V(v2), toSpec(*m));
}
REF_API void ref_scene_add_sphere(void *s, const double *c, double radius, const ptw_material *m) {
  static_cast<RefScene *>(s)->scene.addSphere(V(c), radius, toSpec(*m));
}
REF_API void ref_scene_set_environment(void *s, const double *c) {
  static_cast<RefScene *>(s)->scene.setEnvironmentColour(V(c));
}
// Feed a whole flattened scene in insertion order.  Spheres and triangles live in separate
// vectors in dod::Scene, so their relative order does not matter.
REF_API void ref_scene_from_view(void *s, const ptw_scene_view *v) {
  auto &scene = static_cast<RefScene *>(s)->scene;
  for (uint32_t i = 0; i < v->num_triangles; ++i) {
    const double *t = v->tri_vertices + 9 * static_cast<size_t>(i);
    scene.addTriangle(V(t), V(t + 3), V(t + 6), toSpec(v->materials[v->tri_material[i]]));
  }
  for (uint32_t i = 0; i < v->num_spheres; ++i) {
    const double *sp = v->sph_centre_radius + 4 * static_cast<size_t>(i);
    scene.addSphere(V(sp), sp[3], toSpec(v->materials[v->sph_material[i]]));
  }
  scene.setEnvironmentColour(V(v->environment));
}

// hit_out[17]: distance(-1 miss), inside, pos xyz, normal xyz, material (9 doubles).
// ray6 = two points: origin and a second point (Ray::fromTwoPoints, as the reference's tests).
REF_API void ref_intersect_two_points(void *s, const double *p1, const double *p2, int which,
                                      double nearer_than, double *hit_out) {
  auto &scene = static_cast<RefScene *>(s)->scene;
  Ray ray = Ray::fromTwoPoints(V(p1), V(p2));
  if (which == 0)
    putHit(scene.intersect(ray), hit_out);
  else if (which == 1)
    putHit(scene.intersectSpheres(ray, nearer_than), hit_out);
  else
    putHit(scene.intersectTriangles(ray, nearer_than), hit_out);
}
// The normalised direction Ray::fromTwoPoints produces (to feed identical rays to the oracle).
REF_API void ref_ray_from_two_points(const double *p1, const double *p2, double *ray_out) {
  Ray ray = Ray::fromTwoPoints(V(p1), V(p2));
  ray_out[0] = ray.origin().x(), ray_out[1] = ray.origin().y(), ray_out[2] = ray.origin().z();
  ray_out[3] = ray.direction().x(), ray_out[4] = ray.direction().y(),
  ray_out[5] = ray.direction().z();
}

REF_API void ref_camera_ray(const CamDesc *d, int width, int height, int px, int py,
                            uint32_t seed, double *ray_out) {
  Camera camera = makeCamera(*d, width, height);
  std::mt19937 rng(seed);
  Ray ray = camera.randomRay(px, py, rng);
  ray_out[0] = ray.origin().x(), ray_out[1] = ray.origin().y(), ray_out[2] = ray.origin().z();
  ray_out[3] = ray.direction().x(), ray_out[4] = ray.direction().y(),
  ray_out[5] = ray.direction().z();
}

REF_API void ref_mt_words(uint32_t seed, uint32_t n, uint32_t *out) {
  std::mt19937 rng(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = static_cast<uint32_t>(rng());
}
REF_API void ref_unit_doubles(uint32_t seed, uint32_t n, double *out) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<> unit(0, 1.0);
  for (uint32_t i = 0; i < n; ++i) out[i] = unit(rng);
}
REF_API void ref_uniform_doubles(uint32_t seed, double a, double b, uint32_t n, double *out) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<> dist(a, b);
  for (uint32_t i = 0; i < n; ++i) out[i] = dist(rng);
}

REF_API void ref_render_pass(void *s, const CamDesc *d, const ptw_render_params *p, int pass,
                             double *radiance_out, uint32_t *words_out) {
  auto &scene = static_cast<RefScene *>(s)->scene;
  RenderParams rp = toParams(*p);
  Camera camera = makeCamera(*d, rp.width, rp.height);
  const int rowEnd = (p->row_begin == 0 && p->row_end > 0 && p->row_end < rp.height) ? p->row_end : rp.height;
  renderPass(scene, camera, rp, pass, p->first_pass, radiance_out, words_out, rowEnd);
}

// All passes on `threads` threads (one whole-frame pass per thread at a time, like the
// reference's std::async tasks), merged with ArrayOutput::operator+= in pass order.
REF_API void ref_render(void *s, const CamDesc *d, const ptw_render_params *p, int threads,
                        double *rgb_sum, uint32_t *counts) {
  auto &scene = static_cast<RefScene *>(s)->scene;
  RenderParams rp = toParams(*p);
  Camera camera = makeCamera(*d, rp.width, rp.height);
  const int spp = rp.samplesPerPixel;
  const size_t npix = static_cast<size_t>(rp.width) * rp.height;
  std::vector<std::unique_ptr<ArrayOutput>> passes(spp);
  std::atomic<int> next{0};
  auto work = [&] {
    for (;;) {
      int pass = next.fetch_add(1);
      if (pass >= spp) break;
      auto out = std::make_unique<ArrayOutput>(rp.width, rp.height);
      std::mt19937 rng(rp.seed + p->first_pass + pass);
      for (auto y = 0; y < rp.height; ++y)
        for (auto x = 0; x < rp.width; ++x) {
          auto ray = camera.randomRay(x, y, rng);
          out->addSamples(x, y, scene.radiance(rng, ray, 0, rp), 1);
        }
      passes[pass] = std::move(out);
    }
  };
  std::vector<std::thread> pool;
  for (int i = 0; i < std::max(1, threads); ++i) pool.emplace_back(work);
  for (auto &t : pool) t.join();
  // Accumulate the per-pass buffers in pass order (the order ArrayOutput::operator+= would see
  // with --max-cpus 1).  Each pass pixel has n == 1, so rawPixelAt() == the raw sum.
  for (size_t pix = 0; pix < npix; ++pix) {
    int x = static_cast<int>(pix % rp.width), y = static_cast<int>(pix / rp.width);
    Vec3 sum;
    for (auto &pass : passes) sum += pass->rawPixelAt(x, y); // n == 1 => result() == colour*1.0
    rgb_sum[pix * 3 + 0] += sum.x();
    rgb_sum[pix * 3 + 1] += sum.y();
    rgb_sum[pix * 3 + 2] += sum.z();
    counts[pix] += static_cast<uint32_t>(spp);
  }
}

// ArrayOutput surface: build an ArrayOutput whose pixel (x,y) holds (sum, n) and use the
// reference's save()/pixelAt()/load().
namespace {
ArrayOutput fromBuffers(int w, int h, const double *rgb_sum, const uint32_t *counts) {
  ArrayOutput ao(w, h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      size_t pix = static_cast<size_t>(x) + static_cast<size_t>(y) * w;
      ao.addSamples(x, y, V(rgb_sum + pix * 3), static_cast<int>(counts[pix]));
    }
  return ao;
}
} // namespace
REF_API int ref_raw_save(const char *path, int w, int h, const double *rgb_sum,
                         const uint32_t *counts) {
  try {
    fromBuffers(w, h, rgb_sum, counts).save(path);
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}
REF_API void ref_pixels_rgb8(int w, int h, const double *rgb_sum, const uint32_t *counts,
                             uint8_t *out) {
  ArrayOutput ao = fromBuffers(w, h, rgb_sum, counts);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      auto px = ao.pixelAt(x, y);
      size_t pix = static_cast<size_t>(x) + static_cast<size_t>(y) * w;
      out[pix * 3 + 0] = px[0], out[pix * 3 + 1] = px[1], out[pix * 3 + 2] = px[2];
    }
}
// Loads a .raw with the reference's ArrayOutput::load and reports means + total samples.
REF_API int ref_raw_load_means(const char *path, int w, int h, double *means_out,
                               uint64_t *total_samples) {
  try {
    ArrayOutput ao = ArrayOutput::load(path);
    if (ao.width() != w || ao.height() != h) return 2;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        Vec3 m = ao.rawPixelAt(x, y);
        size_t pix = static_cast<size_t>(x) + static_cast<size_t>(y) * w;
        means_out[pix * 3 + 0] = m.x(), means_out[pix * 3 + 1] = m.y(),
                            means_out[pix * 3 + 2] = m.z();
      }
    *total_samples = ao.totalSamples();
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}

// capi_host.cpp — the host half of the C ABI declared in include/ptw.h: scene building,
// camera, materials, the ArrayOutput file formats, error reporting.  No HIP in this file.
#include "capi_common.h"

#include "../host/framebuffer.h"
#include "../host/obj_loader.h"
#include "../host/precompute.h"
#include "../host/prefilter.h"
#include "../host/scenes.h"

#include <cstring>
#include <sstream>

namespace ptw {
namespace {
thread_local std::string g_lastError;
}

void setLastError(const std::string &message) { g_lastError = message; }

int translateException() {
  try {
    throw;
  } catch (const UnknownScene &e) {
    setLastError(e.what());
    return PTW_ERR_UNKNOWN_SCENE;
  } catch (const ParseError &e) {
    setLastError(e.what());
    return PTW_ERR_PARSE;
  } catch (const IoError &e) {
    setLastError(e.what());
    return PTW_ERR_IO;
  } catch (const SizeMismatch &e) {
    setLastError(e.what());
    return PTW_ERR_SIZE_MISMATCH;
  } catch (const DeviceError &e) {
    setLastError(e.what());
    return e.status;
  } catch (const std::bad_alloc &) {
    setLastError("out of host memory");
    return PTW_ERR_INVALID;
  } catch (const std::exception &e) {
    setLastError(e.what());
    return PTW_ERR_INVALID;
  } catch (...) {
    setLastError("unknown error");
    return PTW_ERR_INVALID;
  }
}

int invalid(const char *what) {
  setLastError(std::string("invalid argument: ") + what);
  return PTW_ERR_INVALID;
}
} // namespace ptw

using namespace ptw;

#define PTW_GUARD_BEGIN try {
#define PTW_GUARD_END                                                                          \
  }                                                                                            \
  catch (...) {                                                                                \
    return translateException();                                                               \
  }

extern "C" {

const char *ptw_last_error(void) { return g_lastError.c_str(); }
int ptw_abi_version(void) { return PTW_ABI_VERSION; }

void ptw_debug_defaults(ptw_debug_options *out) {
  if (!out) return;
  std::memset(out, 0, sizeof *out);
  out->seq_two_masters = out->seq_pairing = out->seq_lds_tables = out->seq_small_kernel = -1;
  out->fail_shard = out->fail_collective = out->silent_shard = -1;
  out->seq_unit_ufirst = -1;
}

void ptw_default_params(ptw_render_params *out) {
  if (!out) return;
  std::memset(out, 0, sizeof *out);
  out->width = 1920;
  out->height = 1080;
  out->preview = 0;
  out->samples_per_pixel = 40;
  out->max_depth = 5;
  out->first_bounce_u = 4;
  out->first_bounce_v = 4;
  out->seed = 0;
  out->rng_policy = PTW_RNG_SEQUENTIAL;
}

void ptw_default_material(ptw_material *out) {
  if (out) *out = material::defaults();
}
void ptw_material_diffuse(const double colour[3], ptw_material *out) {
  if (colour && out) *out = material::makeDiffuse(Vec3d(colour));
}
void ptw_material_specular(const double colour[3], double index, ptw_material *out) {
  if (colour && out) *out = material::makeSpecular(Vec3d(colour), index);
}
void ptw_material_light(const double colour[3], ptw_material *out) {
  if (colour && out) *out = material::makeLight(Vec3d(colour));
}
void ptw_material_glossy(const double colour[3], double index, double cone_degrees,
                         ptw_material *out) {
  if (colour && out) *out = material::makeGlossy(Vec3d(colour), index, cone_degrees);
}
void ptw_material_reflective(const double colour[3], double reflectivity, double cone_degrees,
                             ptw_material *out) {
  if (colour && out) *out = material::makeReflective(Vec3d(colour), reflectivity, cone_degrees);
}

int ptw_scene_create(ptw_scene **out) {
  if (!out) return invalid("out");
  PTW_GUARD_BEGIN
  *out = new ptw_scene();
  return PTW_OK;
  PTW_GUARD_END
}
void ptw_scene_destroy(ptw_scene *scene) { delete scene; }

int ptw_scene_add_triangle(ptw_scene *scene, const double v0[3], const double v1[3],
                           const double v2[3], const ptw_material *mat) {
  if (!scene || !v0 || !v1 || !v2 || !mat) return invalid("null pointer");
  PTW_GUARD_BEGIN
  scene->builder.addTriangle(Vec3d(v0), Vec3d(v1), Vec3d(v2), *mat);
  return PTW_OK;
  PTW_GUARD_END
}
int ptw_scene_add_sphere(ptw_scene *scene, const double centre[3], double radius,
                         const ptw_material *mat) {
  if (!scene || !centre || !mat) return invalid("null pointer");
  PTW_GUARD_BEGIN
  scene->builder.addSphere(Vec3d(centre), radius, *mat);
  return PTW_OK;
  PTW_GUARD_END
}
int ptw_scene_set_environment(ptw_scene *scene, const double colour[3]) {
  if (!scene || !colour) return invalid("null pointer");
  scene->builder.setEnvironmentColour(Vec3d(colour));
  return PTW_OK;
}

int ptw_scene_load_obj(ptw_scene *scene, const char *obj_path, const char *mtl_dir) {
  if (!scene || !obj_path || !mtl_dir) return invalid("null pointer");
  PTW_GUARD_BEGIN
  std::ifstream in(obj_path);
  if (!in) throw IoError(std::string("Unable to open ") + obj_path);
  loadObj(in, dirRelativeOpener(mtl_dir), scene->builder);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_scene_load_obj_text(ptw_scene *scene, const char *obj_text, const char *mtl_text) {
  if (!scene || !obj_text) return invalid("null pointer");
  PTW_GUARD_BEGIN
  std::istringstream in(obj_text);
  MtlOpener opener;
  if (mtl_text) {
    std::string mtl(mtl_text);
    opener = [mtl](const std::string &) -> std::unique_ptr<std::istream> {
      return std::make_unique<std::istringstream>(mtl);
    };
  }
  loadObj(in, opener, scene->builder);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_scene_build_named(ptw_scene *scene, const char *name, const char *scenes_dir,
                          int32_t width, int32_t height, ptw_camera *camera_out) {
  if (!scene || !name || !scenes_dir || !camera_out) return invalid("null pointer");
  if (width <= 0 || height <= 0) return invalid("width/height");
  PTW_GUARD_BEGIN
  *camera_out = buildNamedScene(scene->builder, name, scenes_dir, width, height);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_scene_view_of(const ptw_scene *scene, ptw_scene_view *out) {
  if (!scene || !out) return invalid("null pointer");
  *out = scene->builder.view();
  return PTW_OK;
}

int ptw_scene_prefilter_records(const ptw_scene *scene, float *out, uint64_t capacity_floats,
                                uint64_t *needed_floats, int32_t *usable) {
  if (!scene || (!out && capacity_floats)) return invalid("null pointer");
  PTW_GUARD_BEGIN
  const ptw_scene_view view = scene->builder.view();
  const ptw::DeviceSceneData data = ptw::precomputeScene(view); // (v0, e1, e2 exactly as the device gets them)
  const ptw::PrefilterData pre = ptw::buildPrefilter(data.triGeom.data(), view.num_triangles, view.sph_centre_radius, view.num_spheres);
  const uint64_t full = view.num_triangles ? pre.pairs.size() : 0;
  if (needed_floats) *needed_floats = full;
  if (usable) *usable = pre.usable ? 1 : 0;
  const uint64_t n = full < capacity_floats ? full : capacity_floats;
  for (uint64_t i = 0; i < n; ++i) out[i] = pre.pairs[i];
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_scene_unit_coherence(const ptw_scene *scene, double *out) {
  if (!scene || !out) return invalid("null pointer");
  PTW_GUARD_BEGIN
  const ptw_scene_view view = scene->builder.view();
  const ptw::DeviceSceneData data = ptw::precomputeScene(view);
  *out = ptw::unitUSkipFraction(data.triGeom.data(), view.num_triangles);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_camera_look_at(const double eye[3], const double look_at[3], const double up[3],
                       int32_t width, int32_t height, double vertical_fov_degrees,
                       ptw_camera *out) {
  if (!eye || !look_at || !up || !out) return invalid("null pointer");
  if (width <= 0 || height <= 0) return invalid("width/height");
  *out = makeCamera(Vec3d(eye), Vec3d(look_at), Vec3d(up), width, height, vertical_fov_degrees);
  return PTW_OK;
}
int ptw_camera_set_focus(ptw_camera *camera, const double focal_point[3],
                         double aperture_radius) {
  if (!camera || !focal_point) return invalid("null pointer");
  setFocus(*camera, Vec3d(focal_point), aperture_radius);
  return PTW_OK;
}

int ptw_raw_save(const char *path, int32_t width, int32_t height, const double *rgb_sum,
                 const uint32_t *counts) {
  if (!path || !rgb_sum || !counts) return invalid("null pointer");
  if (width < 0 || height < 0) return invalid("width/height");
  PTW_GUARD_BEGIN
  saveRaw(path, width, height, rgb_sum, counts);
  return PTW_OK;
  PTW_GUARD_END
}
int ptw_raw_read_header(const char *path, int32_t *width, int32_t *height) {
  if (!path || !width || !height) return invalid("null pointer");
  PTW_GUARD_BEGIN
  int w = 0, h = 0;
  readRawHeader(path, w, h);
  *width = w;
  *height = h;
  return PTW_OK;
  PTW_GUARD_END
}
int ptw_raw_load_accumulate(const char *path, int32_t width, int32_t height, double *rgb_sum,
                            uint32_t *counts) {
  if (!path || !rgb_sum || !counts) return invalid("null pointer");
  PTW_GUARD_BEGIN
  loadRawAccumulate(path, width, height, rgb_sum, counts);
  return PTW_OK;
  PTW_GUARD_END
}
int ptw_pixels_rgb8(int32_t width, int32_t height, const double *rgb_sum,
                    const uint32_t *counts, uint8_t *rgb8_out) {
  if (!rgb_sum || !counts || !rgb8_out) return invalid("null pointer");
  if (width < 0 || height < 0) return invalid("width/height");
  toRgb8(width, height, rgb_sum, counts, rgb8_out);
  return PTW_OK;
}
int ptw_png_save(const char *path, int32_t width, int32_t height, const uint8_t *rgb8) {
  if (!path || !rgb8) return invalid("null pointer");
  if (width <= 0 || height <= 0) return invalid("width/height");
  PTW_GUARD_BEGIN
  savePng(path, width, height, rgb8);
  return PTW_OK;
  PTW_GUARD_END
}
uint64_t ptw_total_samples(int32_t width, int32_t height, const uint32_t *counts) {
  if (!counts || width <= 0 || height <= 0) return 0;
  uint64_t total = 0;
  const size_t n = static_cast<size_t>(width) * height;
  for (size_t i = 0; i < n; ++i) total += counts[i];
  return total;
}

} // extern "C"

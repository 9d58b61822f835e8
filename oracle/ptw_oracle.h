/* ptw_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the reference's DoD radiance path, used only as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product (pt-three-ways_amd/, the C-ABI library, the CLI) may include, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks this restatement against the
 * reference's own compiled sources (oracle/_ref, built by oracle/Makefile from
 * /root/reference/src where they lie) and tests/test_oracle_golden.py checks it against the
 * committed vectors under tests/golden/ that oracle/make_golden.py generated from that build.
 *
 * The POD types of the boundary (scene view, camera, params) come from include/ptw.h.
 */
#ifndef PTW_ORACLE_H_
#define PTW_ORACLE_H_

#include "../include/ptw.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- std::mt19937 + std::uniform_real_distribution<double> (libstdc++ 11) ------------- */
typedef struct oracle_mt19937 {
  uint32_t x[624];
  uint32_t pos;
} oracle_mt19937;
void oracle_mt_seed(oracle_mt19937 *mt, uint32_t seed);
uint32_t oracle_mt_next(oracle_mt19937 *mt);
/* KAT helpers: first n raw words / first n canonical doubles of mt19937(seed). */
void oracle_mt_words(uint32_t seed, uint32_t n, uint32_t *out);
void oracle_mt_unit_doubles(uint32_t seed, uint32_t n, double *out);
/* PERPIXEL policy stream (sfc32 keyed by pass seed and pixel index): first n words. */
void oracle_perpixel_words(uint32_t pass_seed, uint32_t pixel_index, uint32_t n, uint32_t *out);

/* ---- Scene::intersect* (src/dod/Scene.cpp:13-122) ------------------------------------- */
/* hit_out[9] = distance (-1 for a miss), inside, position xyz, normal xyz, material index. */
void oracle_intersect(const ptw_scene_view *scene, const double ray[6], double hit_out[9]);
void oracle_intersect_spheres(const ptw_scene_view *scene, const double ray[6],
                              double nearer_than, double hit_out[9]);
void oracle_intersect_triangles(const ptw_scene_view *scene, const double ray[6],
                                double nearer_than, double hit_out[9]);

/* ---- Camera (src/math/Camera.h) ------------------------------------------------------- */
int oracle_camera_look_at(const double eye[3], const double look_at[3], const double up[3],
                          int32_t width, int32_t height, double vfov_degrees, ptw_camera *out);
void oracle_camera_set_focus(ptw_camera *cam, const double focal_point[3], double aperture);
/* Camera::randomRay for pixel (px,py) drawing from mt19937(seed) from its start; ray_out[6]. */
void oracle_camera_ray(const ptw_camera *cam, int32_t px, int32_t py, uint32_t seed,
                       double ray_out[6]);

/* ---- One pass of Scene::render's worker lambda (src/dod/Scene.cpp:209-219) -------------
 * radiance_out: width*height*3 doubles (this pass's per-pixel radiance, NOT accumulated);
 * words_out (may be NULL): width*height uint32, RNG words consumed per pixel.
 * Honours params->rng_policy, the row window and row_stride/row_phase (PERPIXEL only), first_pass. */
int oracle_render_pass(const ptw_scene_view *scene, const ptw_camera *camera,
                       const ptw_render_params *params, int32_t pass_index,
                       double *radiance_out, uint32_t *words_out);

/* The same with the per-pixel pick checksum of the pass (picks_out, width*height uint32, may be NULL):
 * sum over a sample's intersect() calls r = 0, 1, ... of (r + 1) * (combined primitive index + 1) mod
 * 2^32, a miss counting 0 (include/ptw.h, ptw_debug_options.d_picks). */
int oracle_render_pass_picks(const ptw_scene_view *scene, const ptw_camera *camera,
                             const ptw_render_params *params, int32_t pass_index,
                             double *radiance_out, uint32_t *words_out, uint32_t *picks_out);

/* All passes, merged in pass order into rgb_sum/counts (+=), `threads` worker threads
 * (one full-frame pass per thread at a time, as the reference).  words_out (may be NULL) is
 * [pass][y][x].  rays_out (may be NULL) receives the number of intersect() calls made. */
int oracle_render(const ptw_scene_view *scene, const ptw_camera *camera,
                  const ptw_render_params *params, int32_t threads, double *rgb_sum,
                  uint32_t *counts, uint32_t *words_out, uint64_t *rays_out);

/* ... and picks_out (may be NULL) [pass][y][x]. */
int oracle_render_picks(const ptw_scene_view *scene, const ptw_camera *camera,
                        const ptw_render_params *params, int32_t threads, double *rgb_sum,
                        uint32_t *counts, uint32_t *words_out, uint64_t *rays_out, uint32_t *picks_out);

/* ArrayOutput::pixelAt component conversion (src/util/ArrayOutput.cpp:9-12). */
uint8_t oracle_component_to_int(double x);

#ifdef __cplusplus
}
#endif
#endif

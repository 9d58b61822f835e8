/* ptw.h — C ABI of the MI355X-native "hip" way of pt-three-ways' DoD radiance path.
 *
 * This header is the drop-in boundary: plain C, plain pointers and sizes, no C++ or torch
 * types.  A host in any language binds these symbols (see INTEGRATION.md for the C++ stub a
 * pt-three-ways maintainer would add next to `oo`/`fp`/`dod` in src/main/main.cpp:350-366,
 * and for a ctypes stub).  Every entry point cites the reference interface it replaces;
 * paths are relative to the reference repository root.
 *
 * Conventions
 *  - All real arithmetic is IEEE binary64, as in the reference (src/math/Vec3.h:9).
 *  - Vectors are double[3] = {x, y, z}.
 *  - Every function returning `int` returns PTW_OK (0) or a ptw_status error code; the
 *    message for the calling thread's last error is available from ptw_last_error().
 *    Nothing throws across this boundary and nothing aborts the process.
 *  - The caller owns every buffer it passes in.  Objects created by *_create are released
 *    by the matching *_destroy.
 *  - There is no CPU fallback: rendering entry points fail with PTW_ERR_NO_DEVICE when no
 *    gfx950-class HIP device is usable.
 */
#ifndef PTW_H_
#define PTW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): PTW_ACCEL_PREFILTER and its known-answer entry ptw_scene_prefilter_records;
 * ptw_debug_options.seq_small_kernel = 3; seq_pairing / gang_groups retired (refused, layout kept). */
/* 7 (round 6, third session): ptw_scene_unit_coherence and ptw_debug_options.seq_unit_ufirst (the new field takes
 * the four bytes of padding before d_picks: the layout of v6 is unchanged). */
#define PTW_ABI_VERSION 7

typedef enum ptw_status {
  PTW_OK = 0,
  PTW_ERR_INVALID = 1,       /* bad argument (null pointer, non-positive size, ...)            */
  PTW_ERR_NO_DEVICE = 2,     /* no usable HIP device / extension built without one             */
  PTW_ERR_HIP = 3,           /* a HIP runtime call or kernel launch failed                     */
  PTW_ERR_IO = 4,            /* std::runtime_error("Unable to open ...") in the reference      */
  PTW_ERR_PARSE = 5,         /* OBJ/MTL parse error (message mirrors the reference's text)     */
  PTW_ERR_UNKNOWN_SCENE = 6, /* "Unknown scene <name>", src/main/main.cpp:308                  */
  PTW_ERR_SIZE_MISMATCH = 7, /* std::logic_error in ArrayOutput::operator+=, ArrayOutput.cpp:50 */
  PTW_ERR_UNSUPPORTED = 8
} ptw_status;

/* How random numbers are assigned to samples.
 *  SEQUENTIAL — the reference's policy, bit-compatible stream: one std::mt19937 per pass,
 *               seeded `seed + pass`, consumed by the pixels of that pass in row-major order
 *               with a data-dependent number of draws per pixel (src/dod/Scene.cpp:208-217).
 *               Results match the reference DoD renderer at matched seed.
 *  PERPIXEL   — one independent counter-seeded stream per (pass, pixel); pixels become
 *               independent, so image tiles can be sharded.  Not seed-matched with the
 *               reference (same estimator, different random numbers). */
typedef enum ptw_rng_policy { PTW_RNG_SEQUENTIAL = 0, PTW_RNG_PERPIXEL = 1 } ptw_rng_policy;

/* How Scene::intersect finds the nearest hit.
 *  NONE — the reference's algorithm: every ray tests every primitive (README.md:5-6,
 *         src/dod/Scene.cpp:13-113).  The default, and what every headline number is measured on.
 *  BVH  — a SEPARATE, separately reported mode: triangles a ray cannot hit are culled by a
 *         bounding-volume hierarchy and the same Moller-Trumbore arithmetic runs on the rest, so
 *         every sample is bit-identical to the NONE result while the work (tests per ray) is not
 *         the reference's.  PTW_RNG_PERPIXEL only.
 *  PREFILTER — the other separate mode ("fp32 intersect"): every ray still looks at
 *         every triangle (src/dod/Scene.cpp:62-98), but first in fp32 - two triangles per packed
 *         instruction - and the reference's fp64 test runs only where the fp32 evaluation, with a
 *         stated forward error bound, cannot PROVE that the fp64 test rejects.  Bit-identical samples
 *         (a plain fp32 test would flip decisions).  Under PTW_RNG_PERPIXEL both kernel forms; under
 *         PTW_RNG_SEQUENTIAL the worker-wave kernels (scenes beyond 128 triangles: the worker lanes hold
 *         their triangles in fp32 and fetch the fp64 data of the few survivors).  Refused with
 *         PTW_ERR_UNSUPPORTED for scenes with coordinates beyond 1e12 (fp32 products could overflow). */
typedef enum ptw_accel { PTW_ACCEL_NONE = 0, PTW_ACCEL_BVH = 1, PTW_ACCEL_PREFILTER = 2 } ptw_accel;

/* Which of the PTW_RNG_PERPIXEL policy's two radiance kernels runs (same samples, same bytes;
 * which one is faster depends on how uniformly long the scene's paths are, which no host-side
 * number says - closed scenes favour LOCKSTEP, open ones PERSISTENT):
 *  AUTO        the kernel ptw_context_calibrate() last measured faster for this scene, camera and
 *              frame shape on the context; PERSISTENT when nothing was calibrated;
 *  LOCKSTEP    a lane traces whole samples, the lanes of a wave run each level's shading together;
 *  PERSISTENT  a lane whose path ends takes the next sample from a device-wide queue.
 * Ignored under PTW_RNG_SEQUENTIAL and in the accelerated mode. */
typedef enum ptw_pix_kernel {
  PTW_PIX_KERNEL_AUTO = 0, PTW_PIX_KERNEL_LOCKSTEP = 1, PTW_PIX_KERNEL_PERSISTENT = 2
} ptw_pix_kernel;

/* MaterialSpec, src/util/MaterialSpec.h:7-12 (same field order, 72 bytes). */
typedef struct ptw_material {
  double emission[3];
  double diffuse[3];
  double index_of_refraction;      /* default 1.0 */
  double reflectivity;             /* default -1 => Fresnel-ish reflectance, Norm3.cpp:7-24 */
  double reflection_cone_angle_rad; /* default 0.0 */
} ptw_material;

/* What dod::Scene holds after the SceneBuilder calls (src/dod/Scene.h:22-31), flattened.
 * Primitive order is insertion order; it is the nearest-hit tie-break order. */
typedef struct ptw_scene_view {
  uint32_t num_triangles;
  uint32_t num_spheres;
  uint32_t num_materials;
  uint32_t reserved;
  const double *tri_vertices;      /* [num_triangles][3 vertices][3]                        */
  const uint32_t *tri_material;    /* [num_triangles] index into materials                  */
  const double *sph_centre_radius; /* [num_spheres][4] = centre xyz, radius (not squared)   */
  const uint32_t *sph_material;    /* [num_spheres]                                         */
  const ptw_material *materials;   /* [num_materials]                                       */
  double environment[3];           /* Scene::environment_, default (0,0,0)                  */
} ptw_scene_view;

/* Everything Camera keeps private (src/math/Camera.h:11-18). */
typedef struct ptw_camera {
  double centre[3];
  double axis_x[3], axis_y[3], axis_z[3]; /* OrthoNormalBasis::fromZY(dir, up)              */
  double aspect_ratio;                    /* width / height                                 */
  double camera_plane_dist;               /* 1 / tan(vfov * pi / 360)                       */
  double reciprocal_height;
  double reciprocal_width;
  double aperture_radius;                 /* 0 => pinhole: 2 draws per primary ray, else 4  */
  double focal_distance;
} ptw_camera;

/* RenderParams (src/util/RenderParams.h:3-13) plus what the hip way adds. */
typedef struct ptw_render_params {
  int32_t width;                 /* default 1920 */
  int32_t height;                /* default 1080 */
  int32_t preview;               /* default 0    */
  int32_t samples_per_pixel;     /* default 40; the number of passes rendered AND kept      */
  int32_t max_depth;             /* default 5    */
  int32_t first_bounce_u;        /* default 4    */
  int32_t first_bounce_v;        /* default 4    */
  int32_t seed;                  /* pass k uses mt19937(uint32(seed + first_pass + k))      */
  int32_t first_pass;            /* default 0: index of the first pass (multi-GPU / resume) */
  int32_t rng_policy;            /* ptw_rng_policy                                          */
  /* Pixel window for PERPIXEL tile sharding: of the rows [row_begin, row_end) those with
   * (y % row_stride) == row_phase are rendered; the output buffers always describe the full
   * width x height frame.  row_begin == row_end == 0 => all rows; row_begin == row_end != 0 =>
   * an empty shard (nothing is rendered); row_stride 0 or 1 => every row of the window.
   * Interleaved rows (stride = number of GPUs, phase = rank) balance the shards: what a pixel
   * costs depends on what it sees.  Under PTW_RNG_SEQUENTIAL the pixels of a pass are serially
   * dependent (shard by first_pass): a stride, or a window that does not start at row 0, is
   * PTW_ERR_UNSUPPORTED; a PREFIX [0, row_end) is allowed - it is exactly what the full render
   * produces for those rows (timed sub-runs of very large frames). */
  int32_t row_begin;
  int32_t row_end;
  int32_t device;                /* HIP device ordinal for ptw_render()                     */
  int32_t row_stride;
  int32_t row_phase;
  int32_t accel;                 /* ptw_accel; default PTW_ACCEL_NONE                       */
  int32_t pix_kernel;            /* ptw_pix_kernel; default PTW_PIX_KERNEL_AUTO             */
} ptw_render_params;

/* Progress callback, the analogue of `updateFunc(output)` (src/dod/Scene.cpp:245) and of
 * Progressifier (src/util/Progressifier.cpp:11-21): called between device launches with the
 * number of finished samples - on the calling thread for single-device renders; with
 * ptw_render_options.num_devices > 1 from the host thread that drives the first device, while the
 * calling thread waits for it (never concurrently with itself).  Return non-zero to cancel: the
 * render ends with PTW_ERR_INVALID ("cancelled by the callback"), every shard stopping at its next
 * band and none entering the collective. */
typedef int (*ptw_progress_fn)(void *user, uint64_t samples_done, uint64_t samples_total);
/* The full form of `updateFunc(output)` (src/dod/Scene.cpp:245, used by src/main/main.cpp:331-343
 * for --save-every): called on the calling thread between device launches with the caller's
 * own rgb_sum / counts buffers brought up to date - every pixel finished so far holds all of its
 * samples, the others hold what they held before the call (the hip way completes the frame
 * band by band, top to bottom, with every pass of a band in one launch; the reference completes
 * it pass by pass).  The buffers are a valid ArrayOutput at every call (sums + counts), so a
 * snapshot can be saved, merged and resumed like the reference's.  Return non-zero to cancel. */
typedef int (*ptw_update_fn)(void *user, uint64_t samples_done, uint64_t samples_total,
                             const double *rgb_sum, const uint32_t *counts);

const char *ptw_last_error(void);
int ptw_abi_version(void);
void ptw_default_params(ptw_render_params *out);           /* RenderParams.h:3-13 defaults */
void ptw_default_material(ptw_material *out);              /* MaterialSpec.h:8-12 defaults */

/* ---- MaterialSpec factories, src/util/MaterialSpec.h:13-32 ------------------------------ */
void ptw_material_diffuse(const double colour[3], ptw_material *out);
void ptw_material_specular(const double colour[3], double index, ptw_material *out);
void ptw_material_light(const double colour[3], ptw_material *out);
void ptw_material_glossy(const double colour[3], double index, double cone_degrees,
                         ptw_material *out);
void ptw_material_reflective(const double colour[3], double reflectivity, double cone_degrees,
                             ptw_material *out);

/* ---- SceneBuilder concept, src/dod/Scene.h:37-42 and src/dod/Scene.cpp:181-195 ---------- */
typedef struct ptw_scene ptw_scene;
int ptw_scene_create(ptw_scene **out);
void ptw_scene_destroy(ptw_scene *scene);
int ptw_scene_add_triangle(ptw_scene *scene, const double v0[3], const double v1[3],
                           const double v2[3], const ptw_material *material);
int ptw_scene_add_sphere(ptw_scene *scene, const double centre[3], double radius,
                         const ptw_material *material);
int ptw_scene_set_environment(ptw_scene *scene, const double colour[3]);
/* loadObjFile + DirRelativeOpener, src/util/ObjLoaderImpl.h:55-103, src/main/main.cpp:27-38.
 * `obj_path` is opened directly; `mtllib` names are resolved relative to `mtl_dir`. */
int ptw_scene_load_obj(ptw_scene *scene, const char *obj_path, const char *mtl_dir);
/* Same parser fed from memory (the reference's tests feed strings, ObjLoaderTests.cpp:29-35);
 * `mtl_text` may be NULL, in which case any `mtllib` directive fails like the reference's
 * ThrowingObjLoaderOpener. */
int ptw_scene_load_obj_text(ptw_scene *scene, const char *obj_text, const char *mtl_text);
/* createScene(sb, name, params), src/main/main.cpp:291-309: the built-in scenes "cornell",
 * "suzanne", "ce", "single-sphere", "multi-sphere", "example1", "bbc-owl".  OBJ-backed scenes
 * read `<scenes_dir>/<file>` (the reference hard-codes "scenes"). */
int ptw_scene_build_named(ptw_scene *scene, const char *name, const char *scenes_dir,
                          int32_t width, int32_t height, ptw_camera *camera_out);
/* Borrowed view, valid until the scene is modified or destroyed. */
int ptw_scene_view_of(const ptw_scene *scene, ptw_scene_view *out);
/* Known-answer hook for PTW_ACCEL_PREFILTER (host only, no device): the fp32 records the prefilter kernel
 * reads - one per PAIR of triangles (2k, 2k + 1), 22 floats: v0 / e1 / e2 component-interleaved (x of A, x of
 * B, y of A, ...), then the error-bound coefficients EA(A, B), EB(A, B) with E = EA + |ray origin|_inf * EB
 * (the geometry rounded to nearest, the coefficients rounded up).  Writes min(capacity_floats, 22 *
 * ceil(ntri / 2)) floats, *needed_floats = the full size, *usable = 0 when the mode refuses the scene
 * (a coordinate that is not finite or beyond 1e12).  The tests check the conservativeness of the
 * criterion with these very records (src/dod/Scene.cpp:62-98 is what must never be contradicted). */
int ptw_scene_prefilter_records(const ptw_scene *scene, float *out, uint64_t capacity_floats,
                                uint64_t *needed_floats, int32_t *usable);

/* The statistic behind ptw_debug_options.seq_unit_ufirst (host only, no device): the fraction of (ray, unit of 64
 * consecutive triangles) pairs, over a fixed pseudo-random sample of rays starting on the scene's triangles, in
 * which NO triangle of the unit passes the u test of src/dod/Scene.cpp:79-89.  Meshes whose faces follow each other
 * in space score high (ce 0.7), random soups 0.  Decides a schedule, never a result. */
int ptw_scene_unit_coherence(const ptw_scene *scene, double *out);

/* ---- Camera ctor / setFocus, src/math/Camera.h:40-51 ------------------------------------ */
int ptw_camera_look_at(const double eye[3], const double look_at[3], const double up[3],
                       int32_t width, int32_t height, double vertical_fov_degrees,
                       ptw_camera *out);
int ptw_camera_set_focus(ptw_camera *camera, const double focal_point[3],
                         double aperture_radius);

/* ---- dod::Scene::render, src/dod/Scene.h:44-46 / src/dod/Scene.cpp:197-254 ---------------
 * Renders `samples_per_pixel` passes on HIP device `params->device` and ADDS them into the
 * caller's ArrayOutput-shaped buffers: rgb_sum[(x + y*width)*3 + c] += sample radiance,
 * counts[x + y*width] += 1 per pass (ArrayOutput::addSamples / operator+=,
 * src/util/ArrayOutput.cpp:39-56).  Unlike the reference's scheduler (Scene.cpp:251) no
 * launched pass is dropped: exactly samples_per_pixel passes are accumulated, in pass order. */
int ptw_render(const ptw_scene_view *scene, const ptw_camera *camera,
               const ptw_render_params *params, double *rgb_sum, uint32_t *counts,
               ptw_progress_fn progress, void *user);

/* ---- Tests and A/B measurements ONLY: the library's dispatch forced from outside ----------
 * Which kernel instantiation a render runs is the dispatcher's decision (scene size, pass count,
 * LDS budget).  The test-suite has to reach every instantiation of the shipped binary with small
 * scenes, and a measurement sometimes wants the road not taken: this struct says so explicitly,
 * per context (ptw_context_set_debug) or per call (ptw_render_options.debug) - the process
 * environment of a host that loads this library is not read for any of it.  A production host
 * never sets it.  ptw_debug_defaults() fills in "the dispatcher decides" for every field. */
typedef struct ptw_debug_options {
  int32_t seq_two_masters;      /* worker-wave kernels (scenes beyond 128 triangles), two passes per
                                   workgroup: -1 the dispatcher's rule (more passes than CUs), 0 never,
                                   1 always                                                           */
  int32_t seq_pairing;          /* RETIRED (kept for the layout of ABI v5): 1 asked for round 5's paired form
                                   of the two-master kernels, which left the tree in round 6 (LAB.md) -
                                   PTW_ERR_UNSUPPORTED now; -1 / 0 = off                               */
  int32_t seq_lds_tables;       /* shading tables: -1 in LDS when they fit, 0 in global memory        */
  int32_t seq_small_kernel;     /* scenes of at most 64 triangles: -1 the dispatcher's rule, 0 the
                                   plain single-wave kernel (LDS tables, LDS stack), 1 the register
                                   variant, 2 the speculative four-wave kernel whatever the pass count,
                                   3 that kernel in its round-5 form (no camera ray of the next pixel
                                   traced ahead in a pixel's last round) - the A/B switch of round 6 -,
                                   4 its two-wave form (frontier + one candidate, two workgroups per CU) */
  int32_t seq_units[3];         /* worker-wave kernels: resident units of 64 triangles of an older /
                                   younger / master-side worker wave; {0, 0, 0} = the library's split */
  int32_t pix_samples_per_lane; /* lock-step PERPIXEL kernel's grid-stride depth; 0 = default (8)     */
  int32_t pix_waves_per_simd;   /* persistent PERPIXEL kernel: 0 = default (4), 2, 3 or 4             */
  int32_t gang_groups;          /* RETIRED like seq_pairing: CUs per pass of round 3's traceSequentialGang;
                                   > 0 is PTW_ERR_UNSUPPORTED now, 0 = off                            */
  int32_t fail_shard;           /* ptw_render_ex(num_devices > 1) failure injection, -1 = none:       */
  int32_t fail_collective;      /*   shard that fails its set-up / its collective call / reports      */
  int32_t silent_shard;         /*   success WITHOUT entering the collective (the watchdog ends it)   */
  int32_t trace;                /* 1: ptw_context_calibrate prints its two timings to stderr          */
  int32_t intersect_accel;      /* ptw_context_intersect (the known-answer entry): 0 = the brute-force
                                   search, PTW_ACCEL_PREFILTER = through the fp32 prefilter - same hits */
  int32_t seq_unit_ufirst;      /* worker-wave kernels: the unit-level u-first early-out of the worker waves
                                   (a unit of 64 consecutive triangles none of which passes the u test of
                                   src/dod/Scene.cpp:79-89 skips the rest of the test: same decisions, same
                                   values): -1 the library's rule (ptw_scene_unit_coherence >= 0.4), 0 off, 1 on */
  /* Pick checksum (parity instrumentation, PTW_RNG_SEQUENTIAL only): a DEVICE pointer to
   * [pass][y][x] uint32 receiving, per sample, sum over the sample's intersect() calls r = 0, 1, ...
   * in the reference's call order of (r + 1) * (combined primitive index + 1) mod 2^32, a miss
   * counting 0 - combined index = position in Scene::intersect's scan order: spheres [0, nsph),
   * then triangles nsph + k (src/dod/Scene.cpp:115-122).  With it a comparison can tell WHICH
   * primitive every ray hit, where radiance and RNG word counts cannot (ce: every path ends on an
   * emitter of diffuse 0 and every ray hits something).  oracle/ptw_oracle.c computes the same
   * number.  NULL = off; PTW_ERR_UNSUPPORTED under PTW_RNG_PERPIXEL.                                */
  void *d_picks;
} ptw_debug_options;
void ptw_debug_defaults(ptw_debug_options *out);

/* The same call with everything the reference's driver does around it (src/main/main.cpp:
 * 326-366): ONE context and one scene upload for the whole render, `update` handed the running
 * framebuffer (see ptw_update_fn), and - with num_devices > 1 - the reference's decomposition
 * over workers (one task per pass, src/dod/Scene.cpp:208-246) spread over the GPUs of the node:
 *   PTW_RNG_SEQUENTIAL  device g renders a contiguous range of the passes (full frames),
 *                       merged by ONE RCCL reduce(sum) of the fp64 sums + u32 counts
 *                       (ArrayOutput::operator+=, src/util/ArrayOutput.cpp:48-56);
 *   PTW_RNG_PERPIXEL    device g renders the image rows y with y % num_devices == g (all
 *                       passes), assembled by ONE RCCL gather of the rows to the first device.
 * One host thread per device; the collective runs over xGMI on the devices' own streams; the
 * frame crosses PCIe once, from the first device.  Zero-initialise the struct for defaults. */
typedef struct ptw_render_options {
  int32_t num_devices;         /* 0 or 1: params->device only                                  */
  int32_t min_updates;         /* with `update`: cut the frame into at least this many bands
                                  (0 => 16); more bands = more frequent snapshots              */
  const int32_t *devices;      /* [num_devices] HIP ordinals; NULL => device, device + 1, ...  */
  ptw_progress_fn progress;    /* may be NULL                                                  */
  void *progress_user;
  ptw_update_fn update;        /* may be NULL; single-device renders only                      */
  void *update_user;
  int32_t share_device;        /* hosts with ONE GPU (and the tests): every shard on params->device.
                                  1: the shards one after another, accumulated on the device, no
                                     collective;
                                  2: the N-GPU code path itself - a host thread, context and stream
                                     per shard, the collective through the in-process loopback
                                     transport (ptw_comm_create_loopback)                        */
  int32_t reserved;
  const ptw_debug_options *debug; /* tests / A-B runs only (see ptw_debug_options); NULL = none */
} ptw_render_options;
int ptw_render_ex(const ptw_scene_view *scene, const ptw_camera *camera,
                  const ptw_render_params *params, double *rgb_sum, uint32_t *counts,
                  const ptw_render_options *options);

/* ---- Device-resident form of the same call (HBM in, HBM out) ---------------------------- */
typedef struct ptw_context ptw_context;
int ptw_context_create(int32_t device, ptw_context **out);
void ptw_context_destroy(ptw_context *ctx);
/* Upload + per-primitive precompute (the work of addTriangle/addSphere, Scene.cpp:181-195). */
int ptw_context_set_scene(ptw_context *ctx, const ptw_scene_view *scene);
/* Enqueue the render on `hip_stream` (a hipStream_t, NULL = default stream).  d_rgb_sum and
 * d_counts are DEVICE pointers to width*height*3 doubles / width*height uint32 and are
 * accumulated into.  Asynchronous: nothing here waits for the device; the caller synchronises
 * the stream.  A context owns one set of scratch buffers (generator states, staging): renders
 * of one context must be enqueued on the SAME stream (they then run one after another in stream
 * order); to change streams, or before ptw_context_set_scene, synchronise the stream of the
 * previous render.  Renders on different contexts are independent.  If d_words is not NULL it
 * receives, per pass and pixel ([pass][y][x], uint32), the number of 32-bit RNG words that sample
 * consumed (parity instrumentation; SEQUENTIAL and PERPIXEL). */
int ptw_context_render(ptw_context *ctx, const ptw_camera *camera,
                       const ptw_render_params *params, void *d_rgb_sum, void *d_counts,
                       void *d_words, void *hip_stream);
/* PTW_RNG_PERPIXEL: times a trial of the policy's two kernels (about two million samples each,
 * image rows spread over the frame `params` describes, outputs into the context's staging buffer
 * only) on `hip_stream`, WAITS for it (10-30 ms), remembers the faster one on the context for this
 * scene + camera + frame shape - what PTW_PIX_KERNEL_AUTO then resolves to - and returns it in
 * *kernel_out (a ptw_pix_kernel; may be NULL).  Multi-GPU hosts calibrate on ONE rank and hand the
 * answer to the others in ptw_render_params.pix_kernel, so that every shard of a frame runs the
 * same kernel.  ptw_render / ptw_render_ex (synchronous calls) do this themselves for renders of
 * 16 M samples or more.  Under PTW_RNG_SEQUENTIAL / the accelerated mode: PTW_PIX_KERNEL_AUTO, no
 * trial. */
/* (Round 6: under PTW_RNG_SEQUENTIAL the same call times the two small-scene kernels - one wave per pass
 * against four speculating waves against two - when the scene has at most 64 triangles and the launch between one and six
 * passes per compute unit, where which one wins depends on the scene; the context remembers the winner for this
 * scene, camera, frame shape and pass count, *kernel_out is PTW_PIX_KERNEL_AUTO.  A no-op for every other launch.) */
int ptw_context_calibrate(ptw_context *ctx, const ptw_camera *camera,
                          const ptw_render_params *params, void *hip_stream, int32_t *kernel_out);
/* Tests / A-B runs only: the context's renders from now on follow `options` (copied; NULL restores
 * the defaults).  See ptw_debug_options. */
int ptw_context_set_debug(ptw_context *ctx, const ptw_debug_options *options);
/* Per-kernel timing gathered with hipEvents on the launch stream when enabled. */
typedef struct ptw_kernel_stats {
  uint64_t trace_launches;   /* launches of the radiance kernel since the last reset   */
  double trace_ms;           /* summed device time of those launches                   */
  uint64_t resolve_launches; /* launches of the pass-ordered accumulate kernel         */
  double resolve_ms;
  uint64_t samples;          /* samples traced by those launches                       */
  uint64_t rays;             /* intersect() calls made by those launches (0 if unknown) */
  char trace_kernel[64];     /* the radiance kernel variant the library last launched,
                                e.g. "traceSequentialSpec" (as the profiler names it)   */
} ptw_kernel_stats;
int ptw_context_enable_stats(ptw_context *ctx, int32_t enable);
/* Synchronises the events it reads. */
int ptw_context_get_stats(ptw_context *ctx, ptw_kernel_stats *out, int32_t reset);

/* Which kernel would run?  The dispatch rules of the library (csrc/dispatch.hip and the family launchers) for
 * a scene of the given size and a launch of `samples_per_pixel` passes, WITHOUT a device or a launch: writes
 * the kernel's name - the string ptw_kernel_stats.trace_kernel reports after a real render - into `out`.
 * `debug` may be NULL (the dispatcher decides).  scripts/dispatch_sweep.py and the CPU tests use it.  One decision
 * needs the scene itself and is not in the plan: the `,unit` form of the worker-wave kernels (ptw_scene_unit_coherence
 * >= 0.4) - `debug->seq_unit_ufirst = 1` names it. */
typedef struct ptw_dispatch_query {
  uint32_t num_triangles, num_spheres, num_materials;
  int32_t max_depth;         /* RenderParams::maxDepth (5)                                        */
  int32_t samples_per_pixel; /* passes of the launch                                              */
  int32_t rng_policy;        /* ptw_rng_policy                                                    */
  int32_t accel;             /* ptw_accel                                                         */
  int32_t pix_kernel;        /* ptw_pix_kernel (AUTO = the uncalibrated default)                  */
  int32_t compute_units;     /* 0: the current device's count (256 when there is no device)       */
  int32_t reserved[3];
} ptw_dispatch_query;
int ptw_dispatch_plan(const ptw_dispatch_query *query, const struct ptw_debug_options *debug, char *out,
                      size_t capacity);

/* Batch form of Scene::intersect (src/dod/Scene.cpp:115-122) for known-answer tests:
 * rays[n][6] = origin, direction (direction already normalised) -> hit[n]: distance (or -1
 * for a miss), inside flag, position, normal, material index (as doubles: 9 per ray). */
int ptw_context_intersect(ptw_context *ctx, const double *rays, uint64_t n, double *hits_out);

/* Known-answer hook for the device RNG: the first n values a
 * std::uniform_real_distribution<double>(0,1) draws from std::mt19937(seed) (SEQUENTIAL,
 * src/dod/Scene.cpp:149,211) or from the PERPIXEL stream of (pass_seed = seed, pixel), produced
 * by the same device code the render kernels use. */
int ptw_context_rng_doubles(ptw_context *ctx, int32_t rng_policy, uint32_t seed, uint32_t pixel,
                            uint32_t n, double *out);

/* ---- Multi-GPU: the framebuffer collectives (RCCL over xGMI) ----------------------------
 * One communicator per participating GPU.  Multi-process hosts (one process per GPU): rank 0
 * calls ptw_comm_unique_id, ships the 128 bytes to the other ranks by whatever means it has
 * (torch.distributed, MPI, a file), every rank calls ptw_comm_create.  Single-process hosts:
 * ptw_comm_create_all.  The collectives are enqueued on `hip_stream` and are asynchronous. */
#define PTW_COMM_ID_BYTES 128
typedef struct ptw_comm ptw_comm;
int ptw_comm_unique_id(uint8_t id_out[PTW_COMM_ID_BYTES]);
/* (Both set-up calls give up with PTW_ERR_HIP after PTW_COLLECTIVE_TIMEOUT_S - default 300 s - when a
 * rank of the world never arrives; ncclCommInitRank would otherwise wait for it for ever.  The thread that
 * made the RCCL call is then ABANDONED inside RCCL's bootstrap, with its sockets and its device context, for
 * the rest of the process: treat the process as poisoned for multi-GPU work after this error - report it and
 * restart rather than retry.  The library counts such threads (ptw_comm_describe: `abandoned_setups`) and
 * refuses a fifth set-up attempt outright.) */
int ptw_comm_create(const uint8_t id[PTW_COMM_ID_BYTES], int32_t world_size, int32_t rank,
                    int32_t device, ptw_comm **out);
int ptw_comm_create_all(int32_t num_devices, const int32_t *devices, ptw_comm **out_comms);
/* The same collectives between `world_size` communicators that share ONE device inside one
 * process (RCCL refuses two ranks on a GPU): device-to-device copies and an accumulate kernel,
 * ordered across the ranks' streams by HIP events.  Call the collectives from one host thread per
 * rank (they rendezvous).  For single-GPU hosts and for testing the sharded render end to end. */
int ptw_comm_create_loopback(int32_t world_size, int32_t device, ptw_comm **out_comms);
/* Gives up on a communicator whose peer will never arrive (a rank failed before its collective):
 * releases every rank that waits in a collective of it - ncclCommAbort / wakes the loopback
 * rendezvous - after which the only valid call is ptw_comm_destroy. */
int ptw_comm_abort(ptw_comm *comm);
void ptw_comm_destroy(ptw_comm *comm);
/* Waits until everything enqueued on `hip_stream` - the collectives above included - has finished,
 * WATCHING the communicator while it waits: an asynchronous RCCL error (ncclCommGetAsyncError: a
 * peer died, a link failed) or `timeout_ms` without completion (0: PTW_COLLECTIVE_TIMEOUT_S from the
 * environment, default 300 s) aborts the communicator - which ends its kernels on the device - and
 * returns PTW_ERR_HIP.  The collectives are asynchronous and a peer that disappears after the
 * enqueue would otherwise leave hipStreamSynchronize waiting forever. */
int ptw_comm_wait(ptw_comm *comm, void *hip_stream, int32_t timeout_ms);
/* Which wire the communicator's collectives use, as a JSON object in `out` (NUL-terminated; at most
 * `capacity` bytes - 4096 are plenty): {"kind": "rccl" | "loopback", "world", "rank", "device",
 * "links": [{"device", "type": "xgmi" | "pcie" | ..., "hops", "peer_access"}...] (HIP's report for this
 * rank's GPU against every other visible one), "p2p_disabled", "shm_disabled" (NCCL_P2P_DISABLE /
 * NCCL_SHM_DISABLE), "expected": the transport RCCL should therefore pick ("P2P/xGMI", "SHM",
 * "NET/Socket" ...), "rccl_log": the transports RCCL itself names in its channel lines ("P2P/IPC",
 * "NET/Socket/0" ...) when its log goes to a file (NCCL_DEBUG=INFO with NCCL_DEBUG_FILE; null otherwise;
 * channels are connected at the first collective, so ask after one)}.  So that the first run on an
 * 8-GPU node can say whether the framebuffer gather crossed xGMI. */
int ptw_comm_describe(ptw_comm *comm, char *out, size_t capacity);
/* `output += pass` for whole framebuffers (ArrayOutput::operator+=, ArrayOutput.cpp:48-56):
 * the fp64 sums (npix * 3) and the u32 counts (npix) of every rank are summed into rank `root`'s
 * buffers (ncclReduce; the other ranks' buffers are left as they are).  In place. */
int ptw_comm_reduce_framebuffer(ptw_comm *comm, void *d_rgb_sum, void *d_counts, uint64_t npix,
                                int32_t root, void *hip_stream);
/* Assembles a frame rendered with interleaved rows (row_stride = world_size, row_phase = rank):
 * every rank sends the rows it owns, rank `root` receives them into the same rows of its own
 * buffers (rows of other ranks are overwritten there, its own are kept).  A gather of 1/world of
 * the frame per rank - point-to-point over the xGMI links into `root`. */
int ptw_comm_gather_rows(ptw_comm *comm, void *d_rgb_sum, void *d_counts, int32_t width,
                         int32_t height, int32_t root, void *hip_stream);

/* ---- ArrayOutput surface, src/util/ArrayOutput.cpp ------------------------------------- */
/* .raw: header {u32 signature=1, version=1, height, width} then per pixel 3 x f64 sum +
 * u32 count (ArrayOutput.cpp:21-28,65-81). */
int ptw_raw_save(const char *path, int32_t width, int32_t height, const double *rgb_sum,
                 const uint32_t *counts);
int ptw_raw_read_header(const char *path, int32_t *width, int32_t *height);
/* Adds the file's sums/counts into the buffers (ArrayOutput::load + operator+=,
 * ArrayOutput.cpp:83-110, raw_to_png.cpp:39-58). */
int ptw_raw_load_accumulate(const char *path, int32_t width, int32_t height, double *rgb_sum,
                            uint32_t *counts);
/* ArrayOutput::pixelAt for every pixel: lround(pow(clamp(mean,0,1), 1/2.2) * 255)
 * (ArrayOutput.cpp:9-12,30-35); rgb8_out is width*height*3 bytes. */
int ptw_pixels_rgb8(int32_t width, int32_t height, const double *rgb_sum,
                    const uint32_t *counts, uint8_t *rgb8_out);
/* PngWriter (src/main/PngWriter.cpp): 8-bit RGB, non-interlaced PNG. */
int ptw_png_save(const char *path, int32_t width, int32_t height, const uint8_t *rgb8);
/* ArrayOutput::totalSamples, ArrayOutput.cpp:58-63. */
uint64_t ptw_total_samples(int32_t width, int32_t height, const uint32_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* PTW_H_ */

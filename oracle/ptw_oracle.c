/* ptw_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See ptw_oracle.h.
 *
 * A from-scratch CPU restatement of pt-three-ways' DoD radiance path in plain C.  Every
 * function cites the reference file:line whose arithmetic (including operation ORDER, which
 * fixes the rounding) it follows.  Compile with -ffp-contract=off for the checker build so
 * that a*b+c is two roundings, as in a strict build of the reference.
 *
 * Third-party arithmetic restated here because it is not in /root/reference:
 *  - libstdc++ (GCC 11.4) <random>: std::mt19937 (fixed by the C++ standard) and the
 *    implementation-defined word->double mapping of std::generate_canonical<double,53>
 *    (bits/random.tcc:3348-3380) used by std::uniform_real_distribution<double>
 *    (bits/random.h operator(): canonical * (b - a) + a).
 *  - glibc libm sqrt/sin/cos/acos/tan/pow are CALLED, not restated.
 */
#include "ptw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_EPSILON 0.000000001 /* src/math/Epsilon.h:3 */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------- */
/* Vec3 / Norm3 value math (src/math/Vec3.h, Norm3.impl.h)                                */
/* ------------------------------------------------------------------------------------- */
typedef struct v3 {
  double x, y, z;
} v3;

static inline v3 v3_make(double x, double y, double z) {
  v3 r = {x, y, z};
  return r;
}
static inline v3 v3_from(const double p[3]) { return v3_make(p[0], p[1], p[2]); }
/* Vec3::operator+ Vec3.h:16-18 */
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
/* Vec3::operator- Vec3.h:26-28 */
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
/* Vec3::operator*(double) Vec3.h:39-41 (also Norm3::operator*(double) Norm3.impl.h:13-15) */
static inline v3 v3_scale(v3 a, double b) { return v3_make(a.x * b, a.y * b, a.z * b); }
/* operator*(double, Vec3) Vec3.h:36-38 */
static inline v3 v3_lscale(double a, v3 b) { return v3_make(a * b.x, a * b.y, a * b.z); }
/* Vec3::operator*(Vec3) Vec3.h:61-63 */
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
/* Vec3::operator/(double): multiplies by the reciprocal, Vec3.h:51-54 */
static inline v3 v3_div(v3 a, double b) {
  const double reciprocal = 1.0 / b;
  return v3_make(a.x * reciprocal, a.y * reciprocal, a.z * reciprocal);
}
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* Vec3::dot Vec3.h:81-83 */
static inline double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* Vec3::cross Vec3.h:85-90 / Norm3::cross Norm3.impl.h:21-26 */
static inline v3 v3_cross(v3 a, v3 b) {
  double x = a.y * b.z - a.z * b.y;
  double y = a.z * b.x - a.x * b.z;
  double z = a.x * b.y - a.y * b.x;
  return v3_make(x, y, z);
}
/* Vec3::length Vec3.h:77 ; Vec3::normalised Vec3.impl.h:5-7 */
static inline double v3_length(v3 a) { return sqrt(v3_dot(a, a)); }
static inline v3 v3_normalised(v3 a) { return v3_div(a, v3_length(a)); }

/* Norm3::reflect Norm3.impl.h:41-44: incoming - (n * 2) * n.dot(incoming) */
static inline v3 n3_reflect(v3 n, v3 incoming) {
  return v3_sub(incoming, v3_scale(v3_scale(n, 2), v3_dot(n, incoming)));
}

/* Norm3::reflectance src/math/Norm3.cpp:7-24 (rParallel deliberately == rPerpendicular) */
static double n3_reflectance(v3 n, v3 incoming, double iorFrom, double iorTo) {
  double iorRatio = iorFrom / iorTo;
  double cosThetaI = -v3_dot(n, incoming);
  double sinThetaTSquared = iorRatio * iorRatio * (1 - cosThetaI * cosThetaI);
  if (sinThetaTSquared > 1) return 1.0;
  double cosThetaT = sqrt(1 - sinThetaTSquared);
  double rPerpendicular =
      (iorFrom * cosThetaI - iorTo * cosThetaT) / (iorFrom * cosThetaI + iorTo * cosThetaT);
  double rParallel =
      (iorFrom * cosThetaI - iorTo * cosThetaT) / (iorFrom * cosThetaI + iorTo * cosThetaT);
  return (rPerpendicular * rPerpendicular + rParallel * rParallel) / 2;
}

/* OrthoNormalBasis, src/math/OrthoNormalBasis.cpp */
typedef struct onb {
  v3 x, y, z;
} onb;
/* fromZ :44-51 */
static onb onb_from_z(v3 z) {
  const double Coincident = 0.9999;
  v3 xAxis = {1, 0, 0}, yAxis = {0, 1, 0};
  /* z.dot(Norm3::xAxis()) = z.x*1 + z.y*0 + z.z*0 */
  double zdotx = z.x * 1.0 + z.y * 0.0 + z.z * 0.0;
  v3 a = fabs(zdotx) > Coincident ? yAxis : xAxis;
  onb b;
  b.x = v3_normalised(v3_cross(a, z));
  b.y = v3_normalised(v3_cross(z, b.x));
  b.z = z;
  return b;
}
/* fromZY :34-38 (Norm3::fromNormal only asserts) */
static onb onb_from_zy(v3 z, v3 y) {
  onb b;
  b.x = v3_normalised(v3_cross(y, z));
  b.y = v3_cross(z, b.x);
  b.z = z;
  return b;
}
/* transform OrthoNormalBasis.h:18-20 */
static inline v3 onb_transform(const onb *b, v3 p) {
  return v3_add(v3_add(v3_scale(b->x, p.x), v3_scale(b->y, p.y)), v3_scale(b->z, p.z));
}

/* ------------------------------------------------------------------------------------- */
/* RNG                                                                                    */
/* ------------------------------------------------------------------------------------- */
/* std::mersenne_twister_engine<uint32,32,624,397,31,0x9908b0df,11,0xffffffff,7,0x9d2c5680,
 * 15,0xefc60000,18,1812433253>: ISO C++ [rand.eng.mers]. */
void oracle_mt_seed(oracle_mt19937 *mt, uint32_t seed) {
  mt->x[0] = seed;
  for (uint32_t i = 1; i < 624; ++i) {
    uint32_t prev = mt->x[i - 1];
    mt->x[i] = 1812433253u * (prev ^ (prev >> 30)) + i;
  }
  mt->pos = 624;
}

static void mt_regenerate(oracle_mt19937 *mt) {
  const uint32_t upper = 0x80000000u, lower = 0x7fffffffu, matrix = 0x9908b0dfu;
  uint32_t *x = mt->x;
  for (uint32_t k = 0; k < 624 - 397; ++k) {
    uint32_t y = (x[k] & upper) | (x[k + 1] & lower);
    x[k] = x[k + 397] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  }
  for (uint32_t k = 624 - 397; k < 623; ++k) {
    uint32_t y = (x[k] & upper) | (x[k + 1] & lower);
    x[k] = x[k + 397 - 624] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  }
  uint32_t y = (x[623] & upper) | (x[0] & lower);
  x[623] = x[396] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  mt->pos = 0;
}

uint32_t oracle_mt_next(oracle_mt19937 *mt) {
  if (mt->pos >= 624) mt_regenerate(mt);
  uint32_t z = mt->x[mt->pos++];
  z ^= (z >> 11) & 0xffffffffu;
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

/* PERPIXEL policy generator: sfc32 (Chris Doty-Humphrey's "small fast chaotic" 32-bit),
 * keyed a = pixel index, b = pass seed, c = golden-ratio constant, counter = 1, 12 warm-up
 * rounds.  This policy is the build's own definition (the reference has no per-pixel DoD
 * policy); the device implements the same recurrence. */
typedef struct sfc32 {
  uint32_t a, b, c, counter;
} sfc32;
static inline uint32_t sfc32_next(sfc32 *s) {
  uint32_t t = s->a + s->b + s->counter;
  s->counter += 1u;
  s->a = s->b ^ (s->b >> 9);
  s->b = s->c + (s->c << 3);
  s->c = ((s->c << 21) | (s->c >> 11)) + t;
  return t;
}
static void sfc32_seed(sfc32 *s, uint32_t pass_seed, uint32_t pixel_index) {
  s->a = pixel_index;
  s->b = pass_seed;
  s->c = 0x9E3779B9u;
  s->counter = 1u;
  for (int i = 0; i < 12; ++i) (void)sfc32_next(s);
}

typedef struct rng_t {
  int policy;
  oracle_mt19937 mt;
  sfc32 sfc;
  uint64_t words; /* 32-bit words consumed so far */
  uint64_t rays;  /* intersect() calls (statistics only) */
  /* Pick checksum of the current sample (parity instrumentation, include/ptw.h ptw_debug_options.d_picks):
   * sum over the sample's intersect() calls r = 0, 1, ... of (r + 1) * (combined primitive index + 1),
   * misses 0; combined index = position in Scene::intersect's scan: spheres, then triangles.  The
   * reference's IntersectionRecord carries no index, so this lives in the restatement - whose picks
   * are pinned to oracle/_ref through radiance and materials on every scene where those differ. */
  uint32_t sample_rays, picks;
} rng_t;

static inline uint32_t rng_word(rng_t *r) {
  r->words++;
  return r->policy == PTW_RNG_SEQUENTIAL ? oracle_mt_next(&r->mt) : sfc32_next(&r->sfc);
}

/* std::generate_canonical<double, 53>(urng) for a 32-bit engine, libstdc++ random.tcc:3348:
 * m = 2 calls; sum = w0 * 1.0 + w1 * 2^32 (rounded once, to nearest even);
 * ret = sum / 2^64; ret >= 1 -> nextafter(1, 0). */
static inline double rng_canonical(rng_t *r) {
  double sum = 0.0;
  double tmp = 1.0;
  sum += (double)rng_word(r) * tmp;
  tmp *= 4294967296.0;
  sum += (double)rng_word(r) * tmp;
  tmp *= 4294967296.0;
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}
/* std::uniform_real_distribution<double>(a, b)(rng) = canonical * (b - a) + a */
static inline double rng_uniform(rng_t *r, double a, double b) {
  return rng_canonical(r) * (b - a) + a;
}

void oracle_mt_words(uint32_t seed, uint32_t n, uint32_t *out) {
  oracle_mt19937 mt;
  oracle_mt_seed(&mt, seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = oracle_mt_next(&mt);
}
void oracle_mt_unit_doubles(uint32_t seed, uint32_t n, double *out) {
  rng_t r;
  memset(&r, 0, sizeof r);
  r.policy = PTW_RNG_SEQUENTIAL;
  oracle_mt_seed(&r.mt, seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = rng_uniform(&r, 0.0, 1.0);
}
void oracle_perpixel_words(uint32_t pass_seed, uint32_t pixel_index, uint32_t n, uint32_t *out) {
  sfc32 s;
  sfc32_seed(&s, pass_seed, pixel_index);
  for (uint32_t i = 0; i < n; ++i) out[i] = sfc32_next(&s);
}

/* ------------------------------------------------------------------------------------- */
/* Intersection                                                                           */
/* ------------------------------------------------------------------------------------- */
typedef struct ray_t {
  v3 o, d;
} ray_t;
typedef struct hit_t {
  int valid;
  double distance;
  int inside;
  v3 position, normal;
  uint32_t material;
  uint32_t prim; /* combined primitive index (pick checksum only; never read by the radiance path) */
} hit_t;

/* Ray::positionAlong Ray.h:25-27: origin + direction * t */
static inline v3 ray_position_along(const ray_t *r, double t) {
  return v3_add(r->o, v3_scale(r->d, t));
}

/* Scene::intersectSpheres src/dod/Scene.cpp:13-49.  dod::Sphere stores radius*radius
 * (Sphere.h:11-12). */
static hit_t intersect_spheres(const ptw_scene_view *s, const ray_t *ray, double nearerThan) {
  hit_t h;
  memset(&h, 0, sizeof h);
  double currentNearestDist = nearerThan;
  int64_t nearestIndex = -1;
  for (uint32_t i = 0; i < s->num_spheres; ++i) {
    const double *sp = s->sph_centre_radius + 4 * (size_t)i;
    v3 centre = v3_make(sp[0], sp[1], sp[2]);
    double radiusSquared = sp[3] * sp[3];
    v3 op = v3_sub(centre, ray->o);
    double b = v3_dot(op, ray->d);
    double determinant = b * b - v3_dot(op, op) + radiusSquared;
    if (determinant < 0) continue;
    determinant = sqrt(determinant);
    double minusT = b - determinant;
    double plusT = b + determinant;
    if (minusT < ORACLE_EPSILON && plusT < ORACLE_EPSILON) continue;
    double t = minusT > ORACLE_EPSILON ? minusT : plusT;
    if (t < currentNearestDist) {
      nearestIndex = i;
      currentNearestDist = t;
    }
  }
  if (nearestIndex < 0) return h;
  const double *sp = s->sph_centre_radius + 4 * (size_t)nearestIndex;
  v3 hitPosition = ray_position_along(ray, currentNearestDist);
  v3 normal = v3_normalised(v3_sub(hitPosition, v3_make(sp[0], sp[1], sp[2])));
  int inside = v3_dot(normal, ray->d) > 0;
  if (inside) normal = v3_neg(normal);
  h.valid = 1;
  h.distance = currentNearestDist;
  h.inside = inside;
  h.position = hitPosition;
  h.normal = normal;
  h.material = s->sph_material[nearestIndex];
  h.prim = (uint32_t)nearestIndex;
  return h;
}

/* Scene::intersectTriangles src/dod/Scene.cpp:51-113; TriangleVertices.h:25-35;
 * addTriangle stores faceNormal() three times, Scene.cpp:181-187. */
static hit_t intersect_triangles(const ptw_scene_view *s, const ray_t *ray, double nearerThan) {
  hit_t h;
  memset(&h, 0, sizeof h);
  double currentNearestDist = nearerThan;
  int64_t nIndex = -1;
  double nDet = 0, nU = 0, nV = 0;
  for (uint32_t i = 0; i < s->num_triangles; ++i) {
    const double *tv = s->tri_vertices + 9 * (size_t)i;
    v3 v0 = v3_from(tv), v1 = v3_from(tv + 3), v2 = v3_from(tv + 6);
    v3 uVector = v3_sub(v1, v0);
    v3 vVector = v3_sub(v2, v0);
    v3 pVec = v3_cross(ray->d, vVector);
    double det = v3_dot(uVector, pVec);
    if (fabs(det) < ORACLE_EPSILON) continue;
    double invDet = 1.0 / det;
    v3 tVec = v3_sub(ray->o, v0);
    double u = v3_dot(tVec, pVec) * invDet;
    v3 qVec = v3_cross(tVec, uVector);
    double v = v3_dot(ray->d, qVec) * invDet;
    /* Unpredictable::any(u < 0, u > 1, v < 0, u + v > 1), Unpredictable.h:8-10 */
    if ((unsigned)(u < 0.0) | (unsigned)(u > 1.0) | (unsigned)(v < 0.0) | (unsigned)(u + v > 1))
      continue;
    double t = v3_dot(vVector, qVec) * invDet;
    if (t > ORACLE_EPSILON && t < currentNearestDist) {
      nIndex = i;
      nDet = det;
      nU = u;
      nV = v;
      currentNearestDist = t;
    }
  }
  if (nIndex < 0) return h;
  const double *tv = s->tri_vertices + 9 * (size_t)nIndex;
  v3 v0 = v3_from(tv), v1 = v3_from(tv + 3), v2 = v3_from(tv + 6);
  /* faceNormal(): uVector().cross(vVector()).normalised(), TriangleVertices.h:33-35 */
  v3 faceNormal = v3_normalised(v3_cross(v3_sub(v1, v0), v3_sub(v2, v0)));
  v3 tn0 = faceNormal, tn1 = faceNormal, tn2 = faceNormal;
  v3 normalUdelta = v3_sub(tn1, tn0);
  v3 normalVdelta = v3_sub(tn2, tn0);
  v3 normal =
      v3_normalised(v3_add(v3_add(v3_lscale(nU, normalUdelta), v3_lscale(nV, normalVdelta)), tn0));
  int backfacing = nDet < ORACLE_EPSILON;
  h.valid = 1;
  h.distance = currentNearestDist;
  h.inside = backfacing;
  h.position = ray_position_along(ray, currentNearestDist);
  h.normal = backfacing ? v3_neg(normal) : normal;
  h.material = s->tri_material[nIndex];
  h.prim = s->num_spheres + (uint32_t)nIndex;
  return h;
}

/* Scene::intersect src/dod/Scene.cpp:115-122 */
static hit_t intersect(const ptw_scene_view *s, const ray_t *ray) {
  hit_t sphereRec = intersect_spheres(s, ray, INFINITY);
  hit_t triangleRec = intersect_triangles(s, ray, sphereRec.valid ? sphereRec.distance : INFINITY);
  return triangleRec.valid ? triangleRec : sphereRec;
}

static void hit_to_array(const hit_t *h, double out[9]) {
  if (!h->valid) {
    out[0] = -1.0;
    for (int i = 1; i < 9; ++i) out[i] = 0.0;
    return;
  }
  out[0] = h->distance;
  out[1] = h->inside ? 1.0 : 0.0;
  out[2] = h->position.x;
  out[3] = h->position.y;
  out[4] = h->position.z;
  out[5] = h->normal.x;
  out[6] = h->normal.y;
  out[7] = h->normal.z;
  out[8] = (double)h->material;
}
static ray_t ray_from_array(const double r[6]) {
  ray_t ray;
  ray.o = v3_make(r[0], r[1], r[2]);
  ray.d = v3_make(r[3], r[4], r[5]);
  return ray;
}
void oracle_intersect(const ptw_scene_view *scene, const double ray[6], double hit_out[9]) {
  ray_t r = ray_from_array(ray);
  hit_t h = intersect(scene, &r);
  hit_to_array(&h, hit_out);
}
void oracle_intersect_spheres(const ptw_scene_view *scene, const double ray[6],
                              double nearer_than, double hit_out[9]) {
  ray_t r = ray_from_array(ray);
  hit_t h = intersect_spheres(scene, &r, nearer_than);
  hit_to_array(&h, hit_out);
}
void oracle_intersect_triangles(const ptw_scene_view *scene, const double ray[6],
                                double nearer_than, double hit_out[9]) {
  ray_t r = ray_from_array(ray);
  hit_t h = intersect_triangles(scene, &r, nearer_than);
  hit_to_array(&h, hit_out);
}

/* ------------------------------------------------------------------------------------- */
/* Sampling (src/math/Samples.cpp)                                                        */
/* ------------------------------------------------------------------------------------- */
/* coneSample :6-19 */
static v3 cone_sample(v3 direction, double coneTheta, double u, double v) {
  if (coneTheta < ORACLE_EPSILON) return direction;
  coneTheta = coneTheta * (1.0 - (2.0 * acos(u) / M_PI));
  const double radius = sin(coneTheta);
  const double zScale = cos(coneTheta);
  const double randomTheta = v * 2 * M_PI;
  const onb basis = onb_from_z(direction);
  return v3_normalised(onb_transform(
      &basis, v3_make(cos(randomTheta) * radius, sin(randomTheta) * radius, zScale)));
}
/* hemisphereSample :21-30 */
static v3 hemisphere_sample(const onb *basis, double u, double v) {
  double theta = 2 * M_PI * u;
  double radiusSquared = v;
  double radius = sqrt(radiusSquared);
  return v3_normalised(onb_transform(
      basis, v3_make(cos(theta) * radius, sin(theta) * radius, sqrt(1 - radiusSquared))));
}

/* ------------------------------------------------------------------------------------- */
/* Scene::radiance src/dod/Scene.cpp:124-179                                              */
/* ------------------------------------------------------------------------------------- */
static v3 radiance(const ptw_scene_view *s, rng_t *rng, const ray_t *ray, int depth,
                   const ptw_render_params *rp) {
  int numUSamples = depth == 0 ? rp->first_bounce_u : 1;
  int numVSamples = depth == 0 ? rp->first_bounce_v : 1;
  if (depth >= rp->max_depth) return v3_make(0, 0, 0);

  rng->rays++;
  const hit_t hit = intersect(s, ray);
  rng->sample_rays++;
  if (hit.valid) rng->picks += rng->sample_rays * (hit.prim + 1u);
  if (!hit.valid) return v3_from(s->environment);

  const ptw_material *mat = &s->materials[hit.material];
  if (rp->preview) return v3_from(mat->diffuse);

  double iorFrom = hit.inside ? mat->index_of_refraction : 1.0;
  double iorTo = hit.inside ? 1.0 : mat->index_of_refraction;
  const double reflectivity = mat->reflectivity < 0
                                  ? n3_reflectance(hit.normal, ray->d, iorFrom, iorTo)
                                  : mat->reflectivity;
  const onb basis = onb_from_z(hit.normal);
  const v3 emission = v3_from(mat->emission);
  const v3 diffuse = v3_from(mat->diffuse);
  v3 result = v3_make(0, 0, 0);

  for (int uSample = 0; uSample < numUSamples; ++uSample) {
    for (int vSample = 0; vSample < numVSamples; ++vSample) {
      const double u = ((double)uSample + rng_uniform(rng, 0, 1.0)) / (double)numUSamples;
      const double v = ((double)vSample + rng_uniform(rng, 0, 1.0)) / (double)numVSamples;
      const double p = rng_uniform(rng, 0, 1.0);
      if (p < reflectivity) {
        ray_t newRay;
        newRay.o = hit.position;
        newRay.d = cone_sample(n3_reflect(hit.normal, ray->d), mat->reflection_cone_angle_rad, u, v);
        result = v3_add(result, v3_add(emission, radiance(s, rng, &newRay, depth + 1, rp)));
      } else {
        ray_t newRay;
        newRay.o = hit.position;
        newRay.d = hemisphere_sample(&basis, u, v);
        result = v3_add(
            result, v3_add(emission, v3_mul(diffuse, radiance(s, rng, &newRay, depth + 1, rp))));
      }
    }
  }
  /* Vec3::operator/(double) with the int product converted to double */
  return v3_div(result, (double)(numUSamples * numVSamples));
}

/* ------------------------------------------------------------------------------------- */
/* Camera (src/math/Camera.h)                                                             */
/* ------------------------------------------------------------------------------------- */
int oracle_camera_look_at(const double eye[3], const double look_at[3], const double up[3],
                          int32_t width, int32_t height, double vfov_degrees, ptw_camera *out) {
  /* ctor :40-46; callers pass camUp.normalised() (main.cpp:81) */
  v3 e = v3_from(eye), l = v3_from(look_at);
  v3 upn = v3_normalised(v3_from(up));
  onb axis = onb_from_zy(v3_normalised(v3_sub(l, e)), upn);
  memset(out, 0, sizeof *out);
  out->centre[0] = e.x, out->centre[1] = e.y, out->centre[2] = e.z;
  out->axis_x[0] = axis.x.x, out->axis_x[1] = axis.x.y, out->axis_x[2] = axis.x.z;
  out->axis_y[0] = axis.y.x, out->axis_y[1] = axis.y.y, out->axis_y[2] = axis.y.z;
  out->axis_z[0] = axis.z.x, out->axis_z[1] = axis.z.y, out->axis_z[2] = axis.z.z;
  out->aspect_ratio = (double)width / height;
  out->camera_plane_dist = 1.0 / tan(vfov_degrees * M_PI / 360.0);
  out->reciprocal_height = 1.0 / height;
  out->reciprocal_width = 1.0 / width;
  out->aperture_radius = 0.0;
  out->focal_distance = 0.0;
  return 0;
}
void oracle_camera_set_focus(ptw_camera *cam, const double focal_point[3], double aperture) {
  /* setFocus :48-51 */
  cam->focal_distance = v3_length(v3_sub(v3_from(focal_point), v3_from(cam->centre)));
  cam->aperture_radius = aperture;
}

/* Camera::rayFromUnit :20-37 */
static ray_t camera_ray_from_unit(const ptw_camera *c, double x, double y, rng_t *rng) {
  v3 ax = v3_from(c->axis_x), ay = v3_from(c->axis_y), az = v3_from(c->axis_z);
  v3 centre = v3_from(c->centre);
  v3 xContrib = v3_scale(v3_scale(ax, -x), c->aspect_ratio);
  v3 yContrib = v3_scale(ay, -y);
  v3 zContrib = v3_scale(az, c->camera_plane_dist);
  v3 direction = v3_normalised(v3_add(v3_add(xContrib, yContrib), zContrib));
  ray_t r;
  if (c->aperture_radius == 0) {
    r.o = centre;
    r.d = direction;
    return r;
  }
  v3 focalPoint = v3_add(centre, v3_scale(direction, c->focal_distance));
  double angle = rng_uniform(rng, 0, 2 * M_PI);
  double radius = rng_uniform(rng, 0, c->aperture_radius);
  v3 origin = v3_add(v3_add(centre, v3_scale(v3_scale(ax, cos(angle)), radius)),
                     v3_scale(v3_scale(ay, sin(angle)), radius));
  /* Ray::fromTwoPoints Ray.h:12-15 */
  r.o = origin;
  r.d = v3_normalised(v3_sub(focalPoint, origin));
  return r;
}
/* Camera::randomRay :54-60 */
static ray_t camera_random_ray(const ptw_camera *c, int pixelX, int pixelY, rng_t *rng) {
  double x = (pixelX + rng_uniform(rng, 0.0, 1.0)) * c->reciprocal_width;
  double y = (pixelY + rng_uniform(rng, 0.0, 1.0)) * c->reciprocal_height;
  return camera_ray_from_unit(c, 2 * x - 1, 2 * y - 1, rng);
}

void oracle_camera_ray(const ptw_camera *cam, int32_t px, int32_t py, uint32_t seed,
                       double ray_out[6]) {
  rng_t r;
  memset(&r, 0, sizeof r);
  r.policy = PTW_RNG_SEQUENTIAL;
  oracle_mt_seed(&r.mt, seed);
  ray_t ray = camera_random_ray(cam, px, py, &r);
  ray_out[0] = ray.o.x, ray_out[1] = ray.o.y, ray_out[2] = ray.o.z;
  ray_out[3] = ray.d.x, ray_out[4] = ray.d.y, ray_out[5] = ray.d.z;
}

/* ------------------------------------------------------------------------------------- */
/* The pass loop: worker lambda of Scene::render, src/dod/Scene.cpp:209-219               */
/* ------------------------------------------------------------------------------------- */
static int render_pass_impl(const ptw_scene_view *scene, const ptw_camera *camera,
                            const ptw_render_params *rp, int32_t pass_index,
                            double *radiance_out, uint32_t *words_out, uint64_t *rays_out,
                            uint32_t *picks_out) {
  const int width = rp->width, height = rp->height;
  if (width <= 0 || height <= 0) return PTW_ERR_INVALID;
  /* std::mt19937 rng(renderParams.seed + curSample++): int -> unsigned long -> mod 2^32 */
  const uint32_t pass_seed = (uint32_t)(rp->seed + rp->first_pass + pass_index);
  rng_t rng;
  memset(&rng, 0, sizeof rng);
  rng.policy = rp->rng_policy;
  int row_begin = 0, row_end = height, row_stride = 1, row_phase = 0;
  if (rp->rng_policy == PTW_RNG_SEQUENTIAL) {
    oracle_mt_seed(&rng.mt, pass_seed);
    /* include/ptw.h: under SEQUENTIAL only a PREFIX [0, row_end) of the rows can be rendered on
     * its own - it is what the full pass produces for those rows */
    if (rp->row_begin == 0 && rp->row_end > 0 && rp->row_end < height) row_end = rp->row_end;
  } else {
    /* the row window of include/ptw.h: (0,0) = all rows, begin == end != 0 = empty shard */
    if (rp->row_begin != 0 || rp->row_end != 0) {
      row_begin = rp->row_begin;
      row_end = rp->row_end;
    }
    if (rp->row_stride > 1) {
      row_stride = rp->row_stride;
      row_phase = rp->row_phase;
    }
  }
  for (int y = row_begin; y < row_end; ++y) {
    if (y % row_stride != row_phase) continue;
    for (int x = 0; x < width; ++x) {
      const size_t pix = (size_t)x + (size_t)y * width;
      if (rp->rng_policy == PTW_RNG_PERPIXEL) sfc32_seed(&rng.sfc, pass_seed, (uint32_t)pix);
      const uint64_t w0 = rng.words;
      rng.sample_rays = 0, rng.picks = 0;
      ray_t ray = camera_random_ray(camera, x, y, &rng);
      v3 c = radiance(scene, &rng, &ray, 0, rp);
      radiance_out[pix * 3 + 0] = c.x;
      radiance_out[pix * 3 + 1] = c.y;
      radiance_out[pix * 3 + 2] = c.z;
      if (words_out) words_out[pix] = (uint32_t)(rng.words - w0);
      if (picks_out) picks_out[pix] = rng.picks;
    }
  }
  if (rays_out) *rays_out = rng.rays;
  return PTW_OK;
}

int oracle_render_pass(const ptw_scene_view *scene, const ptw_camera *camera,
                       const ptw_render_params *params, int32_t pass_index,
                       double *radiance_out, uint32_t *words_out) {
  return render_pass_impl(scene, camera, params, pass_index, radiance_out, words_out, NULL, NULL);
}
int oracle_render_pass_picks(const ptw_scene_view *scene, const ptw_camera *camera,
                             const ptw_render_params *params, int32_t pass_index,
                             double *radiance_out, uint32_t *words_out, uint32_t *picks_out) {
  return render_pass_impl(scene, camera, params, pass_index, radiance_out, words_out, NULL, picks_out);
}

typedef struct job_t {
  const ptw_scene_view *scene;
  const ptw_camera *camera;
  const ptw_render_params *rp;
  double **pass_buffers; /* [spp] each width*height*3, allocated by the worker */
  uint32_t *words_out;
  uint32_t *picks_out;
  int next_pass;
  uint64_t rays;
  int failed;
  pthread_mutex_t lock;
} job_t;

static void *worker_main(void *arg) {
  job_t *job = (job_t *)arg;
  const size_t npix = (size_t)job->rp->width * job->rp->height;
  for (;;) {
    pthread_mutex_lock(&job->lock);
    int pass = job->next_pass < job->rp->samples_per_pixel ? job->next_pass++ : -1;
    pthread_mutex_unlock(&job->lock);
    if (pass < 0) break;
    double *buf = (double *)calloc(npix * 3, sizeof(double));
    uint64_t rays = 0;
    int rc = buf ? render_pass_impl(job->scene, job->camera, job->rp, pass, buf,
                                    job->words_out ? job->words_out + npix * (size_t)pass : NULL,
                                    &rays, job->picks_out ? job->picks_out + npix * (size_t)pass : NULL)
                 : PTW_ERR_INVALID;
    pthread_mutex_lock(&job->lock);
    job->pass_buffers[pass] = buf;
    job->rays += rays;
    if (rc != PTW_OK) job->failed = 1;
    pthread_mutex_unlock(&job->lock);
  }
  return NULL;
}

int oracle_render(const ptw_scene_view *scene, const ptw_camera *camera,
                  const ptw_render_params *params, int32_t threads, double *rgb_sum,
                  uint32_t *counts, uint32_t *words_out, uint64_t *rays_out) {
  return oracle_render_picks(scene, camera, params, threads, rgb_sum, counts, words_out, rays_out, NULL);
}

int oracle_render_picks(const ptw_scene_view *scene, const ptw_camera *camera,
                        const ptw_render_params *params, int32_t threads, double *rgb_sum,
                        uint32_t *counts, uint32_t *words_out, uint64_t *rays_out, uint32_t *picks_out) {
  if (!scene || !camera || !params || !rgb_sum || !counts) return PTW_ERR_INVALID;
  const int spp = params->samples_per_pixel;
  if (spp < 0 || params->width <= 0 || params->height <= 0) return PTW_ERR_INVALID;
  if (threads < 1) threads = 1;
  if (threads > spp && spp > 0) threads = spp;
  const size_t npix = (size_t)params->width * params->height;
  job_t job;
  memset(&job, 0, sizeof job);
  job.scene = scene;
  job.camera = camera;
  job.rp = params;
  job.words_out = words_out;
  job.picks_out = picks_out;
  job.pass_buffers = (double **)calloc((size_t)(spp > 0 ? spp : 1), sizeof(double *));
  pthread_mutex_init(&job.lock, NULL);
  pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  for (int i = 0; i < threads; ++i) pthread_create(&tids[i], NULL, worker_main, &job);
  for (int i = 0; i < threads; ++i) pthread_join(tids[i], NULL);
  free(tids);
  pthread_mutex_destroy(&job.lock);

  int row_begin = 0, row_end = params->height, row_stride = 1, row_phase = 0;
  if (params->rng_policy == PTW_RNG_PERPIXEL) {
    if (params->row_begin != 0 || params->row_end != 0) {
      row_begin = params->row_begin;
      row_end = params->row_end;
    }
    if (params->row_stride > 1) {
      row_stride = params->row_stride;
      row_phase = params->row_phase;
    }
  } else if (params->row_begin == 0 && params->row_end > 0 && params->row_end < params->height) {
    row_end = params->row_end; /* SEQUENTIAL: a prefix of the rows (see render_pass_impl) */
  }
  /* output += pass (ArrayOutput::operator+=, ArrayOutput.cpp:48-56), in pass order */
  for (int pass = 0; pass < spp; ++pass) {
    double *buf = job.pass_buffers[pass];
    if (!buf) continue;
    for (int y = row_begin; y < row_end; ++y) {
      if (y % row_stride != row_phase) continue;
      for (size_t pix = (size_t)y * params->width; pix < (size_t)(y + 1) * params->width; ++pix) {
        rgb_sum[pix * 3 + 0] += buf[pix * 3 + 0];
        rgb_sum[pix * 3 + 1] += buf[pix * 3 + 1];
        rgb_sum[pix * 3 + 2] += buf[pix * 3 + 2];
        counts[pix] += 1;
      }
    }
    free(buf);
  }
  (void)npix;
  free(job.pass_buffers);
  if (rays_out) *rays_out = job.rays;
  return job.failed ? PTW_ERR_INVALID : PTW_OK;
}

/* componentToInt src/util/ArrayOutput.cpp:9-12 */
uint8_t oracle_component_to_int(double x) {
  double c = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); /* std::clamp(x, 0.0, 1.0) */
  return (uint8_t)lround(pow(c, 1.0 / 2.2) * 255);
}

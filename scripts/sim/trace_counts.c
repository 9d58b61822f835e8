/* scripts/sim/trace_counts.c — measurement helper (NOT product code, NOT part of the oracle).
 * Includes the oracle's C restatement and walks one pass like render_pass_impl(), recording for
 * every pixel how many canonical draws each first-bounce sub-sample consumed (3 per level it
 * reached).  Feeds scripts/sim/spec_sim.py, which evaluates speculation schedules for the
 * SEQUENTIAL kernel against real count sequences.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared scripts/sim/trace_counts.c -o scripts/sim/libtrace.so -lm -lpthread */
#include "../../oracle/ptw_oracle.c"

/* counts_out[pix*17 + 0] = draws of the camera part (+0 when the primary ray misses: then
 * entries 1..16 are 0); counts_out[pix*17 + 1 + j] = draws of sub-sample j. */
int sim_trace_pass(const ptw_scene_view *scene, const ptw_camera *camera,
                   const ptw_render_params *rp, int32_t pass_index, uint8_t *counts_out) {
  const int width = rp->width, height = rp->height;
  const uint32_t pass_seed = (uint32_t)(rp->seed + rp->first_pass + pass_index);
  rng_t rng;
  memset(&rng, 0, sizeof rng);
  rng.policy = PTW_RNG_SEQUENTIAL;
  oracle_mt_seed(&rng.mt, pass_seed);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const size_t pix = (size_t)x + (size_t)y * width;
      uint8_t *c = counts_out + pix * 17;
      memset(c, 0, 17);
      uint64_t w0 = rng.words;
      ray_t ray = camera_random_ray(camera, x, y, &rng);
      c[0] = (uint8_t)((rng.words - w0) / 2);
      const hit_t hit = intersect(scene, &ray);
      if (!hit.valid) continue;
      const ptw_material *mat = &scene->materials[hit.material];
      double iorFrom = hit.inside ? mat->index_of_refraction : 1.0;
      double iorTo = hit.inside ? 1.0 : mat->index_of_refraction;
      const double reflectivity = mat->reflectivity < 0
                                      ? n3_reflectance(hit.normal, ray.d, iorFrom, iorTo)
                                      : mat->reflectivity;
      const onb basis = onb_from_z(hit.normal);
      int j = 0;
      for (int uS = 0; uS < rp->first_bounce_u; ++uS)
        for (int vS = 0; vS < rp->first_bounce_v; ++vS, ++j) {
          w0 = rng.words;
          const double u = ((double)uS + rng_uniform(&rng, 0, 1.0)) / (double)rp->first_bounce_u;
          const double v = ((double)vS + rng_uniform(&rng, 0, 1.0)) / (double)rp->first_bounce_v;
          const double p = rng_uniform(&rng, 0, 1.0);
          ray_t nr;
          nr.o = hit.position;
          if (p < reflectivity)
            nr.d = cone_sample(n3_reflect(hit.normal, ray.d), mat->reflection_cone_angle_rad, u, v);
          else
            nr.d = hemisphere_sample(&basis, u, v);
          (void)radiance(scene, &rng, &nr, 1, rp);
          if (j < 16) c[1 + j] = (uint8_t)((rng.words - w0) / 2);
        }
    }
  return 0;
}

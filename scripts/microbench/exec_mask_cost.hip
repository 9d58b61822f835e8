// Micro-benchmark: does gfx950 issue an fp64 VALU instruction faster when EXEC
// covers only some of the wave's four 16-lane groups?  If a quarter-wave EXEC made v_fma_f64 take
// one pass instead of four, the wave-uniform shading of the sequential kernels (175 of 201 VALU
// instructions per ray) could run in a quarter of its issue slots.
// One wave on a SIMD of its own; 8 v_fma_f64 (dependent / independent) per iteration, timed with
// s_memtime, empty loop subtracted; EXEC set with s_mov_b64 before the loop.
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask_cost exec_mask_cost.hip && ./exec_mask_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 4000
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void bench(unsigned long long *out, double *sink, double seed, unsigned long long mask) {
  double a = seed, b = seed * 0.5, c = seed * 0.25, d = 1.0 + seed, e = seed + 3, f = seed + 4, g = seed + 5, h = seed + 6;
  float fa = static_cast<float>(seed), fb = 0.5f, fc = 0.25f;
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1\n" : "=s"(saved) : "s"(mask));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if (KIND == 0) {
      asm volatile("" ::: "memory");
    } else if (KIND == 1) {
      asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    } else if (KIND == 2) {
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                   : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b), "+v"(c) : "v"(1.0000001), "v"(1e-9));
    } else if (KIND == 3) {
      asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(fa) : "v"(fb), "v"(fc));
    } else if (KIND == 4) {
      asm volatile(REP8("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));
    } else if (KIND == 5) {
      asm volatile(REP8("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_mov_b64 exec, %0\n" : : "s"(saved));
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = a + b + c + d + e + f + g + h + fa;
}

template <int KIND>
double run(unsigned long long *dOut, double *dSink, unsigned long long mask) {
  unsigned long long best = ~0ull, v;
  for (int r = 0; r < 5; ++r) {
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(64), 0, 0, dOut, dSink, 1.25, mask);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&v, dOut, 8, hipMemcpyDeviceToHost);
    if (v < best) best = v;
  }
  return double(best) / ITERS;
}

int main() {
  unsigned long long *dOut;
  double *dSink;
  (void)hipMalloc(&dOut, 8);
  (void)hipMalloc(&dSink, 64 * 8);
  const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0xffull, 0x1ull, 0x0001000100010001ull, 0xffff000000000000ull};
  const char *names[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-7", "lane 0", "one lane per 16-group", "lanes 48-63"};
  printf("%-24s %10s %10s %10s %10s %10s   (s_memtime ticks per instruction; 1 tick = 10 ns at 100 MHz)\n", "EXEC", "dep fma64",
         "ind fma64", "dep fma32", "dep mul64", "dep add64");
  for (int m = 0; m < 7; ++m) {
    const double base = run<0>(dOut, dSink, masks[m]);
    printf("%-24s %10.3f %10.3f %10.3f %10.3f %10.3f\n", names[m], (run<1>(dOut, dSink, masks[m]) - base) / 8,
           (run<2>(dOut, dSink, masks[m]) - base) / 8, (run<3>(dOut, dSink, masks[m]) - base) / 8,
           (run<4>(dOut, dSink, masks[m]) - base) / 8, (run<5>(dOut, dSink, masks[m]) - base) / 8);
  }
  return 0;
}

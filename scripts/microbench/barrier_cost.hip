// What a workgroup barrier costs when every wave arrives at once (gfx950): 2, 4, 5 and 8 waves of
// one workgroup loop over s_barrier (with the LDS release/acquire fences the kernels use).
//   hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip && ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2000
__global__ void bench(unsigned long long *out, int withFence) {
  __shared__ int x[64];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    if (withFence) {
      if (threadIdx.x == 0) x[i & 63] = i;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0 + (x[3] & 0);
}
int main() {
  unsigned long long *d, v;
  (void)hipMalloc(&d, 8);
  for (int waves : {1, 2, 4, 5, 8})
    for (int f : {0, 1}) {
      unsigned long long best = ~0ull;
      for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(bench, dim3(1), dim3(64 * waves), 0, 0, d, f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost);
        if (v < best) best = v;
      }
      printf("%d waves, %s: %.1f cycles per barrier\n", waves, f ? "LDS store + fences + s_barrier" : "s_barrier only", double(best) / ITERS);
    }
  return 0;
}

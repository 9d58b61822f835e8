// Micro-benchmark: what single-wave instruction sequences cost on gfx950 when ONE wave owns a SIMD
// (the situation of the SEQUENTIAL trace kernel).  Each construct is repeated REP times inside a
// loop of ITERS iterations, timed with s_memtime; the empty-loop time is subtracted.
//   hipcc --offload-arch=gfx950 -O3 -o issue_costs issue_costs.hip && ./issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 2000

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int KIND>
__global__ __launch_bounds__(64) void bench(unsigned long long *out, double *sink, double seed) {
  __shared__ double lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = seed + i;
  __syncthreads();
  double a = seed, b = seed * 0.5, c = seed * 0.25, d = 1.0 + seed, e = seed + 3, f = seed + 4, g = seed + 5, h = seed + 6;
  unsigned ldsAddr = (threadIdx.x & 7) * 8;
  unsigned long long sacc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  for (int it = 0; it < ITERS; ++it) {
    if (KIND == 0) {
      asm volatile("" ::: "memory");
    } else if (KIND == 1) { // 8 dependent v_fma_f64
      asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    } else if (KIND == 2) { // 8 independent v_fma_f64
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                   : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b), "+v"(c) : "v"(1.0000001), "v"(1e-9));
    } else if (KIND == 3) { // v_cmp -> s_cbranch_vccz not taken, x4
      asm volatile(REP4("v_cmp_lt_f64 vcc, %0, %1\n s_cbranch_vccz 1f\n") "1:\n" : : "v"(a), "v"(d) : "vcc");
    } else if (KIND == 4) { // v_cmp -> s_cbranch_vccnz TAKEN over one instruction, x4
      asm volatile("v_cmp_lt_f64 vcc, %0, %1\n s_cbranch_vccnz 1f\n v_mov_b32 v255, 0\n1:\n"
                   "v_cmp_lt_f64 vcc, %0, %1\n s_cbranch_vccnz 2f\n v_mov_b32 v255, 0\n2:\n"
                   "v_cmp_lt_f64 vcc, %0, %1\n s_cbranch_vccnz 3f\n v_mov_b32 v255, 0\n3:\n"
                   "v_cmp_lt_f64 vcc, %0, %1\n s_cbranch_vccnz 4f\n v_mov_b32 v255, 0\n4:\n"
                   : : "v"(a), "v"(d) : "vcc", "v255");
    } else if (KIND == 5) { // ds_read_b64 + wait, x4 (independent addresses)
      asm volatile(REP4("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=v"(a) : "v"(ldsAddr) : "memory");
    } else if (KIND == 6) { // 4 ds_read_b128 then one wait
      double2 x0, x1, x2, x3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(ldsAddr) : "memory");
      a += x0.x + x1.x + x2.x + x3.x;
    } else if (KIND == 7) { // 8 v_readlane (dynamic lane in sgpr) + one VALU use
      int lane = __builtin_amdgcn_readfirstlane(it & 63);
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("v_readlane_b32 %0, %8, %9\n v_readlane_b32 %1, %8, %9\n v_readlane_b32 %2, %8, %9\n v_readlane_b32 %3, %8, %9\n"
                   "v_readlane_b32 %4, %8, %9\n v_readlane_b32 %5, %8, %9\n v_readlane_b32 %6, %8, %9\n v_readlane_b32 %7, %8, %9\n"
                   : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3), "=s"(r4), "=s"(r5), "=s"(r6), "=s"(r7) : "v"(ldsAddr), "s"(lane));
      sacc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    } else if (KIND == 8) { // s_and_saveexec + s_cbranch_execz (not taken) + 1 VALU + restore, x4
      asm volatile(REP4("s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 1f\n v_add_f64 %0, %0, 1.0\n s_or_b64 exec, exec, s[20:21]\n") "1:\n s_or_b64 exec, exec, s[20:21]\n"
                   : "+v"(a) : "s"(0xffffull) : "s20", "s21");
    } else if (KIND == 9) { // 8 dependent SALU
      unsigned s = it;
      asm volatile(REP8("s_add_u32 %0, %0, 3\n") : "+s"(s)::"scc");
      sacc += s;
    } else if (KIND == 10) { // 4 dependent v_rcp_f64
      asm volatile(REP4("v_rcp_f64 %0, %0\n") : "+v"(d));
    } else if (KIND == 11) { // ballot -> s_cmp -> branch (the uniformBool idiom), x4, not taken
      asm volatile(REP4("v_cmp_lt_f64 s[20:21], %0, %1\n s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc0 1f\n") "1:\n" : : "v"(a), "v"(d) : "s20", "s21", "scc");
    } else if (KIND == 12) { // v_cmp -> v_cndmask (select), x4, no branch
      asm volatile(REP4("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n") : : "v"(a), "v"(d), "v"(ldsAddr), "v"(ldsAddr) : "vcc");
    } else if (KIND == 13) { // 8 s_nop 0
      asm volatile(REP8("s_nop 0\n"));
    } else if (KIND == 14) { // ds_write_b64 x4 (no wait)
      asm volatile(REP4("ds_write_b64 %0, %1\n") : : "v"(ldsAddr), "v"(a) : "memory");
    } else if (KIND == 15) { // dependent LDS chain: address from previous load, x2
      unsigned ad = ldsAddr;
      asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf8, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xf8, %0\n" : "+v"(ad)::"memory");
      ldsAddr = ad;
    } else if (KIND == 16) { // 4 x (v_readlane -> VALU using the SGPR immediately)
      int lane = __builtin_amdgcn_readfirstlane(it & 63);
      asm volatile(REP4("v_readlane_b32 s20, %1, %2\n v_add_u32 %0, s20, %0\n") : "+v"(ldsAddr) : "v"(ldsAddr), "s"(lane) : "s20");
    } else if (KIND == 18) { // s_and_saveexec + s_cbranch_execz not taken + restore, x4
      asm volatile("s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 1f\n v_add_f64 %0, %0, 1.0\n1:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 2f\n v_add_f64 %0, %0, 1.0\n2:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 3f\n v_add_f64 %0, %0, 1.0\n3:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 4f\n v_add_f64 %0, %0, 1.0\n4:\n s_or_b64 exec, exec, s[20:21]\n"
                   : "+v"(a) : "s"(0xffffull) : "s20", "s21", "scc");
    } else if (KIND == 19) { // same but exec becomes zero: branch TAKEN
      asm volatile("s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 1f\n v_add_f64 %0, %0, 1.0\n1:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 2f\n v_add_f64 %0, %0, 1.0\n2:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 3f\n v_add_f64 %0, %0, 1.0\n3:\n s_or_b64 exec, exec, s[20:21]\n"
                   "s_and_saveexec_b64 s[20:21], %1\n s_cbranch_execz 4f\n v_add_f64 %0, %0, 1.0\n4:\n s_or_b64 exec, exec, s[20:21]\n"
                   : "+v"(a) : "s"(0ull) : "s20", "s21", "scc");
    } else if (KIND == 20) { // s_cmp -> s_cbranch_scc1 NOT taken, x4
      asm volatile("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_cmp_eq_u32 0, 1\n s_cbranch_scc1 2f\n s_nop 0\n2:\n"
                   "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 3f\n s_nop 0\n3:\n s_cmp_eq_u32 0, 1\n s_cbranch_scc1 4f\n s_nop 0\n4:\n" ::: "scc");
    } else if (KIND == 21) { // s_cbranch_vccz not taken, vcc written long before (no VALU dependency), x4
      asm volatile("s_cbranch_vccz 1f\n s_nop 0\n1:\n s_cbranch_vccz 2f\n s_nop 0\n2:\n s_cbranch_vccz 3f\n s_nop 0\n3:\n s_cbranch_vccz 4f\n s_nop 0\n4:\n");
    } else if (KIND == 22) { // unconditional s_branch over one instruction, x4
      asm volatile("s_branch 1f\n s_nop 0\n1:\n s_branch 2f\n s_nop 0\n2:\n s_branch 3f\n s_nop 0\n3:\n s_branch 4f\n s_nop 0\n4:\n");
    } else if (KIND == 23) { // v_cmp, 6 independent VALU, then s_cbranch_vccz not taken (distance hides the dependency?), x2
      asm volatile("v_cmp_lt_f64 vcc, %0, %1\n" REP4("v_fma_f64 %2, %2, %3, %4\n") "v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n s_cbranch_vccz 1f\n s_nop 0\n1:\n"
                   "v_cmp_lt_f64 vcc, %0, %1\n" REP4("v_fma_f64 %2, %2, %3, %4\n") "v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n s_cbranch_vccz 2f\n s_nop 0\n2:\n"
                   : : "v"(a), "v"(d), "v"(e), "v"(b), "v"(c) : "vcc");
    } else if (KIND == 24) { // 4 x ds_read_b64 then one wait
      double x0, x1, x2, x3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:64\n ds_read_b64 %2, %4 offset:128\n ds_read_b64 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(ldsAddr) : "memory");
      a += x0; d += x1; e += x2; f += x3;
    } else if (KIND == 25) { // 1 x ds_read_b128 + wait
      double2 x0;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(x0) : "v"(ldsAddr) : "memory");
      a += x0.x;
    } else if (KIND == 26) { // 4 x ds_read_b128 + wait, all lanes the same address
      double2 x0, x1, x2, x3;
      unsigned ad = 0;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(ad) : "memory");
      a += x0.x; d += x1.x; e += x2.x; f += x3.x;
    } else if (KIND == 27) { // 8 x ds_read_b64 + wait, all lanes the same address
      double x0, x1, x2, x3, x4, x5, x6, x7;
      unsigned ad = 0;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                   "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(ad) : "memory");
      a += x0 + x4; d += x1 + x5; e += x2 + x6; f += x3 + x7;
    } else if (KIND == 28) { // v_readlane with a CONSTANT lane, x8
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %8, 4\n v_readlane_b32 %2, %8, 5\n v_readlane_b32 %3, %8, 6\n"
                   "v_readlane_b32 %4, %8, 7\n v_readlane_b32 %5, %8, 8\n v_readlane_b32 %6, %8, 9\n v_readlane_b32 %7, %8, 10\n"
                   : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3), "=s"(r4), "=s"(r5), "=s"(r6), "=s"(r7) : "v"(ldsAddr));
      sacc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    } else if (KIND == 29) { // v_readfirstlane x8
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("v_readfirstlane_b32 %0, %8\n v_readfirstlane_b32 %1, %8\n v_readfirstlane_b32 %2, %8\n v_readfirstlane_b32 %3, %8\n"
                   "v_readfirstlane_b32 %4, %8\n v_readfirstlane_b32 %5, %8\n v_readfirstlane_b32 %6, %8\n v_readfirstlane_b32 %7, %8\n"
                   : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3), "=s"(r4), "=s"(r5), "=s"(r6), "=s"(r7) : "v"(ldsAddr));
      sacc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    } else if (KIND == 30) { // v_mov_b32_dpp row_bcast-style moves x8
      asm volatile(REP8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(ldsAddr));
    } else if (KIND == 17) { // taken scalar branch (s_cbranch_scc1 taken), x4
      asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 2f\n s_nop 0\n2:\n"
                   "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 3f\n s_nop 0\n3:\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 4f\n s_nop 0\n4:\n" ::: "scc");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = a + b + c + d + e + f + g + h + ldsAddr + sacc;
}

template <int KIND>
double run(unsigned long long *dOut, double *dSink) {
  unsigned long long best = ~0ull, v;
  for (int r = 0; r < 5; ++r) {
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(64), 0, 0, dOut, dSink, 1.25);
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
  const double base = run<0>(dOut, dSink);
  printf("empty loop: %.1f cycles/iter (s_memtime ticks)\n", base);
#define R(K, n, what) printf("%-62s %6.1f cycles each\n", what, (run<K>(dOut, dSink) - base) / n)
  R(1, 8, "dependent v_fma_f64");
  R(2, 8, "independent v_fma_f64");
  R(10, 4, "dependent v_rcp_f64");
  R(9, 8, "dependent s_add_u32");
  R(13, 8, "s_nop 0");
  R(3, 4, "v_cmp_f64 -> s_cbranch_vccz (not taken)");
  R(4, 4, "v_cmp_f64 -> s_cbranch_vccnz (taken, skips 1 instr)");
  R(11, 4, "v_cmp_f64 -> s_cmp_lg_u64 -> s_cbranch_scc0 (not taken)");
  R(17, 4, "s_cmp -> s_cbranch_scc1 (taken)");
  R(20, 4, "s_cmp -> s_cbranch_scc1 (not taken)");
  R(21, 4, "s_cbranch_vccz alone (not taken, vcc old)");
  R(22, 4, "s_branch (unconditional)");
  R(23, 2, "v_cmp; 6 x v_fma; s_cbranch_vccz not taken (whole group)");
  R(18, 4, "s_and_saveexec; s_cbranch_execz NOT taken; v_add; s_or exec");
  R(19, 4, "s_and_saveexec; s_cbranch_execz TAKEN; s_or exec");
  R(12, 4, "v_cmp_f64 -> v_cndmask_b32");
  R(8, 4, "s_and_saveexec + s_cbranch_execz(not taken) + v_add + s_or exec");
  R(5, 4, "ds_read_b64 + s_waitcnt");
  R(6, 1, "4 x ds_read_b128 + one s_waitcnt (+4 v_add)");
  R(15, 2, "dependent ds_read_b32 (+wait +v_and)");
  R(24, 1, "4 x ds_read_b64 + one s_waitcnt (+4 v_add)");
  R(25, 1, "1 x ds_read_b128 + s_waitcnt (+1 v_add)");
  R(26, 1, "4 x ds_read_b128 same address + one wait (+4 v_add)");
  R(27, 1, "8 x ds_read_b64 same address + one wait (+8 v_add)");
  R(28, 8, "v_readlane_b32 (constant lane)");
  R(29, 8, "v_readfirstlane_b32");
  R(30, 8, "v_mov_b32_dpp + s_nop 1");
  R(14, 4, "ds_write_b64 (no wait)");
  R(7, 8, "v_readlane_b32 (sgpr lane)");
  R(16, 4, "v_readlane_b32 -> v_add_u32 using it");
  return 0;
}

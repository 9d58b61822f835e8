// ptw_device.h — device-side fp64 math of the hip way (gfx950 / CDNA4, wave64).
//
// Restates, for one GPU lane, every value-type function the reference's radiance path calls
// (src/math/*.h, src/math/Samples.cpp, src/math/Norm3.cpp) with the reference's operation
// order.  The compiler may contract a*b+c into v_fma_f64 (one rounding instead of two); that
// is within the reference's own build-to-build spread (its CMakeLists.txt:21 builds with
// -march=native -funsafe-math-optimizations, i.e. with FMA contraction), and parity is
// asserted to 1e-12 relative with bit-exact RNG word counts (tests/test_gpu_parity.py).
// Division and square roots are NOT the IEEE sequences by default: v_rcp_f64 / v_rsq_f64 seeds refined by
// two Newton steps (PTW_FAST_MATH below, about 1 ulp; -DPTW_FAST_MATH=0 restores the correctly rounded
// operations, `make ieee`); no compiler fast-math flag is used.  DESIGN.md section 4 says what either
// deviation can and cannot change.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ptwd {

constexpr double kEpsilon = 0.000000001; // src/math/Epsilon.h:3
constexpr double kPi = 3.14159265358979323846;
constexpr double kInf = __builtin_huge_val();

// ---- reciprocal / square roots ------------------------------------------------------------
// PTW_FAST_MATH (default): v_rcp_f64 / v_rsq_f64 seeds refined by Newton steps in fma form,
// accurate to about 1 ulp, instead of the ~13 / ~25 instruction IEEE-exact sequences.  They
// feed only continuous quantities (directions, positions, hit distances); the discrete
// outcomes of the algorithm (hit/miss, lobe choice) change only if a comparison lands within
// ~1e-15 relative of its threshold - the same exposure FMA contraction already has - and
// the radiance VALUE depends only on those discrete outcomes.  Parity tests assert exact RNG
// word counts, which would expose any systematic effect.  -DPTW_FAST_MATH=0 restores IEEE ops.
#ifndef PTW_FAST_MATH
#define PTW_FAST_MATH 1
#endif

__device__ __forceinline__ double rcp(double x) {
#if PTW_FAST_MATH
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
#else
  return 1.0 / x;
#endif
}
// 1 / sqrt(x) for x > 0
__device__ __forceinline__ double rsqrt(double x) {
#if PTW_FAST_MATH
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-(x * y), y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  e = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y * 0.5, e, y);
#else
  return 1.0 / __builtin_sqrt(x);
#endif
}
// sqrt(x) for x >= 0
__device__ __forceinline__ double sqrtPos(double x) {
#if PTW_FAST_MATH
  const double y = rsqrt(x);
  double s = x * y;
  const double r = __builtin_fma(-s, s, x);
  s = __builtin_fma(r, 0.5 * y, s);
  return x == 0.0 ? 0.0 : s;
#else
  return __builtin_sqrt(x);
#endif
}

struct d3 {
  double x, y, z;
};

__device__ __forceinline__ d3 mk(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator-(d3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// Vec3::dot, src/math/Vec3.h:81-83
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vec3::cross, src/math/Vec3.h:85-90
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Vec3::normalised = *this / length(): multiply by 1.0 / sqrt(dot) (Vec3.h:51-54, impl.h:5-7)
__device__ __forceinline__ d3 normalised(d3 a) {
  const double reciprocal = rsqrt(dot(a, a));
  return mk(a.x * reciprocal, a.y * reciprocal, a.z * reciprocal);
}

struct Basis {
  d3 x, y, z;
};
// OrthoNormalBasis::fromZ, src/math/OrthoNormalBasis.cpp:44-51
__device__ __forceinline__ Basis basisFromZ(d3 z) {
  const bool coincident = __builtin_fabs(z.x * 1.0 + z.y * 0.0 + z.z * 0.0) > 0.9999;
  const d3 a = coincident ? mk(0, 1, 0) : mk(1, 0, 0);
  Basis b;
  b.x = normalised(cross(a, z));
  b.y = normalised(cross(z, b.x));
  b.z = z;
  return b;
}
// OrthoNormalBasis::transform, src/math/OrthoNormalBasis.h:18-20
__device__ __forceinline__ d3 transform(const Basis &b, d3 p) {
  return (b.x * p.x + b.y * p.y) + b.z * p.z;
}

// Norm3::reflect, src/math/Norm3.impl.h:41-44: incoming - (n * 2) * n.dot(incoming)
__device__ __forceinline__ d3 reflect(d3 n, d3 incoming) {
  return incoming - (n * 2.0) * dot(n, incoming);
}

// Norm3::reflectance, src/math/Norm3.cpp:7-24.  rParallel is the same expression as
// rPerpendicular upstream, so the result is (r*r + r*r) / 2.  iorRatio = iorFrom / iorTo is
// passed in (the host precomputes 1/ior with the same correctly rounded division).
__device__ __forceinline__ double reflectance(d3 n, d3 incoming, double iorFrom, double iorTo,
                                              double iorRatio) {
  const double cosThetaI = -dot(n, incoming);
  const double sinThetaTSquared = iorRatio * iorRatio * (1 - cosThetaI * cosThetaI);
  if (sinThetaTSquared > 1) return 1.0;
  const double cosThetaT = sqrtPos(1 - sinThetaTSquared);
  const double r =
      (iorFrom * cosThetaI - iorTo * cosThetaT) * rcp(iorFrom * cosThetaI + iorTo * cosThetaT);
  return (r * r + r * r) / 2;
}

// sin and cos of x for the arguments this path produces (theta = 2*pi*u, cone angles: all in
// [0, 2*pi]).  Cody-Waite reduction by multiples of pi/2 in three exact-product steps, then the
// classic minimax kernels on [-pi/4, pi/4] (the fdlibm/FreeBSD k_sin / k_cos polynomials,
// error < 1 ulp).  Arguments outside [-6.5, 6.5] take ocml's sincos (out of line).
__device__ __noinline__ void sincosGeneral(double x, double *s, double *c) { sincos(x, s, c); }

// A floating-point constant held in a SCALAR register pair at its point of use.  The Horner chains
// below add a constant per step; left to itself the compiler keeps a VECTOR-register copy of every
// addend (v_fmac wants the addend in its destination) and hoists the copies out of the pixel loop.
// Where registers are short - the master path of the two-master worker-wave kernels, the PERPIXEL
// kernels at four waves per SIMD - it then spills them: round 3's two-master kernels reloaded four of
// them from scratch memory, three serialised vmcnt(0) waits, in the middle of every first-bounce
// scatter.  v_fma_f64 takes one scalar operand directly: with the constants in scalar registers
// those kernels spill nothing (28 -> 0 and 10 -> 0 registers).  It is NOT free where a wave has its
// SIMD to itself: two s_mov_b32 per constant are two issue slots on the serial path, and the
// single-wave-per-SIMD kernels (traceSequentialSpec: 7.60 against 7.76 Msamples/s on the headline
// scene, same box) never spilled - so the choice is a template parameter (SC), per kernel family.
// Same operations in the same order either way: same bits.
template <bool SC>
__device__ __forceinline__ double sconst(double c) {
  if constexpr (SC) asm volatile("" : "+s"(c));
  return c;
}

template <bool IN_RANGE = false, bool SC = false> // IN_RANGE: the caller guarantees 0 <= x <= 6.5
__device__ __forceinline__ void sinCos(double x, double &sn, double &cs) {
  // |x| <= 6.5 runs the reduction below (it is exact for negative multiples of pi/2 as well: fn is
  // then negative and every product fn * c_i stays exact); cone angles are in [-pi, pi]
  if (!IN_RANGE && !(__builtin_fabs(x) <= 6.5)) {
    sincosGeneral(x, &sn, &cs);
    return;
  }
  const double fn = __builtin_rint(x * 6.36619772367581382433e-01); // x * 2/pi -> 0..4
  // pi/2 = c1 + c2 + c3 (+ ...): c1 has 33 significant bits, so fn * c1 is exact
  double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-fn, 6.07710050630396597660e-11, r);
  r = __builtin_fma(-fn, 2.02226624879595063154e-21, r);
  const double z = r * r;
  // k_sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, sconst<SC>(S5)), sconst<SC>(S4)), sconst<SC>(S3)), sconst<SC>(S2));
  const double ks = __builtin_fma(z * r, __builtin_fma(z, ps, sconst<SC>(S1)), r);
  // k_cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, sconst<SC>(C5)), sconst<SC>(C4)), sconst<SC>(C3)), sconst<SC>(C2)), sconst<SC>(C1));
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double kc = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = static_cast<int>(fn) & 3;
  const double s0 = (q & 1) ? kc : ks;
  const double c0 = (q & 1) ? ks : kc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// normalised() for a vector whose length is 1 up to rounding (|len^2 - 1| << 1e-7):
// 1/sqrt(1 + e) = 1 - e/2 + O(e^2), and the O(e^2) term is below half an ulp.
__device__ __forceinline__ d3 normalisedNearUnit(d3 a) {
#if PTW_FAST_MATH
  const double reciprocal = __builtin_fma(-0.5, dot(a, a) - 1.0, 1.0);
  return mk(a.x * reciprocal, a.y * reciprocal, a.z * reciprocal);
#else
  return normalised(a);
#endif
}

// hemisphereSample, src/math/Samples.cpp:21-30
template <bool SC = false>
__device__ __forceinline__ d3 hemisphereSample(const Basis &basis, double u, double v) {
  const double theta = (2 * kPi) * u;
  const double radius = sqrtPos(v);
  double s, c;
  sinCos<true, SC>(theta, s, c); // u is a (stratified) canonical draw: theta in [0, 2 pi)
  // (c r, s r, sqrt(1 - v)) has squared length v + (1 - v) and the basis is orthonormal, so the
  // transformed vector is unit length up to a few ulp
  return normalisedNearUnit(transform(basis, mk(c * radius, s * radius, sqrtPos(1 - v))));
}

// coneSample, src/math/Samples.cpp:6-19
__device__ __noinline__ d3 coneSample(d3 direction, double coneTheta, double u, double v) {
  if (coneTheta < kEpsilon) return direction;
  coneTheta = coneTheta * (1.0 - (2.0 * acos(u) / kPi));
  double radius, zScale;
  sinCos<false, true>(coneTheta, radius, zScale);
  const double randomTheta = v * 2 * kPi;
  double s, c;
  sinCos<true, true>(randomTheta, s, c); // v is a (stratified) canonical draw
  const Basis basis = basisFromZ(direction);
  return normalised(transform(basis, mk(c * radius, s * radius, zScale)));
}

// ---- wave64 cross-lane helpers ----------------------------------------------------------
__device__ __forceinline__ int lo32(double x) { return __double2loint(x); }
__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }
__device__ __forceinline__ double mk64(int lo, int hi) { return __hiloint2double(hi, lo); }

__device__ __forceinline__ double readLane(double x, int lane) {
  return mk64(__builtin_amdgcn_readlane(lo32(x), lane), __builtin_amdgcn_readlane(hi32(x), lane));
}
__device__ __forceinline__ double readFirstLane(double x) {
  return mk64(__builtin_amdgcn_readfirstlane(lo32(x)), __builtin_amdgcn_readfirstlane(hi32(x)));
}
// A condition every active lane agrees on, as a scalar branch condition: the compare's lane
// mask tested against zero (v_cmp -> s_cmp_lg_u64 -> s_cbranch), no VGPR round trip.
__device__ __forceinline__ bool uniformBool(bool b) {
  return __builtin_amdgcn_ballot_w64(b) != 0;
}

// Orders this wave's LDS/global accesses for cross-LANE communication through memory.  The
// lanes of a wave execute in lockstep, but the compiler only sees one thread: without a fence
// it may hoist a later load above an earlier store to an address that (for this thread) does
// not alias, which breaks data another lane wrote.  No instruction is emitted beyond waits.
__device__ __forceinline__ void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not
// for its outstanding global stores (which __syncthreads() would also drain).
__device__ __forceinline__ void ldsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int Ctrl, int RowMask>
__device__ __forceinline__ double dppMove(double x) {
  // lanes the DPP pattern does not feed keep their own value (old = x, bound_ctrl = 0)
  const int lo = __builtin_amdgcn_update_dpp(lo32(x), lo32(x), Ctrl, RowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(hi32(x), hi32(x), Ctrl, RowMask, 0xf, false);
  return mk64(lo, hi);
}
template <int Ctrl, int RowMask>
__device__ __forceinline__ unsigned dppMoveU(unsigned x) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(
      static_cast<int>(x), static_cast<int>(x), Ctrl, RowMask, 0xf, false));
}

// Minimum over the 64 lanes of a wave, returned wave-uniform.  DPP row shifts inside each
// 16-lane row, then row_bcast15 / row_bcast31 (gfx9-family DPP) into lane 63.
// v_min_f64 without the v_max(x, x) canonicalisation fmin() adds: inputs are never NaN here.
__device__ __forceinline__ double vmin64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double waveMin(double x) {
  x = vmin64(x, dppMove<0x111, 0xf>(x)); // row_shr:1
  x = vmin64(x, dppMove<0x112, 0xf>(x)); // row_shr:2
  x = vmin64(x, dppMove<0x114, 0xf>(x)); // row_shr:4
  x = vmin64(x, dppMove<0x118, 0xf>(x)); // row_shr:8
  x = vmin64(x, dppMove<0x142, 0xa>(x)); // row_bcast:15 -> rows 1,3
  x = vmin64(x, dppMove<0x143, 0xc>(x)); // row_bcast:31 -> rows 2,3
  return readLane(x, 63);
}
// Wave-wide unsigned minimum with the DPP operand fused into v_min_u32 (one VALU op per step;
// lanes whose DPP source is outside the row / masked rows are not written and keep their own
// value).  `s_nop 1` covers the VALU-write -> DPP-read hazard inside the asm block.
__device__ __forceinline__ unsigned waveMinUFused(unsigned x) {
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(x));
  return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(x), 63));
}
// Minimum of non-negative doubles (or +inf) over the wave: their IEEE bit patterns order like
// unsigned 64-bit integers, so reduce the high words, then the low words of the lanes that hold
// the minimal high word.
__device__ __forceinline__ double waveMinPositive(double x, unsigned &hiOut, unsigned &loOut) {
  const unsigned hi = static_cast<unsigned>(hi32(x)), lo = static_cast<unsigned>(lo32(x));
  const unsigned mhi = waveMinUFused(hi);
  const unsigned mlo = waveMinUFused(hi == mhi ? lo : 0xffffffffu);
  hiOut = mhi;
  loOut = mlo;
  return mk64(static_cast<int>(mlo), static_cast<int>(mhi));
}

__device__ __forceinline__ unsigned waveMinU(unsigned x) {
  x = min(x, dppMoveU<0x111, 0xf>(x));
  x = min(x, dppMoveU<0x112, 0xf>(x));
  x = min(x, dppMoveU<0x114, 0xf>(x));
  x = min(x, dppMoveU<0x118, 0xf>(x));
  x = min(x, dppMoveU<0x142, 0xa>(x));
  x = min(x, dppMoveU<0x143, 0xc>(x));
  return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(x), 63));
}

// ---- std::mt19937 pieces ------------------------------------------------------------------
__device__ __forceinline__ uint32_t mtTwist(uint32_t cur, uint32_t next, uint32_t far) {
  const uint32_t y = (cur & 0x80000000u) | (next & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mtTemper(uint32_t z) {
  z ^= (z >> 11);
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}
// std::generate_canonical<double,53> for a 32-bit engine (libstdc++ random.tcc:3348-3380):
// (w0 + w1 * 2^32) rounded once to nearest-even, / 2^64, clamped below 1.
__device__ __forceinline__ double canonicalFromWords(uint32_t w0, uint32_t w1) {
  const double sum =
      __builtin_fma(static_cast<double>(w1), 4294967296.0, static_cast<double>(w0));
  double ret = sum * 0x1p-64;
  if (ret >= 1.0) ret = 0x1.fffffffffffffp-1; // nextafter(1.0, 0.0)
  return ret;
}

// sfc32, the PERPIXEL policy stream (see oracle/ptw_oracle.c for the definition).
struct Sfc32 {
  uint32_t a, b, c, counter;
  __device__ __forceinline__ uint32_t next() {
    const uint32_t t = a + b + counter;
    counter += 1u;
    a = b ^ (b >> 9);
    b = c + (c << 3);
    c = ((c << 21) | (c >> 11)) + t;
    return t;
  }
  __device__ __forceinline__ void seed(uint32_t passSeed, uint32_t pixelIndex) {
    a = pixelIndex;
    b = passSeed;
    c = 0x9E3779B9u;
    counter = 1u;
#pragma unroll
    for (int i = 0; i < 12; ++i) (void)next();
  }
};

} // namespace ptwd

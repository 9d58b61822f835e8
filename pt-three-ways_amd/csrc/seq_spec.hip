// seq_spec.hip - traceSequentialSpec: the SEQUENTIAL policy for scenes of at most 64 triangles with the
// first-bounce fan-out traced speculatively by four waves (the headline kernel).
#include "ptw_launch.h"
#include "ptw_seq_ctx.h"

namespace ptw {
using namespace ptwd;
namespace {

// -----------------------------------------------------------------------------------------
// traceSequentialSpec: SEQUENTIAL policy for scenes of at most 64 triangles, with the
// sub-samples of the first-bounce fan-out traced SPECULATIVELY in parallel.
//
// The stream makes everything serial: sub-sample j+1 starts where sub-sample j stopped, and how
// many draws j consumes (3 per level it reaches) is known only when it is done.  But that count
// takes few values, and the values repeat (a closed scene mostly runs every path to the depth cap,
// an open one mostly loses the first ray).  So a workgroup of kSpecWaves waves - one per SIMD of
// a CU, each holding the whole scene in registers like the single-wave kernel - works per round
// on:  wave 0: sub-sample j at the true stream position (the frontier);
//      wave 1: sub-sample j+1, assuming j consumes m1 (the most recent count);
//      wave 2: sub-sample j+1 assuming m2 (the most recent different count) - or, while no second
//              value has been seen, sub-sample j+3 assuming m1 three times;
//      wave 3: sub-sample j+2, assuming m1 twice.
// After a barrier every wave reads the four counts and commits, in order, as many sub-samples as the
// assumptions allow (always j; j+1 if a wave started where j really stopped; and so on).  Wrong
// guesses cost nothing but the energy: the result is the one the serial order defines, bit for bit
// - the committed contributions are added in sub-sample order (by the generator wave: see PixFin).
//
// For that the stream must be readable ahead of the frontier: the generator output sits in a
// ring of two blocks (SeqCtx SPEC mode); while the frontier is in one block the next one is
// already there, and when the frontier crosses into it, a fifth wave - the generator - produces the
// block after it into the slot that just became free.
// -----------------------------------------------------------------------------------------
constexpr int kSpecWaves = 4;
struct alignas(16) SpecResult { // one per wave and result set, in LDS
  double L[3]; // radiance of the sub-path below the first-bounce surface
  int meta;    // canonical doubles consumed | lobe at the first-bounce surface << 8 | rays << 16
  int pad;
};

// CROSS (round 6): the NEXT pixel's camera ray and first hit, traced ahead by a wave that has no sub-sample
// left in this pixel's last round(s) - see the kernel.  One record per tracing wave and pixel parity.
struct alignas(16) PrimRec {
  double o[3], d[3]; // the camera ray
  double t;          // its nearest hit: distance (+inf: none) ...
  uint32_t idxSign;  // ... combined primitive index << 1 | (det < epsilon), kMiss for none (packAnswer)
  uint32_t pick;     // PICKS: the pick checksum's term of this call (call 0 of the sample)
};
static_assert(sizeof(PrimRec) == 64, "PrimRec layout");
constexpr size_t kSpecPrimBytes = 2 * kSpecWaves * sizeof(PrimRec);

// The GENERATOR wave - idle except once per 312 draws - adds the committed radiance in sub-sample order and stores the
// sample: with every barrier the tracing waves tell it which of the PREVIOUS round's results were committed, and it
// folds them while the next round is being traced (three result sets in turn: a set is read one round late).  That
// takes the fold (two LDS round trips and fifteen vector instructions per committed sub-sample) off the tracing waves'
// path between two rounds: +2.8 % on Cornell together with the one-note histogram below (LAB.md, round 6).
// What the generator needs to know about a pixel whose fan-out it sums: the first-bounce surface's
// emission and diffuse colour, the pixel's index in the band, the pick checksum's term of the primary ray.
// Three records in turn for the pixels that have a fan-out: a pixel's record is read one barrier after its first
// round's, and a pixel can be over in one round - the record written two such pixels later must be another one.
struct alignas(16) PixFin {
  double e[3], dif[3];
  uint32_t pixel, pick0;
  uint32_t pad[2];
};
static_assert(sizeof(PixFin) == 64, "PixFin layout");
constexpr int kSpecResultSets = 3; // result sets (the generator reads a round's set one barrier later)
// The word the tracing waves hand the generator with every barrier: bits 0-1 the stream command (kGen*), and -
// bit 2: bits 3-12 describe a committed round; bits 3-6: ok1, ok2a, ok3, ok2b (wave 0's result always
// counts); bits 7-8: the round's result set; bit 9: first round of its pixel (take PixFin[bits 10-11]); bit 12: last.
constexpr uint32_t kFinValid = 4, kFinOk1 = 8, kFinOk2a = 16, kFinOk3 = 32, kFinOk2b = 64, kFinSetShift = 7,
                   kFinFirst = 512, kFinRecShift = 10, kFinLast = 4096;
constexpr int kSpecPixFins = 3;

__host__ __device__ inline size_t specLdsBytes(uint32_t ntri, uint32_t nmat, uint32_t nsph, bool wholeCu = true) {
  size_t n = 2 * kRingStride;                          // the ring
  n += kMtWords * sizeof(uint32_t);                    // raw generator state
  n += kSpecResultSets * kSpecWaves * sizeof(SpecResult); // results: three sets in turn
  n += 64 + kSeqCamBytes + kSpecPrimBytes;             // generator commands, the camera, the primary-ray records
  n += kSpecPixFins * sizeof(PixFin);
  n = (n + 63) & ~static_cast<size_t>(63);
  n += static_cast<size_t>(nsph) * sizeof(SphereRec);
  n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
  n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  // more than half of a CU's 160 KB: one workgroup per CU, so its four waves get a SIMD each.  (Not for the
  // two-wave form: two of its workgroups - three waves each - share a CU, which is what fills the SIMDs when there
  // are two passes per CU.)
  const size_t floor = wholeCu ? 84 * 1024 : 0;
  return n < floor ? floor : n;
}

// NW: tracing waves per pass.  4: the form described above (a CU per pass).  2 (round 6; more passes than CUs): the
// frontier plus ONE candidate - sub-sample j+1 assuming m1 - and two workgroups per CU: with between one and two
// passes per CU it fills the SIMDs that one wave per pass leaves empty (1.5 commits per round instead of 2.04; the
// LDS areas keep their four-wave layout, slots 2 and 3 are never written and never read).
template <bool PICKS, bool CROSS, int NW = kSpecWaves>
__global__ __launch_bounds__(64 * (NW + 1)) void traceSequentialSpec(
    const TraceParams p, const double *__restrict__ triGeom, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, double *__restrict__ specState, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters, uint32_t *__restrict__ picks) {
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  static_assert(NW == 2 || NW == kSpecWaves, "two or four tracing waves");
  constexpr int kBlock = 64 * (NW + 1);
  constexpr int has23 = NW > 2 ? -1 : 0; // (all-ones / zero mask: waves 2 and 3 exist)
  char *ring = reinterpret_cast<char *>(ldsRaw);
  uint32_t *mt = reinterpret_cast<uint32_t *>(ldsRaw + 2 * kRingStride);
  SpecResult *results = reinterpret_cast<SpecResult *>(mt + kMtWords);
  // (taken from ldsRaw inside the lambdas too: a captured pointer loses its LDS address space and
  // the stores turn into flat instructions with a vmcnt wait)
  constexpr size_t kGenCmdOffset = 2 * kRingStride + kMtWords * sizeof(uint32_t) + kSpecResultSets * kSpecWaves * sizeof(SpecResult);
  uint32_t *genCmd = reinterpret_cast<uint32_t *>(ldsRaw + kGenCmdOffset);
  size_t off = kGenCmdOffset + 64;
  // (the camera in LDS: as part of the kernel argument its 36 dwords were spilled to vector-register lanes
  // and read back for every pixel - see kSeqCamBytes)
  ptw_camera *camLds = reinterpret_cast<ptw_camera *>(ldsRaw + off);
  off += kSeqCamBytes;
  PrimRec *primRecs = reinterpret_cast<PrimRec *>(ldsRaw + off); // [pixel parity][wave]
  off += kSpecPrimBytes;
  PixFin *pixFin = reinterpret_cast<PixFin *>(ldsRaw + off);
  off += kSpecPixFins * sizeof(PixFin);
  off = (off + 63) & ~static_cast<size_t>(63);
  if (threadIdx.x < sizeof(ptw_camera) / sizeof(double))
    reinterpret_cast<double *>(camLds)[threadIdx.x] = reinterpret_cast<const double *>(&p.cam)[threadIdx.x];

  const int pass = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;

  using Ctx = SeqCtx<1, 1, true, true, true, 1, PICKS>;
  Ctx ctx;
  ctx.triCompactGlobal = triCompact;
  ctx.matTableGlobal = matTable;
  ctx.p = &p;
  ctx.envColour = ld3(p.env);
  asm volatile("" : "+v"(ctx.envColour.x), "+v"(ctx.envColour.y), "+v"(ctx.envColour.z));
  ctx.triGeom = triGeom;
  ctx.spheresGlobal = spheres;
  ctx.sh = nullptr;
  ctx.tid = lane; // every wave owns the whole scene: lane k holds triangle k
  ctx.stack = nullptr;
  ctx.partials = nullptr;
  ctx.cmd = nullptr;
  ctx.words = 0;
  ctx.rays = 0;
  ctx.parity = 0;
  ctx.picksOn = PICKS && picks != nullptr;
  ctx.pickReset();
  ctx.ringBase = ring;
  {
    SphereRec *ls = reinterpret_cast<SphereRec *>(ldsRaw + off);
    double *lt = reinterpret_cast<double *>(ls + p.nsph);
    double *lm = lt + static_cast<size_t>(p.ntri) * kTriCompactDoubles;
    const double *gs = reinterpret_cast<const double *>(spheres);
    double *lsd = reinterpret_cast<double *>(ls);
    for (uint32_t i = threadIdx.x; i < p.nsph * (sizeof(SphereRec) / 8); i += kBlock) lsd[i] = gs[i];
    for (uint32_t i = threadIdx.x; i < p.ntri * kTriCompactDoubles; i += kBlock) lt[i] = triCompact[i];
    for (uint32_t i = threadIdx.x; i < p.nmat * kMatDoubles; i += kBlock) lm[i] = matTable[i];
    ctx.tab.sph = ls;
    ctx.tab.tri = lt;
    ctx.tab.mat = lm;
  }
  const bool isGenerator = wave == NW; // the last wave only produces the stream
  if (!isGenerator) ctx.loadPrimitives();

  // ---- the stream: resume (or start) this pass's generator ring ----
  uint32_t *myState = mtState + static_cast<size_t>(pass) * kMtWords;
  double *myPark = specState + static_cast<size_t>(pass) * kSpecStateDoubles;
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) mt[i] = myState[i];
  unsigned fOff = 0; // frontier: ring slot (0 or kRingStride) ...
  int fQ = 0;        // ... and position in it
  if (p.firstBand) {
    __syncthreads();
    if (isGenerator) {
      specGenerateBlock(mt, ring, 0, lane);           // block 0
      specGenerateBlock(mt, ring, kRingStride, lane); // block 1 (completes block 0's overlap)
    }
  } else {
    for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
      const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
      reinterpret_cast<double *>(ring + slot * kRingStride)[k] = myPark[i];
    }
    fOff = __builtin_amdgcn_readfirstlane(static_cast<int>(myPark[2 * kRingCanonDoubles])) ? kRingStride : 0u;
    fQ = __builtin_amdgcn_readfirstlane(static_cast<int>(myPark[2 * kRingCanonDoubles + 1]));
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kMtDoubles; i += kBlock) {
      const int slot = i / kMtDoubles, q = i - slot * kMtDoubles;
      const double *cn = reinterpret_cast<const double *>(ring + slot * kRingStride);
      hemiEntry(cn[q], cn[q + 1], reinterpret_cast<double *>(ring + slot * kRingStride + kRingHemiOff) + 3 * q);
    }
  }
  __syncthreads();

  // ---- the generator wave: serves one command per workgroup barrier until told to exit ----
  if (isGenerator) {
    // the sample of the pixel whose rounds are being reported, summed in sub-sample order
    d3 acc = mk(0, 0, 0), finE = mk(0, 0, 0), finD = mk(0, 0, 0);
    uint32_t finPixel = 0, finPick = 0, finPickBase = 1;
    unsigned long long finRays = 0; // rays of the committed sub-samples (the tracing waves count the primary rays)
    double *myStageG = stage + static_cast<size_t>(pass) * p.pixCount * 3;
    for (unsigned k = 0;; ++k) {
      ldsBarrier();
      const uint32_t word = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(genCmd[k & 1])));
      const uint32_t cmd = word & 3u;
      if (word & kFinValid) {
        if (word & kFinFirst) {
          const PixFin &f = pixFin[(word >> kFinRecShift) & 3u];
          finE = mk(f.e[0], f.e[1], f.e[2]), finD = mk(f.dif[0], f.dif[1], f.dif[2]);
          finPixel = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(f.pixel)));
          finPick = f.pick0, finPickBase = 1;
          acc = mk(0, 0, 0);
        }
        const SpecResult *slot = results + ((word >> kFinSetShift) & 3u) * kSpecWaves;
        auto add = [&](int wv) {
          const SpecResult &r = slot[wv];
          const d3 child = mk(r.L[0], r.L[1], r.L[2]);
          const int meta = __builtin_amdgcn_readfirstlane(r.meta);
          acc = acc + ((meta & 0x100) ? finE + child : finE + finD * child);
          finRays += static_cast<unsigned>(meta >> 16);
          if (PICKS) {
            const uint32_t pw = static_cast<uint32_t>(r.pad);
            finPick += finPickBase * (pw & 0xffffu) + (pw >> 16);
            finPickBase += static_cast<uint32_t>(meta) >> 16;
          }
        };
        add(0);
        if (word & kFinOk1) add(1);
        if (word & kFinOk2a) add(2);
        if (word & kFinOk3) add(3);
        if (word & kFinOk2b) add(2);
        if (word & kFinLast) {
          const d3 L = acc * p.invFirstBounce;
          if (lane == 0) {
            myStageG[finPixel * 3 + 0] = L.x;
            myStageG[finPixel * 3 + 1] = L.y;
            myStageG[finPixel * 3 + 2] = L.z;
            if (PICKS && picks) picks[static_cast<size_t>(pass) * p.npix + p.pixBegin + finPixel] = finPick;
          }
        }
      }
      if (cmd == kGenExit) break;
      if (cmd == kGenSlot0) specGenerateBlock(mt, ring, 0, lane);
      if (cmd == kGenSlot1) specGenerateBlock(mt, ring, kRingStride, lane);
    }
    if (lane == 0 && rayCounters) atomicAdd(rayCounters + pass, finRays);
  } else {
  // Stream bookkeeping of the tracing waves (identical in all of them).  When the frontier
  // enters the other slot, the slot it left is handed to the generator wave with the next
  // barrier (genState 1 -> 2); the block is complete once the barrier after that has been passed
  // (2 -> 0), because the generator arrives there only when it is done.  The tracing waves
  // read at most `ahead` draws beyond the frontier, so they only have to wait for an
  // outstanding block when the frontier comes that close to the end of its slot.
  unsigned barriers = 0;
  int genState = 0;
  unsigned genSlot = 0;
  const int ahead = 12 * (p.maxDepth > 0 ? p.maxDepth : 1) + 8;
  uint32_t finInfo = 0; // the committed round not yet reported to the generator (kFin* bits; 0: none)
  auto roundBarrier = [&](uint32_t exitCmd) {
    if (threadIdx.x == 0)
      reinterpret_cast<uint32_t *>(ldsRaw + kGenCmdOffset)[barriers & 1] =
          (exitCmd ? exitCmd : (genState == 1 ? (genSlot ? kGenSlot1 : kGenSlot0) : kGenNone)) | finInfo;
    finInfo = 0;
    ldsBarrier();
    ++barriers;
    genState = genState == 1 ? 2 : 0;
  };
  auto ensureAhead = [&]() {
    while (genState != 0 && fQ + ahead >= kMtDoubles) roundBarrier(0);
  };
  auto advanceFrontier = [&](int n) { // n < kMtDoubles
    const int np = fQ + n;
    if (np >= kMtDoubles) {
      genSlot = fOff; // the slot left behind takes the block after the next
      genState = 1;
      fQ = np - kMtDoubles;
      fOff ^= kRingStride;
    } else {
      fQ = np;
    }
  };

  const int w = p.width;
  const bool lens = uniformBool(camLds->aperture_radius != 0);
  const int nSub = p.fbU * p.fbV;
  const int vShift = p.fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(p.fbV)) : 0;
  // Per-round constants in vector registers: as kernel arguments they sit in a 16-register
  // scalar tuple that does not survive the rounds and would be re-read from its spill lanes
  // (eighteen v_readlane) for every sub-sample.
  double invU = p.invU, invV = p.invV;
  asm volatile("" : "+v"(invU), "+v"(invV));
  const bool fastFan = (p.uPow2 & p.vPow2) != 0;
  const int vMask = p.fbV - 1;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
  unsigned long long raysTotal = 0;
  // The guesses: m1 = the most frequent count of draws a sub-sample has consumed so far in this
  // pass (3 per level reached), m2 = the second most frequent (m2 == m1 while only one value has
  // been seen).  `hist` counts them in 6-bit fields, halved when a field passes 31.
  int m1 = 3 * (p.maxDepth > 0 ? p.maxDepth : 1), m2 = m1;
  unsigned long long hist = 0;
  int parity = 0;
#if PTW_PROFILE_PHASES
  unsigned long long stRounds = 0, stCommits = 0, stWork = 0, stWait = 0, stCommit = 0, stPrimary = 0;
  unsigned long long stOk1 = 0, stOk2a = 0, stOk3 = 0, stOk2b = 0, stIdle = 0;
  unsigned long long stPrimTried = 0, stPrimTaken = 0, stCommitHist[5] = {0, 0, 0, 0, 0};
  const unsigned long long stT0 = __builtin_amdgcn_s_memtime();
#endif

  // CROSS: the wave (1..3) whose record holds THIS pixel's camera ray and first hit, traced during the previous
  // pixel's last round at the stream position that pixel really ended at; 0 = none (trace it now)
  int primWave = 0;
  int finRec = 0; // the PixFin record of the latest pixel with a fan-out
  int px = static_cast<int>(p.pixBegin % static_cast<uint32_t>(w));
  int py = static_cast<int>(p.pixBegin / static_cast<uint32_t>(w));
  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    // ---- every wave: camera ray and first hit at the frontier (redundant, in parallel) ----
    PTW_T(tP0);
    ensureAhead();
    const int camDraws = lens ? 4 : 2;
    d3 o, d;
    int sampleDraws = camDraws;
    uint32_t pickSum = 0; // the pick checksum's term of the primary ray (the generator wave adds the fan-out's)
    d3 L = mk(0, 0, 0);
    bool traced = false, genSums = false;
    HitKey k0;
    k0.t = kInf, k0.idx = kMiss, k0.det = 0;
#if PTW_PROFILE_PHASES
    stPrimTaken += primWave != 0;
#endif
    if (CROSS && primWave != 0) {
      // traced ahead (only ever with maxDepth > 0 and a fan-out: see below): one LDS round trip
      const PrimRec &r = primRecs[(i & 1u) * kSpecWaves + primWave];
      o = mk(r.o[0], r.o[1], r.o[2]), d = mk(r.d[0], r.d[1], r.d[2]);
      k0.t = r.t;
      const uint32_t pw = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(r.idxSign)));
      k0.idx = pw == kMiss ? kMiss : (pw >> 1);
      k0.det = (pw & 1u) ? -1.0 : 1.0; // (only its sign test is ever used)
      raysTotal++;
      if (PICKS) pickSum = r.pick;
      if (uniformBool(k0.idx == kMiss)) {
        L = ld3(p.env);
      } else {
        traced = true;
      }
    } else {
      ctx.setStream(fOff, fQ);
      double r0, r1, r2 = 0, r3 = 0;
      if (lens) {
        ctx.draw4(r0, r1, r2, r3);
      } else {
        r0 = ctx.draw();
        r1 = ctx.draw();
      }
      cameraRay(*camLds, px, py, r0, r1, r2, r3, o, d);
      ctx.pickReset();
      if (p.maxDepth > 0) {
        k0 = ctx.intersect(o, d);
        raysTotal++;
        if (PICKS) pickSum = ctx.pickS2; // (the primary ray is call 0)
        if (uniformBool(k0.idx == kMiss)) {
          L = ld3(p.env);
        } else {
          traced = true;
        }
      }
    }
    primWave = 0;
    // the next pixel (row-major; `pix` itself is only used for the optional per-sample outputs)
    int pxNext = px + 1, pyNext = py;
    if (pxNext == w) pxNext = 0, ++pyNext;
    advanceFrontier(camDraws);
#if PTW_PROFILE_PHASES
    stPrimary += __builtin_amdgcn_s_memtime() - tP0;
#endif
    if (traced) {
      const Surface first = ctx.surfaceAt(k0, o, d);
      if (p.preview) {
        L = first.diffuse; // Scene.cpp:137-138
      } else {
        // (`result`: what the tracing waves themselves would store - nothing is ever added to it here, the
        // generator wave holds the sum; it stays zero, and only a fan-out without sub-samples would store it)
        d3 result = mk(0, 0, 0);
        int j = 0;
        if (nSub > 0) {
          // the generator wave sums this pixel's fan-out: what it needs, before the first round's barrier
          genSums = true;
          finRec = finRec == kSpecPixFins - 1 ? 0 : finRec + 1;
          if (threadIdx.x == 0) {
            PixFin &f = pixFin[finRec];
            f.e[0] = first.emission.x, f.e[1] = first.emission.y, f.e[2] = first.emission.z;
            f.dif[0] = first.diffuse.x, f.dif[1] = first.diffuse.y, f.dif[2] = first.diffuse.z;
            f.pixel = i, f.pick0 = pickSum;
          }
        }
        if (hist & 0x0820820820820820ull) hist = (hist >> 1) & 0x07df7df7df7df7dfull; // (a field never passes 63)
        if (hist != 0) { // refresh the guesses once per sample
          int best = 0, bestN = -1, second = 0, secondN = 0;
#pragma unroll
          for (int f = 1; f <= 9; ++f) {
            const int n = static_cast<int>(hist >> (6 * f)) & 63;
            const bool top = n > bestN;
            const bool sec = !top & (n > secondN);
            second = top ? best : (sec ? f : second);
            secondN = top ? bestN : (sec ? n : secondN);
            best = top ? f : best;
            bestN = top ? n : bestN;
          }
          m1 = 3 * best;
          m2 = secondN > 0 ? 3 * second : m1;
        }
        // CROSS: what the pixel's last round looked like (read after the loop)
        int lastJ = 0, lastCur = 0, lastD1 = 0, lastD2 = 0, lastD3 = 0, lastOne = 0;
        while (j < nSub) {
          ensureAhead();
          // ---- this wave's assignment: sub-sample j + ioff, stream position frontier + delta ----
          const bool oneMode = m2 == m1;
          int ioff = 0, delta = 0;
          if (wave == 1) ioff = 1, delta = m1;
          if (wave == 2) ioff = oneMode ? 3 : 1, delta = oneMode ? 3 * m1 : m2;
          if (wave == 3) ioff = 2, delta = 2 * m1;
          const int myIdx = j + ioff;
          PTW_T(tW0);
          SpecResult mine;
          mine.L[0] = mine.L[1] = mine.L[2] = 0;
          mine.meta = 0, mine.pad = 0;
          if (myIdx < nSub) {
            const int np = fQ + delta; // delta < kMtDoubles
            const bool wrap = np >= kMtDoubles;
            ctx.setStream(wrap ? fOff ^ kRingStride : fOff, wrap ? np - kMtDoubles : np);
            ctx.words = 0;
            ctx.rays = 0;
            ctx.pickReset();
            // sub-sample index -> stratum (uS, vS) -> stratified (u, v); ONE decision for the
            // usual power-of-two fan-outs (shift / mask / multiply), the general case apart
            double xu, xv, pd;
            ctx.draw3(xu, xv, pd);
            double u, v;
            if (fastFan) {
              const int uS = myIdx >> vShift, vS = myIdx & vMask;
              u = (static_cast<double>(uS) + xu) * invU;
              v = (static_cast<double>(vS) + xv) * invV;
            } else {
              const int uS = myIdx / p.fbV, vS = myIdx - uS * p.fbV;
              const double ur = static_cast<double>(uS) + xu, vr = static_cast<double>(vS) + xv;
              u = p.uPow2 ? ur * invU : ur / static_cast<double>(p.fbU);
              v = p.vPow2 ? vr * invV : vr / static_cast<double>(p.fbV);
            }
            d3 nd;
            const bool refl = scatter(ctx, first, d, u, v, pd, nd);
            const d3 child = ctx.chainHot(p, first.pos, nd);
            mine.L[0] = child.x, mine.L[1] = child.y, mine.L[2] = child.z;
            mine.meta = static_cast<int>(ctx.words >> 1) | (refl ? 0x100 : 0) |
                        (static_cast<int>(ctx.rays) << 16);
            if (PICKS) mine.pad = static_cast<int>(ctx.pickS1 | (ctx.pickS2 << 16)); // (<= 9 calls of <= 127 primitives)
          } else if (CROSS && myIdx == nSub && i + 1 < p.pixCount) {
            // No sub-sample left for this wave, and its assignment is exactly "the sub-sample after the last
            // one": that is the NEXT pixel's camera ray.  Trace it and its first hit at the stream position the
            // assignment implies (this pixel ends `delta` draws after the round's frontier) into this wave's
            // record; it is taken if the pixel really ends there (checked after the loop) - the same commit rule
            // as for a sub-sample, so the value is the serial one bit for bit, and a wrong guess costs energy only.
            const int np = fQ + delta; // delta + 4 camera draws stay within `ahead`
            const bool wrap = np >= kMtDoubles;
            ctx.setStream(wrap ? fOff ^ kRingStride : fOff, wrap ? np - kMtDoubles : np);
            double r0, r1, r2 = 0, r3 = 0;
            if (lens) {
              ctx.draw4(r0, r1, r2, r3);
            } else {
              r0 = ctx.draw();
              r1 = ctx.draw();
            }
            d3 no, nd;
            cameraRay(*camLds, pxNext, pyNext, r0, r1, r2, r3, no, nd);
            ctx.pickReset();
            const HitKey nk = ctx.intersect(no, nd);
#if PTW_PROFILE_PHASES
            stPrimTried++;
#endif
            if (lane == 0) {
              PrimRec &r = primRecs[((i + 1) & 1u) * kSpecWaves + wave];
              r.o[0] = no.x, r.o[1] = no.y, r.o[2] = no.z;
              r.d[0] = nd.x, r.d[1] = nd.y, r.d[2] = nd.z;
              r.t = nk.t;
              r.idxSign = packAnswer(nk);
              r.pick = PICKS ? ctx.pickS2 : 0u;
            }
          }
          SpecResult *slot = results + parity * kSpecWaves;
          if (lane == 0) slot[wave] = mine;
          PTW_T(tW1);
          roundBarrier(0);
          PTW_T(tW2);
          // ---- commit (the scalar part identical in every wave) ----
          const int metaV = slot[lane & 3].meta;
          const int meta0 = __builtin_amdgcn_readlane(metaV, 0), meta1 = __builtin_amdgcn_readlane(metaV, 1);
          const int meta2 = __builtin_amdgcn_readlane(metaV, 2), meta3 = __builtin_amdgcn_readlane(metaV, 3);
          const int c0 = meta0 & 0xff, c1 = meta1 & 0xff, c2 = meta2 & 0xff, c3 = meta3 & 0xff;
          // The assignments this round was made with, and which of them held - as all-ones /
          // zero integer masks in scalar registers (conditions kept as C++ bools become lane masks
          // that take a trip through a vector register per use).
          auto eq = [](int a, int b) { return ((a ^ b) - 1) >> 31; };  // a, b >= 0: -1 if equal
          auto lt = [](int a, int b) { return (a - b) >> 31; };        // -1 if a < b
          const int d1 = m1, d2 = oneMode ? 3 * m1 : m2, d3v = 2 * m1;
          const int w2Second = oneMode ? 0 : -1; // wave 2 ran sub-sample j+1 (else j+3)
          const int ok1 = lt(j + 1, nSub) & eq(d1, c0);
          const int ok2a = has23 & lt(j + 1, nSub) & ~ok1 & w2Second & eq(d2, c0);
          const int cur1 = c0 + (c1 & ok1) + (c2 & ok2a);
          const int two = ok1 | ok2a;
          const int ok3 = has23 & two & lt(j + 2, nSub) & eq(d3v, cur1);
          const int cur2 = cur1 + (c3 & ok3);
          const int ok2b = ok3 & lt(j + 3, nSub) & ~w2Second & eq(d2, cur2);
          const int cur = cur2 + (c2 & ok2b);
          const int nIdx = 1 - two - ok3 - ok2b;
          // Histogram of the counts (6-bit fields indexed by count / 3): the FRONTIER's own count only - every round
          // has exactly one, it is as good a sample of the distribution as the counts of all committed sub-samples,
          // and four notes fewer are forty scalar instructions fewer on every wave's path between two rounds.
          hist += 1ull << (6 * ((c0 * 11) >> 5));
          // What the generator wave needs to fold this round's committed results (it counts their rays and picks too)
          finInfo = kFinValid | (static_cast<uint32_t>(ok1) & kFinOk1) | (static_cast<uint32_t>(ok2a) & kFinOk2a) |
                    (static_cast<uint32_t>(ok3) & kFinOk3) | (static_cast<uint32_t>(ok2b) & kFinOk2b) |
                    (static_cast<uint32_t>(parity) << kFinSetShift) | (j == 0 ? kFinFirst : 0u) |
                    (static_cast<uint32_t>(finRec) << kFinRecShift) | (j + nIdx >= nSub ? kFinLast : 0u);
          if (CROSS) lastJ = j, lastCur = cur, lastD1 = d1, lastD2 = d2, lastD3 = d3v, lastOne = oneMode ? -1 : 0;
          j += nIdx;
          sampleDraws += cur;
          parity = parity == kSpecResultSets - 1 ? 0 : parity + 1;
          advanceFrontier(cur);
#if PTW_PROFILE_PHASES
          stRounds++, stCommits += nIdx;
          stCommitHist[nIdx]++;
          stOk1 -= ok1, stOk2a -= ok2a, stOk3 -= ok3, stOk2b -= ok2b, stIdle += !(myIdx < nSub);
          stWork += tW1 - tW0, stWait += tW2 - tW1, stCommit += __builtin_amdgcn_s_memtime() - tW2;
#endif
        }
        // (Three multiplications per pixel that no pixel with a fan-out uses - and that stay: where the code of the
        // round loop lands in memory decides +-1.5 % for waves that have their SIMD to themselves, every taken branch
        // exposes the fetch of its target, and this form's layout is the fastest of the equivalent ones measured:
        // LAB.md round 6, profiles/r06o_*, r06r_*.)
        L = result * p.invFirstBounce;
        if (CROSS && i + 1 < p.pixCount) {
          // Did a wave trace the next pixel's primary ray in the last round, and from where the pixel ended?
          // Wave v did if its assignment was sub-sample index nSub (lastJ + ioff == nSub); it is right if its
          // delta is what the round really consumed.  (The round that ends the pixel commits every remaining
          // sub-sample, so `lastCur` is the distance from that round's frontier to the pixel's end.)
          auto eq = [](int a, int b) { return ((a ^ b) - 1) >> 31; }; // a, b >= 0: -1 if equal
          const int ioff2 = lastOne ? 3 : 1;
          const int p1 = eq(lastJ + 1, nSub) & eq(lastD1, lastCur);
          const int p2 = has23 & eq(lastJ + ioff2, nSub) & eq(lastD2, lastCur);
          const int p3 = has23 & eq(lastJ + 2, nSub) & eq(lastD3, lastCur);
          primWave = p1 ? 1 : (p2 ? 2 : (p3 ? 3 : 0));
        }
      }
    }
    if (threadIdx.x == 0) {
      if (!genSums) { // (else the generator wave stores the sample and its picks)
        myStage[i * 3 + 0] = L.x;
        myStage[i * 3 + 1] = L.y;
        myStage[i * 3 + 2] = L.z;
        if (PICKS && picks) picks[static_cast<size_t>(pass) * p.npix + pix] = pickSum;
      }
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = 2u * static_cast<unsigned>(sampleDraws);
    }
    px = pxNext, py = pyNext;
  }

  while (genState != 0) roundBarrier(0); // an outstanding block must be in the ring that gets parked
  roundBarrier(kGenExit);
#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const double n = static_cast<double>(p.pixCount);
    printf("SPEC wave %d: cycles/sample=%.0f rounds/sample=%.2f commits/round=%.2f primary=%.0f work=%.0f "
           "wait=%.0f commit+advance=%.0f (per sample)\n",
           wave, (__builtin_amdgcn_s_memtime() - stT0) / n, stRounds / n,
           static_cast<double>(stCommits) / stRounds, stPrimary / n, stWork / n, stWait / n, stCommit / n);
    printf("SPEC wave %d: per round ok1=%.3f ok2a=%.3f ok3=%.3f ok2b=%.3f idle=%.3f\n", wave,
           (double)stOk1 / stRounds, (double)stOk2a / stRounds, (double)stOk3 / stRounds,
           (double)stOk2b / stRounds, (double)stIdle / stRounds);
    printf("SPEC wave %d: rounds by sub-samples committed 1/2/3/4 = %.3f / %.3f / %.3f / %.3f of the rounds; next pixel's "
           "primary ray traced ahead by this wave in %.3f rounds per pixel, pixels that started from such a record %.3f\n",
           wave, (double)stCommitHist[1] / stRounds, (double)stCommitHist[2] / stRounds, (double)stCommitHist[3] / stRounds,
           (double)stCommitHist[4] / stRounds, stPrimTried / n, stPrimTaken / n);
  }
#endif
  if (threadIdx.x == 0) {
    myPark[2 * kRingCanonDoubles] = fOff ? 1.0 : 0.0;
    myPark[2 * kRingCanonDoubles + 1] = static_cast<double>(fQ);
  }
  if (threadIdx.x == 0 && rayCounters) atomicAdd(rayCounters + pass, raysTotal); // (the primary rays; the generator adds the rest)
  } // tracing waves
  // ---- park the stream for the next band ----
  __syncthreads();
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = mt[i];
  for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
    const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
    myPark[i] = reinterpret_cast<const double *>(ring + slot * kRingStride)[k];
  }
}

} // namespace

// the scenes the register-resident kernels handle (the REG variant of the single-wave kernel, traceSequentialSpec)
bool specApplies(const TraceParams &p) {
  return p.ntri <= 64 && p.nsph <= 64 && p.nsph + p.ntri <= 127 && p.maxDepth <= 9 &&
         seqLdsBytes(1, p.maxDepth, true, p.ntri, p.nmat, p.nsph) <= kLdsTableBudget;
}

hipError_t launchSeqSpec(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  // (LaunchHints::seqSmallKernel == 3: round 5's form, without the next pixel's primary ray traced ahead; 4: the
  // two-wave form, two workgroups per CU)
  const bool ahead = hints.seqSmallKernel != 3, two = hints.seqSmallKernel == 4;
  setVariant(two ? "traceSequentialSpec<2 waves>" : ahead ? "traceSequentialSpec" : "traceSequentialSpec<no cross-pixel candidate>");
  if (hints.dryRun) return hipSuccess;
  const size_t lds = specLdsBytes(p.ntri, p.nmat, p.nsph, !two);
  auto kernel = two ? (b.picks ? traceSequentialSpec<true, true, 2> : traceSequentialSpec<false, true, 2>)
                    : b.picks ? (ahead ? traceSequentialSpec<true, true> : traceSequentialSpec<true, false>)
                              : (ahead ? traceSequentialSpec<false, true> : traceSequentialSpec<false, false>);
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(p.npass), dim3(64 * ((two ? 2 : kSpecWaves) + 1)), lds, stream, p,
                     b.triGeom, b.spheres, b.triCompact, b.matTable, b.mtState, b.specState, b.stage,
                     b.words, b.rays, b.picks);
  return hipGetLastError();
}

} // namespace ptw

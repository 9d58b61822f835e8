// ptw_seq_kernel.h - traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS> and its launcher templates; the
// translation units seq_single.hip / seq_worker.hip / seq_worker2.hip each instantiate one family of it.
// Internal to csrc/.
#pragma once
#include "ptw_launch.h"
#include "ptw_seq_ctx.h"

#include <algorithm>
#include <cstdio>

namespace ptw {
using namespace ptwd;
namespace {

template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, int MASTERS = 1, bool PICKS = true, bool PRE = false,
          bool UNIT = false>
__global__ __launch_bounds__(WAVES == 1 ? 64 : 64 * (WAVES + MASTERS)) void traceSequential(
    const TraceParams p, const double *__restrict__ triGeom,
    const TriShade *__restrict__ triShade, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, uint32_t *__restrict__ mtPos, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters, uint32_t *__restrict__ picks,
    const float *__restrict__ triPacked) {
  using Ctx = SeqCtx<SLOTS, WAVES, LDS_TABLES, REG, false, MASTERS, PICKS, PRE, UNIT>;
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  (void)triShade;
  const int depthSlots = p.maxDepth > 0 ? p.maxDepth : 1;
  // the master wave(s): wave m < MASTERS runs pass MASTERS * blockIdx.x + m on its own generator
  const int hwWave = static_cast<int>(threadIdx.x >> 6);
  const int master = MASTERS == 1 ? 0 : (hwWave < MASTERS ? hwWave : 0);
  SeqShared &sh = reinterpret_cast<SeqShared *>(ldsRaw)[master];
  Level *stacks = reinterpret_cast<Level *>(ldsRaw + MASTERS * sizeof(SeqShared));
  PartialHit *partials = reinterpret_cast<PartialHit *>(
      ldsRaw + ((MASTERS * sizeof(SeqShared) + static_cast<size_t>(WAVES) * (p.maxDepth > 0 ? p.maxDepth : 1) * sizeof(Level) + 15) &
                ~static_cast<size_t>(15)));
  // (the answers are read with ds_read_b128: their offset is rounded up to 16 bytes - with seven
  // worker waves and an odd maxDepth the stacks end on 8 mod 16)
  const size_t partialsOff = (MASTERS * sizeof(SeqShared) + static_cast<size_t>(WAVES) * depthSlots * sizeof(Level) + 15) &
                             ~static_cast<size_t>(15);
  size_t off = partialsOff + 2 * MASTERS * static_cast<size_t>(WAVES) * sizeof(PartialHit) + kSeqCmdBytes;
  off = (off + 63) & ~static_cast<size_t>(63);
  ptw_camera *camLds = reinterpret_cast<ptw_camera *>(ldsRaw + off);
  if (WAVES > 1) off += kSeqCamBytes;

  const int pass = blockIdx.x * MASTERS + master;
  const bool hasPass = MASTERS == 1 || static_cast<uint32_t>(pass) < p.npass; // (odd pass count)
  Ctx ctx;
  ctx.triCompactGlobal = triCompact;
  ctx.matTableGlobal = matTable;
  ctx.p = &p;
  ctx.envColour = ld3(p.env);
  asm volatile("" : "+v"(ctx.envColour.x), "+v"(ctx.envColour.y), "+v"(ctx.envColour.z));
  ctx.triGeom = triGeom;
  ctx.triPacked = triPacked;
  ctx.spheresGlobal = spheres;
  ctx.sh = &sh;
  constexpr int kBlock = Ctx::kBlock;
  const bool isWorker = WAVES > 1 && hwWave >= MASTERS;
  const int lane = threadIdx.x & 63;
  // Worker waves in `tid` order: those on a SIMD of their own pair first, those that share a SIMD
  // with a master wave (waves go to the four SIMDs round robin: wave 4 sits with wave 0, wave 5
  // with wave 1) last - they get the scene's empty slots (SeqCtx::slotTriangle), because the master
  // beside them uses the search time for its look-ahead.
  int workerRank = hwWave - MASTERS;
  if (WAVES > 1 && isWorker) {
    const int firstShared = 4 - MASTERS, nShared = MASTERS; // worker indices of waves 4 .. 3 + MASTERS
    if (workerRank >= firstShared + nShared) workerRank -= nShared;
    else if (workerRank >= firstShared) workerRank += WAVES - firstShared - nShared;
  }
  ctx.tid = WAVES == 1 ? threadIdx.x : (isWorker ? workerRank * 64 + lane : lane);
  ctx.unitBase = 0, ctx.myUnits = 0;
  if (WAVES > 1 && isWorker) {
    // ranks [0, nA): the workers that share a SIMD with another worker; [nA, WAVES): beside a master
    constexpr int nA = WAVES - Ctx::kSideB;
    const bool sideB = workerRank >= nA;
    // ranks [0, nA / 2): the OLDER wave of each worker pair (lower hardware wave index), [nA / 2, nA):
    // the younger one
    static_assert(nA % 2 == 0, "the worker-only SIMDs carry two workers each");
    const bool young = !sideB && workerRank >= nA / 2;
    ctx.myUnits = __builtin_amdgcn_readfirstlane(sideB ? p.seqUnitsB : (young ? p.seqUnitsY : p.seqUnitsA));
    ctx.unitBase = __builtin_amdgcn_readfirstlane(
        sideB   ? (nA / 2) * (p.seqUnitsA + p.seqUnitsY) + (workerRank - nA) * p.seqUnitsB
        : young ? (nA / 2) * p.seqUnitsA + (workerRank - nA / 2) * p.seqUnitsY
                : workerRank * p.seqUnitsA);
  }
  // only the master waves use a radiance stack: one each
  ctx.stack = stacks + master * depthSlots;
  // [MASTERS][WAVES] partial results (in an area of twice that: seqLdsBytes), then the commands (128 B each)
  // and the worker waves' atomic slots
  ctx.allCmds = reinterpret_cast<SeqCommand *>(partials + 2 * MASTERS * WAVES);
  ctx.partials = isWorker ? partials : partials + master * WAVES;
  ctx.cmd = ctx.allCmds + master;
  ctx.minSlot = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(ctx.allCmds) + kSeqMinSlotOffset) +
                2 * (isWorker ? workerRank : 0);
  ctx.masterIndex = master;
  ctx.picksOn = PICKS && picks != nullptr;
  ctx.pickReset();
  static_assert(2 * sizeof(SeqCommand) <= kSeqMinSlotOffset && WAVES <= 8 &&
                    kSeqMinSlotOffset + 8 * 2 * sizeof(unsigned long long) <= kSeqCmdBytes, "commands and slots fit");
  ctx.tick = 0;
  ctx.laArmed = false;
  ctx.laPos = -1;
  ctx.laMisses = 0;
  ctx.laTick = 0;
  ctx.pendKind = 0;
  ctx.words = 0;
  ctx.rays = 0;
  ctx.parity = 0;
  if (LDS_TABLES) {
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
  } else {
    ctx.tab.sph = spheres;
    ctx.tab.tri = triCompact;
    ctx.tab.mat = matTable;
  }
  if (WAVES == 1 || isWorker) {
    ctx.loadPrimitives();
  } else {
    ctx.hasSphere = false;
  }
  ctx.cam = &p.cam;
  if (WAVES > 1) {
    if (threadIdx.x < sizeof(ptw_camera) / sizeof(double))
      reinterpret_cast<double *>(camLds)[threadIdx.x] = reinterpret_cast<const double *>(&p.cam)[threadIdx.x];
    ctx.cam = camLds; // (visible after the barrier below)
  }

  // resume this pass's generator
  uint32_t *myState = mtState + static_cast<size_t>(hasPass ? pass : 0) * kMtWords;
  if (MASTERS == 1) {
    for (int i = threadIdx.x; i < kMtWords; i += kBlock) sh.mt[i] = myState[i];
    ctx.pos = __builtin_amdgcn_readfirstlane(static_cast<int>(mtPos[pass]));
    __syncthreads();
    if (ctx.pos < kMtDoubles) ctx.rebuildCanon();
  } else {
    ctx.pos = kMtDoubles;
    if (!isWorker) {
      if (hasPass) {
        for (int i = lane; i < kMtWords; i += 64) sh.mt[i] = myState[i];
        ctx.pos = __builtin_amdgcn_readfirstlane(static_cast<int>(mtPos[pass]));
        waveSync();
        if (ctx.pos < kMtDoubles) ctx.rebuildCanonWave();
      }
      // "live" / "no rays as of barrier 0"
      if (lane == 0) ctx.cmd->op = hasPass ? kCmdLive : 0u;
    }
    __syncthreads();
  }

  if (isWorker) {
    ctx.workerLoop();
  } else if (!hasPass) {
    ctx.stopWorkers();
  } else {
  if (MASTERS == 2 && master == 1) {
    ldsBarrier(); // the second master runs one barrier behind the first
    ctx.tick = 1;
  }
  // (raising the master waves' priority over the worker that shares their SIMD - s_setprio 1..3 -
  // measured no difference on suzanne and ce: profiles/r03e_master_priority_and_balance.txt)
  const int w = p.width;
  const bool lens = WAVES > 1 ? uniformBool(ctx.cam->aperture_radius != 0) : p.cam.aperture_radius != 0;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
#if PTW_PROFILE_PHASES
  for (int i = 0; i < 12; ++i) ctx.prof[i] = 0;
  for (int i = 0; i < 6; ++i) ctx.mprof[i] = 0;
  ctx.g00 = ctx.g01 = ctx.g10 = ctx.g11 = ctx.g20 = ctx.g21 = 0;
  ctx.n00 = ctx.n01 = ctx.n10 = ctx.n11 = ctx.n20 = ctx.n21 = 0;
  ctx.lastExit = 0, ctx.rayKind = 2, ctx.lastKind = 0, ctx.lastMiss = 0;
  const unsigned long long tStart = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    const int px = static_cast<int>(pix % static_cast<uint32_t>(w));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(w));
    ctx.words = 0;
    ctx.pickReset();
    const unsigned long long tC0 = ctx.now();
    double r0, r1, r2 = 0, r3 = 0;
    if (lens) {
      ctx.draw4(r0, r1, r2, r3);
    } else {
      r0 = ctx.draw();
      r1 = ctx.draw();
    }
    d3 o, d;
    cameraRay<MASTERS == 2>(*ctx.cam, px, py, r0, r1, r2, r3, o, d);
    ctx.acc(10, tC0, d.x);
    const d3 L = radiance0(ctx, p, triShade, spheres, o, d);
    if (lane == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = ctx.words;
      if (PICKS && picks) picks[static_cast<size_t>(pass) * p.npix + pix] = ctx.pickS2;
    }
  }

#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const unsigned long long tEnd = __builtin_amdgcn_s_memtime();
    const double r = static_cast<double>(ctx.rays);
    printf("PHASES rays=%llu total/ray=%.0f tests=%.0f reduce=%.0f surface=%.0f (lds1=%.0f) scatter|arm=%.0f "
           "xwave=%.0f first=%.0f fold|accum=%.0f lastE=%.0f regen=%.0f camera=%.0f hotlevel=%.0f other=%.0f\n",
           ctx.rays, (tEnd - tStart) / r, ctx.prof[0] / r, ctx.prof[1] / r, ctx.prof[2] / r,
           ctx.prof[4] / r, ctx.prof[3] / r, ctx.prof[5] / r, ctx.prof[6] / r, ctx.prof[7] / r,
           ctx.prof[8] / r, ctx.prof[9] / r, ctx.prof[10] / r, ctx.prof[11] / r,
           ((tEnd - tStart) - ctx.prof[0] - ctx.prof[1] - ctx.prof[2] - ctx.prof[3] - ctx.prof[5] -
            ctx.prof[6] - ctx.prof[7] - ctx.prof[8] - ctx.prof[10] - ctx.prof[11]) / r);
    if (WAVES > 1)
      printf("MASTER per ray, inside intersect(): publish=%.0f waitB1=%.0f shadow(flush+lookahead)=%.0f waitB2=%.0f pick=%.0f; "
             "outside intersect()=%.0f\n",
             ctx.mprof[0] / r, ctx.mprof[1] / r, ctx.mprof[2] / r, ctx.mprof[3] / r, ctx.mprof[4] / r,
             ((tEnd - tStart) - ctx.prof[5]) / r);
    if (WAVES > 1) {
      const double px = static_cast<double>(p.pixCount);
      auto avg = [](unsigned long long sum, unsigned long long n) { return n ? static_cast<double>(sum) / n : 0.0; };
      printf("MASTER cycles from an answer to the next published ray (and answers per sample): primary hit %.0f (%.2f) miss %.0f "
             "(%.2f) | first ray of a sub-sample hit %.0f (%.2f) miss %.0f (%.2f) | deeper hit %.0f (%.2f) miss %.0f (%.2f)\n",
             avg(ctx.g00, ctx.n00), ctx.n00 / px, avg(ctx.g01, ctx.n01), ctx.n01 / px, avg(ctx.g10, ctx.n10), ctx.n10 / px,
             avg(ctx.g11, ctx.n11), ctx.n11 / px, avg(ctx.g20, ctx.n20), ctx.n20 / px, avg(ctx.g21, ctx.n21), ctx.n21 / px);
    }
  }
#endif
  ctx.stopWorkers();
  } // master
  // park the generator for the next band
  if (MASTERS == 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = sh.mt[i];
    if (threadIdx.x == 0) {
      mtPos[pass] = static_cast<uint32_t>(ctx.pos); // thread 0 belongs to the master wave
      if (rayCounters) rayCounters[pass] += ctx.rays;
    }
  } else if (!isWorker && hasPass) {
    waveSync();
    for (int i = lane; i < kMtWords; i += 64) myState[i] = sh.mt[i];
    if (lane == 0) {
      mtPos[pass] = static_cast<uint32_t>(ctx.pos);
      if (rayCounters) rayCounters[pass] += ctx.rays;
    }
  }
}

// The units of 64 triangles per worker wave: older / younger wave of a worker pair, master-side wave.
// Equal shares (seqUnitSplit); two masters, scenes from 12 units on: shares by the wave's place
// (seqUnitSplitByPlace) - below 31 units only inside the instantiation the equal shares choose (a larger one
// costs more than the shares bring: profiles/r06w_*).  LaunchHints::seqUnits sets them outright (tests, A/B
// runs; what does not fit the waves' shares is streamed from memory).
void seqUnitsFor(uint32_t ntri, int nA, int nB, int cap, const LaunchHints &hints, int &uO, int &uY, int &uM, bool pre = false) {
  int uA, uB;
  seqUnitSplit(ntri, nA, nB, 100, cap, uA, uB);
  uO = uY = uA, uM = uB;
  if (nA == 4 && nB == 2) {
    int capPlace = cap;
    if ((ntri + 63u) / 64u < 31u) { // the two-master instantiations: 1, 2, 3, 4, 6, 9, 10, 11 slots
      const int need = uA > uB ? uA : uB;
      const int inst = need <= 4 ? need : need <= 6 ? 6 : need <= 9 ? 9 : need;
      if (inst < capPlace) capPlace = inst;
    }
    (void)seqUnitSplitByPlace(ntri, PTW_SEQ_YOUNG_PERCENT, capPlace, uO, uY, uM);
  } else if (nA == 6 && nB == 1 && (!pre || (ntri + 63u) / 64u < 31u)) { // (the prefilter's one-master kernels: only inside
    // the equal shares' instantiation - suzanne +6.4 %; ce 9 / 6 / 9 needs its 12-slot kernel: -3.4 %, profiles/r06y_*)
    int capPlace = cap;
    if ((ntri + 63u) / 64u < 31u) { // the one-master instantiations: 1, 2, 3, 4, 6, 8, 9, 10, 12 slots
      const int need = uA > uB ? uA : uB;
      const int inst = need <= 4 ? need : need <= 6 ? 6 : need <= 8 ? 8 : need;
      if (inst < capPlace) capPlace = inst;
    }
    (void)seqUnitSplitByPlaceOneMaster(ntri, PTW_SEQ_YOUNG_PERCENT, capPlace, uO, uY, uM);
  }
  const int o = hints.seqUnits[0], y = hints.seqUnits[1], m = hints.seqUnits[2];
  if ((o | y | m) != 0 && o >= 0 && y >= 0 && m >= 0 && o <= cap && y <= cap && m <= cap) uO = o, uY = y, uM = m;
}

template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, int MASTERS = 1, bool PRE = false, bool UNIT = false>
hipError_t launchSeq(const TraceParams &pIn, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  TraceParams p = pIn;
  if (WAVES > 1) {
    int uO, uY, uM;
    seqUnitsFor(p.ntri, WAVES - MASTERS, MASTERS, SLOTS, hints, uO, uY, uM, PRE);
    p.seqUnitsA = uO, p.seqUnitsY = uY, p.seqUnitsB = uM;
  }
  // (one wave per pass: the pick checksum is its own instantiation, see SeqCtx::picksOn)
  auto kernel = WAVES == 1 && !b.picks ? traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS, WAVES != 1, PRE, UNIT>
                                       : traceSequential<SLOTS, WAVES, LDS_TABLES, REG, MASTERS, true, PRE, UNIT>;
  setVariant("traceSequential<%d,%d,%s,%s%s%s>", SLOTS, WAVES, LDS_TABLES ? "lds" : "global", REG ? "reg" : "stack",
             MASTERS == 2 ? ",2 masters" : "", PRE ? ",prefilter" : UNIT ? ",unit" : "");
  if (hints.dryRun) return hipSuccess;
  const size_t lds = seqLdsBytes(WAVES, p.maxDepth, LDS_TABLES, p.ntri, p.nmat, p.nsph, MASTERS);
  if (lds > 48 * 1024) { // per launch: the attribute belongs to the current device's copy of the kernel
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3((p.npass + MASTERS - 1) / MASTERS),
                     dim3(WAVES == 1 ? 64 : 64 * (WAVES + MASTERS)), lds, stream, p, b.triGeom, b.triShade,
                     b.spheres, b.triCompact, b.matTable, b.mtState, b.mtPos, b.stage, b.words,
                     b.rays, b.picks, b.triPacked);
  return hipGetLastError();
}

// ... with the shading tables in LDS when they fit (LaunchHints::seqLdsTables == 0: global memory
// whatever their size - the tests reach the global-table instantiations with small scenes that way)
template <int SLOTS, int WAVES, int MASTERS = 1, bool PRE = false, bool UNIT = false>
hipError_t launchSeqAuto(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  const size_t tables = seqLdsBytes(WAVES, p.maxDepth, true, p.ntri, p.nmat, p.nsph, MASTERS);
  if (hints.seqLdsTables != 0 && tables <= kLdsTableBudget)
    return launchSeq<SLOTS, WAVES, true, false, MASTERS, PRE, UNIT>(p, b, hints, stream);
  return launchSeq<SLOTS, WAVES, false, false, MASTERS, PRE, UNIT>(p, b, hints, stream);
}

} // namespace
} // namespace ptw

// ptw_spec8.h - EXPERIMENT (make experiments; never in the shipped libptw_hip.so): the speculative
// sequential kernel with eight tracing waves.  Included by ptw_kernels.hip inside its anonymous
// namespace (it uses SeqCtx in SPEC mode) when PTW_EXPERIMENTS is set.  Measured slower than the
// four-wave kernel on the headline scene: kept for reference and A/B runs only.
// -----------------------------------------------------------------------------------------
// traceSequentialSpec8: the speculative kernel with EIGHT tracing waves - two per SIMD - and no
// generator wave.
//
// Round 2 measured that a SIMD carries two of these waves at the speed of one: the single-wave
// kernel does 17.7 Msamples/s at 1024 passes (one wave per SIMD), 36.1 at 2048 and 37.1 at 4096
// (profiles/r02k_cfg5_per_gpu_shares.txt) - a lone wave leaves half of its SIMD's issue slots
// unused, and a second wave takes them without slowing the first.  At 256 passes the four-wave
// kernel therefore uses half of what a CU offers.  Here eight waves trace eight candidates per
// round, each the way a wave of the four-wave kernel does (a lane per primitive, SeqCtx in SPEC
// mode): candidate c is a node (m, D) of the candidate set the many-candidate kernel uses
// (ptw_wide.hip: prefix-closed, maximising the expected number of committed sub-samples for the
// measured distribution of draw counts, rebuilt on the device before every band), and the commit
// walks the chain of candidates through that set's successor table.  With eight candidates a
// Cornell round commits ~2.5 sub-samples instead of 2.04 (scripts/sim/spec_sim2.py).  When the
// frontier crosses into the other ring block all eight waves regenerate the block left behind
// together (the mt19937 twist in three data-parallel phases).  Same ring, same parking format,
// same results as every other sequential kernel.
// -----------------------------------------------------------------------------------------
constexpr int kSpec8Waves = 8;

__host__ __device__ inline size_t spec8LdsBytes(uint32_t ntri, uint32_t nmat, uint32_t nsph) {
  size_t n = 2 * kRingStride;                          // the ring
  n += kMtWords * sizeof(uint32_t);                    // raw generator state
  n += 2 * kSpec8Waves * sizeof(SpecResult);           // results, double-buffered
  n = (n + 63) & ~static_cast<size_t>(63);
  n += static_cast<size_t>(nsph) * sizeof(SphereRec);
  n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
  n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  // more than half of a CU's 160 KB: one workgroup per CU, its eight waves two to a SIMD
  const size_t floor = 84 * 1024;
  return n < floor ? floor : n;
}

// The next block of the stream into the ring slot at `slotOff`, by all 512 lanes of the workgroup
// (uniform call; the layout is traceSequentialSpec's: ring at 0, raw state behind it).  The twist
// x[k] = f(x[k], x[k+1], x[k+397 mod 624]) reads words at most 227 behind its own position that
// this regeneration has already rewritten, so k in [0,227), [227,454), [454,623) are three
// data-parallel phases (all reads, barrier, all writes, barrier), then x[623].
__device__ __noinline__ void spec8GenerateBlock(unsigned char *lds, unsigned slotOff) {
  uint32_t *x = reinterpret_cast<uint32_t *>(lds + 2 * kRingStride);
  const int tid = threadIdx.x;
  for (int base = 0; base < 623; base += 227) {
    const int k = base + tid;
    const bool mine = tid < 227 && k < 623;
    uint32_t nv = 0;
    if (mine) nv = mtTwist(x[k], x[k + 1], base == 0 ? x[k + 397] : x[k - 227]);
    ldsBarrier();
    if (mine) x[k] = nv;
    ldsBarrier();
  }
  if (tid == 0) x[623] = mtTwist(x[623], x[0], x[396]);
  ldsBarrier();
  double *canon = reinterpret_cast<double *>(lds + slotOff);
  double *hemi = reinterpret_cast<double *>(lds + slotOff + kRingHemiOff);
  double *otherCanon = reinterpret_cast<double *>(lds + (slotOff ^ kRingStride));
  double *otherHemi = reinterpret_cast<double *>(lds + (slotOff ^ kRingStride) + kRingHemiOff);
  if (tid < kMtDoubles) {
    const double c = canonicalFromWords(mtTemper(x[2 * tid]), mtTemper(x[2 * tid + 1]));
    canon[tid] = c;
    if (tid < kRingCanonDoubles - kMtDoubles) otherCanon[kMtDoubles + tid] = c;
  }
  ldsBarrier();
  if (tid + 1 < kMtDoubles) hemiEntry(canon[tid], canon[tid + 1], hemi + 3 * tid);
  if (tid == kMtDoubles - 1)
    hemiEntry(otherCanon[kMtDoubles - 1], otherCanon[kMtDoubles], otherHemi + 3 * (kMtDoubles - 1));
  ldsBarrier();
}

__global__ __launch_bounds__(64 * kSpec8Waves) void traceSequentialSpec8(
    const TraceParams p, const double *__restrict__ triGeom, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, double *__restrict__ specState, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters,
    const WideCandidates *__restrict__ candSet, unsigned long long *__restrict__ countHist) {
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  constexpr int kBlock = 64 * kSpec8Waves;
  char *ring = reinterpret_cast<char *>(ldsRaw);
  uint32_t *mt = reinterpret_cast<uint32_t *>(ldsRaw + 2 * kRingStride);
  SpecResult *results = reinterpret_cast<SpecResult *>(mt + kMtWords); // [2][kSpec8Waves]
  size_t off = 2 * kRingStride + kMtWords * sizeof(uint32_t) + 2 * kSpec8Waves * sizeof(SpecResult);
  off = (off + 63) & ~static_cast<size_t>(63);

  const int pass = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;

  using Ctx = SeqCtx<1, 1, true, true, true>;
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
  ctx.loadPrimitives();

  // ---- the stream: resume (or start) this pass's generator ring (format of traceSequentialSpec) ----
  uint32_t *myState = mtState + static_cast<size_t>(pass) * kMtWords;
  double *myPark = specState + static_cast<size_t>(pass) * kSpecStateDoubles;
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) mt[i] = myState[i];
  unsigned fOff = 0; // frontier: ring slot (0 or kRingStride) ...
  int fQ = 0;        // ... and position in it
  __syncthreads();
  if (p.firstBand) {
    spec8GenerateBlock(ldsRaw, 0);
    spec8GenerateBlock(ldsRaw, kRingStride); // (completes block 0's overlap and its last table entry)
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

  // The block after the next is generated as soon as the frontier has crossed into the other
  // slot: the slot it left is free then (every wave is past the round's barrier and reads the
  // ring again only in the next round).
  auto advanceFrontier = [&](int n) { // n < kMtDoubles; uniform over the workgroup
    const int np = fQ + n;
    if (np >= kMtDoubles) {
      const unsigned left = fOff;
      fQ = np - kMtDoubles;
      fOff ^= kRingStride;
      spec8GenerateBlock(ldsRaw, left);
    } else {
      fQ = np;
    }
  };

  // this wave's candidate, and - for the walk after a round - what lane c knows about candidate c
  const int nCand = __builtin_amdgcn_readfirstlane(candSet->count);
  const unsigned myNode = __builtin_amdgcn_readfirstlane(wave < nCand ? static_cast<int>(candSet->node[wave]) : 0xffff);
  const int myM = static_cast<int>(myNode >> 8), myD = static_cast<int>(myNode & 0xffu);
  const unsigned succV = lane < nCand ? candSet->succ[lane] : 0xffffffffu;

  const int w = p.width;
  const bool lens = p.cam.aperture_radius != 0;
  const int nSub = p.fbU * p.fbV;
  const int vShift = p.fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(p.fbV)) : 0;
  // Per-round constants in vector registers (see traceSequentialSpec)
  double invU = p.invU, invV = p.invV;
  asm volatile("" : "+v"(invU), "+v"(invV));
  const bool fastFan = (p.uPow2 & p.vPow2) != 0;
  const int vMask = p.fbV - 1;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
  unsigned long long raysTotal = 0;
  unsigned h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0; // committed sub-samples by levels reached (wave 0)
  int parity = 0;
#if PTW_PROFILE_PHASES
  unsigned long long stRounds = 0, stCommits = 0, stWork = 0, stWait = 0, stCommit = 0, stPrimary = 0, stIdle = 0;
  const unsigned long long stT0 = __builtin_amdgcn_s_memtime();
#endif

  for (uint32_t i = 0; i < p.pixCount; ++i) {
    const uint32_t pix = p.pixBegin + i;
    const int px = static_cast<int>(pix % static_cast<uint32_t>(w));
    const int py = static_cast<int>(pix / static_cast<uint32_t>(w));
    // ---- every wave: camera ray and first hit at the frontier (redundant, in parallel) ----
    PTW_T(tP0);
    ctx.setStream(fOff, fQ);
    double r0, r1, r2 = 0, r3 = 0;
    if (lens) {
      ctx.draw4(r0, r1, r2, r3);
    } else {
      r0 = ctx.draw();
      r1 = ctx.draw();
    }
    const int camDraws = lens ? 4 : 2;
    d3 o, d;
    cameraRay(p.cam, px, py, r0, r1, r2, r3, o, d);
    int sampleDraws = camDraws;
    d3 L = mk(0, 0, 0);
    bool traced = false;
    HitKey k0;
    k0.t = kInf, k0.idx = kMiss, k0.det = 0;
    if (p.maxDepth > 0) {
      k0 = ctx.intersect(o, d);
      raysTotal++;
      if (uniformBool(k0.idx == kMiss)) {
        L = ld3(p.env);
      } else {
        traced = true;
      }
    }
    advanceFrontier(camDraws);
#if PTW_PROFILE_PHASES
    stPrimary += __builtin_amdgcn_s_memtime() - tP0;
#endif
    if (traced) {
      const Surface first = ctx.surfaceAt(k0, o, d);
      if (p.preview) {
        L = first.diffuse; // Scene.cpp:137-138
      } else {
        d3 result = mk(0, 0, 0);
        int j = 0;
        unsigned pixHist = 0; // 6-bit fields
        while (j < nSub) {
          // ---- this wave's candidate: sub-sample j + myM, stream position frontier + myD ----
          const int myIdx = j + myM;
          PTW_T(tW0);
          SpecResult mine;
          mine.L[0] = mine.L[1] = mine.L[2] = 0;
          mine.meta = 0, mine.pad = 0;
          if (myIdx < nSub) { // (a wave without a candidate carries node 0xffff: myM = 255)
            const int np = fQ + myD; // myD < kMtDoubles
            const bool wrap = np >= kMtDoubles;
            ctx.setStream(wrap ? fOff ^ kRingStride : fOff, wrap ? np - kMtDoubles : np);
            ctx.words = 0;
            ctx.rays = 0;
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
          }
          SpecResult *slot = results + parity * kSpec8Waves;
          if (lane == 0) slot[wave] = mine;
          PTW_T(tW1);
          ldsBarrier();
          PTW_T(tW2);
          // ---- commit (identical in every wave): lane c prepares what the walk needs to know
          //      about candidate c - its draw count, its rays, the candidate that continues it ----
          const int metaV = lane < kSpec8Waves ? slot[lane & (kSpec8Waves - 1)].meta : 0;
          int packV;
          {
            const int cnt = metaV & 0xff;
            const int levels = (cnt * 11) >> 5; // cnt / 3 for cnt <= 27
            const unsigned next = levels >= 1 && levels <= 5 ? (succV >> (6 * (levels - 1))) & 63u : 63u;
            packV = static_cast<int>(next | (static_cast<unsigned>(cnt) << 8) |
                                     (static_cast<unsigned>(metaV >> 16) << 16));
          }
          int m = 0, D = 0;
          unsigned raysRound = 0;
          unsigned chain = 0; // committed candidate indices, 3 bits each (at most 8)
          for (int c = 0; c != 63 && j + m < nSub;) {
            const unsigned wd = static_cast<unsigned>(__builtin_amdgcn_readlane(packV, c));
            chain |= static_cast<unsigned>(c) << (3 * m);
            D += static_cast<int>((wd >> 8) & 0xffu);
            raysRound += wd >> 16;
            ++m;
            c = m < kSpec8Waves ? static_cast<int>(wd & 63u) : 63;
          }
          raysTotal += raysRound;
          if (wave == 0) { // only the wave that stores the sample needs the radiance (and the statistics)
            for (int q = 0; q < m; ++q) {
              const int src = static_cast<int>((chain >> (3 * q)) & 7u);
              const SpecResult &r = slot[src];
              const int meta = r.meta;
              const d3 child = mk(r.L[0], r.L[1], r.L[2]);
              result = result + ((meta & 0x100) ? first.emission + child
                                                : first.emission + first.diffuse * child);
              const int levels = ((meta & 0xff) * 11) >> 5;
              pixHist += 1u << (6 * ((levels < 5 ? levels : 5) - 1));
            }
          }
          j += m;
          sampleDraws += D;
          parity ^= 1;
          advanceFrontier(D); // (may regenerate a block: uniform, with barriers)
#if PTW_PROFILE_PHASES
          stRounds++, stCommits += m;
          stIdle += !(myIdx < nSub);
          stWork += tW1 - tW0, stWait += tW2 - tW1, stCommit += __builtin_amdgcn_s_memtime() - tW2;
#endif
        }
        L = result * p.invFirstBounce;
        h1 += pixHist & 63u, h2 += (pixHist >> 6) & 63u, h3 += (pixHist >> 12) & 63u;
        h4 += (pixHist >> 18) & 63u, h5 += (pixHist >> 24) & 63u;
      }
    }
    if (threadIdx.x == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = 2u * static_cast<unsigned>(sampleDraws);
    }
  }

#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const double n = static_cast<double>(p.pixCount);
    printf("SPEC8 wave %d (node m=%d D=%d): cycles/sample=%.0f rounds/sample=%.2f commits/round=%.2f idle=%.2f "
           "primary=%.0f work=%.0f wait=%.0f commit+advance=%.0f (per sample)\n",
           wave, myM, myD, (__builtin_amdgcn_s_memtime() - stT0) / n, stRounds / n,
           static_cast<double>(stCommits) / stRounds, static_cast<double>(stIdle) / stRounds, stPrimary / n,
           stWork / n, stWait / n, stCommit / n);
  }
#endif
  if (threadIdx.x == 0) {
    myPark[2 * kRingCanonDoubles] = fOff ? 1.0 : 0.0;
    myPark[2 * kRingCanonDoubles + 1] = static_cast<double>(fQ);
    if (rayCounters) rayCounters[pass] += raysTotal;
    if (countHist) {
      atomicAdd(&countHist[0], static_cast<unsigned long long>(h1));
      atomicAdd(&countHist[1], static_cast<unsigned long long>(h2));
      atomicAdd(&countHist[2], static_cast<unsigned long long>(h3));
      atomicAdd(&countHist[3], static_cast<unsigned long long>(h4));
      atomicAdd(&countHist[4], static_cast<unsigned long long>(h5));
    }
  }
  // ---- park the stream for the next band ----
  __syncthreads();
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = mt[i];
  for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
    const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
    myPark[i] = reinterpret_cast<const double *>(ring + slot * kRingStride)[k];
  }
}


// ptw_gang.h - EXPERIMENT (make experiments; never in the shipped libptw_hip.so): the candidate-set
// builder of the many-candidate sequential kernels and traceSequentialGang, the speculative kernel
// spread over several CUs per pass.  Included by ptw_kernels.hip inside its anonymous namespace when
// PTW_EXPERIMENTS is set.  Measured (round 3, profiles/r03d_gang_*): bit-identical to the other
// sequential kernels, and no faster than one CU per pass on the headline scene - what 32 candidates
// commit more per round than 4, the meeting of eight CUs through memory costs again.
// The candidate set: prefix-closed greedy on the probability that the true chain reaches a node
// with every node before it in the set (= the expected number of sub-samples a round commits),
// for sub-sample draw counts distributed like the histogram the previous band of this render
// measured (`hist[k]`: committed sub-samples that reached k + 1 levels, i.e. consumed 3 (k + 1)
// draws; read and reset here) or, while there is too little of it, like a prior.  One lane; it
// runs once per band, in stream order before the band's trace kernel.
constexpr int kCandMaxM = 17, kCandMaxK = 5 * kCandMaxM + 1;
__global__ __launch_bounds__(64) void wideBuildCandidates(unsigned long long *__restrict__ hist, int n, int nSub,
                                                          int maxAhead, WideCandidates *__restrict__ out) {
  __shared__ float reach[kCandMaxM][kCandMaxK]; // frontier: reach probability if the node were added
  __shared__ unsigned char taken[kCandMaxM][kCandMaxK]; // 1 + candidate index
  if (threadIdx.x != 0) return;
  // sub-samples that consume 3, 6, 9, 12, 15 (and more) draws: a closed box mostly runs every path
  // to the depth cap, an open scene mostly loses the first ray
  float prob[6] = {0.f, 0.01f, 0.23f, 0.12f, 0.09f, 0.55f};
  unsigned long long total = 0;
  for (int k = 0; k < 5; ++k) total += hist[k];
  if (total >= 4096) {
    for (int k = 0; k < 5; ++k) prob[k + 1] = static_cast<float>(static_cast<double>(hist[k]) / static_cast<double>(total));
  }
  for (int k = 0; k < 5; ++k) hist[k] = 0;
  for (int m = 0; m < kCandMaxM; ++m)
    for (int k = 0; k < kCandMaxK; ++k) reach[m][k] = 0.f, taken[m][k] = 0;
  auto expand = [&](int m, int k, float r) {
    if (m + 1 >= nSub || m + 1 >= kCandMaxM) return;
    for (int c = 1; c <= 5; ++c)
      if (prob[c] > 0.f && 3 * (k + c) <= maxAhead) reach[m + 1][k + c] += r * prob[c];
  };
  int count = 0, maxD = 0;
  out->node[count++] = 0;
  taken[0][0] = 1;
  expand(0, 0, 1.f);
  while (count < n) {
    int bm = -1, bk = 0;
    float best = 0.f;
    for (int m = 1; m < kCandMaxM; ++m)
      for (int k = m; k <= 5 * m; ++k)
        if (!taken[m][k] && reach[m][k] > best) best = reach[m][k], bm = m, bk = k;
    if (bm < 0) break;
    taken[bm][bk] = static_cast<unsigned char>(count + 1);
    out->node[count++] = static_cast<uint16_t>((bm << 8) | (3 * bk));
    maxD = 3 * bk > maxD ? 3 * bk : maxD;
    expand(bm, bk, best);
  }
  for (int i = count; i < kWideMaxCand; ++i) out->node[i] = 0xffffu;
  for (int i = 0; i < kWideMaxCand; ++i) {
    uint32_t row = 0;
    for (int c = 1; c <= 5; ++c) {
      uint32_t next = 63;
      if (i < count) {
        const int m = out->node[i] >> 8, k = (out->node[i] & 0xff) / 3;
        if (m + 1 < kCandMaxM && k + c < kCandMaxK && taken[m + 1][k + c]) next = taken[m + 1][k + c] - 1u;
      }
      row |= next << (6 * (c - 1));
    }
    out->succ[i] = row;
  }
  out->count = count;
  out->maxD = maxD;
}


// -----------------------------------------------------------------------------------------
// traceSequentialGang: the speculative kernel spread over SEVERAL CUs per pass - for renders
// that hold fewer passes than the GPU has CUs (cfg2's 256 passes split over 8 GPUs leave 32 per
// GPU: with one CU per pass 7/8 of the chip idles and the frame takes as long as on one GPU).
//
// A pass is served by G workgroups (G = 2, 4 or 8), each four tracing waves - one per SIMD of its
// CU, each holding the whole scene in registers like traceSequentialSpec's - so 4 G candidates per
// round instead of 4.  Candidate c = 4 * member + wave is node (m, D) of the prefix-closed candidate
// set that maximises the expected number of sub-samples a round commits for the measured
// distribution of per-sub-sample draw counts (buildCandidates below, per band): "sub-sample j + m of
// the pixel, starting D draws after the stream frontier".  With 16 candidates a Cornell round
// commits 3.3 sub-samples, with 32 about 4.2, against 2.04 for four (scripts/sim/spec_sim2.py; the
// many-candidate experiments of round 2 measured the same rates inside one CU).
//
// The workgroups of a pass meet once per round, through global memory: every wave writes its
// result record (radiance of the sub-path, draws consumed, lobe, rays) and then the record's tag -
// the round number - and polls the tags of all candidates of the round (lane c polls candidate c);
// no workgroup barrier and no atomic read-modify-write is involved, and records are double-buffered
// by round parity (a wave can only be one round ahead of the slowest reader, which has to publish
// its own record before anybody can commit the round).  All accesses to the records are
// agent-scope atomics (they bypass the per-XCD L2's non-coherent lines); the blockIdx -> (pass,
// member) map puts the G workgroups of a pass on the same XCD (workgroups go to the XCDs round
// robin).  Every wave then walks the chain of committed candidates through the set's successor
// table; wave 0 of member 0 adds the committed contributions in sub-sample order (the value the
// serial evaluation defines, bit for bit) and stores the sample.  Each workgroup keeps its own copy
// of the stream ring and regenerates blocks itself (all four waves together, when the frontier
// crosses into the other slot).  Same ring, same parking format, same results as every other
// sequential kernel.  The launch is cooperative (the runtime guarantees that all workgroups are
// resident); a poll that never succeeds gives up after a few seconds and poisons the pass's output
// with NaN instead of hanging the device.
// -----------------------------------------------------------------------------------------
constexpr int kGangWaves = 4;
constexpr int kGangMaxGroups = 8;
constexpr int kGangMaxCand = kGangWaves * kGangMaxGroups;

struct alignas(32) GangRecord { // one per pass, round parity and candidate, in global memory
  unsigned long long L[3];      // radiance of the sub-path below the first-bounce surface (bits)
  unsigned long long tag;       // low: draws consumed | lobe << 8 | rays << 16; high: round number (from 1)
};
static_assert(sizeof(GangRecord) == 32, "record size");

__host__ __device__ inline size_t gangLdsBytes(uint32_t ntri, uint32_t nmat, uint32_t nsph) {
  size_t n = 2 * kRingStride;                          // the ring
  n += kMtWords * sizeof(uint32_t);                    // raw generator state
  n = (n + 63) & ~static_cast<size_t>(63);
  n += static_cast<size_t>(nsph) * sizeof(SphereRec);
  n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
  n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  const size_t floor = 84 * 1024; // more than half of a CU's LDS: one workgroup per CU
  return n < floor ? floor : n;
}

// The next block of the stream into the ring slot at `slotOff`, by all lanes of the workgroup
// (uniform call; ring at 0, raw state behind it).  The twist x[k] = f(x[k], x[k+1], x[k+397 mod 624])
// reads words at most 227 behind its own position that this regeneration has already rewritten, so
// k in [0,227), [227,454), [454,623) are three data-parallel phases (all reads, barrier, all
// writes, barrier), then x[623].
template <int NT>
__device__ __noinline__ void coopGenerateBlock(unsigned char *lds, unsigned slotOff) {
  static_assert(NT >= 227, "a phase of the twist in one go");
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
  for (int i = tid; i < kMtDoubles; i += NT) {
    const double c = canonicalFromWords(mtTemper(x[2 * i]), mtTemper(x[2 * i + 1]));
    canon[i] = c;
    if (i < kRingCanonDoubles - kMtDoubles) otherCanon[kMtDoubles + i] = c;
  }
  ldsBarrier();
  for (int i = tid; i + 1 < kMtDoubles; i += NT) hemiEntry(canon[i], canon[i + 1], hemi + 3 * i);
  if (tid == NT - 1)
    hemiEntry(otherCanon[kMtDoubles - 1], otherCanon[kMtDoubles], otherHemi + 3 * (kMtDoubles - 1));
  ldsBarrier();
}

__device__ __forceinline__ unsigned long long gangLoad(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gangStore(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64 * kGangWaves) void traceSequentialGang(
    const TraceParams p, const double *__restrict__ triGeom, const SphereRec *__restrict__ spheres,
    const double *__restrict__ triCompact, const double *__restrict__ matTable,
    uint32_t *__restrict__ mtState, double *__restrict__ specState, double *__restrict__ stage,
    uint32_t *__restrict__ words, unsigned long long *__restrict__ rayCounters,
    const WideCandidates *__restrict__ candSet, unsigned long long *__restrict__ countHist,
    GangRecord *__restrict__ records, int G) {
  extern __shared__ __attribute__((aligned(64))) unsigned char ldsRaw[];
  constexpr int kBlock = 64 * kGangWaves;
  char *ring = reinterpret_cast<char *>(ldsRaw);
  uint32_t *mt = reinterpret_cast<uint32_t *>(ldsRaw + 2 * kRingStride);
  size_t off = 2 * kRingStride + kMtWords * sizeof(uint32_t);
  off = (off + 63) & ~static_cast<size_t>(63);

  // blockIdx -> (pass, member): the G workgroups of a pass on one XCD (workgroup b runs on XCD b % 8)
  const int xcd = blockIdx.x & 7, slotInXcd = blockIdx.x >> 3;
  const int member = __builtin_amdgcn_readfirstlane(slotInXcd % G);
  const int pass = __builtin_amdgcn_readfirstlane((slotInXcd / G) * 8 + xcd);
  if (static_cast<uint32_t>(pass) >= p.npass) return;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const bool leader = member == 0 && wave == 0; // stores the samples, keeps the statistics

  using Ctx = SeqCtx<1, 1, true, true, true, 1, false, false>;
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
  ctx.picksOn = false; // (no pick checksum in this kernel: the dispatcher refuses d_picks for it)
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
  ctx.loadPrimitives();

  // ---- the stream: resume (or start) this pass's generator ring (format of traceSequentialSpec) ----
  uint32_t *myState = mtState + static_cast<size_t>(pass) * kMtWords;
  double *myPark = specState + static_cast<size_t>(pass) * kSpecStateDoubles;
  for (int i = threadIdx.x; i < kMtWords; i += kBlock) mt[i] = myState[i];
  unsigned fOff = 0; // frontier: ring slot (0 or kRingStride) ...
  int fQ = 0;        // ... and position in it
  __syncthreads();
  if (p.firstBand) {
    coopGenerateBlock<kBlock>(ldsRaw, 0);
    coopGenerateBlock<kBlock>(ldsRaw, kRingStride); // (completes block 0's overlap and its last table entry)
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

  auto advanceFrontier = [&](int n) { // n < kMtDoubles; uniform over the workgroup (and over the pass)
    const int np = fQ + n;
    if (np >= kMtDoubles) {
      const unsigned left = fOff;
      fQ = np - kMtDoubles;
      fOff ^= kRingStride;
      coopGenerateBlock<kBlock>(ldsRaw, left);
    } else {
      fQ = np;
    }
  };

  // this wave's candidate, and - for the walk after a round - what lane c knows about candidate c
  const int nCandSet = __builtin_amdgcn_readfirstlane(candSet->count);
  const int nCand = nCandSet < kGangWaves * G ? nCandSet : kGangWaves * G;
  const int myCand = member * kGangWaves + wave;
  const unsigned myNode = __builtin_amdgcn_readfirstlane(myCand < nCand ? static_cast<int>(candSet->node[myCand]) : 0xffff);
  const int myM = static_cast<int>(myNode >> 8), myD = static_cast<int>(myNode & 0xffu);
  const unsigned succV = lane < nCand ? candSet->succ[lane] : 0xffffffffu;
  GangRecord *passRecords = records + static_cast<size_t>(pass) * 2 * kGangMaxCand;

  const int w = p.width;
  const bool lens = p.cam.aperture_radius != 0;
  const int nSub = p.fbU * p.fbV;
  const int vShift = p.fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(p.fbV)) : 0;
  double invU = p.invU, invV = p.invV;
  asm volatile("" : "+v"(invU), "+v"(invV));
  const bool fastFan = (p.uPow2 & p.vPow2) != 0;
  const int vMask = p.fbV - 1;
  double *myStage = stage + static_cast<size_t>(pass) * p.pixCount * 3;
  unsigned long long raysTotal = 0;
  unsigned h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0; // committed sub-samples by levels reached (leader)
  unsigned roundNo = 0;                            // rounds of this launch, the same in every wave of the pass
  bool failed = false;
#if PTW_PROFILE_PHASES
  unsigned long long stRounds = 0, stCommits = 0, stWork = 0, stWait = 0, stCommit = 0, stPrimary = 0, stIdle = 0;
  const unsigned long long stT0 = __builtin_amdgcn_s_memtime();
#endif

  for (uint32_t i = 0; i < p.pixCount && !failed; ++i) {
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
        // leader: the committed candidates of the previous round, their radiance still on its way
        int pendM = 0, pendMeta = 0;
        unsigned long long pendChain = 0;
        double pendLx = 0, pendLy = 0, pendLz = 0;
        auto addPending = [&]() {
          for (int q = 0; q < pendM; ++q) {
            const int src = static_cast<int>((pendChain >> (5 * q)) & 31u);
            const int meta = __builtin_amdgcn_readlane(pendMeta, src);
            const d3 child = mk(readLane(pendLx, src), readLane(pendLy, src), readLane(pendLz, src));
            result = result + ((meta & 0x100) ? first.emission + child
                                              : first.emission + first.diffuse * child);
            const int levels = ((meta & 0xff) * 11) >> 5;
            pixHist += 1u << (6 * ((levels < 5 ? levels : 5) - 1));
          }
          pendM = 0;
        };
        while (j < nSub) {
          // ---- this wave's candidate: sub-sample j + myM, stream position frontier + myD ----
          const int myIdx = j + myM;
          PTW_T(tW0);
          d3 mineL = mk(0, 0, 0);
          unsigned mineMeta = 0;
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
            mineL = ctx.chainHot(p, first.pos, nd);
            mineMeta = (ctx.words >> 1) | (refl ? 0x100u : 0u) | (static_cast<unsigned>(ctx.rays) << 16);
          }
          // ---- publish: the record, then its tag (the round number) ----
          ++roundNo;
          GangRecord *slot = passRecords + (roundNo & 1u) * kGangMaxCand;
          if (myCand < nCand && lane == 0) {
            GangRecord *mine = slot + myCand;
            gangStore(&mine->L[0], static_cast<unsigned long long>(__builtin_bit_cast(long long, mineL.x)));
            gangStore(&mine->L[1], static_cast<unsigned long long>(__builtin_bit_cast(long long, mineL.y)));
            gangStore(&mine->L[2], static_cast<unsigned long long>(__builtin_bit_cast(long long, mineL.z)));
            // The three stores above are agent-scope atomics (written through to the coherence point,
            // nothing of this record sits dirty in a cache): once they are acknowledged the tag may
            // go out.  (An agent-scope release fence does that too, and writes the whole L2 back first:
            // measured 10 k cycles per round, profiles/r03c_gang_*.)  The wait also covers the leader's
            // outstanding loads of the previous round's records (see below).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gangStore(&mine->tag, static_cast<unsigned long long>(mineMeta) | (static_cast<unsigned long long>(roundNo) << 32));
          }
          PTW_T(tW1);
          // ---- meet: lane c waits for candidate c's record of this round ----
          unsigned long long tagV = 0;
          {
            unsigned spins = 0;
            for (;;) {
              bool ready = true;
              if (lane < nCand) {
                tagV = gangLoad(&slot[lane].tag);
                ready = static_cast<unsigned>(tagV >> 32) == roundNo;
              }
              if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
              if (++spins > (1u << 22)) { // seconds: a peer is gone (never on a cooperative launch)
                failed = true;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            // (no acquire fence: the records are only ever read with agent-scope atomic loads, which
            // do not look at the non-coherent cache levels)
          }
          if (failed) break;
          PTW_T(tW2);
          // ---- commit (identical in every wave of the pass): lane c prepares what the walk needs to
          //      know about candidate c - its draw count, its rays, the candidate that continues it ----
          const int metaV = static_cast<int>(static_cast<unsigned>(tagV));
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
          unsigned long long chain = 0; // committed candidate indices, 5 bits each (at most 12 fit; nSub caps m)
          for (int c = 0; c != 63 && j + m < nSub && m < 12;) {
            const unsigned wd = static_cast<unsigned>(__builtin_amdgcn_readlane(packV, c));
            chain |= static_cast<unsigned long long>(c) << (5 * m);
            D += static_cast<int>((wd >> 8) & 0xffu);
            raysRound += wd >> 16;
            ++m;
            c = static_cast<int>(wd & 63u);
            if (c >= nCand) c = 63;
          }
          raysTotal += raysRound;
          if (leader) { // only the wave that stores the sample needs the radiance (and the statistics)
            // The radiance of the committed candidates is fetched now and ADDED one round later (or
            // when the pixel ends): the loads - a trip to the coherence point and back - travel while
            // this wave, which also traces the frontier candidate, is already on its next sub-path.
            // The records of this round cannot be overwritten before this wave has published its
            // next record, and it waits for these loads before it does (the vmcnt wait above).
            addPending();
            if (lane < nCand) {
              pendLx = __builtin_bit_cast(double, static_cast<long long>(gangLoad(&slot[lane].L[0])));
              pendLy = __builtin_bit_cast(double, static_cast<long long>(gangLoad(&slot[lane].L[1])));
              pendLz = __builtin_bit_cast(double, static_cast<long long>(gangLoad(&slot[lane].L[2])));
            }
            pendM = m, pendChain = chain, pendMeta = metaV;
          }
          j += m;
          sampleDraws += D;
          advanceFrontier(D); // (may regenerate a block: uniform, with workgroup barriers)
#if PTW_PROFILE_PHASES
          stRounds++, stCommits += m;
          stIdle += !(myIdx < nSub);
          stWork += tW1 - tW0, stWait += tW2 - tW1, stCommit += __builtin_amdgcn_s_memtime() - tW2;
#endif
        }
        if (leader) addPending();
        L = result * p.invFirstBounce;
        h1 += pixHist & 63u, h2 += (pixHist >> 6) & 63u, h3 += (pixHist >> 12) & 63u;
        h4 += (pixHist >> 18) & 63u, h5 += (pixHist >> 24) & 63u;
      }
    }
    if (leader && lane == 0) {
      myStage[i * 3 + 0] = L.x;
      myStage[i * 3 + 1] = L.y;
      myStage[i * 3 + 2] = L.z;
      if (words) words[static_cast<size_t>(pass) * p.npix + pix] = 2u * static_cast<unsigned>(sampleDraws);
    }
  }
  if (failed && leader && lane == 0) { // fail loudly: the pass's samples of this band become NaN
    const double bad = __builtin_nan("");
    for (uint32_t i = 0; i < p.pixCount * 3; ++i) myStage[i] = bad;
  }

#if PTW_PROFILE_PHASES
  if (pass == 0 && lane == 0) {
    const double n = static_cast<double>(p.pixCount);
    printf("GANG member %d wave %d (node m=%d D=%d): cycles/sample=%.0f rounds/sample=%.2f commits/round=%.2f idle=%.2f "
           "primary=%.0f work=%.0f wait=%.0f commit+advance=%.0f (per sample)\n",
           member, wave, myM, myD, (__builtin_amdgcn_s_memtime() - stT0) / n, stRounds / n,
           static_cast<double>(stCommits) / stRounds, static_cast<double>(stIdle) / stRounds, stPrimary / n,
           stWork / n, stWait / n, stCommit / n);
  }
#endif
  if (leader && lane == 0) {
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
  // ---- park the stream for the next band (member 0 only: the members hold identical copies) ----
  __syncthreads();
  if (member == 0) {
    for (int i = threadIdx.x; i < kMtWords; i += kBlock) myState[i] = mt[i];
    for (int i = threadIdx.x; i < 2 * kRingCanonDoubles; i += kBlock) {
      const int slot = i / kRingCanonDoubles, k = i - slot * kRingCanonDoubles;
      myPark[i] = reinterpret_cast<const double *>(ring + slot * kRingStride)[k];
    }
  }
}


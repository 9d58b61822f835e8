// ptw_pair.h - EXPERIMENTS BUILD ONLY (make experiments; included INSIDE struct SeqCtx of ptw_kernels.hip):
// the PAIR form of the two-master worker-wave kernels - two sub-samples of the first-bounce fan-out in
// flight per master, two rays per request to the workers.  Round 5 built it in four protocols, every
// one bit-identical to the oracle (radiance, RNG word counts, pick checksums: tests/test_gpu_round5.py),
// every one SLOWER than round 4's lock step on BASELINE cfg3 (suzanne 8.5-9.2 against 12.0-12.2
// Msamples/s) and no faster on cfg4 (ce 1.87-2.21 against 2.15-2.17): the master's work per answer -
// not the search - is what a tick lasts, it is the same work whichever protocol delivers the answer, and
// a request with two rays doubles the workers' search.  DESIGN.md 3.1e has the anatomy; this file is the
// last of the four forms (one barrier per tick, both masters busy in every tick, command slots
// alternating), kept for the record like csrc/experiments/ptw_gang.h.
// (no include guard: it is part of a class body)

  // The same for TWO rays at once (PAIR): every resident triangle is tested against both while its nine
  // doubles sit in registers.  The rays are wave-uniform and arrive in scalar registers (every
  // instruction of the test takes at most one of their components as its scalar operand), so a second
  // ray costs the worker no vector registers beyond its own best-so-far triple.
  __device__ __forceinline__ void localNearest2(d3 oA, d3 dA, d3 oB, d3 dB, HitKey &keyA, HitKey &keyB) {
    PTW_T(tA);
    double bestTA = kInf, bestDetA = 0, bestTB = kInf, bestDetB = 0;
    uint32_t bestIdxA = kMiss, bestIdxB = kMiss;
    const uint32_t nsph = p->nsph;
    if (hasSphere) {
      testSphere(oA, dA, mk(scx, scy, scz), sr2, static_cast<uint32_t>(tid), bestTA, bestIdxA);
      testSphere(oB, dB, mk(scx, scy, scz), sr2, static_cast<uint32_t>(tid), bestTB, bestIdxB);
    }
    if (nsph > static_cast<uint32_t>(kThreads)) // rare: more spheres than lanes
      for (uint32_t i = tid + kThreads; i < nsph; i += kThreads) {
        const SphereRec &r = spheresGlobal[i];
        testSphere(oA, dA, ld3(r.centre), r.radiusSquared, i, bestTA, bestIdxA);
        testSphere(oB, dB, ld3(r.centre), r.radiusSquared, i, bestTB, bestIdxB);
      }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (s >= myUnits) continue; // (a guard, not a break: see localNearest)
      const d3 v0 = mk(v0x[s], v0y[s], v0z[s]), e1 = mk(e1x[s], e1y[s], e1z[s]), e2 = mk(e2x[s], e2y[s], e2z[s]);
      testTriangle(oA, dA, v0, e1, e2, nsph + slotTriangle(s), bestTA, bestIdxA, bestDetA);
      testTriangle(oB, dB, v0, e1, e2, nsph + slotTriangle(s), bestTB, bestIdxB, bestDetB);
    }
    if (p->ntri > residentTriangles()) // rare: more triangles than resident slots
      for (uint32_t k = residentTriangles() + tid; k < p->ntri; k += kThreads) {
        const double *g = triGeom + 9 * static_cast<size_t>(k);
        const d3 v0 = ld3(g), e1 = ld3(g + 3), e2 = ld3(g + 6);
        testTriangle(oA, dA, v0, e1, e2, nsph + k, bestTA, bestIdxA, bestDetA);
        testTriangle(oB, dB, v0, e1, e2, nsph + k, bestTB, bestIdxB, bestDetB);
      }
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(bestTA), "+v"(bestTB));
#endif
    PTW_T(tB);
    PTW_ACC(0, tA, tB);
    keyA = pickNearest(bestTA, bestIdxA, bestDetA, minSlot);
    keyB = pickNearest(bestTB, bestIdxB, bestDetB, minSlot + 1);
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(keyA.t), "+v"(keyB.t));
#endif
    PTW_T(tC);
    PTW_ACC(1, tB, tC);
  }


  // =========================================================================================
  // PAIR (two-master kernels, round 5): TWO sub-samples of the first-bounce fan-out in flight per
  // master.  The stream makes a pass serial but not unpredictable (see traceSequentialSpec): sub-sample
  // j + 1 starts where j stops, and the number of draws j consumes - three per level it reaches - takes
  // few values that repeat (suzanne: 3 in four cases of five; ce: always 15).  So next to the chain X of
  // the sub-sample at the stream frontier the master keeps a chain Y for the sub-sample after it,
  // started at the position X WOULD leave the stream at if it consumed g = max(m1, levels X has
  // consumed already) levels, m1 being the most frequent count of this pass so far.  Every request to
  // the workers carries the next ray of both (command slots 0 and 1; SeqCommand::nrays is the mask of
  // slots that hold a ray) and comes back with two nearest hits: one barrier pair, one hand-off, one
  // pick round for two rays - the worker waves, idle half of every tick on suzanne, test their
  // resident triangles against both while they have them in registers.  When X ends, Y is the
  // frontier's sub-sample if and only if it started where X stopped (Y.start == pos): then it is
  // promoted - with whatever it has traced since - and a new Y is started behind it; otherwise it is
  // dropped and costs nothing but the workers' time.  X's contribution is added when X ends, so the
  // contributions are added in sub-sample order and the value is the one the serial evaluation defines,
  // bit for bit; the RNG word count of a sample is the frontier's progress, the ray counter counts
  // committed sub-samples only.  A chain's in-flight ray lives in its command slot, its (E, T) levels
  // in its own LDS stack; what stays in registers is a dozen wave-uniform integers per chain.
  // Y is only started where its worst case (3 maxDepth draws) ends inside the generator block, so a
  // speculated chain never regenerates - and neither does X while a Y exists (X.start <= Y.start).
  // =========================================================================================
  // (All of a chain's state here is wave-uniform and kept in plain integers, so that it lives in scalar
  // registers: the master's vector registers are full - 250 of 256 before this form existed - and what
  // does not fit is spilled to scratch memory, whose reloads on the serial path cost more than the
  // pairing wins (first version: 24 spilled registers, 5.3 k cycles from answers to the next request;
  // profiles/r05b_*).  For the same reason the first-bounce surface lives in LDS (pixRec), a finished
  // chain's radiance waits in its command slot, and a hit key is scalarised before it is used.)
  struct Chain {
    int slot;       // command slot, answers and stack of this chain (0 or 1)
    int sub;        // sub-sample index
    int start, pos; // Y: block position of its first / next draw (X draws at the frontier: ctx.pos)
    int depth;      // depth of the ray in flight (1: the ray that leaves the first-bounce surface)
    int levels;     // groups of three draws consumed so far (the scatter at the first-bounce surface = 1)
    int nlev;       // levels on its stack
    int refl0;      // lobe taken at the first-bounce surface (1: reflective)
    int live, done;
    int pend;       // the level `pendLevel` (a diffuse triangle bounce) still has to be written to the stack
    int pendLevel;
    uint32_t pendIdx;
    uint32_t rays, s1, s2; // intersect() calls so far; pick checksum partial sums
  };
  // The first-bounce scatter of the sub-samples to come, for every stream position they may start at:
  // 64 (sub-sample, position) pairs evaluated by the master's 64 lanes AT ONCE - one pass through the
  // scatter's sincos and square roots instead of one per chain start on the serial path (fanBuild,
  // called while the workers search).  Lane 16 a + m holds sub-sample fanJ0 + a at block position
  // fanQ0 + 3 (a + m): a sub-sample cannot start before its predecessors have consumed a level each.
  FanEntry *fanTable;
  int fanJ0, fanQ0;
  int fanOk;               // the table belongs to this pixel's surface and this generator block
  // The pixel's first-bounce surface and incoming direction, in LDS (written once per pixel by lane 0).
  double *pixRec;
  static constexpr int kPxPos = 0, kPxNormal = 3, kPxBx = 6, kPxBy = 9, kPxRefl = 12, kPxCone = 13, kPxE = 14,
                       kPxD = 17, kPxDir = 20; // (kSeqPixRecDoubles doubles)
  int stackStride;         // levels per chain stack: chain c uses stack[c.slot * stackStride + level]
  unsigned long long hist; // levels consumed by this pass's committed sub-samples (6-bit fields 1..9)
  int m1;                  // the most frequent of them (refreshed once per pixel)

  // (kernel prologue) `area`: this master's kSeqFanBytes of LDS
  __device__ __forceinline__ void pairInit(const TraceParams &tp, unsigned char *area, int depthSlots, bool isMaster) {
    stackStride = depthSlots;
    hist = 0;
    m1 = tp.maxDepth;
    fanTable = reinterpret_cast<FanEntry *>(area);
    pixRec = reinterpret_cast<double *>(fanTable + 64);
    (void)isMaster;
    fanOk = 0, fanJ0 = 0, fanQ0 = 0;
  }

  __device__ __forceinline__ void writeRay(int slot, d3 o, d3 d) {
    if ((threadIdx.x & 63) == 0) {
      double *po = slot ? cmd->o2 : cmd->o, *pd = slot ? cmd->d2 : cmd->d;
      po[0] = o.x, po[1] = o.y, po[2] = o.z;
      pd[0] = d.x, pd[1] = d.y, pd[2] = d.z;
    }
  }
  __device__ __forceinline__ void readRay(int slot, d3 &o, d3 &d) const {
    const double *po = slot ? cmd->o2 : cmd->o, *pd = slot ? cmd->d2 : cmd->d;
    o = mk(po[0], po[1], po[2]);
    d = mk(pd[0], pd[1], pd[2]);
  }
  // a finished chain's radiance waits for its commit where its ray used to be
  __device__ __forceinline__ void storeL(int slot, d3 L) {
    if ((threadIdx.x & 63) == 0) {
      double *po = slot ? cmd->o2 : cmd->o;
      po[0] = L.x, po[1] = L.y, po[2] = L.z;
    }
  }
  __device__ __forceinline__ d3 loadL(int slot) const {
    const double *po = slot ? cmd->o2 : cmd->o;
    return mk(po[0], po[1], po[2]);
  }
  __device__ __forceinline__ void pixRecStore(const Surface &s, d3 dirIn) {
    if ((threadIdx.x & 63) == 0) {
      double *r = pixRec;
      r[kPxPos] = s.pos.x, r[kPxPos + 1] = s.pos.y, r[kPxPos + 2] = s.pos.z;
      r[kPxNormal] = s.normal.x, r[kPxNormal + 1] = s.normal.y, r[kPxNormal + 2] = s.normal.z;
      r[kPxBx] = s.basis.x.x, r[kPxBx + 1] = s.basis.x.y, r[kPxBx + 2] = s.basis.x.z;
      r[kPxBy] = s.basis.y.x, r[kPxBy + 1] = s.basis.y.y, r[kPxBy + 2] = s.basis.y.z;
      r[kPxRefl] = s.reflectivity, r[kPxCone] = s.coneAngle;
      r[kPxE] = s.emission.x, r[kPxE + 1] = s.emission.y, r[kPxE + 2] = s.emission.z;
      r[kPxD] = s.diffuse.x, r[kPxD + 1] = s.diffuse.y, r[kPxD + 2] = s.diffuse.z;
      r[kPxDir] = dirIn.x, r[kPxDir + 1] = dirIn.y, r[kPxDir + 2] = dirIn.z;
    }
  }
  __device__ __forceinline__ void chainPush(const Chain &c, int level, d3 e, d3 dif, bool refl) {
    if ((threadIdx.x & 63) == 0) {
      Level lv;
      lv.emission = e;
      lv.diffuse = dif;
      lv.reflective = refl;
      stack[c.slot * stackStride + level] = lv;
    }
  }
  // (called while the workers search: the colours' fetch - triangle record -> material index ->
  // material, dependent round trips - is off the serial path, as with flushPending())
  __device__ __forceinline__ void chainFlush(Chain &c) {
    if (!(c.live & c.pend)) return;
    const double *r = tab.tri + static_cast<size_t>(c.pendIdx - p->nsph) * kTriCompactDoubles;
    const double *m = tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
    chainPush(c, c.pendLevel, ld3(m), ld3(m + 3), false);
    c.pend = 0;
  }
  // The chain has ended with radiance `L` at its innermost level: fold its stack (Scene.cpp:163-175).
  __device__ __forceinline__ d3 chainFinish(Chain &c, d3 L) {
    chainFlush(c);
    for (int i = c.nlev - 1; i >= 0; --i) {
      const Level lv = stack[c.slot * stackStride + i];
      L = uniformBool(lv.reflective) ? lv.emission + L : lv.emission + lv.diffuse * L;
    }
    c.done = 1;
    return L;
  }
  // The answer `k` to the chain's ray in flight: radianceChain()'s / chainMasterFrom()'s level, once.
  // IS_X: the frontier chain draws at ctx.pos (and may regenerate); a speculated chain at c.pos.
  // Returns true when the chain has ended: `Lout` is radiance(depth 1) of its sub-sample.
  template <bool IS_X>
  __device__ __forceinline__ bool chainAdvance(Chain &c, const HitKey &k, d3 &Lout) {
    const uint32_t idx = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(k.idx)));
    const uint32_t pv = idx == kMiss ? 0u : idx + 1u;
    c.rays += 1u;
    c.s1 += pv;
    c.s2 += c.rays * pv;
    if (idx == kMiss) { // Scene.cpp:131-133
      Lout = chainFinish(c, envColour);
      return true;
    }
    HitKey ks = k;
    ks.idx = idx;
    if (c.depth + 1 >= p->maxDepth) { // last level: see radianceChain()
      if (IS_X) skip3(); else c.pos += 3;
      c.levels += 1;
      Lout = chainFinish(c, emissionAt(ks));
      return true;
    }
    d3 o, d;
    readRay(c.slot, o, d);
    const uint32_t nsph = p->nsph, ntri = p->ntri;
    const int q = IS_X ? pos : c.pos;
    const bool inBlock = IS_X ? q + 3 <= kMtDoubles : true;
    const bool isTri = (idx - nsph) < ntri; // unsigned: spheres fail
    bool handled = false;
    if (isTri & inBlock) {
      // the common level (chainMasterFrom): a triangle, the diffuse lobe decided from the record's lobe
      // threshold as lane-mask logic, all LDS operands waited for once
      const double *r = tab.tri + static_cast<size_t>(idx - nsph) * kTriCompactDoubles;
      d3 n = ld3(r), bx = ld3(r + 3), by = ld3(r + 6);
      double thr = r[kTriLobeThreshold];
      double pd = sh->canon[q + 2];
      const double *hm = sh->hemi[q];
      d3 local = mk(hm[0], hm[1], hm[2]);
      asm volatile("" : "+v"(n.x), "+v"(bx.x), "+v"(by.x), "+v"(thr), "+v"(pd), "+v"(local.x)); // one wait
      const bool backfacing = uniformBool(k.det < kEpsilon); // Scene.cpp:107
      const double ndotd = dot(n, d);
      const double cosThetaI = backfacing ? ndotd : -ndotd;
      const unsigned long long mNotRefl = __builtin_amdgcn_ballot_w64(!(pd < thr));
      const unsigned long long mPlain = __builtin_amdgcn_ballot_w64(thr >= 0.0);
      const unsigned long long mCos = __builtin_amdgcn_ballot_w64(cosThetaI >= 1e-3);
      const unsigned long long mPos = __builtin_amdgcn_ballot_w64(pd > 0.0);
      if ((mNotRefl & (mPlain | (mCos & mPos))) != 0) {
        if (IS_X) pos += 3, words += 6; else c.pos += 3;
        Basis b;
        b.x = bx, b.y = by, b.z = n;
        const double sgn = backfacing ? -1.0 : 1.0;
        const d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
        writeRay(c.slot, o + d * k.t, nd);
        c.pend = 1, c.pendLevel = c.nlev, c.pendIdx = idx;
        c.nlev += 1;
        handled = true;
      }
    }
    if (!handled) { // sphere, reflective lobe, Fresnel evaluation, draws straddling a regeneration
      const Surface s = surfaceAt(ks, o, d, false);
      d3 nd;
      bool refl;
      if (IS_X) {
        refl = scatterChain(s, d, nd);
      } else {
        refl = scatterChainAt(q, s, d, nd);
        c.pos += 3;
      }
      chainPush(c, c.nlev, s.emission, s.diffuse, refl);
      c.nlev += 1;
      writeRay(c.slot, s.pos, nd);
    }
    c.depth += 1;
    c.levels += 1;
    return false;
  }

  // Fills the table for sub-samples j0 .. j0 + 3 from block position q0 on (every lane its own pair).
  // Only the diffuse lobe is tabulated (hemisphereSample: the same function scatter() calls); an entry
  // whose draws choose the reflective lobe is marked unusable and that chain start evaluates
  // scatter() itself.
  __device__ __forceinline__ void fanBuild(int j0, int q0, int nSub, int vShift) {
    const int lane = threadIdx.x & 63, a = lane >> 4, m = lane & 15;
    const int sub = j0 + a, q = q0 + 3 * (a + m);
    const bool ok = (sub < nSub) & (q + 3 <= kMtDoubles) & (q >= 0);
    const int qq = ok ? q : 0;
    const double xu = sh->canon[qq], xv = sh->canon[qq + 1], pd = sh->canon[qq + 2];
    const double *r = pixRec;
    Basis basis;
    basis.z = ld3(r + kPxNormal), basis.x = ld3(r + kPxBx), basis.y = ld3(r + kPxBy);
    const double reflectivity = r[kPxRefl];
    const int fbV = p->fbV;
    const int uS = p->vPow2 ? sub >> vShift : sub / fbV, vS = sub - uS * fbV;
    double u, v;
    stratify(*p, uS, vS, xu, xv, p->invU, p->invV, u, v);
    const d3 nd = hemisphereSample<kScalarConsts>(basis, u, v); // Scene.cpp:169-175
    FanEntry e;
    e.dir[0] = nd.x, e.dir[1] = nd.y, e.dir[2] = nd.z;
    e.ok = (ok & !(pd < reflectivity)) ? 1u : 0u; // Scene.cpp:163
    e.pad = 0;
    fanTable[lane] = e;
    fanJ0 = j0, fanQ0 = q0, fanOk = 1;
#if PTW_PROFILE_PHASES
    fanBuilds++;
#endif
  }
  // The first-bounce ray of sub-sample `sub` started at block position q, from the table: origin and
  // direction.  False: not tabulated (out of range, another block, or the reflective lobe).
  __device__ __forceinline__ bool fanLookup(int sub, int q, d3 &origin, d3 &nd) const {
    const int a = sub - fanJ0, r = q - fanQ0 - 3 * a;
    const int m = r / 3;
    if (!((fanOk != 0) & (a >= 0) & (a < 4) & (r >= 0) & (m < 16) & (m * 3 == r))) return false;
    const FanEntry *e = fanTable + (16 * a + m);
    const double x = e->dir[0], y = e->dir[1], z = e->dir[2];
    const uint32_t ok = e->ok;
    origin = ld3(pixRec + kPxPos);
    if (!uniformBool(ok != 0u)) return false;
    nd = mk(x, y, z);
    return true;
  }

  // The PAIR master's whole pass: Scene::render's pixel loop (Scene.cpp:211-217) and radiance0() as a
  // machine that does ONE thing per workgroup barrier.  Barrier b is followed by the workers' search of
  // the rays in command slot b & 1 of BOTH masters (one request, two rays: workerLoop) - and by each
  // master's work on its OTHER slot, c = (b + 1) & 1, whose answers that barrier has just made visible:
  // advance the chain that lives there (or take the primary ray's hit), commit what has ended, put the
  // slot's next ray in place - a chain's next ray, a new chain, the next pixel's primary ray - and say
  // so in bit c of the command's mask, all before barrier b + 1, after which that slot is searched.
  // So a chain advances one ray per two barriers, a master with two chains in flight is busy in every
  // tick, both masters work at the same time (on different SIMDs), and the six worker waves answer one
  // two-ray request per tick.  Round 4's lock step had one master shading while the other one's ray was
  // searched: a ray per master and 4.9 k cycles on suzanne, where a tick of this form is the longer of
  // one two-ray search and one answer's worth of shading.
  __device__ __forceinline__ void pairRun(const TraceParams &tp, double *myStage, uint32_t *wordsOut, uint32_t *picksOut,
                                          int pass) {
    const int lane = threadIdx.x & 63;
    const int width = tp.width;
    const bool lens = uniformBool(cam->aperture_radius != 0);
    const int nSub = tp.fbU * tp.fbV, fbV = tp.fbV;
    const int vShift = fbV > 0 ? 31 - __builtin_clz(static_cast<unsigned>(fbV)) : 0;
    const int maxDepth = tp.maxDepth;
    const uint32_t pixCount = tp.pixCount;
    Chain X, Y;
    X.slot = 0, X.sub = 0, X.start = 0, X.pos = 0, X.depth = 0, X.levels = 0, X.nlev = 0, X.refl0 = 0, X.live = 0, X.done = 0;
    X.pend = 0, X.pendLevel = 0, X.pendIdx = 0, X.rays = 0, X.s1 = 0, X.s2 = 0;
    Y = X;
    uint32_t i = 0;     // pixel of the band
    int phase = 0;      // 0: between pixels, 1: the primary ray is in flight (slot primSlot), 2: the fan-out
    int primSlot = 0;
    uint32_t mask = 0;  // command slots that hold a ray to be searched
    int j = 0;          // sub-samples of the pixel committed
    d3 result = mk(0, 0, 0);
#if PTW_PROFILE_PHASES
    unsigned long long tBusy = 0, tWait = 0, tSlack = 0, nTicks = 0, nIdle = 0;
    // by what the tick did: 0 nothing, 1 primary hit, 2 primary miss, 3 X ended, 4 X goes on, 5 Y ended, 6 Y goes on,
    // 7 a chain start only; (six scalars each, not arrays: see mprof)
    unsigned long long k0n = 0, k1n = 0, k2n = 0, k3n = 0, k4n = 0, k5n = 0, k6n = 0, k7n = 0;
    unsigned long long k0t = 0, k1t = 0, k2t = 0, k3t = 0, k4t = 0, k5t = 0, k6t = 0, k7t = 0;
    unsigned long long tAnswer = 0, tRefill = 0;
#endif

    // one sub-sample's contribution (Scene.cpp:163-175 at depth 0), in sub-sample order
    auto commit = [&](const Chain &c, d3 L) {
      const d3 e0 = ld3(pixRec + kPxE), d0 = ld3(pixRec + kPxD);
      result = result + (c.refl0 ? e0 + L : e0 + d0 * L);
      rays += c.rays;
      pickS2 += pickN * c.s1 + c.s2;
      pickN += c.rays;
      hist += 1ull << (6 * c.levels);
      if (hist & 0x0820820820820820ull) hist = (hist >> 1) & 0x07df7df7df7df7dfull;
      ++j;
    };
    // ArrayOutput::addSamples of this pass's buffer (Scene.cpp:216)
    auto finishPixel = [&](d3 L) {
      if (lane == 0) {
        const uint32_t pix = tp.pixBegin + i;
        myStage[i * 3 + 0] = L.x;
        myStage[i * 3 + 1] = L.y;
        myStage[i * 3 + 2] = L.z;
        if (wordsOut) wordsOut[static_cast<size_t>(pass) * tp.npix + pix] = words;
        if (picksOut) picksOut[static_cast<size_t>(pass) * tp.npix + pix] = pickS2;
      }
      ++i;
      phase = 0;
    };
    // Camera::randomRay of the next pixel into slot c (Scene.cpp:214)
    auto startPrimary = [&](int c) {
      const uint32_t pix = tp.pixBegin + i;
      const int px = static_cast<int>(pix % static_cast<uint32_t>(width));
      const int py = static_cast<int>(pix / static_cast<uint32_t>(width));
      words = 0;
      pickReset();
      double r0, r1, r2 = 0, r3 = 0;
      if (lens) {
        draw4(r0, r1, r2, r3);
      } else {
        r0 = draw();
        r1 = draw();
      }
      d3 o, d;
      cameraRay<true>(*cam, px, py, r0, r1, r2, r3, o, d);
      writeRay(c, o, d);
      primSlot = c;
      phase = 1;
    };

    if (pixCount > 0) {
      startPrimary(0);
      mask = 1u;
    }
    if (lane == 0) cmd->nrays = mask;
    unsigned b = 0;
    bool more = pixCount > 0;
#if PTW_PROFILE_PHASES
    unsigned long long tMark = __builtin_amdgcn_s_memtime();
#endif
    for (; more; ++b) {
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long tIn = __builtin_amdgcn_s_memtime();
#endif
      ldsBarrier(); // barrier b: slot b & 1 is searched now; the answers of the other slot are in
#if PTW_PROFILE_PHASES
      const unsigned long long tOut = __builtin_amdgcn_s_memtime();
      tWait += tOut - tIn, nTicks++;
#endif
      const int c = static_cast<int>((b + 1u) & 1u);
      bool busy = false;
#if PTW_PROFILE_PHASES
      int kind = 0;
#endif
      // ---- the answers of slot c ----
      if ((phase == 1) & (primSlot == c) & (b > 0u)) { // the primary ray's hit
        busy = true;
        const HitKey k0 = pickPartials(c);
        rays++;
        pickNote(k0);
#if PTW_PROFILE_PHASES
        kind = k0.idx == kMiss ? 2 : 1;
#endif
        if (uniformBool(k0.idx == kMiss)) {
          finishPixel(ld3(tp.env)); // Scene.cpp:131-133
        } else {
          d3 o, d;
          readRay(c, o, d);
          const Surface s0 = surfaceAt(k0, o, d);
          if (tp.preview) {
            finishPixel(s0.diffuse); // Scene.cpp:137-138
          } else {
            pixRecStore(s0, d);
            result = mk(0, 0, 0);
            j = 0;
            int best = maxDepth, bestN = 0; // the guess of this pixel: the most frequent number of levels so far
#pragma unroll
            for (int f = 1; f <= 9; ++f) {
              const int n = static_cast<int>(hist >> (6 * f)) & 63;
              const bool top = n > bestN;
              best = top ? f : best;
              bestN = top ? n : bestN;
            }
            m1 = best;
            fanOk = 0;
            if (pos + 3 <= kMtDoubles) fanBuild(0, pos, nSub, vShift); // the first sub-samples' scatters in one go
            phase = 2;
          }
        }
      } else if ((X.live != 0) & (X.slot == c)) { // the frontier chain's answer
        busy = true;
        d3 LX;
#if PTW_PROFILE_PHASES
        kind = 4;
#endif
        if (chainAdvance<true>(X, pickPartials(c), LX)) {
#if PTW_PROFILE_PHASES
          kind = 3;
#endif
          commit(X, LX);
          if ((Y.live != 0) & (Y.start == pos)) { // the guess held: Y is the sub-sample at the frontier -
            X = Y;                                 // with whatever it has traced since, in its own slot
            words += 2u * static_cast<unsigned>(Y.pos - Y.start);
            pos = Y.pos;
            Y.live = 0;
            if (X.done != 0) { // (it had ended already)
              commit(X, loadL(X.slot));
              X.live = 0;
            }
          } else {
            X.live = 0, Y.live = 0; // (a wrong guess costs the workers' time, nothing else)
          }
        } else if ((Y.live != 0) & (Y.start < pos)) {
          Y.live = 0; // X has gone past the position Y was started at: the guess is known to be wrong
        }
      } else if ((Y.live != 0) & (Y.slot == c) & (Y.done == 0)) { // the speculated chain's answer
        busy = true;
        d3 LY;
#if PTW_PROFILE_PHASES
        kind = 6;
#endif
        if (chainAdvance<false>(Y, pickPartials(c), LY)) {
          storeL(Y.slot, LY);
#if PTW_PROFILE_PHASES
          kind = 5;
#endif
        }
      }
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long tAns = __builtin_amdgcn_s_memtime();
      tAnswer += tAns - tOut;
#endif
      // ---- the pixel's last sub-sample committed? ----
      if ((phase == 2) & (X.live == 0) & (j >= nSub)) finishPixel(result * tp.invFirstBounce); // Vec3::operator/(double)
      // ---- slot c's next ray ----
      uint32_t bit = 0;
      if (phase == 0) {
        if (i < pixCount) {
          startPrimary(c);
          bit = 1u;
          busy = true;
        }
      } else if (phase == 2) {
        const bool xHere = (X.live != 0) & (X.slot == c), yHere = (Y.live != 0) & (Y.slot == c);
        if (xHere | yHere) {
          bit = xHere ? 1u : (Y.done == 0 ? 1u : 0u);
        } else {
          // a free slot: X at the frontier if there is none, else Y behind it at the guessed position
          const bool forX = X.live == 0;
          bool start = false;
          int sub = 0, q = 0;
          if (forX) {
            start = j < nSub;
            sub = j, q = pos;
          } else if ((Y.live == 0) & (X.sub + 1 < nSub)) {
            const int g = m1 > X.levels ? m1 : X.levels;
            sub = X.sub + 1, q = pos + 3 * (g - X.levels);
            start = q + 3 * maxDepth <= kMtDoubles; // a speculated chain never leaves the block
          }
          if (start) {
            busy = true;
            d3 origin, nd;
            int refl = 0;
            if (fanLookup(sub, q, origin, nd)) { // tabulated (the diffuse lobe): the three draws are consumed
              if (forX) pos += 3, words += 6;
#if PTW_PROFILE_PHASES
              fanHits++;
#endif
            } else {
              double xu, xv, pd;
              if (forX) {
                draw3(xu, xv, pd); // the frontier's draws (may regenerate: no Y exists then)
              } else {
                xu = sh->canon[q], xv = sh->canon[q + 1], pd = sh->canon[q + 2];
              }
              const int uS = tp.vPow2 ? sub >> vShift : sub / fbV, vS = sub - uS * fbV;
              double u, v;
              stratify(tp, uS, vS, xu, xv, tp.invU, tp.invV, u, v);
              const double *r = pixRec;
              Surface s0;
              s0.pos = ld3(r + kPxPos), s0.normal = ld3(r + kPxNormal);
              s0.basis.x = ld3(r + kPxBx), s0.basis.y = ld3(r + kPxBy), s0.basis.z = s0.normal;
              s0.reflectivity = r[kPxRefl], s0.coneAngle = r[kPxCone];
              origin = s0.pos;
              refl = scatter(*this, s0, ld3(r + kPxDir), u, v, pd, nd) ? 1 : 0;
#if PTW_PROFILE_PHASES
              fanMisses++;
#endif
            }
            Chain n;
            n.slot = c;
            n.sub = sub, n.start = q, n.pos = q + 3, n.depth = 1, n.levels = 1, n.nlev = 0;
            n.refl0 = refl, n.live = 1, n.done = 0, n.pend = 0, n.pendLevel = 0, n.pendIdx = 0;
            n.rays = 0, n.s1 = 0, n.s2 = 0;
            writeRay(c, origin, nd);
            if (forX) X = n; else Y = n;
            bit = 1u;
          }
        }
      }
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tRefill += __builtin_amdgcn_s_memtime() - tAns;
      if (kind == 0 && busy) kind = 7;
#endif
      mask = (mask & ~(1u << c)) | (bit << c);
      more = (phase != 0) | (i < pixCount);
      if (lane == 0) {
        cmd->nrays = mask;
        if (!more) cmd->op = b + 1u; // no rays from this master as of the next barrier
      }
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long tPub = __builtin_amdgcn_s_memtime();
      tBusy += tPub - tOut, nIdle += busy ? 0 : 1;
      const unsigned long long dtk = tPub - tOut;
      if (kind == 0) k0n++, k0t += dtk;
      if (kind == 1) k1n++, k1t += dtk;
      if (kind == 2) k2n++, k2t += dtk;
      if (kind == 3) k3n++, k3t += dtk;
      if (kind == 4) k4n++, k4t += dtk;
      if (kind == 5) k5n++, k5t += dtk;
      if (kind == 6) k6n++, k6t += dtk;
      if (kind == 7) k7n++, k7t += dtk;
#endif
      // ---- what is left of the tick: deferred stack entries, and the scatters of the sub-samples that
      // start next if they are not tabulated ----
      chainFlush(X);
      chainFlush(Y);
      if (phase == 2) {
        const int ns = ((Y.live != 0) ? Y.sub : ((X.live != 0) ? X.sub : j - 1)) + 1;
        const int qn = (Y.live != 0) ? Y.pos : pos;
        const int a = ns - fanJ0, r = qn - fanQ0 - 3 * a;
        const bool covered = (fanOk != 0) & (a >= 0) & (a <= 2) & (r >= 0) & (r < 24);
        if ((ns < nSub) & !covered & (qn + 3 <= kMtDoubles)) fanBuild(ns, qn, nSub, vShift);
      }
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tSlack += __builtin_amdgcn_s_memtime() - tPub;
#endif
      (void)busy;
    }
    // keep the cadence until the other master is done too (workerLoop)
    if (pixCount == 0 && lane == 0) cmd->op = 0u;
    for (unsigned n = b;; ++n) {
      ldsBarrier();
      if (allCmds[0].op <= n && allCmds[1].op <= n) break;
    }
#if PTW_PROFILE_PHASES
    if (pass == 0 && lane == 0) {
      const double px = static_cast<double>(pixCount), t = static_cast<double>(nTicks);
      printf("PAIR master (pass 0): %.2f ticks per sample (%.2f with nothing to do), %.2f rays, %.2f table builds, chain starts "
             "%.2f tabulated + %.2f evaluated | per tick: work (barrier -> mask published)=%.0f slack work=%.0f barrier wait=%.0f "
             "total=%.0f\n",
             t / px, nIdle / px, rays / px, fanBuilds / px, fanHits / px, fanMisses / px, tBusy / t, tSlack / t, tWait / t,
             (__builtin_amdgcn_s_memtime() - tMark) / t);
      auto av = [](unsigned long long a, unsigned long long n) { return n ? static_cast<double>(a) / n : 0.0; };
      printf("PAIR master work by tick kind, cycles (ticks per sample): nothing %.0f (%.2f) | primary hit %.0f (%.2f) miss %.0f (%.2f) | "
             "X ended %.0f (%.2f) goes on %.0f (%.2f) | Y ended %.0f (%.2f) goes on %.0f (%.2f) | start only %.0f (%.2f) || answers "
             "part %.0f refill part %.0f per tick\n",
             av(k0t, k0n), k0n / px, av(k1t, k1n), k1n / px, av(k2t, k2n), k2n / px, av(k3t, k3n), k3n / px, av(k4t, k4n), k4n / px,
             av(k5t, k5n), k5n / px, av(k6t, k6n), k6n / px, av(k7t, k7n), k7n / px, tAnswer / t, tRefill / t);
    }
#endif
  }

  // The PAIR form of workerLoop(): barrier n is followed by the search of the rays in command slot n & 1 of
  // BOTH masters (pairRun): one request, up to two rays, tested against the resident triangles in one pass.
  // A command's `op` holds the barrier index from which its master has no more rays (kCmdLive while it
  // has): a value that reads the same whenever it is looked at, so all waves leave after the same barrier.
  __device__ __forceinline__ void workerLoopPair(unsigned long long &nreq, unsigned long long &nreq2) {
    (void)nreq, (void)nreq2;
    // Barrier n is followed by the search of the rays in command slot n & 1 of BOTH masters (pairRun):
    // one request, up to two rays, tested against the resident triangles in one pass.  A command's
    // `op` holds the barrier index from which its master has no more rays (kCmdLive while it has): a
    // value that reads the same whenever it is looked at, so all waves leave after the same barrier.
    for (unsigned n = 0;; ++n) {
      ldsBarrier();
      const SeqCommand &cA = allCmds[0], &cB = allCmds[1];
      const uint32_t opA = cA.op, opB = cB.op;
      if (opA <= n && opB <= n) break;
      const int c = static_cast<int>(n & 1u);
      const uint32_t live = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(
          ((opA > n ? (cA.nrays >> c) & 1u : 0u)) | ((opB > n ? (cB.nrays >> c) & 1u : 0u) << 1))));
      if (live == 0u) continue;
      const double *oA = c ? cA.o2 : cA.o, *dA = c ? cA.d2 : cA.d;
      const double *oB = c ? cB.o2 : cB.o, *dB = c ? cB.d2 : cB.d;
      PartialHit *outA = partials + (0 * 2 + c) * WAVES + (tid >> 6);
      PartialHit *outB = partials + (1 * 2 + c) * WAVES + (tid >> 6);
      if (live == 3u) {
        // both rays into scalar registers: they are operands of every test of the search
        const d3 rOA = mk(readFirstLane(oA[0]), readFirstLane(oA[1]), readFirstLane(oA[2]));
        const d3 rDA = mk(readFirstLane(dA[0]), readFirstLane(dA[1]), readFirstLane(dA[2]));
        const d3 rOB = mk(readFirstLane(oB[0]), readFirstLane(oB[1]), readFirstLane(oB[2]));
        const d3 rDB = mk(readFirstLane(dB[0]), readFirstLane(dB[1]), readFirstLane(dB[2]));
        HitKey fA, fB;
        localNearest2(rOA, rDA, rOB, rDB, fA, fB);
        if ((tid & 63) == 0) {
          PartialHit ph;
          ph.t = fA.t, ph.pad = 0, ph.idxSign = packAnswer(fA);
          *outA = ph;
          ph.t = fB.t, ph.idxSign = packAnswer(fB);
          *outB = ph;
        }
#if PTW_PROFILE_PHASES
        nreq2++;
#endif
      } else {
        const bool second = live == 2u;
        const double *ro = second ? oB : oA, *rd = second ? dB : dA;
        const HitKey found = localNearest(mk(ro[0], ro[1], ro[2]), mk(rd[0], rd[1], rd[2]));
        if ((tid & 63) == 0) {
          PartialHit ph;
          ph.t = found.t, ph.pad = 0, ph.idxSign = packAnswer(found);
          *(second ? outB : outA) = ph;
        }
      }
#if PTW_PROFILE_PHASES
      nreq++;
#endif
    }
  }

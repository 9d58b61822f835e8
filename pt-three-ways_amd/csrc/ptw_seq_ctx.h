// ptw_seq_ctx.h - SEQUENTIAL policy: the per-pass execution context (one workgroup per pass) shared by the
// single-wave, worker-wave and speculative kernels.  Internal to csrc/.
#pragma once
#include "ptw_radiance.h"

namespace ptw {
using namespace ptwd;
namespace {

// -----------------------------------------------------------------------------------------
// SEQUENTIAL policy context: one workgroup (WAVES x 64 lanes) per pass.
// -----------------------------------------------------------------------------------------
struct SeqShared {
  uint32_t mt[kMtWords];
  double canon[kMtDoubles];
  // For every position q of the block: the local cosine-hemisphere direction that
  // hemisphereSample() builds from (u, v) = (canon[q], canon[q + 1]) before the basis transform:
  // (cos(2 pi u) sqrt(v), sin(2 pi u) sqrt(v), sqrt(1 - v)).  It depends only on the draws, so all
  // positions are evaluated 64 lanes at a time when the block is generated, instead of one
  // sincos + two square roots on the serial path of every bounce.
  double hemi[kMtDoubles][3];
};

// A worker wave's answer: its nearest hit.  16 bytes, one ds_read_b128 for the master: the distance
// and the combined primitive index with the only fact ever used of the determinant - the sign test
// `det < epsilon` of Scene.cpp:107 - in bit 0 (kMiss, all ones, with t = +inf for "nothing hit").
struct alignas(16) PartialHit {
  double t;
  uint32_t idxSign; // combined index << 1 | (det < epsilon); kMiss (all ones) with t = +inf for "nothing"
  uint32_t pad;
};
__device__ __forceinline__ uint32_t packAnswer(const HitKey &k) {
  return k.idx == kMiss ? kMiss : ((k.idx << 1) | (k.det < kEpsilon ? 1u : 0u));
}
// The nearest of n answers with the reference's tie-break (strictly nearer wins, an exact tie goes to
// the lower combined index: Scene.cpp:31,95,118), computed by every lane alike - no cross-lane traffic:
// the minimum distance, then the lowest packed index among the answers that have it (the packing
// keeps the order of the indices; a miss is +inf / all ones and loses against everything).
template <int N>
__device__ __forceinline__ HitKey pickOfAnswers(const PartialHit (&ph)[N]) {
  HitKey key;
  double bt = ph[0].t;
#pragma unroll
  for (int w = 1; w < N; ++w) bt = vmin64(bt, ph[w].t);
  uint32_t bw = kMiss;
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const uint32_t c = ph[w].t == bt ? ph[w].idxSign : kMiss;
    bw = c < bw ? c : bw;
  }
  key.t = bt;
  key.idx = bw == kMiss ? kMiss : (bw >> 1);
  key.det = (bw & 1u) ? -1.0 : 1.0; // (only its sign test is ever used)
  return key;
}
// Behind the answers: the masters' commands (128 bytes each: two rays + the request word), then 8 bytes
// per worker wave and ray (pickNearest's LDS atomic).
constexpr size_t kSeqCmdBytes = 384;
constexpr size_t kSeqMinSlotOffset = 256; // into the command area; [8 waves][2 rays] x 8 bytes
// Worker-wave kernels: a copy of the camera in LDS.  As part of the kernel argument its 36 dwords sit in
// scalar registers the master's loop has no room for: they were spilled to vector-register lanes and read
// back for every pixel; from LDS the camera ray reads them with one wait.
constexpr size_t kSeqCamBytes = (sizeof(ptw_camera) + 63) & ~static_cast<size_t>(63);

// Master -> worker request of the multi-wave sequential kernels: one ray.  (128 bytes: the second half held
// the second ray of round 5's paired requests, LAB.md; the layout is kept.)
constexpr uint32_t kCmdTrace = 1, kCmdExit = 2;
constexpr uint32_t kCmdLive = 0xffffffffu; // two masters: this master still has rays
struct alignas(16) SeqCommand {
  double o[3], d[3];   // the ray
  double unused[6];
  uint32_t op;         // one master: kCmdTrace / kCmdExit; two masters: see workerLoop
  uint32_t pad[7];     // 128 bytes
};
static_assert(sizeof(SeqCommand) == 128, "SeqCommand layout");

// std::mt19937 regeneration (the "twist") + tempering + generate_canonical for all 312
// doubles, by the 64 lanes of one wave.  Chunks of 64 consecutive k are processed in order;
// inside a chunk every lane reads its inputs, waveSync(), then writes, waveSync() - the
// fences keep the compiler from reordering one lane's loads across another lane's stores.
// Kept out of line: it runs once per 312 draws and would otherwise be cloned into every
// draw() site.
__device__ __forceinline__ void fillHemiTable(SeqShared *sh, int lane) {
  for (int q = lane; q + 1 < kMtDoubles; q += 64) {
    const double u = sh->canon[q], v = sh->canon[q + 1];
    const double theta = (2 * kPi) * u;
    const double radius = sqrtPos(v);
    double sn, cs;
    sinCos<true>(theta, sn, cs);
    sh->hemi[q][0] = cs * radius;
    sh->hemi[q][1] = sn * radius;
    sh->hemi[q][2] = sqrtPos(1 - v);
  }
}

__device__ __noinline__ void mtRegenerateWave(SeqShared *sh, int lane) {
  uint32_t *x = sh->mt;
  mtTwistWave(x, lane);
  for (int i = lane; i < kMtDoubles; i += 64)
    sh->canon[i] = canonicalFromWords(mtTemper(x[2 * i]), mtTemper(x[2 * i + 1]));
  waveSync();
  fillHemiTable(sh, lane);
  waveSync();
}

// Shading tables in LDS (filled once per launch): compact triangle records, the material
// table and the sphere records.  A hit costs one LDS round trip instead of a scalar + vector
// global fetch on the critical path of every ray.
struct SeqTables {
  const double *tri;     // [ntri][kTriCompactDoubles]   (LDS or global)
  const double *mat;     // [nmat][kMatDoubles]           (LDS or global)
  const SphereRec *sph;  // [nsph]                        (LDS or global)
};

// Layout of the per-lane shading record of the REG path (doubles).
constexpr int kRecEmission = 0, kRecDiffuse = 3, kRecDoubles = 6;

// PRE (worker-wave kernels, PTW_ACCEL_PREFILTER under the SEQUENTIAL policy - a separate, separately reported
// mode): the worker lanes hold their triangles in fp32 - two slots per register pair - and look at them with the
// conservative prefilter of host/prefilter.h (two triangles per packed instruction: 38 VALU instructions per PAIR
// of slots against 50 per slot); the reference's fp64 test runs only for the slots whose rejection fp32 cannot
// prove, on v0 / e1 / e2 fetched from memory by the lanes that need them.  Same hits, bit for bit.
template <int SLOTS, int WAVES, bool LDS_TABLES, bool REG = false, bool SPEC = false, int MASTERS = 1, bool PICKS = true,
          bool PRE = false, bool UNIT = false>
struct SeqCtx {
  static_assert(!UNIT || (WAVES > 1 && !PRE), "the unit-level u-first early-out exists for the plain worker-wave kernels");
  static_assert(!PRE || (WAVES > 1 && !REG && !SPEC), "the prefilter form exists for the worker-wave kernels");
  static constexpr int kPairs = (SLOTS + 1) / 2;
  Float2 pv0x[kPairs], pv0y[kPairs], pv0z[kPairs], pe1x[kPairs], pe1y[kPairs], pe1z[kPairs], pe2x[kPairs], pe2y[kPairs],
      pe2z[kPairs], pea[kPairs], peb[kPairs]; // PRE only (never touched otherwise)
  const float *triPacked;                       // PRE: [(ntri + 1) / 2][22] floats (host/prefilter.h)
  // REG (single wave, one triangle per lane, at most 127 primitives, maxDepth <= 9): every lane
  // also keeps the emission and diffuse colour of its triangle in registers, and the (E, T)
  // stack is one byte per level (combined primitive index + lobe flag) in a scalar register
  // pair.  Pushing a level is three scalar instructions; folding one fetches the colours from
  // the owner lane with v_readlane (only the diffuse colour when the emission is zero): no LDS
  // traffic, and with one wave per SIMD nothing would hide an LDS wait.  (Fetching the whole
  // surface record that way was measured and is slower: one ds_read_b128 moves what four
  // v_readlane do.)
  static_assert(!REG || (SLOTS == 1 && WAVES == 1), "REG needs one wave and one triangle per lane");
  double rec[kRecDoubles]; // REG only (never touched otherwise, so it costs nothing there)
  unsigned long long stackBits; // REG: level i in bits [8i, 8i+8): combined index | lobe << 7
  unsigned long long emissiveMask; // REG: lanes whose triangle has a non-zero emission

  // WAVES == 1: one wave does everything.  WAVES > 1: WAVES worker waves hold the primitives
  // and one extra master wave (wave 0, no resident primitives) runs the path logic.
  static constexpr int kThreads = 64 * WAVES;                        // lanes that hold primitives
  // WAVES > 1: while the workers search a ray the master has nothing to do.  Most sub-samples of
  // an open scene end with a ray that leaves it, and then the next sub-sample's first-bounce
  // scatter (one sincos and two square roots on the serial path) starts at the stream position
  // the master is looking at right now: it is evaluated in that idle time, and taken if the
  // position still matches when the next sub-sample starts (lookAhead / takeLookAhead).
  static constexpr bool kLookAhead = WAVES > 1;
  // sincos constants in scalar registers (ptw_device.h, sconst()): the two-master kernels, whose
  // master path is short of vector registers
  static constexpr bool kScalarConsts = MASTERS == 2;
  // two masters per workgroup: radiance0 and the chain use chainMaster / chainMasterFrom
  static constexpr bool kMasterChain = WAVES > 1 && MASTERS == 2;
  // worker waves that share their SIMD with a master wave (the others sit two to a SIMD among themselves)
  static constexpr int kSideB = MASTERS;
  Surface laSurf;  // look-ahead inputs: the first-bounce surface, the incoming direction (set once per
  d3 laDir;        // pixel, before the fan-out: inside its loop they are the caller's own values), ...
  double laInvU, laInvV;
  int laU, laV;    // ... the stratum of the next sub-sample
  bool laArmed;
  int laMisses;    // look-aheads in a row that were not taken (closed scenes: nearly all) ...
  unsigned laTick; // ... after two of them only every eighth sub-sample tries again
  int laPos;       // stream position the result was evaluated for (-1: none)
  d3 laOut;
  bool laRefl;
  // WAVES > 1: the (E, T) stack entry of the level the master has just left is written while the
  // workers search the next ray (flushPending, called between the two barriers of intersect()):
  // its material fetch - two dependent LDS round trips for a triangle - is off the serial path.
  int pendKind;      // 0 none, 1 a triangle hit that took the diffuse lobe (colours still to be fetched)
  int pendLevel;
  uint32_t pendIdx;
  // workgroup size.  MASTERS == 2 (traceSequentialMM): two passes share the worker waves - wave
  // m < 2 runs pass 2 * blockIdx.x + m, and the workers alternate between the two masters' rays,
  // so that one master shades while the other one's ray is being searched.
  static constexpr int kBlock = WAVES == 1 ? 64 : 64 * (WAVES + MASTERS);
  static_assert(MASTERS == 1 || (MASTERS == 2 && WAVES > 1 && !REG && !SPEC), "two masters need worker waves");

  // per-lane resident triangles (SoA in registers)
  double v0x[SLOTS], v0y[SLOTS], v0z[SLOTS];
  double e1x[SLOTS], e1y[SLOTS], e1z[SLOTS];
  double e2x[SLOTS], e2y[SLOTS], e2z[SLOTS];
  // per-lane resident sphere (lane tid owns sphere tid when tid < nsph)
  double scx, scy, scz, sr2;
  bool hasSphere;

  const TraceParams *p;
  const ptw_camera *cam; // worker-wave kernels: the LDS copy of p->cam
  const double *triGeom;
  const SphereRec *spheresGlobal;
  const double *triCompactGlobal; // REG: source of the per-lane shading records
  const double *matTableGlobal;
  SeqTables tab;
  SeqShared *sh;
  Level *stack;          // this wave's private radiance stack in LDS
  PartialHit *partials;  // [WAVES] cross-wave exchange (WAVES > 1)
  SeqCommand *cmd;       // master -> workers (WAVES > 1); MASTERS == 2: this master's of allCmds[2]
  SeqCommand *allCmds;   // MASTERS == 2: both masters' commands
  unsigned tick;         // MASTERS == 2, lock step: workgroup barriers this wave has executed
  unsigned long long *minSlot; // worker waves: this wave's 2 x 8 bytes of LDS for pickNearest's atomic form
  int masterIndex;
  // Pick checksum (ptw_debug_options.d_picks): sum over the sample's intersect() calls r = 0, 1, ... of
  // (r + 1) * (combined index + 1), misses 0.  pickS1 / pickS2 accumulate sum (idx + 1) and
  // sum (i + 1) (idx + 1) over the calls since pickReset(); a segment that starts at call ordinal b
  // contributes b * S1 + S2 (the speculative kernels commit whole sub-samples at once).
  // PICKS = false compiles it out: the kernels whose wave has its SIMD to itself (WAVES == 1, the
  // speculative kernel) pay an issue slot for every instruction, so their shipped instantiation carries
  // none of this and a second one (launched when d_picks is set) does.
  bool picksOn;
  uint32_t pickS1, pickS2, pickN;
  int tid;               // index among the primitive-holding lanes (workers); master: lane id
  int pos;               // next canonical double in sh->canon (wave-uniform)
  d3 envColour;          // chainHot: the environment colour, kept in vector registers
  char *ringBase;        // SPEC: LDS address of ring slot 0
  unsigned ringOff;      // SPEC: 0 or kRingStride - the slot `pos` indexes
  unsigned words;        // RNG words consumed by the current sample
  unsigned long long rays;
  unsigned parity;
#if PTW_PROFILE_PHASES
  unsigned long long prof[12];
  unsigned long long mprof[6]; // master, inside intersect(): publish, wait B1, shadow work, wait B2, pick
  // master, OUTSIDE intersect(): cycles from one answer to the next published ray, by what the answer
  // was: [ray kind 0 primary / 1 first ray of a sub-sample / 2 deeper][0 hit / 1 miss]
  // (six scalars, not an array: a run-time index would put it in scratch memory and ruin the timing)
  unsigned long long g00, g01, g10, g11, g20, g21, n00, n01, n10, n11, n20, n21, lastExit;
  int rayKind, lastKind, lastMiss;
#endif

  // Triangle held in slot s of this lane.  WAVES == 1: slot-major (slot s of all lanes covers
  // triangles [64 s, 64 s + 64)).  WAVES > 1: a worker wave holds `myUnits` consecutive units of 64
  // triangles starting at unit `unitBase` - fewer for the waves that share their SIMD with another
  // worker, more for those beside a master (seqUnitSplit) - one unit per slot; slots from myUnits on
  // are empty and skipped with one scalar branch.
  int unitBase, myUnits;
  __device__ __forceinline__ uint32_t slotTriangle(int s) const {
    if (WAVES == 1) return static_cast<uint32_t>(s) * kThreads + static_cast<uint32_t>(tid);
    return static_cast<uint32_t>(unitBase + s) * 64u + static_cast<uint32_t>(tid & 63);
  }
  // triangles resident in the workers' registers (the rest is streamed: localNearest)
  __device__ __forceinline__ uint32_t residentTriangles() const {
    if (WAVES == 1) return static_cast<uint32_t>(kThreads) * SLOTS;
    constexpr int nB = kSideB, nA = WAVES - kSideB;
    return static_cast<uint32_t>((nA / 2) * (p->seqUnitsA + p->seqUnitsY) + nB * p->seqUnitsB) * 64u;
  }

  __device__ __forceinline__ void loadPrimitives() {
    const uint32_t ntri = p->ntri;
    if constexpr (PRE) {
      // fp32 copies from the pair records (triangle k: record k / 2, half k % 2); a slot without a triangle gets
      // E = -1: r = max(0 + E, E - 0) < 0 - "rejected for certain" at no extra instruction
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        float v[2][11];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = 2 * q + h;
          const uint32_t k = slotTriangle(s);
          const bool valid = s < SLOTS && k < ntri && s < myUnits;
          const float *rec = triPacked + 22 * static_cast<size_t>(valid ? (k >> 1) : 0u) + (valid ? (k & 1u) : 0u);
#pragma unroll
          for (int c = 0; c < 11; ++c) v[h][c] = valid ? rec[2 * c] : (c == 9 ? -1.0f : 0.0f);
        }
        pv0x[q] = (Float2){v[0][0], v[1][0]}, pv0y[q] = (Float2){v[0][1], v[1][1]}, pv0z[q] = (Float2){v[0][2], v[1][2]};
        pe1x[q] = (Float2){v[0][3], v[1][3]}, pe1y[q] = (Float2){v[0][4], v[1][4]}, pe1z[q] = (Float2){v[0][5], v[1][5]};
        pe2x[q] = (Float2){v[0][6], v[1][6]}, pe2y[q] = (Float2){v[0][7], v[1][7]}, pe2z[q] = (Float2){v[0][8], v[1][8]};
        pea[q] = (Float2){v[0][9], v[1][9]}, peb[q] = (Float2){v[0][10], v[1][10]};
      }
    }
#pragma unroll
    for (int s = 0; s < (PRE ? 0 : SLOTS); ++s) {
      // Branch-free on purpose: with an if/else the compiler sinks the two stores into one with a
      // runtime slot index, which sends the slot arrays to scratch memory.  An unused slot gets a
      // degenerate triangle (det == 0 -> always skipped); triGeom holds at least one record.
      const uint32_t k = slotTriangle(s);
      const bool valid = k < ntri && (WAVES == 1 || s < myUnits);
      const double *g = triGeom + 9 * static_cast<size_t>(valid ? k : 0u);
      v0x[s] = valid ? g[0] : 0.0, v0y[s] = valid ? g[1] : 0.0, v0z[s] = valid ? g[2] : 0.0;
      e1x[s] = valid ? g[3] : 0.0, e1y[s] = valid ? g[4] : 0.0, e1z[s] = valid ? g[5] : 0.0;
      e2x[s] = valid ? g[6] : 0.0, e2y[s] = valid ? g[7] : 0.0, e2z[s] = valid ? g[8] : 0.0;
    }
    if (REG) {
      // lanes without a triangle never win a hit, their record is never read
#pragma unroll
      for (int i = 0; i < kRecDoubles; ++i) rec[i] = 0.0;
      if (static_cast<uint32_t>(tid) < ntri) {
        const double *r = triCompactGlobal + static_cast<size_t>(tid) * kTriCompactDoubles;
        const double *m = matTableGlobal + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
#pragma unroll
        for (int i = 0; i < 6; ++i) rec[i] = m[i]; // emission, diffuse
      }
      const bool emissive = rec[kRecEmission] != 0.0 || rec[kRecEmission + 1] != 0.0 ||
                            rec[kRecEmission + 2] != 0.0;
      emissiveMask = __builtin_amdgcn_ballot_w64(emissive);
      stackBits = 0;
    }
    hasSphere = static_cast<uint32_t>(tid) < p->nsph;
    if (hasSphere) {
      const SphereRec &r = spheresGlobal[tid];
      scx = r.centre[0], scy = r.centre[1], scz = r.centre[2], sr2 = r.radiusSquared;
    } else {
      scx = scy = scz = sr2 = 0;
    }
  }

  // Only the master wave draws random numbers, so it regenerates on its own.
  __device__ __forceinline__ void regenerate() {
#if PTW_PROFILE_PHASES
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    mtRegenerateWave(sh, threadIdx.x & 63);
    laPos = -1; // positions of the old block mean nothing in the new one
#if PTW_PROFILE_PHASES
    prof[9] += __builtin_amdgcn_s_memtime() - t0;
#endif
  }

  // The same by the calling wave alone, without a workgroup barrier (several masters).
  __device__ __forceinline__ void rebuildCanonWave() {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < kMtDoubles; i += 64)
      sh->canon[i] = canonicalFromWords(mtTemper(sh->mt[2 * i]), mtTemper(sh->mt[2 * i + 1]));
    waveSync();
    fillHemiTable(sh, lane);
    waveSync();
  }

  // Rebuild canon[] from the current raw state without twisting (state resumed mid-block).
  __device__ __forceinline__ void rebuildCanon() {
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      for (int i = lane; i < kMtDoubles; i += 64)
        sh->canon[i] = canonicalFromWords(mtTemper(sh->mt[2 * i]), mtTemper(sh->mt[2 * i + 1]));
      waveSync();
      fillHemiTable(sh, lane);
    }
    __syncthreads();
  }

  // SPEC accessors: canon / hemi of the slot `pos` indexes
  __device__ __forceinline__ const double *ringCanon() const {
    return reinterpret_cast<const double *>(ringBase + ringOff);
  }
  __device__ __forceinline__ const double *ringHemi(int q) const {
    return reinterpret_cast<const double *>(ringBase + ringOff + kRingHemiOff) + 3 * q;
  }
  // SPEC: consume n draws (branch-free wrap into the other slot)
  __device__ __forceinline__ void advance(int n) {
    const int np = pos + n;
    const bool wrap = np >= kMtDoubles;
    pos = wrap ? np - kMtDoubles : np;
    ringOff = wrap ? ringOff ^ kRingStride : ringOff;
    words += 2 * n;
  }
  __device__ __forceinline__ void setStream(unsigned off, int q) {
    ringOff = off;
    pos = q;
  }

  __device__ __forceinline__ double draw() {
    if (SPEC) {
      const double v = ringCanon()[pos];
      advance(1);
      return v;
    }
    if (pos == kMtDoubles) {
      regenerate();
      pos = 0;
    }
    words += 2;
    return sh->canon[pos++];
  }
  // consecutive draws with one LDS round trip when they do not straddle a regeneration
  __device__ __forceinline__ void draw3(double &a, double &b, double &c) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      a = cn[0], b = cn[1], c = cn[2];
      advance(3);
      return;
    }
    if (pos + 3 <= kMtDoubles) {
      a = sh->canon[pos];
      b = sh->canon[pos + 1];
      c = sh->canon[pos + 2];
      pos += 3;
      words += 6;
    } else {
      a = draw();
      b = draw();
      c = draw();
    }
  }
  __device__ __forceinline__ void draw4(double &a, double &b, double &c, double &d) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      a = cn[0], b = cn[1], c = cn[2], d = cn[3];
      advance(4);
      return;
    }
    if (pos + 4 <= kMtDoubles) {
      a = sh->canon[pos];
      b = sh->canon[pos + 1];
      c = sh->canon[pos + 2];
      d = sh->canon[pos + 3];
      pos += 4;
      words += 8;
    } else {
      a = draw();
      b = draw();
      c = draw();
      d = draw();
    }
  }

  // The nearest of the lanes' candidates (t, combined index, determinant) with the reference's
  // tie-break, as a wave-uniform result.
  // `slot`: the calling wave's 8 bytes of LDS for the atomic form of the many-candidates case (worker
  // waves), nullptr for the DPP form.
  __device__ __forceinline__ static HitKey pickNearest(double bestT, uint32_t bestIdx, double bestDet,
                                                        unsigned long long *slot = nullptr) {
    HitKey key;
    // Most rays leave at most two lanes with a candidate (the line through a closed scene crosses
    // few primitives on its positive side): pick the nearer of them with scalar code instead of
    // a 64-lane reduction.
    const unsigned long long cands = __builtin_amdgcn_ballot_w64(bestIdx != kMiss);
    const int ncand = __builtin_popcountll(cands);
    // Only the sign test `det < epsilon` of the winner's determinant is ever used: inside this function
    // it travels as bit 31 of the index word (the answers exchanged BETWEEN waves carry it in bit 0:
    // packAnswer), which saves the two cross-lane reads of the determinant (a v_readlane with a computed
    // lane costs a lone wave four issue slots).
    const uint32_t packed = bestIdx | (bestDet < kEpsilon ? 0x80000000u : 0u);
    uint32_t pw;
    if (ncand == 0) {
      key.t = kInf, key.idx = kMiss, key.det = 0;
      return key;
    } else if (ncand == 1) {
      const int la = __builtin_ctzll(cands);
      key.t = readLane(bestT, la);
      pw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), la));
    } else if (ncand == 2) {
      const int la = __builtin_ctzll(cands);
      const int lb = __builtin_ctzll(cands & (cands - 1));
      const double ta = readLane(bestT, la), tb = readLane(bestT, lb);
      const uint32_t pa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), la));
      const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), lb));
      // strictly nearer wins; an exact tie goes to the lower combined index (Scene.cpp:31,95,118)
      const bool pickB = uniformBool((tb < ta) | ((tb == ta) & ((pb & 0x7fffffffu) < (pa & 0x7fffffffu))));
      key.t = pickB ? tb : ta;
      pw = pickB ? pb : pa;
    } else {
      // (walking three to six candidates with scalar code instead - three v_readlane and a few
      // scalar compares each - measured slower: Cornell 7.28 against 7.69 Msamples/s,
      // profiles/r02r_pick_loop_probe.txt)
      unsigned tHi, tLo;
      double tmin;
      if (WAVES > 1 && slot) {
        // Distances are positive doubles: their bit patterns order like unsigned 64-bit integers.  The
        // first candidate lane resets the slot, every candidate lane folds its distance in with one
        // ds_min_u64, everybody reads the result - three LDS instructions of one wave to one address,
        // served in order - instead of twelve DPP steps.
        typedef unsigned long long __attribute__((address_space(3))) LdsU64;
        LdsU64 *ls = (LdsU64 *)(slot);
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(bestT));
        if (static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(cands)) *(volatile LdsU64 *)ls = ~0ull;
        asm volatile("" ::: "memory");
        if (bestIdx != kMiss) (void)__hip_atomic_fetch_min(ls, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        asm volatile("" ::: "memory");
        const unsigned long long tb = *(volatile LdsU64 *)ls;
        tHi = static_cast<unsigned>(tb >> 32), tLo = static_cast<unsigned>(tb);
        tmin = __longlong_as_double(static_cast<long long>(tb));
      } else {
        tmin = waveMinPositive(bestT, tHi, tLo);
      }
      unsigned long long owner = __builtin_amdgcn_ballot_w64(
          static_cast<unsigned>(hi32(bestT)) == tHi && static_cast<unsigned>(lo32(bestT)) == tLo);
      if (__builtin_popcountll(owner) != 1) { // exact tie between lanes: lowest combined index wins
        const uint32_t imin = waveMinUFused(bestT == tmin ? bestIdx : kMiss);
        owner = __builtin_amdgcn_ballot_w64(bestIdx == imin);
      }
      key.t = tmin;
      pw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(packed), __builtin_ctzll(owner)));
    }
    key.idx = pw & 0x7fffffffu;
    key.det = (pw >> 31) ? -1.0 : 1.0;
    return key;
  }

  // This wave's part of Scene::intersect (Scene.cpp:115-122): its lanes' resident primitives
  // against the ray, then the wave-level nearest hit with the reference's tie-break.
  __device__ __forceinline__ HitKey localNearest(d3 o, d3 d) {
    PTW_T(tA);
    double bestT = kInf, bestDet = 0;
    uint32_t bestIdx = kMiss;
    const uint32_t nsph = p->nsph;
    // spheres first (lower combined index)
    if (hasSphere) testSphere(o, d, mk(scx, scy, scz), sr2, static_cast<uint32_t>(tid), bestT, bestIdx);
    if (!REG && nsph > static_cast<uint32_t>(kThreads)) // rare: more spheres than lanes
      for (uint32_t i = tid + kThreads; i < nsph; i += kThreads) {
        const SphereRec &r = spheresGlobal[i];
        testSphere(o, d, ld3(r.centre), r.radiusSquared, i, bestT, bestIdx);
      }
    if constexpr (PRE) {
      // the fp32 look at every resident slot, two per instruction; bit s of `keep`: slot s goes to the fp64 test
      const PrefilterRay ray = prefilterRay(o, d);
      uint32_t keep = 0;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        if (2 * q >= myUnits) continue; // (wave-uniform: this wave's share ends at myUnits)
        const Float2 r = prefilterPair(ray, pv0x[q], pv0y[q], pv0z[q], pe1x[q], pe1y[q], pe1z[q], pe2x[q], pe2y[q], pe2z[q],
                                       pea[q], peb[q]);
        keep |= (!(r.x < 0.0f) ? 1u : 0u) << (2 * q) | (!(r.y < 0.0f) ? 2u : 0u) << (2 * q);
      }
      // ... and the reference's test for the slots that are left, lowest slot first (a lane's slots are in index
      // order, `<` is strict: the tie-break of the plain loop), each lane on the triangle it still has to look at
      while (__builtin_amdgcn_ballot_w64(keep != 0) != 0) {
        if (keep != 0) {
          const int s = __builtin_ctz(keep);
          keep &= keep - 1;
          const uint32_t k = slotTriangle(s);
          const double *g = triGeom + 9 * static_cast<size_t>(k);
          testTriangle(o, d, ld3(g), ld3(g + 3), ld3(g + 6), nsph + k, bestT, bestIdx, bestDet);
        }
      }
    }
    // a worker wave holds 64 consecutive triangles per slot: for scenes whose units mostly fail the u test as a whole
    // (decided on the host per scene, TraceParams::seqUnitUFirst) the dispatcher picks the UNIT instantiation - the
    // unit-level early-out - and the fused test for the others.  (One kernel with both loops behind a wave-uniform
    // flag was measured first: the second copy of the unrolled loop cost either path 2-4 %, profiles/r06ab_*.)
    constexpr bool unitUFirst = UNIT;
#pragma unroll
    for (int s = 0; s < (PRE ? 0 : SLOTS); ++s) {
      // (wave-uniform: this wave's share of the scene ends at myUnits.  A guard, not a `break`: with a
      // second loop exit the compiler stops unrolling from nine slots on, indexes the slot arrays
      // at run time and moves them to scratch memory)
      if (WAVES > 1 && s >= myUnits) continue;
      if constexpr (UNIT)
        testTriangleUnit<true>(o, d, mk(v0x[s], v0y[s], v0z[s]), mk(e1x[s], e1y[s], e1z[s]),
                         mk(e2x[s], e2y[s], e2z[s]), nsph + slotTriangle(s), bestT, bestIdx, bestDet);
      else
        testTriangle(o, d, mk(v0x[s], v0y[s], v0z[s]), mk(e1x[s], e1y[s], e1z[s]),
                     mk(e2x[s], e2y[s], e2z[s]), nsph + slotTriangle(s),
                     bestT, bestIdx, bestDet);
    }
    // rare: more triangles than resident slots -> stream the remainder from memory
    if (!REG && p->ntri > residentTriangles())
      for (uint32_t k = residentTriangles() + tid; k < p->ntri; k += kThreads) {
        const double *g = triGeom + 9 * static_cast<size_t>(k);
        // (a wave's lanes hold 64 consecutive triangles here too)
        if (unitUFirst) testTriangleUnit<false>(o, d, ld3(g), ld3(g + 3), ld3(g + 6), nsph + k, bestT, bestIdx, bestDet);
        else testTriangle(o, d, ld3(g), ld3(g + 3), ld3(g + 6), nsph + k, bestT, bestIdx, bestDet);
      }

    // wave reduction: lexicographic min of (t, idx)
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(bestT));
#endif
    PTW_T(tB);
    PTW_ACC(0, tA, tB);
    HitKey key = pickNearest(bestT, bestIdx, bestDet, WAVES > 1 ? minSlot : nullptr);
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(key.t));
#endif
    PTW_T(tC);
    PTW_ACC(1, tB, tC);
    return key;
  }

  __device__ __forceinline__ void setLookAheadFrame(const Surface &s, d3 dirIn, double invU, double invV) {
    laSurf = s, laDir = dirIn, laInvU = invU, laInvV = invV;
  }
  __device__ __forceinline__ void armLookAhead(bool on, int nu, int nv) {
    laArmed = on && (laMisses < 2 || (++laTick & 7u) == 0);
    laU = nu, laV = nv;
  }
  __device__ __forceinline__ void lookAhead() {
    laPos = -1;
    if (pos + 3 <= kMtDoubles) { // (a sub-sample whose draws straddle a regeneration takes the plain path)
      const double xu = sh->canon[pos], xv = sh->canon[pos + 1], pd = sh->canon[pos + 2];
      double u, v;
      stratify(*p, laU, laV, xu, xv, laInvU, laInvV, u, v);
      laRefl = scatter(*this, laSurf, laDir, u, v, pd, laOut);
      laPos = pos;
    }
  }
  __device__ __forceinline__ bool takeLookAhead(d3 &dirOut, bool &refl) {
    const bool hit = laPos == pos; // wave-uniform (pos >= 0)
    if (laArmed) laMisses = hit ? 0 : (laMisses < 2 ? laMisses + 1 : 2);
    laPos = -1;
    laArmed = false;
    if (!hit) return false;
    dirOut = laOut;
    refl = laRefl;
    pos += 3;
    words += 6;
    return true;
  }

  __device__ __forceinline__ void flushPending() {
    if (pendKind == 0) return;
    const double *r = tab.tri + static_cast<size_t>(pendIdx - p->nsph) * kTriCompactDoubles;
    const double *m = tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
    push(pendLevel, ld3(m), ld3(m + 3), false, pendIdx);
    pendKind = 0;
  }
  __device__ __forceinline__ void setPending(int level, uint32_t idx) {
    pendKind = 1, pendLevel = level, pendIdx = idx;
  }

  // The nearest of the WAVES workers' answers with the reference's tie-break (strictly nearer wins,
  // an exact tie goes to the lower combined index: Scene.cpp:31,95,118).  Every lane of the master
  // reads all answers (broadcast LDS reads, one wait) and runs the same pick: no cross-lane traffic at
  // all, where the general pick spends 0.5-0.9 k cycles on ballots, readlanes and - from three
  // candidates on - a 64-lane reduction.
  __device__ __forceinline__ HitKey pickPartials() const {
    PartialHit ph[WAVES];
    // One ds_read_b128 per answer, all issued before the first is used (copied member by member the
    // compiler reads 8 + 4 bytes each - twelve LDS instructions for six answers): the
    // empty asm statement makes all four dwords of every answer count as used.
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    const U4 *src = reinterpret_cast<const U4 *>(partials);
    static_assert(WAVES == 1 || WAVES == 6 || WAVES == 7, "six or seven worker waves");
    U4 raw[WAVES];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) raw[w] = src[w];
    if constexpr (WAVES == 6)
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]));
    if constexpr (WAVES == 7)
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]));
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      ph[w].t = mk64(static_cast<int>(raw[w].x), static_cast<int>(raw[w].y));
      ph[w].idxSign = raw[w].z;
      ph[w].pad = 0;
    }
    return pickOfAnswers(ph);
  }

  // Pick checksum bookkeeping (see picksOn).
  __device__ __forceinline__ void pickReset() { pickS1 = 0, pickS2 = 0, pickN = 0; }
  __device__ __forceinline__ void pickNote(const HitKey &k) {
    const uint32_t v = k.idx == kMiss ? 0u : k.idx + 1u;
    pickN += 1u;
    pickS1 += v;
    pickS2 += pickN * v;
  }

  // Scene::intersect for the whole workgroup.  WAVES == 1: the wave's own result.  WAVES > 1:
  // wave 0 (the master, the only wave that runs the path logic) publishes the ray, every wave
  // searches its resident primitives, the partial results meet in LDS.  The worker waves sit in
  // workerLoop() and do nothing but this - while the master shades, their SIMDs are free for
  // the waves of other passes.
  __device__ __forceinline__ HitKey intersect(d3 o, d3 d) {
    rays++;
    if (WAVES == 1) {
      const HitKey key = localNearest(o, d);
      if constexpr (PICKS) if (picksOn) pickNote(key);
      return key;
    }
    PTW_T(tM0);
#if PTW_PROFILE_PHASES
    if (lastExit) {
      const unsigned long long gap = tM0 - lastExit;
      const int which = lastKind * 2 + lastMiss;
      if (which == 0) g00 += gap, n00++;
      if (which == 1) g01 += gap, n01++;
      if (which == 2) g10 += gap, n10++;
      if (which == 3) g11 += gap, n11++;
      if (which == 4) g20 += gap, n20++;
      if (which == 5) g21 += gap, n21++;
    }
#endif
    if ((threadIdx.x & 63) == 0) { // (the master wave's first lane; cmd / partials are this master's)
      cmd->o[0] = o.x, cmd->o[1] = o.y, cmd->o[2] = o.z;
      cmd->d[0] = d.x, cmd->d[1] = d.y, cmd->d[2] = d.z;
      if (MASTERS == 1) cmd->op = kCmdTrace;
    }
#if PTW_PROFILE_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMa);
    ldsBarrier(); // B1: ray visible to the workers
    PTW_T(tMb);
    // the search takes a thousand cycles and more: the stack entry of the level just left ...
    flushPending();
    if (laArmed) lookAhead(); // ... and the next sub-sample's first-bounce scatter
#if PTW_PROFILE_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PTW_T(tMc);
    ldsBarrier(); // B2: partial results visible
    PTW_T(tMd);
#if PTW_PROFILE_PHASES
    mprof[0] += tMa - tM0, mprof[1] += tMb - tMa, mprof[2] += tMc - tMb, mprof[3] += tMd - tMc;
#endif
    // (MASTERS == 2: the same two barriers - the workers search this ray between them, and the
    // other master's ray between B2 and this master's next B1, i.e. while this one shades)
    if (MASTERS == 2) tick += 2;
    const HitKey key = pickPartials();
    if constexpr (PICKS) if (picksOn) pickNote(key);
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(const_cast<HitKey &>(key).t));
#endif
    PTW_T(tM1);
    PTW_ACC(5, tM0, tM1);
#if PTW_PROFILE_PHASES
    mprof[4] += tM1 - tMd;
    lastKind = rayKind, lastMiss = key.idx == kMiss ? 1 : 0, lastExit = __builtin_amdgcn_s_memtime();
    rayKind = 2; // (whoever traces a primary ray or a sub-sample's first ray says so before the call)
#endif
    return key;
  }

  // Worker waves (WAVES > 1, wave != 0): serve nearest-hit requests until told to stop.
  __device__ __forceinline__ void workerLoop() {
#if PTW_PROFILE_PHASES
    for (int i = 0; i < 12; ++i) prof[i] = 0;
    unsigned long long nreq = 0;
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
    if (MASTERS == 2) {
      // Barrier n is followed by the search of master (n & 1)'s ray, which that master published
      // before it.  A command's `op` holds the barrier index from which its master has no more
      // rays (kCmdLive while it has): a value that reads the same whenever it is looked at, so all
      // waves leave after the same barrier.
      for (unsigned n = 0;; ++n) {
        ldsBarrier();
        const SeqCommand &c = allCmds[n & 1];
        const uint32_t mine = c.op, other = allCmds[(n & 1) ^ 1].op;
        if (mine <= n) {
          if (other <= n) break;
          continue;
        }
        const d3 o = mk(c.o[0], c.o[1], c.o[2]);
        const d3 d = mk(c.d[0], c.d[1], c.d[2]);
        const HitKey found = localNearest(o, d);
        if ((tid & 63) == 0) {
          PartialHit ph;
          ph.t = found.t, ph.pad = 0;
          ph.idxSign = packAnswer(found);
          partials[(n & 1) * WAVES + (tid >> 6)] = ph;
        }
        // (Round 5 also had the worker wave that answers LAST - an LDS counter per master - pick the
        // nearest of the six answers, so that the master reads one: the 0.4 k cycles moved from the
        // master's tick to the search's tail, and suzanne ran 10.0 against 11.7, ce 1.90 against 2.11
        // Msamples/s, profiles/r05h_*: the tick is the maximum of both, not the master's alone.)
#if PTW_PROFILE_PHASES
        nreq++;
#endif
      }
    } else
    for (;;) {
      ldsBarrier(); // B1
      if (cmd->op == kCmdExit) break;
      const d3 o = mk(cmd->o[0], cmd->o[1], cmd->o[2]);
      const d3 d = mk(cmd->d[0], cmd->d[1], cmd->d[2]);
      const HitKey mine = localNearest(o, d);
      if ((tid & 63) == 0) {
        PartialHit ph;
        ph.t = mine.t, ph.pad = 0;
        ph.idxSign = packAnswer(mine);
        partials[tid >> 6] = ph;
      }
#if PTW_PROFILE_PHASES
      nreq++;
#endif
      ldsBarrier(); // B2
    }
#if PTW_PROFILE_PHASES
    if (blockIdx.x == 0 && (tid & 63) == 0) { // every worker wave: which ones are the slow ones?
      const unsigned long long w1 = __builtin_amdgcn_s_memtime();
      printf("WORKER rank=%d (hardware wave %d, %d units) requests=%llu total/req=%.0f tests=%.0f reduce=%.0f\n", tid >> 6,
             (int)(threadIdx.x >> 6), myUnits, nreq, (double)(w1 - w0) / nreq, (double)prof[0] / nreq, (double)prof[1] / nreq);
    }
#endif
  }
  __device__ __forceinline__ void stopWorkers() {
    if (WAVES == 1) return;
    if (MASTERS == 2) {
      // no more rays from this master as of its next barrier; keep the cadence until the other
      // one is done too (see workerLoop)
      if ((threadIdx.x & 63) == 0) cmd->op = tick;
      for (unsigned n = tick;; ++n) {
        ldsBarrier();
        if (allCmds[0].op <= n && allCmds[1].op <= n) break;
      }
      return;
    }
    if (threadIdx.x == 0) cmd->op = kCmdExit;
    ldsBarrier(); // pairs with the workers' B1
  }

  __device__ __forceinline__ bool branch(bool b) const { return uniformBool(b); }

  __device__ __forceinline__ d3 recD3(int at, int lane) const {
    return mk(readLane(rec[at], lane), readLane(rec[at + 1], lane), readLane(rec[at + 2], lane));
  }
  __device__ __forceinline__ d3 emissionAt(const HitKey &k) const {
    if (REG && k.idx >= p->nsph) {
      const int lane = static_cast<int>(k.idx - p->nsph);
      if (!((emissiveMask >> lane) & 1ull)) return mk(0, 0, 0);
      return recD3(kRecEmission, lane);
    }
    if (k.idx >= p->nsph) {
      const double *r = tab.tri + static_cast<size_t>(k.idx - p->nsph) * kTriCompactDoubles;
      return ld3(tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles);
    }
    return ld3(tab.sph[k.idx].emission);
  }
  // consume three draws without looking at them
  __device__ __forceinline__ void skip3() {
    if (SPEC) {
      advance(3);
      return;
    }
    if (pos + 3 <= kMtDoubles) {
      pos += 3;
      words += 6;
    } else {
      (void)draw();
      (void)draw();
      (void)draw();
    }
  }

  // The same scatter with the three draws at block position q (q + 3 <= kMtDoubles), without touching
  // the stream position (scatterChain() calls it with the frontier's).
  __device__ __forceinline__ bool scatterChainAt(int q, const Surface &s, d3 dirIn, d3 &dirOut) const {
    const double pd = sh->canon[q + 2];
    const d3 local = mk(sh->hemi[q][0], sh->hemi[q][1], sh->hemi[q][2]);
    if (uniformBool(lobeIsReflective(s, dirIn, pd))) { // Scene.cpp:163-168
      dirOut = coneSample(reflect(s.normal, dirIn), s.coneAngle, sh->canon[q], sh->canon[q + 1]);
      return true;
    }
    dirOut = normalisedNearUnit(transform(s.basis, local)); // Scene.cpp:169-175
    return false;
  }

  // The scatter of a single-sample level (depth >= 1): u = xi1, v = xi2, p = xi3 drawn in that
  // order (Scene.cpp:157-161).  When the three draws sit inside the current block, the diffuse
  // lobe takes its local direction from the precomputed table.
  __device__ __forceinline__ bool scatterChain(const Surface &s, d3 dirIn, d3 &dirOut) {
    if (SPEC) {
      const double *cn = ringCanon() + pos;
      const double *hm = ringHemi(pos);
      const double u = cn[0], v = cn[1], pd = cn[2];
      const d3 local = mk(hm[0], hm[1], hm[2]);
      advance(3);
      if (uniformBool(lobeIsReflective(s, dirIn, pd))) { // Scene.cpp:163-168
        dirOut = coneSample(reflect(s.normal, dirIn), s.coneAngle, u, v);
        return true;
      }
      dirOut = normalisedNearUnit(transform(s.basis, local)); // Scene.cpp:169-175
      return false;
    }
    if (pos + 3 <= kMtDoubles) {
      const int q = pos;
      pos += 3;
      words += 6;
      return scatterChainAt(q, s, dirIn, dirOut);
    }
    double u, v, pd;
    draw3(u, v, pd); // straddles a regeneration
    Surface r = s;
    r.reflectivity = resolveReflectivity(s, dirIn);
    return scatter(*this, r, dirIn, u, v, pd, dirOut);
  }

  // `p < reflectivity` (Scene.cpp:143-146,163) without always evaluating Norm3::reflectance.
  // For ior == 1 on both sides the reflectance is ((c - c') / (c + c'))^2 with c' = sqrt(1 - (1 -
  // c^2)) differing from c = cos(theta_i) only by rounding: |c'^2 - c^2| <= 3e-16, so for
  // c >= 1e-3 the value is below 2^-64, the spacing of the canonical draws - `p < reflectivity`
  // can then only hold for p == 0.  Everything else takes the exact evaluation.
  __device__ __forceinline__ bool lobeIsReflective(const Surface &s, d3 dirIn, double pd) const {
    if (uniformBool(s.matReflectivity >= 0)) return pd < s.matReflectivity;
    if (uniformBool(s.iorFrom == 1.0 && s.iorTo == 1.0)) {
      const double cosThetaI = -dot(s.normal, dirIn);
      if (uniformBool(cosThetaI >= 1e-3 && pd > 0.0)) return false;
    }
    return pd < reflectance(s.normal, dirIn, s.iorFrom, s.iorTo, s.iorRatio);
  }
#if PTW_PROFILE_PHASES
  __device__ __forceinline__ void markRay(int kind) { rayKind = kind; }
  __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void acc(int slot, unsigned long long t0, double &keep) {
    asm volatile("" : "+v"(keep));
    prof[slot] += __builtin_amdgcn_s_memtime() - t0;
  }
#else
  __device__ __forceinline__ void markRay(int) {}
  __device__ __forceinline__ unsigned long long now() const { return 0; }
  __device__ __forceinline__ void acc(int, unsigned long long, double &) {}
#endif
  __device__ __forceinline__ void push(int level, d3 e, d3 dif, bool refl, uint32_t idx) {
    if (REG) {
      const unsigned sh8 = static_cast<unsigned>(level) * 8u;
      const unsigned long long w =
          static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(idx)) & 0x7f) |
          (refl ? 0x80ull : 0ull);
      stackBits = (stackBits & ~(0xffull << sh8)) | (w << sh8);
      return;
    }
    // One lane stores (64 lanes writing one address would serialise in the LDS); every lane
    // reads it back later.  The address is the same for the store and the loads, so the
    // compiler keeps them ordered.
    if ((tid & 63) == 0) {
      Level lv;
      lv.emission = e;
      lv.diffuse = dif;
      lv.reflective = refl;
      stack[level] = lv;
    }
  }
  __device__ __forceinline__ Level top(int level) const {
    if (REG) {
      const unsigned w = static_cast<unsigned>(stackBits >> (static_cast<unsigned>(level) * 8u)) & 0xffu;
      const uint32_t idx = w & 0x7fu;
      Level lv;
      lv.reflective = (w >> 7) != 0;
      if (idx >= p->nsph) {
        const int lane = static_cast<int>(idx - p->nsph);
        lv.emission = recD3(kRecEmission, lane);
        lv.diffuse = recD3(kRecDiffuse, lane);
      } else {
        lv.emission = ld3(tab.sph[idx].emission);
        lv.diffuse = ld3(tab.sph[idx].diffuse);
      }
      return lv;
    }
    return stack[level];
  }

  __device__ __forceinline__ d3 runChain(const TraceParams &tp, const TriShade *ts, const SphereRec *sp,
                                         d3 o, d3 d) {
    if (REG) return chainHot(tp, o, d);
    // (two masters: suzanne 512 passes +1 %, ce +6 %; with one master per workgroup it measured 5 %
    // slower than radianceChain - profiles/r03c_worker_wave_master_path.txt - and is not used there)
    if (WAVES > 1 && MASTERS == 2) return chainMaster(tp, o, d);
    return radianceChain(*this, tp, ts, sp, o, d);
  }

  // radianceChain() for the master wave of the worker-wave kernels (WAVES > 1), arranged like
  // chainHot() around what the master's serial path pays for.  With two masters per workgroup a
  // tick of the protocol lasts as long as the slower of "the workers search one master's ray" and
  // "the other master reads the answers, picks, shades and publishes" - on suzanne the latter
  // (1.2 k against 2.1 k cycles per ray).  The common level - a triangle hit, the
  // diffuse lobe, the draws inside the generator block - is decided by two branches with all its
  // LDS operands (triangle record with the lobe threshold, the draw, the draw-derived local
  // direction) waited for once; the level's colours are not needed before the fold, so their fetch
  // (triangle record -> material index -> material: dependent round trips) and the stack entry are
  // left to flushPending(), which runs while the workers search the NEXT ray.  Everything else
  // takes the general code, which is the sequence radianceChain() runs.
  __device__ __forceinline__ d3 chainMaster(const TraceParams &tp, d3 o, d3 d) {
    if (tp.maxDepth <= 1) return mk(0, 0, 0); // Scene.cpp:128 at depth 1
    const HitKey k = intersect(o, d);
    if (uniformBool(k.idx == kMiss)) return envColour; // Scene.cpp:131-133
    return chainMasterFrom(tp, o, d, k);
  }
  // ... from the first hit `k` of the ray (o, d) on (not a miss: the fan-out loop of radiance0 deals
  // with the sub-sample whose first ray leaves the scene - four of five on suzanne - itself, in a
  // handful of instructions).
  __device__ __forceinline__ d3 chainMasterFrom(const TraceParams &tp, d3 o, d3 d, HitKey k) {
    int nlev = 0;
    d3 L;
    const int maxDepth = tp.maxDepth;
    const uint32_t nsph = tp.nsph, ntri = tp.ntri;
    int depth = 1;
    for (;; k = intersect(o, d)) {
      const int notLast = depth + 1 - maxDepth;   // < 0
      const int inBlock = pos + 2 - kMtDoubles;   // < 0  <=>  pos + 3 <= kMtDoubles
      const bool isTri = (k.idx - nsph) < ntri;   // unsigned: kMiss and spheres fail
      if (isTri & ((notLast & inBlock) < 0)) {
        const unsigned long long tH0 = now();
        const double *r = tab.tri + static_cast<size_t>(k.idx - nsph) * kTriCompactDoubles;
        const int q = pos;
        d3 n = ld3(r), bx = ld3(r + 3), by = ld3(r + 6);
        double thr = r[kTriLobeThreshold];
        double pd = sh->canon[q + 2];
        const double *hm = sh->hemi[q];
        d3 local = mk(hm[0], hm[1], hm[2]);
        asm volatile("" : "+v"(n.x), "+v"(bx.x), "+v"(by.x), "+v"(thr), "+v"(pd), "+v"(local.x)); // one wait
        const bool backfacing = k.det < kEpsilon; // Scene.cpp:107
        const double ndotd = dot(n, d);
        const double cosThetaI = backfacing ? ndotd : -ndotd; // -dot(+-n, d)
        const unsigned long long mNotRefl = __builtin_amdgcn_ballot_w64(!(pd < thr));
        const unsigned long long mPlain = __builtin_amdgcn_ballot_w64(thr >= 0.0);
        const unsigned long long mCos = __builtin_amdgcn_ballot_w64(cosThetaI >= 1e-3);
        const unsigned long long mPos = __builtin_amdgcn_ballot_w64(pd > 0.0);
        if ((mNotRefl & (mPlain | (mCos & mPos))) != 0) { // the diffuse lobe (see chainHot / ptw_layout.h)
          pos += 3;
          words += 6;
          Basis b;
          b.x = bx, b.y = by, b.z = n;
          const double sgn = backfacing ? -1.0 : 1.0;
          const d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
          o = o + d * k.t;
          d = nd;
          setPending(nlev++, k.idx);
          ++depth;
          acc(11, tH0, d.x);
          continue;
        }
      }
      // ---- general path: miss, last level, sphere, reflective lobe, straddling draws ----
      if (uniformBool(k.idx == kMiss)) { // Scene.cpp:131-133
        L = envColour;
        break;
      }
      if (depth + 1 >= maxDepth) { // last level: see radianceChain()
        skip3();
        L = emissionAt(k);
        break;
      }
      const Surface s = surfaceAt(k, o, d, false);
      d3 nd;
      const bool refl = scatterChain(s, d, nd);
      push(nlev++, s.emission, s.diffuse, refl, k.idx); // (rare: written at once)
      o = s.pos;
      d = nd;
      ++depth;
    }
    flushPending();
    for (int i = nlev - 1; i >= 0; --i) L = fold(i, L);
    return L;
  }

  // radianceChain() for the REG variant, arranged around what a single wave per SIMD pays for:
  // every instruction is one issue slot, and a branch - even an untaken one - costs five to ten
  // of them (scripts/microbench/issue_costs.hip).  The common case, a chain level that hits a
  // triangle and takes the diffuse lobe with its three draws inside the current generator block,
  // is decided by TWO branches: one on scalar facts about the hit, one on the lobe predicate
  // (evaluated as lane-mask logic from the per-triangle lobe threshold, see ptw_layout.h).  All
  // its LDS operands (triangle record, draws, draw-derived local direction) are requested
  // together and waited for once.  Everything else - miss, last level, sphere, reflective lobe,
  // Fresnel evaluation, draws straddling a regeneration - takes the general code below, which
  // is the same sequence radianceChain() runs.
  //   backfacing hits: surfaceAt() negates the normal and basis.x; negation commutes exactly
  //   with the products of transform(), so the signs go onto the local direction instead.
  __device__ __forceinline__ d3 chainHot(const TraceParams &tp, d3 o, d3 d) {
    int nlev = 0;
    d3 L;
    const int maxDepth = tp.maxDepth;
    const uint32_t nsph = tp.nsph, ntri = tp.ntri;
    if (maxDepth <= 1) return mk(0, 0, 0); // Scene.cpp:128 at depth 1
    int depth = 1;
    for (;;) {
      HitKey k;
      // ---- hot loop: stays inside while every level is a diffuse triangle bounce ----
      for (;;) {
        k = intersect(o, d);
        // hit a triangle (not a miss, not a sphere), not the last level, draws inside the block:
        // three differences that are all negative exactly then
        const int notLast = depth + 1 - maxDepth;       // < 0
        const int inBlock = SPEC ? -1 : pos + 2 - kMtDoubles; // < 0  <=>  pos + 3 <= kMtDoubles
        const bool isTri = (k.idx - nsph) < ntri;       // unsigned: kMiss and spheres fail
        if (!(isTri & ((notLast & inBlock) < 0))) break;
        const double *r = tab.tri + static_cast<size_t>(k.idx - nsph) * kTriCompactDoubles;
        const int q = pos;
        d3 n = ld3(r), bx = ld3(r + 3), by = ld3(r + 6);
        double thr = r[kTriLobeThreshold];
        double pd = SPEC ? ringCanon()[q + 2] : sh->canon[q + 2];
        const double *hm = SPEC ? ringHemi(q) : sh->hemi[q];
        d3 local = mk(hm[0], hm[1], hm[2]);
        // one wait for everything: without this the loads the lobe test does not need sink below
        // its branch and are waited for a second time
        asm volatile("" : "+v"(n.x), "+v"(bx.x), "+v"(by.x), "+v"(thr), "+v"(pd), "+v"(local.x));
        const bool backfacing = k.det < kEpsilon; // Scene.cpp:107
        const double ndotd = dot(n, d);
        const double cosThetaI = backfacing ? ndotd : -ndotd; // -dot(+-n, d)
        // lobe predicate as lane-mask logic (all lanes agree)
        const unsigned long long mNotRefl = __builtin_amdgcn_ballot_w64(!(pd < thr));
        const unsigned long long mPlain = __builtin_amdgcn_ballot_w64(thr >= 0.0);
        const unsigned long long mCos = __builtin_amdgcn_ballot_w64(cosThetaI >= 1e-3);
        const unsigned long long mPos = __builtin_amdgcn_ballot_w64(pd > 0.0);
        if ((mNotRefl & (mPlain | (mCos & mPos))) == 0) break;
        if (SPEC) {
          advance(3);
        } else {
          pos += 3;
          words += 6;
        }
        Basis b;
        b.x = bx, b.y = by, b.z = n;
        const double sgn = backfacing ? -1.0 : 1.0;
        const d3 nd = normalisedNearUnit(transform(b, mk(local.x * sgn, local.y, local.z * sgn)));
        o = o + d * k.t;
        d = nd;
        push(nlev++, mk(0, 0, 0), mk(0, 0, 0), false, k.idx);
        ++depth;
      }
      // ---- general path: miss, last level, sphere, reflective lobe, straddling draws ----
      if (uniformBool(k.idx == kMiss)) { // Scene.cpp:131-133
        L = envColour;
        break;
      }
      if (depth + 1 >= maxDepth) { // last level: see radianceChain()
        skip3();
        L = emissionAt(k);
        break;
      }
      const Surface s = surfaceAt(k, o, d, false);
      d3 nd;
      const bool refl = scatterChain(s, d, nd);
      push(nlev++, s.emission, s.diffuse, refl, k.idx);
      o = s.pos;
      d = nd;
      ++depth;
    }
    for (int i = nlev - 1; i >= 0; --i) L = fold(i, L);
    return L;
  }

  // One step of the innermost-first fold: L_level = E + T * L_child (Scene.cpp:163-175).
  __device__ __forceinline__ d3 fold(int level, d3 L) const {
    if (REG) {
      const unsigned w = static_cast<unsigned>(stackBits >> (static_cast<unsigned>(level) * 8u)) & 0xffu;
      const uint32_t idx = w & 0x7fu;
      const bool refl = (w >> 7) != 0;
      if (idx >= p->nsph) {
        const int lane = static_cast<int>(idx - p->nsph);
        if (!((emissiveMask >> lane) & 1ull)) {
          // E == +0 and the child radiance is never negative: E + x == x exactly
          return refl ? L : recD3(kRecDiffuse, lane) * L;
        }
        const d3 e = recD3(kRecEmission, lane);
        return refl ? e + L : e + recD3(kRecDiffuse, lane) * L;
      }
    }
    const Level lv = top(level);
    return uniformBool(lv.reflective) ? lv.emission + L : lv.emission + lv.diffuse * L;
  }

  // Surface at a hit from the shading tables (same values as makeSurface()).
  __device__ __forceinline__ Surface surfaceAt(const HitKey &k, d3 o, d3 d, bool eager = true) {
    PTW_T(tA);
    Surface s;
    s.pos = o + d * k.t;
    double ior, invIor, reflectivity;
    bool inside;
    if (k.idx >= p->nsph) {
      const double *r = tab.tri + static_cast<size_t>(k.idx - p->nsph) * kTriCompactDoubles;
      const bool backfacing = k.det < kEpsilon;
#if PTW_PROFILE_PHASES
      PTW_T(tL0);
#endif
      d3 n = ld3(r), bx = ld3(r + 3);
#if PTW_PROFILE_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n.x), "+v"(bx.x)::"memory");
      PTW_T(tL1);
      PTW_ACC(4, tL0, tL1);
#endif
      s.normal = backfacing ? -n : n;
      s.basis.x = backfacing ? -bx : bx;
      s.basis.y = ld3(r + 6);
      s.basis.z = s.normal;
      const double *m = tab.mat + static_cast<size_t>(static_cast<uint32_t>(r[kTriMaterialIndex])) * kMatDoubles;
      s.emission = ld3(m);
      s.diffuse = ld3(m + 3);
      ior = m[6], invIor = m[7], reflectivity = m[8];
      s.coneAngle = m[9];
      inside = backfacing;
    } else {
      const SphereRec &r = tab.sph[k.idx];
      d3 n = normalised(s.pos - ld3(r.centre));
      inside = dot(n, d) > 0;
      if (inside) n = -n;
      s.normal = n;
      s.basis = basisFromZ(n);
      s.emission = ld3(r.emission);
      s.diffuse = ld3(r.diffuse);
      s.coneAngle = r.coneAngle;
      ior = r.ior, invIor = r.invIor, reflectivity = r.reflectivity;
    }
    s.iorFrom = inside ? ior : 1.0;
    s.iorTo = inside ? 1.0 : ior;
    s.iorRatio = inside ? ior : invIor;
    s.matReflectivity = reflectivity;
    s.reflectivity = eager ? resolveReflectivity(s, d) : 0.0;
#if PTW_PROFILE_PHASES
    asm volatile("" : "+v"(s.reflectivity), "+v"(s.pos.x), "+v"(s.basis.y.z), "+v"(s.diffuse.x));
#endif
    PTW_T(tB);
    PTW_ACC(2, tA, tB);
    return s;
  }
};

// Bytes of dynamic LDS traceSequential needs (also computed on the host for the launch).
__host__ __device__ inline size_t seqLdsBytes(int waves, int maxDepth, bool ldsTables, uint32_t ntri,
                                              uint32_t nmat, uint32_t nsph, int masters = 1) {
  size_t n = masters * sizeof(SeqShared);
  n += static_cast<size_t>(waves) * (maxDepth > 0 ? maxDepth : 1) * sizeof(Level);
  n = (n + 15) & ~static_cast<size_t>(15); // the answers: 16-byte aligned (ds_read_b128)
  // (room for two answer sets per master - round 5's paired requests, LAB.md - of which one is used)
  n += 2 * masters * static_cast<size_t>(waves) * sizeof(PartialHit) + kSeqCmdBytes;
  n = (n + 63) & ~static_cast<size_t>(63);
  if (waves > 1) n += kSeqCamBytes; // the camera (worker-wave kernels: see traceSequential)
  if (ldsTables) {
    n += static_cast<size_t>(nsph) * sizeof(SphereRec);
    n += static_cast<size_t>(ntri) * kTriCompactDoubles * sizeof(double);
    n += static_cast<size_t>(nmat) * kMatDoubles * sizeof(double);
  }
  return n;
}

} // namespace
} // namespace ptw

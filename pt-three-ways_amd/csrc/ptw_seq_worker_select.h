// ptw_seq_worker_select.h - which <SLOTS> instantiation of the worker-wave kernels a scene gets; shared by the
// plain (seq_worker.hip, seq_worker2.hip) and the prefilter (seq_worker_pre.hip, seq_worker2_pre.hip) files, each of
// which instantiates its own set.  Internal to csrc/.
#pragma once
#include "ptw_seq_kernel.h"

namespace ptw {
namespace {

// Seven worker waves + one master wave = 8 waves = 2 per SIMD of one CU (256 registers per lane each): the smallest
// SLOTS (triangles per worker lane) that keeps every triangle resident.
template <bool PRE, bool UNIT = false>
hipError_t selectSeqOneMaster(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  int uO, uY, uM;
  seqUnitsFor(p.ntri, 6, 1, 12, hints, uO, uY, uM, PRE);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 7, 1, PRE, UNIT>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 7, 1, PRE, UNIT>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 7, 1, PRE, UNIT>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 7, 1, PRE, UNIT>(p, b, hints, stream);
  if (need <= 6) return launchSeq<6, 7, false, false, 1, PRE, UNIT>(p, b, hints, stream);
  if (need <= 8) return launchSeq<8, 7, false, false, 1, PRE, UNIT>(p, b, hints, stream);
  if constexpr (!PRE) { // (the shares by place of a large scene: ce 9 / 6 / 9; fp64 triangles only)
    if (need <= 9) return launchSeq<9, 7, false, false, 1, PRE, UNIT>(p, b, hints, stream);
    if (need <= 10) return launchSeq<10, 7, false, false, 1, PRE, UNIT>(p, b, hints, stream);
  }
  return launchSeq<12, 7, false, false, 1, PRE, UNIT>(p, b, hints, stream); // beyond 5376 the tail is streamed from memory
}

// The two-master kernels by the largest share of 64-triangle units any worker wave gets; shares are
// capped at what the register file holds without spilling inside the search loop (11 units = 198
// registers), the rest of a larger scene is streamed from memory.
template <bool PRE, bool UNIT = false>
hipError_t selectSeqTwoMasters(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  int uO, uY, uM;
  seqUnitsFor(p.ntri, 4, 2, 11, hints, uO, uY, uM);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 6, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 6, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 6, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 6, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 6) return launchSeqAuto<6, 6, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 9) return launchSeq<9, 6, false, false, 2, PRE, UNIT>(p, b, hints, stream);
  if (need <= 10) return launchSeq<10, 6, false, false, 2, PRE, UNIT>(p, b, hints, stream);
  return launchSeq<11, 6, false, false, 2, PRE, UNIT>(p, b, hints, stream);
}

} // namespace
} // namespace ptw

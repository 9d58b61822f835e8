// seq_worker.hip - traceSequential with seven worker waves and ONE master per pass (scenes beyond 128
// triangles, at most as many passes as CUs): the workers hold the triangles in registers, the master runs
// the path logic and exchanges ray / nearest hit with them through LDS (SeqCtx::intersect).
#include "ptw_seq_kernel.h"

namespace ptw {

hipError_t launchSeqOneMaster(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  // Smallest configuration that keeps every triangle resident in VGPRs: 7 worker waves + 1 master wave
  // = 8 waves = 2 per SIMD of one CU (256 registers per lane each), SLOTS triangles per worker lane.
  int uO, uY, uM;
  seqUnitsFor(p.ntri, 6, 1, 12, hints, uO, uY, uM);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 7>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 7>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 7>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 7>(p, b, hints, stream);
  if (need <= 6) return launchSeq<6, 7, false>(p, b, hints, stream);
  if (need <= 8) return launchSeq<8, 7, false>(p, b, hints, stream);
  return launchSeq<12, 7, false>(p, b, hints, stream); // beyond 5376 the tail is streamed from memory
}

} // namespace ptw

// seq_worker2.hip - traceSequential with six worker waves shared by TWO masters: two passes per workgroup,
// one barrier apart - the workers search one master's ray while the other master shades (more passes than
// CUs: BASELINE cfg3 / cfg4).
#include "ptw_seq_kernel.h"

namespace ptw {

// The two-master kernels by the largest share of 64-triangle units any worker wave gets; shares are
// capped at what the register file holds without spilling inside the search loop (11 units = 198
// registers), the rest of a larger scene is streamed from memory.
hipError_t launchSeqTwoMasters(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  int uO, uY, uM;
  seqUnitsFor(p.ntri, 4, 2, 11, hints, uO, uY, uM);
  const int need = std::max(uO, std::max(uY, uM));
  if (need <= 1) return launchSeqAuto<1, 6, 2>(p, b, hints, stream);
  if (need <= 2) return launchSeqAuto<2, 6, 2>(p, b, hints, stream);
  if (need <= 3) return launchSeqAuto<3, 6, 2>(p, b, hints, stream);
  if (need <= 4) return launchSeqAuto<4, 6, 2>(p, b, hints, stream);
  if (need <= 6) return launchSeqAuto<6, 6, 2>(p, b, hints, stream);
  if (need <= 9) return launchSeq<9, 6, false, false, 2>(p, b, hints, stream);
  if (need <= 10) return launchSeq<10, 6, false, false, 2>(p, b, hints, stream);
  return launchSeq<11, 6, false, false, 2>(p, b, hints, stream);
}

} // namespace ptw

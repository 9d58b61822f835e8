// seq_single.hip - traceSequential with ONE wave per pass (scenes of at most 128 triangles): lane k owns
// triangle k (and k + 64), the wave searches, picks, shades and draws by itself.  The REG variant keeps the
// shading records in registers and the (E, T) stack in a scalar register pair (SeqCtx, ptw_seq_ctx.h).
#include "ptw_seq_kernel.h"

namespace ptw {

hipError_t launchSeqSingle(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream,
                           int slots, bool reg) {
  if (slots == 1 && reg) return launchSeq<1, 1, true, true>(p, b, hints, stream);
  if (slots == 1) return launchSeqAuto<1, 1>(p, b, hints, stream);
  return launchSeqAuto<2, 1>(p, b, hints, stream);
}

} // namespace ptw

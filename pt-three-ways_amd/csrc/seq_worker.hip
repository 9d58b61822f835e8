// seq_worker.hip - traceSequential with seven worker waves and ONE master per pass (scenes beyond 128
// triangles, at most as many passes as CUs): the workers hold the triangles in registers, the master runs
// the path logic and exchanges ray / nearest hit with them through LDS (SeqCtx::intersect).
#include "ptw_seq_worker_select.h"

namespace ptw {

hipError_t launchSeqOneMaster(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  return selectSeqOneMaster<false>(p, b, hints, stream);
}

} // namespace ptw

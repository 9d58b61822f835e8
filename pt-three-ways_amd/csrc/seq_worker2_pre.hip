// seq_worker2_pre.hip - the two-master worker-wave kernels with the fp32 PREFILTER in the worker lanes
// (PTW_ACCEL_PREFILTER under the SEQUENTIAL policy: a separate, separately reported mode; SeqCtx PRE).
#include "ptw_seq_worker_select.h"

namespace ptw {

hipError_t launchSeqTwoMastersPrefilter(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  return selectSeqTwoMasters<true>(p, b, hints, stream);
}

} // namespace ptw

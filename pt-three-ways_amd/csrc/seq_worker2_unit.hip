// seq_worker2_unit.hip - the two-master worker-wave kernels (seq_worker2.hip) with the unit-level u-first early-out in
// the worker waves (see seq_worker_unit.hip): BASELINE cfg4's scene gets these.
#include "ptw_seq_worker_select.h"

namespace ptw {

hipError_t launchSeqTwoMastersUnit(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  return selectSeqTwoMasters<false, true>(p, b, hints, stream);
}

} // namespace ptw

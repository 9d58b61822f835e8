// seq_worker_unit.hip - the one-master worker-wave kernels (seq_worker.hip) with the UNIT-LEVEL U-FIRST EARLY-OUT in the
// worker waves (testTriangleUnit, ptw_trace_common.h): picked for scenes whose units of 64 consecutive triangles
// mostly fail the u test as a whole (TraceParams::seqUnitUFirst, host/precompute.h).  Same samples, bit for bit.
#include "ptw_seq_worker_select.h"

namespace ptw {

hipError_t launchSeqOneMasterUnit(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  return selectSeqOneMaster<false, true>(p, b, hints, stream);
}

} // namespace ptw

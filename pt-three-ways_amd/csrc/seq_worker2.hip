// seq_worker2.hip - traceSequential with six worker waves shared by TWO masters: two passes per workgroup,
// one barrier apart - the workers search one master's ray while the other master shades (more passes than
// CUs: BASELINE cfg3 / cfg4).
#include "ptw_seq_worker_select.h"

namespace ptw {

hipError_t launchSeqTwoMasters(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream) {
  return selectSeqTwoMasters<false>(p, b, hints, stream);
}

} // namespace ptw

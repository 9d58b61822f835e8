// ptw_launch.h - what the kernel translation units of csrc/ share on the HOST side: one launcher per kernel
// family (each family is its own .hip file, compiled in parallel), the name of the variant the last launch
// picked, and the dispatch constants.  ptw_kernels.h is the interface towards capi_render.hip; this header is
// internal to the kernel files.
//
//   dispatch.hip      launchTraceSequential / dispatchSequential: which family, which instantiation
//   seq_single.hip    traceSequential<SLOTS, 1, ...>            one wave per pass (<= 128 triangles)
//   seq_spec.hip      traceSequentialSpec                       four speculating waves per pass (<= 64 triangles)
//   seq_worker.hip    traceSequential<SLOTS, 7, ...>            seven worker waves + one master
//   seq_worker2.hip   traceSequential<SLOTS, 6, ..., 2 masters> six worker waves shared by two passes
//   seq_worker_unit.hip, seq_worker2_unit.hip   the same two families with the unit-level u-first early-out
//   seq_worker_pre.hip, seq_worker2_pre.hip   the same two families with the fp32 prefilter in the worker lanes
//   perpixel.hip      tracePerPixel, tracePerPixelPersistent    PERPIXEL policy, brute force
//   accel.hip         tracePerPixelBvh, tracePerPixelPrefilter  the separate accelerated modes
//   resolve_kat.hip   resolveKernel, intersectBatchKernel, rngKatKernel
#pragma once
#include "ptw_kernels.h"

#include <cstddef>

namespace ptw {

// Name of the variant the last launch*() call of this thread picked (reported through ptw_kernel_stats so
// that callers do not have to re-derive the dispatch rules).  printf-style; the text lives in a
// thread-local buffer until the next call.
void setVariant(const char *format, ...) __attribute__((format(printf, 1, 2)));
const char *lastVariant();

constexpr size_t kLdsTableBudget = 150 * 1024; // bytes of LDS we are willing to spend on shading tables
int deviceCus();
inline int cusFor(const LaunchHints &hints) { return hints.cus > 0 ? hints.cus : deviceCus(); }

// ---- SEQUENTIAL families (each defined in the file named above) --------------------------------------------
// the scenes the register-resident kernels handle (REG variant of the single-wave kernel, traceSequentialSpec)
bool specApplies(const TraceParams &p);
hipError_t launchSeqSpec(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
// one wave per pass: slots = 1 (<= 64 triangles; reg: the register-resident REG variant) or 2 (<= 128)
hipError_t launchSeqSingle(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream,
                           int slots, bool reg);
hipError_t launchSeqOneMaster(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
hipError_t launchSeqTwoMasters(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
// ... the same two families with the unit-level u-first early-out in the worker waves (seq_worker_unit.hip,
// seq_worker2_unit.hip: scenes with TraceParams::seqUnitUFirst set)
hipError_t launchSeqOneMasterUnit(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
hipError_t launchSeqTwoMastersUnit(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
// ... and the same two families with the fp32 prefilter in the worker lanes (seq_worker_pre.hip, seq_worker2_pre.hip:
// PTW_ACCEL_PREFILTER under the SEQUENTIAL policy)
hipError_t launchSeqOneMasterPrefilter(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);
hipError_t launchSeqTwoMastersPrefilter(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);

// ---- PERPIXEL: the accelerated modes (accel.hip; launchTracePerPixel in perpixel.hip hands over) -----------
hipError_t launchTraceAccel(const TraceParams &p, const TraceBuffers &b, const LaunchHints &hints, hipStream_t stream);

} // namespace ptw

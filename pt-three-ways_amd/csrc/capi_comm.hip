// capi_comm.hip — the multi-GPU half of the C ABI (include/ptw.h, "framebuffer collectives"):
// RCCL over xGMI behind plain C entry points.
//
// The reference merges its workers' frames with ArrayOutput::operator+= (src/dod/Scene.cpp:242,
// src/util/ArrayOutput.cpp:48-56) and, across processes, offline with raw_to_png
// (src/main/raw_to_png.cpp:39-58).  Here the same merge is ONE collective on device memory:
//   ptw_comm_reduce_framebuffer  pass-sharded renders: ncclReduce(sum) of fp64 sums + u32 counts
//   ptw_comm_gather_rows         row-interleaved renders: every rank sends the 1/world of the
//                                frame it owns to the root (grouped ncclSend/ncclRecv -
//                                point-to-point over the xGMI links into the root GPU)
//
// librccl is bound at first use (dlopen of the soname): a process that already carries an RCCL
// (PyTorch-ROCm bundles one as librccl.so.1) keeps a single copy, and the library still loads on
// hosts without a GPU.
//
// Two transports sit under the same entry points (and under the same packing / slot arithmetic /
// unpacking code of the gather):
//   RCCL      ptw_comm_create / ptw_comm_create_all: one communicator per GPU, xGMI;
//   loopback  ptw_comm_create_loopback: `world` communicators that share ONE device and meet in
//             host memory of one process - device-to-device copies and an accumulate kernel ordered
//             by HIP events across the ranks' streams.  RCCL refuses two ranks on one GPU; this is
//             how a host with a single GPU (and the tests) run the sharded render end to end:
//             ptw_render_ex(num_devices = N, share_device = 2).
#include "capi_common.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace ptw {
namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommInitAll) commInitAll = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclCommAbort) commAbort = nullptr;
  decltype(&ncclCommGetAsyncError) commGetAsyncError = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
};

const Rccl &rccl() {
  static Rccl api;
  static std::once_flag once;
  static std::string failure;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      failure = std::string("cannot load librccl: ") + dlerror();
      return;
    }
    auto sym = [&](auto &fn, const char *name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
      if (!fn && failure.empty()) failure = std::string("librccl lacks ") + name;
    };
    sym(api.getUniqueId, "ncclGetUniqueId");
    sym(api.commInitRank, "ncclCommInitRank");
    sym(api.commInitAll, "ncclCommInitAll");
    sym(api.commDestroy, "ncclCommDestroy");
    sym(api.commAbort, "ncclCommAbort");
    sym(api.commGetAsyncError, "ncclCommGetAsyncError");
    sym(api.getErrorString, "ncclGetErrorString");
    sym(api.reduce, "ncclReduce");
    sym(api.send, "ncclSend");
    sym(api.recv, "ncclRecv");
    sym(api.groupStart, "ncclGroupStart");
    sym(api.groupEnd, "ncclGroupEnd");
  });
  if (!failure.empty()) throw DeviceError(PTW_ERR_UNSUPPORTED, failure);
  return api;
}

void checkNccl(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return;
  throw DeviceError(PTW_ERR_HIP, std::string(what) + ": " + rccl().getErrorString(r));
}
void checkHip(hipError_t e, const char *what) {
  if (e == hipSuccess) return;
  throw DeviceError(PTW_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

static_assert(PTW_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

// How long a rank waits for its peers before it gives up on the communicator (ptw_comm_wait with
// timeout 0, and the host rendezvous of the loopback transport): PTW_COLLECTIVE_TIMEOUT_S, default
// 300 s - far above any framebuffer collective (462 MB over xGMI: milliseconds), far below forever.
std::chrono::milliseconds collectiveTimeout(int32_t timeoutMs = 0) {
  if (timeoutMs > 0) return std::chrono::milliseconds(timeoutMs);
  if (const char *v = std::getenv("PTW_COLLECTIVE_TIMEOUT_S")) {
    const double s = std::atof(v);
    if (s > 0) return std::chrono::milliseconds(static_cast<long long>(s * 1e3));
  }
  return std::chrono::milliseconds(300000);
}

// Runs `body` - a group of RCCL calls - so that it can be given up on.  The calls of a blocking
// communicator may wait on the HOST for their peers (RCCL connects two ranks at their first
// send / receive: a peer that died before that leaves ncclGroupEnd waiting for good), where no
// stream watchdog reaches.  So they run on a helper thread; if that thread is not back within the
// timeout, the caller aborts the communicator from here - ncclCommAbort is the one RCCL call that
// may be made while another thread is inside the library, and it is what unblocks that thread.
// (The handle is an atomic: whoever exchanges it for null - this function's timeout path, or
// ptw_comm_abort on another thread - is the one caller of ncclCommAbort; the gate's mutex keeps that call
// apart from ptw_comm_wait's ncclCommGetAsyncError on the same handle, which ncclCommAbort frees.)
// One helper thread per group of collective calls: a frame has ONE such group (the reduce or the gather
// at its end, seconds to minutes of rendering apart), so the thread's few tens of microseconds do not show.
//
// Enqueue against abort.  ncclCommAbort FREES the handle, so a thread that is about to START a
// group of calls on it must not meet an abort from another thread (ptw_render_ex: the shard whose collective
// failed aborts every shard's communicator from its own thread, while the healthy shards may only just be
// entering theirs).  `CallGate` makes the two exclusive without making the abort wait for a call that is
// stuck: the enqueuing thread takes the handle and marks the communicator "in a call" under the guard; an
// abort that arrives meanwhile leaves a request and waits a short grace period for the call to return - the
// normal case, an enqueue takes microseconds to milliseconds - after which the thread that made the call
// aborts its own communicator; only a call that is still inside RCCL after the grace period (it waits for
// a peer on the host) is aborted from outside, which is what ncclCommAbort is documented for.
struct CallGate {
  std::mutex m;                 // also keeps ncclCommAbort apart from ptw_comm_wait's ncclCommGetAsyncError
  std::condition_variable cv;
  bool inCall = false;          // a group of RCCL calls is being made on the handle
  bool abortRequested = false;  // ... and somebody wants the communicator gone when it returns
};
constexpr std::chrono::milliseconds kAbortGrace(2000);

// The one place that calls ncclCommAbort.  `fromCaller`: the thread that owns the in-flight call.
// Returns the abort's result (ncclSuccess when there was nothing left to abort).
ncclResult_t abortHandle(std::atomic<ncclComm_t> &comm, CallGate &gate, bool fromCaller) {
  std::unique_lock<std::mutex> lock(gate.m);
  if (!fromCaller && gate.inCall) {
    gate.abortRequested = true;
    if (gate.cv.wait_for(lock, kAbortGrace, [&] { return !gate.inCall; })) {
      // the call returned; its thread has aborted (or will find the request under this mutex) - nothing
      // is in flight on the handle now, so finishing the job here is safe either way
    }
  }
  ncclResult_t r = ncclSuccess;
  if (const ncclComm_t c = comm.exchange(nullptr)) r = rccl().commAbort(c);
  return r;
}

template <typename Body>
void runAbortable(std::atomic<ncclComm_t> &comm, CallGate &gate, int device, const char *what, Body &&body) {
  ncclComm_t nc = nullptr;
  {
    std::lock_guard<std::mutex> lock(gate.m);
    nc = comm.load();
    if (nc) gate.inCall = true;
  }
  if (!nc) throw DeviceError(PTW_ERR_HIP, "communicator aborted");
  auto leave = [&]() { // the call is over: wake an abort that waits for it; true if one was requested
    std::lock_guard<std::mutex> lock(gate.m);
    gate.inCall = false;
    const bool wanted = gate.abortRequested;
    gate.cv.notify_all();
    return wanted;
  };
  std::promise<void> done;
  std::future<void> fut = done.get_future();
  std::thread helper([&] {
    try {
      checkHip(hipSetDevice(device), "hipSetDevice");
      body(nc);
      done.set_value();
    } catch (...) {
      done.set_exception(std::current_exception());
    }
  });
  if (fut.wait_for(collectiveTimeout()) != std::future_status::ready) {
    // still inside RCCL at the deadline (a peer that never arrived): abort under the call - the one RCCL
    // call that may be made while another thread is inside the library, and what unblocks that thread
    {
      std::lock_guard<std::mutex> lock(gate.m);
      if (const ncclComm_t c = comm.exchange(nullptr)) (void)rccl().commAbort(c);
    }
    helper.join();
    (void)leave();
    throw DeviceError(PTW_ERR_HIP, std::string(what) + " did not return within the timeout (a peer that never "
                                       "arrived?): communicator aborted");
  }
  helper.join();
  if (leave()) {
    (void)abortHandle(comm, gate, true);
    throw DeviceError(PTW_ERR_HIP, "communicator aborted");
  }
  fut.get();
}

// Communicator set-up under the same deadline.  ncclCommInitRank blocks until EVERY rank of the world
// has called it: a rank that never arrives (a process that died before its set-up) used to leave the
// others in it for good, where no later watchdog reaches.  There is no communicator to abort yet, so
// the call runs on a helper thread that is ABANDONED when the deadline passes (it stays blocked in the
// bootstrap until the process ends; should it ever return, it destroys what it created): the caller gets
// PTW_ERR_HIP instead of a hang.
struct InitState {
  std::mutex m;
  std::condition_variable cv;
  bool done = false, abandoned = false;
  std::exception_ptr error;
  std::vector<ncclComm_t> comms;
};
// Every abandoned set-up leaves a thread inside ncclCommInitRank with its bootstrap sockets for the rest of
// the process: counted (ptw_comm_describe: `abandoned_setups`) and capped - a host that keeps
// retrying against a world that never assembles is told to restart instead of leaking a thread per attempt.
constexpr int kMaxAbandonedInits = 4;
std::atomic<int> &abandonedInits() {
  static std::atomic<int> n{0};
  return n;
}
template <typename Body>
std::vector<ncclComm_t> runInitAbortable(const char *what, Body body) {
  if (abandonedInits().load() >= kMaxAbandonedInits)
    throw DeviceError(PTW_ERR_HIP, std::string(what) + ": " + std::to_string(kMaxAbandonedInits) +
                                       " earlier communicator set-ups of this process were given up at their deadline and "
                                       "their threads are still inside RCCL's bootstrap - restart the process");
  auto state = std::make_shared<InitState>();
  std::thread helper([state, body] {
    std::vector<ncclComm_t> made;
    std::exception_ptr err;
    try {
      made = body();
    } catch (...) {
      err = std::current_exception();
    }
    std::unique_lock<std::mutex> lock(state->m);
    if (state->abandoned) { // nobody is waiting any more: nobody will own these
      lock.unlock();
      for (ncclComm_t c : made)
        if (c) (void)rccl().commAbort(c);
      return;
    }
    state->comms = std::move(made);
    state->error = err;
    state->done = true;
    state->cv.notify_all();
  });
  std::unique_lock<std::mutex> lock(state->m);
  if (!state->cv.wait_for(lock, collectiveTimeout(), [&] { return state->done; })) {
    state->abandoned = true;
    lock.unlock();
    helper.detach();
    abandonedInits().fetch_add(1);
    throw DeviceError(PTW_ERR_HIP, std::string(what) + " did not return within the timeout (PTW_COLLECTIVE_TIMEOUT_S; a rank "
                                       "that never called it?): communicator set-up given up");
  }
  lock.unlock();
  helper.join();
  if (state->error) std::rethrow_exception(state->error);
  return state->comms;
}

// ---- loopback transport ---------------------------------------------------------------------
// A message is a device buffer plus the event after which its contents are final on the sender's
// stream.  The receiver makes its own stream wait for that event, consumes the buffer (copy or
// accumulate), records a `consumed` event and hands it back; the sender's stream waits for it
// before it may touch the buffer again.  The host threads of the ranks rendezvous on a mutex +
// condition variable; abort() wakes every waiter with an error.
struct LoopMessage {
  const void *ptr = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr;    // owned by the sender (one per peer and channel, re-recorded per message)
  hipEvent_t consumed = nullptr; // owned by the receiver (likewise)
  int state = 0;                 // 0 empty, 1 posted, 2 consumed
};

struct LoopbackHub {
  std::mutex m;
  std::condition_variable cv;
  int world = 0;
  bool aborted = false;
  std::vector<LoopMessage> box; // [src * world + dst] * 2 + channel (two messages per pair in flight)
  explicit LoopbackHub(int w) : world(w), box(static_cast<size_t>(w) * w * 2) {}
  LoopMessage &at(int src, int dst, int channel) {
    return box[(static_cast<size_t>(src) * world + dst) * 2 + channel];
  }
  void abort() {
    std::lock_guard<std::mutex> lock(m);
    aborted = true;
    cv.notify_all();
  }
  // Waits (lock held) until `ready()` or the communicator is given up; a peer that never arrives is
  // an error after collectiveTimeout(), not a hang - and the error aborts the hub, so that whoever
  // waits for THIS rank is released too.
  template <typename Ready>
  void await(std::unique_lock<std::mutex> &lock, const char *what, Ready &&ready) {
    const bool ok = cv.wait_for(lock, collectiveTimeout(), [&] { return aborted || ready(); });
    if (aborted) throw DeviceError(PTW_ERR_HIP, "communicator aborted");
    if (!ok) {
      aborted = true;
      cv.notify_all();
      throw DeviceError(PTW_ERR_HIP, std::string("timed out waiting for a peer (") + what +
                                         "; PTW_COLLECTIVE_TIMEOUT_S): communicator aborted");
    }
  }
};

__global__ void loopAccumulateF64(double *__restrict__ dst, const double *__restrict__ src, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] += src[i];
}
__global__ void loopAccumulateU32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] += src[i];
}

} // namespace
} // namespace ptw

using namespace ptw;

struct ptw_comm {
  std::atomic<ncclComm_t> comm{nullptr}; // RCCL transport (null once aborted: see runAbortable)
  CallGate gate;                          // enqueue / ncclCommGetAsyncError against ncclCommAbort (frees the handle)
  std::shared_ptr<LoopbackHub> hub;   // loopback transport (comm == nullptr)
  // loopback: one `ready` event per (destination, channel) and one `consumed` event per (source,
  // channel), created on first use and re-recorded for every message - a long-lived communicator
  // holds 4 x world events, however many collectives it has run
  std::vector<hipEvent_t> events;
  int world = 1, rank = 0, device = 0;
  // packed rows of the gather: [rows][width][3] doubles then [rows][width] u32, per rank slot
  void *pack = nullptr;
  size_t packBytes = 0;
  ~ptw_comm() {
    (void)hipSetDevice(device);
    if (pack) (void)hipFree(pack);
    for (hipEvent_t e : events)
      if (e) (void)hipEventDestroy(e);
    if (const ncclComm_t c = comm.exchange(nullptr)) (void)rccl().commDestroy(c);
  }
  hipEvent_t eventFor(int kind, int peer, int channel) { // kind 0: ready (send), 1: consumed (recv)
    if (events.empty()) events.assign(static_cast<size_t>(world) * 4, nullptr);
    hipEvent_t &e = events[(static_cast<size_t>(kind) * world + peer) * 2 + channel];
    if (!e) checkHip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    return e;
  }
  // ---- the two point-to-point primitives the collectives are written in -----------------------
  // (RCCL: ncclSend / ncclRecv inside the caller's group; loopback: see LoopbackHub)
  void loopSend(const void *ptr, size_t bytes, int dst, int channel, hipStream_t stream) {
    // (re-recording the event of the previous message to this peer is safe: that message reached
    // state 2, i.e. the receiver's hipStreamWaitEvent on it has been issued)
    const hipEvent_t ready = eventFor(0, dst, channel);
    checkHip(hipEventRecord(ready, stream), "hipEventRecord");
    hipEvent_t consumed = nullptr;
    {
      std::unique_lock<std::mutex> lock(hub->m);
      LoopMessage &msg = hub->at(rank, dst, channel);
      hub->await(lock, "send: slot free", [&] { return msg.state == 0; });
      msg.ptr = ptr, msg.bytes = bytes, msg.ready = ready, msg.state = 1;
      hub->cv.notify_all();
      hub->await(lock, "send: receiver", [&] { return msg.state == 2; });
      consumed = msg.consumed;
      msg = LoopMessage();
      hub->cv.notify_all();
    }
    // the buffer may be reused on this stream only after the receiver has consumed it
    checkHip(hipStreamWaitEvent(stream, consumed, 0), "hipStreamWaitEvent");
  }
  // Waits for the message, lets `consume(ptr, bytes)` enqueue its work on `stream`.
  template <typename Consume>
  void loopRecv(int src, int channel, size_t expectBytes, hipStream_t stream, Consume &&consume) {
    LoopMessage got;
    {
      std::unique_lock<std::mutex> lock(hub->m);
      LoopMessage &msg = hub->at(src, rank, channel);
      hub->await(lock, "receive: sender", [&] { return msg.state == 1; });
      got = msg;
    }
    if (got.bytes != expectBytes) {
      hub->abort(); // the sender waits for this message to be consumed: release it
      throw DeviceError(PTW_ERR_SIZE_MISMATCH, "loopback message of " + std::to_string(got.bytes) +
                                                   " bytes where " + std::to_string(expectBytes) + " were expected");
    }
    checkHip(hipStreamWaitEvent(stream, got.ready, 0), "hipStreamWaitEvent");
    consume(got.ptr, got.bytes);
    // (the previous `consumed` event of this pair has been waited for by the sender's stream before
    // the sender could post this message)
    const hipEvent_t consumed = eventFor(1, src, channel);
    checkHip(hipEventRecord(consumed, stream), "hipEventRecord");
    {
      std::lock_guard<std::mutex> lock(hub->m);
      LoopMessage &msg = hub->at(src, rank, channel);
      msg.consumed = consumed, msg.state = 2;
      hub->cv.notify_all();
    }
  }
  void reservePack(size_t bytes) {
    if (bytes <= packBytes) return;
    if (pack) (void)hipFree(pack);
    pack = nullptr;
    packBytes = 0;
    checkHip(hipMalloc(&pack, bytes), "hipMalloc");
    packBytes = bytes;
  }
};

#define PTW_GUARD_BEGIN try {
#define PTW_GUARD_END                                                                          \
  }                                                                                            \
  catch (...) {                                                                                \
    return translateException();                                                               \
  }

extern "C" {

int ptw_comm_unique_id(uint8_t id_out[PTW_COMM_ID_BYTES]) {
  if (!id_out) return invalid("id_out");
  PTW_GUARD_BEGIN
  ncclUniqueId id;
  checkNccl(rccl().getUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id_out, id.internal, PTW_COMM_ID_BYTES);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_create(const uint8_t id[PTW_COMM_ID_BYTES], int32_t world_size, int32_t rank,
                    int32_t device, ptw_comm **out) {
  if (!id || !out) return invalid("null pointer");
  if (world_size < 1 || rank < 0 || rank >= world_size) return invalid("world_size / rank");
  PTW_GUARD_BEGIN
  checkHip(hipSetDevice(device), "hipSetDevice");
  auto c = std::make_unique<ptw_comm>();
  c->world = world_size;
  c->rank = rank;
  c->device = device;
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, PTW_COMM_ID_BYTES);
  const Rccl &api = rccl();
  const std::vector<ncclComm_t> made = runInitAbortable("ncclCommInitRank", [&api, world_size, uid, rank, device] {
    checkHip(hipSetDevice(device), "hipSetDevice");
    ncclComm_t raw = nullptr;
    checkNccl(api.commInitRank(&raw, world_size, uid, rank), "ncclCommInitRank");
    return std::vector<ncclComm_t>{raw};
  });
  c->comm = made.at(0);
  *out = c.release();
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_create_all(int32_t num_devices, const int32_t *devices, ptw_comm **out_comms) {
  if (!out_comms || num_devices < 1) return invalid("num_devices / out_comms");
  PTW_GUARD_BEGIN
  std::vector<int> devs(static_cast<size_t>(num_devices));
  for (int i = 0; i < num_devices; ++i) devs[i] = devices ? devices[i] : i;
  const Rccl &api = rccl();
  const std::vector<ncclComm_t> raw = runInitAbortable("ncclCommInitAll", [&api, num_devices, devs] {
    std::vector<ncclComm_t> made(static_cast<size_t>(num_devices), nullptr);
    checkNccl(api.commInitAll(made.data(), num_devices, devs.data()), "ncclCommInitAll");
    return made;
  });
  for (int i = 0; i < num_devices; ++i) {
    auto *c = new ptw_comm;
    c->comm = raw[i];
    c->world = num_devices;
    c->rank = i;
    c->device = devs[i];
    out_comms[i] = c;
  }
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_create_loopback(int32_t world_size, int32_t device, ptw_comm **out_comms) {
  if (!out_comms || world_size < 1) return invalid("world_size / out_comms");
  PTW_GUARD_BEGIN
  checkHip(hipSetDevice(device), "hipSetDevice");
  auto hub = std::make_shared<LoopbackHub>(world_size);
  for (int i = 0; i < world_size; ++i) {
    auto *c = new ptw_comm;
    c->hub = hub;
    c->world = world_size;
    c->rank = i;
    c->device = device;
    out_comms[i] = c;
  }
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_abort(ptw_comm *comm) {
  if (!comm) return invalid("comm");
  PTW_GUARD_BEGIN
  if (comm->hub) {
    comm->hub->abort();
  } else {
    // ncclCommAbort frees the communicator: kernels of it that wait for a peer on the device end.  Not
    // while ptw_comm_wait looks at the handle, and not while another thread is STARTING calls on it (CallGate)
    checkNccl(abortHandle(comm->comm, comm->gate, false), "ncclCommAbort");
  }
  return PTW_OK;
  PTW_GUARD_END
}

void ptw_comm_destroy(ptw_comm *comm) { delete comm; }

int ptw_comm_wait(ptw_comm *comm, void *hip_stream, int32_t timeout_ms) {
  if (!comm) return invalid("comm");
  if (timeout_ms < 0) return invalid("timeout_ms");
  PTW_GUARD_BEGIN
  checkHip(hipSetDevice(comm->device), "hipSetDevice");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const auto deadline = std::chrono::steady_clock::now() + collectiveTimeout(timeout_ms);
  auto giveUp = [&](const std::string &why) {
    (void)ptw_comm_abort(comm); // ends the communicator's kernels on the device / wakes the rendezvous
    throw DeviceError(PTW_ERR_HIP, why + ": communicator aborted");
  };
  for (unsigned spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return PTW_OK;
    if (q != hipErrorNotReady) checkHip(q, "hipStreamQuery");
    if (comm->hub) {
      bool aborted;
      {
        std::lock_guard<std::mutex> lock(comm->hub->m);
        aborted = comm->hub->aborted;
      }
      if (aborted) throw DeviceError(PTW_ERR_HIP, "communicator aborted");
    } else {
      // (the handle is looked at under the guard: a ptw_comm_abort on another thread - the documented way
      // to release a rank that waits here - frees it)
      ncclResult_t r = ncclSuccess, async = ncclSuccess;
      bool gone = false;
      {
        std::lock_guard<std::mutex> lock(comm->gate.m);
        if (const ncclComm_t nc = comm->comm.load())
          r = rccl().commGetAsyncError(nc, &async);
        else
          gone = true;
      }
      if (gone) throw DeviceError(PTW_ERR_HIP, "communicator aborted");
      if (r != ncclSuccess) giveUp(std::string("ncclCommGetAsyncError: ") + rccl().getErrorString(r));
      if (async != ncclSuccess && async != ncclInProgress)
        giveUp(std::string("asynchronous RCCL error: ") + rccl().getErrorString(async));
    }
    if (std::chrono::steady_clock::now() >= deadline)
      giveUp("the collective did not complete within the timeout (a peer that never arrived?)");
    // a framebuffer collective takes well under a millisecond to a few: poll closely first
    std::this_thread::sleep_for(std::chrono::microseconds(spin < 200 ? 50 : 1000));
  }
  PTW_GUARD_END
}

} // extern "C"

// ---- which wire? ---------------------------------------------------------------------------
namespace {
const char *linkTypeName(uint32_t t) {
  switch (t) { // hsa_amd_link_info_type_t
  case 0: return "hypertransport";
  case 1: return "qpi";
  case 2: return "pcie";
  case 3: return "infiniband";
  case 4: return "xgmi";
  default: return "unknown";
  }
}
bool envIsOne(const char *name) { // RCCL reads these as integers: any non-zero value switches the transport off
  const char *v = std::getenv(name);
  return v && *v && std::strtol(v, nullptr, 0) != 0;
}
// NCCL_DEBUG_FILE with RCCL's %h (host name) and %p (process id) filled in; empty: not set
std::string rcclDebugFile() {
  const char *v = std::getenv("NCCL_DEBUG_FILE");
  if (!v || !*v) return std::string();
  std::string out;
  for (const char *c = v; *c; ++c) {
    if (c[0] == '%' && c[1] == 'h') {
      char host[256] = "";
      (void)gethostname(host, sizeof host - 1);
      out += host;
      ++c;
    } else if (c[0] == '%' && c[1] == 'p') {
      out += std::to_string(static_cast<long>(getpid()));
      ++c;
    } else {
      out += *c;
    }
  }
  return out;
}
} // namespace

extern "C" {

int ptw_comm_describe(ptw_comm *comm, char *out, size_t capacity) {
  if (!comm || !out || capacity == 0) return invalid("null pointer");
  PTW_GUARD_BEGIN
  std::ostringstream js;
  js << "{\"kind\": \"" << (comm->hub ? "loopback" : "rccl") << "\", \"world\": " << comm->world << ", \"rank\": " << comm->rank
     << ", \"device\": " << comm->device;
  if (comm->hub) {
    js << ", \"expected\": \"in-process device-to-device copies (one GPU)\", \"links\": [], \"rccl_log\": null}";
  } else {
    // the wires of this rank's GPU: link type and hops to every other visible device (HIP's own report)
    int count = 0;
    (void)hipGetDeviceCount(&count);
    bool allXgmi = count > 1, anyPeer = false;
    js << ", \"links\": [";
    bool first = true;
    for (int d = 0; d < count; ++d) {
      if (d == comm->device) continue;
      uint32_t type = 0xffffffffu, hops = 0;
      const hipError_t e = hipExtGetLinkTypeAndHopCount(comm->device, d, &type, &hops);
      int canAccess = 0;
      (void)hipDeviceCanAccessPeer(&canAccess, comm->device, d);
      if (!first) js << ", ";
      first = false;
      js << "{\"device\": " << d << ", \"type\": \"" << (e == hipSuccess ? linkTypeName(type) : "unknown") << "\", \"hops\": " << hops
         << ", \"peer_access\": " << (canAccess ? "true" : "false") << "}";
      anyPeer = true;
      if (!(e == hipSuccess && type == 4 && canAccess)) allXgmi = false;
    }
    js << "]";
    const bool p2pOff = envIsOne("NCCL_P2P_DISABLE"), shmOff = envIsOne("NCCL_SHM_DISABLE");
    js << ", \"p2p_disabled\": " << (p2pOff ? "true" : "false") << ", \"shm_disabled\": " << (shmOff ? "true" : "false");
    const char *expected = p2pOff ? (shmOff ? "NET/Socket" : "SHM")
                                  : (!anyPeer ? "one visible GPU: the peers are other hosts to RCCL (NET)"
                                              : (allXgmi ? "P2P/xGMI" : "P2P (not every peer over xGMI: see links)"));
    // (`expected` is a GUESS from HIP's link types and the two switches above: it does not know
    // NCCL_P2P_LEVEL, HIP_VISIBLE_DEVICES orderings or worlds that span hosts - `rccl_log` is what RCCL says)
    js << ", \"expected\": \"" << expected << "\", \"expected_is\": \"a guess from HIP link types and NCCL_P2P_DISABLE / NCCL_SHM_DISABLE\"";
    js << ", \"rccl_log_scope\": \"every communicator of this process that logs to the file\"";
    js << ", \"abandoned_setups\": " << abandonedInits().load();
    // ... and what RCCL says it chose, when its log goes to a file (NCCL_DEBUG=INFO, NCCL_DEBUG_FILE): the
    // transports named after "via" in its channel lines
    const std::string file = rcclDebugFile();
    std::set<std::string> seen;
    bool haveLog = false;
    if (!file.empty()) {
      std::ifstream in(file);
      std::string line;
      while (std::getline(in, line)) {
        const size_t at = line.find(" via ");
        if (at == std::string::npos) continue;
        std::string word = line.substr(at + 5);
        const size_t end = word.find_first_of(" \t\r\n");
        if (end != std::string::npos) word.resize(end);
        if (!word.empty() && word.size() < 48 && word.find('"') == std::string::npos) seen.insert(word), haveLog = true;
      }
    }
    js << ", \"rccl_log_file\": ";
    if (file.empty()) js << "null";
    else js << "\"" << (file.find_first_of("\"\\") == std::string::npos ? file : std::string("(unprintable)")) << "\"";
    if (!haveLog) { // no file, or one without channel lines (NCCL_DEBUG below INFO): RCCL did not say
      js << ", \"rccl_log\": null}";
    } else {
      js << ", \"rccl_log\": [";
      bool f2 = true;
      for (const std::string &w : seen) {
        if (!f2) js << ", ";
        f2 = false;
        js << "\"" << w << "\"";
      }
      js << "]}";
    }
  }
  const std::string text = js.str();
  if (text.size() + 1 > capacity) throw std::invalid_argument("ptw_comm_describe: buffer too small");
  std::memcpy(out, text.c_str(), text.size() + 1);
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_reduce_framebuffer(ptw_comm *comm, void *d_rgb_sum, void *d_counts, uint64_t npix,
                                int32_t root, void *hip_stream) {
  if (!comm || !d_rgb_sum || !d_counts) return invalid("null pointer");
  if (root < 0 || root >= comm->world) return invalid("root");
  PTW_GUARD_BEGIN
  checkHip(hipSetDevice(comm->device), "hipSetDevice");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  if (comm->hub) {
    // loopback: the root adds the other ranks' buffers to its own, in rank order
    if (comm->world == 1) return PTW_OK;
    if (comm->rank != root) {
      comm->loopSend(d_rgb_sum, npix * 3 * sizeof(double), root, 0, stream);
      comm->loopSend(d_counts, npix * sizeof(uint32_t), root, 1, stream);
      return PTW_OK;
    }
    for (int r = 0; r < comm->world; ++r) {
      if (r == root) continue;
      comm->loopRecv(r, 0, npix * 3 * sizeof(double), stream, [&](const void *src, size_t) {
        hipLaunchKernelGGL(loopAccumulateF64, dim3(1024), dim3(256), 0, stream, static_cast<double *>(d_rgb_sum),
                           static_cast<const double *>(src), static_cast<size_t>(npix) * 3);
        checkHip(hipGetLastError(), "accumulate launch");
      });
      comm->loopRecv(r, 1, npix * sizeof(uint32_t), stream, [&](const void *src, size_t) {
        hipLaunchKernelGGL(loopAccumulateU32, dim3(1024), dim3(256), 0, stream, static_cast<uint32_t *>(d_counts),
                           static_cast<const uint32_t *>(src), static_cast<size_t>(npix));
        checkHip(hipGetLastError(), "accumulate launch");
      });
    }
    return PTW_OK;
  }
  const Rccl &api = rccl();
  // (the handle is taken inside runAbortable, under the gate: an abort from another thread either comes
  // first - "communicator aborted" - or waits for these calls to return)
  runAbortable(comm->comm, comm->gate, comm->device, "ncclReduce of the framebuffer", [&](ncclComm_t nc) {
    // one group: both reductions are launched together
    checkNccl(api.groupStart(), "ncclGroupStart");
    checkNccl(api.reduce(d_rgb_sum, d_rgb_sum, npix * 3, ncclDouble, ncclSum, root, nc, stream),
              "ncclReduce(rgb_sum)");
    checkNccl(api.reduce(d_counts, d_counts, npix, ncclUint32, ncclSum, root, nc, stream),
              "ncclReduce(counts)");
    checkNccl(api.groupEnd(), "ncclGroupEnd");
  });
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_gather_rows(ptw_comm *comm, void *d_rgb_sum, void *d_counts, int32_t width,
                         int32_t height, int32_t root, void *hip_stream) {
  if (!comm || !d_rgb_sum || !d_counts) return invalid("null pointer");
  if (width <= 0 || height <= 0) return invalid("width / height");
  if (root < 0 || root >= comm->world) return invalid("root");
  PTW_GUARD_BEGIN
  checkHip(hipSetDevice(comm->device), "hipSetDevice");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const int world = comm->world, rank = comm->rank;
  if (world == 1) return PTW_OK;
  const bool loop = static_cast<bool>(comm->hub);
  if (!loop && !comm->comm.load()) throw DeviceError(PTW_ERR_HIP, "communicator aborted");
  const Rccl *api = loop ? nullptr : &rccl();
  const size_t w = static_cast<size_t>(width);
  const size_t rgbRow = w * 3 * sizeof(double), cntRow = w * sizeof(uint32_t);
  auto rowsOf = [&](int r) { return static_cast<size_t>(height > r ? (height - r + world - 1) / world : 0); };
  const size_t maxRows = rowsOf(0);
  const size_t slotRgb = maxRows * rgbRow, slotCnt = maxRows * cntRow;
  auto *rgb = static_cast<char *>(d_rgb_sum);
  auto *cnt = static_cast<char *>(d_counts);

  if (rank != root) {
    // pack my rows (row y = rank + k * world) into a contiguous buffer, send it
    const size_t rows = rowsOf(rank);
    comm->reservePack(slotRgb + slotCnt);
    char *packRgb = static_cast<char *>(comm->pack), *packCnt = packRgb + slotRgb;
    if (rows) {
      checkHip(hipMemcpy2DAsync(packRgb, rgbRow, rgb + rank * rgbRow, world * rgbRow, rgbRow, rows,
                                hipMemcpyDeviceToDevice, stream), "pack rgb");
      checkHip(hipMemcpy2DAsync(packCnt, cntRow, cnt + rank * cntRow, world * cntRow, cntRow, rows,
                                hipMemcpyDeviceToDevice, stream), "pack counts");
    }
    if (loop) {
      if (rows) {
        comm->loopSend(packRgb, rows * rgbRow, root, 0, stream);
        comm->loopSend(packCnt, rows * cntRow, root, 1, stream);
      }
      return PTW_OK;
    }
    runAbortable(comm->comm, comm->gate, comm->device, "ncclSend of the rows", [&](ncclComm_t nc) {
      checkNccl(api->groupStart(), "ncclGroupStart");
      if (rows) {
        checkNccl(api->send(packRgb, rows * w * 3, ncclDouble, root, nc, stream), "ncclSend(rgb)");
        checkNccl(api->send(packCnt, rows * w, ncclUint32, root, nc, stream), "ncclSend(counts)");
      }
      checkNccl(api->groupEnd(), "ncclGroupEnd");
    });
    return PTW_OK;
  }
  // root: receive every other rank's packed rows, then scatter them into their image rows
  comm->reservePack(static_cast<size_t>(world) * (slotRgb + slotCnt));
  char *base = static_cast<char *>(comm->pack);
  if (loop) {
    for (int r = 0; r < world; ++r) {
      if (r == root || rowsOf(r) == 0) continue;
      char *slot = base + static_cast<size_t>(r) * (slotRgb + slotCnt);
      comm->loopRecv(r, 0, rowsOf(r) * rgbRow, stream, [&](const void *src, size_t bytes) {
        checkHip(hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToDevice, stream), "recv rgb");
      });
      comm->loopRecv(r, 1, rowsOf(r) * cntRow, stream, [&](const void *src, size_t bytes) {
        checkHip(hipMemcpyAsync(slot + slotRgb, src, bytes, hipMemcpyDeviceToDevice, stream), "recv counts");
      });
    }
  } else {
    runAbortable(comm->comm, comm->gate, comm->device, "ncclRecv of the rows", [&](ncclComm_t nc) {
      checkNccl(api->groupStart(), "ncclGroupStart");
      for (int r = 0; r < world; ++r) {
        if (r == root || rowsOf(r) == 0) continue;
        char *slot = base + static_cast<size_t>(r) * (slotRgb + slotCnt);
        checkNccl(api->recv(slot, rowsOf(r) * w * 3, ncclDouble, r, nc, stream), "ncclRecv(rgb)");
        checkNccl(api->recv(slot + slotRgb, rowsOf(r) * w, ncclUint32, r, nc, stream), "ncclRecv(counts)");
      }
      checkNccl(api->groupEnd(), "ncclGroupEnd");
    });
  }
  for (int r = 0; r < world; ++r) {
    if (r == root || rowsOf(r) == 0) continue;
    char *slot = base + static_cast<size_t>(r) * (slotRgb + slotCnt);
    checkHip(hipMemcpy2DAsync(rgb + r * rgbRow, world * rgbRow, slot, rgbRow, rgbRow, rowsOf(r),
                              hipMemcpyDeviceToDevice, stream), "unpack rgb");
    checkHip(hipMemcpy2DAsync(cnt + r * cntRow, world * cntRow, slot + slotRgb, cntRow, cntRow, rowsOf(r),
                              hipMemcpyDeviceToDevice, stream), "unpack counts");
  }
  return PTW_OK;
  PTW_GUARD_END
}

} // extern "C"

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
#include "capi_common.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace ptw {
namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommInitAll) commInitAll = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
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

} // namespace
} // namespace ptw

using namespace ptw;

struct ptw_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  // packed rows of the gather: [rows][width][3] doubles then [rows][width] u32, per rank slot
  void *pack = nullptr;
  size_t packBytes = 0;
  ~ptw_comm() {
    (void)hipSetDevice(device);
    if (pack) (void)hipFree(pack);
    if (comm) (void)rccl().commDestroy(comm);
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
  checkNccl(rccl().commInitRank(&c->comm, world_size, uid, rank), "ncclCommInitRank");
  *out = c.release();
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_create_all(int32_t num_devices, const int32_t *devices, ptw_comm **out_comms) {
  if (!out_comms || num_devices < 1) return invalid("num_devices / out_comms");
  PTW_GUARD_BEGIN
  std::vector<int> devs(static_cast<size_t>(num_devices));
  for (int i = 0; i < num_devices; ++i) devs[i] = devices ? devices[i] : i;
  std::vector<ncclComm_t> raw(static_cast<size_t>(num_devices), nullptr);
  checkNccl(rccl().commInitAll(raw.data(), num_devices, devs.data()), "ncclCommInitAll");
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

void ptw_comm_destroy(ptw_comm *comm) { delete comm; }

int ptw_comm_reduce_framebuffer(ptw_comm *comm, void *d_rgb_sum, void *d_counts, uint64_t npix,
                                int32_t root, void *hip_stream) {
  if (!comm || !d_rgb_sum || !d_counts) return invalid("null pointer");
  if (root < 0 || root >= comm->world) return invalid("root");
  PTW_GUARD_BEGIN
  const Rccl &api = rccl();
  checkHip(hipSetDevice(comm->device), "hipSetDevice");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  // one group: both reductions are launched together
  checkNccl(api.groupStart(), "ncclGroupStart");
  checkNccl(api.reduce(d_rgb_sum, d_rgb_sum, npix * 3, ncclDouble, ncclSum, root, comm->comm, stream),
            "ncclReduce(rgb_sum)");
  checkNccl(api.reduce(d_counts, d_counts, npix, ncclUint32, ncclSum, root, comm->comm, stream),
            "ncclReduce(counts)");
  checkNccl(api.groupEnd(), "ncclGroupEnd");
  return PTW_OK;
  PTW_GUARD_END
}

int ptw_comm_gather_rows(ptw_comm *comm, void *d_rgb_sum, void *d_counts, int32_t width,
                         int32_t height, int32_t root, void *hip_stream) {
  if (!comm || !d_rgb_sum || !d_counts) return invalid("null pointer");
  if (width <= 0 || height <= 0) return invalid("width / height");
  if (root < 0 || root >= comm->world) return invalid("root");
  PTW_GUARD_BEGIN
  const Rccl &api = rccl();
  checkHip(hipSetDevice(comm->device), "hipSetDevice");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const int world = comm->world, rank = comm->rank;
  if (world == 1) return PTW_OK;
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
    checkNccl(api.groupStart(), "ncclGroupStart");
    if (rows) {
      checkNccl(api.send(packRgb, rows * w * 3, ncclDouble, root, comm->comm, stream), "ncclSend(rgb)");
      checkNccl(api.send(packCnt, rows * w, ncclUint32, root, comm->comm, stream), "ncclSend(counts)");
    }
    checkNccl(api.groupEnd(), "ncclGroupEnd");
    return PTW_OK;
  }
  // root: receive every other rank's packed rows, then scatter them into their image rows
  comm->reservePack(static_cast<size_t>(world) * (slotRgb + slotCnt));
  char *base = static_cast<char *>(comm->pack);
  checkNccl(api.groupStart(), "ncclGroupStart");
  for (int r = 0; r < world; ++r) {
    if (r == root || rowsOf(r) == 0) continue;
    char *slot = base + static_cast<size_t>(r) * (slotRgb + slotCnt);
    checkNccl(api.recv(slot, rowsOf(r) * w * 3, ncclDouble, r, comm->comm, stream), "ncclRecv(rgb)");
    checkNccl(api.recv(slot + slotRgb, rowsOf(r) * w, ncclUint32, r, comm->comm, stream), "ncclRecv(counts)");
  }
  checkNccl(api.groupEnd(), "ncclGroupEnd");
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

// The ONE collective of a multi-GPU run, behind the C ABI.
//
// Queries are independent and the model is replicated per GPU, so nothing on the data path
// crosses GPUs.  At the end of a run every rank holds a latency histogram and a few scalars
// (query count, sum of latencies, first / last completion time); the orchestrator's formulae
// (reference DeepRecSys.py:168-175: QPS = completed queries / (last - first
// inference_end_time), p95/p99 over all response latencies) need their SUM / MIN / MAX over
// the ranks.  That is one grouped RCCL all-reduce of ~32 KB over xGMI: latency-bound (tens of
// microseconds), nowhere near the 7 x ~153 GB/s per-link ceiling.
//
// RCCL is bound at run time (dlopen): libdrs_hip.so keeps loading on a box without RCCL, and
// when the host process already carries a copy (PyTorch-ROCm bundles one) that copy is the one
// used -- two RCCLs in one process would fight over the same symbols.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#include "drs_internal.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;   // why loading failed
};

thread_local std::string g_comm_error;

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.why = std::string("RCCL is not loadable: ") + dlerror(); return; }
#define SYM(field, name)                                                     \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));         \
  if (!r.field) { r.why = std::string("RCCL lacks ") + name; r.lib = nullptr; return; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  });
  return r.lib ? &r : nullptr;
}

int32_t cfail(int32_t code, const char* what, const char* detail) {
  g_comm_error = std::string(what) + ": " + (detail ? detail : "");
  return code;
}

}  // namespace

struct drs_comm_s {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void* d_buf = nullptr;     // [hist int64 x cap | 4 doubles]
  size_t cap_bins = 0;
};

#define NCCL_TRY(r, call)                                                              \
  do {                                                                                 \
    ncclResult_t _n = (call);                                                          \
    if (_n != ncclSuccess) return cfail(DRS_ERR_HIP, #call, (r)->GetErrorString(_n));  \
  } while (0)
#define CHIP_TRY(call)                                                                 \
  do {                                                                                 \
    hipError_t _h = (call);                                                            \
    if (_h != hipSuccess)                                                              \
      return cfail(_h == hipErrorOutOfMemory ? DRS_ERR_OOM : DRS_ERR_HIP, #call, hipGetErrorString(_h)); \
  } while (0)

extern "C" {

const char* drs_comm_last_error(void) { return g_comm_error.c_str(); }

int32_t drs_comm_unique_id(uint8_t* id) {
  if (!id) return cfail(DRS_ERR_BAD_ARG, "drs_comm_unique_id", "null id");
  Rccl* r = rccl();
  if (!r) return cfail(DRS_ERR_UNSUPPORTED, "drs_comm_unique_id", rccl() ? "" : "RCCL not loadable");
  ncclUniqueId u;
  NCCL_TRY(r, r->GetUniqueId(&u));
  static_assert(sizeof u == DRS_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &u, sizeof u);
  return DRS_OK;
}

int32_t drs_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device_id, drs_comm* out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return cfail(DRS_ERR_BAD_ARG, "drs_comm_create", "bad rank / world / id");
  *out = nullptr;
  Rccl* r = rccl();
  if (!r) return cfail(DRS_ERR_UNSUPPORTED, "drs_comm_create", "RCCL not loadable");
  CHIP_TRY(hipSetDevice(device_id));
  drs_comm_s* c = new drs_comm_s();
  c->rank = rank; c->world = world; c->device = device_id;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclResult_t n = r->CommInitRank(&c->comm, world, u, rank);
  if (n != ncclSuccess) { delete c; return cfail(DRS_ERR_HIP, "ncclCommInitRank", r->GetErrorString(n)); }
  hipError_t h = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (h != hipSuccess) { r->CommDestroy(c->comm); delete c; return cfail(DRS_ERR_HIP, "hipStreamCreate", hipGetErrorString(h)); }
  *out = c;
  return DRS_OK;
}

int32_t drs_comm_destroy(drs_comm c) {
  if (!c) return DRS_OK;
  (void)hipSetDevice(c->device);
  Rccl* r = rccl();
  if (c->stream) { (void)hipStreamSynchronize(c->stream); }
  if (r && c->comm) r->CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_buf) (void)hipFree(c->d_buf);
  delete c;
  return DRS_OK;
}

static int32_t ensure_buf(drs_comm c, size_t bins) {
  if (c->d_buf && c->cap_bins >= bins) return DRS_OK;
  if (c->d_buf) (void)hipFree(c->d_buf);
  c->d_buf = nullptr;
  CHIP_TRY(hipMalloc(&c->d_buf, sizeof(int64_t) * bins + sizeof(double) * 4));
  c->cap_bins = bins;
  return DRS_OK;
}

// hist[nbins]: SUM over ranks.  sum_min_max[4]: [0] SUM, [1] SUM, [2] MIN, [3] MAX over ranks
// (query count, sum of latencies, first and last completion time).  In place, every rank gets
// the result.  One grouped RCCL launch on the communicator's own stream.
int32_t drs_stats_allreduce(drs_comm c, int64_t* hist, int32_t nbins, double* sum_min_max) {
  if (!c || nbins < 0 || (nbins > 0 && !hist) || !sum_min_max) return cfail(DRS_ERR_BAD_ARG, "drs_stats_allreduce", "bad arguments");
  Rccl* r = rccl();
  if (!r) return cfail(DRS_ERR_UNSUPPORTED, "drs_stats_allreduce", "RCCL not loadable");
  CHIP_TRY(hipSetDevice(c->device));
  const size_t bins = nbins > 0 ? (size_t)nbins : 1;
  int32_t rc = ensure_buf(c, bins);
  if (rc) return rc;
  int64_t* d_hist = reinterpret_cast<int64_t*>(c->d_buf);
  double* d_s = reinterpret_cast<double*>(d_hist + c->cap_bins);
  if (nbins > 0) CHIP_TRY(hipMemcpyAsync(d_hist, hist, sizeof(int64_t) * nbins, hipMemcpyHostToDevice, c->stream));
  CHIP_TRY(hipMemcpyAsync(d_s, sum_min_max, sizeof(double) * 4, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(r, r->GroupStart());
  if (nbins > 0) NCCL_TRY(r, r->AllReduce(d_hist, d_hist, (size_t)nbins, ncclInt64, ncclSum, c->comm, c->stream));
  NCCL_TRY(r, r->AllReduce(d_s, d_s, 2, ncclFloat64, ncclSum, c->comm, c->stream));
  NCCL_TRY(r, r->AllReduce(d_s + 2, d_s + 2, 1, ncclFloat64, ncclMin, c->comm, c->stream));
  NCCL_TRY(r, r->AllReduce(d_s + 3, d_s + 3, 1, ncclFloat64, ncclMax, c->comm, c->stream));
  NCCL_TRY(r, r->GroupEnd());
  if (nbins > 0) CHIP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(int64_t) * nbins, hipMemcpyDeviceToHost, c->stream));
  CHIP_TRY(hipMemcpyAsync(sum_min_max, d_s, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
  CHIP_TRY(hipStreamSynchronize(c->stream));
  return DRS_OK;
}

// All ranks have arrived and their device is idle: an all-reduce of one word, then a device
// sync (bench.py brackets its timed region with it).
int32_t drs_comm_barrier(drs_comm c) {
  if (!c) return cfail(DRS_ERR_BAD_ARG, "drs_comm_barrier", "null communicator");
  double v[4] = {1.0, 0.0, 0.0, 0.0};
  int32_t rc = drs_stats_allreduce(c, nullptr, 0, v);
  if (rc) return rc;
  if (v[0] != (double)c->world) return cfail(DRS_ERR_HIP, "drs_comm_barrier", "rank count mismatch");
  CHIP_TRY(hipDeviceSynchronize());
  return DRS_OK;
}

}  // extern "C"

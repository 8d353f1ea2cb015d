// Host side of libdrs_hip.so: the C ABI of include/drs.h on top of the kernels in
// sls.hip / mlp.hip / gemm.hip.  One engine = one GPU = one process (accelInferenceEngine
// counterpart, reference accelInferenceEngine.py:18-86).
//
// HBM layout (all hipMalloc'ed once in drs_create / first use):
//   tables   one arena, table t at a 256-B aligned offset, rows*D fp32 row-major
//   weights  one arena: all biases back to back (layer order, padded to 4 floats), then per
//            layer W [N, K] dense row-major (as fed by the reference)
//   batches  per staged batch: dense [max_batch, m_den] f32 | idx [T, cap] i32 |
//            off [T, max_batch+1] i32 (exclusive prefix sums of the lengths)
//   slots    per in-flight launch set (up to DRS_MAX_COALESCE coalesced queries): interaction buffer(s),
//            layer scratch, device output buffer, [flag | err | out] in host-mapped pinned
//            memory, and a host-mapped pinned input block for per-call inputs
// Streams: every gather on stream_g, the rest of each launch set on a second stream behind
// an event; completion is a flag in pinned memory, not a stream sync.
#include <immintrin.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "drs_internal.h"

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

using namespace drs;

// Per-device one-time setup: HIP function attributes and allocations belong to a
// device, not to the process -- an engine on GPU 1 created after one on GPU 0 needs its own
// > 64 KB LDS opt-in and its own zero page.  Thread-safe; the table is indexed by device id.
namespace drs {
hipError_t device_init(int device, const float** zero_page) {
  constexpr int kMaxDevices = 64;
  static std::mutex mu;
  static bool done[kMaxDevices];
  static float* zero[kMaxDevices];
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!done[device]) {
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = mlp_set_attrs();
    if (e == hipSuccess) e = gemm_set_attrs();
    if (e == hipSuccess && !zero[device]) {
      e = hipMalloc(reinterpret_cast<void**>(&zero[device]), 256);
      if (e == hipSuccess) e = hipMemset(zero[device], 0, 256);
    }
    if (e != hipSuccess) return e;
    done[device] = true;
  }
  *zero_page = zero[device];
  return hipSuccess;
}

// "name<...>[grid] " appended to the slot's dispatch log (drs_last_dispatch); a full log drops what does not fit
void log_launch(DispatchLog* log, const char* fmt, ...) {
  if (!log) return;
  const int room = (int)sizeof(log->text) - log->len;
  if (room <= 2) return;
  if (log->len > 0) { log->text[log->len++] = ' '; log->text[log->len] = 0; }
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(log->text + log->len, (size_t)(sizeof(log->text) - log->len), fmt, ap);
  va_end(ap);
  if (n > 0) log->len = log->len + n < (int)sizeof(log->text) ? log->len + n : (int)sizeof(log->text) - 1;
}
}  // namespace drs

namespace {

thread_local std::string g_create_error;

// pinned host block of a slot: [flag | err | pad | pad | outputs...]: outputs 16-B aligned
constexpr int kOutOffset = 4;

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct Layer {
  float* W = nullptr;
  float* b = nullptr;
  int32_t m = 0, n = 0;  // W is [m, n]
  bool set = false;
  bool packed = false;   // a packed twin (MFMA operand order, mlp.hip) follows W in the arena
};

struct Mlp {
  std::vector<int32_t> ln;
  std::vector<Layer> layers;  // ln.size()-1
  int32_t sigmoid_layer = -1; // 1-based, -1 none
};

struct Batch {
  float* dense = nullptr;
  int32_t* idx = nullptr;
  int32_t* off = nullptr;
  int32_t n_samples = 0;
  int32_t uniform_len = -1;    // all bags of all tables have this length, else -1
  std::vector<int32_t> h_off;  // [T][max_batch+1] host copy (gather_bytes, validation)
  bool staged = false;
};

struct Slot {
  bool split_last = false;        // the set in flight read its dense columns in place ("gemm_split"): s.T holds none
  hipStream_t stream = nullptr;   // the stream the job in flight launches its MLP side on
  hipStream_t base_stream = nullptr;   // ... as assigned by apply_stream_mode (shared or own)
  hipStream_t own_stream = nullptr;
  hipStream_t early_stream = nullptr;    // "mlp_early": the MLP launch of a small set whose gather runs on own_stream
  uint32_t* d_gflag = nullptr;           // ... and the word that launch polls: seq, written by a stream-ordered write behind the gather
  hipStream_t gather_stream = nullptr;   // where the gather is launched (== stream unless pipelined)
  hipStream_t cur = nullptr;             // "mlp_layout" 1: the stream the set's latest MLP launch went on
  hipEvent_t ev_k[4] = {nullptr, nullptr, nullptr, nullptr};   // ... events that order a set's launches across the two kinds of stream
  int n_ev = 0;
  hipEvent_t ev_sls = nullptr;           // pipelined mode: gather done -> the MLP stream may go on
  hipEvent_t ev_in = nullptr;            // pipelined mode: per-call inputs copied -> the gather may start
  Batch zc;                              // per-call inputs read in place from host-mapped pinned memory
  float* T = nullptr;        // [max_batch, ldT]  concat buffer: dense_out | emb_0 | ...
  float* R = nullptr;        // [max_batch, ldR]  dot-interaction output (dot only)
  float* H = nullptr;        // [max_batch, ldH]  inter-segment MLP scratch (ping)
  float* Hb = nullptr;       // (pong)
  float* H2 = nullptr;       // NCF: concat(mf, mlp_out)
  float* H3 = nullptr;       // MT-WnD: output of the shared top MLP (input of every task head)
  float* d_out = nullptr;    // [max_batch*n_out] device outputs (copy path only)
  uint32_t* d_err = nullptr; // device error word (bit0: index out of range)
  uint32_t* d_counter = nullptr;  // arrival counter of the completion hand-off
  float* xbuf = nullptr;          // stream4_kernel's column-split form: exchange buffer [xrows, xcols] of the split layer's outputs ...
  uint32_t* xcnt = nullptr;       // ... and one arrival ticket per 16-row slab (zero between launches)
  int64_t xrows = 0;
  int32_t xcols = 0;
  uint32_t* h_out = nullptr; // pinned host: [flag | err | outputs...]
  uint32_t* dm_out = nullptr;// the same memory as seen from the device (zero-copy path)
  uint32_t seq = 0;          // sequence number of the query in flight on this slot
  uint64_t* d_ts = nullptr;  // [2 * max gather workgroups] device clock stamps (profiling)
  std::vector<uint64_t> h_ts;
  int64_t ts_blocks = 0, ts_blocks_done = 0;
  uint64_t* d_span_acc = nullptr;  // device [2]: running (min, max) of the stamps
  uint64_t* h_span = nullptr;  // pinned [2]: (min start, max end) of the gather launch
  uint64_t* dm_span = nullptr;
  Batch scratch;             // drs_forward_inputs staging
  char* d_stage = nullptr;   // device copy of the staging block (one-DMA-copy input path)
  Batch dc;                  // ... viewed as a batch: [dense | idx | off]
  void* h_stage = nullptr;   // pinned host staging for forward_inputs
  size_t h_stage_bytes = 0;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool ev_pending = false;
  int64_t ts_bytes = 0;      // algorithmic bytes of the gather launch being timed
  int32_t last_bs = 0;       // total valid samples of the job in flight
  int32_t last_n = 0;        // queries coalesced into it
  int32_t q_bs[DRS_MAX_COALESCE] = {0};
  int32_t q_vstart[DRS_MAX_COALESCE] = {0};
  bool busy = false;
  bool polled = false;       // completion arrives through the host flag
  // per-call inputs of a whole launch set (drs_run_queues_multi_async; allocated on first use):
  // DRS_MAX_COALESCE blocks [dense | idx | off] back to back in ONE pinned allocation and their
  // twins in ONE HBM allocation, so that a set's inputs cross the bus in one DMA copy
  char* h_multi = nullptr;
  char* d_multi = nullptr;
  size_t multi_block = 0;    // bytes from one block to the next
  std::vector<Batch> mq;     // block i viewed as a batch (device pointers into d_multi)
  int32_t launch_rc = 0;     // status of the launches the launcher thread made for the job in flight
  std::string launch_err;
  DispatchLog dlog = {{0}, 0};   // what the launch functions chose for the set last enqueued here (drs_last_dispatch)
};

// One copy of the table arena.  kind 0: a plain hipMalloc.  kind 1: built with the virtual-memory API --
// a reserved address range of chosen alignment, physical memory created in chunks of a chosen size
// (0: one handle for the whole arena) and mapped into it ("table_alloc" and friends, DESIGN.md 5:
// what a table is, models/dlrm_s_caffe2.py:297-299, does not say where it lives).
struct VaRange { void* base = nullptr; size_t reserved = 0; float* p = nullptr; };   // a reserved address range and the (aligned) arena address inside it
struct Arena {
  float* p = nullptr;
  int kind = 0;
  size_t va_bytes = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;   // kind 1: the physical memory, in chunks ...
  std::vector<size_t> place;                              // ... handle i sits at chunk position place[i] of the range
  std::vector<hipMemGenericAllocationHandle_t> pads;      // "table_va_perturb": 4 KiB allocations made between address candidates
  std::vector<VaRange> vas;                               // address ranges reserved for it ("table_va_next"); [va_cur] is mapped
  int va_cur = 0;
  size_t align = 0;
};

}  // namespace

namespace { class HostPool; class Launcher; }

struct drs_engine {
  int device = 0;
  int32_t kind = 0, T = 0, D = 0;
  std::vector<int64_t> rows;
  std::vector<int64_t> tab_off;  // element offsets
  float* tables = nullptr;
  // "table_placement": further copies of the arena in other places of HBM; `tables` is the one in use (see drs_set_option)
  std::vector<Arena> arenas;
  std::vector<hipMemGenericAllocationHandle_t> spacers;   // "table_spacer": device memory taken (never mapped) between placement candidates
  size_t tables_bytes = 0;
  // how the NEXT arena is built (drs_create's first one, "table_placement" -1 candidates)
  int table_alloc = 0;              // 0 hipMalloc | 1 virtual-memory API
  int64_t vmm_chunk = -1;           // bytes of physical memory per handle (0: one handle | -1: 1 GiB handles from 1 GiB on, else one); rounded up to whole 2 MiB pages
  int64_t vmm_align = 0;            // alignment of the reserved address range (0: the allocation granularity)
  // arena_alloc_selected: scratch of the gather probe and what the last selection saw
  int32_t* probe_idx = nullptr;
  float* probe_out = nullptr;
  int64_t* probe_tab = nullptr;
  int32_t* probe_err = nullptr;
  int64_t probe_rows = 0;
  int32_t probe_bags = 0, probe_L = 0;
  int64_t sel_want_pool = 0;        // "table_select_pool": chunks the next selection allocates (0: 2 n + 8)
  int64_t sel_pool = 0, sel_kept = 0, sel_best_ns = 0, sel_worst_ns = 0, sel_kept_worst_ns = 0, sel_ms = 0;
  int64_t probe_gather_ns = 0;      // result of the last "table_probe_gather"
  int64_t probe_mbs = 0;            // result of the last "table_probe"
  int probe_windows = 0, probe_sorted = 0, probe_row_bytes = 256, probe_nt = 1, probe_loads = 20;
  int64_t probe_ps = 0;             // result of the last "table_probe_latency": picoseconds per dependent load
  int vmm_shuffle = 0;              // lab: map the chunks in a permuted order (neighbouring addresses, distant memory)
  int64_t* d_tab_off = nullptr;
  int64_t* d_tab_rows = nullptr;
  std::vector<bool> table_set;
  Mlp bot, top, fin;
  std::vector<Mlp> tasks;        // MT-WnD task heads
  std::vector<Mlp> att;          // DIN attention units (one small MLP per behaviour table)
  const float** d_att = nullptr; // device: 4 pointers per unit (W1, b1, W2, b2) ...
  float* d_att_packed = nullptr; // ... and the units' weights packed for the DIN kernels (din.hip)
  bool att_dirty = true;         // a unit's weights changed since the last pack
  int dien_fuse_top = 1;         // DIEN: the top MLP inside the recurrence's launch when it fits (din.hip dien_top_fusable)
  int dien_mfma = 2;             // DIEN recurrence on the matrix cores, 16 samples per workgroup: 2 = one wave set per layer | 1 = every wave both layers | 0 one wave per sample (VALU)
  int din_fused = 1;             // gather + attention units + Concat in one launch (default mode)
  std::vector<Mlp> rnn;          // DIEN: the two BasicRNN layers, each {i2h, gates_t}; packed into d_att_packed
  float* w_arena = nullptr;      // all FC weights + biases in ONE allocation (large pages: the
  size_t w_arena_floats = 0;     // MLP kernels' per-CU TLBs then hold every weight page)
  size_t w_arena_used = 0;
  int32_t interaction_op = DRS_INTERACT_CAT, itself = 0;
  int32_t max_batch = 0, max_lookups = 0, n_batches = 0, n_slots = 0;
  int32_t m_den = 0, w0 = 0;     // dense input width, dense_out width
  int32_t num_int = 0, n_out = 0;
  int64_t ldT = 0, ldR = 0, ldH = 0, cap = 0;
  int64_t max_rows = 0;          // virtual rows of a slot's activation buffers
  std::vector<Batch> batches;
  std::vector<Slot> slots;
  // op-level scratch
  int64_t* d_op_tab = nullptr;   // [2]: tab_off, tab_rows for drs_sls
  // options
  int sls_exact = 0, mlp_split = 1, zero_copy = 1, sls_uniform = 1, shared_stream = 2, mlp_fuse = 1;
  int dispatch_log = 0;             // "dispatch_log": keep the per-slot record of the kernel forms chosen (drs_last_dispatch)
  hipStream_t stream_g = nullptr;   // shared_stream == 2: all gathers, back to back
#ifdef DRS_LAB
  hipStream_t stream_g2 = nullptr;  // lab ("gather_streams" 2): the gathers of consecutive slots alternate between two streams
  int gather_streams = 1;
#endif
  hipStream_t stream_h2d = nullptr; // input copies of drs_run_queues_multi_async (created on first use)
  int mlp_streams = 1;              // pipelined mode: streams the MLP launches alternate between (set in drs_create)
  // "mlp_layout" 1 (MLP-bound models, pipelined mode): streams by KERNEL TYPE instead of by launch set -- the
  // gather and every wide-layer GEMM of every set go on stream_g, strictly one after the other (each fills
  // the chip by itself: the gather then has the HBM to itself instead of sharing every CU with two
  // overlapping GEMM launches), the latency-bound chain launches go on the slots' MLP streams beside them;
  // an event per change of stream orders a set's launches.  0: a set's MLP launches all on its own stream.
  int mlp_layout = 0;
  int gather_bound = 0;             // set by choose_launch_forms (read only for callers)
  int gemm_split = 1;               // W&D / MT-WnD: the first top layer reads the dense columns from the queries' arrays (no copy launch)
  int gather_priority = 0;
  int mlp_cu_mask = 0, gather_cu_complement = 1;   // "mlp_cu_mask": CUs reserved for the MLP streams (0: none)
  int sls_short_bag = 8;            // uniform bag length up to which the lane-group-per-bag gather is used (drs_create: 2048 / D)
  // launch sets whose outputs are at least this many bytes (0: never) leave by a copy-engine transfer queued behind the
  // last kernel + a stream-ordered flag write, instead of the last workgroup's in-kernel copy: MT-WnD's 2 MB per
  // 16-query set (72.7 k -> 75.2 k queries/s, what leaving the copy out altogether gives); NCF's 1 MB sets lose with it
  // (two more HIP calls per 30-us set: 472 k -> 398 k at six sets in flight), hence the threshold
  int64_t out_dma = 1536 * 1024;
  int zero_copy_inputs = 1;         // drs_forward_inputs: 0 per-array copies | 1 read in place over PCIe | 2 one DMA copy | 3 by size
  int64_t mlp_wide_kn = 256 * 1024;   // K*N from which a layer gets its own 2-D launch (RM3's 1024x256 included)
  int64_t mlp_fuse_rows = 0;          // fuse bottom+top only from this many rows on
  int64_t mlp_small_rows = 1024;      // launch sets up to this many rows: MLP side on the slot's own stream
  int mlp_early = 0;                  // small sets of staged DLRM queries: the fused MLP launch starts beside the gather (Done::wait_flag)
  int small_piped = 0;                // ... and their gather: 0 = on the slot's own stream too, 1 = on the shared gather stream
  // profiling
  int profiling = 0;             // 0 off | 1 device clock stamps | 2 stamps + HIP events
  double k_ms[DRS_KERNEL_COUNT] = {0, 0, 0};
  int64_t k_n[DRS_KERNEL_COUNT] = {0, 0, 0};
  int64_t k_bytes[DRS_KERNEL_COUNT] = {0, 0, 0};   // algorithmic bytes of exactly the launches in k_ms / k_n
  double wall_clock_khz = 100000.0;
  Tune tune;                     // per-engine tunables + this device's zero page
  std::unique_ptr<HostPool> pool;   // workers of the per-call input pass (created on first use)
  // Per-call inputs, "launch_thread" 1: the calling thread converts a query's arrays (they are consumed
  // before the call returns, as the ABI promises) and hands everything that is a HIP call -- the DMA
  // copy, the events, the launches -- to this thread: the caller's time per call drops from 20 to
  // 14 us.  Off by default: the path's throughput does not move (36 k queries/s either way, round 3:
  // it is bound by how fast 0.76 MB per query crosses PCIe in sub-megabyte pieces).
  std::unique_ptr<Launcher> launcher;
  std::unique_ptr<std::atomic<int>[]> launch_state;   // per slot: 0 idle | 1 handed over, not launched yet | 2 launched
  int launch_thread = 0;
  std::mutex err_mu;             // e->err is written by both threads
  int host_threads = -1;         // "host_threads": workers beside the caller (-1 = auto: min(T, 7))
  std::string err;
};

namespace {

int32_t fail(drs_engine* e, int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) { std::lock_guard<std::mutex> l(e->err_mu); e->err = buf; } else g_create_error = buf;
  return code;
}

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _r = (call);                                                                \
    if (_r != hipSuccess)                                                                  \
      return fail((e), _r == hipErrorOutOfMemory ? DRS_ERR_OOM : DRS_ERR_HIP, "%s: %s",    \
                  #call, hipGetErrorString(_r));                                           \
  } while (0)

int32_t set_device(drs_engine* e) {
  HIP_TRY(e, hipSetDevice(e->device));
  return DRS_OK;
}

int32_t alloc_batch(drs_engine* e, Batch& b) {
  if (e->m_den > 0)
    HIP_TRY(e, hipMalloc(&b.dense, sizeof(float) * (size_t)e->max_batch * e->m_den));
  HIP_TRY(e, hipMalloc(&b.idx, sizeof(int32_t) * (size_t)e->T * e->cap));
  HIP_TRY(e, hipMalloc(&b.off, sizeof(int32_t) * (size_t)e->T * (e->max_batch + 1)));
  b.h_off.assign((size_t)e->T * (e->max_batch + 1), 0);
  return DRS_OK;
}

void free_batch(Batch& b) {
  if (b.dense) (void)hipFree(b.dense);
  if (b.idx) (void)hipFree(b.idx);
  if (b.off) (void)hipFree(b.off);
  b = Batch();
}

// ---- table arenas ---------------------------------------------------------------------------------
hipError_t va_reserve(size_t total, size_t align, VaRange* out) {
  // hipMemAddressReserve does not honour its alignment argument beyond the granularity (measured: "1 GiB aligned"
  // came back 2 MiB aligned): reserve `align` bytes more and align inside the range by hand
  void* base = nullptr;
  const size_t reserved = total + align;
  hipError_t r = hipMemAddressReserve(&base, reserved, 0, nullptr, 0);
  if (r != hipSuccess) return r;
  out->base = base;
  out->reserved = reserved;
  out->p = static_cast<float*>(align ? reinterpret_cast<void*>(((uintptr_t)base + align - 1) / align * align) : base);
  return hipSuccess;
}

// the arena's physical handles at address `at` (handle i at place[i]); on failure nothing stays mapped there
hipError_t arena_map(const Arena& a, float* at, int device) {
  const size_t n = a.handles.size(), csz = a.va_bytes / n;
  hipError_t r = hipSuccess;
  std::vector<size_t> done;
  for (size_t i = 0; i < n && r == hipSuccess; ++i) {
    r = hipMemMap(reinterpret_cast<char*>(at) + a.place[i] * csz, csz, 0, a.handles[i], 0);
    if (r == hipSuccess) done.push_back(a.place[i]);
  }
  if (r == hipSuccess) {
    hipMemAccessDesc desc;
    memset(&desc, 0, sizeof desc);
    desc.location.type = hipMemLocationTypeDevice;
    desc.location.id = device;
    desc.flags = hipMemAccessFlagsProtReadWrite;
    r = hipMemSetAccess(at, a.va_bytes, &desc, 1);
  }
  if (r != hipSuccess) {
    for (size_t pl : done) (void)hipMemUnmap(reinterpret_cast<char*>(at) + pl * csz, csz);
    (void)hipGetLastError();
  }
  return r;
}

void arena_free(Arena& a) {
  if (!a.p) { a = Arena(); return; }
  if (a.kind == 0) {
    (void)hipFree(a.p);
  } else {
    (void)hipMemUnmap(a.p, a.va_bytes);
    for (auto& h : a.handles) (void)hipMemRelease(h);
    for (auto& h : a.pads) (void)hipMemRelease(h);
    for (auto& v : a.vas) (void)hipMemAddressFree(v.base, v.reserved);
  }
  a = Arena();
}

#ifdef DRS_LAB
// The model's OWN gather kernel on a one-table problem laid over [base, base + bytes): `bags` bags of L sorted,
// distinct rows each (one row per L-th of the range, what np.unique leaves of a bag's draws,
// data_generator/dlrm_data_caffe2.py:105-110), the launch the engine would make for them.  *us = average
// duration of a launch.  Why the real kernel: synthetic row-read probes that saturate the memory system read every
// gigabyte of HBM equally fast; the gather kernels, with a handful of loads in flight per lane, do not
// (profiles/r05_placement/).
hipError_t probe_gather(drs_engine* e, const float* base, size_t bytes, double* us) {
  *us = 0;
  const int D = e->D;
  const int64_t rows = (int64_t)(bytes / ((size_t)D * 4));
  const int L = e->max_lookups > 256 ? 256 : e->max_lookups;
  const int bags = 16384;
  if (rows < L || rows >= (1ll << 31)) return hipErrorInvalidValue;
  hipError_t r = hipSuccess;
  if (!e->probe_idx || e->probe_rows != rows || e->probe_L != L) {
    if (e->probe_idx) { (void)hipFree(e->probe_idx); e->probe_idx = nullptr; }
    std::vector<int32_t> idx((size_t)bags * L);
    uint32_t z = 0x2545F491u;
    for (int b = 0; b < bags; ++b)
      for (int j = 0; j < L; ++j) {
        z ^= z << 13; z ^= z >> 17; z ^= z << 5;
        const int64_t lo = rows * j / L, hi = rows * (j + 1) / L;
        idx[(size_t)b * L + j] = (int32_t)(lo + (int64_t)(z % (uint32_t)(hi > lo ? hi - lo : 1)));
      }
    if ((r = hipMalloc(&e->probe_idx, sizeof(int32_t) * idx.size())) != hipSuccess) return r;
    if ((r = hipMemcpy(e->probe_idx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice)) != hipSuccess) return r;
    if (!e->probe_out && (r = hipMalloc(&e->probe_out, sizeof(float) * (size_t)bags * D)) != hipSuccess) return r;
    if (!e->probe_tab && (r = hipMalloc(&e->probe_tab, sizeof(int64_t) * 2)) != hipSuccess) return r;
    if (!e->probe_err && (r = hipMalloc(&e->probe_err, sizeof(int32_t))) != hipSuccess) return r;
    const int64_t tab[2] = {0, rows};
    if ((r = hipMemcpy(e->probe_tab, tab, sizeof tab, hipMemcpyHostToDevice)) != hipSuccess) return r;
    if ((r = hipMemset(e->probe_err, 0, sizeof(int32_t))) != hipSuccess) return r;
    e->probe_rows = rows; e->probe_L = L; e->probe_bags = bags;
  }
  SlsArgs a;
  memset(&a, 0, sizeof a);
  a.tables = base; a.tab_off = e->probe_tab; a.tab_rows = e->probe_tab + 1;
  a.q.n_q = 1; a.q.vstart[1] = bags; a.q.cum[1] = bags; a.q.bs[0] = bags;
  a.idx[0] = e->probe_idx; a.off[0] = nullptr; a.uniform_len[0] = L;
  a.idx_stride = (int64_t)bags * L; a.off_stride = bags + 1;
  a.out = e->probe_out; a.ld_out = D; a.col0 = 0; a.T = 1; a.D = D; a.err = e->probe_err; a.ts = nullptr;
  Tune t = e->tune;
  t.log = nullptr;
  const int exact = L <= e->sls_short_bag && !sls_flat_applicable(a, t);
  hipEvent_t e0, e1;
  if ((r = hipEventCreate(&e0)) != hipSuccess) return r;
  if ((r = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return r; }
  const int warm = 2, reps = 6;
  for (int i = 0; i < warm + reps && r == hipSuccess; ++i) {
    if (i == warm) r = hipEventRecord(e0, nullptr);
    if (r == hipSuccess) r = launch_sls(a, exact, t, nullptr);
  }
  if (r == hipSuccess) r = hipEventRecord(e1, nullptr);
  if (r == hipSuccess) r = hipEventSynchronize(e1);
  float ms = 0.f;
  if (r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
  if (r == hipSuccess) *us = (double)ms * 1e3 / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return r;
}

// "table_alloc" 3: the arena out of the fastest gigabytes of a pool.  ceil(bytes / 1 GiB) = n chunks are needed;
// up to 2 n + 8 one-GiB handles are created (never more than half of the free memory), each is timed with
// probe_gather, the n fastest are mapped back to back as the arena and the rest is released at once: one copy of
// the tables, nothing held.  Costs ~1 ms per pool chunk at engine start.
hipError_t arena_alloc_selected(drs_engine* e, size_t bytes, Arena* out) {
  *out = Arena();
  const size_t chunk = (size_t)1 << 30;
  const size_t n = (bytes + chunk - 1) / chunk;
  const auto t_begin = std::chrono::steady_clock::now();
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  size_t free_b = 0, total_b = 0;
  hipError_t r = hipMemGetInfo(&free_b, &total_b);
  if (r != hipSuccess) return r;
  size_t m = 2 * n + 8;
  if (e->sel_want_pool > 0) m = (size_t)e->sel_want_pool;
  if (m > 192) m = 192;
  while (m > n && m * chunk > free_b / 2) --m;
  if (m < n) m = n;
  std::vector<hipMemGenericAllocationHandle_t> pool;
  for (size_t i = 0; i < m; ++i) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    pool.push_back(h);
  }
  auto release_all = [&]() { for (auto& h : pool) (void)hipMemRelease(h); };
  if (pool.size() < n) { release_all(); return hipErrorOutOfMemory; }
  m = pool.size();
  std::vector<double> us(m, 0.0);
  if (m > n) {
    Arena all;
    all.kind = 1; all.va_bytes = m * chunk; all.handles = pool;
    for (size_t i = 0; i < m; ++i) all.place.push_back(i);
    VaRange v;
    r = va_reserve(all.va_bytes, 0, &v);
    if (r == hipSuccess) {
      r = arena_map(all, v.p, e->device);
      if (r == hipSuccess) {
        // two rounds, the faster reading of each chunk counts (a single reading can catch a clock ramp)
        for (int round = 0; round < 2 && r == hipSuccess; ++round)
          for (size_t i = 0; i < m && r == hipSuccess; ++i) {
            double t = 0;
            r = probe_gather(e, reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.p) + i * chunk), chunk, &t);
            if (r == hipSuccess && (round == 0 || t < us[i])) us[i] = t;
          }
        (void)hipMemUnmap(v.p, all.va_bytes);
      }
      (void)hipMemAddressFree(v.base, v.reserved);
    }
    if (r != hipSuccess) { release_all(); (void)hipGetLastError(); return r; }
  }
  std::vector<size_t> order(m);
  for (size_t i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return us[x] < us[y]; });
  std::vector<size_t> keep(order.begin(), order.begin() + (long)n);
  std::sort(keep.begin(), keep.end());
  Arena a;
  a.kind = 1;
  a.va_bytes = n * chunk;
  a.align = e->vmm_align > 0 ? (size_t)e->vmm_align : 0;
  std::vector<bool> kept(m, false);
  for (size_t k : keep) { a.handles.push_back(pool[k]); a.place.push_back(a.place.size()); kept[k] = true; }
  for (size_t i = 0; i < m; ++i) if (!kept[i]) (void)hipMemRelease(pool[i]);
  VaRange v;
  r = va_reserve(a.va_bytes, a.align, &v);
  if (r == hipSuccess) {
    r = arena_map(a, v.p, e->device);
    if (r != hipSuccess) (void)hipMemAddressFree(v.base, v.reserved);
  }
  if (r != hipSuccess) { for (auto& h : a.handles) (void)hipMemRelease(h); (void)hipGetLastError(); return r; }
  a.vas.push_back(v);
  a.p = v.p;
  e->sel_pool = (int64_t)m; e->sel_kept = (int64_t)n;
  e->sel_best_ns = (int64_t)(us[order[0]] * 1e3); e->sel_worst_ns = (int64_t)(us[order[m - 1]] * 1e3);
  e->sel_kept_worst_ns = (int64_t)(us[order[n - 1]] * 1e3);
  e->sel_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_begin).count();
  *out = a;
  return hipSuccess;
}

#endif  // DRS_LAB
// `bytes` of device memory for the tables, built as e->table_alloc / vmm_* say.  On failure nothing stays
// allocated and *out is empty.
hipError_t arena_alloc(drs_engine* e, size_t bytes, Arena* out) {
  *out = Arena();
#ifdef DRS_LAB
  if (e->table_alloc == 3) {
    if (arena_alloc_selected(e, bytes, out) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();           // (no room for a pool, no virtual-memory API ...: a plain allocation)
  }
#endif  // DRS_LAB
  if (e->table_alloc == 0 || e->table_alloc == 2 || e->table_alloc == 3) {
    // 2: physically contiguous device memory, best effort (hipDeviceMallocContiguous: the driver assembles the
    // allocation from neighbouring free blocks instead of taking whatever blocks head its free lists -- DESIGN.md 5)
    void* p = nullptr;
    hipError_t r = hipErrorOutOfMemory;
    if (e->table_alloc == 2) {
      r = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
      if (r != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    if (r != hipSuccess) r = hipMalloc(&p, bytes);
    if (r != hipSuccess) return r;
    out->p = static_cast<float*>(p);
    out->va_bytes = bytes;
    return hipSuccess;
  }
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  size_t gran = 0;
  hipError_t r = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (r != hipSuccess) return r;
  if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;      // (the runtime reports 4 KiB; whole 2 MiB pages keep every mapping a huge page)
  // "table_vmm_chunk" -1 (the default): 1 GiB handles for arenas of at least 1 GiB, one handle for smaller ones
  const int64_t want = e->vmm_chunk < 0 ? (bytes >= ((size_t)1 << 30) ? (int64_t)1 << 30 : 0) : e->vmm_chunk;
  const size_t chunk = want > 0 ? (size_t)round_up(want, (int64_t)gran) : 0;
  const size_t total = (size_t)round_up((int64_t)bytes, (int64_t)(chunk ? chunk : gran));
  Arena a;
  a.kind = 1;
  a.va_bytes = total;
  a.align = e->vmm_align > 0 ? (size_t)round_up(e->vmm_align, (int64_t)gran) : 0;
  const size_t n = chunk ? total / chunk : 1, csz = chunk ? chunk : total;
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    r = hipMemCreate(&h, csz, &prop, 0);
    if (r != hipSuccess) break;
    a.handles.push_back(h);
    // chunk i of physical memory goes to place perm(i) of the range ("table_vmm_shuffle", a lab option: a fixed odd-multiplier walk)
    size_t pl = i;
    if (e->vmm_shuffle && n > 2) pl = (n & (n - 1)) == 0 ? (i * ((n / 2) | 1) + n / 3) % n : n - 1 - i;
    a.place.push_back(pl);
  }
  VaRange v;
  if (r == hipSuccess) r = va_reserve(total, a.align, &v);
  if (r == hipSuccess) {
    r = arena_map(a, v.p, e->device);
    if (r != hipSuccess) (void)hipMemAddressFree(v.base, v.reserved);
  }
  if (r != hipSuccess) {
    for (auto& h : a.handles) (void)hipMemRelease(h);
    (void)hipGetLastError();
    return r;
  }
  a.vas.push_back(v);
  a.va_cur = 0;
  a.p = v.p;
  *out = a;
  return hipSuccess;
}

#ifdef DRS_LAB
// The arena in use moves to another address range: "table_va_next" reserves one more range and maps the arena's
// memory there (the ranges tried so far stay reserved -- address space only, no memory), "table_va_select" k goes
// back to candidate k and gives the other ranges up.  A lab instrument (tools/placement_lab.py): it showed that the
// gather's speed on an arena does NOT depend on the address -- one arena read the same at 24-60 address ranges, and
// fast / slow memory stayed fast / slow wherever it was mapped (profiles/r05_placement/README.md).
int32_t arena_move(drs_engine* e, Arena& a, int64_t to /* -1: a fresh range */) {
  if (a.kind != 1) return fail(e, DRS_ERR_STATE, "the table arena was not built with the virtual-memory API (\"table_alloc\" 1)");
  if (to >= (int64_t)a.vas.size()) return fail(e, DRS_ERR_BAD_ARG, "address candidate %lld of %zu", (long long)to, a.vas.size());
  if (to < 0) {
    if (a.vas.size() >= 64) return fail(e, DRS_ERR_BAD_ARG, "64 address candidates are the limit");
    VaRange v;
    hipError_t r = va_reserve(a.va_bytes, a.align, &v);
    if (r != hipSuccess) { (void)hipGetLastError(); return fail(e, DRS_ERR_OOM, "hipMemAddressReserve: %s", hipGetErrorString(r)); }
    a.vas.push_back(v);
    to = (int64_t)a.vas.size() - 1;
  }
  if (to == a.va_cur) return DRS_OK;
  HIP_TRY(e, hipMemUnmap(a.p, a.va_bytes));
  hipError_t r = arena_map(a, a.vas[(size_t)to].p, e->device);
  if (r != hipSuccess) {
    // back to where it was: the engine must not be left without its tables
    hipError_t r2 = arena_map(a, a.p, e->device);
    return fail(e, DRS_ERR_HIP, "mapping the tables at another address: %s%s", hipGetErrorString(r), r2 == hipSuccess ? "" : " (and the old mapping could not be restored)");
  }
  const bool in_use = e->tables == a.p;
  a.va_cur = (int)to;
  a.p = a.vas[(size_t)to].p;
  if (in_use) e->tables = a.p;
  return DRS_OK;
}

#endif  // DRS_LAB
// ---- host-side worker pool for the per-call input pass ----------------------------------------
// drs_forward_inputs converts 160 k indices per RMC1 query on the host; one thread doing that
// (plus the Python call) capped the PCIe-inclusive path at 12 k queries/s.  The
// tables of a query are independent, so they are spread over a few workers.  Workers spin for a
// short while after a job (the next query usually follows within microseconds) and then sleep on
// a condition variable; the calling thread always takes part, so a pool of zero workers is just
// the plain loop.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_.store(true, std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int workers() const { return (int)th_.size(); }
  // fn(i) for i in [0, n), on the caller and the workers; returns when all are done
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    // a worker of the PREVIOUS job may still be between its last pending_ decrement and its next
    // next_ increment inside work(): re-arming fn_ / n_ / next_ under it would be a data race, and a
    // worker that grabs an item before pending_ is stored would leave run() spinning forever
    // (ADVICE r2).  Wait until nobody is inside work(), then arm pending_ BEFORE next_.
    while (active_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    fn_ = &fn; n_.store(n, std::memory_order_relaxed);
    pending_.store(n, std::memory_order_relaxed);
    next_.store(0, std::memory_order_release);
    {
      std::lock_guard<std::mutex> l(mu_);     // pairs with the sleepers' predicate check
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    work();
    while (pending_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    next_.store(1 << 30, std::memory_order_release);   // closed: a worker that wakes up late finds no item
  }

 private:
  void work() {
    active_.fetch_add(1, std::memory_order_acq_rel);
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_acq_rel);
      if (i >= n_.load(std::memory_order_relaxed)) break;
      (*fn_)(i);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }
  void loop() {
    // gen_ is 0 when the constructor starts the workers: a worker that gets its first time slice only
    // after run() -- or the destructor -- has already bumped gen_ must still notice that bump (reading
    // gen_ here instead left such a worker asleep for ever and the destructor's join() with it: pools
    // that are created and destroyed without work in between, small models on a busy host)
    uint64_t seen = 0;
    for (;;) {
      // spin ~50 us for the next job, then sleep
      bool got = false;
      for (int spin = 0; spin < 20000; ++spin) {
        if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
        __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> l(mu_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> next_{1 << 30}, pending_{0}, sleepers_{0}, active_{0};   // (next_ past any n_ while idle)
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> n_{0};
  std::atomic<bool> stop_{false};
};

// int64 -> int32 with the range ENFORCE, branch-free so it vectorises (AVX2 where the host has
// it); returns the position of the first offending index or -1
template <int>
static inline int64_t narrow_checked_impl(const int64_t* __restrict__ src, int64_t n, int64_t rows,
                                          int32_t* __restrict__ dst) {
  uint64_t bad = 0;
  const uint64_t R = (uint64_t)rows;
  for (int64_t j = 0; j < n; ++j) {
    const uint64_t v = (uint64_t)src[j];       // negative -> huge: one unsigned compare
    bad |= (uint64_t)(v >= R);
    dst[j] = (int32_t)v;
  }
  if (!bad) return -1;
  for (int64_t j = 0; j < n; ++j)
    if ((uint64_t)src[j] >= R) return j;
  return -1;
}
// AVX2 form: 8 indices per step, packed into one 32-byte NON-TEMPORAL store -- the destination is a pinned
// block the DMA engine reads next, never this core: streaming stores skip the read-for-ownership of every
// destination line (a third of the pass's memory reads; 12-query sets on the GPU box's host: 43-57 k -> 59-64 k
// queries/s, the bus then carries 48 GB/s)
__attribute__((target("avx2"))) static int64_t narrow_checked_avx2(const int64_t* s, int64_t n, int64_t r, int32_t* d) {
  int64_t j = 0;
  uint64_t bad = 0;
  const uint64_t R = (uint64_t)r;
  for (; j < n && ((uintptr_t)(d + j) & 31); ++j) {
    const uint64_t v = (uint64_t)s[j];
    bad |= (uint64_t)(v >= R);
    d[j] = (int32_t)v;
  }
  const __m256i sign = _mm256_set1_epi64x((long long)0x8000000000000000ull);
  const __m256i lim = _mm256_set1_epi64x((long long)((R - 1) ^ 0x8000000000000000ull));   // v > R - 1, unsigned
  const __m256i pick = _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6);
  __m256i acc = _mm256_setzero_si256();
  if (R > 0)
    for (; j + 8 <= n; j += 8) {
      const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + j));
      const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + j + 4));
      acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_cmpgt_epi64(_mm256_xor_si256(a, sign), lim),
                                                 _mm256_cmpgt_epi64(_mm256_xor_si256(b, sign), lim)));
      const __m256i lo = _mm256_permutevar8x32_epi32(a, pick), hi = _mm256_permutevar8x32_epi32(b, pick);
      const __m256i o = _mm256_blend_epi32(lo, hi, 0xf0);
      _mm256_stream_si256(reinterpret_cast<__m256i*>(d + j), o);
    }
  bad |= (uint64_t)!_mm256_testz_si256(acc, acc);
  for (; j < n; ++j) {
    const uint64_t v = (uint64_t)s[j];
    bad |= (uint64_t)(v >= R);
    d[j] = (int32_t)v;
  }
  _mm_sfence();
  if (!bad) return -1;
  for (int64_t k = 0; k < n; ++k)
    if ((uint64_t)s[k] >= R) return k;
  return -1;
}
static int64_t narrow_checked_base(const int64_t* s, int64_t n, int64_t r, int32_t* d) {
  return narrow_checked_impl<0>(s, n, r, d);
}
static int64_t narrow_checked(const int64_t* s, int64_t n, int64_t r, int32_t* d) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  return avx2 ? narrow_checked_avx2(s, n, r, d) : narrow_checked_base(s, n, r, d);
}

// Validate (the Caffe2 ENFORCEs) and narrow int64 -> int32 (the Cast op,
// models/dlrm_s_caffe2.py:308-309) into caller-provided host buffers: one table of one query ...
struct ConvRes { int32_t code = DRS_OK; int32_t bag = 0; int64_t pos = 0, val = 0, total = 0; bool same = true; };
void convert_table(const drs_engine* e, int32_t n, int t, const int64_t* idx_t, int64_t n_idx_t,
                   const int32_t* len_t, int32_t* idx32_t /*[cap]*/, int32_t* off_t /*[max_batch+1]*/, ConvRes& r) {
  r = ConvRes();
  if (!idx_t && n_idx_t > 0) { r.code = DRS_ERR_BAD_ARG; r.pos = -1; return; }
  if (!len_t) { r.code = DRS_ERR_BAD_ARG; r.pos = -2; return; }
  if (n_idx_t < 0 || n_idx_t > e->cap) { r.code = DRS_ERR_BAD_ARG; r.pos = -3; return; }
  int64_t total = 0;
  off_t[0] = 0;
  const int32_t L0 = n > 0 ? len_t[0] : 0;
  bool same = true;
  for (int b = 0; b < n; ++b) {
    if (len_t[b] < 0) { r.code = DRS_ERR_LENGTHS_SUM; r.bag = b; r.pos = -1; return; }
    same = same && len_t[b] == L0;
    total += len_t[b];
    if (total > n_idx_t) break;
    off_t[b + 1] = (int32_t)total;
  }
  r.total = total;
  r.same = same;
  if (total != n_idx_t) { r.code = DRS_ERR_LENGTHS_SUM; r.pos = 0; return; }
  for (int b = n; b < e->max_batch; ++b) off_t[b + 1] = (int32_t)total;
  const int64_t j = narrow_checked(idx_t, n_idx_t, e->rows[t], idx32_t);
  if (j >= 0) { r.code = DRS_ERR_INDEX_RANGE; r.pos = j; r.val = idx_t[j]; }
}
// ... and what the lowest failing table of a query reports (what a sequential pass would have hit first)
int32_t convert_report(drs_engine* e, const ConvRes* res, const int64_t* n_idx, const char* who = "") {
  for (int t = 0; t < e->T; ++t) {
    const ConvRes& r = res[t];
    if (r.code == DRS_OK) continue;
    if (r.code == DRS_ERR_BAD_ARG) {
      if (r.pos == -1) return fail(e, DRS_ERR_BAD_ARG, "%sh_idx[%d] is NULL", who, t);
      if (r.pos == -2) return fail(e, DRS_ERR_BAD_ARG, "%sh_len[%d] is NULL", who, t);
      return fail(e, DRS_ERR_BAD_ARG, "%stable %d: %lld indices exceed staging capacity %lld", who, t,
                  (long long)n_idx[t], (long long)e->cap);
    }
    if (r.code == DRS_ERR_LENGTHS_SUM) {
      if (r.pos == -1) return fail(e, DRS_ERR_LENGTHS_SUM, "%stable %d bag %d: negative length", who, t, r.bag);
      return fail(e, DRS_ERR_LENGTHS_SUM, "%stable %d: sum(lengths)=%lld != len(indices)=%lld", who, t,
                  (long long)r.total, (long long)n_idx[t]);
    }
    return fail(e, DRS_ERR_INDEX_RANGE, "%stable %d: index %lld at position %lld outside [0, %lld)", who, t,
                (long long)r.val, (long long)r.pos, (long long)e->rows[t]);
  }
  return DRS_OK;
}

int32_t convert_inputs(drs_engine* e, int32_t n, const int64_t* const* h_idx, const int64_t* n_idx,
                       const int32_t* const* h_len, int32_t* idx32 /*[T][cap]*/,
                       int32_t* off /*[T][max_batch+1]*/, HostPool* pool = nullptr,
                       const std::function<void()>* also = nullptr /*one more independent work item*/) {
  std::vector<ConvRes> res((size_t)e->T);
  auto one = [&](int t) {
    convert_table(e, n, t, h_idx[t], n_idx[t], h_len[t], idx32 + (size_t)t * e->cap, off + (size_t)t * (e->max_batch + 1), res[t]);
  };
  int64_t work = 0;
  for (int t = 0; t < e->T; ++t) work += n_idx[t] > 0 ? n_idx[t] : 0;
  auto item = [&](int i) { if (i < e->T) one(i); else (*also)(); };
  const int n_items = e->T + (also ? 1 : 0);
  if (pool && work >= 32768) pool->run(n_items, item);
  else for (int i = 0; i < n_items; ++i) item(i);
  return convert_report(e, res.data(), n_idx);
}

int32_t mlp_ready(drs_engine* e, const Mlp& m, const char* name) {
  for (size_t i = 0; i < m.layers.size(); ++i)
    if (!m.layers[i].set) return fail(e, DRS_ERR_STATE, "%s layer %zu has no weights", name, i);
  return DRS_OK;
}

int act_of(const Mlp& m, int l) { return (l + 1 == m.sigmoid_layer) ? DRS_ACT_SIGMOID : DRS_ACT_RELU; }

// A layer big enough to deserve its own 2-D launch (many workgroups, W streamed once per
// 16-row slab would be too much traffic): RM3's 2560x1024.  RM1's 576x256 is not.
bool is_wide(const drs_engine* e, const Mlp& m, int l) {
  return e->mlp_split && (int64_t)m.ln[l] * m.ln[l + 1] >= e->mlp_wide_kn;
}

void fill_chain(ChainArgs& c, const Mlp& m, int l0, int cnt, const float* x, int64_t ldx, int64_t M,
                float* y, int64_t ldy) {
  memset(&c, 0, sizeof c);
  c.x = x; c.ldx = ldx; c.M = M; c.n_layers = cnt; c.y = y; c.ldy = ldy;
  for (int i = 0; i <= cnt; ++i) c.width[i] = m.ln[l0 + i];
  for (int i = 0; i < cnt; ++i) {
    c.W[i] = m.layers[l0 + i].W;
    c.b[i] = m.layers[l0 + i].b;
    c.act[i] = act_of(m, l0 + i);
  }
}

constexpr size_t kChainLds = 156 * 1024;

// The stream the set's next MLP launch goes on.  "mlp_layout" 0: the set's own MLP stream.  1: wide-layer
// GEMMs of full launch sets on the gather stream, everything else on the set's MLP stream; when the kind
// changes inside a set, the new stream waits for an event recorded behind the set's previous launch.
hipError_t mlp_launch_stream(drs_engine* e, Slot& s, bool wide, int64_t M, hipStream_t* out) {
  hipStream_t want = s.stream;
  if (e->mlp_layout == 1 && e->shared_stream == 2 && wide && M > e->mlp_small_rows) want = e->stream_g;
  if (s.cur && s.cur != want) {
    hipEvent_t ev = s.ev_k[s.n_ev];
    s.n_ev = (s.n_ev + 1) & 3;
    hipError_t r = hipEventRecord(ev, s.cur);
    if (r == hipSuccess) r = hipStreamWaitEvent(want, ev, 0);
    if (r != hipSuccess) return r;
  }
  s.cur = want;
  *out = want;
  return hipSuccess;
}

// "mlp_layout" 1: a launch or copy that goes straight on s.stream (interaction, row copies, DIN attention, the
// output copy and flag write, the timing event) must sit behind the set's latest MLP launch, which may have gone
// on the gather's stream (a wide layer): bring the set back to s.stream first.  A no-op otherwise.
hipError_t rejoin_stream(drs_engine* e, Slot& s) {
  hipStream_t st;
  return mlp_launch_stream(e, s, false, 0, &st);
}

// Run all layers of `m` on x -> y.  A huge layer runs as its own 2-D launch; runs of
// ordinary layers are fused into one LDS-resident chain.  Segment outputs that are not
// the final one ping-pong between s.H and s.Hb.
int32_t run_mlp(drs_engine* e, Slot& s, const Mlp& m, const float* x, int64_t ldx, int64_t M,
                float* y, int64_t ldy, const Done* done = nullptr, const XSrc* xs = nullptr) {
  const int n_layers = (int)m.layers.size();
  int l0 = 0;
  const float* in = x;
  int64_t ldin = ldx;
  while (l0 < n_layers) {
    int cnt = 1;
    ChainArgs c;
    memset(&c, 0, sizeof c);
    bool standalone = is_wide(e, m, l0);
    if (!standalone) {
      cnt = 0;
      while (l0 + cnt < n_layers && cnt < DRS_MAX_CHAIN && !is_wide(e, m, l0 + cnt)) ++cnt;
      for (;;) {
        fill_chain(c, m, l0, cnt, in, ldin, M, nullptr, 0);
        if (chain_lds_bytes(c, e->tune) <= kChainLds) break;
        if (cnt == 1) { standalone = true; break; }
        --cnt;
      }
    }
    const bool last = l0 + cnt == n_layers;
    float* out = last ? y : (in == s.H ? s.Hb : s.H);
    const int64_t ldo = last ? ldy : e->ldH;
    hipStream_t st = s.stream;
    HIP_TRY(e, mlp_launch_stream(e, s, standalone && is_wide(e, m, l0), M, &st));
    if (standalone) {
      HIP_TRY(e, launch_fc(in, ldin, M, m.ln[l0], m.layers[l0].W, m.layers[l0].b, m.ln[l0 + 1],
                           act_of(m, l0), out, ldo, e->tune, st, last ? done : nullptr,
                           l0 == 0 ? xs : nullptr));
    } else {
      c.y = out; c.ldy = ldo;
      HIP_TRY(e, launch_chain(c, e->tune, st, last ? done : nullptr, l0 == 0 ? xs : nullptr));
    }
    in = out; ldin = ldo; l0 += cnt;
  }
  return DRS_OK;
}

// DLRM: bottom MLP, interaction and top MLP of a 16-row slab in ONE launch (the slab's
// dense_out never waits for a kernel boundary).  "cat": the top chain reads the buffer the
// bottom chain wrote; "dot": the stream kernel computes T.T^T in LDS between the chains.
struct FusedPlan {
  bool ok = false;
  ChainArgs a, b;
  DotArgs dot;
  bool has_dot = false;
};

FusedPlan fused_plan(const drs_engine* e, const Slot& s, int64_t Mv, float* out, const XSrc* xs, bool* can_defer = nullptr) {
  if (can_defer) *can_defer = false;
  FusedPlan p;
  if (!e->mlp_fuse || Mv < e->mlp_fuse_rows || e->kind != DRS_MODEL_DLRM) return p;
  const int nb = (int)e->bot.layers.size(), nt = (int)e->top.layers.size();
  if (nb < 1 || nt < 1 || nb > DRS_MAX_CHAIN || nt > DRS_MAX_CHAIN) return p;
  for (int l = 0; l < nb; ++l) if (is_wide(e, e->bot, l)) return p;
  for (int l = 0; l < nt; ++l) if (is_wide(e, e->top, l)) return p;
  fill_chain(p.a, e->bot, 0, nb, nullptr, e->m_den, Mv, s.T, e->ldT);
  if (e->interaction_op == DRS_INTERACT_CAT) {
    fill_chain(p.b, e->top, 0, nt, s.T, e->ldT, Mv, out, e->n_out);
    p.ok = can_defer ? (stream_applicable(p.a, p.b, e->tune, xs, nullptr, nullptr, can_defer) || chain2_lds_bytes(p.a, p.b, e->tune) <= kChainLds)
                     : (chain2_lds_bytes(p.a, p.b, e->tune) <= kChainLds || stream_applicable(p.a, p.b, e->tune, xs, nullptr));
  } else {
    fill_chain(p.b, e->top, 0, nt, s.R, e->ldR, Mv, out, e->n_out);
    p.dot.T = s.T; p.dot.ldt = e->ldT; p.dot.F = e->T + 1; p.dot.D = e->D; p.dot.itself = e->itself;
    p.dot.R = s.R; p.dot.ldr = e->ldR;
    p.has_dot = true;
    p.ok = stream_applicable(p.a, p.b, e->tune, xs, &p.dot, nullptr, can_defer);   // only the stream kernel has the interaction
  }
  return p;
}

bool fused_applicable(const drs_engine* e, const Slot& s, int64_t Mv, const XSrc* xs) {
  return fused_plan(e, s, Mv, s.d_out, xs).ok;
}

bool try_fused_bottom_top(drs_engine* e, Slot& s, int64_t Mv, float* out, const Done* dp,
                          const XSrc* xs, int32_t* rc) {
  *rc = DRS_OK;
  FusedPlan p = fused_plan(e, s, Mv, out, xs);
  if (!p.ok) return false;
  hipError_t r = launch_chain2(p.a, &p.b, e->tune, s.stream, dp, xs, p.has_dot ? &p.dot : nullptr);
  if (r != hipSuccess) *rc = fail(e, DRS_ERR_HIP, "launch_chain2: %s", hipGetErrorString(r));
  return true;
}

// shared_stream: 1 = one stream for everything (launch sets strictly back to back);
// 0 = one stream per slot; 2 = pipelined: every gather on stream_g, everything else on the
// first slot's stream behind an event, so the HBM-bound gather of set i+1 runs under the
// latency-bound MLP of set i and the gathers themselves never overlap each other.
void apply_stream_mode(drs_engine* e) {
  // pipelined mode: the MLP launches may alternate between `mlp_streams` streams, so that the
  // latency-bound tail of one set's MLP launch (completion hand-off: one workgroup active)
  // overlaps the start of the next set's
  const int nm = e->mlp_streams < 1 ? 1 : (e->mlp_streams > (int)e->slots.size() ? (int)e->slots.size() : e->mlp_streams);
  int k = 0;
  for (auto& s : e->slots) {
    if (e->shared_stream == 2) s.stream = e->slots[k % nm].own_stream;
    else s.stream = e->shared_stream ? e->slots[0].own_stream : s.own_stream;
    s.base_stream = s.stream;
    s.gather_stream = e->shared_stream == 2 ? e->stream_g : s.stream;
#ifdef DRS_LAB
    if (e->shared_stream == 2 && e->gather_streams == 2 && e->stream_g2 && (k & 1)) s.gather_stream = e->stream_g2;
#endif
    ++k;
  }
}

// the stream the MLP side of a job of Mv virtual rows goes on (see enqueue_forward)
hipStream_t job_stream(const drs_engine* e, const Slot& s, int64_t Mv) {
  return (e->shared_stream == 2 && Mv <= e->mlp_small_rows) ? s.own_stream : s.base_stream;
}
// ... and the stream its gather goes on: a small set runs entirely on the slot's own stream (no
// cross-stream event; short gathers of different slots may overlap -- they are latency-bound,
// by PCIe when the inputs are read in place from host memory)
hipStream_t job_gather_stream(const drs_engine* e, const Slot& s, int64_t Mv) {
  return (e->shared_stream == 2 && Mv <= e->mlp_small_rows && !e->small_piped) ? s.own_stream : s.gather_stream;
}

// Enqueue n >= 1 coalesced queries (query i = first bs[i] samples of *bts[i]) as ONE set of
// launches on the slot's stream.
int32_t enqueue_forward(drs_engine* e, Slot& s, int n, const Batch* const* bts, const int32_t* bss) {
  if (n < 1 || n > DRS_MAX_COALESCE) return fail(e, DRS_ERR_BAD_ARG, "1..%d queries per launch, got %d", DRS_MAX_COALESCE, n);
  for (int t = 0; t < e->T; ++t)
    if (!e->table_set[t]) return fail(e, DRS_ERR_STATE, "table %d has no data", t);
  int32_t rc;
  if ((rc = mlp_ready(e, e->bot, "bottom")) || (rc = mlp_ready(e, e->top, "top")) ||
      (rc = mlp_ready(e, e->fin, "final")))
    return rc;
  for (auto& tk : e->tasks)
    if ((rc = mlp_ready(e, tk, "task"))) return rc;
  for (auto& au : e->att)
    if ((rc = mlp_ready(e, au, "attention"))) return rc;
  for (auto& rn : e->rnn)
    if ((rc = mlp_ready(e, rn, "rnn"))) return rc;
  if (!e->rnn.empty() && e->att_dirty) {
    const int H = e->rnn[0].ln[1];
    if (!e->d_att) {
      std::vector<const float*> hp;
      for (auto& rn : e->rnn) { hp.push_back(rn.layers[0].W); hp.push_back(rn.layers[0].b); hp.push_back(rn.layers[1].W); hp.push_back(rn.layers[1].b); }
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att), sizeof(float*) * hp.size()));
      HIP_TRY(e, hipMemcpy(e->d_att, hp.data(), sizeof(float*) * hp.size(), hipMemcpyHostToDevice));
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att_packed), sizeof(float) * (size_t)dien_packed_floats(e->D, H)));
    }
    HIP_TRY(e, launch_dien_pack(e->d_att, e->d_att_packed, e->D, H, nullptr));
    HIP_TRY(e, hipStreamSynchronize(nullptr));
    e->att_dirty = false;
  }
  if (!e->att.empty() && e->att_dirty) {
    const int U = (int)e->att.size(), h = e->att[0].ln[1];
    if (!e->d_att) {
      std::vector<const float*> hp;
      for (auto& au : e->att) { hp.push_back(au.layers[0].W); hp.push_back(au.layers[0].b); hp.push_back(au.layers[1].W); hp.push_back(au.layers[1].b); }
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att), sizeof(float*) * hp.size()));
      HIP_TRY(e, hipMemcpy(e->d_att, hp.data(), sizeof(float*) * hp.size(), hipMemcpyHostToDevice));
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att_packed), sizeof(float) * (size_t)U * din_unit_stride(e->D, h)));
    }
    // (drs_set_fc is synchronous; nothing of this engine is in flight while weights change)
    HIP_TRY(e, launch_din_pack(e->d_att, e->d_att_packed, U, e->D, h, nullptr));
    HIP_TRY(e, hipStreamSynchronize(nullptr));
    e->att_dirty = false;
  }
  // layout of the job: zero-sized queries take no rows
  QTable q;
  memset(&q, 0, sizeof q);
  const Batch* qb[DRS_MAX_COALESCE];
  int32_t v = 0, c = 0;
  for (int i = 0; i < n; ++i) {
    if (bss[i] < 0 || bss[i] > bts[i]->n_samples)
      return fail(e, DRS_ERR_BAD_ARG, "bs=%d outside [0, %d]", bss[i], bts[i]->n_samples);
    s.q_bs[i] = bss[i];
    s.q_vstart[i] = v;
    if (bss[i] == 0) continue;
    qb[q.n_q] = bts[i];
    q.vstart[q.n_q] = v;
    q.cum[q.n_q] = c;
    q.bs[q.n_q] = bss[i];
    q.n_q++;
    v += (bss[i] + 63) / 64 * 64;   // whole 64-row MLP blocks per query
    c += bss[i];
  }
  q.vstart[q.n_q] = v;
  q.cum[q.n_q] = c;
  if (v > e->max_rows) return fail(e, DRS_ERR_BAD_ARG, "%d coalesced rows exceed the slot capacity %lld", v, (long long)e->max_rows);
  s.last_n = n;
  s.last_bs = c;
  s.busy = true;
  s.polled = false;
  if (c == 0) return DRS_OK;
  const int64_t Mv = v;
  // Pipelined mode, small launch set (a single query: 256 rows = 16 MLP workgroups on a 256-CU
  // chip): its latency-bound MLP launch goes on the SLOT's own stream, so the MLP launches of
  // consecutive sets overlap each other instead of queueing on the one shared MLP stream
  // (one query per launch set: 23 k -> see DESIGN 3.5).  Full sets (8 queries, 128 workgroups)
  // keep the shared stream: there the extra concurrency only takes CUs from the gather.
  // Safe: a slot is reused only after its previous job was observed complete on the host.
  s.stream = job_stream(e, s, Mv);
  s.cur = nullptr;               // (the set's first MLP launch needs no event: join() orders it behind the gather)
  s.dlog.len = 0; s.dlog.text[0] = 0;
  // ("dispatch_log" 1: the launch functions note what they choose for this set -- drs_last_dispatch; off by default: four
  //  to eight vsnprintf per set are ~1 us of the ~11 us a small set costs the host)
  e->tune.log = e->dispatch_log ? &s.dlog : nullptr;
  e->tune.xbuf = s.xbuf; e->tune.xcnt = s.xcnt; e->tune.xbuf_rows = s.xrows; e->tune.xbuf_cols = s.xcols;
  // (both belong to THIS slot: launches made outside this function -- the operator-level entry points -- must not see them)
  struct TuneScope { Tune& t; ~TuneScope() { t.log = nullptr; t.xbuf = nullptr; t.xcnt = nullptr; } } tune_scope{e->tune};
  log_launch(e->tune.log, "set[%d queries, %d rows, gather on %s, mlp on %s]", q.n_q, (int)Mv,
             job_gather_stream(e, s, Mv) == e->stream_g ? "stream_g" : "own", s.stream == s.own_stream ? "own" : "shared");
  const hipStream_t gstream = job_gather_stream(e, s, Mv);
  const bool prof = e->profiling >= 1;
  const bool evts = e->profiling >= 2;
  const bool piped = gstream != s.stream;
  if (evts) HIP_TRY(e, hipEventRecord(s.ev[0], gstream));

  SlsArgs a;
  memset(&a, 0, sizeof a);
  a.tables = e->tables; a.tab_off = e->d_tab_off; a.tab_rows = e->d_tab_rows;
  a.q = q;
  for (int i = 0; i < q.n_q; ++i) {
    a.idx[i] = qb[i]->idx;
    a.off[i] = qb[i]->off;
    a.uniform_len[i] = e->sls_uniform ? qb[i]->uniform_len : -1;
  }
  a.idx_stride = e->cap; a.off_stride = e->max_batch + 1;
  a.out = s.T; a.ld_out = e->ldT; a.col0 = e->kind == DRS_MODEL_NCF ? 0 : e->w0;
  a.T = e->T; a.D = e->D; a.err = reinterpret_cast<int32_t*>(s.d_err);
  a.ts = prof ? s.d_ts : nullptr;
  // Bags of a few rows (W&D / NCF: one lookup per table) would leave most of a wave idle in the
  // wave-per-bag variant: a lane group per bag is both faster there and bit-exact.
  bool short_bags = true;
  for (int i = 0; i < q.n_q; ++i) short_bags = short_bags && qb[i]->uniform_len >= 0 && qb[i]->uniform_len <= e->sls_short_bag;
  // ... unless the flat variant takes the launch (fixed-length bags of >= 2 rows: several short
  // bags share a wave and all of its row loads are in flight at once)
  const int exact_now = e->sls_exact || (short_bags && !sls_flat_applicable(a, e->tune));
  // DIN, default mode: the attention units are fused into the gather launch (din.hip)
  const bool din_fused = e->kind == DRS_MODEL_DIN && !e->sls_exact && e->din_fused &&
                         din_fused_applicable(e->D, e->att[0].ln[1]);
  s.ts_blocks = prof ? (din_fused ? din_fused_grid(a, e->tune) : sls_grid_blocks(a, exact_now, e->tune)) : 0;
  if (prof) {
    // algorithmic bytes of THIS launch (rows + int32 indices + length + pooled output
    // per bag), so that achieved GB/s = sum(bytes) / sum(duration) over exactly the timed launches
    int64_t bytes = 0;
    for (int i = 0; i < q.n_q; ++i)
      for (int t = 0; t < e->T; ++t)
        bytes += (int64_t)qb[i]->h_off[(size_t)t * (e->max_batch + 1) + q.bs[i]] * ((int64_t)e->D * 4 + 4) +
                 (int64_t)q.bs[i] * (4 + (int64_t)e->D * 4);
    // (the fused DIN launch writes the 4 D floats of the top MLP's input row per sample instead
    // of T pooled vectors)
    if (din_fused) bytes -= (int64_t)c * (e->T - 4) * e->D * 4;
    s.ts_bytes = bytes;
  }
  // pipelined mode: the event the MLP stream waits for is recorded by the gather dispatch itself
  // (hipExtLaunchKernel's stop event = the packet's completion signal): no marker packet sits
  // between consecutive gathers (a hipEventRecord there costs ~2 us per set)
  if (din_fused)
    HIP_TRY(e, launch_din_fused(a, e->att[0].ln[1], e->d_att_packed, s.R, e->ldR, e->tune, gstream, piped ? s.ev_sls : nullptr));
  else
    HIP_TRY(e, launch_sls(a, exact_now, e->tune, gstream, piped ? s.ev_sls : nullptr));
  if (evts) HIP_TRY(e, hipEventRecord(s.ev[1], gstream));
  bool joined = !piped;   // has s.stream been made to wait for the gather yet?
  auto join = [&]() -> hipError_t {
    if (joined) return hipSuccess;
    joined = true;
    return hipStreamWaitEvent(s.stream, s.ev_sls, 0);
  };

  // last kernel of the job: outputs either go straight to host-mapped pinned memory
  // followed by a flag store (zero copy, no stream sync), or to a device buffer + memcpy
  s.seq += 1;
  if (s.seq == 0) s.seq = 1;
  Done done;
  memset(&done, 0, sizeof done);
  done.counter = s.d_counter; done.host_flag = s.dm_out; done.host_err = s.dm_out + 1;
  done.dev_err = s.d_err; done.seq = s.seq;
  if (prof && e->zero_copy) { done.ts = s.d_ts; done.ts_blocks = (uint32_t)s.ts_blocks; done.span_acc = s.d_span_acc; done.host_span = s.dm_span; }
  const Done* dp = e->zero_copy ? &done : nullptr;
  float* out = s.d_out;          // kernels store to the device buffer; see Done::host_out
  done.dev_out = s.d_out; done.host_out = reinterpret_cast<float*>(s.dm_out + kOutOffset);
  done.out_words = (uint32_t)(Mv * e->n_out);
  // "out_dma": the outputs leave through a copy-engine transfer queued behind the last kernel, and the flag through
  // a stream-ordered 32-bit write behind that -- the last workgroup then hands over the error word only
  const bool out_dma = e->zero_copy && e->out_dma && (int64_t)done.out_words * 4 >= e->out_dma;
  if (out_dma) { done.out_words = 0; done.host_flag = nullptr; }
  XSrc xs;
  memset(&xs, 0, sizeof xs);
  xs.q = q;
  for (int i = 0; i < q.n_q; ++i) xs.x[i] = qb[i]->dense;
  if (e->kind == DRS_MODEL_DIEN) {
    // the two recurrent layers over the pooled behaviour rows -> top MLP input R [rows, H + 3D]
    HIP_TRY(e, join());
    const float* rw[8];
    for (int l = 0; l < 2; ++l) {
      rw[4 * l + 0] = e->rnn[l].layers[0].W; rw[4 * l + 1] = e->rnn[l].layers[0].b;
      rw[4 * l + 2] = e->rnn[l].layers[1].W; rw[4 * l + 3] = e->rnn[l].layers[1].b;
    }
    // the top MLP in the recurrence's own launch when it fits ("dien_fuse_top", default on): one workgroup per 16
    // samples instead of two, one launch less per set
    const int Hh = e->rnn[0].ln[1];
    DienTop tp;
    memset(&tp, 0, sizeof tp);
    const int nt = (int)e->top.layers.size();
    if (e->dien_fuse_top && e->dien_mfma && Hh % 16 == 0 && nt >= 1 && nt <= 4 && e->top.layers[0].packed &&
        dien_top_fusable(nt, e->top.ln.data(), Hh) && e->top.ln[0] == Hh + 3 * e->D) {
      tp.n = nt; tp.sc1 = dp ? 1 : 0; tp.out = out; tp.ldo = e->n_out;
      for (int l = 0; l < nt; ++l) {
        const Layer& L = e->top.layers[l];
        tp.Wp[l] = L.W + ((size_t)L.m * L.n + 63) / 64 * 64; tp.b[l] = L.b;
        tp.K[l] = e->top.ln[l]; tp.N[l] = e->top.ln[l + 1]; tp.act[l] = act_of(e->top, l);
      }
      tp.kmax = dien_top_kmax(nt, e->top.ln.data());
    }
    log_launch(e->tune.log, "%s<%d,%d%s>[%d wg]", e->dien_mfma && Hh % 16 == 0 ? "dien_rnn_mfma_kernel" : "dien_rnn_kernel", e->D, Hh,
               tp.n ? ",top" : "", e->dien_mfma && Hh % 16 == 0 ? (c + 15) / 16 : (c + 3) / 4);
    HIP_TRY(e, launch_dien_rnn(s.T, e->ldT, q, e->T, e->D, Hh, e->d_att_packed, rw, e->dien_mfma, s.R,
                               e->ldR, s.stream, tp.n ? &tp : nullptr, dp));
    if (!tp.n && (rc = run_mlp(e, s, e->top, s.R, e->ldR, Mv, out, e->n_out, dp))) return rc;
  } else if (e->kind == DRS_MODEL_DIN) {
    // attention units over the pooled rows -> top MLP input R [rows, 4D] -> top MLP (all ReLU)
    HIP_TRY(e, join());
    if (!din_fused) log_launch(e->tune.log, "din_attention_kernel[%lld wg]", (long long)((Mv + 3) / 4));
    if (!din_fused)
      HIP_TRY(e, launch_din_attention(s.T, e->ldT, Mv, e->T, e->D, e->att[0].ln[1], e->d_att_packed, s.R, e->ldR, s.stream));
    if ((rc = run_mlp(e, s, e->top, s.R, e->ldR, Mv, out, e->n_out, dp))) return rc;
  } else if (e->kind == DRS_MODEL_NCF) {
    // mf = Sum(sls0, sls1); mlp = Concat(sls2, sls3) -> MLP; Concat(mf, mlp_out) -> FC+Relu
    const int D = e->D;
    const int wl = e->top.ln.back();
    const int64_t ldc = D + wl;
    HIP_TRY(e, join());
    // one launch when it fits: Sum, MLP branch and predictor of a 16-row slab in the stream kernel
    bool fused = false;
    const int nt = (int)e->top.layers.size();
    if (e->mlp_fuse && nt >= 1 && nt <= DRS_MAX_CHAIN && e->fin.layers.size() == 1) {
      ChainArgs ca, cb;
      fill_chain(ca, e->top, 0, nt, s.T + 2 * D, e->ldT, Mv, s.H2 + D, ldc);
      fill_chain(cb, e->fin, 0, 1, s.H2, ldc, Mv, out, e->n_out);
      SumArgs sum = {s.T, e->ldT, 0, D, D, s.H2, ldc};
      bool wide = false;
      for (int l = 0; l < nt; ++l) wide = wide || is_wide(e, e->top, l);
      if (!wide && !is_wide(e, e->fin, 0) && stream_applicable(ca, cb, e->tune, nullptr, nullptr, &sum)) {
        HIP_TRY(e, launch_chain2(ca, &cb, e->tune, s.stream, dp, nullptr, nullptr, &sum));
        fused = true;
      }
    }
    if (!fused) {
      log_launch(e->tune.log, "add_rows_kernel");
      HIP_TRY(e, launch_add_rows(s.T, e->ldT, s.T + D, e->ldT, s.H2, ldc, Mv, D, s.stream));
      if ((rc = run_mlp(e, s, e->top, s.T + 2 * D, e->ldT, Mv, s.H2 + D, ldc))) return rc;
      if ((rc = run_mlp(e, s, e->fin, s.H2, ldc, Mv, out, e->n_out, dp))) return rc;
    }
  } else {
    bool fused = false;
    bool split_top = false;      // the first top layer reads the dense columns in place (xs_top)
    s.split_last = false;
    XSrc xs_top;
    memset(&xs_top, 0, sizeof xs_top);
    if (!e->bot.layers.empty()) {
      // "mlp_early": a small set of staged queries (gather on the slot's own stream, nothing else of the set there) whose
      // bottom + top MLP is ONE stream4_kernel launch: the launch goes on a second stream WITHOUT waiting for the gather,
      // runs its prologue and the bottom chain beside it and polls the slot's flag -- a 32-bit write queued behind the
      // gather -- before it fetches the pooled rows (mlp.hip, Done::wait_flag).  <= 512 rows: at most 32 workgroups spin.
      bool early = false;
      if (e->mlp_early && e->shared_stream == 2 && !piped && gstream == s.own_stream && Mv <= 512 && dp && s.early_stream &&
          e->kind == DRS_MODEL_DLRM) {
        bool staged = true;
        for (int i = 0; i < q.n_q; ++i)
          staged = staged && qb[i] >= e->batches.data() && qb[i] < e->batches.data() + e->batches.size();
        if (staged) {
          FusedPlan fp = fused_plan(e, s, Mv, out, &xs, &early);
          early = early && fp.ok;
        }
      }
      if (early) {
        HIP_TRY(e, hipStreamWriteValue32(gstream, s.d_gflag, s.seq, 0));
        done.wait_flag = s.d_gflag; done.wait_val = s.seq;
        s.stream = s.early_stream;
        log_launch(e->tune.log, "early");
      }
      if (fused_applicable(e, s, Mv, &xs)) HIP_TRY(e, join());
      fused = try_fused_bottom_top(e, s, Mv, out, dp, &xs, &rc);
      if (rc) return rc;
    }
    if (fused) {
      // nothing else to launch
    } else if (e->bot.layers.empty()) {
      // W&D / MT-WnD: Concat(dense, pooled embeddings) feeds the first top layer.  When that layer goes to a GEMM form
      // that can read a split row ("gemm_split", launch_gemm) it takes the dense columns from the queries' own arrays;
      // otherwise the dense rows are copied in front of the embeddings first.
      XSrc xsp = xs;
      xsp.ksplit = e->m_den;
      if (e->gemm_split && !e->top.layers.empty() && is_wide(e, e->top, 0) &&
          gemm_split_applicable(s.T, e->ldT, Mv, e->top.ln[0], e->top.layers[0].W, e->top.ln[1], xsp, e->tune)) {
        xs_top = xsp;
        split_top = true;
        s.split_last = true;
      } else {
        log_launch(e->tune.log, "copy_rows_multi_kernel");
        HIP_TRY(e, launch_copy_rows_multi(xs, e->m_den, s.T, e->ldT, s.stream));
        s.cur = s.stream;          // ("mlp_layout" 1: a wide first layer routed to the gather's stream must wait for this copy)
      }
    } else {
      if ((rc = run_mlp(e, s, e->bot, nullptr, e->m_den, Mv, s.T, e->ldT, nullptr, &xs))) return rc;
    }
    HIP_TRY(e, join());   // (the bottom MLP above ran beside the gather)
    const float* top_in = s.T;
    int64_t ld_top = e->ldT;
    if (!fused && e->kind == DRS_MODEL_DLRM && e->interaction_op == DRS_INTERACT_DOT) {
      HIP_TRY(e, rejoin_stream(e, s));
      log_launch(e->tune.log, "interact_dot_kernel[%lld wg]", (long long)((Mv + 3) / 4));
      HIP_TRY(e, launch_interact_dot(s.T, e->ldT, Mv, e->T + 1, e->D, e->itself, s.R, e->ldR, s.stream));
      top_in = s.R;
      ld_top = e->ldR;
    }
    if (e->kind == DRS_MODEL_MTWND) {
      // shared top MLP (all ReLU) -> H3, then every task head reads H3 and writes its block of
      // the output row; the last head's last launch carries the completion hand-off
      const int wt = e->top.ln.back(), wo = e->tasks[0].ln.back();
      if ((rc = run_mlp(e, s, e->top, top_in, ld_top, Mv, s.H3, wt, nullptr, split_top ? &xs_top : nullptr))) return rc;
      for (size_t k = 0; k < e->tasks.size(); ++k)
        if ((rc = run_mlp(e, s, e->tasks[k], s.H3, wt, Mv, out + k * wo, e->n_out,
                          k + 1 == e->tasks.size() ? dp : nullptr)))
          return rc;
    } else if (!fused && (rc = run_mlp(e, s, e->top, top_in, ld_top, Mv, out, e->n_out, dp, split_top ? &xs_top : nullptr))) return rc;
  }
  HIP_TRY(e, rejoin_stream(e, s));     // ("mlp_layout" 1: the tail below is ordered behind a last launch on the gather's stream)
  if (evts) {
    HIP_TRY(e, hipEventRecord(s.ev[2], s.stream));
    s.ev_pending = true;
  }
  if (!e->zero_copy) {
    HIP_TRY(e, hipMemcpyAsync(s.h_out + kOutOffset, s.d_out, sizeof(float) * (size_t)Mv * e->n_out,
                              hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(e, hipMemcpyAsync(s.h_out + 1, s.d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
  }
  if (out_dma) {
    log_launch(e->tune.log, "out_dma[%lld B]", (long long)(sizeof(float) * (size_t)Mv * e->n_out));
    HIP_TRY(e, hipMemcpyAsync(s.h_out + kOutOffset, s.d_out, sizeof(float) * (size_t)Mv * e->n_out,
                              hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(e, hipStreamWriteValue32(s.stream, s.dm_out, s.seq, 0));
  }
  s.polled = e->zero_copy != 0;
  return DRS_OK;
}

// Everything of a per-call-input query that is a HIP call: the one DMA copy of its converted block
// (copy mode 2), the cross-stream events, the launches.  Runs on the calling thread or, with
// "launch_thread" 1, on the launcher thread.
int32_t finish_inputs(drs_engine* e, Slot& s, int mode, int32_t bs, size_t used, bool need_off) {
  const Batch* bt;
  const int64_t Mv = ((int64_t)bs + 63) / 64 * 64;
  if (mode == 2) {
    const hipStream_t gstream = job_gather_stream(e, s, Mv);
    const size_t dense_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1);
    const size_t idx_bytes = sizeof(int32_t) * (size_t)e->T * e->cap;
    HIP_TRY(e, hipMemcpyAsync(s.d_stage, s.h_stage, used, hipMemcpyHostToDevice, gstream));
    // ragged bags -- or "sls_uniform" 0, which makes enqueue_forward hand the kernels uniform_len = -1
    // for fixed-length bags too: the kernels then read the prefix sums as well (same predicate)
    if (need_off)
      HIP_TRY(e, hipMemcpyAsync(s.d_stage + dense_bytes + idx_bytes, static_cast<char*>(s.h_stage) + dense_bytes + idx_bytes,
                                sizeof(int32_t) * (size_t)e->T * (e->max_batch + 1), hipMemcpyHostToDevice, gstream));
    if (gstream != s.stream) {   // the MLP side reads the dense rows: order it behind the copy
      HIP_TRY(e, hipEventRecord(s.ev_in, gstream));
      HIP_TRY(e, hipStreamWaitEvent(s.stream, s.ev_in, 0));
    }
    bt = &s.dc;
  } else if (mode == 1) {
    bt = &s.zc;
  } else {
    const hipStream_t gstream = job_gather_stream(e, s, Mv);
    if (gstream != s.stream) {   // the gather runs on another stream: order it behind the copies
      HIP_TRY(e, hipEventRecord(s.ev_in, s.stream));
      HIP_TRY(e, hipStreamWaitEvent(gstream, s.ev_in, 0));
    }
    bt = &s.scratch;
  }
  return enqueue_forward(e, s, 1, &bt, &bs);
}

// The launcher thread of the per-call input path: takes jobs in FIFO order and makes their HIP calls
// (finish_inputs).  Spins ~50 us for the next job, then sleeps.
class Launcher {
 public:
  explicit Launcher(drs_engine* e) : e_(e), th_([this] { loop(); }) {}
  ~Launcher() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
    cv_.notify_all();
    th_.join();
  }
  void push(int slot, int mode, int32_t bs, size_t used, bool need_off) {
    e_->launch_state[slot].store(1, std::memory_order_release);
    { std::lock_guard<std::mutex> l(mu_); q_.push_back(Job{slot, mode, bs, used, need_off}); }
    pushed_.fetch_add(1, std::memory_order_release);
    if (sleeping_.load(std::memory_order_acquire)) cv_.notify_one();
  }
  // every job handed over so far has been launched (other entry points call this before they touch
  // streams or slots themselves)
  void drain() {
    while (done_.load(std::memory_order_acquire) != pushed_.load(std::memory_order_acquire)) __builtin_ia32_pause();
  }

 private:
  struct Job { int slot; int mode; int32_t bs; size_t used; bool need_off; };
  bool pop(Job* j) {
    std::lock_guard<std::mutex> l(mu_);
    if (q_.empty()) return false;
    *j = q_.front();
    q_.erase(q_.begin());
    return true;
  }
  void loop() {
    (void)hipSetDevice(e_->device);
    for (;;) {
      Job j;
      bool got = false;
      for (int spin = 0; spin < 20000 && !got; ++spin) {
        if (done_.load(std::memory_order_relaxed) != pushed_.load(std::memory_order_acquire)) got = pop(&j);
        else __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> l(mu_);
        sleeping_.store(true, std::memory_order_release);
        cv_.wait(l, [&] { return stop_ || !q_.empty(); });
        sleeping_.store(false, std::memory_order_release);
        if (q_.empty()) { if (stop_) return; continue; }
        j = q_.front();
        q_.erase(q_.begin());
      }
      Slot& s = e_->slots[j.slot];
      s.launch_rc = finish_inputs(e_, s, j.mode, j.bs, j.used, j.need_off);
      if (s.launch_rc) { std::lock_guard<std::mutex> l(e_->err_mu); s.launch_err = e_->err; }
      e_->launch_state[j.slot].store(2, std::memory_order_release);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  drs_engine* e_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Job> q_;
  std::atomic<uint64_t> pushed_{0}, done_{0};
  std::atomic<bool> sleeping_{false};
  bool stop_ = false;
  std::thread th_;      // (last: the members above exist before it starts)
};

int32_t wait_slot(drs_engine* e, Slot& s, float* h_out, int64_t h_cap = -1) {
  if (!s.busy) return DRS_OK;
  if (e->launch_state) {
    // the job's launches may still be with the launcher thread
    std::atomic<int>& st = e->launch_state[&s - e->slots.data()];
    if (st.load(std::memory_order_acquire) != 0) {
      while (st.load(std::memory_order_acquire) == 1) __builtin_ia32_pause();
      st.store(0, std::memory_order_relaxed);
      if (s.launch_rc) {
        s.busy = false;
        const int32_t rc = s.launch_rc;
        s.launch_rc = 0;
        return fail(e, rc, "%s", s.launch_err.c_str());
      }
    }
  }
  // the caller's buffer must hold what was SUBMITTED on this slot (a mismatched bs
  // after a multi-query submit used to overflow the heap silently); the job stays in flight
  if (h_out && h_cap >= 0 && h_cap < (int64_t)s.last_bs * e->n_out)
    return fail(e, DRS_ERR_BAD_ARG, "output buffer holds %lld floats, the %d queries on this slot produce %lld",
                (long long)h_cap, s.last_n, (long long)s.last_bs * e->n_out);
  if (s.polled && s.last_bs > 0) {
    // spin on the flag the last kernel publishes (bounded: fall back to a stream sync)
    volatile uint32_t* flag = s.h_out;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (*flag != s.seq) {
      __builtin_ia32_pause();
      // 8 ranks on a node share its cores with each other's runtime threads: do not starve them
      if ((++spins & 0xfff) == 0) sched_yield();
      if ((spins & 0xfffff) == 0 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
        HIP_TRY(e, hipStreamSynchronize(s.stream));
        if (*flag != s.seq) {
          s.busy = false;
          return fail(e, DRS_ERR_HIP, "completion flag never arrived (seq %u, flag %u)", s.seq, *flag);
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    HIP_TRY(e, hipStreamSynchronize(s.stream));
  }
  s.busy = false;
  if (s.ev_pending) {
    HIP_TRY(e, hipEventSynchronize(s.ev[2]));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) { e->k_ms[DRS_KERNEL_SLS] += ms; e->k_n[DRS_KERNEL_SLS]++; e->k_bytes[DRS_KERNEL_SLS] += s.ts_bytes; }
    if (hipEventElapsedTime(&ms, s.ev[1], s.ev[2]) == hipSuccess) { e->k_ms[DRS_KERNEL_MLP] += ms; e->k_n[DRS_KERNEL_MLP]++; }
    s.ev_pending = false;
  }
  if (s.ts_blocks > 0) {
    uint64_t lo = ~0ull, hi = 0;
    if (s.polled) {
      lo = s.h_span[0]; hi = s.h_span[1];       // reduced on the device, see Done
    } else {
      HIP_TRY(e, hipMemcpy(s.h_ts.data(), s.d_ts, sizeof(uint64_t) * 2 * (size_t)s.ts_blocks, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < s.ts_blocks; ++i) {
        lo = s.h_ts[2 * i] < lo ? s.h_ts[2 * i] : lo;
        hi = s.h_ts[2 * i + 1] > hi ? s.h_ts[2 * i + 1] : hi;
      }
    }
    if (hi > lo) {
      e->k_ms[DRS_KERNEL_SLS_CLOCK] += (double)(hi - lo) / e->wall_clock_khz;
      e->k_n[DRS_KERNEL_SLS_CLOCK]++;
      e->k_bytes[DRS_KERNEL_SLS_CLOCK] += s.ts_bytes;
    }
    s.ts_blocks_done = s.ts_blocks;
    s.ts_blocks = 0;
  }
  if (s.last_bs > 0 && s.h_out[1] != 0) {
    const uint32_t bits = s.h_out[1];
    s.h_out[1] = 0;
    HIP_TRY(e, hipMemsetAsync(s.d_err, 0, sizeof(uint32_t), s.stream));
    HIP_TRY(e, hipStreamSynchronize(s.stream));
    if (bits & 2u) return fail(e, DRS_ERR_HIP, "mlp_early: the gather's flag never reached the MLP launch");
    return fail(e, DRS_ERR_INDEX_RANGE, "an embedding index was out of range on the device");
  }
  if (h_out && s.last_bs > 0) {
    // queries sit at 16-row aligned virtual offsets: pack them back to back
    const float* src = reinterpret_cast<const float*>(s.h_out + kOutOffset);
    size_t o = 0;
    for (int i = 0; i < s.last_n; ++i) {
      memcpy(h_out + o, src + (size_t)s.q_vstart[i] * e->n_out, sizeof(float) * (size_t)s.q_bs[i] * e->n_out);
      o += (size_t)s.q_bs[i] * e->n_out;
    }
  }
  return DRS_OK;
}

// (hot = the per-call input path itself, which may have jobs with the launcher thread; every other
// entry point first lets that thread finish what it was handed)
int32_t check_handle(drs_engine* e, bool hot = false) {
  if (!e) return fail(nullptr, DRS_ERR_BAD_ARG, "null handle");
  if (!hot && e->launcher) e->launcher->drain();
  return set_device(e);
}

}  // namespace

// =============================================================================
extern "C" {

int32_t drs_abi_version(void) { return DRS_ABI_VERSION; }
const char* drs_backend(void) { return "hip:gfx950"; }

int32_t drs_device_count(int32_t* out_count) {
  if (!out_count) return DRS_ERR_BAD_ARG;
  int n = 0;
  hipError_t r = hipGetDeviceCount(&n);
  if (r != hipSuccess) {
    *out_count = 0;
    return fail(nullptr, DRS_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(r));
  }
  *out_count = n;
  return DRS_OK;
}

const char* drs_last_error(drs_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// ---- which launch forms a model's sets take ------------------------------------------------------------
// Decided ONCE per engine, here and nowhere else, from the model's shape; what each launch then becomes also depends
// on its row count (mlp.hip stream_plan, gemm.hip launch_gemm, sls.hip flat_plan) -- the resulting table is
// DESIGN.md 3 / profiles/r05_dispatch.md, read back through drs_last_dispatch and asserted by
// test_dispatch_table_of_the_bench_workloads.  The numbers behind every choice are same-session A/Bs
// (docs/DESIGN_rounds_1-4.md).
//
//   class (MLP FLOP per gathered byte, per sample)     MLP streams  stream kernel              rows32 from   GEMM forms
//   gather-bound DLRM (RMC1 2, RM2 0.6)                1            stream4, one WG per CU     2 048 rows    --
//   in-between DLRM (dlrm_rm1.json: MLP launch         2            stream4, two WGs per CU    4 096 rows    --
//     outlasts its gather)
//   MLP-bound DLRM (RM3 230)                           up to 4      stream4, two WGs per CU    8 192 rows    2cu; gemm32 64 x 128 from 256 tiles
//   W&D (440), DIEN (200)                              up to 4      stream4, two WGs per CU    never         W&D: 2cu; gemm32 64 x 128 from 512 tiles
//   MT-WnD                                             up to 4      stream_kernel, two per CU  --            gemm32 64 x 128 from 256 tiles
//   DIN                                                1            stream_kernel, two per CU  --            --
//   NCF (145)                                          up to 4      stream_kernel, one per CU  --            --
static void choose_launch_forms(drs_engine* e) {
  const int T = e->T, D = e->D;
  // A wave of the wave-split gather takes 256/D rows per load instruction: a bag shorter than 8 such instructions cannot
  // fill its load rings, and a lane group per bag (the sequential variant, which is also bit-exact) is faster: RM3
  // (D=32, L=20) 16.9 -> 11.6 us, W&D / NCF (L=1) 2x; RM1 (L=80) stays wave-split.
  e->sls_short_bag = 2048 / D;
  // Which side bounds a launch set?  Gather-bound models keep ONE MLP stream (more only takes CUs from the gather that sets
  // the pace); MLP-bound ones let the MLP launches of consecutive sets overlap on one stream per slot (W&D 57 k -> 68 k q/s,
  // RM3 39 k -> 50 k, NCF 128 k -> 200 k; RM1 122 k -> 100 k, hence the rule).
  double flop = 0;
  for (const Mlp* mm : {&e->bot, &e->top, &e->fin})
    for (size_t i = 0; i + 1 < mm->ln.size(); ++i) flop += 2.0 * mm->ln[i] * (mm->ln[i + 1] > 0 ? mm->ln[i + 1] : 64);
  // (DIEN: the recurrence, (T - 3) steps of two layers)
  for (const Mlp& rn : e->rnn) flop += 2.0 * (T - 3) * ((double)rn.ln[0] * rn.ln[1] + (double)rn.ln[1] * rn.ln[2]);
  const double bytes = (double)T * e->max_lookups * D * 4.0;
  const bool mlp_bound = flop / bytes > 20.0;
  e->mlp_streams = mlp_bound ? (e->n_slots < 4 ? e->n_slots : 4) : 1;
  // In between: a gather-bound DLRM whose full launch set gathers FASTER than its latency-bound MLP launch runs (the
  // reference's own dlrm_rm1.json, D = 32: 33 us of gather against a 40 us launch): two MLP streams hand the pace back to
  // the gather (186 k -> 200 k queries/s; RMC1 BASELINE within noise; DIN 158 k -> 147 k, hence an estimate instead of a
  // blanket 2: gather at 5.5 TB/s, MLP launch 12 us + 1 us per 4 500 weights).
  bool in_between = false;
  if (!mlp_bound && e->n_slots >= 2 && e->kind == DRS_MODEL_DLRM) {
    double weights = 0;
    for (const Mlp* mm : {&e->bot, &e->top})
      for (size_t i = 0; i + 1 < mm->ln.size(); ++i) weights += (double)mm->ln[i] * mm->ln[i + 1];
    const double gather_us = 2048.0 * bytes / 5.5e6, mlp_us = 12.0 + weights / 4500.0;
    if (mlp_us > gather_us) { e->mlp_streams = 2; in_between = true; }
  }
  const bool dlrm = e->kind == DRS_MODEL_DLRM;
  const bool gather_bound_dlrm = dlrm && !mlp_bound && !in_between;
  // ("gather_bound", read only: the models whose set period is their gather launch -- where the tables live and which
  //  policy their rows are read with is worth a search, DLRM_Net.tune_table_placement)
  e->gather_bound = (dlrm && !mlp_bound) || e->kind == DRS_MODEL_DIN;
  // stream kernel: stream4_kernel for DLRM, W&D, DIEN and DIN (W&D 95.1 k -> 96.2 k, DIEN 168 k -> 172 k, DIN beside the
  // pipelined fused launch 170.6 k -> 172.7 k; MT-WnD -4 %, NCF -9 % keep stream_kernel on the packed twins)
  if (dlrm || e->kind == DRS_MODEL_WND || e->kind == DRS_MODEL_DIEN || e->kind == DRS_MODEL_DIN) e->tune.mlp_stream = 4;
  // 32 rows per workgroup: gather-bound DLRM from 2 048 rows (96 workgroups beside the next set's gather instead of 192:
  // +1.6-4 %), in-between DLRM from 4 096 (237.6 k -> 241.8 k), MLP-bound DLRM from 8 192 (RM3 config 3's top chain 80 -> 67 us;
  // at 4 096 rows the form loses: W&D 96.1 k -> 94.5 k)
  if (gather_bound_dlrm) e->tune.mlp_rows32 = 2048;
  else if (in_between) e->tune.mlp_rows32 = 4096;
  else if (dlrm) e->tune.mlp_rows32 = 8192;
  // two workgroups per CU (the 128-VGPR builds) for every model whose MLP launches overlap each other (DIEN +8 %, W&D +5 %,
  // MT-WnD +4 %, DIN +3 %, RM3 +2 %; NCF -2 %; gather-bound DLRM keeps one per CU: 54.6 k against 53.4 k at one query per set)
  e->tune.mlp_stream_2cu = e->kind != DRS_MODEL_NCF && !gather_bound_dlrm;
  // column-split form of the fused DLRM launch (mlp.hip NSplit): launch sets of one or two queries spread their widest
  // layer over four workgroups per slab of rows (RMC1, one query per set: 54.3 k -> 59.5 k queries/s, two: 84.7 k ->
  // 87-91 k, three: equal, four: 103 k -> 89 k -- 256 workgroups that each repeat the bottom chain; profiles/r06_nsplit/)
  e->tune.mlp_nsplit_rows = 512;
  e->tune.mlp_nsplit = dlrm ? 4 : 0;      // (dlrm_rm1.json: 65.0 k -> 73.5 k, 111.9 k -> 120.0 k; dot interaction: equal, +5 %)
  // wide layers: two 64 x 64 gemm_kernel workgroups per CU where that measured faster; gemm32_kernel's 64 x 128 workgroups
  // for launches below 512 tiles of 128 x 128 when they number at least "mlp_gemm32_small_blocks" (k queries/s, off | >= 0 | >= 512:
  // MT-WnD 69.2 | 71.8 | 66.4; RM3 reference JSON 66.5 | 68.1 | 72.2; RM3 config 3 34.8 | 35.2 | 34.7; W&D 96.0 | 94.7 | 97.0)
  e->tune.gemm_2cu = dlrm || e->kind == DRS_MODEL_WND;
  if (e->kind == DRS_MODEL_MTWND || (dlrm && e->mlp_streams > 1)) { e->tune.gemm32_small = 12; e->tune.gemm32_small_blocks = 256; }
  if (e->kind == DRS_MODEL_WND) { e->tune.gemm32_small = 12; e->tune.gemm32_small_blocks = 512; }
}

int32_t drs_create(const drs_model_cfg* cfg, int32_t device_id, drs_handle* out) {
  if (!cfg || !out) return fail(nullptr, DRS_ERR_BAD_ARG, "null cfg/out");
  *out = nullptr;
  if (cfg->num_tables <= 0 || !cfg->table_rows || cfg->n_bot < 1 || !cfg->ln_bot || cfg->n_top < 2 ||
      !cfg->ln_top || cfg->max_batch <= 0 || cfg->max_lookups <= 0 || cfg->num_staged_batches < 0)
    return fail(nullptr, DRS_ERR_BAD_ARG, "bad model config");
  const int D = cfg->sparse_dim;
  if (D <= 0 || D > 256 || (D & 3))
    return fail(nullptr, DRS_ERR_UNSUPPORTED, "sparse_dim=%d must be a multiple of 4 in [4, 256]", D);
  int ndev = 0;
  hipError_t r = hipGetDeviceCount(&ndev);
  if (r != hipSuccess || ndev <= 0)
    return fail(nullptr, DRS_ERR_HIP, "no HIP device visible (%s); this library has no CPU fallback",
                r == hipSuccess ? "device count 0" : hipGetErrorString(r));
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, DRS_ERR_BAD_ARG, "device %d of %d", device_id, ndev);

  drs_engine* e = new drs_engine();
  e->device = device_id;
  e->kind = cfg->model_kind; e->T = cfg->num_tables; e->D = D;
  e->rows.assign(cfg->table_rows, cfg->table_rows + e->T);
  e->interaction_op = cfg->interaction_op; e->itself = cfg->interaction_itself ? 1 : 0;
  e->max_batch = cfg->max_batch; e->max_lookups = cfg->max_lookups;
  e->n_batches = cfg->num_staged_batches; e->n_slots = cfg->num_slots > 0 ? cfg->num_slots : 1;
  e->bot.ln.assign(cfg->ln_bot, cfg->ln_bot + cfg->n_bot);
  e->top.ln.assign(cfg->ln_top, cfg->ln_top + cfg->n_top);
  e->bot.layers.resize(cfg->n_bot - 1);
  e->top.layers.resize(cfg->n_top - 1);
  e->top.sigmoid_layer = cfg->sigmoid_top;
  const int T = e->T, F = T + 1;

  auto bail = [&](int32_t code, const char* msg) {
    g_create_error = msg;
    drs_destroy(e);
    return code;
  };
  // shape algebra of the reference builders
  switch (e->kind) {
    case DRS_MODEL_DLRM: {
      e->m_den = e->bot.ln.front();
      e->w0 = e->bot.ln.back();
      if (e->w0 != D) return bail(DRS_ERR_BAD_ARG, "arch_sparse_feature_size does not match last dim of bottom mlp");
      if (e->interaction_op == DRS_INTERACT_DOT)
        e->num_int = (e->itself ? F * (F + 1) / 2 : F * (F - 1) / 2) + D;
      else if (e->interaction_op == DRS_INTERACT_CAT)
        e->num_int = F * D;
      else
        return bail(DRS_ERR_BAD_ARG, "unknown interaction op");
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_WND: {
      if (cfg->n_bot != 1) return bail(DRS_ERR_BAD_ARG, "W&D has no bottom MLP layers");
      e->m_den = e->w0 = e->bot.ln.front();
      if (e->w0 & 3) return bail(DRS_ERR_UNSUPPORTED, "dense width must be a multiple of 4");
      e->num_int = T * D + e->w0;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "num_int does not match first dim of top mlp");
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_MTWND: {
      if (cfg->n_bot != 1) return bail(DRS_ERR_BAD_ARG, "MT-W&D has no bottom MLP layers");
      e->m_den = e->w0 = e->bot.ln.front();
      if (e->w0 & 3) return bail(DRS_ERR_UNSUPPORTED, "dense width must be a multiple of 4");
      e->num_int = T * D + e->w0;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "num_int does not match first dim of top mlp");
      if (cfg->n_task < 2 || !cfg->ln_task || cfg->num_tasks < 1 || cfg->num_tasks > 64)
        return bail(DRS_ERR_BAD_ARG, "MT-W&D needs arch_mlp_tasks and 1..64 task heads");
      if (cfg->ln_task[0] != e->top.ln.back())
        return bail(DRS_ERR_BAD_ARG, "Shared top layer and task MLP layers must have same input/output dimension");
      e->tasks.resize(cfg->num_tasks);
      for (auto& tk : e->tasks) {
        tk.ln.assign(cfg->ln_task, cfg->ln_task + cfg->n_task);
        tk.layers.resize(cfg->n_task - 1);
        tk.sigmoid_layer = cfg->sigmoid_top;     // multi_task_wnd.py:309 passes self.sigmoid_top to the heads
      }
      e->top.sigmoid_layer = -1;                 // :301 create_mlp(self.ln_top, -1, ...)
      e->n_out = cfg->num_tasks * cfg->ln_task[cfg->n_task - 1];
      break;
    }
    case DRS_MODEL_DIN: {
      if (T < 4) return bail(DRS_ERR_BAD_ARG, "DIN needs at least 4 embedding tables");
      if (cfg->n_bot != 3 || e->bot.ln[0] != 3 * D || e->bot.ln[2] != D || e->bot.ln[1] < 1 || e->bot.ln[1] > 64)
        return bail(DRS_ERR_UNSUPPORTED, "DIN attention unit must be 3*D -> h -> D with 1 <= h <= 64");
      // the two-launch attention kernel (the only path for sls_exact = 1 and for shapes the fused launch is
      // not instantiated for) keeps 4 samples x (T - 3) units x h hidden values in 64 KB of LDS
      if ((int64_t)(T - 3) * e->bot.ln[1] > 4096)
        return bail(DRS_ERR_UNSUPPORTED, "DIN: (num_tables - 3) * hidden width must not exceed 4096");
      e->m_den = 0; e->w0 = 0;
      e->num_int = 4 * D;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->att.resize(T - 3);
      for (auto& au : e->att) { au.ln = e->bot.ln; au.layers.resize(2); au.sigmoid_layer = -1; }
      e->bot.ln = {0}; e->bot.layers.clear();      // no bottom MLP of its own
      e->top.sigmoid_layer = -1;
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_DIEN: {
      if (T < 4) return bail(DRS_ERR_BAD_ARG, "DIEN needs at least 4 embedding tables");
      if (cfg->n_bot != 2 || e->bot.ln[0] != D || !dien_applicable(D, e->bot.ln[1]))
        return bail(DRS_ERR_UNSUPPORTED, "DIEN: ln_bot must be [D, hidden_size], D in {16, 32, 64}, hidden_size in {8, 16, 32, 64}");
      const int H = e->bot.ln[1];
      e->m_den = 0; e->w0 = 0;
      e->num_int = H + 3 * D;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->rnn.resize(2);
      e->rnn[0].ln = {D, H, H};
      e->rnn[1].ln = {H, H, H};
      for (auto& rn : e->rnn) { rn.layers.resize(2); rn.sigmoid_layer = -1; }
      e->bot.ln = {0}; e->bot.layers.clear();
      e->top.sigmoid_layer = -1;
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_NCF: {
      if (T != 4) return bail(DRS_ERR_BAD_ARG, "NCF has 4 embedding tables");
      if (e->top.ln.front() != 2 * D) return bail(DRS_ERR_BAD_ARG, "NCF MLP branch input must be 2*D");
      e->m_den = 0; e->w0 = 0;
      e->num_int = D + e->top.ln.back();
      e->top.sigmoid_layer = -1;
      e->fin.ln = {e->num_int, 0};  // output width arrives with drs_set_fc(DRS_MLP_FINAL)
      e->fin.layers.resize(1);
      e->n_out = 0;
      break;
    }
    default:
      return bail(DRS_ERR_BAD_ARG, "unknown model kind");
  }
  for (int t = 0; t < T; ++t) {
    if (e->rows[t] <= 0) return bail(DRS_ERR_BAD_ARG, "table with no rows");
    // row offsets travel as 32-bit counts of load-width units (8 or 16 bytes): 32 GiB per table
    if (e->rows[t] * (int64_t)D >= (1ll << 33)) return bail(DRS_ERR_UNSUPPORTED, "rows*D must be < 2^33 per table");
  }

  // prefix sums and bag * length products are int32 on the device
  if ((int64_t)cfg->max_batch * cfg->max_lookups >= (1ll << 31) / DRS_MAX_COALESCE)
    return bail(DRS_ERR_UNSUPPORTED, "max_batch * max_lookups must stay below 2^31 / 8");
  if (set_device(e)) return bail(DRS_ERR_HIP, e->err.c_str());
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) == hipSuccess && khz > 0)
      e->wall_clock_khz = khz;
  }
  // table arena
  int64_t off = 0;
  e->tab_off.resize(T);
  for (int t = 0; t < T; ++t) {
    e->tab_off[t] = off;
    off += round_up(e->rows[t] * D, 64);  // 256-B aligned
  }
  e->table_set.assign(T, false);
  hipError_t last_rr = hipSuccess;
  auto hip_ok = [&](hipError_t rr) { last_rr = rr; if (rr != hipSuccess) { e->err = hipGetErrorString(rr); return false; } return true; };
  // only an allocation failure is DRS_ERR_OOM; stream/event creation, bad device ... are DRS_ERR_HIP
#define CREATE_TRY(call) if (!hip_ok(call)) return bail(last_rr == hipErrorOutOfMemory ? DRS_ERR_OOM : DRS_ERR_HIP, (std::string(#call ": ") + e->err).c_str())
  e->tune.device = device_id;
  CREATE_TRY(device_init(device_id, &e->tune.zero));
  e->tables_bytes = sizeof(float) * (size_t)off;
  CREATE_TRY(hipMalloc(&e->d_tab_off, sizeof(int64_t) * T));
  CREATE_TRY(hipMalloc(&e->d_tab_rows, sizeof(int64_t) * T));
  CREATE_TRY(hipMalloc(&e->d_op_tab, sizeof(int64_t) * 2));
  CREATE_TRY(hipMemcpy(e->d_tab_off, e->tab_off.data(), sizeof(int64_t) * T, hipMemcpyHostToDevice));
  CREATE_TRY(hipMemcpy(e->d_tab_rows, e->rows.data(), sizeof(int64_t) * T, hipMemcpyHostToDevice));

  e->cap = (int64_t)e->max_batch * e->max_lookups;
  e->max_rows = (int64_t)DRS_MAX_COALESCE * ((e->max_batch + 63) / 64 * 64);
  e->ldT = e->kind == DRS_MODEL_NCF ? 4 * D : e->w0 + (int64_t)T * D;   // (DIN: w0 == 0)
  e->ldR = round_up(e->num_int, 4);
  int maxw = 4;
  for (int w : e->bot.ln) maxw = w > maxw ? w : maxw;
  for (int w : e->top.ln) maxw = w > maxw ? w : maxw;
  for (auto& tk : e->tasks) for (int w : tk.ln) maxw = w > maxw ? w : maxw;
  e->ldH = round_up(maxw, 4);
  e->batches.resize(e->n_batches);
  for (auto& b : e->batches)
    if (alloc_batch(e, b)) return bail(DRS_ERR_OOM, e->err.c_str());
  e->slots.resize(e->n_slots);
  const int n_out_cap = e->kind == DRS_MODEL_NCF ? 1024 : e->n_out;
  for (auto& s : e->slots) {
    CREATE_TRY(hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc(&s.T, sizeof(float) * (size_t)e->max_rows * e->ldT));
    CREATE_TRY(hipMemset(s.T, 0, sizeof(float) * (size_t)e->max_rows * e->ldT));
    CREATE_TRY(hipMalloc(&s.R, sizeof(float) * (size_t)e->max_rows * e->ldR));
    CREATE_TRY(hipMemset(s.R, 0, sizeof(float) * (size_t)e->max_rows * e->ldR));
    CREATE_TRY(hipMalloc(&s.H, sizeof(float) * (size_t)e->max_rows * e->ldH));
    CREATE_TRY(hipMalloc(&s.Hb, sizeof(float) * (size_t)e->max_rows * e->ldH));
    CREATE_TRY(hipMalloc(&s.H2, sizeof(float) * (size_t)e->max_rows * (e->num_int + 4)));
    if (e->kind == DRS_MODEL_MTWND) CREATE_TRY(hipMalloc(&s.H3, sizeof(float) * (size_t)e->max_rows * e->ldH));
    const size_t out_words = kOutOffset + (size_t)e->max_rows * n_out_cap;
    CREATE_TRY(hipMalloc(&s.d_out, sizeof(float) * out_words));
    CREATE_TRY(hipMalloc(&s.d_err, sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&s.d_counter, sizeof(uint32_t)));
    CREATE_TRY(hipStreamCreateWithFlags(&s.early_stream, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc(&s.d_gflag, sizeof(uint32_t)));
    CREATE_TRY(hipMemset(s.d_gflag, 0, sizeof(uint32_t)));
    CREATE_TRY(hipMemset(s.d_err, 0, sizeof(uint32_t)));
    CREATE_TRY(hipMemset(s.d_counter, 0, sizeof(uint32_t)));
    // column-split MLP launches (mlp.hip NSplit; DLRM's first top layer): up to 4 096 rows of that layer's outputs
    if (e->kind == DRS_MODEL_DLRM && e->top.ln.size() >= 3 && e->top.ln[1] >= 128 && e->top.ln[1] <= 1024 && !(e->top.ln[1] & 63)) {
      s.xrows = e->max_rows < 4096 ? e->max_rows : 4096;
      s.xcols = e->top.ln[1];
      CREATE_TRY(hipMalloc(&s.xbuf, sizeof(float) * (size_t)s.xrows * s.xcols));
      CREATE_TRY(hipMalloc(&s.xcnt, sizeof(uint32_t) * (size_t)(s.xrows / 16 + 1)));
      CREATE_TRY(hipMemset(s.xcnt, 0, sizeof(uint32_t) * (size_t)(s.xrows / 16 + 1)));
    }
    // coherent (fine-grained) pinned memory: device stores become visible to a polling CPU
    CREATE_TRY(hipHostMalloc(&s.h_out, sizeof(uint32_t) * out_words, hipHostMallocMapped | hipHostMallocCoherent));
    memset(s.h_out, 0, sizeof(uint32_t) * out_words);
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.dm_out), s.h_out, 0));
    CREATE_TRY(hipMalloc(&s.d_ts, sizeof(uint64_t) * 2 * ((size_t)e->max_rows * T + 8)));   // + 8: the XCD-ordered grid is rounded up to 8
    CREATE_TRY(hipMalloc(&s.d_span_acc, sizeof(uint64_t) * 2 * 65536));
    CREATE_TRY(hipHostMalloc(&s.h_span, sizeof(uint64_t) * 2, hipHostMallocMapped | hipHostMallocCoherent));
    s.h_span[0] = s.h_span[1] = 0;
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.dm_span), s.h_span, 0));
    s.h_ts.resize(2 * ((size_t)e->max_rows * T + 8));
    for (auto& ev : s.ev) CREATE_TRY(hipEventCreate(&ev));
    // Cross-stream ordering on ONE device only (no host reader): the kernels' own agent-scope
    // release/acquire at their boundaries carries the data; the system-scope fence an event
    // record adds by default costs ~3 us between consecutive gathers (measured: 128 k -> 132 k QPS)
    CREATE_TRY(hipEventCreateWithFlags(&s.ev_sls, hipEventDisableTiming | hipEventDisableSystemFence));
    CREATE_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming | hipEventDisableSystemFence));
    for (auto& ev : s.ev_k) CREATE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    if (alloc_batch(e, s.scratch)) return bail(DRS_ERR_OOM, e->err.c_str());
    s.scratch.n_samples = 0;
    s.h_stage_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1) +
                      sizeof(int32_t) * (size_t)T * e->cap +
                      sizeof(int32_t) * (size_t)T * (e->max_batch + 1);
    CREATE_TRY(hipHostMalloc(&s.h_stage, s.h_stage_bytes, hipHostMallocMapped));
    {
      // device view of the same block, laid out like a staged batch: [dense | idx | off]
      char* dm = nullptr;
      CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&dm), s.h_stage, 0));
      const size_t dense_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1);
      const size_t idx_bytes = sizeof(int32_t) * (size_t)T * e->cap;
      s.zc.dense = reinterpret_cast<float*>(dm);
      s.zc.idx = reinterpret_cast<int32_t*>(dm + dense_bytes);
      s.zc.off = reinterpret_cast<int32_t*>(dm + dense_bytes + idx_bytes);
      s.zc.h_off.assign((size_t)T * (e->max_batch + 1), 0);
      // the same layout once more in HBM: target of the one-copy input path
      CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&s.d_stage), s.h_stage_bytes));
      s.dc.dense = reinterpret_cast<float*>(s.d_stage);
      s.dc.idx = reinterpret_cast<int32_t*>(s.d_stage + dense_bytes);
      s.dc.off = reinterpret_cast<int32_t*>(s.d_stage + dense_bytes + idx_bytes);
      s.dc.h_off.assign((size_t)T * (e->max_batch + 1), 0);
    }
  }
  CREATE_TRY(hipStreamCreateWithFlags(&e->stream_g, hipStreamNonBlocking));
  choose_launch_forms(e);
  apply_stream_mode(e);
  {
    // The table arena, last: one hipMalloc (where it lands in HBM, and what DLRM_Net.tune_table_placement does about
    // it: DESIGN.md 5).
#ifdef DRS_LAB
    // lab build, DRS_TABLE_SELECT=1: "table_alloc" 3 (arena_alloc_selected: the fastest gigabytes of a pool by a one-table
    // run of the model's gather kernel) -- an experiment that did NOT work (profiles/r05_placement/README.md)
    const char* env = getenv("DRS_TABLE_SELECT");
    if (env && atoi(env) != 0 && e->kind == DRS_MODEL_DLRM && e->mlp_streams <= 2 && e->max_lookups >= 8 && e->tables_bytes >= ((size_t)1 << 30)) e->table_alloc = 3;
#endif
    Arena first;
    CREATE_TRY(arena_alloc(e, e->tables_bytes, &first));
    e->tables = first.p;
    e->arenas.assign(1, first);
  }
#undef CREATE_TRY
  *out = e;
  return DRS_OK;
}

int32_t drs_destroy(drs_handle e) {
  if (!e) return DRS_OK;
  const bool trc = getenv("DRS_TRACE_DESTROY") != nullptr;
#define DTR(x) do { if (trc) { fprintf(stderr, "destroy %p: %s\n", (void*)e, x); fflush(stderr); } } while (0)
  DTR("begin");
  e->launcher.reset();           // (finishes the jobs it holds, then joins)
  DTR("launcher gone");
  (void)hipSetDevice(e->device);
  if (e->stream_g) { (void)hipStreamSynchronize(e->stream_g); (void)hipStreamDestroy(e->stream_g); }
#ifdef DRS_LAB
  if (e->stream_g2) { (void)hipStreamSynchronize(e->stream_g2); (void)hipStreamDestroy(e->stream_g2); }
#endif
  if (e->stream_h2d) { (void)hipStreamSynchronize(e->stream_h2d); (void)hipStreamDestroy(e->stream_h2d); }
  DTR("g and h2d streams gone");
  for (auto& s : e->slots) {
    DTR("slot");
    if (s.h_multi) (void)hipHostFree(s.h_multi);
    if (s.d_multi) (void)hipFree(s.d_multi);
    s.mq.clear();
    DTR("multi freed");
    if (s.own_stream) { (void)hipStreamSynchronize(s.own_stream); (void)hipStreamDestroy(s.own_stream); }
    if (s.early_stream) { (void)hipStreamSynchronize(s.early_stream); (void)hipStreamDestroy(s.early_stream); }
    if (s.d_gflag) (void)hipFree(s.d_gflag);
    DTR("own stream gone");
    if (s.ev_sls) (void)hipEventDestroy(s.ev_sls);
    if (s.ev_in) (void)hipEventDestroy(s.ev_in);
    for (auto& ev : s.ev_k) if (ev) (void)hipEventDestroy(ev);
    if (s.T) (void)hipFree(s.T);
    if (s.R) (void)hipFree(s.R);
    if (s.H) (void)hipFree(s.H);
    if (s.Hb) (void)hipFree(s.Hb);
    if (s.H2) (void)hipFree(s.H2);
    if (s.H3) (void)hipFree(s.H3);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_err) (void)hipFree(s.d_err);
    if (s.d_ts) (void)hipFree(s.d_ts);
    if (s.h_span) (void)hipHostFree(s.h_span);
    if (s.d_span_acc) (void)hipFree(s.d_span_acc);
    if (s.d_counter) (void)hipFree(s.d_counter);
    if (s.xbuf) (void)hipFree(s.xbuf);
    if (s.xcnt) (void)hipFree(s.xcnt);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.d_stage) (void)hipFree(s.d_stage);
    s.dc = Batch();
    for (auto& ev : s.ev) if (ev) (void)hipEventDestroy(ev);
    free_batch(s.scratch);
  }
  DTR("slots freed");
  for (auto& b : e->batches) free_batch(b);
  for (Mlp* m : {&e->bot, &e->top, &e->fin})
    for (auto& l : m->layers) { l.W = l.b = nullptr; }
  e->tasks.clear();
  e->att.clear();
  e->rnn.clear();
  if (e->d_att) (void)hipFree(e->d_att);
  if (e->d_att_packed) (void)hipFree(e->d_att_packed);
  if (e->w_arena) (void)hipFree(e->w_arena);
  for (Arena& a : e->arenas) arena_free(a);
  for (auto& h : e->spacers) (void)hipMemRelease(h);
  e->tables = nullptr;
  if (e->probe_idx) (void)hipFree(e->probe_idx);
  if (e->probe_out) (void)hipFree(e->probe_out);
  if (e->probe_tab) (void)hipFree(e->probe_tab);
  if (e->probe_err) (void)hipFree(e->probe_err);
  if (e->d_tab_off) (void)hipFree(e->d_tab_off);
  if (e->d_tab_rows) (void)hipFree(e->d_tab_rows);
  if (e->d_op_tab) (void)hipFree(e->d_op_tab);
  DTR("device memory freed");
  delete e;
  if (trc) { fprintf(stderr, "destroy: done\n"); fflush(stderr); }
#undef DTR
  return DRS_OK;
}

// new table contents make the other placement candidates stale: only the arena in use survives
static void drop_spacers(drs_engine* e);
static void drop_other_placements(drs_engine* e) {
  drop_spacers(e);
  if (e->arenas.size() <= 1) return;
  std::vector<Arena> keep;
  for (Arena& a : e->arenas) { if (a.p == e->tables) keep.push_back(a); else arena_free(a); }
  e->arenas = keep;
}
static void drop_spacers(drs_engine* e) {
  for (auto& h : e->spacers) (void)hipMemRelease(h);
  e->spacers.clear();
}

int32_t drs_set_table(drs_handle e, int32_t t, const float* h_W, int64_t rows) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (t < 0 || t >= e->T || !h_W) return fail(e, DRS_ERR_BAD_ARG, "bad table id / null data");
  if (rows != e->rows[t]) return fail(e, DRS_ERR_BAD_ARG, "table %d has %lld rows, got %lld", t, (long long)e->rows[t], (long long)rows);
  if ((rc = drs_sync(e))) return rc;
  drop_other_placements(e);
  HIP_TRY(e, hipMemcpy(e->tables + e->tab_off[t], h_W, sizeof(float) * (size_t)rows * e->D, hipMemcpyHostToDevice));
  e->table_set[t] = true;
  return DRS_OK;
}

int32_t drs_fill_table_uniform(drs_handle e, int32_t t, float lo, float hi, uint64_t seed) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (t < 0 || t >= e->T) return fail(e, DRS_ERR_BAD_ARG, "bad table id");
  if (e->arenas.size() > 1) { if ((rc = drs_sync(e))) return rc; drop_other_placements(e); }
  HIP_TRY(e, launch_fill_uniform(e->tables + e->tab_off[t], e->rows[t] * e->D, t, lo, hi, seed, e->slots[0].stream));
  HIP_TRY(e, hipStreamSynchronize(e->slots[0].stream));
  e->table_set[t] = true;
  return DRS_OK;
}

int32_t drs_set_fc(drs_handle e, int32_t mlp, int32_t layer, const float* h_W, const float* h_b,
                   int32_t m, int32_t n) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!h_W || !h_b) return fail(e, DRS_ERR_BAD_ARG, "null weights");
  Mlp* M = mlp == DRS_MLP_BOT ? &e->bot : mlp == DRS_MLP_TOP ? &e->top : mlp == DRS_MLP_FINAL ? &e->fin : nullptr;
  if (mlp >= DRS_MLP_TASK0 && mlp - DRS_MLP_TASK0 < (int)e->tasks.size()) M = &e->tasks[mlp - DRS_MLP_TASK0];
  if (mlp >= DRS_MLP_ATT0 && mlp - DRS_MLP_ATT0 < (int)e->att.size()) { M = &e->att[mlp - DRS_MLP_ATT0]; e->att_dirty = true; }
  if ((mlp == DRS_MLP_RNN0 || mlp == DRS_MLP_RNN1) && e->rnn.size() == 2) { M = &e->rnn[mlp - DRS_MLP_RNN0]; e->att_dirty = true; }
  if (!M || layer < 0 || layer >= (int)M->layers.size()) return fail(e, DRS_ERR_BAD_ARG, "no such layer");
  if (mlp == DRS_MLP_FINAL && M->ln[1] == 0) {
    if (m <= 0 || m > 1024) return fail(e, DRS_ERR_BAD_ARG, "bad predictor width");
    M->ln[1] = m;
    e->n_out = m;
  }
  if (n != M->ln[layer] || m != M->ln[layer + 1])
    return fail(e, DRS_ERR_BAD_ARG, "layer %d expects W[%d,%d], got [%d,%d]", layer, M->ln[layer + 1], M->ln[layer], m, n);
  Layer& L = M->layers[layer];
  if (!e->w_arena) {
    // ONE allocation for every FC layer of the model, laid out up front:
    //   [64 zeros | all biases, back to back in layer order, each padded to 4 floats (a fused MLP
    //    launch pulls every bias it needs into LDS with one flat copy) |
    //    per layer of the bottom / top / final / task MLPs: W [N, K] row-major, then its PACKED twin
    //    (stream_packed_floats(K, N): the same weights in MFMA-operand order, mlp.hip) |
    //    per layer of the attention units / recurrent layers: W only ]
    // (the final predictor's width is known only when it is set: sized for 1024)
    std::vector<Mlp*> packed = {&e->bot, &e->top, &e->fin}, plain;
    for (auto& tk : e->tasks) packed.push_back(&tk);
    for (auto& au : e->att) plain.push_back(&au);
    for (auto& rn : e->rnn) plain.push_back(&rn);
    auto width = [](const Mlp* mm, size_t i) { return mm->ln[i] > 0 ? (size_t)mm->ln[i] : (size_t)1024; };
    auto wsz = [](size_t k, size_t n) { return (k * n + 63) / 64 * 64; };
    size_t need = 0, nbias = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (Mlp* mm : pass == 0 ? packed : plain)
        for (size_t i = 0; i + 1 < mm->ln.size(); ++i) {
          const size_t k = width(mm, i), n = width(mm, i + 1);
          need += wsz(k, n) + (pass == 0 ? (size_t)stream_packed_floats((int)k, (int)n) : 0);
          nbias += (n + 3) / 4 * 4;
        }
    nbias = (nbias + 63) / 64 * 64;
    const size_t zeros = 64;       // a zero page inside the arena (stream kernel: k beyond a layer's K)
    need += nbias + zeros;
    e->w_arena_floats = need < (1u << 20) ? (1u << 20) : need;   // >= 4 MiB
    HIP_TRY(e, hipMalloc(&e->w_arena, sizeof(float) * e->w_arena_floats));
    HIP_TRY(e, hipMemset(e->w_arena, 0, sizeof(float) * zeros));
    e->tune.w_arena = e->w_arena; e->tune.w_arena_floats = e->w_arena_floats; e->tune.w_zero_off = 0;
    size_t boff = zeros, woff = zeros + nbias;
    e->tune.w_packed_lo = woff;
    for (int pass = 0; pass < 2; ++pass) {
      for (Mlp* mm : pass == 0 ? packed : plain)
        for (size_t i = 0; i + 1 < mm->ln.size(); ++i) {
          const size_t k = width(mm, i), n = width(mm, i + 1);
          mm->layers[i].b = e->w_arena + boff;
          boff += (n + 3) / 4 * 4;
          mm->layers[i].W = e->w_arena + woff;
          mm->layers[i].packed = pass == 0;
          woff += wsz(k, n) + (pass == 0 ? (size_t)stream_packed_floats((int)k, (int)n) : 0);
        }
      if (pass == 0) e->tune.w_packed_hi = woff;
    }
    e->w_arena_used = woff;
  }
  HIP_TRY(e, hipMemcpy(L.W, h_W, sizeof(float) * (size_t)m * n, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(L.b, h_b, sizeof(float) * (size_t)m, hipMemcpyHostToDevice));
  if (L.packed) {
    // the MFMA-operand-order twin sits right behind W (at W + roundup64(K N): stream_plan relies on it)
    HIP_TRY(e, launch_pack_stream_weights(L.W, n, m, L.W + ((size_t)m * n + 63) / 64 * 64, nullptr));
    HIP_TRY(e, hipStreamSynchronize(nullptr));
  }
  L.m = m; L.n = n; L.set = true;
  return DRS_OK;
}

static int32_t stage_into(drs_engine* e, Batch& b, int32_t n, const float* h_dense,
                          const int64_t* const* h_idx, const int64_t* n_idx,
                          const int32_t* const* h_len, hipStream_t stream, void* pinned,
                          bool in_place = false) {
  if (n < 0 || n > e->max_batch) return fail(e, DRS_ERR_BAD_ARG, "n_samples=%d exceeds max_batch=%d", n, e->max_batch);
  if (!h_idx || !n_idx || !h_len) return fail(e, DRS_ERR_BAD_ARG, "null index/length arrays");
  if (e->m_den > 0 && !h_dense && n > 0) return fail(e, DRS_ERR_BAD_ARG, "null dense input");
  const size_t dense_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1);
  const size_t idx_bytes = sizeof(int32_t) * (size_t)e->T * e->cap;
  const size_t off_bytes = sizeof(int32_t) * (size_t)e->T * (e->max_batch + 1);
  std::vector<int32_t> tmp_idx, tmp_off;
  int32_t* idx32;
  int32_t* off32;
  float* dense_stage = nullptr;
  if (pinned) {
    dense_stage = reinterpret_cast<float*>(pinned);
    idx32 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(pinned) + dense_bytes);
    off32 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(pinned) + dense_bytes + idx_bytes);
  } else {
    // validate into temporaries: a failure part-way (index range on table 3) must leave a
    // previously staged batch exactly as it was
    tmp_idx.resize((size_t)e->T * e->cap);
    tmp_off.resize((size_t)e->T * (e->max_batch + 1));
    idx32 = tmp_idx.data();
    off32 = tmp_off.data();
  }
  if (pinned && !e->pool) {
    int w = e->host_threads >= 0 ? e->host_threads : (e->T < 7 ? e->T : 7);   // T tables + the dense rows
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && w > hw - 1) w = hw - 1;
    e->pool.reset(new HostPool(w < 0 ? 0 : w));
  }
  // zero-copy path: the dense rows' copy into the pinned block rides along as one more work item
  std::function<void()> copy_dense = [&] { memcpy(dense_stage, h_dense, sizeof(float) * (size_t)n * e->m_den); };
  const bool dense_in_pool = in_place && e->m_den > 0 && n > 0;
  int32_t rc = convert_inputs(e, n, h_idx, n_idx, h_len, idx32, off32, pinned ? e->pool.get() : nullptr,
                              dense_in_pool ? &copy_dense : nullptr);
  if (rc) {
    if (pinned) { b.staged = false; b.n_samples = 0; }   // the slot's pinned block was overwritten: nothing valid in it
    return rc;
  }
  memcpy(b.h_off.data(), off32, off_bytes);
  if (in_place) {
    // `b` aliases the pinned block: the converted indices/offsets are already where the
    // kernels will read them (over PCIe, once); only the dense rows need a host copy
    // (dense rows: copied beside the index conversion above)
  } else {
  // copy only what is used of each table's index row
  for (int t = 0; t < e->T; ++t)
    if (n_idx[t] > 0)
      HIP_TRY(e, hipMemcpyAsync(b.idx + (size_t)t * e->cap, idx32 + (size_t)t * e->cap,
                                sizeof(int32_t) * (size_t)n_idx[t], hipMemcpyHostToDevice, stream));
  HIP_TRY(e, hipMemcpyAsync(b.off, off32, off_bytes, hipMemcpyHostToDevice, stream));
  if (e->m_den > 0 && n > 0) {
    const float* src = h_dense;
    if (pinned) {
      memcpy(dense_stage, h_dense, sizeof(float) * (size_t)n * e->m_den);
      src = dense_stage;
    }
    HIP_TRY(e, hipMemcpyAsync(b.dense, src, sizeof(float) * (size_t)n * e->m_den, hipMemcpyHostToDevice, stream));
  }
  }
  if (!pinned) HIP_TRY(e, hipStreamSynchronize(stream));
  b.n_samples = n;
  b.staged = true;
  b.uniform_len = -1;
  if (n > 0) {
    const int32_t L = h_len[0][0];
    bool same = true;
    for (int t = 0; t < e->T && same; ++t)
      for (int i = 0; i < n; ++i)
        if (h_len[t][i] != L) { same = false; break; }
    if (same) b.uniform_len = L;
  }
  return DRS_OK;
}

int32_t drs_stage_batch(drs_handle e, int32_t batch_id, int32_t n_samples, const float* h_dense,
                        const int64_t* const* h_idx, const int64_t* n_idx,
                        const int32_t* const* h_len) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (batch_id < 0 || batch_id >= e->n_batches) return fail(e, DRS_ERR_BAD_ARG, "batch_id %d of %d", batch_id, e->n_batches);
  // make sure no in-flight query still reads this batch
  for (auto& s : e->slots) if (s.busy) HIP_TRY(e, hipStreamSynchronize(s.stream));
  return stage_into(e, e->batches[batch_id], n_samples, h_dense, h_idx, n_idx, h_len, e->slots[0].stream, nullptr);
}

int32_t drs_forward_async(drs_handle e, int32_t slot, int32_t batch_id, int32_t bs) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (batch_id < 0 || batch_id >= e->n_batches || !e->batches[batch_id].staged)
    return fail(e, DRS_ERR_STATE, "batch %d is not staged", batch_id);
  Slot& s = e->slots[slot];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  const Batch* bt = &e->batches[batch_id];
  return enqueue_forward(e, s, 1, &bt, &bs);
}

int32_t drs_forward_multi_async(drs_handle e, int32_t slot, int32_t n, const int32_t* batch_ids,
                                const int32_t* bs) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (!batch_ids || !bs || n < 1 || n > DRS_MAX_COALESCE) return fail(e, DRS_ERR_BAD_ARG, "1..%d queries per launch", DRS_MAX_COALESCE);
  const Batch* bts[DRS_MAX_COALESCE];
  for (int i = 0; i < n; ++i) {
    if (batch_ids[i] < 0 || batch_ids[i] >= e->n_batches || !e->batches[batch_ids[i]].staged)
      return fail(e, DRS_ERR_STATE, "batch %d is not staged", batch_ids[i]);
    bts[i] = &e->batches[batch_ids[i]];
  }
  Slot& s = e->slots[slot];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  return enqueue_forward(e, s, n, bts, bs);
}

int32_t drs_wait(drs_handle e, int32_t slot, float* h_out, int64_t h_out_floats) {
  int32_t rc = check_handle(e, true);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (h_out && h_out_floats < 0) return fail(e, DRS_ERR_BAD_ARG, "negative output capacity");
  return wait_slot(e, e->slots[slot], h_out, h_out_floats);
}

int32_t drs_forward(drs_handle e, int32_t batch_id, int32_t bs, float* h_out) {
  int32_t rc = drs_forward_async(e, 0, batch_id, bs);
  if (rc) return rc;
  return wait_slot(e, e->slots[0], h_out);
}

int32_t drs_sync(drs_handle e) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  int32_t first = DRS_OK;
  for (auto& s : e->slots) {
    rc = wait_slot(e, s, nullptr);
    if (rc && !first) first = rc;
  }
  return first;
}

int32_t drs_forward_inputs_async(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                                 const int64_t* const* h_idx, const int64_t* n_idx,
                                 const int32_t* const* h_len) {
  int32_t rc = check_handle(e, true);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  Slot& s = e->slots[slot];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  // the copies below must go on the stream the job's MLP side will use
  if (bs >= 0) s.stream = job_stream(e, s, ((int64_t)bs + 63) / 64 * 64);
  // how the converted inputs reach the kernels: 1 = read in place from the pinned block over PCIe
  // (no copy: best for small queries, kernel-issued PCIe reads top out near 20 GB/s), 2 = ONE
  // DMA copy of the packed block into its HBM twin (the copy engine moves it at PCIe rate beside
  // the kernels of the other slots), 3 = 2 when the query carries >= 128 KB, else 1 (default: 1)
  int mode = e->zero_copy_inputs;
  if (mode == 3) {
    int64_t bytes = (int64_t)bs * e->m_den * 4;
    for (int t = 0; t < e->T && n_idx; ++t) bytes += n_idx[t] * 4;
    mode = bytes >= 128 * 1024 ? 2 : 1;
  }
  size_t used = 0;
  bool need_off = false;
  if (mode == 2) {
    if ((rc = stage_into(e, s.dc, bs, h_dense, h_idx, n_idx, h_len, s.stream, s.h_stage, true))) return rc;
    // the used prefix of the block: dense rows, then index rows up to the last table's last index
    const size_t dense_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1);
    used = dense_bytes + sizeof(int32_t) * ((size_t)(e->T - 1) * e->cap + (size_t)n_idx[e->T - 1]);
    need_off = !e->sls_uniform || s.dc.uniform_len < 0;
  } else if (mode == 1) {
    // no H2D copies at all: convert straight into the slot's host-mapped pinned block and let
    // the gather / first MLP layer read it in place (795 KB per RMC1 query, read once)
    if ((rc = stage_into(e, s.zc, bs, h_dense, h_idx, n_idx, h_len, s.stream, s.h_stage, true))) return rc;
  } else {
    if ((rc = stage_into(e, s.scratch, bs, h_dense, h_idx, n_idx, h_len, s.stream, s.h_stage))) return rc;
  }
  // the arrays are consumed; what is left are HIP calls
  if (e->launch_thread && bs > 0 && mode != 0) {
    if (!e->launcher) {          // (created on first use)
      e->launch_state.reset(new std::atomic<int>[e->slots.size()]);
      for (size_t i = 0; i < e->slots.size(); ++i) e->launch_state[i].store(0, std::memory_order_relaxed);
      e->launcher.reset(new Launcher(e));
    }
    s.busy = true;                 // (enqueue_forward sets it too; wait_slot needs it before that ran)
    s.launch_rc = 0;
    e->launcher->push(slot, mode, bs, used, need_off);
    return DRS_OK;
  }
  return finish_inputs(e, s, mode, bs, used, need_off);
}

int32_t drs_run_queues_async(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                             const int64_t* h_ids, int64_t ids_row_stride, int64_t n_idx_per_table,
                             const int32_t* h_lengths, int64_t len_row_stride) {
  if (!e) return fail(nullptr, DRS_ERR_BAD_ARG, "null handle");
  if (!h_ids || !h_lengths || n_idx_per_table < 0 || e->T > 256) return fail(e, DRS_ERR_BAD_ARG, "bad 2-D input arrays");
  const int64_t* ip[256];
  const int32_t* lp[256];
  int64_t ni[256];
  for (int t = 0; t < e->T; ++t) {
    ip[t] = h_ids + (int64_t)t * ids_row_stride;
    lp[t] = h_lengths + (int64_t)t * len_row_stride;
    ni[t] = n_idx_per_table;
  }
  return drs_forward_inputs_async(e, slot, bs, h_dense, ip, ni, lp);
}

// n queries' per-call arrays as ONE launch set: every query is narrowed / ENFORCE-checked into its
// own block of the slot's multi-block pinned allocation, the blocks cross the bus in one DMA copy on
// a copy stream of their own (so the copy of this set runs under the gathers of the sets before it),
// and the set is launched like a coalesced set of staged batches.
int32_t drs_run_queues_multi_async(drs_handle e, int32_t slot, int32_t n, const int32_t* bs,
                                   const float* const* h_dense, const int64_t* const* h_ids,
                                   const int64_t* ids_row_stride, const int64_t* n_idx_per_table,
                                   const int32_t* const* h_lengths, const int64_t* len_row_stride) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (n < 1 || n > DRS_MAX_COALESCE) return fail(e, DRS_ERR_BAD_ARG, "1..%d queries per launch", DRS_MAX_COALESCE);
  if (!bs || !h_dense || !h_ids || !ids_row_stride || !n_idx_per_table || !h_lengths || !len_row_stride || e->T > 256)
    return fail(e, DRS_ERR_BAD_ARG, "bad per-query array tables");
  Slot& s = e->slots[slot];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  const size_t dense_bytes = sizeof(float) * (size_t)e->max_batch * (e->m_den > 0 ? e->m_den : 1);
  const size_t idx_bytes = sizeof(int32_t) * (size_t)e->T * e->cap;
  const size_t off_bytes = sizeof(int32_t) * (size_t)e->T * (e->max_batch + 1);
  if (s.mq.empty()) {
    // first use: both allocations or neither (a failed second one must not leave a half-built slot behind:
    // the next call would index device pointers derived from null -- ADVICE r3)
    s.multi_block = (size_t)round_up((int64_t)(dense_bytes + idx_bytes + off_bytes), 256);
    hipError_t r1 = hipHostMalloc(reinterpret_cast<void**>(&s.h_multi), s.multi_block * DRS_MAX_COALESCE, hipHostMallocDefault);
    hipError_t r2 = r1 == hipSuccess ? hipMalloc(reinterpret_cast<void**>(&s.d_multi), s.multi_block * DRS_MAX_COALESCE) : r1;
    if (r2 != hipSuccess) {
      if (r1 == hipSuccess) (void)hipHostFree(s.h_multi);
      s.h_multi = nullptr; s.d_multi = nullptr; s.multi_block = 0;
      return fail(e, r2 == hipErrorOutOfMemory ? DRS_ERR_OOM : DRS_ERR_HIP, "per-call input blocks of a launch set: %s", hipGetErrorString(r2));
    }
    s.mq.assign(DRS_MAX_COALESCE, Batch());
    for (int i = 0; i < DRS_MAX_COALESCE; ++i) {
      char* d = s.d_multi + (size_t)i * s.multi_block;
      s.mq[i].dense = reinterpret_cast<float*>(d);
      s.mq[i].idx = reinterpret_cast<int32_t*>(d + dense_bytes);
      s.mq[i].off = reinterpret_cast<int32_t*>(d + dense_bytes + idx_bytes);
      s.mq[i].h_off.assign((size_t)e->T * (e->max_batch + 1), 0);
    }
  }
  if (!e->stream_h2d) HIP_TRY(e, hipStreamCreateWithFlags(&e->stream_h2d, hipStreamNonBlocking));
  // host pass: ONE fork-join over the tables (and dense rows) of every query of the set
  for (int i = 0; i < n; ++i) {
    if (bs[i] < 0 || bs[i] > e->max_batch) return fail(e, DRS_ERR_BAD_ARG, "query %d: n_samples=%d exceeds max_batch=%d", i, bs[i], e->max_batch);
    if (!h_ids[i] || !h_lengths[i] || n_idx_per_table[i] < 0) return fail(e, DRS_ERR_BAD_ARG, "bad 2-D input arrays of query %d", i);
    if (e->m_den > 0 && !h_dense[i] && bs[i] > 0) return fail(e, DRS_ERR_BAD_ARG, "query %d: null dense input", i);
  }
  if (!e->pool) {
    int w = e->host_threads >= 0 ? e->host_threads : (e->T < 7 ? e->T : 7);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && w > hw - 1) w = hw - 1;
    e->pool.reset(new HostPool(w < 0 ? 0 : w));
  }
  const int T = e->T, per_q = T + 1;
  std::vector<ConvRes> res((size_t)n * T);
  auto item = [&](int k) {
    const int i = k / per_q, t = k % per_q;
    char* blk = s.h_multi + (size_t)i * s.multi_block;
    if (t == T) {   // the dense rows
      if (e->m_den > 0 && bs[i] > 0) memcpy(blk, h_dense[i], sizeof(float) * (size_t)bs[i] * e->m_den);
      return;
    }
    int32_t* off_t = reinterpret_cast<int32_t*>(blk + dense_bytes + idx_bytes) + (size_t)t * (e->max_batch + 1);
    convert_table(e, bs[i], t, h_ids[i] + (int64_t)t * ids_row_stride[i], n_idx_per_table[i],
                  h_lengths[i] + (int64_t)t * len_row_stride[i],
                  reinterpret_cast<int32_t*>(blk + dense_bytes) + (size_t)t * e->cap, off_t, res[(size_t)i * T + t]);
    memcpy(s.mq[i].h_off.data() + (size_t)t * (e->max_batch + 1), off_t, sizeof(int32_t) * (size_t)(e->max_batch + 1));
  };
  e->pool->run(n * per_q, item);
  const Batch* bts[DRS_MAX_COALESCE];
  int64_t Mv = 0;
  size_t used_sum = 0, used[DRS_MAX_COALESCE];
  bool need_off = false;
  for (int i = 0; i < n; ++i) {
    Batch& b = s.mq[i];
    int64_t ni[256];
    for (int t = 0; t < T; ++t) ni[t] = n_idx_per_table[i];
    char who[32];
    snprintf(who, sizeof who, "query %d: ", i);
    if ((rc = convert_report(e, res.data() + (size_t)i * T, ni, who))) {
      for (int k = 0; k < n; ++k) { s.mq[k].staged = false; s.mq[k].n_samples = 0; }
      return rc;
    }
    b.n_samples = bs[i];
    b.staged = true;
    b.uniform_len = -1;
    if (bs[i] > 0) {
      const int32_t L0 = h_lengths[i][0];
      bool same = true;
      for (int t = 0; t < T && same; ++t) same = res[(size_t)i * T + t].same && h_lengths[i][(int64_t)t * len_row_stride[i]] == L0;
      if (same) b.uniform_len = L0;
    }
    bts[i] = &b;
    Mv += ((int64_t)bs[i] + 63) / 64 * 64;
    used[i] = dense_bytes + sizeof(int32_t) * ((size_t)(T - 1) * e->cap + (size_t)n_idx_per_table[i]);
    used_sum += used[i];
    need_off = need_off || !e->sls_uniform || b.uniform_len < 0;
  }
  if (Mv > e->max_rows) return fail(e, DRS_ERR_BAD_ARG, "%lld coalesced rows exceed the slot capacity %lld", (long long)Mv, (long long)e->max_rows);
  // one copy of the n blocks when they are mostly full; else the used prefix of each block and, where the
  // kernels will read prefix sums (ragged bags, or "sls_uniform" 0), that block's offsets region as a
  // second copy -- a set of small ragged queries must not move n full-capacity blocks
  if (2 * used_sum >= (size_t)n * s.multi_block) {
    const size_t bytes = need_off ? (size_t)n * s.multi_block : (size_t)(n - 1) * s.multi_block + used[n - 1];
    HIP_TRY(e, hipMemcpyAsync(s.d_multi, s.h_multi, bytes, hipMemcpyHostToDevice, e->stream_h2d));
  } else {
    for (int i = 0; i < n; ++i) {
      const size_t base = (size_t)i * s.multi_block;
      HIP_TRY(e, hipMemcpyAsync(s.d_multi + base, s.h_multi + base, used[i], hipMemcpyHostToDevice, e->stream_h2d));
      if (!e->sls_uniform || s.mq[i].uniform_len < 0)
        HIP_TRY(e, hipMemcpyAsync(s.d_multi + base + dense_bytes + idx_bytes, s.h_multi + base + dense_bytes + idx_bytes,
                                  off_bytes, hipMemcpyHostToDevice, e->stream_h2d));
    }
  }
  HIP_TRY(e, hipEventRecord(s.ev_in, e->stream_h2d));
  const hipStream_t ms = job_stream(e, s, Mv), gs = job_gather_stream(e, s, Mv);
  HIP_TRY(e, hipStreamWaitEvent(gs, s.ev_in, 0));
  if (ms != gs) HIP_TRY(e, hipStreamWaitEvent(ms, s.ev_in, 0));   // the MLP side reads the dense rows
  return enqueue_forward(e, s, n, bts, bs);
}

int32_t drs_forward_inputs(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                           const int64_t* const* h_idx, const int64_t* n_idx,
                           const int32_t* const* h_len, float* h_out) {
  int32_t rc = drs_forward_inputs_async(e, slot, bs, h_dense, h_idx, n_idx, h_len);
  if (rc) return rc;
  return wait_slot(e, e->slots[slot], h_out);
}

int32_t drs_fetch_interaction(drs_handle e, int32_t slot, int32_t bs, float* h_R) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  // rows are VIRTUAL rows of the slot: coalesced query i sits at the 64-row aligned offset
  // sum of round_up(bs_j, 64) over j < i; a single query starts at row 0
  if (slot < 0 || slot >= e->n_slots || !h_R || bs < 0 || bs > e->max_rows) return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  Slot& s = e->slots[slot];
  HIP_TRY(e, hipStreamSynchronize(s.stream));
  if (e->mlp_layout == 1) HIP_TRY(e, hipStreamSynchronize(e->stream_g));
  const float* src;
  int64_t ld;
  if (e->kind == DRS_MODEL_NCF) { src = s.H2; ld = e->num_int; }
  else if (e->kind == DRS_MODEL_DIN || e->kind == DRS_MODEL_DIEN) { src = s.R; ld = e->ldR; }
  else if (e->kind == DRS_MODEL_DLRM && e->interaction_op == DRS_INTERACT_DOT) { src = s.R; ld = e->ldR; }
  else {
    if (s.split_last)
      return fail(e, DRS_ERR_STATE, "the set's dense columns were read in place (\"gemm_split\" 1): set it to 0 to materialise the interaction tensor");
    src = s.T; ld = e->ldT;
  }
  HIP_TRY(e, hipMemcpy2D(h_R, sizeof(float) * e->num_int, src, sizeof(float) * ld,
                         sizeof(float) * e->num_int, bs, hipMemcpyDeviceToHost));
  return DRS_OK;
}

int32_t drs_out_width(drs_handle e, int32_t* n_out) {
  if (!e || !n_out) return DRS_ERR_BAD_ARG;
  *n_out = e->n_out;
  return DRS_OK;
}

int32_t drs_interaction_width(drs_handle e, int32_t* num_int) {
  if (!e || !num_int) return DRS_ERR_BAD_ARG;
  *num_int = e->num_int;
  return DRS_OK;
}

// ---- operator-level entry points ---------------------------------------------
int32_t drs_sls(drs_handle e, const float* d_W, int64_t rows, int32_t D, const int32_t* d_idx,
                const int32_t* d_len, int64_t n_bags, int64_t n_idx, float* d_out, int32_t exact_order) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!d_W || !d_len || !d_out || (!d_idx && n_idx > 0) || n_bags < 0 || n_idx < 0 || rows <= 0)
    return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  if (D <= 0 || D > 256 || (D & 3)) return fail(e, DRS_ERR_UNSUPPORTED, "D=%d must be a multiple of 4 in [4,256]", D);
  if (rows * (int64_t)D >= (1ll << 33) || n_bags >= (1ll << 31) || n_idx >= (1ll << 31))
    return fail(e, DRS_ERR_UNSUPPORTED, "operand too large");
  if (n_bags == 0) return n_idx == 0 ? DRS_OK : fail(e, DRS_ERR_LENGTHS_SUM, "indices without bags");
  Slot& s = e->slots[0];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  std::vector<int32_t> len((size_t)n_bags), off((size_t)n_bags + 1);
  HIP_TRY(e, hipMemcpy(len.data(), d_len, sizeof(int32_t) * (size_t)n_bags, hipMemcpyDeviceToHost));
  int64_t total = 0;
  off[0] = 0;
  for (int64_t b = 0; b < n_bags; ++b) {
    if (len[b] < 0) return fail(e, DRS_ERR_LENGTHS_SUM, "negative length");
    total += len[b];
    if (total > n_idx) return fail(e, DRS_ERR_LENGTHS_SUM, "sum(lengths) exceeds len(indices)");
    off[b + 1] = (int32_t)total;
  }
  if (total != n_idx) return fail(e, DRS_ERR_LENGTHS_SUM, "sum(lengths)=%lld != len(indices)=%lld", (long long)total, (long long)n_idx);
  int32_t* d_off = nullptr;
  int32_t* d_err = nullptr;
  HIP_TRY(e, hipMalloc(&d_off, sizeof(int32_t) * ((size_t)n_bags + 1)));
  hipError_t r = hipMalloc(&d_err, sizeof(int32_t));
  if (r != hipSuccess) { (void)hipFree(d_off); return fail(e, DRS_ERR_OOM, "hipMalloc"); }
  const int64_t tab[2] = {0, rows};
  int32_t h_err = 0;
  r = hipMemcpy(d_off, off.data(), sizeof(int32_t) * ((size_t)n_bags + 1), hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipMemcpy(e->d_op_tab, tab, sizeof tab, hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipMemset(d_err, 0, sizeof(int32_t));
  if (r == hipSuccess) {
    SlsArgs a;
    memset(&a, 0, sizeof a);
    a.tables = d_W; a.tab_off = e->d_op_tab; a.tab_rows = e->d_op_tab + 1;
    a.q.n_q = 1; a.q.vstart[1] = (int32_t)n_bags; a.q.cum[1] = (int32_t)n_bags; a.q.bs[0] = (int32_t)n_bags;
    a.idx[0] = d_idx; a.off[0] = d_off; a.uniform_len[0] = -1;
    a.out = d_out; a.ld_out = D; a.col0 = 0; a.T = 1; a.D = D; a.err = d_err; a.ts = nullptr;
    r = launch_sls(a, exact_order, e->tune, s.stream);
  }
  if (r == hipSuccess) r = hipStreamSynchronize(s.stream);
  if (r == hipSuccess) r = hipMemcpy(&h_err, d_err, sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_off);
  (void)hipFree(d_err);
  if (r != hipSuccess) return fail(e, DRS_ERR_HIP, "drs_sls: %s", hipGetErrorString(r));
  if (h_err) return fail(e, DRS_ERR_INDEX_RANGE, "an index is outside [0, %lld)", (long long)rows);
  return DRS_OK;
}

int32_t drs_fc(drs_handle e, const float* d_x, int64_t M, int32_t K, const float* d_W, const float* d_b,
               int32_t N, int32_t act, float* d_y) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!d_x || !d_W || !d_y || M < 0 || K <= 0 || N <= 0 || act < 0 || act > 2) return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  Slot& s = e->slots[0];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  HIP_TRY(e, launch_fc(d_x, K, M, K, d_W, d_b, N, act, d_y, N, e->tune, s.stream));
  HIP_TRY(e, hipStreamSynchronize(s.stream));
  return DRS_OK;
}

int32_t drs_interact_dot(drs_handle e, const float* d_T, int64_t B, int32_t F, int32_t D, int32_t itself, float* d_R) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!d_T || !d_R || B < 0 || F <= 0 || D <= 0) return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  Slot& s = e->slots[0];
  if (s.busy && (rc = wait_slot(e, s, nullptr))) return rc;
  const int P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
  hipError_t r = launch_interact_dot(d_T, (int64_t)F * D, B, F, D, itself, d_R, D + P, s.stream);
  if (r == hipErrorInvalidValue) return fail(e, DRS_ERR_UNSUPPORTED, "F=%d D=%d does not fit LDS", F, D);
  HIP_TRY(e, r);
  HIP_TRY(e, hipStreamSynchronize(s.stream));
  return DRS_OK;
}

// ---- tuning / measurement ------------------------------------------------------
int32_t drs_set_option(drs_handle e, const char* key, int64_t value) {
  if (!e || !key) return DRS_ERR_BAD_ARG;
  if (!strcmp(key, "sls_exact")) e->sls_exact = value ? 1 : 0;
  else if (!strcmp(key, "sls_flat") && value >= 0 && value <= 2) e->tune.sls_flat = (int)value;
  else if (!strcmp(key, "din_fused")) e->din_fused = value ? 1 : 0;
  else if (!strcmp(key, "dien_mfma") && value >= 0 && value <= 2) e->dien_mfma = (int)value;
  else if (!strcmp(key, "dien_fuse_top") && (value == 0 || value == 1)) e->dien_fuse_top = (int)value;
  else if (!strcmp(key, "din_s") && (value == 0 || value == 1 || value == 2 || value == 4)) e->tune.din_s = (int)value;
  else if (!strcmp(key, "sls_nt")) e->tune.sls_nt = value ? 1 : 0;
  else if (!strcmp(key, "din_nt")) e->tune.din_nt = value ? 1 : 0;
  else if (!strcmp(key, "din_pipe")) e->tune.din_pipe = value ? 1 : 0;
  else if (!strcmp(key, "gemm_split")) e->gemm_split = value ? 1 : 0;
  else if (!strcmp(key, "sls_bpw") && (value == 0 || value == 1 || value == 2 || value == 4)) e->tune.sls_bpw = (int)value;
  else if (!strcmp(key, "mlp_split")) e->mlp_split = value ? 1 : 0;
  else if (!strcmp(key, "sls_uniform")) e->sls_uniform = value ? 1 : 0;
  else if (!strcmp(key, "shared_stream")) {
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    if (value < 0 || value > 2) return fail(e, DRS_ERR_BAD_ARG, "shared_stream is 0, 1 or 2");
    e->shared_stream = (int)value;
    apply_stream_mode(e);
  }
  else if (!strcmp(key, "mlp_fuse")) e->mlp_fuse = value ? 1 : 0;
  else if (!strcmp(key, "dispatch_log")) e->dispatch_log = value ? 1 : 0;
  else if (!strcmp(key, "zero_copy_inputs") && value >= 0 && value <= 3) e->zero_copy_inputs = (int)value;
  else if (!strcmp(key, "host_threads") && value >= -1 && value <= 64) { e->host_threads = (int)value; e->pool.reset(); }
  else if (!strcmp(key, "mlp_streams") && value >= 1 && value <= 8) {
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    e->mlp_streams = (int)value;
    apply_stream_mode(e);
  }
#ifdef DRS_LAB
  else if (!strcmp(key, "gather_streams") && (value == 1 || value == 2)) {
    // experiment: does a second gather stream close the ~1.4 us between back-to-back gather launches?
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    if (value == 2 && !e->stream_g2) HIP_TRY(e, hipStreamCreateWithFlags(&e->stream_g2, hipStreamNonBlocking));
    e->gather_streams = (int)value;
    apply_stream_mode(e);
  }
  else if (!strcmp(key, "mlp_cu_mask") && value >= 0 && value <= 248) {
    // experiment: the MLP side's streams run on `value` CUs only (bits 0 .. value-1 of the queue's CU
    // mask; 0 = every CU, the default), the gather stream on the others ("gather_cu_complement" 1, default) or everywhere (0)
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    int ncu = 0;
    HIP_TRY(e, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, e->device));
    if (value >= ncu) return fail(e, DRS_ERR_BAD_ARG, "mlp_cu_mask %lld of %d CUs", (long long)value, ncu);
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> mlp((size_t)words, 0u), rest((size_t)words, 0u);
    for (int c = 0; c < ncu; ++c) ((value == 0 || c < value) ? mlp : rest)[(size_t)c / 32] |= 1u << (c % 32);
    if (value == 0 || !e->gather_cu_complement) rest = std::vector<uint32_t>((size_t)words, 0xffffffffu);
    for (auto& s : e->slots) {
      if (s.own_stream) { (void)hipStreamSynchronize(s.own_stream); (void)hipStreamDestroy(s.own_stream); s.own_stream = nullptr; }
      HIP_TRY(e, hipExtStreamCreateWithCUMask(&s.own_stream, (uint32_t)words, mlp.data()));
    }
    if (e->stream_g) { (void)hipStreamSynchronize(e->stream_g); (void)hipStreamDestroy(e->stream_g); e->stream_g = nullptr; }
    HIP_TRY(e, hipExtStreamCreateWithCUMask(&e->stream_g, (uint32_t)words, rest.data()));
    e->mlp_cu_mask = (int)value;
    apply_stream_mode(e);
  }
  else if (!strcmp(key, "gather_cu_complement") && (value == 0 || value == 1)) e->gather_cu_complement = (int)value;
  else if (!strcmp(key, "gather_priority") && value >= -1 && value <= 1) {
    // experiment: the gather stream at the device's highest (1) or lowest (-1) queue priority, 0 = default
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    int lo = 0, hi = 0;
    HIP_TRY(e, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (numerically: hi <= lo)
    if (e->stream_g) { (void)hipStreamSynchronize(e->stream_g); (void)hipStreamDestroy(e->stream_g); e->stream_g = nullptr; }
    HIP_TRY(e, hipStreamCreateWithPriority(&e->stream_g, hipStreamNonBlocking, value > 0 ? hi : value < 0 ? lo : (lo + hi) / 2));
    e->gather_priority = (int)value;
    apply_stream_mode(e);
  }
#endif  // DRS_LAB
  else if (!strcmp(key, "mlp_layout") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_layout = (int)value; }
  else if (!strcmp(key, "sls_short_bag") && value >= -1 && value <= 1 << 20) e->sls_short_bag = (int)value;
  else if (!strcmp(key, "mlp_wide_kn") && value > 0) e->mlp_wide_kn = value;
  else if (!strcmp(key, "mlp_fuse_rows") && value >= 0) e->mlp_fuse_rows = value;
  else if (!strcmp(key, "mlp_early") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_early = (int)value; }
  else if (!strcmp(key, "small_piped") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->small_piped = (int)value; }
  else if (!strcmp(key, "mlp_small_rows") && value >= 0) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_small_rows = value; }
  else if (!strcmp(key, "mlp_preload")) e->tune.mlp_preload = value ? 1 : 0;
  else if (!strcmp(key, "mlp_stream") && value >= 0 && value <= 4 && value != 3) e->tune.mlp_stream = (int)value;
  else if (!strcmp(key, "mlp_stream_2cu") && (value == 0 || value == 1)) e->tune.mlp_stream_2cu = (int)value;
  else if (!strcmp(key, "mlp_gemm_2cu") && (value == 0 || value == 1)) e->tune.gemm_2cu = (int)value;
  else if (!strcmp(key, "launch_thread") && (value == 0 || value == 1)) e->launch_thread = (int)value;
  else if (!strcmp(key, "mlp_gemm")) e->tune.mlp_gemm = value ? 1 : 0;
  else if (!strcmp(key, "mlp_gemm_tile") && (value == 0 || value == 22 || value == 12 || value == 21 || value == 11 || value == 214 || value == 322 || value == 321 || value == 312 || value == 311)) e->tune.gemm_tile = (int)value;
  else if (!strcmp(key, "mlp_gemm32") && (value == 0 || value == 1)) e->tune.gemm32 = (int)value;
  else if (!strcmp(key, "mlp_gemm32_small") && (value == 0 || value == 22 || value == 21 || value == 12 || value == 11)) e->tune.gemm32_small = (int)value;
  else if (!strcmp(key, "mlp_gemm32_small_blocks") && value >= 0 && value <= 65536) e->tune.gemm32_small_blocks = (int)value;
  else if (!strcmp(key, "mlp_gemm32_blocks") && value >= 1 && value <= 65536) e->tune.gemm32_blocks = (int)value;
  else if (!strcmp(key, "mlp_debug")) e->tune.mlp_debug = (int)value;
  else if (!strcmp(key, "mlp_rows32") && value >= 0) e->tune.mlp_rows32 = value;
  else if (!strcmp(key, "mlp_nsplit") && (value == 0 || value == 2 || value == 4)) e->tune.mlp_nsplit = (int)value;
  else if (!strcmp(key, "mlp_nsplit_rows") && value >= 0) e->tune.mlp_nsplit_rows = value;
  else if (!strcmp(key, "mlp_kc") && (value == 0 || value == 64 || value == 128 || value == 192 || value == 256)) e->tune.mlp_kc = (int)value;
  else if (!strcmp(key, "table_placement")) {
    // Where a multi-gigabyte allocation lands in HBM moves the gather by up to 6 % and stays for the allocation's
    // lifetime: the feeder may try a few places with the model's own launch sets and keep the best.
    //   -1: copy the tables into one more allocation and use that one (the earlier ones stay allocated, or the allocator
    //       hands the same pages out again) | k >= 0: use candidate k | -2: free every candidate but the one in use.
    // Refused (DRS_ERR_OOM, nothing changes) when one more copy would not leave 3/4 of the device's memory free.
    int32_t rc = drs_sync(e);
    if (rc) return rc;
#ifdef DRS_LAB
    const bool scan = value == -3;
#else
    const bool scan = false;
#endif
    if (value == -1 || scan) {
      // (-3, a lab's request: one more copy as long as it fits beside 4 GB of headroom -- tools/placement_lab.py scans
      // the whole of HBM with it)
      size_t free_b = 0, total_b = 0;
      if (e->arenas.size() >= 256 || hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
          (!scan ? e->tables_bytes > free_b / 4 : e->tables_bytes + ((size_t)4 << 30) > free_b))
        return fail(e, DRS_ERR_OOM, "table_placement: no room for one more copy of the tables (%zu bytes)", e->tables_bytes);
      Arena fresh;
      hipError_t ar = arena_alloc(e, e->tables_bytes, &fresh);
      if (ar != hipSuccess) { (void)hipGetLastError(); return fail(e, DRS_ERR_OOM, "table_placement: arena allocation: %s", hipGetErrorString(ar)); }
      // (a device-to-device hipMemcpy may return before the copy is done, and the engine's streams do not wait for the
      // null stream: without the synchronize the next gather read a half-copied arena)
      if (hipMemcpy(fresh.p, e->tables, e->tables_bytes, hipMemcpyDeviceToDevice) != hipSuccess ||
          hipStreamSynchronize(nullptr) != hipSuccess) {
        arena_free(fresh);
        return fail(e, DRS_ERR_HIP, "table_placement: copy");
      }
      e->arenas.push_back(fresh);
      e->tables = fresh.p;
    } else if (value >= 0 && (size_t)value < e->arenas.size()) {
      e->tables = e->arenas[(size_t)value].p;
    } else if (value == -2) {
      drop_other_placements(e);
    } else {
      return fail(e, DRS_ERR_BAD_ARG, "table_placement %lld (candidates: %zu)", (long long)value, e->arenas.size());
    }
  }
#ifdef DRS_LAB
  else if (!strcmp(key, "table_va_next") || !strcmp(key, "table_va_select") || !strcmp(key, "table_va_goto")) {
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    auto it = std::find_if(e->arenas.begin(), e->arenas.end(), [&](const Arena& a) { return a.p == e->tables; });
    if (it == e->arenas.end()) return fail(e, DRS_ERR_STATE, "no table arena in use");
    if (!strcmp(key, "table_va_next")) {
      // value k > 0: k allocations of 4 KiB first -- they take the device-memory pages the driver would otherwise
      // hand to the page-table blocks of the new range, i.e. the range's page tables land somewhere else
      if (value > 0 && it->kind == 1) {
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = e->device;
        for (int64_t k = 0; k < value && it->pads.size() < 65536; ++k) {
          hipMemGenericAllocationHandle_t h;
          if (hipMemCreate(&h, 4096, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
          it->pads.push_back(h);
        }
      }
      return arena_move(e, *it, -1);
    }
    if (value < 0) return fail(e, DRS_ERR_BAD_ARG, "%s %lld", key, (long long)value);
    if ((rc = arena_move(e, *it, value))) return rc;
    if (!strcmp(key, "table_va_goto")) return DRS_OK;      // (every candidate stays reserved)
    // the ranges not in use are given back (address space only)
    for (size_t k = 0; k < it->vas.size(); ++k)
      if ((int)k != it->va_cur) (void)hipMemAddressFree(it->vas[k].base, it->vas[k].reserved);
    const VaRange keep = it->vas[(size_t)it->va_cur];
    it->vas.assign(1, keep);
    it->va_cur = 0;
  }
  else if (!strcmp(key, "table_probe_windows") && value >= 0 && value <= 4096) e->probe_windows = (int)value;
  else if (!strcmp(key, "table_probe_sorted") && (value == 0 || value == 1)) e->probe_sorted = (int)value;
  else if (!strcmp(key, "table_probe_row_bytes") && (value == 128 || value == 256 || value == 512)) e->probe_row_bytes = (int)value;
  else if (!strcmp(key, "table_probe_nt") && (value == 0 || value == 1)) e->probe_nt = (int)value;
  else if (!strcmp(key, "table_probe_loads") && (value == 10 || value == 20)) e->probe_loads = (int)value;
  else if (!strcmp(key, "table_probe_gather")) {
    // lab: the selection's probe (the model's own gather kernel on a one-table problem) over arena `value` as a whole;
    // result "table_probe_gather_ns"
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    if (value < 0 || (size_t)value >= e->arenas.size()) return fail(e, DRS_ERR_BAD_ARG, "table_probe_gather %lld", (long long)value);
    const Arena& a = e->arenas[(size_t)value];
    double us = 0, us2 = 0;
    HIP_TRY(e, probe_gather(e, a.p, std::min(a.va_bytes, e->tables_bytes), &us));
    HIP_TRY(e, probe_gather(e, a.p, std::min(a.va_bytes, e->tables_bytes), &us2));
    e->probe_gather_ns = (int64_t)(std::min(us, us2) * 1e3);
  }
  else if (!strcmp(key, "table_probe_latency")) {
    // lab: dependent-load latency over arena `value` as ONE chunk; result "table_probe_ns" (picoseconds per load)
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    if (value < 0 || (size_t)value >= e->arenas.size()) return fail(e, DRS_ERR_BAD_ARG, "table_probe_latency %lld", (long long)value);
    const Arena& a = e->arenas[(size_t)value];
    Slot& s0 = e->slots[0];
    const int steps = 4096;
    uint64_t ticks = 0;
    for (int pass = 0; pass < 2; ++pass) {
      HIP_TRY(e, probe_latency(a.p, std::min(a.va_bytes, e->tables_bytes), 1, steps, s0.d_ts, s0.own_stream));
      HIP_TRY(e, hipStreamSynchronize(s0.own_stream));
      HIP_TRY(e, hipMemcpy(&ticks, s0.d_ts, sizeof ticks, hipMemcpyDeviceToHost));
    }
    e->probe_ps = (int64_t)((double)ticks / e->wall_clock_khz * 1e9 / steps);     // ticks / kHz = ms; -> ps per load
  }
  else if (!strcmp(key, "table_probe")) {
    // lab: the row-read probe over arena `value` as a whole, or (value = -(k + 1)) over 1 GiB chunk k of the arena in
    // use; the result is read with drs_get_option "table_probe_mbs" (MB/s of row bytes)
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    const char* base = nullptr;
    size_t bytes = 0;
    if (value >= 0 && (size_t)value < e->arenas.size()) { base = reinterpret_cast<const char*>(e->arenas[(size_t)value].p); bytes = e->arenas[(size_t)value].va_bytes; }
    else if (value < 0 && (size_t)(-(value + 1)) * ((size_t)1 << 30) < e->tables_bytes) {
      const size_t off = (size_t)(-(value + 1)) << 30;
      base = reinterpret_cast<const char*>(e->tables) + off;
      bytes = std::min((size_t)1 << 30, e->tables_bytes - off);
    } else return fail(e, DRS_ERR_BAD_ARG, "table_probe %lld", (long long)value);
    double gbs = 0;
    HIP_TRY(e, probe_rows(base, bytes, 24576, 64, e->slots[0].d_out, e->slots[0].own_stream, &gbs, e->probe_windows, e->probe_sorted, e->probe_row_bytes, e->probe_nt, e->probe_loads));   // warm-up pass
    HIP_TRY(e, probe_rows(base, bytes, 24576, 24, e->slots[0].d_out, e->slots[0].own_stream, &gbs, e->probe_windows, e->probe_sorted, e->probe_row_bytes, e->probe_nt, e->probe_loads));
    e->probe_mbs = (int64_t)(gbs * 1e3);
  }
  else if (!strcmp(key, "table_vmm_swap")) {
    // lab (tools/placement_lab.py): is the gather's speed on an arena a property of its MEMORY or of its ADDRESS?
    // (i << 16) | j: the physical handles of arenas i and j change places (both built with "table_alloc" 1 and the same
    // chunking; both hold the same tables, so results do not change)
    int32_t rc = drs_sync(e);
    if (rc) return rc;
    const size_t i = (size_t)(value >> 16), j = (size_t)(value & 0xffff);
    if (value < 0 || i >= e->arenas.size() || j >= e->arenas.size() || i == j) return fail(e, DRS_ERR_BAD_ARG, "table_vmm_swap: no such arenas");
    Arena &a = e->arenas[i], &b = e->arenas[j];
    if (a.kind != 1 || b.kind != 1 || a.va_bytes != b.va_bytes || a.handles.size() != b.handles.size())
      return fail(e, DRS_ERR_BAD_ARG, "table_vmm_swap: both arenas must come from the virtual-memory API with the same chunking");
    HIP_TRY(e, hipMemUnmap(a.p, a.va_bytes));
    HIP_TRY(e, hipMemUnmap(b.p, b.va_bytes));
    std::swap(a.handles, b.handles);
    std::swap(a.place, b.place);
    HIP_TRY(e, arena_map(a, a.p, e->device));
    HIP_TRY(e, arena_map(b, b.p, e->device));
  }
#endif  // DRS_LAB
#ifdef DRS_LAB
  else if (!strcmp(key, "table_alloc") && value == 3) e->table_alloc = 3;
#endif
  else if (!strcmp(key, "table_alloc") && value >= 0 && value <= 2) e->table_alloc = (int)value;
  else if (!strcmp(key, "table_spacer") && value >= 0) {
    // `value` bytes of device memory are taken in 1 GiB pieces and never mapped: the next placement candidate comes from
    // further on in HBM.  "table_placement" -2 gives them back (as does drs_destroy).
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = e->device;
    size_t free_b = 0, total_b = 0;
    for (int64_t got = 0; got < value; got += (int64_t)1 << 30) {
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < ((size_t)8 << 30)) break;
      hipMemGenericAllocationHandle_t h;
      if (hipMemCreate(&h, (size_t)1 << 30, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
      e->spacers.push_back(h);
    }
  }
#ifdef DRS_LAB
  else if (!strcmp(key, "table_select_pool") && value >= 0 && value <= 192) e->sel_want_pool = value;
#endif  // DRS_LAB
  else if (!strcmp(key, "table_vmm_chunk") && value >= -1) e->vmm_chunk = value;
  else if (!strcmp(key, "table_vmm_align") && value >= 0) e->vmm_align = value;
#ifdef DRS_LAB
  else if (!strcmp(key, "table_vmm_shuffle") && (value == 0 || value == 1)) e->vmm_shuffle = (int)value;
#endif  // DRS_LAB
  else if (!strcmp(key, "out_dma") && value >= 0) { int32_t rc = drs_sync(e); if (rc) return rc; e->out_dma = value; }
  else if (!strcmp(key, "zero_copy")) { int32_t rc = drs_sync(e); if (rc) return rc; e->zero_copy = value ? 1 : 0; }
  else return fail(e, DRS_ERR_BAD_ARG, "unknown option %s=%lld", key, (long long)value);
  return DRS_OK;
}

int32_t drs_set_profiling(drs_handle e, int32_t enabled) {
  if (!e) return DRS_ERR_BAD_ARG;
  int32_t rc = drs_sync(e);
  e->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
  return rc;
}

int32_t drs_kernel_time(drs_handle e, int32_t kernel, double* sum_ms, int64_t* launches) {
  if (!e || kernel < 0 || kernel >= DRS_KERNEL_COUNT || !sum_ms || !launches) return DRS_ERR_BAD_ARG;
  *sum_ms = e->k_ms[kernel];
  *launches = e->k_n[kernel];
  return DRS_OK;
}

int32_t drs_debug_gather_stamps(drs_handle e, int32_t slot, uint64_t* out, int64_t cap, int64_t* n_blocks) {
  if (!e || slot < 0 || slot >= e->n_slots || !out || !n_blocks) return DRS_ERR_BAD_ARG;
  Slot& s = e->slots[slot];
  if (hipSetDevice(e->device) != hipSuccess) return DRS_ERR_HIP;
  const int64_t n = s.ts_blocks_done < cap / 2 ? s.ts_blocks_done : cap / 2;
  if (n > 0 && hipMemcpy(s.h_ts.data(), s.d_ts, sizeof(uint64_t) * 2 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
    return DRS_ERR_HIP;
  memcpy(out, s.h_ts.data(), sizeof(uint64_t) * 2 * (size_t)n);
  *n_blocks = n;
  return DRS_OK;
}

int32_t drs_reset_kernel_time(drs_handle e) {
  if (!e) return DRS_ERR_BAD_ARG;
  for (int i = 0; i < DRS_KERNEL_COUNT; ++i) { e->k_ms[i] = 0; e->k_n[i] = 0; e->k_bytes[i] = 0; }
  return DRS_OK;
}

int32_t drs_kernel_bytes(drs_handle e, int32_t kernel, int64_t* bytes) {
  if (!e || kernel < 0 || kernel >= DRS_KERNEL_COUNT || !bytes) return DRS_ERR_BAD_ARG;
  *bytes = e->k_bytes[kernel];
  return DRS_OK;
}

int32_t drs_get_option(drs_handle e, const char* key, int64_t* value) {
  if (!e || !key || !value) return DRS_ERR_BAD_ARG;
  const Tune& t = e->tune;
  struct { const char* k; int64_t v; } tab[] = {
      {"sls_exact", e->sls_exact}, {"sls_flat", t.sls_flat},
      {"sls_bpw", t.sls_bpw}, {"din_fused", e->din_fused}, {"dien_mfma", e->dien_mfma}, {"dien_fuse_top", e->dien_fuse_top}, {"din_s", t.din_s}, {"sls_nt", t.sls_nt}, {"din_nt", t.din_nt}, {"din_pipe", t.din_pipe}, {"gemm_split", e->gemm_split}, {"sls_uniform", e->sls_uniform}, {"sls_short_bag", e->sls_short_bag},
      {"mlp_split", e->mlp_split}, {"mlp_wide_kn", e->mlp_wide_kn}, {"mlp_fuse", e->mlp_fuse}, {"dispatch_log", e->dispatch_log},
      {"mlp_fuse_rows", e->mlp_fuse_rows}, {"mlp_small_rows", e->mlp_small_rows}, {"small_piped", e->small_piped}, {"mlp_early", e->mlp_early}, {"mlp_gemm", t.mlp_gemm}, {"mlp_gemm_tile", t.gemm_tile}, {"mlp_gemm_2cu", t.gemm_2cu}, {"mlp_gemm32", t.gemm32}, {"mlp_gemm32_blocks", t.gemm32_blocks}, {"mlp_gemm32_small", t.gemm32_small}, {"mlp_gemm32_small_blocks", t.gemm32_small_blocks}, {"mlp_stream_2cu", t.mlp_stream_2cu}, 
      {"preferred_coalesce", e->mlp_streams > 1 ? DRS_MAX_COALESCE : (e->kind == DRS_MODEL_DLRM ? 12 : 8)},
      // launch sets the feeder should keep in flight: 3 (gather | MLP | enqueue); NCF's sets are one latency-bound
      // launch of small layers that writes 1 MB of outputs over PCIe -- six of them in flight keep three resident
      {"preferred_slots", e->kind == DRS_MODEL_NCF ? 6 : 3}, {"mlp_stream", t.mlp_stream}, {"mlp_preload", t.mlp_preload}, {"mlp_kc", t.mlp_kc},
      {"mlp_debug", t.mlp_debug}, {"mlp_rows32", t.mlp_rows32}, {"mlp_nsplit", t.mlp_nsplit}, {"mlp_nsplit_rows", t.mlp_nsplit_rows}, {"shared_stream", e->shared_stream}, {"mlp_streams", e->mlp_streams}, {"mlp_layout", e->mlp_layout}, {"gather_bound", e->gather_bound},
      {"zero_copy_inputs", e->zero_copy_inputs}, {"host_threads", e->host_threads}, {"launch_thread", e->launch_thread}, {"zero_copy", e->zero_copy}, {"out_dma", e->out_dma}, {"device", e->device},
      {"table_placement", (int64_t)(std::find_if(e->arenas.begin(), e->arenas.end(), [&](const Arena& a) { return a.p == e->tables; }) - e->arenas.begin())},
      {"table_placements", (int64_t)e->arenas.size()}, {"table_bytes", (int64_t)e->tables_bytes},
      {"table_alloc", e->table_alloc}, {"table_vmm_chunk", e->vmm_chunk}, {"table_vmm_align", e->vmm_align},
      {"table_address", (int64_t)(uintptr_t)e->tables}};
  for (auto& kv : tab)
    if (!strcmp(key, kv.k)) { *value = kv.v; return DRS_OK; }
#ifdef DRS_LAB
  struct { const char* k; int64_t v; } lab[] = {
      {"table_select_pool", e->sel_pool}, {"table_select_kept", e->sel_kept},
      {"table_select_best_ns", e->sel_best_ns}, {"table_select_worst_ns", e->sel_worst_ns}, {"table_select_kept_worst_ns", e->sel_kept_worst_ns},
      {"table_select_ms", e->sel_ms}, {"table_probe_mbs", e->probe_mbs}, {"table_probe_gather_ns", e->probe_gather_ns}, {"table_probe_ps", e->probe_ps},
      {"mlp_cu_mask", e->mlp_cu_mask}, {"gather_priority", e->gather_priority}, {"gather_cu_complement", e->gather_cu_complement},
      {"table_vmm_shuffle", e->vmm_shuffle},
      {"table_kind", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return a.kind; return 0; }()},
      {"table_va_candidates", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return (int64_t)a.vas.size(); return 0; }()},
      {"table_va", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return a.va_cur; return 0; }()}};
  for (auto& kv : lab)
    if (!strcmp(key, kv.k)) { *value = kv.v; return DRS_OK; }
#endif

  return fail(e, DRS_ERR_BAD_ARG, "unknown option %s", key);
}

int32_t drs_last_dispatch(drs_handle e, int32_t slot, char* buf, int64_t cap) {
  if (!e || !buf || cap < 1) return DRS_ERR_BAD_ARG;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (!e->dispatch_log) return fail(e, DRS_ERR_STATE, "drs_last_dispatch: the record is off (drs_set_option \"dispatch_log\" 1 before the launch set)");
  if (e->launcher) e->launcher->drain();
  const Slot& s = e->slots[slot];
  const int64_t n = s.dlog.len < cap - 1 ? s.dlog.len : cap - 1;
  memcpy(buf, s.dlog.text, (size_t)n);
  buf[n] = 0;
  return DRS_OK;
}

int32_t drs_gather_bytes(drs_handle e, int32_t batch_id, int32_t bs, int64_t* bytes) {
  if (!e || !bytes) return DRS_ERR_BAD_ARG;
  if (batch_id < 0 || batch_id >= e->n_batches || !e->batches[batch_id].staged) return fail(e, DRS_ERR_STATE, "batch %d is not staged", batch_id);
  const Batch& b = e->batches[batch_id];
  if (bs < 0 || bs > b.n_samples) return fail(e, DRS_ERR_BAD_ARG, "bs out of range");
  int64_t total = 0;
  for (int t = 0; t < e->T; ++t) {
    const int64_t n = b.h_off[(size_t)t * (e->max_batch + 1) + bs];
    total += n * ((int64_t)e->D * 4 + 4) + (int64_t)bs * (4 + (int64_t)e->D * 4);
  }
  *bytes = total;
  return DRS_OK;
}

}  // extern "C"

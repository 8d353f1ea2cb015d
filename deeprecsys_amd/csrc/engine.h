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
//
// Translation units (round 6: one 3 000-line file before): engine_create.hip (device set-up, launch-form choice,
// drs_create / drs_destroy, tables and weights), engine_arena.hip (where the table arena lives), engine_inputs.hip
// (staging and the per-call input path), engine_dispatch.hip (a launch set: enqueue, wait, the operator-level entry
// points), engine_options.hip (the option table, profiling read-outs).  engine_host.h: the worker pool and the
// launcher thread of the per-call input path.
#pragma once
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

#include <type_traits>

namespace drs {
namespace eng {

extern thread_local std::string g_create_error;    // drs_create failures (no handle to hang the text on)


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
  hipEvent_t ev_dma = nullptr;           // "out_dma": last kernel done -> the copy stream may take the outputs
  bool on_dma = false;                   // ... the job in flight hands over through the copy stream
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


class HostPool;
class Launcher;

}  // namespace eng
}  // namespace drs

using namespace drs;
using namespace drs::eng;

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
  bool din_any = false;          // DIN: units din.hip has no form for (depth != 2, wide hidden layer, D not 4 k <= 256) -> din_any.hip
  int din_maxw = 0;              // ... their widest hidden layer
  int32_t* d_att_ln = nullptr;   // ... and the units' widths on the device
  int dien_fuse_top = 1;         // DIEN: the top MLP inside the recurrence's launch when it fits (din.hip dien_top_fusable)
  int dien_mfma = 2;             // DIEN recurrence on the matrix cores, 16 samples per workgroup: 2 = one wave set per layer | 1 = every wave both layers | 0 one wave per sample (VALU) | 3 the any-shape form (din_any.hip)
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
  hipStream_t stream_dma = nullptr; // "out_dma": the copy-engine transfers of the outputs and the flag writes behind them
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
  int mlp_bound = 0;                // ... the other class: MLP FLOP per gathered byte > 20 (RM3, W&D, MT-WnD, NCF, DIEN)
  int pref_slots = 3;               // "preferred_slots": launch sets the engine asks its feeder to keep in flight (choose_launch_forms)
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

namespace drs {
namespace eng {

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _r = (call);                                                                \
    if (_r != hipSuccess)                                                                  \
      return fail((e), _r == hipErrorOutOfMemory ? DRS_ERR_OOM : DRS_ERR_HIP, "%s: %s",    \
                  #call, hipGetErrorString(_r));                                           \
  } while (0)

// engine_create.hip
int32_t fail(drs_engine* e, int32_t code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
int32_t set_device(drs_engine* e);
int32_t alloc_batch(drs_engine* e, Batch& b);
void free_batch(Batch& b);
// engine_arena.hip
hipError_t arena_map(const Arena& a, float* at, int device);
void arena_free(Arena& a);
hipError_t arena_alloc(drs_engine* e, size_t bytes, Arena* out);
void drop_other_placements(drs_engine* e);
void drop_spacers(drs_engine* e);
#ifdef DRS_LAB
hipError_t probe_gather(drs_engine* e, const float* base, size_t bytes, double* us);
int32_t arena_move(drs_engine* e, Arena& a, int64_t to /* -1: a fresh range */);
#endif
// engine_dispatch.hip
void apply_stream_mode(drs_engine* e);
hipStream_t job_stream(const drs_engine* e, const Slot& s, int64_t Mv);
hipStream_t job_gather_stream(const drs_engine* e, const Slot& s, int64_t Mv);
int32_t enqueue_forward(drs_engine* e, Slot& s, int n, const Batch* const* bts, const int32_t* bss);
int32_t wait_slot(drs_engine* e, Slot& s, float* h_out, int64_t h_cap = -1);
int32_t check_handle(drs_engine* e, bool hot = false);
// engine_inputs.hip
int32_t finish_inputs(drs_engine* e, Slot& s, int mode, int32_t bs, size_t used, bool need_off);

}  // namespace eng
}  // namespace drs

#include "engine_host.h"

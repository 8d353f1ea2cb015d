// Internal declarations shared by the HIP translation units of libdrs_hip.so.
// Public surface: include/drs.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/drs.h"

namespace drs {

// Up to this many queries can be coalesced into one set of launches.  A query q owns the
// virtual rows [vstart[q], vstart[q] + bs[q]) of the slot's activation buffers; vstart[]
// is kept a multiple of 64 so a 16- or 64-row MLP block never straddles two queries.
// (DRS_MAX_COALESCE: include/drs.h)
struct QTable {
  int32_t n_q;
  int32_t vstart[DRS_MAX_COALESCE + 1];   // virtual first row per query; [n_q] = total virtual rows
  int32_t cum[DRS_MAX_COALESCE + 1];      // prefix sums of bs (valid samples before query q)
  int32_t bs[DRS_MAX_COALESCE];
};

// One launch of the multi-table gather-reduce (SparseLengthsSum x T tables).
// Bags are numbered sample-major: bag = b * T + t, so the T pooled vectors of a
// sample land next to each other in the interaction buffer
//   out[b * ld_out + col0 + t * D + d]
// which is the [B, F, D] / Concat(axis=1) layout of
// models/dlrm_s_caffe2.py:337-342,358-360 with the dense slot in front.
struct SlsArgs {
  const float* tables;        // base of the table arena
  const int64_t* tab_off;     // [T] element offset of table t in the arena
  const int64_t* tab_rows;    // [T]
  QTable q;                   // which query a bag belongs to
  const int32_t* idx[DRS_MAX_COALESCE];   // per query: [T][idx_stride] int32 indices (Cast op done)
  const int32_t* off[DRS_MAX_COALESCE];   // per query: [T][off_stride] exclusive prefix sums of lengths
  int32_t uniform_len[DRS_MAX_COALESCE];  // per query: >= 0 -> every bag has this many indices
  int64_t idx_stride;
  int64_t off_stride;
  float* out;
  int64_t ld_out;             // floats between consecutive (virtual) samples
  int32_t col0;               // first output column of table 0
  int32_t T;
  int32_t D;
  int32_t* err;               // device error word: bit0 = index out of range
  uint64_t* ts;               // optional [2 * gridDim.x] start/end wall_clock64() per workgroup
  int32_t nt;                 // fused DIN launch: table rows by non-temporal loads ("sls_nt"; set by launch_din_fused)
};

// What the launch functions chose for the launch set being enqueued ("which kernel serves which shape"):
// each launch appends "name<template args>[grid x block] " to the log of the slot (drs_last_dispatch).
// Host-side bookkeeping only; nothing on the device reads it.
struct DispatchLog {
  char text[768];
  int len;
};
void log_launch(DispatchLog* log, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Per-engine tunables (drs_set_option) and the per-device resources every launch needs.
// Nothing here is process-global: two engines in one process (the mixed-model accelerator
// engine; engines on different GPUs) keep their own copy.
struct Tune {
  int device = 0;
  const float* zero = nullptr;   // 256 B of zeros on `device`: source of out-of-range float4 loads
  const float* w_arena = nullptr;   // the engine's FC weight arena (biases + weights, one allocation) ...
  uint64_t w_arena_floats = 0;
  uint64_t w_packed_lo = 0, w_packed_hi = 0;   // float range of the arena whose layers carry a packed twin
  uint32_t w_zero_off = 0;          // ... whose first 64 floats are zeros (float offset of them)
  int sls_flat = 1;              // fixed-length bags: all row loads of a wave in flight at once
  int sls_bpw = 0;               // ... bags per wave of that variant (0 = auto | 1 | 2 | 4)
  int sls_nt = 1;                // table rows are read with non-temporal loads (every gather kernel of sls.hip)
  int sls_one = 1;               // fixed bags of ONE row (W&D, MT-WnD, NCF, DIEN): the copy form (sls_one_kernel); 1 = 64 samples per wave, 16 for small launches | 64 | 16 | 0 off
  int din_nt = 1;                // fused DIN launch: non-temporal row loads ("din_nt": +2.5 % queries/s, 0.527 -> 0.545 of peak)
  int din_pipe = 1;              // fused DIN launch, hidden width 1: indices staged in LDS, units pipelined (din_pipe_kernel)
  int din_s = 0;                 // fused DIN launch: samples per workgroup (0 = by launch size | 1 | 2 | 4)
  int mlp_preload = 0, mlp_kc = 0, mlp_stream = 2, mlp_stream_2cu = 0, mlp_gemm = 1, gemm_tile = 0, gemm_2cu = 0, mlp_debug = 0;
  int gemm32 = 1;                // wide layers through the v_mfma_f32_32x32x2_f32 kernel (gemm.hip gemm32_kernel) ...
  int gemm32_small_blocks = 0;   // ... when that shape gives at least this many workgroups
  int gemm32_small = 0;          // ... and smaller launches this gemm32 shape (0 = gemm_kernel | 22 | 21 | 12 | 11)
  int gemm32_blocks = 512;       // ... when its 128 x 128 workgroups number at least this many (two per CU)
  int64_t mlp_rows32 = 0;        // stream4_kernel: launches of at least this many rows take 32 rows per workgroup (0 = never)
  DispatchLog* log = nullptr;    // where the launch functions note what they chose (the slot being enqueued; may be null)
  // stream4_kernel's column-split form (mlp.hip SArgs::ns): launches of at most mlp_nsplit_rows rows spread the first
  // layer of the second chain over mlp_nsplit (0 = never | 2 | 4) workgroups per slab of rows.  xbuf / xcnt: the
  // exchange buffer ([xbuf_rows, xbuf_cols] floats) and ticket words of the slot being enqueued (like `log`).
  int mlp_nsplit = 0;
  int64_t mlp_nsplit_rows = 0;
  float* xbuf = nullptr;
  uint32_t* xcnt = nullptr;
  int64_t xbuf_rows = 0;
  int32_t xbuf_cols = 0;
};
// Once per DEVICE (thread-safe): the > 64 KB dynamic-LDS attribute of every kernel that needs
// it (HIP function attributes are per device) and the device's zero page.
hipError_t device_init(int device, const float** zero_page);
hipError_t mlp_set_attrs();    // mlp.hip's kernels, on the current device
hipError_t gemm_set_attrs();   // gemm.hip's kernels, on the current device

// launch on `stream`; exact != 0 selects the sequential-order variant.
// stop_event (optional): recorded by the gather dispatch itself when it completes
hipError_t launch_sls(const SlsArgs& a, int exact, const Tune& tune, hipStream_t stream,
                      hipEvent_t stop_event = nullptr);
int64_t sls_grid_blocks(const SlsArgs& a, int exact, const Tune& tune);
bool sls_flat_applicable(const SlsArgs& a, const Tune& tune);   // would a non-exact launch run the flat variant?

// Completion hand-off to the host without a copy or a stream sync: the LAST kernel of a
// query stores its outputs straight into host-mapped pinned memory and, once every one
// of its workgroups has done so (device-scope arrival counter), the last arriver copies
// the device error word and publishes `seq` in a host flag the CPU is polling.
struct Done {
  uint32_t* counter;        // device arrival counter (zero between uses); nullptr = disabled
  uint32_t* host_flag;      // host-mapped: receives seq
  uint32_t* host_err;       // host-mapped: receives *dev_err
  const uint32_t* dev_err;  // device error word written by earlier kernels of the query
  uint32_t seq;
  // live timing of the gather launch: its per-workgroup [start, end] clock stamps are
  // reduced to (min start, max end) by the same last-arriving workgroup and stored in
  // host-mapped memory, so profiling adds no copy, no sync and no extra launch
  const uint64_t* ts;       // [2 * ts_blocks] or nullptr
  uint32_t ts_blocks;
  uint64_t* span_acc;       // device [2 * workgroups of the last kernel]: per-workgroup (min, max)
  uint64_t* host_span;      // host-mapped [2]
  // the outputs themselves: every workgroup stores to the DEVICE buffer; only the last
  // arriver streams them to host-mapped memory (8 KB for 2048 rows) behind ONE system-scope
  // release.  (Letting all workgroups store to host memory and fence at system scope cost
  // 20 us per launch: measured 56 us vs 36 us.)
  const float* dev_out;
  float* host_out;
  uint32_t out_words;
  // early start (stream4_kernel, "mlp_early"): the launch does NOT wait for the gather on its stream; it runs its
  // prologue and the first chain, then polls `wait_flag` for `wait_val` (a stream-ordered 32-bit write queued behind the
  // gather) before it fetches the second chain's pooled rows.  nullptr: the stream orders the launch behind the gather.
  uint32_t wait_val;
  const uint32_t* wait_flag;
};

// y[M, N] (ld = ldy) = act(x[M, K] (ld = ldx) . W[N, K]^T + b), k-ordered fp32 MFMA chain
// Optional per-query sources of the FIRST layer's input rows (dense features live in the
// staged batches, one array per query); rows of y are always virtual rows.
struct XSrc {
  QTable q;
  const float* x[DRS_MAX_COALESCE];
  // > 0 (gemm32_kernel's scalar-base forms only, launch_gemm): columns [0, ksplit) of an input row come from the
  // queries' own staged arrays (rows ksplit floats apart), columns from ksplit on from `x` at the VIRTUAL row -- W&D's and
  // MT-WnD's first layer reads Concat(dense, pooled embeddings) without the dense rows ever being copied next to the
  // embeddings (models/wide_and_deep.py:271-281).  A multiple of 32, >= 64, < K.
  int32_t ksplit;
};

hipError_t launch_fc(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W,
                     const float* b, int32_t N, int32_t act, float* y, int64_t ldy,
                     const Tune& tune, hipStream_t stream, const Done* done = nullptr,
                     const XSrc* xs = nullptr);

// would launch_gemm take a form that reads a split input row (XSrc::ksplit) for this layer?  (gemm.hip)
bool gemm_split_applicable(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, int32_t N, const XSrc& xs,
                           const Tune& tune);
// Register-blocked GEMM for wide layers (gemm.hip); false = not applicable, use launch_fc's
// own kernel.  zero_page: 16 B of zeros in device memory.
bool launch_gemm(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, const float* b,
                 int32_t N, int32_t act, float* y, int64_t ldy, const Tune& tune,
                 hipStream_t stream, const Done& done, const XSrc& xs, hipError_t* err);

// Fused chain of up to DRS_MAX_CHAIN FC layers on 16-row slabs; intermediate
// activations never leave LDS.
#define DRS_MAX_CHAIN 6
struct ChainArgs {
  const float* x;
  int64_t ldx;
  int64_t M;
  int32_t n_layers;
  int32_t width[DRS_MAX_CHAIN + 1];   // width[0] = K of first layer
  const float* W[DRS_MAX_CHAIN];
  const float* b[DRS_MAX_CHAIN];
  int32_t act[DRS_MAX_CHAIN];
  float* y;
  int64_t ldy;
};
hipError_t launch_chain(const ChainArgs& a, const Tune& tune, hipStream_t stream,
                        const Done* done = nullptr, const XSrc* xs = nullptr);
size_t chain_lds_bytes(const ChainArgs& a, const Tune& tune);
// two chains on the same rows in one launch (bottom MLP, then the top MLP that reads the
// buffer the first one wrote its last layer into)
// Optional dot interaction BETWEEN the two chains of launch_chain2 (DLRM "dot",
// models/dlrm_s_caffe2.py:334-354): the first chain writes the dense_out slot of T, the
// interaction turns T [M, F*D] into R [M, D + P], the second chain reads R.
struct DotArgs {
  const float* T;   // == a.y: [M, F*D] sample-major (dense_out | emb_0 | ...), ld = ldt
  int64_t ldt;
  int32_t F, D, itself;
  float* R;         // == b->x: interaction output, ld = ldr (also kept for drs_fetch_interaction)
  int64_t ldr;
};
// Optional NCF-style join (models/ncf.py:301-305,317-346): the second chain's input buffer is
// [ src[:, col_a:col_a+cols] + src[:, col_b:col_b+cols] | output of the first chain ], i.e.
// Sum of two pooled embeddings in front of the MLP branch.  dst (== b->x) receives the summed
// block as well (the first chain stores its output right behind it, a.y == dst + cols).
struct SumArgs {
  const float* src;
  int64_t ld;
  int32_t col_a, col_b, cols;
  float* dst;
  int64_t ldd;
};
hipError_t launch_chain2(const ChainArgs& a, const ChainArgs* b, const Tune& tune, hipStream_t stream,
                         const Done* done = nullptr, const XSrc* xs = nullptr,
                         const DotArgs* dot = nullptr, const SumArgs* sum = nullptr);
// would launch_chain2(a, &b, ..., dot) run as the stream kernel?  (With a dot interaction in
// between it is the only kernel that can.)
bool stream_applicable(const ChainArgs& a, const ChainArgs& b, const Tune& tune, const XSrc* xs,
                       const DotArgs* dot, const SumArgs* sum = nullptr, bool* can_defer = nullptr);
size_t chain2_lds_bytes(const ChainArgs& a, const ChainArgs& b, const Tune& tune);

// T [B, F, D] (sample stride ldt) -> R [B, D + P] (ld = ldr), see drs_interact_dot
hipError_t launch_interact_dot(const float* T, int64_t ldt, int64_t B, int32_t F, int32_t D,
                               int32_t itself, float* R, int64_t ldr, hipStream_t stream);

// out[i] = a[i] + b[i] rows of width D (NCF Sum, models/ncf.py:301-305) and strided copies
hipError_t launch_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                           int64_t ldo, int64_t M, int32_t D, hipStream_t stream);
// dense rows of all coalesced queries (xs.x[q], m_den wide) -> their virtual rows of `out`
hipError_t launch_copy_rows_multi(const XSrc& xs, int32_t m_den, float* out, int64_t ldo, hipStream_t stream);
hipError_t launch_copy_rows(const float* a, int64_t lda, float* out, int64_t ldo, int64_t M,
                            int32_t D, hipStream_t stream);

// DIN (din.hip; models/din.py:247-330).  packed: the attention units' weights, one block of
// din_unit_stride(D, h) floats per unit, built by launch_din_pack from 4 device pointers per unit
// (W1 [h, 3D], b1 [h], W2 [D, h], b2 [D]).  launch_din_attention: T [M, Tn*D] pooled rows ->
// R [M, 4*D] = [ profile | Sum_i unit_i(u_i, ad) | ad | context ], bit-identical to the oracle.
// launch_din_fused: gather + units + Concat in one launch (a: the gather's arguments; a.out unused).
int64_t din_unit_stride(int D, int h);
hipError_t launch_din_pack(const float* const* att, float* packed, int32_t U, int32_t D, int32_t h, hipStream_t stream);
hipError_t launch_din_attention(const float* T, int64_t ldt, int64_t M, int32_t Tn, int32_t D, int32_t h,
                                const float* packed, float* R, int64_t ldr, hipStream_t stream);
bool din_fused_applicable(int32_t D, int32_t h);
int64_t din_fused_grid(const SlsArgs& a, const Tune& tune);
hipError_t launch_din_fused(const SlsArgs& a, int32_t h, const float* packed, float* R, int64_t ldr,
                            const Tune& tune, hipStream_t stream, hipEvent_t stop);

// DIEN (din.hip; models/dien.py:308-432).  w: 8 device pointers {i2h_w, i2h_b, gates_t_w, gates_t_b} of
// layer 1 then layer 2; launch_dien_rnn: T [rows, Tn*D] pooled rows of the coalesced queries q ->
// R [rows, H + 3*D] = [ last state of layer 2 | profile | ad | context ].
// Any-shape forms (din_any.hip): attention units of any depth and width -- d_ln: the unit's n_ln widths on the device,
// d_att: per unit and layer {W, b}, maxw: the widest hidden layer -- and the recurrence for any D and H (packed as
// launch_dien_pack lays it out).  *_fits: the sample's activations fit the 160 KB of LDS.
bool din_any_fits(int32_t D, int32_t maxw);
bool dien_any_fits(int32_t D, int32_t H);
hipError_t launch_din_attention_any(const float* T, int64_t ldt, int64_t M, int32_t Tn, int32_t D, int32_t n_ln,
                                    const int32_t* d_ln, const float* const* d_att, int32_t maxw, float* R, int64_t ldr,
                                    hipStream_t stream);
hipError_t launch_dien_rnn_any(const float* T, int64_t ldt, const QTable& q, int32_t Tn, int32_t D, int32_t H,
                               const float* packed, float* R, int64_t ldr, hipStream_t stream);
bool dien_applicable(int32_t D, int32_t H);
int64_t dien_packed_floats(int32_t D, int32_t H);
hipError_t launch_dien_pack(const float* const* w, float* packed, int32_t D, int32_t H, hipStream_t stream);
// (mfma: the 16-samples-per-workgroup matrix-core form when H % 16 == 0, reading the row-major
// weights w[8] (host array of device pointers); else one wave per sample on `packed`; same bits)
// top (optional, matrix-core form only): the model's top MLP over R's rows in the same launch -- n <= 4 layers,
// every K a multiple of 4 and <= 256 (dien_top_fusable) -- written to out [rows, ldo]; `done`
// then makes the launch sign off the launch set (mlp_dev.h signal_done).  Same bits as the stream kernels' chains.
struct DienTop {
  int32_t n, kmax, sc1, pad_;      // layers (0: none) | rows of an LDS activation buffer (dien_top_kmax) | write-through output stores
  const float* Wp[4];              // the layers' PACKED twins (mlp.hip pack_stream_kernel; W + roundup64(K N) in the arena)
  const float* b[4];
  int32_t K[4], N[4], act[4];
  float* out;
  int64_t ldo;
};
bool dien_top_fusable(int32_t n_layers, const int32_t* widths /* n_layers + 1 */, int32_t H);
int32_t dien_top_kmax(int32_t n_layers, const int32_t* widths);
hipError_t launch_dien_rnn(const float* T, int64_t ldt, const QTable& q, int32_t Tn, int32_t D, int32_t H,
                           const float* packed, const float* const* w, int mfma, float* R, int64_t ldr,
                           hipStream_t stream, const DienTop* top = nullptr, const Done* done = nullptr);

// Weights of one FC layer (W [N, K] row-major) in the stream kernel's MFMA-operand order (mlp.hip):
// per 128-column pass and 64-k chunk, per wave (16 columns), four float4 per lane.
int64_t stream_packed_floats(int K, int N);
hipError_t launch_pack_stream_weights(const float* W, int32_t K, int32_t N, float* Wp, hipStream_t stream);

// random 128- / 256- / 512-byte row reads over [base, base + bytes) in the gather's access shape, timed with events on `s`
// (sls.hip probe_rows_kernel); sink: >= 1 KiB of device memory nobody reads
hipError_t probe_rows(const void* base, size_t bytes, int waves, int reps, float* sink, hipStream_t s, double* gbs,
                      int windows = 0, int sorted = 0, int row_bytes = 256, int nt = 1, int loads = 20);

// dependent-load walk through each of n_chunks chunks of chunk_bytes (one lane per chunk); d_ticks[k] = 100 MHz ticks
hipError_t probe_latency(const void* base, size_t chunk_bytes, int n_chunks, int steps, uint64_t* d_ticks, hipStream_t s);

hipError_t launch_fill_uniform(float* W, int64_t n, int32_t t, float lo, float hi, uint64_t seed,
                               hipStream_t stream);

}  // namespace drs

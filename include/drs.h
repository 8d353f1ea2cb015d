/*
 * drs.h -- C ABI of the MI355X (gfx950) inference engine for the DeepRecSys
 * hot path: the per-query DLRM-style forward (SparseLengthsSum gathers,
 * bottom/top MLP, dot/cat feature interaction) that the reference dispatches
 * from inferenceEngine.py / accelInferenceEngine.py.
 *
 * This header is the drop-in boundary.  The reference has no native FFI of its
 * own (it is 100% Python on top of Caffe2), so each entry point below cites the
 * reference Python interface / Caffe2 operator call site it replaces.  All
 * paths are into the reference tree (harvard-acc/DeepRecSys).
 *
 * Conventions
 *   - every function returns an int32 status: 0 = DRS_OK, negative = error;
 *     no exception or abort ever crosses this boundary
 *     (reference style is print + sys.exit(), accelInferenceEngine.py:28-31).
 *   - h_* pointers are host memory owned by the caller (copied before return
 *     unless the name says "pinned"); d_* pointers are device memory on the
 *     engine's GPU.
 *   - plain pointers and sizes only; no torch types.
 *   - the library needs a GPU: drs_create fails with DRS_ERR_HIP when no
 *     gfx950 device is visible.  There is no CPU fallback behind this ABI.
 */
#ifndef DRS_H_
#define DRS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRS_ABI_VERSION 5

typedef struct drs_engine* drs_handle;

/* status codes */
enum {
  DRS_OK = 0,
  DRS_ERR_BAD_ARG = -1,      /* null pointer, out-of-range id, shape mismatch          */
  DRS_ERR_OOM = -2,          /* hipMalloc / hipHostMalloc failed                       */
  DRS_ERR_HIP = -3,          /* any other HIP runtime error (see drs_last_error)       */
  DRS_ERR_INDEX_RANGE = -4,  /* an index is <0 or >= rows  (Caffe2 CAFFE_ENFORCE)      */
  DRS_ERR_LENGTHS_SUM = -5,  /* sum(lengths) != number of indices (Caffe2 ENFORCE)     */
  DRS_ERR_STATE = -6,        /* e.g. forward before weights/batch were set             */
  DRS_ERR_UNSUPPORTED = -7   /* shape outside what the kernels implement               */
};

/* model wiring: which reference graph builder the engine mirrors */
enum {
  DRS_MODEL_DLRM = 0, /* models/dlrm_s_caffe2.py:367-389  (RMC1/2/3)                   */
  DRS_MODEL_WND = 1,  /* models/wide_and_deep.py:282-305  (SLS ++ raw dense -> top MLP) */
  DRS_MODEL_NCF = 2,  /* models/ncf.py:317-346            (MF Sum ++ MLP branch)        */
  DRS_MODEL_MTWND = 3 /* models/multi_task_wnd.py:286-316 (W&D trunk, all-ReLU shared top MLP,
                         then num_tasks task heads over its output; outputs = the heads' last
                         layers side by side, [bs, num_tasks * ln_task[-1]] -- the reference
                         keeps the last head as `last_output`, :316)                          */
  ,
  DRS_MODEL_DIN = 4   /* models/din.py:247-330: tables = [user profile | U behaviour tables |
                         candidate ad | context], U = num_tables - 3; per behaviour table an
                         attention unit with its OWN MLP over Concat(u_i, ad, u_i + ad);
                         atten_out = Sum over the units; top MLP over Concat(profile, atten_out,
                         ad, context); every activation ReLU.  cfg: ln_bot = the unit's widths
                         [3*D, <arch_mlp_bot>, D] -- any number of hidden layers of any width (the
                         shipped "1": the fused launch; others: slower forms, same results) --,
                         ln_top = [4*D, ...], no dense input                                      */
  ,
  DRS_MODEL_DIEN = 5  /* models/dien.py:308-432: tables as DIN.  The U behaviour embeddings of a query,
                         Concat'ed [bs, U*D] and Reshape'd (row-major reinterpretation, as the reference
                         does it) to [U, bs, D], run through two caffe2 rnn_cell.BasicRNN layers (tanh,
                         zero initial state, D -> H and H -> H; the FC + Softmax between them is dead in
                         the reference graph and not computed); top MLP (all ReLU) over Concat(last
                         state, profile, ad, context).  cfg: ln_bot = [D, H] (--hidden_size),
                         ln_top = [H + 3*D, ...], no dense input.  Any D and H whose D + 4 H floats fit
                         160 KB (D in {16, 32, 64} with H in {8, 16, 32, 64}: the matrix-core form).   */
};

/* feature interaction (models/dlrm_s_caffe2.py:331-365) */
enum { DRS_INTERACT_DOT = 0, DRS_INTERACT_CAT = 1 };

/* FC epilogue (Relu / Sigmoid ops, models/dlrm_s_caffe2.py:268-272) */
enum { DRS_ACT_NONE = 0, DRS_ACT_RELU = 1, DRS_ACT_SIGMOID = 2 };

/* which MLP a layer belongs to in drs_set_fc */
enum { DRS_MLP_BOT = 0, DRS_MLP_TOP = 1, DRS_MLP_FINAL = 2 /* NCF predictor */,
       DRS_MLP_TASK0 = 16 /* + k: task head k of DRS_MODEL_MTWND */,
       DRS_MLP_ATT0 = 1024 /* + i: attention unit i of DRS_MODEL_DIN */,
       DRS_MLP_RNN0 = 32 /* DRS_MODEL_DIEN, first BasicRNN: layer 0 = i2h (W [H, D], b [H]), layer 1 =
                            gates_t (W [H, H], b [H]) */,
       DRS_MLP_RNN1 = 33 /* second BasicRNN: layer 0 = i2h (W [H, H]), layer 1 = gates_t (W [H, H]) */ };

/* kernels that keep live HIP-event timings (drs_kernel_time) */
enum {
  DRS_KERNEL_SLS = 0,   /* multi-table gather-reduce (+ bottom MLP blocks)             */
  DRS_KERNEL_MLP = 1,   /* all top-MLP / interaction launches of a forward             */
  DRS_KERNEL_SLS_CLOCK = 2, /* the gather launch again, timed by the device's own constant-
                           rate clock: max(end) - min(start) over its workgroups.  HIP
                           events bracket a ~10 us launch with several us of packet
                           processing, this does not (Measurement)     */
  DRS_KERNEL_COUNT = 3
};

/*
 * Shape algebra of DLRM_Net.__init__ (models/dlrm_s_caffe2.py:391-476).
 * ln_bot / ln_top are the full width lists including the input width, i.e.
 * ln_top[0] == num_int (":415-430").  drs_create re-derives num_int and
 * rejects a mismatch exactly like the reference's sys.exit checks (:432-440).
 */
typedef struct drs_model_cfg {
  int32_t model_kind;           /* DRS_MODEL_*                                          */
  int32_t num_tables;           /* len(arch_embedding_size)                             */
  const int64_t* table_rows;    /* [num_tables] rows of each table                      */
  int32_t sparse_dim;           /* arch_sparse_feature_size (D), 1 .. 4096.  Multiples of 4 up to 256 (every shipped
                                 * config) take the fast kernels, any other width the generic forms (same results,
                                 * slower)                                                                          */
  int32_t n_bot;                /* len(ln_bot)   (1 => no bottom MLP, W&D style)        */
  const int32_t* ln_bot;        /* [n_bot]                                              */
  int32_t n_top;                /* len(ln_top) including num_int                        */
  const int32_t* ln_top;        /* [n_top]                                              */
  int32_t interaction_op;       /* DRS_INTERACT_*                                       */
  int32_t interaction_itself;   /* arch_interaction_itself                              */
  int32_t sigmoid_top;          /* 1-based layer index that gets Sigmoid, -1 = none     */
  int32_t max_batch;            /* max_mini_batch_size: rows of a staged batch          */
  int32_t max_lookups;          /* upper bound on indices per bag (staging capacity)    */
  int32_t num_staged_batches;   /* num_batches: how many input sets stay device-resident*/
  int32_t num_slots;            /* in-flight queries (streams); >=1                     */
  /* DRS_MODEL_MTWND only (zero / NULL otherwise): arch_mlp_tasks and num_multi_tasks
   * (multi_task_wnd.py:360,304-312); ln_task[0] == ln_top[-1]; sigmoid_top is the 1-based
   * layer index of a TASK head that gets Sigmoid (the reference passes the shared top's
   * ln_top.size-1 there, :399,309), the shared top MLP is all ReLU (:301)                   */
  int32_t n_task;               /* len(ln_task)                                         */
  const int32_t* ln_task;       /* [n_task]                                             */
  int32_t num_tasks;            /* task heads                                           */
} drs_model_cfg;

/* ---- library / device ------------------------------------------------------ */
int32_t drs_abi_version(void);
/* which implementation of this header a loaded library is: "hip:gfx950" for libdrs_hip.so (the
 * product).  The CPU restatement that tests/ drive the host code with (oracle/drs_cpu_abi.cpp)
 * answers "cpu:oracle"; the product binding (deeprecsys_amd/_native.py) refuses to bind anything
 * whose answer does not start with "hip:" -- no environment variable or path can put the
 * arithmetic of a served query on the CPU.  Static string, never NULL.                        */
const char* drs_backend(void);
int32_t drs_device_count(int32_t* out_count);
/* last error text of this handle (or of the failed drs_create when h == NULL);
 * owned by the library, valid until the next call on the same thread */
const char* drs_last_error(drs_handle h);

/* ---- lifecycle --------------------------------------------------------------
 * replaces: DLRM_Wrapper(args) + .create(...)  (models/dlrm_s_caffe2.py:88-158)
 * i.e. everything the engine does before inferenceEngineReadyQueue.put(True)
 * (inferenceEngine.py:81-88,190; accelInferenceEngine.py:34).                  */
int32_t drs_create(const drs_model_cfg* cfg, int32_t device_id, drs_handle* out);
int32_t drs_destroy(drs_handle h);

/* ---- parameters -------------------------------------------------------------
 * replaces: FeedBlob(tbl_s, W)            (models/dlrm_s_caffe2.py:297-303)
 *           FeedBlob(tag_fc_w/_b, W / b)  (models/dlrm_s_caffe2.py:245-251)    */
int32_t drs_set_table(drs_handle h, int32_t t, const float* h_W /*[rows,D]*/, int64_t rows);
/* device-side fill for benchmark-sized tables; value(t,i) is a pure function
 * of (seed, t, i) restated bit-exactly in oracle/               */
int32_t drs_fill_table_uniform(drs_handle h, int32_t t, float lo, float hi, uint64_t seed);
/* drs_set_fc copies W [m, n] (row-major, as the reference feeds it) and b [m] into the engine's
 * weight arena and, for the bottom / top / final / task MLPs, builds the layer's packed twin (the
 * same weights in MFMA-operand order, read by the MLP stream kernel): weights take ~2x their size
 * in HBM.  Synchronous; may be called again to replace a layer's weights.                         */
int32_t drs_set_fc(drs_handle h, int32_t mlp, int32_t layer /*0-based*/,
                   const float* h_W /*[m,n] row-major, NOT transposed*/,
                   const float* h_b /*[m]*/, int32_t m, int32_t n);

/* ---- inputs -----------------------------------------------------------------
 * replaces: the engine holding lX / lS_l / lS_i in process memory
 * (inferenceEngine.py:83) and slicing a prefix per request (:200-206).
 * h_idx[t] holds the concatenated indices of table t for all n_samples bags
 * (int64 as fed by the reference, narrowed here = the Cast op,
 * models/dlrm_s_caffe2.py:308-309); h_len[t][b] is the bag length.
 * Validation = Caffe2's ENFORCEs: DRS_ERR_INDEX_RANGE / DRS_ERR_LENGTHS_SUM.  */
int32_t drs_stage_batch(drs_handle h, int32_t batch_id, int32_t n_samples,
                        const float* h_dense /*[n_samples, m_den] or NULL (NCF)*/,
                        const int64_t* const* h_idx /*[T] -> [n_idx[t]]*/,
                        const int64_t* n_idx /*[T]*/,
                        const int32_t* const* h_len /*[T] -> [n_samples]*/);

/* ---- hot path ---------------------------------------------------------------
 * replaces: run_queues(...) + workspace.RunNet(net) + FetchBlob
 * (models/dlrm_s_caffe2.py:162-174,568; inferenceEngine.py:33,211-215), and the
 * predict_time()+sleep() of accelInferenceEngine.py:63-64.
 * A query is the first `bs` samples of staged batch `batch_id`.
 * h_out receives [bs, n_out] floats (n_out = ln_top[-1]).                      */
int32_t drs_forward(drs_handle h, int32_t batch_id, int32_t bs, float* h_out);
/* asynchronous form: enqueue on slot's stream; result lands in the slot's
 * pinned buffer; drs_wait blocks for that slot and copies to h_out (may be NULL
 * to only wait).  Slots are independent HIP streams and may overlap.           */
int32_t drs_forward_async(drs_handle h, int32_t slot, int32_t batch_id, int32_t bs);
/* Query coalescing: n (1..16) queries -- query i = first bs[i] samples of staged batch
 * batch_ids[i] -- run as ONE set of launches (one gather over all their bags, one MLP
 * pass over all their rows).  This is what the engine does when several requests are
 * already waiting in its queue; drs_wait then returns the n results back to back,
 * [sum(bs), n_out].  drs_forward_async is the n == 1 case.                         */
#define DRS_MAX_COALESCE 16   /* (8 until round 3: sets of 16 x 256 rows give the 16-row MLP workgroups of the
                                   MLP-bound models all 256 CUs; drs_get_option "preferred_coalesce" says what a model wants) */
int32_t drs_forward_multi_async(drs_handle h, int32_t slot, int32_t n, const int32_t* batch_ids,
                                const int32_t* bs);
/* h_out_floats: capacity of h_out in floats; must hold sum(bs) * n_out of what was submitted
 * on the slot (DRS_ERR_BAD_ARG otherwise, the job stays in flight); ignored when h_out is NULL */
int32_t drs_wait(drs_handle h, int32_t slot, float* h_out, int64_t h_out_floats);
int32_t drs_sync(drs_handle h);
/* non-staged inputs (the run_queues(ids, lengths, fc, bs) signature,
 * models/dlrm_s_caffe2.py:162-174): int64 -> int32 narrowing (the Cast op, :308-309) and the
 * Caffe2 ENFORCEs on the host, into the slot's pinned staging block, then the same forward.
 * The caller's arrays are consumed before the call returns.  _async + drs_wait keeps several
 * slots in flight (the next query's host pass overlaps this query's kernels).        */
int32_t drs_forward_inputs(drs_handle h, int32_t slot, int32_t bs,
                           const float* h_dense,
                           const int64_t* const* h_idx, const int64_t* n_idx,
                           const int32_t* const* h_len, float* h_out);
int32_t drs_forward_inputs_async(drs_handle h, int32_t slot, int32_t bs,
                                 const float* h_dense,
                                 const int64_t* const* h_idx, const int64_t* n_idx,
                                 const int32_t* const* h_len);
/* The same call for the arrays exactly as the reference's feeder holds them
 * (DLRM_Wrapper.run_queues(ids, lengths, fc, batch_size), models/dlrm_s_caffe2.py:162-174;
 * sliced from the pre-generated sets at inferenceEngine.py:200-206): ids is a [T, n_idx_per_table]
 * int64 array and lengths a [T, bs] int32 array, rows `*_row_stride` ELEMENTS apart (a column
 * slice of a bigger array is fine).  Saves the binding a pointer table per query. */
int32_t drs_run_queues_async(drs_handle h, int32_t slot, int32_t bs, const float* h_dense,
                             const int64_t* h_ids, int64_t ids_row_stride, int64_t n_idx_per_table,
                             const int32_t* h_lengths, int64_t len_row_stride);
/* Several queued requests' arrays as ONE launch set -- what an engine process does with the
 * requests it finds waiting in its queue (inferenceEngine.py:195-215 takes one request per turn
 * and slices its arrays; accelInferenceEngine.py drains the queue): query i is (bs[i],
 * h_dense[i] [bs, m_den], h_ids[i] [T, n_idx_per_table[i]] int64, h_lengths[i] [T, bs] int32, rows
 * ids_row_stride[i] / len_row_stride[i] ELEMENTS apart), n in 1..DRS_MAX_COALESCE.  Every query
 * is narrowed and ENFORCE-checked like a drs_run_queues_async call (the lowest failing query
 * reports; nothing is launched then); the converted inputs of the whole set cross PCIe in one
 * DMA copy that overlaps the kernels of the sets before it.  drs_wait returns the n results back
 * to back, [sum(bs), n_out], exactly as for drs_forward_multi_async, and a query's bits do not
 * depend on what it was coalesced with.  All arrays are consumed before the call returns. */
int32_t drs_run_queues_multi_async(drs_handle h, int32_t slot, int32_t n, const int32_t* bs,
                                   const float* const* h_dense, const int64_t* const* h_ids,
                                   const int64_t* ids_row_stride, const int64_t* n_idx_per_table,
                                   const int32_t* const* h_lengths, const int64_t* len_row_stride);
/* read back the interaction tensor R [bs, num_int] (the top MLP's input) of the last forward
 * on `slot` (parity tests).  Rows are the slot's virtual rows: a single query starts at row 0,
 * coalesced query i at the sum of round_up(bs_j, 64) over j < i; bs may span several queries. */
int32_t drs_fetch_interaction(drs_handle h, int32_t slot, int32_t bs, float* h_R);
int32_t drs_out_width(drs_handle h, int32_t* n_out);
int32_t drs_interaction_width(drs_handle h, int32_t* num_int);

/* ---- operator-level entry points (same semantics as the Caffe2 ops) ----------
 * All pointers are DEVICE pointers on the engine's GPU; launches go to slot 0's
 * stream and the call returns after the stream is idle.  The operands must be complete
 * when the call is made: a caller that produced them on another stream (e.g. torch's)
 * synchronises that stream first -- the engine cannot order itself behind foreign work.
 *
 * drs_sls  == SparseLengthsSum([tbl, idx, len]) (models/dlrm_s_caffe2.py:317-325)
 *   out[b,:] = sum over the bag's indices of W[idx,:], fp32, sequential in
 *   index order when exact_order != 0 (bit-identical to the Caffe2 CPU
 *   perfkernel); empty bag -> zeros.                                           */
int32_t drs_sls(drs_handle h, const float* d_W, int64_t rows, int32_t D,
                const int32_t* d_idx, const int32_t* d_len, int64_t n_bags,
                int64_t n_idx, float* d_out /*[n_bags, D]*/, int32_t exact_order);
/* drs_fc == FC([x,W,b]) + Relu|Sigmoid (models/dlrm_s_caffe2.py:258-272)
 *   y = act(x . W^T + b); accumulation is a k-ordered fp32 fma chain (MFMA).   */
int32_t drs_fc(drs_handle h, const float* d_x, int64_t M, int32_t K, const float* d_W /*[N,K]*/,
               const float* d_b /*[N], or NULL = no bias (zeros)*/, int32_t N, int32_t act,
               float* d_y /*[M,N]*/);
/* drs_interact_dot == Concat(add_axis) + BatchMatMul(trans_b) + Flatten +
 * BatchGather(tril) + Concat  (models/dlrm_s_caffe2.py:334-354, :529-535)
 *   d_T [B,F,D] -> d_R [B, D + F(F-1)/2 (+F if itself)]                        */
int32_t drs_interact_dot(drs_handle h, const float* d_T, int64_t B, int32_t F, int32_t D,
                         int32_t itself, float* d_R);

/* ---- tuning ------------------------------------------------------------------
 * Integer options of a handle; every key, its values, default and the measurement behind it: docs/OPTIONS.md.
 * Results never depend on an option except where noted (sls_exact: the gather's fp32 summation order).
 * The product library takes the keys below; unknown key or value -> DRS_ERR_BAD_ARG.
 *   gather        "sls_exact" 0|1 (1: sequential order, bit-identical to Caffe2's SparseLengthsSum)
 *                 "sls_flat" 0|1|2   "sls_bpw" 0|1|2|4   "sls_nt" 0|1   "sls_one" 0|1|16|64
 *                 "din_fused" 0|1   "din_pipe" 0|1   "din_s" 0|1|2|4   "din_nt" 0|1
 *                 "dien_mfma" 0|1|2|3   "dien_fuse_top" 0|1
 *   MLP side      "mlp_fuse" 0|1   "mlp_split" 0|1   "mlp_wide_kn" n   "gemm_split" 0|1
 *                 "mlp_stream" 2|4   "mlp_stream_2cu" 0|1   "mlp_rows32" n
 *                 "mlp_nsplit" 0|2|4   "mlp_nsplit_rows" n   "mlp_gemm_tile" 0|22|12|21|11|214|322|321|312|311
 *   streams, host "shared_stream" 0|1|2   "mlp_streams" 1..8   "host_threads" -1..64
 *                 "zero_copy_inputs" 1|2|3   "out_dma" bytes   "dispatch_log" 0|1
 *   table arena   "table_placement" -1|-2|k   "table_alloc" 0|1|2   "table_spacer" bytes
 *   read only     "preferred_coalesce"  "preferred_slots"  "gather_bound"  "device"
 *                 "table_placements"  "table_bytes"  "table_address"
 * Options belong to the handle: two engines in one process (the mixed-model accelerator engine) keep their own.
 * The lab build (make -C deeprecsys_amd/csrc lab-lib, -DDRS_LAB) also takes the lab's instruments and the options
 * of every form that lost its measurement (docs/OPTIONS.md, last section); the product library refuses them.      */
int32_t drs_set_option(drs_handle h, const char* key, int64_t value);
int32_t drs_get_option(drs_handle h, const char* key, int64_t* value);

/* ---- measurement ------------------------------------------------------------
 * Live timing of the engine's own launches (bench.py roofline leg).
 *   level 1: every workgroup of the gather stamps the device's constant-rate clock at
 *            entry and exit; the query's last kernel reduces the stamps to (min start,
 *            max end) and hands them to the host with the results -- no copy, no sync,
 *            no extra launch, so it can stay on inside a timed region
 *            (DRS_KERNEL_SLS_CLOCK).
 *   level 2: additionally brackets the gather and the rest of the launch set with
 *            hipEvents recorded on the stream they are launched on (DRS_KERNEL_SLS,
 *            DRS_KERNEL_MLP); the event packets perturb a ~10-60 us launch by several
 *            us, which is why level 1 exists.                                      */
int32_t drs_set_profiling(drs_handle h, int32_t level);
int32_t drs_kernel_time(drs_handle h, int32_t kernel /*DRS_KERNEL_*/,
                        double* sum_ms, int64_t* launches);
int32_t drs_reset_kernel_time(drs_handle h);
/* algorithmic bytes (drs_gather_bytes' formula) of exactly the gather launches whose durations
 * drs_kernel_time(kernel) has accumulated since the last reset: achieved GB/s =
 * bytes / sum_ms whatever mix of full and partial launch sets was timed            */
int32_t drs_kernel_bytes(drs_handle h, int32_t kernel /*DRS_KERNEL_SLS | _SLS_CLOCK*/, int64_t* bytes);
/* per-workgroup [start, end] device clock ticks (100 MHz) of the last profiled gather
 * launch on `slot`; out holds 2*n_blocks words.  Tuning aid (tools/gather_timeline.py) */
int32_t drs_debug_gather_stamps(drs_handle h, int32_t slot, uint64_t* out, int64_t cap,
                                int64_t* n_blocks);
/* Which kernels served the launch set last enqueued on `slot`, as text: one token per launch in launch
 * order, "name<form>[workgroups, ...]", after a "set[...]" token with the set's size and streams -- e.g.
 * "set[12 queries, 3072 rows, gather on stream_g, mlp on shared] sls_flatc_kernel<16,20,nt>[24576 wg, L=80]
 * stream4_kernel<rows32>[96 wg, 5 layers, 142400 B lds]".  The reference has one operator list per model
 * (models/dlrm_s_caffe2.py:223-389); here the launch form depends on the set's row count, and this is how a
 * caller (tests, tools/dispatch_table.py -> DESIGN.md's dispatch table) sees which one ran.  buf receives at
 * most cap - 1 characters and a terminating 0.  The record is kept only while "dispatch_log" is 1
 * (off by default: DRS_ERR_STATE).                                                                      */
int32_t drs_last_dispatch(drs_handle h, int32_t slot, char* buf, int64_t cap);
/* algorithmic bytes of the gather for a query of `bs` samples of `batch_id`:
 * sum over bags of len*D*4 + len*4 + 4 + D*4  (SURVEY.md 8d / BASELINE.md 2)   */
int32_t drs_gather_bytes(drs_handle h, int32_t batch_id, int32_t bs, int64_t* bytes);

/* ---- multi-GPU: the single collective ---------------------
 * replaces: the parent process merging every engine's responseQueue and computing QPS /
 * tail latency over all of them (DeepRecSys.py:89-135, :168-175).  With one engine process
 * per GPU the per-rank latency histogram and run scalars are combined by ONE grouped RCCL
 * all-reduce over xGMI (~32 KB; nothing on the data path crosses GPUs).
 *   rank 0 calls drs_comm_unique_id and hands the 128 bytes to the other ranks by any
 *   out-of-band means (bench.py: torch.distributed gloo broadcast); every rank then calls
 *   drs_comm_create(id, rank, world, its GPU).
 * RCCL is bound at run time; without it these return DRS_ERR_UNSUPPORTED.          */
#define DRS_COMM_ID_BYTES 128
typedef struct drs_comm_s* drs_comm;
int32_t drs_comm_unique_id(uint8_t* id /*[DRS_COMM_ID_BYTES]*/);
int32_t drs_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device_id,
                        drs_comm* out);
int32_t drs_comm_destroy(drs_comm c);
/* every rank has arrived and its GPU is idle */
int32_t drs_comm_barrier(drs_comm c);
/* in place, result on every rank: hist[nbins] SUM; sum_min_max[0], [1] SUM (query count, sum
 * of latencies), [2] MIN (first completion time), [3] MAX (last completion time / elapsed) */
int32_t drs_stats_allreduce(drs_comm c, int64_t* hist, int32_t nbins, double* sum_min_max /*[4]*/);
const char* drs_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DRS_H_ */

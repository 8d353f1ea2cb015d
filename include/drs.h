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
                         attention unit with its OWN two-layer MLP over Concat(u_i, ad, u_i + ad);
                         atten_out = Sum over the units; top MLP over Concat(profile, atten_out,
                         ad, context); every activation ReLU.  cfg: ln_bot = the unit's widths
                         [3*D, h, D] (arch_mlp_bot "h"), ln_top = [4*D, ...], no dense input      */
  ,
  DRS_MODEL_DIEN = 5  /* models/dien.py:308-432: tables as DIN.  The U behaviour embeddings of a query,
                         Concat'ed [bs, U*D] and Reshape'd (row-major reinterpretation, as the reference
                         does it) to [U, bs, D], run through two caffe2 rnn_cell.BasicRNN layers (tanh,
                         zero initial state, D -> H and H -> H; the FC + Softmax between them is dead in
                         the reference graph and not computed); top MLP (all ReLU) over Concat(last
                         state, profile, ad, context).  cfg: ln_bot = [D, H] (--hidden_size),
                         ln_top = [H + 3*D, ...], no dense input.  H <= 64.                           */
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
  int32_t sparse_dim;           /* arch_sparse_feature_size (D); multiple of 4, <= 256  */
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
 * integer knobs, for A/B measurements inside one process (bench.py --sweep):
 *   "sls_exact"  0 (default) wave-split gather: a wave per bag, rows spread over its lane
 *                groups, wave-wide butterfly at the end (fp32 sum order differs from the
 *                reference: compare with a tolerance) | 1 sequential-order gather,
 *                bit-identical to the Caffe2 CPU SparseLengthsSum (same speed alone, 4-8% slower
 *                beside an MLP launch)
 *   "sls_short_bag" fixed-length batches with at most this many lookups per bag (default
 *                2048 / D: RM3's 20 and W&D's / NCF's 1 qualify, RM1's 80 does not) always take
 *                the sequential-order variant: a lane group per bag instead of a mostly idle
 *                wave per bag | -1 never
 *   "sls_flat"   1 (default) batches whose bags all have one length L >= 2 (every shipped
 *                reference config) run the flat variant when sls_exact is 0: a wave owns 1, 2 or 4
 *                consecutive bags of a sample, reads their indices with one coalesced load and has
 *                ALL of its row loads (up to 20 x 16 B per lane) in flight at once; same tolerance
 *                as the wave-split variant | 0 ring-walk kernels only
 *   "sls_bpw"    bags per wave of the flat variant: 0 (default: as many of 4 | 2 | 1 as divide the
 *                table count and keep a lane at <= 10 loads) | 1 | 2 | 4
 *   "din_fused"  1 (default) DRS_MODEL_DIN with sls_exact 0, D in {32, 64} and hidden width in
 *                {1, 2, 4}: ONE launch gathers the bags, applies the attention units and writes the top
 *                MLP's input row (the [rows, T*D] pooled tensor never exists) | 0 gather launch +
 *                attention launch (what sls_exact 1 always does; bit-identical to the oracle there)
 *   "dien_mfma"  2 (default) DRS_MODEL_DIEN with hidden_size a multiple of 16: the recurrence runs on the
 *                matrix cores, 16 samples per workgroup, one set of waves per layer (layer 2 a step behind
 *                layer 1) | 1 every wave runs both layers of its 16 hidden units | 0 one wave per sample on
 *                the VALU.  Same bits in all three.
 *   "dien_fuse_top" 1 (default) | 0: DRS_MODEL_DIEN, matrix-core recurrence: the top MLP of a workgroup's 16
 *                samples runs in the recurrence's own launch when it fits (<= 4 layers, every input width a
 *                multiple of 4 and <= 256), which then signs off the launch set; 0: a stream-kernel launch
 *                of its own behind it.  Same bits.
 *   "din_s"      samples per workgroup of that launch: 0 (default: 4 | 2 | 1 by launch size) | 1 | 2 | 4
 *                (results do not depend on it)
 *   "gemm_split" 1 (default) | 0: DRS_MODEL_WND / DRS_MODEL_MTWND: when the first top layer runs as a scalar-base
 *                gemm32_kernel (full launch sets), it reads the dense columns of Concat(dense, embeddings) from the
 *                queries' own arrays; 0 (and every other launch form): the dense rows are copied in front of the
 *                embeddings first (copy_rows_multi_kernel).  Same bits.
 *   "din_pipe"   1 (default) | 0: hidden width 1 and launch sets whose bags all have one fixed length <= 3 (din.json)
 *                take the pipelined form of that launch (din_pipe_kernel: the set's indices staged in LDS with one
 *                round trip, two units in flight per lane group); 0: the chained form for every shape.  Same bits.
 *   "sls_nt"     1 (default) | 0: the many-rows-per-bag gather kernels read table rows with non-temporal
 *                loads (rows are read once per launch; same bits either way; the one-lookup models' gather
 *                keeps plain loads: their tables are cache-resident).  "din_nt" 1 (default) | 0: the same for
 *                the fused DIN launch
 *   "sls_uniform" 1 (default) batches whose bags all have one length skip the offset
 *                read (bag b starts at b*L) | 0 always read the staged prefix sums
 *   "mlp_split"  1 (default) a layer with K*N >= "mlp_wide_kn" weights (RM3's 2560x1024)
 *                runs as its own 2-D launch | 0 chain everything that fits LDS
 *   "mlp_wide_kn" that threshold (default 256K weights: RM3's 2560x1024 and 1024x256, W&D's
 *                1376x1024 and 1024x512)
 *   "mlp_fuse"   1 (default) DLRM/"cat": bottom and top MLP of a 16-row slab in ONE launch
 *                | 0 one launch per MLP;  "mlp_fuse_rows": fuse only from this many rows on
 *   "mlp_gemm"   1 (default) stand-alone wide layers run as the register-blocked gemm_kernel
 *                (gemm.hip) | 0 fc_kernel;  "mlp_gemm_tile" 0 (default: by workgroup count)
 *                | 22 | 12 | 21 | 11 forces the per-wave tile shape, 214 = the 2 x 1 shape compiled for TWO
 *                workgroups per CU (ring of two chunks, <= 128 VGPRs, 70 KB of LDS);
 *                "mlp_gemm_2cu" 1 (default for DLRM and W&D) | 0: take that shape by itself whenever
 *                it gives >= 512 workgroups (W&D +5 %, RM3 +5 %; MT-WnD -5 %: off there)
 *                (the full 2 x 2 tile is kept while it still gives 128 workgroups);
 *                "mlp_gemm_tile" 322 | 321 | 312 | 311: gemm32_kernel, the same GEMM on
 *                v_mfma_f32_32x32x2_f32 (four waves, each 2 x 2 | 2 x 1 | 1 x 2 | 1 x 1 tiles of 32 x 32:
 *                workgroup tiles of 128 x 128 .. 64 x 64, operands by ds_read_b128, two workgroups per CU);
 *                "mlp_gemm32" 1 (default) | 0: wide layers whose 128 x 128 tiles number at least
 *                "mlp_gemm32_blocks" (default 512: two workgroups per CU) take the 2 x 2 form (RM3 config 3's
 *                2560 x 1024 layer at 8 192 rows), smaller launches keep gemm_kernel unless
 *                "mlp_gemm32_small" names a gemm32 shape for them (22 | 21 | 12 | 11) that gives at least
 *                "mlp_gemm32_small_blocks" workgroups (defaults: 12 with 256 for MT-WnD and MLP-bound DLRM, 12 with 512 for
 *                W&D, else 0)
 *   "mlp_stream" 2 (default for MT-WnD, NCF) chains run as the weight-tile stream kernel (tiles of all layers requested
 *                six rounds ahead, inputs resident in LDS) when every K % 4 == 0 and the slabs fit, the tiles read from
 *                the layers' PACKED twins (MFMA operand order, built by drs_set_fc) straight into the MFMA operand
 *                registers: no LDS staging of W, a workgroup barrier per layer instead of per 64-k chunk | 4 (default
 *                for DLRM, W&D, DIEN, DIN) stream4_kernel: four waves, every (layer, 64-column-per-wave pass) run by ONE
 *                hand-laid instruction stream (csrc/seg_asm.inc): MFMAs back to back with the weight reloads, operand
 *                prefetch and loop control between them, ring and accumulators in AGPRs under fixed names, the next
 *                segment's first chunk requested while the last one runs ("mlp_rows32" n: its launches of >= n rows take
 *                32 rows per workgroup -- two halves sharing the weight operands; default 8192 for MLP-bound DLRM, 2048
 *                for gather-bound DLRM, else 0 = never) | 1 the first kernel with W staged through LDS | 0 always the
 *                per-layer chain kernel.  Same bits in every form.  (3 was stream3_kernel, removed in round 5: stream4_kernel
 *                serves every launch size it served, faster -- DESIGN.md "Dispatch".)
 *                ("mlp_stream_2cu" 1 (default, except NCF) | 0: "mlp_stream" 2 with a ring of three
 *                register sets instead of six, compiled for 128 VGPRs, so that two of its workgroups
 *                share a CU and overlapping launches interleave on the same SIMDs)
 *   "mlp_preload" 0 (default) | 1 chain kernel only: pull a chain's 16 x K0 input slab into
 *                LDS in one round instead of streaming it per K chunk
 *   "mlp_kc"     chain kernel only: force the K chunk (0 auto | 64 | 128 | 192 | 256)
 *   "mlp_debug"  timing experiments on the stream kernel, honoured by the `make timeline`
 *                build only (bit 0: always fetch the first tile, bit 1: skip the MFMAs;
 *                results are garbage while set)
 *   "shared_stream" how the launch sets of the slots are put on HIP streams:
 *                2 (default) pipelined: every gather on one stream, back to back; the rest of
 *                  each set (MLPs, interaction, completion) behind an event on a second
 *                  stream, so the HBM-bound gather of set i+1 runs beside the latency-bound
 *                  MLP of set i and gathers never overlap each other
 *                  ("mlp_streams" n: alternate the MLP side over n streams; default 1 for
 *                  gather-bound models, one per slot (up to 4) for MLP-bound ones, decided in
 *                  drs_create from MLP FLOP per gathered byte;
 *                  "mlp_layout" 0 (default) the MLP side's streams are dealt by launch set | 1 by
 *                  kernel type: wide-layer GEMM launches go on the GATHER's stream, serialised with it,
 *                  chains on the MLP streams, an event per kernel -- RM3 config 3: the gather runs at
 *                  0.63 instead of 0.26 of the HBM peak, the model 3-10 % slower, so it is not the default)
 *                1 one stream: sets strictly back to back, each kernel has the chip to itself
 *                0 one stream per slot: whole sets overlap freely
 *   "zero_copy_inputs" how drs_forward_inputs' converted inputs (one packed, pinned block per slot:
 *                dense | int32 indices | prefix sums) reach the kernels: 1 read in place over PCIe
 *                (default: no copy; kernel-issued PCIe reads top out near 20-25 GB/s) | 2 ONE DMA
 *                copy of the block's used prefix into its HBM twin, on the job's gather stream
 *                (same throughput at 3 calls in flight, 20 us more latency per query) | 3: 2 for
 *                queries of >= 128 KB, else 1 | 0 one copy per array (first version)
 *   "host_threads" workers of the per-call input pass (int64 -> int32 + ENFORCEs, one table per
 *                work item, the dense rows' copy one more) beside the calling thread:
 *                -1 (default) min(T, 7) | 0 the caller alone | n.  They spin ~50 us after a call and
 *                then sleep; they do not exist until the first per-call-input query.
 *   "launch_thread" 0 (default) | 1: per-call inputs: the caller converts a query's arrays (they are
 *                consumed before the call returns) and hands the HIP calls -- DMA copy, events,
 *                launches -- to a launcher thread; errors of those calls surface at drs_wait.  Cuts
 *                the caller's time per call from 20 to 14 us; throughput is PCIe-bound either way.
 *   "preferred_coalesce" (read only) queries per launch set the engine asks its feeder for: 12
 *                for gather-bound DLRM (the gap between two gather launches is amortised over
 *                more bytes), 8 for DIN, 16 (DRS_MAX_COALESCE) for MLP-bound models
 *   "gather_bound" (read only) 1 for the models whose set period is their gather launch (DLRM with fewer than 20 MLP FLOP
 *                per gathered byte, DIN): where the tables live and the rows' load policy are worth a search
 *   "device"     (read only) the HIP device index the engine was created on
 *   "preferred_slots" (read only) launch sets the engine asks its feeder to keep in flight: 3
 *                (gather | MLP | enqueue), 6 for NCF (one short latency-bound launch per set: 414 k ->
 *                480-492 k queries/s; nothing for the other models, whose p99 only doubles)
 *   "mlp_small_rows" pipelined mode: launch sets of up to this many rows (default 1024; a single
 *                query is 256) put their MLP side on the slot's own stream, so the latency-bound
 *                MLP launches of consecutive small sets overlap each other
 *   "small_piped" 0 (default) | 1: ... and their gather on the shared gather stream instead of the slot's own (one query
 *                per set, 6 sets in flight: 57 k -> 64 k queries/s at p99 0.115 ms; 3 in flight: 53 k -> 47 k)
 *   "mlp_early"  0 (default) | 1: a small set (<= 512 rows) of staged DLRM queries whose bottom + top MLP is one plain
 *                stream4_kernel launch starts that launch beside its gather: prologue and bottom chain run, then the
 *                launch polls a per-slot flag (a stream-ordered write behind the gather) before it fetches the pooled
 *                rows.  Same bits; measured slower with HIP's stream-ordered write (a kernel of its own), see
 *                profiles/r05_single_query/README.md
 *   "table_placement" where the table arena lives.  A multi-gigabyte allocation's place in HBM moves the gather by
 *                up to 6 % and stays for the allocation's lifetime; a feeder that has staged its input sets can
 *                try a few: -1 = copy the tables into one more allocation and use that one (the earlier ones stay
 *                allocated; DRS_ERR_OOM, nothing changed, when one more copy would take more than a quarter of the
 *                free memory or 256 exist) | k >= 0 = use candidate k | -2 = free every candidate but the one in use.
 *                Reading it gives the index in use, "table_placements" (read only) the number of candidates,
 *                "table_bytes" (read only) the size of one.  Freeing gigabytes has a price of its own: the runtime's
 *                copy-engine transfers (and, by a per cent or two, the gather) are slower for the rest of the process
 *                after it, so a feeder may prefer to leave small losers allocated until drs_destroy.
 *                drs_set_table / drs_fill_table_uniform drop the candidates not in use (they would be stale).
 *                Results never depend on it.  (DLRM_Net.tune_table_placement times each with the model's own sets.)
 *   "table_alloc" how the NEXT arena is built (a "table_placement" -1 candidate): 0 hipMalloc | 1 the virtual-memory
 *                API -- physical memory created in handles of "table_vmm_chunk" bytes (-1, the default: 1 GiB handles for
 *                arenas of at least 1 GiB, one handle for smaller ones; 0: one handle) and mapped into an address range
 *                aligned to "table_vmm_align" bytes (0: 2 MiB) | 2 hipDeviceMallocContiguous, best effort.
 *                "table_spacer" n: n bytes of device memory are taken in 1 GiB pieces and never mapped, so that the next
 *                candidate comes from further on in HBM ("table_placement" -2 and drs_destroy give them back).
 *                "table_address" (read only): the arena's address.  What moves the gather is WHICH physical gigabytes
 *                hold the tables x the load policy ("sls_nt"); DLRM_Net.tune_table_placement searches both.
 *                (The lab's instruments -- memory probes, arenas moved between address ranges or built from a probed
 *                pool, CU masks / priorities of the streams -- exist in the lab build only: make lab-lib, -DDRS_LAB.)
 *   "out_dma"    bytes (default 1 572 864; 0 = never): with "zero_copy" 1, launch sets with at least this many bytes of
 *                outputs hand them over by a copy-engine transfer queued behind the last kernel and a
 *                stream-ordered write of the completion flag behind that (MT-WnD's 2 MB per 16-query set);
 *                smaller sets by the last workgroup's own system-scope stores
 *   "zero_copy"  1 (default) last kernel writes outputs + completion flag into
 *                host-mapped pinned memory (no D2H copy, no stream sync) | 0 memcpy
 * unknown key -> DRS_ERR_BAD_ARG.  Options belong to the handle: two engines in one process
 * (the mixed-model accelerator engine) keep their own values.                    */
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
 * most cap - 1 characters and a terminating 0.                                                        */
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

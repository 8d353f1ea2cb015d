/*
 * drs_oracle.c -- CPU restatement of the DeepRecSys hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle, not a product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * shipped engine (deeprecsys_amd/) never links, imports or calls it and fails
 * loudly when its HIP library is missing.
 *
 * What is restated (paths into the reference tree, harvard-acc/DeepRecSys):
 *   - the operator graph emitted by models/dlrm_s_caffe2.py:367-389
 *     (create_emb :281-329, create_mlp :223-279, create_interactions :331-365),
 *     models/wide_and_deep.py:282-305, models/ncf.py:317-346,
 *     models/multi_task_wnd.py:286-316, models/din.py:247-330 and
 *     models/dien.py:308-432 (rnn_cell.BasicRNN = caffe2.python.rnn_cell's
 *     BasicRNNCell: i2h FC over all steps, per step gates_t FC of the previous
 *     state, Sum, Tanh);
 *   - the Caffe2 operator arithmetic those builders invoke.  Caffe2 is a
 *     third-party dependency that is NOT in the reference tree (it ships inside
 *     torch==1.4.0+cu92, build/pip_requirements.txt:28), so the operator
 *     semantics below follow Caffe2's published operator schemas:
 *       SparseLengthsSum: OUT[i] = sum_{j in segment i} DATA[INDICES[j]], fp32,
 *         rows accumulated sequentially in index order (perfkernels
 *         EmbeddingLookup: acc = fma(1.0f, row, acc)); ENFORCE 0<=idx<N and
 *         sum(LENGTHS)==len(INDICES); empty segment -> zeros.
 *       FC: Y = X * W^T + b  (W is [N,K], not transposed).
 *       Relu / Sigmoid: max(x,0) / 1/(1+exp(-x)).
 *       Concat(axis=1[,add_axis=1]), BatchMatMul(trans_b=1), Flatten(axis=1),
 *       BatchGather(tril indices), Sum.
 *
 * Pinning status ("parity partially pinned", see DESIGN.md section Oracle):
 *   pinned  : graph/op list, blob dtypes, weights, inputs, tril indices -- against
 *             fixtures produced by importing the reference's own builders and
 *             data generator here (tools/gen_golden.py, tests/golden/ npz files);
 *   pinned  : SparseLengthsSum / FC arithmetic -- against torch-CPU
 *             embedding_bag(sum) / addmm, the lineal descendants of the Caffe2
 *             kernels (tests/test_oracle.py);
 *   pinned  : the BasicRNN recurrence -- against torch.nn.RNN in fp64
 *             (tests/test_oracle.py::test_oracle_dien_matches_torch_rnn_fp64);
 *   UNPINNED: outputs of the reference executing on Caffe2 itself -- Caffe2 is
 *             not installable here, so no reference-run output vector exists;
 *   UNPINNED: DIEN's recurrent WEIGHTS in a live reference run -- models/dien.py
 *             feeds numpy values (:318-331) that create() then overwrites by
 *             running Caffe2's param_init_net (:528, XavierFill from Caffe2's own
 *             RNG); the fixture holds the fed values and the list of re-initialised
 *             blobs.
 *
 * FC accumulation order: the reference's sgemm order (MKL/Eigen) is not
 * observable, so the oracle fixes a definite one -- a k-ordered fp32 fma chain
 * from 0, bias added after -- which is also exactly what the gfx950
 * v_mfma_f32_16x16x4_f32 path computes, making GPU-vs-oracle comparisons
 * bitwise up to the final expf.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_BAD_ARG (-1)
#define ORC_ERR_OOM (-2)
#define ORC_ERR_INDEX_RANGE (-4)
#define ORC_ERR_LENGTHS_SUM (-5)
#define ORC_ERR_UNSUPPORTED (-7)

enum { ORC_MODEL_DLRM = 0, ORC_MODEL_WND = 1, ORC_MODEL_NCF = 2, ORC_MODEL_MTWND = 3, ORC_MODEL_DIN = 4, ORC_MODEL_DIEN = 5 };
enum { ORC_INTERACT_DOT = 0, ORC_INTERACT_CAT = 1 };
enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_SIGMOID = 2 };

static void set_threads(int32_t nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
}

int32_t orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* counter-based table fill shared bit-exactly with the HIP engine
 * (include/drs.h drs_fill_table_uniform): value = f(seed, table, element). */
static inline float fill_value(uint64_t seed, int32_t t, uint64_t i, float lo, float hi) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + ((uint64_t)(uint32_t)t << 40) + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
  return fmaf(u, hi - lo, lo);
}

float orc_fill_value(uint64_t seed, int32_t t, uint64_t i, float lo, float hi) {
  return fill_value(seed, t, i, lo, hi);
}

int32_t orc_fill_table_uniform(float* W, int64_t rows, int32_t D, int32_t t, float lo, float hi,
                               uint64_t seed, int32_t nthreads) {
  if (!W || rows < 0 || D <= 0) return ORC_ERR_BAD_ARG;
  set_threads(nthreads);
  int64_t n = rows * (int64_t)D;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) W[i] = fill_value(seed, t, (uint64_t)i, lo, hi);
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* SparseLengthsSum  (call site models/dlrm_s_caffe2.py:317-325)              */
static int32_t sls_check(int64_t rows, const int64_t* idx64, const int32_t* idx32,
                         const int32_t* len, int64_t n_bags, int64_t n_idx) {
  int64_t total = 0;
  for (int64_t b = 0; b < n_bags; ++b) {
    if (len[b] < 0) return ORC_ERR_LENGTHS_SUM;
    total += len[b];
  }
  if (total != n_idx) return ORC_ERR_LENGTHS_SUM;
  for (int64_t j = 0; j < n_idx; ++j) {
    int64_t v = idx64 ? idx64[j] : (int64_t)idx32[j];
    if (v < 0 || v >= rows) return ORC_ERR_INDEX_RANGE;
  }
  return ORC_OK;
}

static int32_t sls_impl(const float* W, int64_t rows, int32_t D, const int64_t* idx64,
                        const int32_t* idx32, const int32_t* len, int64_t n_bags, int64_t n_idx,
                        float* out, int64_t out_stride, int32_t nthreads) {
  if (!W || !len || !out || (!idx64 && !idx32 && n_idx > 0) || D <= 0 || n_bags < 0)
    return ORC_ERR_BAD_ARG;
  int32_t rc = sls_check(rows, idx64, idx32, len, n_bags, n_idx);
  if (rc != ORC_OK) return rc;
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_bags + 1));
  if (!off) return ORC_ERR_OOM;
  off[0] = 0;
  for (int64_t b = 0; b < n_bags; ++b) off[b + 1] = off[b] + len[b];
  set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t b = 0; b < n_bags; ++b) {
    float* o = out + b * out_stride;
    for (int32_t d = 0; d < D; ++d) o[d] = 0.0f;
    for (int64_t j = off[b]; j < off[b + 1]; ++j) {
      int64_t r = idx64 ? idx64[j] : (int64_t)idx32[j];
      const float* row = W + r * (int64_t)D;
      if (j + 4 < off[b + 1]) {
        int64_t r4 = idx64 ? idx64[j + 4] : (int64_t)idx32[j + 4];
        __builtin_prefetch(W + r4 * (int64_t)D);
      }
      /* sequential in index order, one plain fp32 add per column */
      for (int32_t d = 0; d < D; ++d) o[d] = o[d] + row[d];
    }
  }
  free(off);
  return ORC_OK;
}

int32_t orc_sls_i64(const float* W, int64_t rows, int32_t D, const int64_t* idx,
                    const int32_t* len, int64_t n_bags, int64_t n_idx, float* out,
                    int32_t nthreads) {
  return sls_impl(W, rows, D, idx, NULL, len, n_bags, n_idx, out, D, nthreads);
}

int32_t orc_sls_i32(const float* W, int64_t rows, int32_t D, const int32_t* idx,
                    const int32_t* len, int64_t n_bags, int64_t n_idx, float* out,
                    int32_t nthreads) {
  return sls_impl(W, rows, D, NULL, idx, len, n_bags, n_idx, out, D, nthreads);
}

/* ------------------------------------------------------------------------- */
/* FC + Relu|Sigmoid  (call site models/dlrm_s_caffe2.py:258-272)             */
static inline float act_apply(float v, int32_t act) {
  if (act == ORC_ACT_RELU) return v > 0.0f ? v : 0.0f;
  if (act == ORC_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

/* y[m, n] = act( (k-ordered fma chain from 0 of x[m,k]*W[n,k]) + b[n] )        */
/* Wt_pre (optional): W already transposed to [K][N] by the caller (the model object does it
 * once at build time -- a CPU engine would not re-transpose its weights per query); NULL =
 * transpose here, per call (operator-level entry point).                               */
static int32_t fc_impl(const float* x, int64_t M, int32_t K, int64_t ldx, const float* W,
                       const float* Wt_pre, const float* b, int32_t N, int32_t act, float* y,
                       int64_t ldy, int32_t nthreads) {
  if (!x || !W || !y || M < 0 || K <= 0 || N <= 0) return ORC_ERR_BAD_ARG;
  /* W as [K][N] so the n-loop vectorises while every output keeps its own k-ordered chain */
  float* Wt_own = NULL;
  const float* Wt = Wt_pre;
  if (!Wt) {
    Wt_own = (float*)malloc(sizeof(float) * (size_t)K * (size_t)N);
    if (!Wt_own) return ORC_ERR_OOM;
    for (int32_t n = 0; n < N; ++n)
      for (int32_t k = 0; k < K; ++k) Wt_own[(size_t)k * N + n] = W[(size_t)n * K + k];
    Wt = Wt_own;
  }
  set_threads(nthreads);
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      const float* xr = x + m * ldx;
      for (int32_t n = 0; n < N; ++n) acc[n] = 0.0f;
      for (int32_t k = 0; k < K; ++k) {
        const float xv = xr[k];
        const float* wr = Wt + (size_t)k * N;
        for (int32_t n = 0; n < N; ++n) acc[n] = fmaf(xv, wr[n], acc[n]);
      }
      float* yr = y + m * ldy;
      for (int32_t n = 0; n < N; ++n) yr[n] = act_apply(b ? acc[n] + b[n] : acc[n], act);
    }
    free(acc);
  }
  free(Wt_own);
  return ORC_OK;
}

int32_t orc_fc(const float* x, int64_t M, int32_t K, const float* W, const float* b, int32_t N,
               int32_t act, float* y, int32_t nthreads) {
  return fc_impl(x, M, K, K, W, NULL, b, N, act, y, N, nthreads);
}

/* ------------------------------------------------------------------------- */
/* dot interaction (models/dlrm_s_caffe2.py:334-354, tril :529-535):
 * T [B,F,D] -> R [B, D + P], P = F(F-1)/2 (+F with itself);
 * R[b, 0:D] = T[b,0,:] (dense_out), then Z[b,i,j] = <T[b,i,:], T[b,j,:]> for
 * i in 0..F-1, j in 0..i-1(+itself), row-major -- the BatchGather order.
 * The dot is a d-ordered fma chain from 0 (matches the MFMA path).            */
int32_t orc_interact_dot(const float* T, int64_t B, int32_t F, int32_t D, int32_t itself,
                         float* R, int32_t nthreads) {
  if (!T || !R || B < 0 || F <= 0 || D <= 0) return ORC_ERR_BAD_ARG;
  const int32_t off = itself ? 1 : 0;
  const int32_t P = F * (F - 1) / 2 + (itself ? F : 0);
  const int64_t ldr = (int64_t)D + P;
  set_threads(nthreads);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    const float* t = T + b * (int64_t)F * D;
    float* r = R + b * ldr;
    for (int32_t d = 0; d < D; ++d) r[d] = t[d];
    int32_t p = 0;
    for (int32_t i = 0; i < F; ++i)
      for (int32_t j = 0; j < i + off; ++j) {
        float acc = 0.0f;
        for (int32_t d = 0; d < D; ++d) acc = fmaf(t[(int64_t)i * D + d], t[(int64_t)j * D + d], acc);
        r[D + p++] = acc;
      }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* whole forward.  Widths:
 *   DLRM: ln_bot = arch_mlp_bot, ln_top = [num_int] + arch_mlp_top
 *         (models/dlrm_s_caffe2.py:415-430); sigmoid on top layer sigmoid_top.
 *   WND : n_bot == 1 (raw dense, width ln_bot[0]); R = [dense | emb_0 | ...]
 *         (models/wide_and_deep.py:271-305), num_int = T*D + ln_bot[0] (:345).
 *   NCF : tables 0,1 -> mf = Sum; tables 2,3 -> Concat -> MLP over ln_top
 *         (all Relu, models/ncf.py:150-196); then Concat(mf, mlp_out) ->
 *         FC(final_W[m_fin, D + ln_top[-1]]) + Relu (:317-346).
 * R_out (optional) receives the interaction tensor fed to the top MLP.        */
typedef struct orc_model {
  int32_t model_kind, T, D;
  const int64_t* rows;
  const float* const* tables;
  int32_t n_bot;
  const int32_t* ln_bot;
  const float* const* bot_W;
  const float* const* bot_b;
  int32_t n_top;
  const int32_t* ln_top;
  const float* const* top_W;
  const float* const* top_b;
  const float* final_W;
  const float* final_b;
  int32_t final_m;
  int32_t interaction_op, itself, sigmoid_top;
  /* optional: the FC weights once more, transposed to [K][N] at model build (NULL = per call) */
  const float* const* bot_Wt;
  const float* const* top_Wt;
  const float* final_Wt;
  /* MT-WnD (models/multi_task_wnd.py:286-316): num_tasks heads of widths ln_task over the
   * shared top MLP's (all-ReLU) output; task_W / task_b / task_Wt hold the heads' layers back
   * to back, [num_tasks * (n_task - 1)]; task_sigmoid = 1-based head layer that gets Sigmoid */
  int32_t n_task;
  const int32_t* ln_task;
  int32_t num_tasks, task_sigmoid;
  const float* const* task_W;
  const float* const* task_b;
  const float* const* task_Wt;
  /* DIN (models/din.py:247-330): tables = [user profile | U behaviour tables | candidate ad |
   * context]; per behaviour table an attention unit with ITS OWN MLP of widths ln_att
   * (ln_att[0] == 3*D: Concat(u_i, ad, Sum(u_i, ad)); ln_att[-1] == D; all ReLU); att_W / att_b
   * hold the units' layers back to back, [U * (n_att - 1)]; atten_out = Sum of the units' outputs
   * in unit order; top input = Concat(user profile, atten_out, ad, context) [B, 4*D]; top MLP all
   * ReLU (din.py:186, no Sigmoid anywhere)                                                    */
  int32_t n_att;
  const int32_t* ln_att;
  const float* const* att_W;
  const float* const* att_b;
  /* DIEN (models/dien.py:308-432): tables as DIN.  The U behaviour embeddings of a query are
   * Concat'ed to [B, U*D] and Reshape'd -- a row-major REINTERPRETATION, not a transpose -- to
   * [U, B, D] (:316-320): step t of "sample" b is embedding n % U of sample n / U, n = t*B + b.
   * Two caffe2 rnn_cell.BasicRNN layers (tanh; h_t = tanh((gates_t_w h_{t-1} + gates_t_b) +
   * (i2h_w x_t + i2h_b)), zero initial state; every step is inside seq_lengths (:495-497,89-90)),
   * D -> H and H -> H; the second reads the first's states (the FC + Softmax between them is dead:
   * the Sum after it overwrites its output with a copy of the states, :336-348).  Top input =
   * Concat(last state of layer 2, user profile, candidate ad, context) [B, H + 3*D] (:411-421); top
   * MLP all ReLU.  rnn_w = {i2h_w, i2h_b, gates_t_w, gates_t_b} of layer 1 then of layer 2.      */
  int32_t rnn_hidden;
  const float* rnn_w[8];
} orc_model;

static int32_t mlp_chain(const float* in, int64_t B, int64_t ld_in, int32_t n_l,
                         const int32_t* ln, const float* const* Ws, const float* const* Wts,
                         const float* const* bs,
                         int32_t sigmoid_layer, float* out_last, int64_t ld_out,
                         int32_t nthreads) {
  /* layers 1..n_l-1; activation Relu except Sigmoid at sigmoid_layer
   * (models/dlrm_s_caffe2.py:268-272) */
  const float* cur = in;
  int64_t ld = ld_in;
  float* tmp[2] = {NULL, NULL};
  int32_t maxw = 0;
  for (int32_t i = 0; i < n_l; ++i)
    if (ln[i] > maxw) maxw = ln[i];
  for (int32_t i = 0; i < 2; ++i) {
    tmp[i] = (float*)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1) * (size_t)maxw);
    if (!tmp[i]) {
      free(tmp[0]);
      return ORC_ERR_OOM;
    }
  }
  int32_t rc = ORC_OK;
  for (int32_t i = 1; i < n_l && rc == ORC_OK; ++i) {
    const int last = (i == n_l - 1);
    float* dst = last ? out_last : tmp[i & 1];
    int64_t ldd = last ? ld_out : ln[i];
    rc = fc_impl(cur, B, ln[i - 1], ld, Ws[i - 1], Wts ? Wts[i - 1] : NULL, bs[i - 1], ln[i],
                 i == sigmoid_layer ? ORC_ACT_SIGMOID : ORC_ACT_RELU, dst, ldd, nthreads);
    cur = dst;
    ld = ldd;
  }
  free(tmp[0]);
  free(tmp[1]);
  return rc;
}

int32_t orc_forward(const orc_model* m, int32_t bs, const float* dense,
                    const int64_t* const* idx, const int64_t* n_idx,
                    const int32_t* const* len, float* out, float* R_out, int32_t nthreads) {
  if (!m || !out || bs < 0 || !idx || !len || !n_idx) return ORC_ERR_BAD_ARG;
  const int32_t T = m->T, D = m->D;
  const int64_t B = bs;
  int32_t rc = ORC_OK;
  if (B == 0) return ORC_OK;

  if (m->model_kind == ORC_MODEL_NCF) {
    if (T != 4) return ORC_ERR_BAD_ARG;
    const int32_t wl = m->ln_top[m->n_top - 1];
    float* emb = (float*)malloc(sizeof(float) * (size_t)B * 4 * D);
    float* cat = (float*)malloc(sizeof(float) * (size_t)B * (size_t)(D + wl));
    float* mlp_in = (float*)malloc(sizeof(float) * (size_t)B * 2 * D);
    if (!emb || !cat || !mlp_in) {
      free(emb); free(cat); free(mlp_in);
      return ORC_ERR_OOM;
    }
    for (int32_t t = 0; t < 4 && rc == ORC_OK; ++t)
      rc = sls_impl(m->tables[t], m->rows[t], D, idx[t], NULL, len[t], B, n_idx[t],
                    emb + (size_t)t * B * D, D, nthreads);
    if (rc == ORC_OK) {
      for (int64_t b = 0; b < B; ++b)
        for (int32_t d = 0; d < D; ++d) {
          /* Sum(mf_sls0, mf_sls1)  (models/ncf.py:301-305) */
          cat[b * (D + wl) + d] = emb[(0 * B + b) * D + d] + emb[(1 * B + b) * D + d];
          mlp_in[b * 2 * D + d] = emb[(2 * B + b) * D + d];
          mlp_in[b * 2 * D + D + d] = emb[(3 * B + b) * D + d];
        }
      rc = mlp_chain(mlp_in, B, 2 * D, m->n_top, m->ln_top, m->top_W, m->top_Wt, m->top_b, -1,
                     cat + D, D + wl, nthreads);
    }
    if (rc == ORC_OK && R_out) memcpy(R_out, cat, sizeof(float) * (size_t)B * (size_t)(D + wl));
    if (rc == ORC_OK)
      rc = fc_impl(cat, B, D + wl, D + wl, m->final_W, m->final_Wt, m->final_b, m->final_m, ORC_ACT_RELU, out,
                   m->final_m, nthreads);
    free(emb); free(cat); free(mlp_in);
    return rc;
  }

  if (m->model_kind == ORC_MODEL_DIN) {
    const int32_t U = T - 3, per = m->n_att - 1;
    if (T < 4 || m->n_att < 2 || m->ln_att[0] != 3 * D || m->ln_att[m->n_att - 1] != D || m->ln_top[0] != 4 * D)
      return ORC_ERR_BAD_ARG;
    const int64_t ldE = (int64_t)T * D;
    float* E = (float*)malloc(sizeof(float) * (size_t)B * (size_t)ldE);
    float* X = (float*)malloc(sizeof(float) * (size_t)B * 3 * D);
    float* Y = (float*)malloc(sizeof(float) * (size_t)B * D);
    float* Rin = (float*)malloc(sizeof(float) * (size_t)B * 4 * D);
    if (!E || !X || !Y || !Rin) { free(E); free(X); free(Y); free(Rin); return ORC_ERR_OOM; }
    for (int32_t t = 0; t < T && rc == ORC_OK; ++t)
      rc = sls_impl(m->tables[t], m->rows[t], D, idx[t], NULL, len[t], B, n_idx[t], E + (int64_t)t * D, ldE, nthreads);
    for (int32_t i = 0; i < U && rc == ORC_OK; ++i) {
      for (int64_t b = 0; b < B; ++b) {
        const float* u = E + b * ldE + (int64_t)(1 + i) * D;
        const float* ad = E + b * ldE + (int64_t)(T - 2) * D;
        float* x = X + b * 3 * D;
        for (int32_t d = 0; d < D; ++d) { x[d] = u[d]; x[D + d] = ad[d]; x[2 * D + d] = u[d] + ad[d]; }   /* Sum, Concat (:262-271) */
      }
      rc = mlp_chain(X, B, 3 * D, m->n_att, m->ln_att, m->att_W + (size_t)i * per, NULL, m->att_b + (size_t)i * per, -1, Y, D, nthreads);
      if (rc == ORC_OK)
        for (int64_t b = 0; b < B; ++b)
          for (int32_t d = 0; d < D; ++d) {
            float* z = Rin + b * 4 * D + D + d;          /* atten_out = Sum(fc_outs), in unit order (:283) */
            *z = i == 0 ? Y[b * D + d] : *z + Y[b * D + d];
          }
    }
    if (rc == ORC_OK)
      for (int64_t b = 0; b < B; ++b) {
        memcpy(Rin + b * 4 * D, E + b * ldE, sizeof(float) * D);                                  /* user profile */
        memcpy(Rin + b * 4 * D + 2 * D, E + b * ldE + (int64_t)(T - 2) * D, sizeof(float) * 2 * D); /* candidate ad, context */
      }
    if (rc == ORC_OK && R_out) memcpy(R_out, Rin, sizeof(float) * (size_t)B * 4 * D);
    if (rc == ORC_OK)
      rc = mlp_chain(Rin, B, 4 * D, m->n_top, m->ln_top, m->top_W, m->top_Wt, m->top_b, -1, out, m->ln_top[m->n_top - 1], nthreads);
    free(E); free(X); free(Y); free(Rin);
    return rc;
  }

  if (m->model_kind == ORC_MODEL_DIEN) {
    const int32_t U = T - 3, Hh = m->rnn_hidden, wr = Hh + 3 * D;
    if (T < 4 || Hh < 1 || m->ln_top[0] != wr) return ORC_ERR_BAD_ARG;
    for (int i = 0; i < 8; ++i) if (!m->rnn_w[i]) return ORC_ERR_BAD_ARG;
    const int64_t ldE = (int64_t)T * D;
    float* E = (float*)malloc(sizeof(float) * (size_t)B * (size_t)ldE);
    float* X = (float*)malloc(sizeof(float) * (size_t)B * D);
    float* A = (float*)malloc(sizeof(float) * (size_t)B * Hh);
    float* G = (float*)malloc(sizeof(float) * (size_t)B * Hh);
    float* h0 = (float*)calloc((size_t)B * Hh, sizeof(float));
    float* h1 = (float*)calloc((size_t)B * Hh, sizeof(float));
    float* Rin = (float*)malloc(sizeof(float) * (size_t)B * wr);
    if (!E || !X || !A || !G || !h0 || !h1 || !Rin) { free(E); free(X); free(A); free(G); free(h0); free(h1); free(Rin); return ORC_ERR_OOM; }
    for (int32_t t = 0; t < T && rc == ORC_OK; ++t)
      rc = sls_impl(m->tables[t], m->rows[t], D, idx[t], NULL, len[t], B, n_idx[t], E + (int64_t)t * D, ldE, nthreads);
    for (int32_t t = 0; t < U && rc == ORC_OK; ++t) {
      for (int64_t b = 0; b < B; ++b) {                       /* the Reshape (:319-320) */
        const int64_t n = (int64_t)t * B + b;
        memcpy(X + b * D, E + (n / U) * ldE + (1 + n % U) * D, sizeof(float) * D);
      }
      /* layer 1: i2h FC, gates FC on the previous state, Sum(gates, i2h), Tanh */
      rc = fc_impl(X, B, D, D, m->rnn_w[0], NULL, m->rnn_w[1], Hh, ORC_ACT_NONE, A, Hh, nthreads);
      if (rc == ORC_OK) rc = fc_impl(h0, B, Hh, Hh, m->rnn_w[2], NULL, m->rnn_w[3], Hh, ORC_ACT_NONE, G, Hh, nthreads);
      if (rc != ORC_OK) break;
      for (int64_t i = 0; i < B * Hh; ++i) h0[i] = tanhf(G[i] + A[i]);
      /* layer 2 on layer 1's new state */
      rc = fc_impl(h0, B, Hh, Hh, m->rnn_w[4], NULL, m->rnn_w[5], Hh, ORC_ACT_NONE, A, Hh, nthreads);
      if (rc == ORC_OK) rc = fc_impl(h1, B, Hh, Hh, m->rnn_w[6], NULL, m->rnn_w[7], Hh, ORC_ACT_NONE, G, Hh, nthreads);
      if (rc != ORC_OK) break;
      for (int64_t i = 0; i < B * Hh; ++i) h1[i] = tanhf(G[i] + A[i]);
    }
    if (rc == ORC_OK)
      for (int64_t b = 0; b < B; ++b) {
        memcpy(Rin + b * wr, h1 + b * Hh, sizeof(float) * Hh);
        memcpy(Rin + b * wr + Hh, E + b * ldE, sizeof(float) * D);                                   /* user profile */
        memcpy(Rin + b * wr + Hh + D, E + b * ldE + (int64_t)(T - 2) * D, sizeof(float) * 2 * D);    /* candidate ad, context */
      }
    if (rc == ORC_OK && R_out) memcpy(R_out, Rin, sizeof(float) * (size_t)B * wr);
    if (rc == ORC_OK)
      rc = mlp_chain(Rin, B, wr, m->n_top, m->ln_top, m->top_W, m->top_Wt, m->top_b, -1, out, m->ln_top[m->n_top - 1], nthreads);
    free(E); free(X); free(A); free(G); free(h0); free(h1); free(Rin);
    return rc;
  }

  /* DLRM / WND: build T0 = [dense_out | emb_0 | ... | emb_{T-1}] */
  const int32_t w0 = m->ln_bot[m->n_bot - 1]; /* dense_out width (== D for DLRM) */
  const int64_t ldc = (int64_t)w0 + (int64_t)T * D;
  float* cat = (float*)malloc(sizeof(float) * (size_t)B * (size_t)ldc);
  if (!cat) return ORC_ERR_OOM;
  for (int32_t t = 0; t < T && rc == ORC_OK; ++t)
    rc = sls_impl(m->tables[t], m->rows[t], D, idx[t], NULL, len[t], B, n_idx[t],
                  cat + w0 + (int64_t)t * D, ldc, nthreads);
  if (rc == ORC_OK) {
    if (m->n_bot > 1) {
      rc = mlp_chain(dense, B, m->ln_bot[0], m->n_bot, m->ln_bot, m->bot_W, m->bot_Wt, m->bot_b, -1, cat, ldc,
                     nthreads);
    } else {
      for (int64_t b = 0; b < B; ++b) memcpy(cat + b * ldc, dense + b * w0, sizeof(float) * w0);
    }
  }
  const float* top_in = cat;
  int64_t ld_top = ldc;
  float* R = NULL;
  if (rc == ORC_OK && m->model_kind == ORC_MODEL_DLRM && m->interaction_op == ORC_INTERACT_DOT) {
    if (w0 != D) {
      free(cat);
      return ORC_ERR_BAD_ARG;
    }
    const int32_t F = T + 1;
    const int32_t P = F * (F - 1) / 2 + (m->itself ? F : 0);
    R = (float*)malloc(sizeof(float) * (size_t)B * (size_t)(D + P));
    if (!R) {
      free(cat);
      return ORC_ERR_OOM;
    }
    rc = orc_interact_dot(cat, B, F, D, m->itself, R, nthreads);
    top_in = R;
    ld_top = D + P;
  }
  if (rc == ORC_OK && ld_top != m->ln_top[0]) rc = ORC_ERR_BAD_ARG;
  if (rc == ORC_OK && R_out) memcpy(R_out, top_in, sizeof(float) * (size_t)B * (size_t)ld_top);
  if (rc == ORC_OK && m->model_kind == ORC_MODEL_MTWND) {
    /* shared top (no sigmoid, :301) -> every head reads it; outputs side by side */
    const int32_t wt = m->ln_top[m->n_top - 1], wo = m->ln_task[m->n_task - 1];
    const int32_t per = m->n_task - 1;
    if (m->n_task < 2 || m->num_tasks < 1 || m->ln_task[0] != wt) rc = ORC_ERR_BAD_ARG;
    float* shared = rc == ORC_OK ? (float*)malloc(sizeof(float) * (size_t)B * (size_t)wt) : NULL;
    if (rc == ORC_OK && !shared) rc = ORC_ERR_OOM;
    if (rc == ORC_OK)
      rc = mlp_chain(top_in, B, ld_top, m->n_top, m->ln_top, m->top_W, m->top_Wt, m->top_b, -1, shared, wt, nthreads);
    for (int32_t k = 0; k < m->num_tasks && rc == ORC_OK; ++k)
      rc = mlp_chain(shared, B, wt, m->n_task, m->ln_task, m->task_W + (size_t)k * per,
                     m->task_Wt ? m->task_Wt + (size_t)k * per : NULL, m->task_b + (size_t)k * per,
                     m->task_sigmoid, out + (size_t)k * wo, (int64_t)m->num_tasks * wo, nthreads);
    free(shared);
  } else if (rc == ORC_OK)
    rc = mlp_chain(top_in, B, ld_top, m->n_top, m->ln_top, m->top_W, m->top_Wt, m->top_b, m->sigmoid_top, out,
                   m->ln_top[m->n_top - 1], nthreads);
  free(R);
  free(cat);
  return rc;
}

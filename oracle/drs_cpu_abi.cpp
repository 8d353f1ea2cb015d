// TEST INFRASTRUCTURE -- not part of the product.
//
// CPU restatement of the C ABI of include/drs.h on top of the oracle (drs_oracle.c): the
// same entry points, argument meaning, validation and status codes as libdrs_hip.so
// (deeprecsys_amd/csrc/engine.hip), with orc_forward() doing the arithmetic.  It exists so
// that
//   * the host logic ABOVE the ABI (deeprecsys_amd/_native.py, dlrm_s_hip.py wrappers, the
//     accelerator engine's request loop) is exercised by the CPU test suite, against the
//     golden fixtures, without a GPU (tests/conftest.py `cpu_abi` fixture monkeypatches the
//     binding inside the test process only);
//   * SURVEY.md 8(b)-3's "CPU backend implementing the identical ABI" has a concrete form.
// Only tests/ load this library.  The product binding (deeprecsys_amd/_native.py) has no
// path to it: without libdrs_hip.so / without a GPU the product fails loudly.
//
// Built by oracle/Makefile into oracle/_build/libdrs_cpu.so together with drs_oracle.c.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/drs.h"

extern "C" {
// drs_oracle.c
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
  const float* const* bot_Wt;
  const float* const* top_Wt;
  const float* final_Wt;
  int32_t n_task;
  const int32_t* ln_task;
  int32_t num_tasks, task_sigmoid;
  const float* const* task_W;
  const float* const* task_b;
  const float* const* task_Wt;
  int32_t n_att;
  const int32_t* ln_att;
  const float* const* att_W;
  const float* const* att_b;
  int32_t rnn_hidden;
  const float* rnn_w[8];
} orc_model;
int32_t orc_forward(const orc_model* m, int32_t bs, const float* dense, const int64_t* const* idx,
                    const int64_t* n_idx, const int32_t* const* len, float* out, float* R_out,
                    int32_t nthreads);
int32_t orc_fill_table_uniform(float* W, int64_t rows, int32_t D, int32_t t, float lo, float hi,
                               uint64_t seed, int32_t nthreads);
int32_t orc_sls_i32(const float* W, int64_t rows, int32_t D, const int32_t* idx, const int32_t* len,
                    int64_t n_bags, int64_t n_idx, float* out, int32_t nthreads);
int32_t orc_fc(const float* x, int64_t M, int32_t K, const float* W, const float* b, int32_t N,
               int32_t act, float* y, int32_t nthreads);
int32_t orc_interact_dot(const float* T, int64_t B, int32_t F, int32_t D, int32_t itself, float* R,
                         int32_t nthreads);
}

namespace {

thread_local std::string g_create_error;

struct Mlp {
  std::vector<int32_t> ln;
  std::vector<std::vector<float>> W, b;
  std::vector<bool> set;
};

struct Batch {
  std::vector<float> dense;                  // [n, m_den]
  std::vector<std::vector<int64_t>> idx;     // [T][n_idx]
  std::vector<std::vector<int32_t>> len;     // [T][n]
  int32_t n_samples = 0;
  bool staged = false;
};

struct Slot {
  std::vector<float> out, R;
  int64_t rows = 0;
  int32_t pending_rc = DRS_OK;
  std::string pending_msg;
  bool busy = false;
};

}  // namespace

struct drs_engine {
  uint32_t magic = 0x43505544;   // "CPUD"
  int32_t kind = 0, T = 0, D = 0;
  std::vector<int64_t> rows;
  std::vector<std::vector<float>> tables;
  std::vector<bool> table_set;
  Mlp bot, top, fin;
  std::vector<Mlp> tasks;        // MT-WnD heads
  std::vector<Mlp> att;          // DIN attention units
  std::vector<Mlp> rnn;          // DIEN: the two BasicRNN layers, each {i2h, gates_t}
  int32_t interaction_op = 0, itself = 0, sigmoid_top = -1;
  int32_t max_batch = 0, max_lookups = 0, n_batches = 0, n_slots = 1;
  int32_t m_den = 0, w0 = 0, num_int = 0, n_out = 0;
  int64_t cap = 0;
  std::vector<Batch> batches;
  std::vector<Slot> slots;
  std::string err;
};

namespace {

int32_t fail(drs_engine* e, int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

int32_t check_handle(drs_engine* e) {
  if (!e || e->magic != 0x43505544) return DRS_ERR_BAD_ARG;
  return DRS_OK;
}

int32_t mlp_ready(drs_engine* e, const Mlp& m, const char* name) {
  for (size_t i = 0; i < m.set.size(); ++i)
    if (!m.set[i]) return fail(e, DRS_ERR_STATE, "%s layer %zu has no weights", name, i);
  return DRS_OK;
}

// the Caffe2 ENFORCEs (same order and codes as engine.hip convert_inputs)
int32_t validate(drs_engine* e, int32_t n, const int64_t* const* h_idx, const int64_t* n_idx,
                 const int32_t* const* h_len) {
  for (int t = 0; t < e->T; ++t) {
    if (!h_idx[t] && n_idx[t] > 0) return fail(e, DRS_ERR_BAD_ARG, "h_idx[%d] is NULL", t);
    if (!h_len[t]) return fail(e, DRS_ERR_BAD_ARG, "h_len[%d] is NULL", t);
    if (n_idx[t] < 0 || n_idx[t] > e->cap)
      return fail(e, DRS_ERR_BAD_ARG, "table %d: %lld indices exceed staging capacity %lld", t,
                  (long long)n_idx[t], (long long)e->cap);
    int64_t total = 0;
    for (int b = 0; b < n; ++b) {
      if (h_len[t][b] < 0) return fail(e, DRS_ERR_LENGTHS_SUM, "table %d bag %d: negative length", t, b);
      total += h_len[t][b];
      if (total > n_idx[t]) break;
    }
    if (total != n_idx[t])
      return fail(e, DRS_ERR_LENGTHS_SUM, "table %d: sum(lengths)=%lld != len(indices)=%lld", t,
                  (long long)total, (long long)n_idx[t]);
    for (int64_t j = 0; j < n_idx[t]; ++j)
      if (h_idx[t][j] < 0 || h_idx[t][j] >= e->rows[t])
        return fail(e, DRS_ERR_INDEX_RANGE, "table %d: index %lld at position %lld outside [0, %lld)",
                    t, (long long)h_idx[t][j], (long long)j, (long long)e->rows[t]);
  }
  return DRS_OK;
}

int32_t store_batch(drs_engine* e, Batch& b, int32_t n, const float* h_dense,
                    const int64_t* const* h_idx, const int64_t* n_idx, const int32_t* const* h_len) {
  if (n < 0 || n > e->max_batch) return fail(e, DRS_ERR_BAD_ARG, "n_samples=%d exceeds max_batch=%d", n, e->max_batch);
  if (!h_idx || !n_idx || !h_len) return fail(e, DRS_ERR_BAD_ARG, "null index/length arrays");
  if (e->m_den > 0 && !h_dense && n > 0) return fail(e, DRS_ERR_BAD_ARG, "null dense input");
  int32_t rc = validate(e, n, h_idx, n_idx, h_len);
  if (rc) return rc;
  b.idx.assign(e->T, {});
  b.len.assign(e->T, {});
  for (int t = 0; t < e->T; ++t) {
    b.idx[t].assign(h_idx[t], h_idx[t] + n_idx[t]);
    b.len[t].assign(h_len[t], h_len[t] + n);
  }
  if (e->m_den > 0 && n > 0) b.dense.assign(h_dense, h_dense + (size_t)n * e->m_den);
  b.n_samples = n;
  b.staged = true;
  return DRS_OK;
}

// queries = prefixes of staged batches (inferenceEngine.py:200-206); results back to back
int32_t run(drs_engine* e, Slot& s, int n, const Batch* const* bts, const int32_t* bss) {
  for (int t = 0; t < e->T; ++t)
    if (!e->table_set[t]) return fail(e, DRS_ERR_STATE, "table %d has no data", t);
  int32_t rc;
  if ((rc = mlp_ready(e, e->bot, "bottom")) || (rc = mlp_ready(e, e->top, "top")) ||
      (rc = mlp_ready(e, e->fin, "final")) || [&] { for (auto& tk : e->tasks) if ((rc = mlp_ready(e, tk, "task"))) return true; return false; }() ||
      [&] { for (auto& au : e->att) if ((rc = mlp_ready(e, au, "attention"))) return true; return false; }() ||
      [&] { for (auto& rn : e->rnn) if ((rc = mlp_ready(e, rn, "rnn"))) return true; return false; }())
    return rc;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (bss[i] < 0 || bss[i] > bts[i]->n_samples)
      return fail(e, DRS_ERR_BAD_ARG, "bs=%d outside [0, %d]", bss[i], bts[i]->n_samples);
    total += bss[i];
  }
  s.out.assign((size_t)total * e->n_out, 0.f);
  s.R.assign((size_t)(bss[n - 1] > 0 ? bss[n - 1] : 1) * e->num_int, 0.f);
  s.rows = total;
  s.busy = true;
  s.pending_rc = DRS_OK;

  std::vector<const float*> tabs(e->T), bw, bb, tw, tb;
  for (int t = 0; t < e->T; ++t) tabs[t] = e->tables[t].data();
  for (size_t l = 0; l < e->bot.W.size(); ++l) { bw.push_back(e->bot.W[l].data()); bb.push_back(e->bot.b[l].data()); }
  for (size_t l = 0; l < e->top.W.size(); ++l) { tw.push_back(e->top.W[l].data()); tb.push_back(e->top.b[l].data()); }
  orc_model m;
  memset(&m, 0, sizeof m);
  m.model_kind = e->kind; m.T = e->T; m.D = e->D; m.rows = e->rows.data(); m.tables = tabs.data();
  m.n_bot = (int32_t)e->bot.ln.size(); m.ln_bot = e->bot.ln.data(); m.bot_W = bw.data(); m.bot_b = bb.data();
  m.n_top = (int32_t)e->top.ln.size(); m.ln_top = e->top.ln.data(); m.top_W = tw.data(); m.top_b = tb.data();
  if (e->kind == DRS_MODEL_NCF) { m.final_W = e->fin.W[0].data(); m.final_b = e->fin.b[0].data(); m.final_m = e->n_out; }
  std::vector<const float*> kw, kb;
  if (e->kind == DRS_MODEL_MTWND) {
    for (auto& tk : e->tasks)
      for (size_t l = 0; l < tk.W.size(); ++l) { kw.push_back(tk.W[l].data()); kb.push_back(tk.b[l].data()); }
    m.n_task = (int32_t)e->tasks[0].ln.size(); m.ln_task = e->tasks[0].ln.data();
    m.num_tasks = (int32_t)e->tasks.size(); m.task_sigmoid = e->sigmoid_top;
    m.task_W = kw.data(); m.task_b = kb.data();
  }
  std::vector<const float*> aw, ab;
  if (e->kind == DRS_MODEL_DIN) {
    for (auto& au : e->att)
      for (size_t l = 0; l < au.W.size(); ++l) { aw.push_back(au.W[l].data()); ab.push_back(au.b[l].data()); }
    m.n_att = (int32_t)e->att[0].ln.size(); m.ln_att = e->att[0].ln.data();
    m.att_W = aw.data(); m.att_b = ab.data();
    m.n_bot = 0; m.ln_bot = nullptr; m.bot_W = nullptr; m.bot_b = nullptr;
  }
  if (e->kind == DRS_MODEL_DIEN) {
    m.rnn_hidden = e->rnn[0].ln[1];
    for (int l = 0; l < 2; ++l) {
      m.rnn_w[4 * l + 0] = e->rnn[l].W[0].data(); m.rnn_w[4 * l + 1] = e->rnn[l].b[0].data();
      m.rnn_w[4 * l + 2] = e->rnn[l].W[1].data(); m.rnn_w[4 * l + 3] = e->rnn[l].b[1].data();
    }
    m.n_bot = 0; m.ln_bot = nullptr; m.bot_W = nullptr; m.bot_b = nullptr;
  }
  m.interaction_op = e->interaction_op; m.itself = e->itself; m.sigmoid_top = e->sigmoid_top;

  int64_t o = 0;
  for (int i = 0; i < n; ++i) {
    const Batch& b = *bts[i];
    const int32_t bs = bss[i];
    if (bs == 0) continue;
    std::vector<const int64_t*> ip(e->T);
    std::vector<const int32_t*> lp(e->T);
    std::vector<int64_t> ni(e->T);
    for (int t = 0; t < e->T; ++t) {
      int64_t c = 0;
      for (int k = 0; k < bs; ++k) c += b.len[t][k];
      ip[t] = b.idx[t].data(); lp[t] = b.len[t].data(); ni[t] = c;
    }
    rc = orc_forward(&m, bs, e->m_den > 0 ? b.dense.data() : nullptr, ip.data(), ni.data(), lp.data(),
                     s.out.data() + (size_t)o * e->n_out, i == n - 1 ? s.R.data() : nullptr, 0);
    if (rc) { s.busy = false; return fail(e, rc, "oracle forward failed with %d", rc); }
    o += bs;
  }
  return DRS_OK;
}

int32_t finish(drs_engine* e, Slot& s, float* h_out, int64_t h_cap = -1) {
  if (!s.busy) return DRS_OK;
  if (h_out && h_cap >= 0 && h_cap < s.rows * e->n_out)
    return fail(e, DRS_ERR_BAD_ARG, "output buffer holds %lld floats, the slot produces %lld",
                (long long)h_cap, (long long)(s.rows * e->n_out));
  s.busy = false;
  if (h_out && s.rows > 0) memcpy(h_out, s.out.data(), sizeof(float) * (size_t)s.rows * e->n_out);
  return DRS_OK;
}

}  // namespace

extern "C" {

int32_t drs_abi_version(void) { return DRS_ABI_VERSION; }
const char* drs_backend(void) { return "cpu:oracle"; }

int32_t drs_device_count(int32_t* out_count) {
  if (!out_count) return DRS_ERR_BAD_ARG;
  *out_count = 1;     // "the CPU"
  return DRS_OK;
}

const char* drs_last_error(drs_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int32_t drs_create(const drs_model_cfg* cfg, int32_t device_id, drs_handle* out) {
  if (!cfg || !out) return fail(nullptr, DRS_ERR_BAD_ARG, "null cfg/out");
  *out = nullptr;
  if (cfg->num_tables <= 0 || !cfg->table_rows || cfg->n_bot < 1 || !cfg->ln_bot || cfg->n_top < 2 ||
      !cfg->ln_top || cfg->max_batch <= 0 || cfg->max_lookups <= 0 || cfg->num_staged_batches < 0)
    return fail(nullptr, DRS_ERR_BAD_ARG, "bad model config");
  const int D = cfg->sparse_dim;
  if (D <= 0 || D > 4096) return fail(nullptr, DRS_ERR_UNSUPPORTED, "sparse_dim=%d must be in [1, 4096]", D);
  if (device_id != 0) return fail(nullptr, DRS_ERR_BAD_ARG, "device %d of 1", device_id);
  drs_engine* e = new drs_engine();
  e->kind = cfg->model_kind; e->T = cfg->num_tables; e->D = D;
  e->rows.assign(cfg->table_rows, cfg->table_rows + e->T);
  e->interaction_op = cfg->interaction_op; e->itself = cfg->interaction_itself ? 1 : 0;
  e->max_batch = cfg->max_batch; e->max_lookups = cfg->max_lookups;
  e->n_batches = cfg->num_staged_batches; e->n_slots = cfg->num_slots > 0 ? cfg->num_slots : 1;
  e->bot.ln.assign(cfg->ln_bot, cfg->ln_bot + cfg->n_bot);
  e->top.ln.assign(cfg->ln_top, cfg->ln_top + cfg->n_top);
  e->sigmoid_top = cfg->sigmoid_top;
  const int T = e->T, F = T + 1;
  auto bail = [&](int32_t code, const char* msg) { g_create_error = msg; delete e; return code; };
  switch (e->kind) {
    case DRS_MODEL_DLRM:
      e->m_den = e->bot.ln.front(); e->w0 = e->bot.ln.back();
      if (e->w0 != D) return bail(DRS_ERR_BAD_ARG, "arch_sparse_feature_size does not match last dim of bottom mlp");
      if (e->interaction_op == DRS_INTERACT_DOT) e->num_int = (e->itself ? F * (F + 1) / 2 : F * (F - 1) / 2) + D;
      else if (e->interaction_op == DRS_INTERACT_CAT) e->num_int = F * D;
      else return bail(DRS_ERR_BAD_ARG, "unknown interaction op");
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->n_out = e->top.ln.back();
      break;
    case DRS_MODEL_WND:
      if (cfg->n_bot != 1) return bail(DRS_ERR_BAD_ARG, "W&D has no bottom MLP layers");
      e->m_den = e->w0 = e->bot.ln.front();
      e->num_int = T * D + e->w0;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "num_int does not match first dim of top mlp");
      e->n_out = e->top.ln.back();
      break;
    case DRS_MODEL_MTWND:
      if (cfg->n_bot != 1) return bail(DRS_ERR_BAD_ARG, "MT-W&D has no bottom MLP layers");
      e->m_den = e->w0 = e->bot.ln.front();
      e->num_int = T * D + e->w0;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "num_int does not match first dim of top mlp");
      if (cfg->n_task < 2 || !cfg->ln_task || cfg->num_tasks < 1 || cfg->num_tasks > 64)
        return bail(DRS_ERR_BAD_ARG, "MT-W&D needs arch_mlp_tasks and 1..64 task heads");
      if (cfg->ln_task[0] != e->top.ln.back())
        return bail(DRS_ERR_BAD_ARG, "Shared top layer and task MLP layers must have same input/output dimension");
      e->tasks.resize(cfg->num_tasks);
      for (auto& tk : e->tasks) tk.ln.assign(cfg->ln_task, cfg->ln_task + cfg->n_task);
      e->n_out = cfg->num_tasks * cfg->ln_task[cfg->n_task - 1];
      break;
    case DRS_MODEL_DIN:
      if (T < 4) return bail(DRS_ERR_BAD_ARG, "DIN needs at least 4 embedding tables");
      if (cfg->n_bot < 2 || e->bot.ln.front() != 3 * D || e->bot.ln.back() != D)
        return bail(DRS_ERR_BAD_ARG, "DIN attention unit must be 3*D -> ... -> D");
      for (int w : e->bot.ln) if (w < 1) return bail(DRS_ERR_BAD_ARG, "DIN attention unit with an empty layer");
      {   // same limit as the HIP engine (din_any.hip: a sample's activations in 160 KB of LDS)
        int maxw = 0;
        for (int l = 1; l + 1 < cfg->n_bot; ++l) maxw = e->bot.ln[l] > maxw ? e->bot.ln[l] : maxw;
        const bool any = cfg->n_bot != 3 || e->bot.ln[1] > 64 || (int64_t)(T - 3) * e->bot.ln[1] > 4096 || (D & 3) || D > 256;
        if (any && sizeof(float) * (4 * (size_t)D + 2 * (size_t)maxw) > 160 * 1024)
          return bail(DRS_ERR_UNSUPPORTED, "DIN: 4*D + 2*(widest hidden layer of a unit) floats must fit 160 KB of LDS");
      }
      e->m_den = 0; e->w0 = 0;
      e->num_int = 4 * D;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->att.resize(T - 3);
      for (auto& au : e->att) au.ln = e->bot.ln;
      e->bot.ln = {0};
      e->sigmoid_top = -1;
      e->n_out = e->top.ln.back();
      break;
    case DRS_MODEL_DIEN: {
      if (T < 4) return bail(DRS_ERR_BAD_ARG, "DIEN needs at least 4 embedding tables");
      if (cfg->n_bot != 2 || e->bot.ln[0] != D || e->bot.ln[1] < 1) return bail(DRS_ERR_BAD_ARG, "DIEN: ln_bot must be [D, hidden_size]");
      if (sizeof(float) * ((size_t)D + 4 * (size_t)e->bot.ln[1]) > 160 * 1024)   // (the HIP engine's any-shape recurrence)
        return bail(DRS_ERR_UNSUPPORTED, "DIEN: D + 4*hidden_size floats must fit 160 KB of LDS");
      const int H = e->bot.ln[1];
      e->m_den = 0; e->w0 = 0;
      e->num_int = H + 3 * D;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->rnn.resize(2);
      e->rnn[0].ln = {D, H, H};
      e->rnn[1].ln = {H, H, H};
      e->bot.ln = {0};
      e->sigmoid_top = -1;
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_NCF:
      if (T != 4) return bail(DRS_ERR_BAD_ARG, "NCF has 4 embedding tables");
      if (e->top.ln.front() != 2 * D) return bail(DRS_ERR_BAD_ARG, "NCF MLP branch input must be 2*D");
      e->m_den = 0; e->w0 = 0;
      e->num_int = D + e->top.ln.back();
      e->sigmoid_top = -1;
      e->fin.ln = {e->num_int, 0};
      e->n_out = 0;
      break;
    default:
      return bail(DRS_ERR_BAD_ARG, "unknown model kind");
  }
  for (int t = 0; t < T; ++t) {
    if (e->rows[t] <= 0) return bail(DRS_ERR_BAD_ARG, "table with no rows");
    if (e->rows[t] * (int64_t)D >= (1ll << 33)) return bail(DRS_ERR_UNSUPPORTED, "rows*D must be < 2^33 per table");   // (the HIP engine's limit: engine.hip)
  }
  auto init_mlp = [](Mlp& m) {
    const size_t n = m.ln.size() > 0 ? m.ln.size() - 1 : 0;
    m.W.assign(n, {}); m.b.assign(n, {}); m.set.assign(n, false);
  };
  init_mlp(e->bot); init_mlp(e->top); init_mlp(e->fin);
  for (auto& tk : e->tasks) init_mlp(tk);
  for (auto& au : e->att) init_mlp(au);
  for (auto& rn : e->rnn) init_mlp(rn);
  e->tables.assign(T, {});
  e->table_set.assign(T, false);
  e->cap = (int64_t)e->max_batch * e->max_lookups;
  e->batches.resize(e->n_batches);
  e->slots.resize(e->n_slots);
  *out = e;
  return DRS_OK;
}

int32_t drs_destroy(drs_handle h) {
  if (h) { h->magic = 0; delete h; }
  return DRS_OK;
}

int32_t drs_set_table(drs_handle e, int32_t t, const float* h_W, int64_t rows) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (t < 0 || t >= e->T || !h_W) return fail(e, DRS_ERR_BAD_ARG, "bad table id / null data");
  if (rows != e->rows[t]) return fail(e, DRS_ERR_BAD_ARG, "table %d has %lld rows, got %lld", t, (long long)e->rows[t], (long long)rows);
  e->tables[t].assign(h_W, h_W + (size_t)rows * e->D);
  e->table_set[t] = true;
  return DRS_OK;
}

int32_t drs_fill_table_uniform(drs_handle e, int32_t t, float lo, float hi, uint64_t seed) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (t < 0 || t >= e->T) return fail(e, DRS_ERR_BAD_ARG, "bad table id");
  e->tables[t].resize((size_t)e->rows[t] * e->D);
  rc = orc_fill_table_uniform(e->tables[t].data(), e->rows[t], e->D, t, lo, hi, seed, 0);
  if (rc) return fail(e, rc, "fill failed");
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
  if (mlp >= DRS_MLP_ATT0 && mlp - DRS_MLP_ATT0 < (int)e->att.size()) M = &e->att[mlp - DRS_MLP_ATT0];
  if ((mlp == DRS_MLP_RNN0 || mlp == DRS_MLP_RNN1) && e->rnn.size() == 2) M = &e->rnn[mlp - DRS_MLP_RNN0];
  if (!M || layer < 0 || layer >= (int)M->set.size()) return fail(e, DRS_ERR_BAD_ARG, "no such layer");
  if (mlp == DRS_MLP_FINAL && M->ln[1] == 0) {
    if (m <= 0 || m > 1024) return fail(e, DRS_ERR_BAD_ARG, "bad predictor width");
    M->ln[1] = m;
    e->n_out = m;
  }
  if (n != M->ln[layer] || m != M->ln[layer + 1])
    return fail(e, DRS_ERR_BAD_ARG, "layer %d expects W[%d,%d], got [%d,%d]", layer, M->ln[layer + 1], M->ln[layer], m, n);
  M->W[layer].assign(h_W, h_W + (size_t)m * n);
  M->b[layer].assign(h_b, h_b + m);
  M->set[layer] = true;
  return DRS_OK;
}

int32_t drs_stage_batch(drs_handle e, int32_t batch_id, int32_t n_samples, const float* h_dense,
                        const int64_t* const* h_idx, const int64_t* n_idx, const int32_t* const* h_len) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (batch_id < 0 || batch_id >= e->n_batches) return fail(e, DRS_ERR_BAD_ARG, "batch_id %d of %d", batch_id, e->n_batches);
  return store_batch(e, e->batches[batch_id], n_samples, h_dense, h_idx, n_idx, h_len);
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
  return run(e, e->slots[slot], n, bts, bs);
}

int32_t drs_forward_async(drs_handle e, int32_t slot, int32_t batch_id, int32_t bs) {
  return drs_forward_multi_async(e, slot, 1, &batch_id, &bs);
}

int32_t drs_wait(drs_handle e, int32_t slot, float* h_out, int64_t h_out_floats) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (h_out && h_out_floats < 0) return fail(e, DRS_ERR_BAD_ARG, "negative output capacity");
  return finish(e, e->slots[slot], h_out, h_out_floats);
}

int32_t drs_sync(drs_handle e) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  for (auto& s : e->slots) s.busy = false;
  return DRS_OK;
}

int32_t drs_forward(drs_handle e, int32_t batch_id, int32_t bs, float* h_out) {
  int32_t rc = drs_forward_async(e, 0, batch_id, bs);
  if (rc) return rc;
  return finish(e, e->slots[0], h_out);
}

int32_t drs_forward_inputs_async(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                                 const int64_t* const* h_idx, const int64_t* n_idx,
                                 const int32_t* const* h_len) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  Batch b;
  if ((rc = store_batch(e, b, bs, h_dense, h_idx, n_idx, h_len))) return rc;
  const Batch* bt = &b;
  return run(e, e->slots[slot], 1, &bt, &bs);
}

int32_t drs_run_queues_async(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                             const int64_t* h_ids, int64_t ids_row_stride, int64_t n_idx_per_table,
                             const int32_t* h_lengths, int64_t len_row_stride) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!h_ids || !h_lengths || n_idx_per_table < 0) return fail(e, DRS_ERR_BAD_ARG, "bad 2-D input arrays");
  std::vector<const int64_t*> ip(e->T);
  std::vector<const int32_t*> lp(e->T);
  std::vector<int64_t> ni(e->T, n_idx_per_table);
  for (int t = 0; t < e->T; ++t) { ip[t] = h_ids + (int64_t)t * ids_row_stride; lp[t] = h_lengths + (int64_t)t * len_row_stride; }
  return drs_forward_inputs_async(e, slot, bs, h_dense, ip.data(), ni.data(), lp.data());
}

int32_t drs_run_queues_multi_async(drs_handle e, int32_t slot, int32_t n, const int32_t* bs,
                                   const float* const* h_dense, const int64_t* const* h_ids,
                                   const int64_t* ids_row_stride, const int64_t* n_idx_per_table,
                                   const int32_t* const* h_lengths, const int64_t* len_row_stride) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots) return fail(e, DRS_ERR_BAD_ARG, "slot %d of %d", slot, e->n_slots);
  if (n < 1 || n > DRS_MAX_COALESCE) return fail(e, DRS_ERR_BAD_ARG, "1..%d queries per launch", DRS_MAX_COALESCE);
  if (!bs || !h_dense || !h_ids || !ids_row_stride || !n_idx_per_table || !h_lengths || !len_row_stride)
    return fail(e, DRS_ERR_BAD_ARG, "bad per-query array tables");
  std::vector<Batch> b(n);
  const Batch* bts[DRS_MAX_COALESCE];
  for (int i = 0; i < n; ++i) {
    if (!h_ids[i] || !h_lengths[i] || n_idx_per_table[i] < 0) return fail(e, DRS_ERR_BAD_ARG, "bad 2-D input arrays of query %d", i);
    std::vector<const int64_t*> ip(e->T);
    std::vector<const int32_t*> lp(e->T);
    std::vector<int64_t> ni(e->T, n_idx_per_table[i]);
    for (int t = 0; t < e->T; ++t) { ip[t] = h_ids[i] + (int64_t)t * ids_row_stride[i]; lp[t] = h_lengths[i] + (int64_t)t * len_row_stride[i]; }
    if ((rc = store_batch(e, b[i], bs[i], h_dense[i], ip.data(), ni.data(), lp.data()))) return rc;
    bts[i] = &b[i];
  }
  return run(e, e->slots[slot], n, bts, bs);
}

int32_t drs_forward_inputs(drs_handle e, int32_t slot, int32_t bs, const float* h_dense,
                           const int64_t* const* h_idx, const int64_t* n_idx,
                           const int32_t* const* h_len, float* h_out) {
  int32_t rc = drs_forward_inputs_async(e, slot, bs, h_dense, h_idx, n_idx, h_len);
  if (rc) return rc;
  return finish(e, e->slots[slot], h_out);
}

int32_t drs_fetch_interaction(drs_handle e, int32_t slot, int32_t bs, float* h_R) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (slot < 0 || slot >= e->n_slots || !h_R || bs < 0 || bs > e->max_batch) return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  const Slot& s = e->slots[slot];
  if ((size_t)bs * e->num_int > s.R.size()) return fail(e, DRS_ERR_STATE, "no forward of that size on this slot");
  memcpy(h_R, s.R.data(), sizeof(float) * (size_t)bs * e->num_int);
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

// operator level: "device" pointers are host pointers here
int32_t drs_sls(drs_handle e, const float* d_W, int64_t rows, int32_t D, const int32_t* d_idx,
                const int32_t* d_len, int64_t n_bags, int64_t n_idx, float* d_out, int32_t) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  rc = orc_sls_i32(d_W, rows, D, d_idx, d_len, n_bags, n_idx, d_out, 0);
  return rc ? fail(e, rc, "sls failed") : DRS_OK;
}

int32_t drs_fc(drs_handle e, const float* d_x, int64_t M, int32_t K, const float* d_W, const float* d_b,
               int32_t N, int32_t act, float* d_y) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  rc = orc_fc(d_x, M, K, d_W, d_b, N, act, d_y, 0);
  return rc ? fail(e, rc, "fc failed") : DRS_OK;
}

int32_t drs_interact_dot(drs_handle e, const float* d_T, int64_t B, int32_t F, int32_t D, int32_t itself,
                         float* d_R) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  rc = orc_interact_dot(d_T, B, F, D, itself, d_R, 0);
  return rc ? fail(e, rc, "interact failed") : DRS_OK;
}

// tuning / measurement knobs have no meaning on the CPU: accepted and ignored
int32_t drs_set_option(drs_handle e, const char* key, int64_t) {   // (options tune the HIP kernels: accepted, ignored)
  if (!e || !key) return DRS_ERR_BAD_ARG;
  return DRS_OK;
}
int32_t drs_set_profiling(drs_handle e, int32_t) { return e ? DRS_OK : DRS_ERR_BAD_ARG; }
int32_t drs_kernel_time(drs_handle e, int32_t, double* sum_ms, int64_t* launches) {
  if (!e || !sum_ms || !launches) return DRS_ERR_BAD_ARG;
  *sum_ms = 0.0; *launches = 0;
  return DRS_OK;
}
int32_t drs_reset_kernel_time(drs_handle e) { return e ? DRS_OK : DRS_ERR_BAD_ARG; }
int32_t drs_kernel_bytes(drs_handle e, int32_t, int64_t* bytes) {
  if (!e || !bytes) return DRS_ERR_BAD_ARG;
  *bytes = 0;
  return DRS_OK;
}
int32_t drs_get_option(drs_handle e, const char* key, int64_t* value) {
  if (!e || !key || !value) return DRS_ERR_BAD_ARG;
  *value = 0;
  if (!strcmp(key, "preferred_coalesce") || !strcmp(key, "preferred_slots")) {
    // the engine's own rule (csrc/engine_create.hip choose_launch_forms, engine_options.hip), so the host code above the ABI
    // runs with the launch-set sizes the product uses: 16 when the model's MLP launches overlap each
    // other (MLP FLOP per gathered byte > 20, or a DLRM whose MLP launch outlasts its gather), 12 for
    // gather-bound DLRM, 8 otherwise
    double flop = 0;
    for (const Mlp* mm : {&e->bot, &e->top, &e->fin})
      for (size_t i = 0; i + 1 < mm->ln.size(); ++i) flop += 2.0 * mm->ln[i] * (mm->ln[i + 1] > 0 ? mm->ln[i + 1] : 64);
    for (const Mlp& rn : e->rnn) flop += 2.0 * (e->T - 3) * ((double)rn.ln[0] * rn.ln[1] + (double)rn.ln[1] * rn.ln[2]);
    const double bytes = (double)e->T * e->max_lookups * e->D * 4.0;
    // launch sets in flight: 6 for the MLP-bound class (their MFMA-bound launches overlap each other), 3 otherwise
    if (!strcmp(key, "preferred_slots")) {
      *value = !(flop / bytes > 20.0) ? 3 : (e->kind == DRS_MODEL_DIEN || e->kind == DRS_MODEL_MTWND || e->kind == DRS_MODEL_WND) ? 4 : 6;
      return DRS_OK;
    }
    const int want = (e->kind == DRS_MODEL_DIEN || e->kind == DRS_MODEL_MTWND || e->kind == DRS_MODEL_WND) ? 2 : e->kind == DRS_MODEL_NCF ? 3 : 4;
    int streams = flop / bytes > 20.0 ? (e->n_slots < want ? e->n_slots : want) : 1;
    if (streams == 1 && e->n_slots >= 2 && e->kind == DRS_MODEL_DLRM) {
      double weights = 0;
      for (const Mlp* mm : {&e->bot, &e->top})
        for (size_t i = 0; i + 1 < mm->ln.size(); ++i) weights += (double)mm->ln[i] * mm->ln[i + 1];
      if (12.0 + weights / 4500.0 > 2048.0 * bytes / 5.5e6) streams = 2;
      // ... or whose MLP side has a stand-alone wide layer between two chains (RM2's 2112 x 128: K N >= 256 K weights)
      for (const Mlp* mm : {&e->bot, &e->top})
        for (size_t i = 0; i + 1 < mm->ln.size(); ++i)
          if ((int64_t)mm->ln[i] * mm->ln[i + 1] >= 262144) streams = 2;
    }
    *value = streams > 1 ? DRS_MAX_COALESCE : (e->kind == DRS_MODEL_DLRM ? 12 : 8);
  }
  return DRS_OK;
}
// the collective is RCCL over xGMI: no CPU restatement (the CPU suite combines ranks over gloo)
int32_t drs_comm_unique_id(uint8_t*) { return DRS_ERR_UNSUPPORTED; }
int32_t drs_comm_create(const uint8_t*, int32_t, int32_t, int32_t, drs_comm* out) {
  if (out) *out = nullptr;
  return DRS_ERR_UNSUPPORTED;
}
int32_t drs_comm_destroy(drs_comm) { return DRS_OK; }
int32_t drs_comm_barrier(drs_comm) { return DRS_ERR_UNSUPPORTED; }
int32_t drs_stats_allreduce(drs_comm, int64_t*, int32_t, double*) { return DRS_ERR_UNSUPPORTED; }
const char* drs_comm_last_error(void) { return "no RCCL behind the CPU restatement of the ABI"; }
int32_t drs_debug_gather_stamps(drs_handle e, int32_t, uint64_t*, int64_t, int64_t* n_blocks) {
  if (!e || !n_blocks) return DRS_ERR_BAD_ARG;
  *n_blocks = 0;
  return DRS_OK;
}
int32_t drs_last_dispatch(drs_handle e, int32_t, char* buf, int64_t cap) {   // (no launches on the CPU: an empty list)
  if (!e || !buf || cap < 1) return DRS_ERR_BAD_ARG;
  buf[0] = 0;
  return DRS_OK;
}
int32_t drs_gather_bytes(drs_handle e, int32_t batch_id, int32_t bs, int64_t* bytes) {
  int32_t rc = check_handle(e);
  if (rc) return rc;
  if (!bytes || batch_id < 0 || batch_id >= e->n_batches || !e->batches[batch_id].staged) return fail(e, DRS_ERR_BAD_ARG, "bad arguments");
  const Batch& b = e->batches[batch_id];
  if (bs < 0 || bs > b.n_samples) return fail(e, DRS_ERR_BAD_ARG, "bs out of range");
  int64_t total = 0;
  for (int t = 0; t < e->T; ++t)
    for (int k = 0; k < bs; ++k) total += (int64_t)b.len[t][k] * e->D * 4 + (int64_t)b.len[t][k] * 4 + 4 + e->D * 4;
  *bytes = total;
  return DRS_OK;
}

}  // extern "C"

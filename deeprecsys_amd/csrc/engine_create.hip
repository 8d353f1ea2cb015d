// libdrs_hip.so, host side: per-device set-up, which launch forms a model takes, drs_create / drs_destroy, tables and FC weights.
#include "engine.h"


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

namespace drs {
namespace eng {

thread_local std::string g_create_error;

int32_t fail(drs_engine* e, int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) { std::lock_guard<std::mutex> l(e->err_mu); e->err = buf; } else g_create_error = buf;
  return code;
}

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

}  // namespace eng
}  // namespace drs

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
  // How many streams, and how many sets in flight the engine asks its feeder for ("preferred_slots"), round 6, one box,
  // (sets in flight, streams) -> k queries/s:  RM3 config 3 (3,3) 34.1 (6,4) 35.0 (4,2) 31.7 | RM3 JSON 67.8 / 70.8 / 70.9 |
  // W&D 98.1 / 104.9 / 100.0 | NCF 297 / 389-417 / 346 | MT-WnD 68.5 / 70.8 / 73.0-74.8 | DIEN 158 / 185 / 214: with three
  // sets one of four streams idles; DIEN's and MT-WnD's launches (two workgroups per CU each) thrash when four of them
  // run at once and do best two at a time with a second set queued behind each (profiles/r06_slots.md).
  // (W&D joined them when its gather became sls_one_kernel: (4, 2) 110.8 k at p99 0.59 ms, (6, 4) 110.4 k at 1.01 ms.)
  const bool two_at_a_time = e->kind == DRS_MODEL_DIEN || e->kind == DRS_MODEL_MTWND || e->kind == DRS_MODEL_WND;
  // (NCF: six sets on three streams 530-552 k against 519-524 k on four, alternating three times on one box.)
  const int want_streams = two_at_a_time ? 2 : e->kind == DRS_MODEL_NCF ? 3 : 4;
  e->mlp_streams = mlp_bound ? (e->n_slots < want_streams ? e->n_slots : want_streams) : 1;
  e->mlp_bound = mlp_bound ? 1 : 0;
  e->pref_slots = !mlp_bound ? 3 : two_at_a_time ? 4 : 6;
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
  // A gather-bound DLRM whose MLP side is THREE launches -- chain, a stand-alone wide layer, chain (RM2's 2112 x 128 first top
  // layer) -- keeps two MLP streams: beside the next set's gather the GEMM launch does not get its 256-thread workgroups
  // onto CUs the gather's one-wave workgroups keep backfilling (kernel trace, profiles/r06_rm2_timeline.txt: 513-533 us for
  // 1.7 GFLOP, ending 28 us after the gather it ran beside), and on ONE stream the next set's bottom chain queues behind it:
  // 58 us between two gather launches.  Two streams let that chain run meanwhile: RM2 21.6 k -> 22.7 k queries/s.
  if (gather_bound_dlrm && e->mlp_split && e->n_slots >= 2) {
    bool wide = false;
    for (const Mlp* mm : {&e->bot, &e->top})
      for (size_t i = 0; i + 1 < mm->ln.size(); ++i) wide = wide || (int64_t)mm->ln[i] * mm->ln[i + 1] >= e->mlp_wide_kn;
    if (wide) e->mlp_streams = 2;
  }
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
  // W&D: the first layer from 3 072 rows on (384 workgroups; the second layer's 256 at full sets stay with gemm_kernel, the
  // 94.7 above).  Until round 6 the bar was 512 = full sets of 256-sample queries only: 15 queries per set 99.9 k -> 106.0 k,
  // 8 / 12 / 16 equal; at 2 560-2 640 rows (10 x 256, or 16 queries of the run scripts' ~165 samples) the two kernels are
  // within 1 % of each other either way (profiles/r06_wnd_set_sizes.txt)
  if (e->kind == DRS_MODEL_WND) { e->tune.gemm32_small = 12; e->tune.gemm32_small_blocks = 384; }
}

int32_t drs_create(const drs_model_cfg* cfg, int32_t device_id, drs_handle* out) {
  if (!cfg || !out) return fail(nullptr, DRS_ERR_BAD_ARG, "null cfg/out");
  *out = nullptr;
  if (cfg->num_tables <= 0 || !cfg->table_rows || cfg->n_bot < 1 || !cfg->ln_bot || cfg->n_top < 2 ||
      !cfg->ln_top || cfg->max_batch <= 0 || cfg->max_lookups <= 0 || cfg->num_staged_batches < 0)
    return fail(nullptr, DRS_ERR_BAD_ARG, "bad model config");
  const int D = cfg->sparse_dim;
  // Every shipped config has D % 4 == 0 and D <= 256 (rows read as 16-byte pieces); any other width goes through the
  // generic forms (sls_any_kernel, chain_kernel / fc_kernel's scalar paths, din_any.hip): the reference only asks
  // m_spa == ln_bot[-1] (models/dlrm_s_caffe2.py:435-437).
  if (D <= 0 || D > 4096) return fail(nullptr, DRS_ERR_UNSUPPORTED, "sparse_dim=%d must be in [1, 4096]", D);
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
      e->num_int = T * D + e->w0;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "num_int does not match first dim of top mlp");
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_MTWND: {
      if (cfg->n_bot != 1) return bail(DRS_ERR_BAD_ARG, "MT-W&D has no bottom MLP layers");
      e->m_den = e->w0 = e->bot.ln.front();
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
      // an attention unit is create_mlp over 3*D - <arch_mlp_bot> - D (models/din.py:255-277): any depth, any widths
      if (cfg->n_bot < 2 || e->bot.ln.front() != 3 * D || e->bot.ln.back() != D)
        return bail(DRS_ERR_BAD_ARG, "DIN attention unit must be 3*D -> ... -> D");
      for (int w : e->bot.ln) if (w < 1) return bail(DRS_ERR_BAD_ARG, "DIN attention unit with an empty layer");
      // din.hip's forms: one hidden layer of <= 64 units, rows in 16-byte pieces; the two-launch attention kernel (the only
      // path for sls_exact = 1 and for shapes the fused launch is not instantiated for) keeps 4 samples x (T - 3) units x
      // h hidden values in 64 KB of LDS.  Everything else: din_any.hip.
      e->din_any = cfg->n_bot != 3 || e->bot.ln[1] > 64 || (int64_t)(T - 3) * e->bot.ln[1] > 4096 || (D & 3) || D > 256;
      for (int l = 1; l + 1 < cfg->n_bot; ++l) e->din_maxw = std::max(e->din_maxw, e->bot.ln[l]);
      if (e->din_any && !din_any_fits(D, e->din_maxw))
        return bail(DRS_ERR_UNSUPPORTED, "DIN: 4*D + 2*(widest hidden layer of a unit) floats must fit 160 KB of LDS");
      e->m_den = 0; e->w0 = 0;
      e->num_int = 4 * D;
      if (e->num_int != e->top.ln.front()) return bail(DRS_ERR_BAD_ARG, "# of feature interactions does not match first dim of top mlp");
      e->att.resize(T - 3);
      for (auto& au : e->att) { au.ln = e->bot.ln; au.layers.resize(cfg->n_bot - 1); au.sigmoid_layer = -1; }
      e->bot.ln = {0}; e->bot.layers.clear();      // no bottom MLP of its own
      e->top.sigmoid_layer = -1;
      e->n_out = e->top.ln.back();
      break;
    }
    case DRS_MODEL_DIEN: {
      if (T < 4) return bail(DRS_ERR_BAD_ARG, "DIEN needs at least 4 embedding tables");
      if (cfg->n_bot != 2 || e->bot.ln[0] != D || e->bot.ln[1] < 1) return bail(DRS_ERR_BAD_ARG, "DIEN: ln_bot must be [D, hidden_size]");
      // D in {16, 32, 64} with hidden_size in {8, 16, 32, 64}: din.hip's forms; any other pair: din_any.hip
      if (!dien_applicable(D, e->bot.ln[1]) && !dien_any_fits(D, e->bot.ln[1]))
        return bail(DRS_ERR_UNSUPPORTED, "DIEN: D + 4*hidden_size floats must fit 160 KB of LDS");
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
#ifdef DRS_LAB
    // "mlp_early" (lab only): its stream, and the word a RUNNING kernel polls while another stream's write lands in it --
    // fine-grained memory, so that the write is visible across the XCDs' L2s
    CREATE_TRY(hipStreamCreateWithFlags(&s.early_stream, hipStreamNonBlocking));
    CREATE_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&s.d_gflag), sizeof(uint32_t), hipDeviceMallocFinegrained));
    CREATE_TRY(hipMemset(s.d_gflag, 0, sizeof(uint32_t)));
#endif
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
    CREATE_TRY(hipEventCreateWithFlags(&s.ev_dma, hipEventDisableTiming | hipEventDisableSystemFence));
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
  CREATE_TRY(hipStreamCreateWithFlags(&e->stream_dma, hipStreamNonBlocking));
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
  if (e->stream_dma) { (void)hipStreamSynchronize(e->stream_dma); (void)hipStreamDestroy(e->stream_dma); }
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
    if (s.ev_dma) (void)hipEventDestroy(s.ev_dma);
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
  if (e->d_att_ln) (void)hipFree(e->d_att_ln);
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

}  // extern "C"

// libdrs_hip.so, host side: ONE launch set -- which kernels, on which streams, in which order (enqueue_forward), its completion
// (wait_slot), and the entry points on top: drs_forward*, drs_wait, drs_sync, drs_fetch_interaction, the operator-level calls.
#include "engine.h"

namespace drs {
namespace eng {

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
  if (!e->att.empty() && e->att_dirty && e->din_any) {
    // any-shape units (din_any.hip): the kernel reads the layers where drs_set_fc put them
    if (!e->d_att) {
      std::vector<const float*> hp;
      for (auto& au : e->att)
        for (auto& L : au.layers) { hp.push_back(L.W); hp.push_back(L.b); }
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att), sizeof(float*) * hp.size()));
      HIP_TRY(e, hipMemcpy(e->d_att, hp.data(), sizeof(float*) * hp.size(), hipMemcpyHostToDevice));
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_att_ln), sizeof(int32_t) * e->att[0].ln.size()));
      HIP_TRY(e, hipMemcpy(e->d_att_ln, e->att[0].ln.data(), sizeof(int32_t) * e->att[0].ln.size(), hipMemcpyHostToDevice));
    }
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
  const bool din_fused = e->kind == DRS_MODEL_DIN && !e->sls_exact && e->din_fused && !e->din_any &&
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
    // 3: the any-shape form (din_any.hip) -- every shape without an instance in din.hip, or on request
    const int form = dien_applicable(e->D, Hh) ? e->dien_mfma : 3;
    const bool mfma_form = form != 3 && form && Hh % 16 == 0;
    if (e->dien_fuse_top && mfma_form && nt >= 1 && nt <= 4 && e->top.layers[0].packed &&
        dien_top_fusable(nt, e->top.ln.data(), Hh) && e->top.ln[0] == Hh + 3 * e->D) {
      tp.n = nt; tp.sc1 = dp ? 1 : 0; tp.out = out; tp.ldo = e->n_out;
      for (int l = 0; l < nt; ++l) {
        const Layer& L = e->top.layers[l];
        tp.Wp[l] = L.W + ((size_t)L.m * L.n + 63) / 64 * 64; tp.b[l] = L.b;
        tp.K[l] = e->top.ln[l]; tp.N[l] = e->top.ln[l + 1]; tp.act[l] = act_of(e->top, l);
      }
      tp.kmax = dien_top_kmax(nt, e->top.ln.data());
    }
    log_launch(e->tune.log, "%s<%d,%d%s>[%d wg]", form == 3 ? "dien_rnn_any_kernel" : mfma_form ? "dien_rnn_mfma_kernel" : "dien_rnn_kernel",
               e->D, Hh, tp.n ? ",top" : "", form == 3 ? c : mfma_form ? (c + 15) / 16 : (c + 3) / 4);
    HIP_TRY(e, launch_dien_rnn(s.T, e->ldT, q, e->T, e->D, Hh, e->d_att_packed, rw, form, s.R,
                               e->ldR, s.stream, tp.n ? &tp : nullptr, dp));
    if (!tp.n && (rc = run_mlp(e, s, e->top, s.R, e->ldR, Mv, out, e->n_out, dp))) return rc;
  } else if (e->kind == DRS_MODEL_DIN) {
    // attention units over the pooled rows -> top MLP input R [rows, 4D] -> top MLP (all ReLU)
    HIP_TRY(e, join());
    if (e->din_any) {
      log_launch(e->tune.log, "din_attention_any_kernel[%lld wg]", (long long)Mv);
      HIP_TRY(e, launch_din_attention_any(s.T, e->ldT, Mv, e->T, e->D, (int)e->att[0].ln.size(), e->d_att_ln, e->d_att, e->din_maxw,
                                          s.R, e->ldR, s.stream));
    }
    if (!din_fused && !e->din_any) log_launch(e->tune.log, "din_attention_kernel[%lld wg]", (long long)((Mv + 3) / 4));
    if (!din_fused && !e->din_any)
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
#ifdef DRS_LAB
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
#endif
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
    // ... on the engine's copy stream, behind an event: queued on s.stream itself, the transfer (2 MB = ~40 us of PCIe
    // per MT-WnD set) held up the NEXT set's launches on that MLP stream -- the chip idled 13.5 % of the time
    // (profiles/r06_mtwnd_kernel_overlap_before.txt)
    HIP_TRY(e, hipEventRecord(s.ev_dma, s.stream));
    HIP_TRY(e, hipStreamWaitEvent(e->stream_dma, s.ev_dma, 0));
    HIP_TRY(e, hipMemcpyAsync(s.h_out + kOutOffset, s.d_out, sizeof(float) * (size_t)Mv * e->n_out,
                              hipMemcpyDeviceToHost, e->stream_dma));
    HIP_TRY(e, hipStreamWriteValue32(e->stream_dma, s.dm_out, s.seq, 0));
  }
  s.on_dma = out_dma;
  s.polled = e->zero_copy != 0;
  return DRS_OK;
}


int32_t wait_slot(drs_engine* e, Slot& s, float* h_out, int64_t h_cap) {
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
        if (s.on_dma) HIP_TRY(e, hipStreamSynchronize(e->stream_dma));
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
int32_t check_handle(drs_engine* e, bool hot) {
  if (!e) return fail(nullptr, DRS_ERR_BAD_ARG, "null handle");
  if (!hot && e->launcher) e->launcher->drain();
  return set_device(e);
}


}  // namespace eng
}  // namespace drs

extern "C" {

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
  if (D <= 0 || D > 4096) return fail(e, DRS_ERR_UNSUPPORTED, "D=%d must be in [1, 4096]", D);
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

}  // extern "C"

// libdrs_hip.so, host side: staged input sets and the per-call input path (int64 -> int32 + Caffe2's ENFORCEs on the host,
// one packed pinned block per query or launch set, one DMA copy) -- drs_stage_batch, drs_forward_inputs*, drs_run_queues*.
#include "engine.h"

namespace drs {
namespace eng {

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


}  // namespace eng
}  // namespace drs

extern "C" {

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

}  // extern "C"

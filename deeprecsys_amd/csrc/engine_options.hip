// libdrs_hip.so, host side: the option table (docs/OPTIONS.md), profiling read-outs, the dispatch record.
#include "engine.h"

// ---- options ------------------------------------------------------------------------------------
// ONE table: key, accepted values, what setting it entails, where it lives.  docs/OPTIONS.md explains every key; the
// product library accepts the keys below and nothing else.  The lab's instruments and every option whose default lost its
// measurement (rounds 1-5: docs/DESIGN_rounds_1-4.md) exist in the lab build only (-DDRS_LAB,
// `make lab-lib`: lab_set_option / lab_get_option).
namespace {

enum : uint32_t {
  O_BOOL = 1,      // any value, stored as 0 | 1
  O_SYNC = 2,      // nothing of the engine may be in flight while it changes: drs_sync first
  O_STREAMS = 4,   // the slots' streams are dealt again afterwards (apply_stream_mode)
  O_POOL = 8,      // the per-call input workers are recreated on next use
  O_RO = 16,       // read only
};
struct OptDesc {
  const char* key;
  int64_t lo, hi;
  bool (*valid)(int64_t);                     // further restriction inside [lo, hi], or nullptr
  uint32_t flags;
  int64_t (*get)(drs_engine*);
  void (*set)(drs_engine*, int64_t);          // plain store, or nullptr with `custom`
  int32_t (*custom)(drs_engine*, int64_t);    // options that are actions (table_placement, table_spacer)
};
#define OPT(KEY, LO, HI, VALID, FLAGS, EXPR)                                                       \
  {KEY, LO, HI, VALID, FLAGS, [](drs_engine* e) -> int64_t { return (int64_t)(e->EXPR); },         \
   [](drs_engine* e, int64_t v) { e->EXPR = static_cast<std::remove_reference_t<decltype(e->EXPR)>>(v); }, nullptr}
#define OPT_RO(KEY, ...) {KEY, 0, 0, nullptr, O_RO, [](drs_engine* e) -> int64_t { __VA_ARGS__ }, nullptr, nullptr}
constexpr int64_t kBig = (int64_t)1 << 62;

int32_t set_table_placement(drs_engine* e, int64_t value) {

    // Where a multi-gigabyte allocation lands in HBM moves the gather by up to 6 % and stays for the allocation's
    // lifetime: the feeder may try a few places with the model's own launch sets and keep the best.
    //   -1: copy the tables into one more allocation and use that one (the earlier ones stay allocated, or the allocator
    //       hands the same pages out again) | k >= 0: use candidate k | -2: free every candidate but the one in use.
    // Refused (DRS_ERR_OOM, nothing changes) when one more copy would not leave 3/4 of the device's memory free.
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
    return DRS_OK;
}

int32_t set_table_spacer(drs_engine* e, int64_t value) {
  if (value < 0) return fail(e, DRS_ERR_BAD_ARG, "table_spacer %lld", (long long)value);

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
    return DRS_OK;
}

int64_t arena_in_use(drs_engine* e) {
  return (int64_t)(std::find_if(e->arenas.begin(), e->arenas.end(), [&](const Arena& a) { return a.p == e->tables; }) - e->arenas.begin());
}

const OptDesc kOptions[] = {
    // gather
    OPT("sls_exact", 0, 1, nullptr, O_BOOL, sls_exact),
    OPT("sls_flat", 0, 2, nullptr, 0, tune.sls_flat),
    OPT("sls_bpw", 0, 4, [](int64_t v) { return v != 3; }, 0, tune.sls_bpw),
    OPT("sls_nt", 0, 1, nullptr, O_BOOL, tune.sls_nt),
    OPT("sls_one", 0, 64, [](int64_t v) { return v == 0 || v == 1 || v == 16 || v == 64; }, 0, tune.sls_one),
    OPT("din_fused", 0, 1, nullptr, O_BOOL, din_fused),
    OPT("din_pipe", 0, 1, nullptr, O_BOOL, tune.din_pipe),
    OPT("din_s", 0, 4, [](int64_t v) { return v != 3; }, 0, tune.din_s),
    OPT("din_nt", 0, 1, nullptr, O_BOOL, tune.din_nt),
    OPT("dien_mfma", 0, 3, nullptr, 0, dien_mfma),
    OPT("dien_fuse_top", 0, 1, nullptr, 0, dien_fuse_top),
    // MLP side
    OPT("gemm_split", 0, 1, nullptr, O_BOOL, gemm_split),
    OPT("mlp_split", 0, 1, nullptr, O_BOOL, mlp_split),
    OPT("mlp_wide_kn", 1, kBig, nullptr, 0, mlp_wide_kn),
    OPT("mlp_fuse", 0, 1, nullptr, O_BOOL, mlp_fuse),
#ifdef DRS_LAB
    OPT("mlp_stream", 0, 4, [](int64_t v) { return v != 3; }, 0, tune.mlp_stream),
#else
    OPT("mlp_stream", 2, 4, [](int64_t v) { return v != 3; }, 0, tune.mlp_stream),
#endif
    OPT("mlp_stream_2cu", 0, 1, nullptr, 0, tune.mlp_stream_2cu),
    OPT("mlp_rows32", 0, kBig, nullptr, 0, tune.mlp_rows32),
    OPT("mlp_nsplit", 0, 4, [](int64_t v) { return v == 0 || v == 2 || v == 4; }, 0, tune.mlp_nsplit),
    OPT("mlp_nsplit_rows", 0, kBig, nullptr, 0, tune.mlp_nsplit_rows),
    OPT("mlp_gemm_tile", 0, 322, [](int64_t v) { return v == 0 || v == 22 || v == 12 || v == 21 || v == 11 || v == 214 || v == 322 || v == 321 || v == 312 || v == 311; }, 0, tune.gemm_tile),
    // streams, host side
    OPT("shared_stream", 0, 2, nullptr, O_SYNC | O_STREAMS, shared_stream),
    OPT("mlp_streams", 1, 8, nullptr, O_SYNC | O_STREAMS, mlp_streams),
    OPT("host_threads", -1, 64, nullptr, O_POOL, host_threads),
#ifdef DRS_LAB
    OPT("zero_copy_inputs", 0, 3, nullptr, 0, zero_copy_inputs),
#else
    OPT("zero_copy_inputs", 1, 3, nullptr, 0, zero_copy_inputs),
#endif
    OPT("out_dma", 0, kBig, nullptr, O_SYNC, out_dma),
    OPT("dispatch_log", 0, 1, nullptr, O_BOOL, dispatch_log),
    // where the tables live
    {"table_placement", -kBig, kBig, nullptr, O_SYNC, [](drs_engine* e) -> int64_t { return arena_in_use(e); }, nullptr, set_table_placement},
#ifdef DRS_LAB
    OPT("table_alloc", 0, 3, nullptr, 0, table_alloc),
#else
    OPT("table_alloc", 0, 2, nullptr, 0, table_alloc),
#endif
    {"table_spacer", 0, kBig, nullptr, 0, [](drs_engine* e) -> int64_t { return (int64_t)e->spacers.size() << 30; }, nullptr, set_table_spacer},
    // what the engine tells its feeder (read only)
    OPT_RO("preferred_coalesce", return e->mlp_streams > 1 ? DRS_MAX_COALESCE : (e->kind == DRS_MODEL_DLRM ? 12 : 8);),
    // launch sets the feeder should keep in flight: 3 for the gather-bound models (gather | MLP | enqueue; more only adds
    // latency: dlrm_rm1.json 258 k queries/s at 3 .. 8, DIN 175-176 k); 6 for the MLP-bound ones, whose sets are chains of
    // MFMA-bound launches that overlap each other on up to four MLP streams (round 6, same box, 3 -> 6: DIEN 158 k -> 185 k
    // (198 k at 5), W&D 98.1 k -> 104.7 k, RM3 config 3 34.1 k -> 35.1 k, RM3 JSON 67.8 k -> 70.6 k, MT-WnD 68.5 k -> 70.6 k,
    // NCF 297 k -> 417 k; p99 doubles and stays under 4 ms against the 25 ms SLA)
    // (DIEN and MT-WnD: 4, two at a time on two MLP streams: 214 k against 185 k, 74 k against 71 k)
    OPT_RO("preferred_slots", return e->pref_slots;),
    OPT_RO("gather_bound", return e->gather_bound;),
    OPT_RO("device", return e->device;),
    OPT_RO("table_placements", return (int64_t)e->arenas.size();),
    OPT_RO("table_bytes", return (int64_t)e->tables_bytes;),
    OPT_RO("table_address", return (int64_t)(uintptr_t)e->tables;),
};
#undef OPT
#undef OPT_RO

#ifdef DRS_LAB
// The lab build's further keys: instruments (probes, address moves, CU masks, priorities) and the options of forms that
// lost their A/B -- kept reachable for the record and for the lab's parity runs (tests with DRS_TEST_LAB=1).
// *handled = false: not a lab key either.
int32_t lab_set_option(drs_engine* e, const char* key, int64_t value, bool* handled) {
  *handled = true;
  if (false) {}
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
  else if (!strcmp(key, "mlp_layout") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_layout = (int)value; }
  else if (!strcmp(key, "sls_short_bag") && value >= -1 && value <= 1 << 20) e->sls_short_bag = (int)value;
  else if (!strcmp(key, "sls_uniform")) e->sls_uniform = value ? 1 : 0;
  else if (!strcmp(key, "mlp_fuse_rows") && value >= 0) e->mlp_fuse_rows = value;
  else if (!strcmp(key, "mlp_early") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_early = (int)value; }
  else if (!strcmp(key, "small_piped") && (value == 0 || value == 1)) { int32_t rc = drs_sync(e); if (rc) return rc; e->small_piped = (int)value; }
  else if (!strcmp(key, "mlp_small_rows") && value >= 0) { int32_t rc = drs_sync(e); if (rc) return rc; e->mlp_small_rows = value; }
  else if (!strcmp(key, "mlp_preload")) e->tune.mlp_preload = value ? 1 : 0;
  else if (!strcmp(key, "mlp_gemm_2cu") && (value == 0 || value == 1)) e->tune.gemm_2cu = (int)value;
  else if (!strcmp(key, "launch_thread") && (value == 0 || value == 1)) e->launch_thread = (int)value;
  else if (!strcmp(key, "mlp_gemm")) e->tune.mlp_gemm = value ? 1 : 0;
  else if (!strcmp(key, "mlp_gemm32") && (value == 0 || value == 1)) e->tune.gemm32 = (int)value;
  else if (!strcmp(key, "mlp_gemm32_small") && (value == 0 || value == 22 || value == 21 || value == 12 || value == 11)) e->tune.gemm32_small = (int)value;
  else if (!strcmp(key, "mlp_gemm32_small_blocks") && value >= 0 && value <= 65536) e->tune.gemm32_small_blocks = (int)value;
  else if (!strcmp(key, "mlp_gemm32_blocks") && value >= 1 && value <= 65536) e->tune.gemm32_blocks = (int)value;
  else if (!strcmp(key, "mlp_debug")) e->tune.mlp_debug = (int)value;
  else if (!strcmp(key, "mlp_kc") && (value == 0 || value == 64 || value == 128 || value == 192 || value == 256)) e->tune.mlp_kc = (int)value;
  else if (!strcmp(key, "zero_copy")) { int32_t rc = drs_sync(e); if (rc) return rc; e->zero_copy = value ? 1 : 0; }
  else if (!strcmp(key, "table_vmm_chunk") && value >= -1) e->vmm_chunk = value;
  else if (!strcmp(key, "table_vmm_align") && value >= 0) e->vmm_align = value;
  else if (!strcmp(key, "table_vmm_shuffle") && (value == 0 || value == 1)) e->vmm_shuffle = (int)value;
  else if (!strcmp(key, "table_select_pool") && value >= 0 && value <= 192) e->sel_want_pool = value;
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
  else *handled = false;
  return DRS_OK;
}

bool lab_get_option(drs_engine* e, const char* key, int64_t* value) {
  const Tune& t = e->tune;
  struct { const char* k; int64_t v; } lab[] = {
      {"sls_uniform", e->sls_uniform}, {"sls_short_bag", e->sls_short_bag}, {"mlp_fuse_rows", e->mlp_fuse_rows}, {"mlp_small_rows", e->mlp_small_rows},
      {"small_piped", e->small_piped}, {"mlp_early", e->mlp_early}, {"mlp_gemm", t.mlp_gemm}, {"mlp_gemm_2cu", t.gemm_2cu}, {"mlp_gemm32", t.gemm32},
      {"mlp_gemm32_blocks", t.gemm32_blocks}, {"mlp_gemm32_small", t.gemm32_small}, {"mlp_gemm32_small_blocks", t.gemm32_small_blocks},
      {"mlp_preload", t.mlp_preload}, {"mlp_kc", t.mlp_kc}, {"mlp_debug", t.mlp_debug}, {"mlp_layout", e->mlp_layout}, {"launch_thread", e->launch_thread},
      {"zero_copy", e->zero_copy}, {"table_vmm_chunk", e->vmm_chunk}, {"table_vmm_align", e->vmm_align},
      {"table_select_pool", e->sel_pool}, {"table_select_kept", e->sel_kept},
      {"table_select_best_ns", e->sel_best_ns}, {"table_select_worst_ns", e->sel_worst_ns}, {"table_select_kept_worst_ns", e->sel_kept_worst_ns},
      {"table_select_ms", e->sel_ms}, {"table_probe_mbs", e->probe_mbs}, {"table_probe_gather_ns", e->probe_gather_ns}, {"table_probe_ps", e->probe_ps},
      {"mlp_cu_mask", e->mlp_cu_mask}, {"gather_priority", e->gather_priority}, {"gather_cu_complement", e->gather_cu_complement},
      {"table_vmm_shuffle", e->vmm_shuffle},
      {"table_kind", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return a.kind; return 0; }()},
      {"table_va_candidates", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return (int64_t)a.vas.size(); return 0; }()},
      {"table_va", [&]() -> int64_t { for (const Arena& a : e->arenas) if (a.p == e->tables) return a.va_cur; return 0; }()}};
  for (auto& kv : lab)
    if (!strcmp(key, kv.k)) { *value = kv.v; return true; }
  return false;
}
#endif  // DRS_LAB

}  // namespace

extern "C" {

int32_t drs_set_option(drs_handle e, const char* key, int64_t value) {
  if (!e || !key) return DRS_ERR_BAD_ARG;
  for (const OptDesc& d : kOptions) {
    if (strcmp(key, d.key)) continue;
    if (d.flags & O_RO) return fail(e, DRS_ERR_BAD_ARG, "option %s is read only", key);
    const int64_t v = (d.flags & O_BOOL) ? (value ? 1 : 0) : value;
    if (v < d.lo || v > d.hi || (d.valid && !d.valid(v))) return fail(e, DRS_ERR_BAD_ARG, "unknown option %s=%lld", key, (long long)value);
    if (d.flags & O_SYNC) {
      const int32_t rc = drs_sync(e);
      if (rc) return rc;
    }
    if (d.custom) return d.custom(e, v);
    d.set(e, v);
    if (d.flags & O_STREAMS) apply_stream_mode(e);
    if (d.flags & O_POOL) e->pool.reset();
    return DRS_OK;
  }
#ifdef DRS_LAB
  bool handled = false;
  const int32_t rc = lab_set_option(e, key, value, &handled);
  if (handled) return rc;
#endif
  return fail(e, DRS_ERR_BAD_ARG, "unknown option %s=%lld", key, (long long)value);
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
  for (const OptDesc& d : kOptions)
    if (!strcmp(key, d.key)) { *value = d.get(e); return DRS_OK; }
#ifdef DRS_LAB
  if (lab_get_option(e, key, value)) return DRS_OK;
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

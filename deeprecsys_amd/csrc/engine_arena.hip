// libdrs_hip.so, host side: where the table arena lives -- arenas by hipMalloc / the virtual-memory API / contiguous,
// placement candidates and spacers; lab build: the probes and address moves of tools/placement_lab.py.
#include "engine.h"

namespace drs {
namespace eng {

// ---- table arenas ---------------------------------------------------------------------------------
hipError_t va_reserve(size_t total, size_t align, VaRange* out) {
  // hipMemAddressReserve does not honour its alignment argument beyond the granularity (measured: "1 GiB aligned"
  // came back 2 MiB aligned): reserve `align` bytes more and align inside the range by hand
  void* base = nullptr;
  const size_t reserved = total + align;
  hipError_t r = hipMemAddressReserve(&base, reserved, 0, nullptr, 0);
  if (r != hipSuccess) return r;
  out->base = base;
  out->reserved = reserved;
  out->p = static_cast<float*>(align ? reinterpret_cast<void*>(((uintptr_t)base + align - 1) / align * align) : base);
  return hipSuccess;
}

// the arena's physical handles at address `at` (handle i at place[i]); on failure nothing stays mapped there
hipError_t arena_map(const Arena& a, float* at, int device) {
  const size_t n = a.handles.size(), csz = a.va_bytes / n;
  hipError_t r = hipSuccess;
  std::vector<size_t> done;
  for (size_t i = 0; i < n && r == hipSuccess; ++i) {
    r = hipMemMap(reinterpret_cast<char*>(at) + a.place[i] * csz, csz, 0, a.handles[i], 0);
    if (r == hipSuccess) done.push_back(a.place[i]);
  }
  if (r == hipSuccess) {
    hipMemAccessDesc desc;
    memset(&desc, 0, sizeof desc);
    desc.location.type = hipMemLocationTypeDevice;
    desc.location.id = device;
    desc.flags = hipMemAccessFlagsProtReadWrite;
    r = hipMemSetAccess(at, a.va_bytes, &desc, 1);
  }
  if (r != hipSuccess) {
    for (size_t pl : done) (void)hipMemUnmap(reinterpret_cast<char*>(at) + pl * csz, csz);
    (void)hipGetLastError();
  }
  return r;
}

void arena_free(Arena& a) {
  if (!a.p) { a = Arena(); return; }
  if (a.kind == 0) {
    (void)hipFree(a.p);
  } else {
    (void)hipMemUnmap(a.p, a.va_bytes);
    for (auto& h : a.handles) (void)hipMemRelease(h);
    for (auto& h : a.pads) (void)hipMemRelease(h);
    for (auto& v : a.vas) (void)hipMemAddressFree(v.base, v.reserved);
  }
  a = Arena();
}

#ifdef DRS_LAB
// The model's OWN gather kernel on a one-table problem laid over [base, base + bytes): `bags` bags of L sorted,
// distinct rows each (one row per L-th of the range, what np.unique leaves of a bag's draws,
// data_generator/dlrm_data_caffe2.py:105-110), the launch the engine would make for them.  *us = average
// duration of a launch.  Why the real kernel: synthetic row-read probes that saturate the memory system read every
// gigabyte of HBM equally fast; the gather kernels, with a handful of loads in flight per lane, do not
// (profiles/r05_placement/).
hipError_t probe_gather(drs_engine* e, const float* base, size_t bytes, double* us) {
  *us = 0;
  const int D = e->D;
  const int64_t rows = (int64_t)(bytes / ((size_t)D * 4));
  const int L = e->max_lookups > 256 ? 256 : e->max_lookups;
  const int bags = 16384;
  if (rows < L || rows >= (1ll << 31)) return hipErrorInvalidValue;
  hipError_t r = hipSuccess;
  if (!e->probe_idx || e->probe_rows != rows || e->probe_L != L) {
    if (e->probe_idx) { (void)hipFree(e->probe_idx); e->probe_idx = nullptr; }
    std::vector<int32_t> idx((size_t)bags * L);
    uint32_t z = 0x2545F491u;
    for (int b = 0; b < bags; ++b)
      for (int j = 0; j < L; ++j) {
        z ^= z << 13; z ^= z >> 17; z ^= z << 5;
        const int64_t lo = rows * j / L, hi = rows * (j + 1) / L;
        idx[(size_t)b * L + j] = (int32_t)(lo + (int64_t)(z % (uint32_t)(hi > lo ? hi - lo : 1)));
      }
    if ((r = hipMalloc(&e->probe_idx, sizeof(int32_t) * idx.size())) != hipSuccess) return r;
    if ((r = hipMemcpy(e->probe_idx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice)) != hipSuccess) return r;
    if (!e->probe_out && (r = hipMalloc(&e->probe_out, sizeof(float) * (size_t)bags * D)) != hipSuccess) return r;
    if (!e->probe_tab && (r = hipMalloc(&e->probe_tab, sizeof(int64_t) * 2)) != hipSuccess) return r;
    if (!e->probe_err && (r = hipMalloc(&e->probe_err, sizeof(int32_t))) != hipSuccess) return r;
    const int64_t tab[2] = {0, rows};
    if ((r = hipMemcpy(e->probe_tab, tab, sizeof tab, hipMemcpyHostToDevice)) != hipSuccess) return r;
    if ((r = hipMemset(e->probe_err, 0, sizeof(int32_t))) != hipSuccess) return r;
    e->probe_rows = rows; e->probe_L = L; e->probe_bags = bags;
  }
  SlsArgs a;
  memset(&a, 0, sizeof a);
  a.tables = base; a.tab_off = e->probe_tab; a.tab_rows = e->probe_tab + 1;
  a.q.n_q = 1; a.q.vstart[1] = bags; a.q.cum[1] = bags; a.q.bs[0] = bags;
  a.idx[0] = e->probe_idx; a.off[0] = nullptr; a.uniform_len[0] = L;
  a.idx_stride = (int64_t)bags * L; a.off_stride = bags + 1;
  a.out = e->probe_out; a.ld_out = D; a.col0 = 0; a.T = 1; a.D = D; a.err = e->probe_err; a.ts = nullptr;
  Tune t = e->tune;
  t.log = nullptr;
  const int exact = L <= e->sls_short_bag && !sls_flat_applicable(a, t);
  hipEvent_t e0, e1;
  if ((r = hipEventCreate(&e0)) != hipSuccess) return r;
  if ((r = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return r; }
  const int warm = 2, reps = 6;
  for (int i = 0; i < warm + reps && r == hipSuccess; ++i) {
    if (i == warm) r = hipEventRecord(e0, nullptr);
    if (r == hipSuccess) r = launch_sls(a, exact, t, nullptr);
  }
  if (r == hipSuccess) r = hipEventRecord(e1, nullptr);
  if (r == hipSuccess) r = hipEventSynchronize(e1);
  float ms = 0.f;
  if (r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
  if (r == hipSuccess) *us = (double)ms * 1e3 / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return r;
}

// "table_alloc" 3: the arena out of the fastest gigabytes of a pool.  ceil(bytes / 1 GiB) = n chunks are needed;
// up to 2 n + 8 one-GiB handles are created (never more than half of the free memory), each is timed with
// probe_gather, the n fastest are mapped back to back as the arena and the rest is released at once: one copy of
// the tables, nothing held.  Costs ~1 ms per pool chunk at engine start.
hipError_t arena_alloc_selected(drs_engine* e, size_t bytes, Arena* out) {
  *out = Arena();
  const size_t chunk = (size_t)1 << 30;
  const size_t n = (bytes + chunk - 1) / chunk;
  const auto t_begin = std::chrono::steady_clock::now();
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  size_t free_b = 0, total_b = 0;
  hipError_t r = hipMemGetInfo(&free_b, &total_b);
  if (r != hipSuccess) return r;
  size_t m = 2 * n + 8;
  if (e->sel_want_pool > 0) m = (size_t)e->sel_want_pool;
  if (m > 192) m = 192;
  while (m > n && m * chunk > free_b / 2) --m;
  if (m < n) m = n;
  std::vector<hipMemGenericAllocationHandle_t> pool;
  for (size_t i = 0; i < m; ++i) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    pool.push_back(h);
  }
  auto release_all = [&]() { for (auto& h : pool) (void)hipMemRelease(h); };
  if (pool.size() < n) { release_all(); return hipErrorOutOfMemory; }
  m = pool.size();
  std::vector<double> us(m, 0.0);
  if (m > n) {
    Arena all;
    all.kind = 1; all.va_bytes = m * chunk; all.handles = pool;
    for (size_t i = 0; i < m; ++i) all.place.push_back(i);
    VaRange v;
    r = va_reserve(all.va_bytes, 0, &v);
    if (r == hipSuccess) {
      r = arena_map(all, v.p, e->device);
      if (r == hipSuccess) {
        // two rounds, the faster reading of each chunk counts (a single reading can catch a clock ramp)
        for (int round = 0; round < 2 && r == hipSuccess; ++round)
          for (size_t i = 0; i < m && r == hipSuccess; ++i) {
            double t = 0;
            r = probe_gather(e, reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.p) + i * chunk), chunk, &t);
            if (r == hipSuccess && (round == 0 || t < us[i])) us[i] = t;
          }
        (void)hipMemUnmap(v.p, all.va_bytes);
      }
      (void)hipMemAddressFree(v.base, v.reserved);
    }
    if (r != hipSuccess) { release_all(); (void)hipGetLastError(); return r; }
  }
  std::vector<size_t> order(m);
  for (size_t i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return us[x] < us[y]; });
  std::vector<size_t> keep(order.begin(), order.begin() + (long)n);
  std::sort(keep.begin(), keep.end());
  Arena a;
  a.kind = 1;
  a.va_bytes = n * chunk;
  a.align = e->vmm_align > 0 ? (size_t)e->vmm_align : 0;
  std::vector<bool> kept(m, false);
  for (size_t k : keep) { a.handles.push_back(pool[k]); a.place.push_back(a.place.size()); kept[k] = true; }
  for (size_t i = 0; i < m; ++i) if (!kept[i]) (void)hipMemRelease(pool[i]);
  VaRange v;
  r = va_reserve(a.va_bytes, a.align, &v);
  if (r == hipSuccess) {
    r = arena_map(a, v.p, e->device);
    if (r != hipSuccess) (void)hipMemAddressFree(v.base, v.reserved);
  }
  if (r != hipSuccess) { for (auto& h : a.handles) (void)hipMemRelease(h); (void)hipGetLastError(); return r; }
  a.vas.push_back(v);
  a.p = v.p;
  e->sel_pool = (int64_t)m; e->sel_kept = (int64_t)n;
  e->sel_best_ns = (int64_t)(us[order[0]] * 1e3); e->sel_worst_ns = (int64_t)(us[order[m - 1]] * 1e3);
  e->sel_kept_worst_ns = (int64_t)(us[order[n - 1]] * 1e3);
  e->sel_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_begin).count();
  *out = a;
  return hipSuccess;
}

#endif  // DRS_LAB
// `bytes` of device memory for the tables, built as e->table_alloc / vmm_* say.  On failure nothing stays
// allocated and *out is empty.
hipError_t arena_alloc(drs_engine* e, size_t bytes, Arena* out) {
  *out = Arena();
#ifdef DRS_LAB
  if (e->table_alloc == 3) {
    if (arena_alloc_selected(e, bytes, out) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();           // (no room for a pool, no virtual-memory API ...: a plain allocation)
  }
#endif  // DRS_LAB
  if (e->table_alloc == 0 || e->table_alloc == 2 || e->table_alloc == 3) {
    // 2: physically contiguous device memory, best effort (hipDeviceMallocContiguous: the driver assembles the
    // allocation from neighbouring free blocks instead of taking whatever blocks head its free lists -- DESIGN.md 5)
    void* p = nullptr;
    hipError_t r = hipErrorOutOfMemory;
    if (e->table_alloc == 2) {
      r = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
      if (r != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    if (r != hipSuccess) r = hipMalloc(&p, bytes);
    if (r != hipSuccess) return r;
    out->p = static_cast<float*>(p);
    out->va_bytes = bytes;
    return hipSuccess;
  }
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = e->device;
  size_t gran = 0;
  hipError_t r = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (r != hipSuccess) return r;
  if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;      // (the runtime reports 4 KiB; whole 2 MiB pages keep every mapping a huge page)
  // "table_vmm_chunk" -1 (the default): 1 GiB handles for arenas of at least 1 GiB, one handle for smaller ones
  const int64_t want = e->vmm_chunk < 0 ? (bytes >= ((size_t)1 << 30) ? (int64_t)1 << 30 : 0) : e->vmm_chunk;
  const size_t chunk = want > 0 ? (size_t)round_up(want, (int64_t)gran) : 0;
  const size_t total = (size_t)round_up((int64_t)bytes, (int64_t)(chunk ? chunk : gran));
  Arena a;
  a.kind = 1;
  a.va_bytes = total;
  a.align = e->vmm_align > 0 ? (size_t)round_up(e->vmm_align, (int64_t)gran) : 0;
  const size_t n = chunk ? total / chunk : 1, csz = chunk ? chunk : total;
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    r = hipMemCreate(&h, csz, &prop, 0);
    if (r != hipSuccess) break;
    a.handles.push_back(h);
    // chunk i of physical memory goes to place perm(i) of the range ("table_vmm_shuffle", a lab option: a fixed odd-multiplier walk)
    size_t pl = i;
    if (e->vmm_shuffle && n > 2) pl = (n & (n - 1)) == 0 ? (i * ((n / 2) | 1) + n / 3) % n : n - 1 - i;
    a.place.push_back(pl);
  }
  VaRange v;
  if (r == hipSuccess) r = va_reserve(total, a.align, &v);
  if (r == hipSuccess) {
    r = arena_map(a, v.p, e->device);
    if (r != hipSuccess) (void)hipMemAddressFree(v.base, v.reserved);
  }
  if (r != hipSuccess) {
    for (auto& h : a.handles) (void)hipMemRelease(h);
    (void)hipGetLastError();
    return r;
  }
  a.vas.push_back(v);
  a.va_cur = 0;
  a.p = v.p;
  *out = a;
  return hipSuccess;
}

#ifdef DRS_LAB
// The arena in use moves to another address range: "table_va_next" reserves one more range and maps the arena's
// memory there (the ranges tried so far stay reserved -- address space only, no memory), "table_va_select" k goes
// back to candidate k and gives the other ranges up.  A lab instrument (tools/placement_lab.py): it showed that the
// gather's speed on an arena does NOT depend on the address -- one arena read the same at 24-60 address ranges, and
// fast / slow memory stayed fast / slow wherever it was mapped (profiles/r05_placement/README.md).
int32_t arena_move(drs_engine* e, Arena& a, int64_t to /* -1: a fresh range */) {
  if (a.kind != 1) return fail(e, DRS_ERR_STATE, "the table arena was not built with the virtual-memory API (\"table_alloc\" 1)");
  if (to >= (int64_t)a.vas.size()) return fail(e, DRS_ERR_BAD_ARG, "address candidate %lld of %zu", (long long)to, a.vas.size());
  if (to < 0) {
    if (a.vas.size() >= 64) return fail(e, DRS_ERR_BAD_ARG, "64 address candidates are the limit");
    VaRange v;
    hipError_t r = va_reserve(a.va_bytes, a.align, &v);
    if (r != hipSuccess) { (void)hipGetLastError(); return fail(e, DRS_ERR_OOM, "hipMemAddressReserve: %s", hipGetErrorString(r)); }
    a.vas.push_back(v);
    to = (int64_t)a.vas.size() - 1;
  }
  if (to == a.va_cur) return DRS_OK;
  HIP_TRY(e, hipMemUnmap(a.p, a.va_bytes));
  hipError_t r = arena_map(a, a.vas[(size_t)to].p, e->device);
  if (r != hipSuccess) {
    // back to where it was: the engine must not be left without its tables
    hipError_t r2 = arena_map(a, a.p, e->device);
    return fail(e, DRS_ERR_HIP, "mapping the tables at another address: %s%s", hipGetErrorString(r), r2 == hipSuccess ? "" : " (and the old mapping could not be restored)");
  }
  const bool in_use = e->tables == a.p;
  a.va_cur = (int)to;
  a.p = a.vas[(size_t)to].p;
  if (in_use) e->tables = a.p;
  return DRS_OK;
}

#endif  // DRS_LAB

// new table contents make the other placement candidates stale: only the arena in use survives

void drop_other_placements(drs_engine* e) {
  drop_spacers(e);
  if (e->arenas.size() <= 1) return;
  std::vector<Arena> keep;
  for (Arena& a : e->arenas) { if (a.p == e->tables) keep.push_back(a); else arena_free(a); }
  e->arenas = keep;
}
void drop_spacers(drs_engine* e) {
  for (auto& h : e->spacers) (void)hipMemRelease(h);
  e->spacers.clear();
}

}  // namespace eng
}  // namespace drs

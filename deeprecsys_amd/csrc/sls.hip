// Multi-table SparseLengthsSum gather-reduce for gfx950 (MI355X).
//
// Replaces the T x SparseLengthsSum([tbl, idx, len]) operators emitted by
// create_emb (reference models/dlrm_s_caffe2.py:281-329, op at :317-325) with ONE
// launch over all B*T bags of a query.  HBM-bound: every pooled row is a random
// D*4-byte read (128 B at D=32, 256 B at D=64) out of multi-GB tables.
//
// Mapping to CDNA4
//   - a row is read by G = D/V adjacent lanes, V floats (16 B for V=4) per lane:
//     one global_load_dwordx4 per lane, 64/G whole rows per wave-instruction,
//     each row a single contiguous, aligned segment (tables are 256-B aligned and
//     D*4 is a multiple of 16).
//   - EXACT variant: each G-lane group owns one bag and walks its rows in index
//     order, so every output column is the sequential fp32 sum
//     ((r0 + r1) + r2) + ... -- bit-identical to the Caffe2 CPU perfkernel the
//     reference runs.  64/G bags progress concurrently in a wave; U independent
//     row loads are kept in flight per lane to cover the ~0.5-1 us HBM latency.
//   - SPLIT variant: the wave owns one bag, lane group g takes rows g, g+64/G, ...
//     and the partial sums are combined with a wave-wide xor butterfly
//     (different fp32 summation order -> tolerance compare, not bitwise).
//   - FLAT variant (fixed-length bags -- every shipped reference config generates
//     num_indices_per_lookup_fixed inputs): a wave owns BPW consecutive bags of one sample
//     (R = BPW*L flattened rows), lane group g takes rows g, g+64/G, ... and ALL of a
//     lane's NL = ceil(R / (64/G)) row loads are issued before the first one is consumed;
//     the indices come from ONE coalesced read (lane i owns row i) and reach the loading
//     lanes through the cross-lane network (ds_bpermute), not LDS.  Two dependent HBM round
//     trips per wave (indices, rows) instead of the five or more of a ring walk: short bags
//     (RM3: 20 x 128 B) and single-query launches spend their time streaming, not waiting.
//   - the bag's offsets come from the staged prefix-sum vector; its indices are
//     staged in LDS by one coalesced read per wave (CH at a time) and then
//     broadcast-read by the lanes of the group; no __syncthreads: a wave only
//     reads what it wrote itself.
//   - 64-thread workgroups: B*T bags of a single query are few (2048 at RMC1
//     b=256), so the launch is cut into as many workgroups as possible to cover
//     all 256 CUs; workgroups never cooperate.
//   - index range is checked per row (Caffe2 ENFORCEs it): an out-of-range index
//     raises bit 0 of *err and contributes zero instead of faulting.
#include <hip/hip_ext.h>

#include "drs_internal.h"

namespace drs {
namespace {

template <int V>
struct Vec;
template <>
struct Vec<4> {
  using type = float4;
};
template <>
struct Vec<2> {
  using type = float2;
};

__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float2 vzero2() { return make_float2(0.f, 0.f); }
template <int V>
__device__ __forceinline__ typename Vec<V>::type vzero();
template <>
__device__ __forceinline__ float4 vzero<4>() { return vzero4(); }
template <>
__device__ __forceinline__ float2 vzero<2>() { return vzero2(); }

__device__ __forceinline__ void vadd(float4& a, const float4& b) {
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}
__device__ __forceinline__ void vadd(float2& a, const float2& b) {
  a.x += b.x; a.y += b.y;
}
__device__ __forceinline__ float4 vshfl_xor(const float4& a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m),
                     __shfl_xor(a.w, m));
}
__device__ __forceinline__ float2 vshfl_xor(const float2& a, int m) {
  return make_float2(__shfl_xor(a.x, m), __shfl_xor(a.y, m));
}

constexpr int kChunk = 128;  // indices staged in LDS per bag per round

template <int V>
__device__ __forceinline__ typename Vec<V>::type vsel(bool keep, const typename Vec<V>::type& v);
template <>
__device__ __forceinline__ float4 vsel<4>(bool keep, const float4& v) {
  return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}
template <>
__device__ __forceinline__ float2 vsel<2>(bool keep, const float2& v) {
  return make_float2(keep ? v.x : 0.f, keep ? v.y : 0.f);
}

// G lanes per row, V floats per lane, U row loads in flight per lane.
// table rows are read once per launch (~1 % reuse inside a batch): "sls_nt" reads them with the
// non-temporal hint (same-session A/B on RMC1, two boxes: +1.5 % queries/s)
typedef float f4v_nt __attribute__((ext_vector_type(4)));
typedef float f2v_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_nt(const float4* p) {
  const f4v_nt t = __builtin_nontemporal_load(reinterpret_cast<const f4v_nt*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float2 ld_nt(const float2* p) {
  const f2v_nt t = __builtin_nontemporal_load(reinterpret_cast<const f2v_nt*>(p));
  return make_float2(t.x, t.y);
}

// NT: the hint must be a COMPILE-TIME property of the load: a run-time `nt ? ld_nt(p) : *p` is if-converted
// into one plain load (the hint is metadata the merge drops): measured in the ISA, 0 of 5 / 2 of 14 loads kept it.
template <int G, int V, int U, bool EXACT, bool NT = false>
__global__ __launch_bounds__(64) void sls_kernel(SlsArgs a) {
  using vec = typename Vec<V>::type;
  constexpr int NG = 64 / G;                  // lane groups per wave
  constexpr int BAGS = EXACT ? NG : 1;        // bags per wave
  constexpr int STEP = EXACT ? 1 : NG;        // row stride between a lane's loads
  constexpr int OWNERS = EXACT ? G : 64;      // lanes that stage one bag's indices
  __shared__ __attribute__((aligned(16))) int32_t s_idx[BAGS][kChunk];

  // live timing (bench.py roofline leg): first/last constant-rate clock tick of every
  // workgroup; the host takes max(end) - min(start) as the launch duration
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();

  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const int col = min(gl * V, a.D - V);       // clamp idle lanes onto valid columns
  const bool col_ok = gl * V < a.D;

  const int64_t n_bags = (int64_t)a.q.cum[a.q.n_q] * a.T;
  const int64_t bag = (int64_t)blockIdx.x * BAGS + (EXACT ? g : 0);
  const bool bag_ok = bag < n_bags;
  const int smp = bag_ok ? (int)(bag / a.T) : 0;            // valid-sample number over all queries
  const int t = bag_ok ? (int)(bag - (int64_t)smp * a.T) : 0;
  // which coalesced query owns this sample: select chain over <= DRS_MAX_COALESCE entries (no dynamic
  // indexing of the kernel-argument arrays)
  int b = smp, vrow = a.q.vstart[0] + smp, ulen = a.uniform_len[0];
  const int32_t* qidx = a.idx[0];
  const int32_t* qoff = a.off[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    b = in ? smp - a.q.cum[i] : b;
    vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
    ulen = in ? a.uniform_len[i] : ulen;
    qidx = in ? a.idx[i] : qidx;
    qoff = in ? a.off[i] : qoff;
  }
  if (a.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < a.q.n_q && smp >= a.q.cum[i];
      b = in ? smp - a.q.cum[i] : b;
      vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
      ulen = in ? a.uniform_len[i] : ulen;
      qidx = in ? a.idx[i] : qidx;
      qoff = in ? a.off[i] : qoff;
    }
  }

  // fixed-length bags (every shipped reference config: num_indices_per_lookup_fixed) need
  // no offsets: one dependent HBM round trip less before the first row load can issue
  int beg, end;
  if (ulen >= 0) {
    beg = bag_ok ? b * ulen : 0;
    end = bag_ok ? beg + ulen : 0;
  } else {
    const int32_t* __restrict__ offp = qoff + (int64_t)t * a.off_stride;
    beg = bag_ok ? offp[b] : 0;
    end = bag_ok ? offp[b + 1] : 0;
  }
  const int32_t* __restrict__ ip = qidx + (int64_t)t * a.idx_stride;
  const float* __restrict__ W = a.tables + a.tab_off[t] + col;
  const uint32_t rows = (uint32_t)a.tab_rows[t];
  const int64_t D = a.D;
  const uint32_t Dv = (uint32_t)a.D / V;   // row stride in load-width units: rows * D / V < 2^32 (rows * D < 2^33 is enforced at table creation)

  int32_t* my_idx = s_idx[EXACT ? g : 0];
  const int me = EXACT ? gl : lane;           // my slot among the owners
  const int first = EXACT ? 0 : g;            // first row (within a chunk) of this lane
  vec acc = vzero<V>();
  bool bad = false;

  // Control flow is kept WAVE-UNIFORM: every lane runs as many rounds as the
  // longest bag in the wave needs (shorter bags re-read their last row, an L1
  // hit, and add +0.0f).  With scalar branches each pipeline arm below is one
  // straight-line block, so the compiler's s_waitcnt vmcnt(N) counts stay exact
  // and U..2U row loads per lane remain outstanding.
  const int len = end - beg;
  int len_max = len;
#pragma unroll
  for (int m = G; m < 64; m <<= 1) len_max = max(len_max, __shfl_xor(len_max, m));
  len_max = __builtin_amdgcn_readfirstlane(len_max);

  for (int c = 0; c < len_max; c += kChunk) {
    const int n = min(kChunk, len - c);            // this lane's rows in the chunk (may be <= 0)
    const int n_u = min(kChunk, len_max - c);      // uniform: rounds the wave runs
    const int last = max(n - 1, 0);
    const int j0 = beg + c;
    // stage the next indices of each bag in LDS: coalesced, clamped (branch-free)
    for (int c0 = 0; c0 < n_u; c0 += 4 * OWNERS) {
      int32_t tmp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) tmp[q] = n > 0 ? ip[j0 + min(c0 + q * OWNERS + me, last)] : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (c0 + q * OWNERS + me < kChunk) my_idx[c0 + q * OWNERS + me] = tmp[q];
    }
    __builtin_amdgcn_wave_barrier();

    // U independent, unconditional row loads
    auto issue = [&](vec (&ring)[U], int pos) {
      uint32_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = (uint32_t)my_idx[min(pos + u * STEP, last)];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bad |= (pos + u * STEP < n) && (r[u] >= rows);
        r[u] = r[u] < rows ? r[u] : 0u;
        const vec* rp_ = reinterpret_cast<const vec*>(W) + (uint64_t)(r[u] * Dv);
        if constexpr (NT) ring[u] = ld_nt(rp_); else ring[u] = *rp_;
      }
    };
    auto consume = [&](const vec (&ring)[U], int pos) {
#pragma unroll
      for (int u = 0; u < U; ++u) vadd(acc, vsel<V>(pos + u * STEP < n, ring[u]));
    };

    // software pipeline over two register rings: while ring A (round k) is summed
    // in index order, ring B (round k+1) is already in flight, and vice versa.
    // Each arm holds its own issue+consume pair; the distinct asm comments keep
    // SimplifyCFG from sinking the common tails into a join.
    constexpr int R = U * STEP;
    vec ringA[U], ringB[U];
    int jj = first;                                 // per-lane row position
    int ju = 0;                                     // uniform round position
    issue(ringA, jj);
    for (;;) {
      if (ju + R < n_u) {
        issue(ringB, jj + R);
        __builtin_amdgcn_sched_barrier(0);   // loads first, then the sums
        consume(ringA, jj);
        asm volatile("; drs sls: A summed, B in flight" ::: "memory");
      } else {
        consume(ringA, jj);
        asm volatile("; drs sls: A summed, tail" ::: "memory");
        break;
      }
      if (ju + 2 * R < n_u) {
        issue(ringA, jj + 2 * R);
        __builtin_amdgcn_sched_barrier(0);
        consume(ringB, jj + R);
        asm volatile("; drs sls: B summed, A in flight" ::: "memory");
      } else {
        consume(ringB, jj + R);
        asm volatile("; drs sls: B summed, tail" ::: "memory");
        break;
      }
      jj += 2 * R;
      ju += 2 * R;
    }
    __builtin_amdgcn_wave_barrier();
  }

  if (!EXACT) {
#pragma unroll
    for (int m = G; m < 64; m <<= 1) vadd(acc, vshfl_xor(acc, m));
  }
  if (bad) atomicOr(a.err, 1);
  if (bag_ok && col_ok && (EXACT || g == 0)) {
    float* o = a.out + (int64_t)vrow * a.ld_out + a.col0 + (int64_t)t * D + col;
    *reinterpret_cast<vec*>(o) = acc;
  }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);   // include the output store in the span
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}


// ---------------------------------------------------------------------------
// ONE lookup per bag (W&D, MT-WnD, NCF, DIEN: num_indices_per_lookup 1, fixed): the pooled "sum" is an indexed row
// copy, and the lane-group-per-bag walk above spends it waiting -- three dependent round trips (index, row, store)
// for the 1 KB a wave has in flight.  Here a wave takes 64 samples of ONE table: lane i reads sample i's index (one
// coalesced request per query the tile touches) and finds its output row; lane group g then copies bags g G ..
// g G + G - 1, M = min(G, 8) rows in flight per lane (8 KB per wave at D 32), the row numbers and output rows
// coming over the cross-lane network.  The value stored is 0.0f + row, the sequential form's single addition:
// the same bits.  D == 4 G exactly.  BW = samples per wave: 64, or 16 for launches that would otherwise be a few dozen
// waves (one query of NCF: 4 tables x 256 samples) -- lanes 0 .. 15 fetch the indices then.
template <int G, int BW>
__global__ __launch_bounds__(64) void sls_one_kernel(SlsArgs a, int tiles) {
  constexpr int NG = 64 / G, PER = BW / NG;          // bags a lane group copies
  constexpr int M = PER < 8 ? PER : 8;               // ... M at a time
  static_assert(PER >= 1 && PER % M == 0, "whole rounds");
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();
  const int lane = threadIdx.x;
  const int g = lane / G, gl = lane - g * G;
  const int n_smp = a.q.cum[a.q.n_q];
  const int t = (int)blockIdx.x / tiles;                   // (uniform: table bases and row counts are scalar loads)
  const int smp = ((int)blockIdx.x - t * tiles) * BW + lane;
  const bool ok = lane < BW && smp < n_smp;
  int b = smp, vrow = a.q.vstart[0] + smp;
  const int32_t* qidx = a.idx[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    b = in ? smp - a.q.cum[i] : b;
    vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
    qidx = in ? a.idx[i] : qidx;
  }
  if (a.q.n_q > 8) {
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < a.q.n_q && smp >= a.q.cum[i];
      b = in ? smp - a.q.cum[i] : b;
      vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
      qidx = in ? a.idx[i] : qidx;
    }
  }
  const uint32_t rows = (uint32_t)a.tab_rows[t];
  uint32_t r = ok ? (uint32_t)qidx[(int64_t)t * a.idx_stride + b] : 0u;
  const bool bad = r >= rows;                               // Caffe2's ENFORCE: flag it, contribute zero
  if (bad) atomicOr(a.err, 1);
  r = bad ? 0u : r;
  // what the copying lanes need of sample i: its row number, and its output row (-1: nothing to store)
  const int dst = ok ? vrow : -1;
  const int keep = bad ? 0 : 1;
  const float4* __restrict__ W = reinterpret_cast<const float4*>(a.tables + a.tab_off[t]) + gl;
  float* __restrict__ out = a.out + a.col0 + (int64_t)t * (4 * G) + gl * 4;
#pragma unroll
  for (int j0 = 0; j0 < PER; j0 += M) {
    float4 v[M];
    int vr[M], kp[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int src = g * PER + j0 + j;
      const uint32_t rj = (uint32_t)__shfl((int)r, src);
      vr[j] = __shfl(dst, src);
      kp[j] = __shfl(keep, src);
      v[j] = W[(uint64_t)(rj * (uint32_t)G)];                // rows * D / 4 < 2^32 (enforced at table creation)
    }
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (vr[j] >= 0) {
        const float4 o = make_float4(0.f + (kp[j] ? v[j].x : 0.f), 0.f + (kp[j] ? v[j].y : 0.f),
                                     0.f + (kp[j] ? v[j].z : 0.f), 0.f + (kp[j] ? v[j].w : 0.f));
        *reinterpret_cast<float4*>(out + (int64_t)vr[j] * a.ld_out) = o;
      }
  }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}

// ---------------------------------------------------------------------------
// FLAT variant: fixed-length bags, G lanes per row (16 B per lane), NL loads per lane, BPW
// bags (same sample, consecutive tables) per wave.  Requires L * BPW <= NL * (64 / G) and
// T % BPW == 0 (checked by launch_sls).
template <int G, int NL, int BPW, bool NT = false>
__global__ __launch_bounds__(64) void sls_flat_kernel(SlsArgs a, int L, int xcd_order) {
  constexpr int NG = 64 / G;                       // lane groups = rows per load instruction
  // Work item w = (table group, sample), numbered TABLE-MAJOR; everything that depends only on
  // the wave (sample, query, tables) is scalar.  XCD-aware order (xcd_order != 0): workgroup id
  // lands on XCD id % 8 (observed dispatch order; speed only, never correctness), and XCD x walks
  // the contiguous slice [x*per, (x+1)*per) of the work list -- so one XCD's L2 and TLBs see one or
  // two tables (and contiguous pieces of their index arrays) instead of all T of them.
  const unsigned wg = blockIdx.x;
  if (a.ts && threadIdx.x == 0) a.ts[2 * wg] = wall_clock64();
  const unsigned n_smp = (unsigned)a.q.cum[a.q.n_q];
  const unsigned n_work = n_smp * (unsigned)(a.T / BPW);
  unsigned w = wg;
  if (xcd_order) {
    const unsigned per = (n_work + 7u) >> 3;
    w = (wg & 7u) * per + (wg >> 3);
    if ((wg >> 3) >= per || w >= n_work) {
      if (a.ts && threadIdx.x == 0) a.ts[2 * wg + 1] = a.ts[2 * wg];   // keep (min, max) well defined
      return;
    }
  }
  const unsigned tg = (unsigned)__builtin_amdgcn_readfirstlane((int)(w / n_smp));

  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const int col = min(gl * 4, a.D - 4);            // clamp idle lanes onto valid columns
  const bool col_ok = gl * 4 < a.D;

  const int smp = (int)(w - tg * n_smp);
  const int t0 = (int)tg * BPW;
  int b = smp, vrow = a.q.vstart[0] + smp;
  const int32_t* qidx = a.idx[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    b = in ? smp - a.q.cum[i] : b;
    vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
    qidx = in ? a.idx[i] : qidx;
  }
  if (a.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < a.q.n_q && smp >= a.q.cum[i];
      b = in ? smp - a.q.cum[i] : b;
      vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
      qidx = in ? a.idx[i] : qidx;
    }
  }
  const int R = BPW * L;
  const uint32_t D4 = (uint32_t)a.D >> 2;          // row stride in 16-byte units: rows * D / 4 < 2^32 (enforced at table creation)
  // table bases and row counts of the wave's BPW tables: scalar loads, issued now and waited
  // for only when the row addresses are formed, i.e. in the shadow of the index loads.  (Left
  // to the compiler they become vector loads -- it cannot prove the arrays are not written by
  // this kernel -- and cost a dependent round trip BEFORE the index loads.)
  uint64_t tab_off_k[BPW], tab_rows_k[BPW];
  {
    const uint32_t boff = (uint32_t)__builtin_amdgcn_readfirstlane(t0) * 8u;
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
      asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(tab_off_k[k]) : "s"(a.tab_off), "s"(boff + 8u * k));
      asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(tab_rows_k[k]) : "s"(a.tab_rows), "s"(boff + 8u * k));
    }
  }
  // which of the wave's bags does flattened row j belong to (j < R)
  auto bag_of = [&](int j) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < BPW; ++q) k += j >= q * L ? 1 : 0;
    return k;
  };

  // ---- phase 1: the index of every row this lane will load.  The G lanes of a group read the
  // same word and the 64/G groups adjacent words: one 32..128-B segment per instruction, all NL
  // of them in flight together -------------------------------------------------------------
  const int32_t* ip[NL];
  int kj[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int jj = min(g + NG * u, R - 1);
    kj[u] = bag_of(jj);
    ip[u] = qidx + (int64_t)(t0 + kj[u]) * a.idx_stride + (int64_t)b * L + (jj - kj[u] * L);
  }
  __builtin_amdgcn_sched_barrier(0);
  uint32_t ridx[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) ridx[u] = (uint32_t)*ip[u];
  __builtin_amdgcn_sched_barrier(0);
  // (the s_loads above: not tracked by the compiler's counters)
  if (BPW == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tab_off_k[0]), "+s"(tab_rows_k[0]));
  else if (BPW == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tab_off_k[0]), "+s"(tab_rows_k[0]), "+s"(tab_off_k[BPW > 1 ? 1 : 0]), "+s"(tab_rows_k[BPW > 1 ? 1 : 0]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tab_off_k[0]), "+s"(tab_rows_k[0]), "+s"(tab_off_k[BPW > 1 ? 1 : 0]), "+s"(tab_rows_k[BPW > 1 ? 1 : 0]),
                    "+s"(tab_off_k[BPW > 2 ? 2 : 0]), "+s"(tab_rows_k[BPW > 2 ? 2 : 0]), "+s"(tab_off_k[BPW > 3 ? 3 : 0]), "+s"(tab_rows_k[BPW > 3 ? 3 : 0]));

  // ---- phase 2: range check (Caffe2 ENFORCE) and every row address of the wave ----------------
  const float* rp[NL];
  bool bad = false;
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const float* W = a.tables + tab_off_k[0];
    uint32_t rk = (uint32_t)tab_rows_k[0];
#pragma unroll
    for (int z = 1; z < BPW; ++z) {
      W = kj[u] == z ? a.tables + tab_off_k[z] : W;
      rk = kj[u] == z ? (uint32_t)tab_rows_k[z] : rk;
    }
    bad |= g + NG * u < R && ridx[u] >= rk;
    const uint32_t ro = (ridx[u] < rk ? ridx[u] : 0u) * D4 + ((uint32_t)col >> 2);
    rp[u] = W + ((uint64_t)ro << 2);
  }
  // ---- phase 3: all row loads, back to back, nothing else in between --------------------------
  __builtin_amdgcn_sched_barrier(0);
  float4 v[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    if constexpr (NT) v[u] = ld_nt(reinterpret_cast<const float4*>(rp[u]));
    else v[u] = *reinterpret_cast<const float4*>(rp[u]);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- phase 4: per-bag sums in arrival order, then the butterfly over the lane groups --------
  float4 acc[BPW];
#pragma unroll
  for (int k = 0; k < BPW; ++k) acc[k] = vzero4();
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int j = g + NG * u;
    if (BPW == 1) {
      vadd(acc[0], vsel<4>(j < R, v[u]));
    } else {
#pragma unroll
      for (int k = 0; k < BPW; ++k) vadd(acc[k], vsel<4>(j < R && kj[u] == k, v[u]));
    }
  }
#pragma unroll
  for (int k = 0; k < BPW; ++k)
#pragma unroll
    for (int m = G; m < 64; m <<= 1) vadd(acc[k], vshfl_xor(acc[k], m));

  if (bad) atomicOr(a.err, 1);
  // every group holds every sum after the butterfly; group 0 stores them, one 128..512-B row per
  // bag.  (Letting group k store bag k needs acc[g]: the optimiser turns that select chain into a
  // dynamically indexed array, i.e. SCRATCH memory -- which capped the BPW > 1 variants at half
  // the speed of BPW == 1 until it was spotted in the ISA.)
  if (col_ok && g == 0) {
    float* o = a.out + (int64_t)vrow * a.ld_out + a.col0 + (int64_t)t0 * a.D + col;
#pragma unroll
    for (int k = 0; k < BPW; ++k) *reinterpret_cast<float4*>(o + (int64_t)k * a.D) = acc[k];
  }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);   // include the output store in the span
    if (threadIdx.x == 0) a.ts[2 * wg + 1] = wall_clock64();
  }
}

// The first form of the flat variant, one bag per wave: ONE coalesced index read (lane i owns
// row i), indices handed to the loading lanes over the cross-lane network, and the row loads /
// sums left to the compiler's schedule -- which turns them into groups of four or five loads in
// flight with the sums of one group under the next.  Measured against the phased form above
// (everything in flight at once) on RMC1's 80 x 256-B bags beside the MLP launch: 0.74 vs 0.72 of
// peak for 8-query launches, 0.57-0.59 vs 0.51 for a single query; so one-bag-per-wave launches
// take this one ("sls_flat" 1) and the phased form serves the several-bags-per-wave shapes.
template <int G, int NL, bool NT>
__global__ __launch_bounds__(64) void sls_flatc_kernel(SlsArgs a, int L) {
  constexpr int BPW = 1;
  constexpr int NG = 64 / G;                       // lane groups = rows per load instruction
  constexpr int NI = (NL * NG + 63) / 64;          // index registers per lane
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();

  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const int col = min(gl * 4, a.D - 4);            // clamp idle lanes onto valid columns
  const bool col_ok = gl * 4 < a.D;

  // the wave's bags: all of one sample (T % BPW == 0), tables t0 .. t0+BPW-1 -- uniform
  const int64_t bag0 = (int64_t)blockIdx.x * BPW;
  const int smp = (int)(bag0 / a.T);
  const int t0 = (int)(bag0 - (int64_t)smp * a.T);
  int b = smp, vrow = a.q.vstart[0] + smp;
  const int32_t* qidx = a.idx[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    b = in ? smp - a.q.cum[i] : b;
    vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
    qidx = in ? a.idx[i] : qidx;
  }
  if (a.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < a.q.n_q && smp >= a.q.cum[i];
      b = in ? smp - a.q.cum[i] : b;
      vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
      qidx = in ? a.idx[i] : qidx;
    }
  }
  const int R = BPW * L;
  const uint32_t D4 = (uint32_t)a.D >> 2;          // row stride in 16-byte units: rows * D / 4 < 2^32 (enforced at table creation)
  const float* Wk[BPW];
  uint32_t rows_k[BPW];
#pragma unroll
  for (int k = 0; k < BPW; ++k) {
    Wk[k] = a.tables + a.tab_off[t0 + k] + col;
    rows_k[k] = (uint32_t)a.tab_rows[t0 + k];
  }
  // which of the wave's bags does flattened row j belong to (j < R)
  auto bag_of = [&](int j) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < BPW; ++q) k += j >= q * L ? 1 : 0;
    return k;
  };

  // ONE coalesced index read: lane i owns flattened rows i, i+64, ...; range check (Caffe2
  // ENFORCE) and the row's element offset inside its table are computed by the owner
  uint32_t roff[NI];
  bool bad = false;
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int i = lane + 64 * q;
    const int ii = min(i, R - 1);
    const int k = bag_of(ii);
    const int32_t* ip = qidx + (int64_t)(t0 + k) * a.idx_stride + (int64_t)b * L + (ii - k * L);
    uint32_t r = (uint32_t)*ip;
    uint32_t rk = rows_k[0];
#pragma unroll
    for (int z = 1; z < BPW; ++z) rk = k == z ? rows_k[z] : rk;
    bad |= i < R && r >= rk;
    r = r < rk ? r : 0u;
    roff[q] = r * D4;
  }

  // every row load of the wave, back to back
  float4 v[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int j = g + NG * u;                      // (j >> 6) == (NG * u) >> 6: compile time
    const uint32_t ro = (uint32_t)__shfl((int)roff[(NG * u) >> 6], j & 63);
    const float* W = Wk[0];
    if (BPW > 1) {
      const int k = bag_of(min(j, R - 1));
#pragma unroll
      for (int z = 1; z < BPW; ++z) W = k == z ? Wk[z] : W;
    }
    // NT ("sls_nt" 1): the rows are read once (~1 % reuse inside a batch): non-temporal loads
    if constexpr (NT) {
      v[u] = ld_nt(reinterpret_cast<const float4*>(W) + (uint64_t)ro);
    } else {
      v[u] = reinterpret_cast<const float4*>(W)[(uint64_t)ro];
    }
  }

  float4 acc[BPW];
#pragma unroll
  for (int k = 0; k < BPW; ++k) acc[k] = vzero4();
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int j = g + NG * u;
    if (BPW == 1) {
      vadd(acc[0], vsel<4>(j < R, v[u]));
    } else {
      const int kj = bag_of(min(j, R - 1));
#pragma unroll
      for (int k = 0; k < BPW; ++k) vadd(acc[k], vsel<4>(j < R && kj == k, v[u]));
    }
  }
#pragma unroll
  for (int k = 0; k < BPW; ++k)
#pragma unroll
    for (int m = G; m < 64; m <<= 1) vadd(acc[k], vshfl_xor(acc[k], m));

  if (bad) atomicOr(a.err, 1);
  // lane group k stores bag k (every group holds every sum after the butterfly)
  if (col_ok && g < BPW) {
    float4 o4 = acc[0];
#pragma unroll
    for (int k = 1; k < BPW; ++k) o4 = g == k ? acc[k] : o4;
    float* o = a.out + (int64_t)vrow * a.ld_out + a.col0 + (int64_t)(t0 + g) * a.D + col;
    *reinterpret_cast<float4*>(o) = o4;
  }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);   // include the output store in the span
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}

// ANY row width: the generic form behind the ABI's total boundary.  The reference only asks m_spa == ln_bot[-1]
// (models/dlrm_s_caffe2.py:435-437); every kernel above reads a row as 16-byte pieces (D % 4 == 0, D <= 256), which
// every shipped config satisfies.  Other widths -- D = 10, 50, 300 -- take this one: a wave per bag, lane c takes columns
// c, c + 64, ... (dword loads: rows need no alignment), rows strictly in index order, i.e. Caffe2's own summation order
// (bit-identical to the oracle), ragged bags through the prefix sums.  Slow by design (one row at a time), never wrong.
__global__ __launch_bounds__(64) void sls_any_kernel(SlsArgs a) {
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();
  const int lane = threadIdx.x;
  const int64_t bag = blockIdx.x;
  const int smp = (int)(bag / a.T);
  const int t = (int)(bag - (int64_t)smp * a.T);
  int b = smp, vrow = a.q.vstart[0] + smp, ulen = a.uniform_len[0];
  const int32_t* qidx = a.idx[0];
  const int32_t* qoff = a.off[0];
#pragma unroll
  for (int i = 1; i < DRS_MAX_COALESCE; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    b = in ? smp - a.q.cum[i] : b;
    vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : vrow;
    ulen = in ? a.uniform_len[i] : ulen;
    qidx = in ? a.idx[i] : qidx;
    qoff = in ? a.off[i] : qoff;
  }
  int beg, end;
  if (ulen >= 0) {
    beg = b * ulen;
    end = beg + ulen;
  } else {
    const int32_t* __restrict__ offp = qoff + (int64_t)t * a.off_stride;
    beg = offp[b];
    end = offp[b + 1];
  }
  const int32_t* __restrict__ ip = qidx + (int64_t)t * a.idx_stride;
  const float* __restrict__ W = a.tables + a.tab_off[t];
  const uint32_t rows = (uint32_t)a.tab_rows[t];
  const int D = a.D;
  float* o = a.out + (int64_t)vrow * a.ld_out + a.col0 + (int64_t)t * D;
  bool bad = false;
  for (int c0 = 0; c0 < D; c0 += 64 * 4) {          // four columns per lane and pass
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = beg; j < end; ++j) {
      uint32_t r = (uint32_t)ip[j];
      bad |= r >= rows;
      r = r < rows ? r : 0u;
      const float* row = W + (int64_t)r * D;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + lane + 64 * k;
        acc[k] += c < D ? row[c] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + lane + 64 * k;
      if (c < D) o[c] = acc[k];
    }
  }
  if (bad) atomicOr(a.err, 1);
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}

// stop: optional event recorded BY the kernel dispatch itself (its completion signal) -- no
// separate marker packet between this launch and the next one on the stream
template <typename K, typename... X>
void launch_kb(K kernel, dim3 grid, dim3 block, hipStream_t s, hipEvent_t stop, const X&... x) {
  if (stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, nullptr, stop, 0, x...);
  else hipLaunchKernelGGL(kernel, grid, block, 0, s, x...);
}
template <typename K, typename... X>
void launch_k(K kernel, dim3 grid, hipStream_t s, hipEvent_t stop, const X&... x) {
  launch_kb(kernel, grid, dim3(64), s, stop, x...);
}

template <int G, int V, int U>
hipError_t launch_variant(const SlsArgs& a, int exact, int nt, hipStream_t s, hipEvent_t stop) {
  const int64_t n_bags = (int64_t)a.q.cum[a.q.n_q] * a.T;
  if (n_bags == 0) return hipSuccess;
  if (exact) {
    constexpr int BAGS = 64 / G;
    const unsigned grid = (unsigned)((n_bags + BAGS - 1) / BAGS);
    launch_k(sls_kernel<G, V, U, true>, grid, s, stop, a);
  } else if (nt) {
    launch_k(sls_kernel<G, V, U, false, true>, (unsigned)n_bags, s, stop, a);
  } else {
    launch_k(sls_kernel<G, V, U, false>, (unsigned)n_bags, s, stop, a);
  }
  return hipGetLastError();
}

// U (row loads per register ring and lane) is 4: two rings, so 4..8 loads in flight per lane, the waves per
// CU provide the rest of the memory-level parallelism.  (8, 16 and 20 were options until round 4 -- measured
// equal or slower on every shape -- as was a 16-lane x 8-byte form for D == 32.)
template <int G, int V>
hipError_t launch_u(const SlsArgs& a, int exact, int nt, hipStream_t s, hipEvent_t stop) {
  return launch_variant<G, V, 4>(a, exact, nt, s, stop);
}

int lanes_per_row(int D) { return D <= 8 ? 2 : D <= 16 ? 4 : D <= 32 ? 8 : D <= 64 ? 16 : D <= 128 ? 32 : 64; }

// Shape of the flat variant for this launch, or ok == false: every coalesced query must have
// the same fixed bag length L >= 1, G must be one of the instantiated widths, BPW must divide T
// (a wave's bags belong to one sample) and BPW * L rows must fit NL loads per lane.
struct FlatPlan {
  bool ok = false;
  int G = 0, NL = 0, BPW = 1, L = 0, xcd = 1, coal = 0, nt = 0;
  unsigned grid = 0;
};
FlatPlan flat_plan(const SlsArgs& a, const Tune& tune) {
  FlatPlan p;
  if (!tune.sls_flat || a.q.n_q < 1) return p;
  const int L = a.uniform_len[0];
  for (int i = 1; i < a.q.n_q; ++i) if (a.uniform_len[i] != L) return p;
  if (L < 2) return p;                      // L == 1 is a copy: the lane-group-per-bag kernel
  const int G = lanes_per_row(a.D);
  if (G != 8 && G != 16 && G != 32) return p;
  const int NG = 64 / G;
  int bpw = 1;
  if (tune.sls_bpw > 0) {
    bpw = tune.sls_bpw;
    if ((bpw != 1 && bpw != 2 && bpw != 4) || a.T % bpw) return p;
  } else {
    // short bags share a wave until it has five loads per lane to issue (measured on RM3,
    // 12 x 10M x 32, L = 20, beside its GEMM launches: 2 bags per wave 0.48 of peak, 1 or 4 bags
    // 0.44; the chip to itself: 0.60 / 0.55 / 0.59)
    for (int c : {4, 2})
      if (a.T % c == 0 && c * L <= 5 * NG) { bpw = c; break; }
  }
  const int need = (bpw * L + NG - 1) / NG;
  // (30 loads per lane -- RM2: L = 120, D = 64 -- in the one-bag-per-wave form was measured in round 4: 497.6 us per
  // launch against the ring walk's 497.4 us; not kept)
  const int nl = need <= 5 ? 5 : need <= 10 ? 10 : need <= 20 ? 20 : 0;
  if (!nl || (bpw > 1 && nl > 10)) return p;
  p.ok = true; p.G = G; p.NL = nl; p.BPW = bpw; p.L = L; p.xcd = 1;
  p.coal = bpw == 1 && tune.sls_flat == 1;      // "sls_flat" 2 forces the phased form
  p.nt = tune.sls_nt;
  const unsigned n_work = (unsigned)a.q.cum[a.q.n_q] * (unsigned)(a.T / bpw);
  p.grid = p.coal ? n_work : (p.xcd ? 8u * ((n_work + 7u) / 8u) : n_work);
  return p;
}

template <int G, int NL>
hipError_t launch_flat_b(const SlsArgs& a, const FlatPlan& p, dim3 grid, hipStream_t s, hipEvent_t stop) {
  if (p.coal && p.nt) launch_k(sls_flatc_kernel<G, NL, true>, grid, s, stop, a, p.L);
  else if (p.coal) launch_k(sls_flatc_kernel<G, NL, false>, grid, s, stop, a, p.L);
  else if (p.nt) {
    if (p.BPW == 1) launch_k(sls_flat_kernel<G, NL, 1, true>, grid, s, stop, a, p.L, p.xcd);
    else if constexpr (NL <= 10) {
      if (p.BPW == 2) launch_k(sls_flat_kernel<G, NL, 2, true>, grid, s, stop, a, p.L, p.xcd);
      else launch_k(sls_flat_kernel<G, NL, 4, true>, grid, s, stop, a, p.L, p.xcd);
    }
  }
  else if (p.BPW == 1) launch_k(sls_flat_kernel<G, NL, 1>, grid, s, stop, a, p.L, p.xcd);
  else if constexpr (NL <= 10) {
    if (p.BPW == 2) launch_k(sls_flat_kernel<G, NL, 2>, grid, s, stop, a, p.L, p.xcd);
    else launch_k(sls_flat_kernel<G, NL, 4>, grid, s, stop, a, p.L, p.xcd);
  }
  return hipGetLastError();
}
template <int G>
hipError_t launch_flat_g(const SlsArgs& a, const FlatPlan& p, dim3 grid, hipStream_t s, hipEvent_t stop) {
  switch (p.NL) {
    case 5: return launch_flat_b<G, 5>(a, p, grid, s, stop);
    case 10: return launch_flat_b<G, 10>(a, p, grid, s, stop);
    default: return launch_flat_b<G, 20>(a, p, grid, s, stop);
  }
}
hipError_t launch_flat(const SlsArgs& a, const FlatPlan& p, hipStream_t s, hipEvent_t stop) {
  const int64_t n_bags = (int64_t)a.q.cum[a.q.n_q] * a.T;
  if (n_bags == 0) return hipSuccess;
  const dim3 grid(p.grid);
  switch (p.G) {
    case 8: return launch_flat_g<8>(a, p, grid, s, stop);
    case 16: return launch_flat_g<16>(a, p, grid, s, stop);
    default: return launch_flat_g<32>(a, p, grid, s, stop);
  }
}

}  // namespace

// Tunables (drs_set_option, kept per engine in Tune): "sls_flat" / "sls_bpw" the flat variant and its bags
// per wave (0 = auto), "sls_nt" non-temporal row loads.
static inline bool any_width(int D) { return (D & 3) || D > 256; }      // widths only sls_any_kernel takes
// the one-lookup copy form: every coalesced query has fixed bags of ONE row, a row is 4 / 8 / 16 / 32 lanes x 16 B
static inline bool one_lookup(const SlsArgs& a, const Tune& tune) {
  if (!tune.sls_one || a.q.n_q < 1 || !(a.D == 16 || a.D == 32 || a.D == 64 || a.D == 128)) return false;
  for (int i = 0; i < a.q.n_q; ++i) if (a.uniform_len[i] != 1) return false;
  return true;
}
// samples per wave: 64, unless that leaves the launch under 1 024 waves ("sls_one" 64 / 16 force one)
static inline int one_lookup_tile(const SlsArgs& a, const Tune& tune) {
  if (tune.sls_one == 64 || tune.sls_one == 16) return tune.sls_one;
  return (int64_t)a.T * ((a.q.cum[a.q.n_q] + 63) / 64) < 1024 ? 16 : 64;
}
static inline int64_t one_lookup_grid(const SlsArgs& a, const Tune& tune) {
  const int bw = one_lookup_tile(a, tune);
  return (int64_t)a.T * ((a.q.cum[a.q.n_q] + bw - 1) / bw);
}

bool sls_flat_applicable(const SlsArgs& a, const Tune& tune) { return !any_width(a.D) && flat_plan(a, tune).ok; }

int64_t sls_grid_blocks(const SlsArgs& a, int exact, const Tune& tune) {
  const int64_t n_bags = (int64_t)a.q.cum[a.q.n_q] * a.T;
  if (any_width(a.D)) return n_bags;
  if (!exact) {
    const FlatPlan p = flat_plan(a, tune);
    return p.ok ? (int64_t)p.grid : n_bags;
  }
  if (one_lookup(a, tune)) return one_lookup_grid(a, tune);
  int G = lanes_per_row(a.D);
  const int bags = 64 / G;
  return (n_bags + bags - 1) / bags;
}

hipError_t launch_sls(const SlsArgs& a, int exact, const Tune& tune, hipStream_t s, hipEvent_t stop) {
  const int D = a.D;
  if (D <= 0) return hipErrorInvalidValue;
  if (any_width(D)) {            // the generic form: any width, sequential order
    const int64_t n_bags = (int64_t)a.q.cum[a.q.n_q] * a.T;
    if (n_bags == 0) return hipSuccess;
    log_launch(tune.log, "sls_any_kernel[%lld wg, D=%d]", (long long)n_bags, D);
    launch_k(sls_any_kernel, dim3((unsigned)n_bags), s, stop, a);
    return hipGetLastError();
  }
  if (!exact) {
    const FlatPlan p = flat_plan(a, tune);
    if (p.ok) {
      log_launch(tune.log, "%s<%d,%d%s%s>[%u wg, L=%d]", p.coal ? "sls_flatc_kernel" : "sls_flat_kernel", p.G, p.NL,
                 p.coal ? "" : (p.BPW == 4 ? ",bpw4" : p.BPW == 2 ? ",bpw2" : ",bpw1"), p.nt ? ",nt" : "", p.grid, p.L);
      return launch_flat(a, p, s, stop);
    }
  }
  if (exact && one_lookup(a, tune)) {
    const int bw = one_lookup_tile(a, tune);
    const int tiles = (a.q.cum[a.q.n_q] + bw - 1) / bw;
    if (tiles == 0) return hipSuccess;
    const dim3 grid((unsigned)one_lookup_grid(a, tune));
    log_launch(tune.log, "sls_one_kernel<%d,%d>[%u wg]", D / 4, bw, grid.x);
    if (bw == 64) {
      if (D == 16) launch_k(sls_one_kernel<4, 64>, grid, s, stop, a, tiles);
      else if (D == 32) launch_k(sls_one_kernel<8, 64>, grid, s, stop, a, tiles);
      else if (D == 64) launch_k(sls_one_kernel<16, 64>, grid, s, stop, a, tiles);
      else launch_k(sls_one_kernel<32, 64>, grid, s, stop, a, tiles);
    } else {
      if (D == 16) launch_k(sls_one_kernel<4, 16>, grid, s, stop, a, tiles);
      else if (D == 32) launch_k(sls_one_kernel<8, 16>, grid, s, stop, a, tiles);
      else if (D == 64) launch_k(sls_one_kernel<16, 16>, grid, s, stop, a, tiles);
      else launch_k(sls_one_kernel<32, 16>, grid, s, stop, a, tiles);
    }
    return hipGetLastError();
  }
  log_launch(tune.log, "sls_kernel<%d,%s>[%lld wg]", lanes_per_row(D), exact ? "sequential" : (tune.sls_nt ? "split,nt" : "split"),
             (long long)sls_grid_blocks(a, exact, tune));
  // the non-temporal hint is for bags of many rows out of big tables; the one-lookup models (W&D, NCF, MT-WnD:
  // the sequential form) keep their rows cacheable -- NCF's tables live in the Infinity Cache (measured: -3 % with it)
  const int nt = exact ? 0 : tune.sls_nt;
  if (D <= 8) return launch_u<2, 4>(a, exact, nt, s, stop);
  if (D <= 16) return launch_u<4, 4>(a, exact, nt, s, stop);
  if (D <= 32) return launch_u<8, 4>(a, exact, nt, s, stop);
  if (D <= 64) return launch_u<16, 4>(a, exact, nt, s, stop);
  if (D <= 128) return launch_u<32, 4>(a, exact, nt, s, stop);
  return launch_u<64, 4>(a, exact, nt, s, stop);
}

// ---------------------------------------------------------------------------
// device-side table fill, bit-identical to oracle/drs_oracle.c fill_value()
__global__ void fill_uniform_kernel(float* W, int64_t n, int32_t t, float lo, float hi,
                                    uint64_t seed) {
  const float span = hi - lo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)i + ((uint64_t)(uint32_t)t << 40) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
    W[i] = __fmaf_rn(u, span, lo);
  }
}

#ifdef DRS_LAB   // probes of tools/placement_lab.py (libdrs_hip_lab.so: make lab-lib); not in the product library
// ---------------------------------------------------------------------------
// Row-read probe of a range of device memory: the access shape of the many-rows-per-bag gather (a wave =
// 80 random 256-byte rows, 4 per load instruction, all 20 loads of a lane in flight, non-temporal) without
// index arrays or outputs.  What it is for: DESIGN.md 5 -- the same gather runs up to 9 % faster on some
// gigabytes of HBM than on others, and the arena builder keeps the gigabytes this probe reads fastest.
// G lanes per row (16 bytes each): 16 = RMC1's 256-byte rows (a wave = 80 rows), 8 = 128-byte rows (dlrm_rm1.json / RM3:
// a wave = 160 rows, eight per load instruction), 32 = 512-byte rows; NT: non-temporal loads (a compile-time property).
template <int G, bool NT, int NLD>
__global__ __launch_bounds__(64) void probe_rows_kernel(const float4* __restrict__ base, uint32_t rows, uint32_t seed,
                                                        float4* __restrict__ sink, uint32_t windows, uint32_t stride_rows, int sorted) {
  // windows == 0: the wave's 80 rows anywhere in [0, rows).  windows > 0: wave w reads inside window w % windows
  // (window k = rows [k * stride_rows, k * stride_rows + rows)), like a bag of one table; sorted: ascending, one row per
  // eightieth of the window (what np.unique leaves of a bag's indices)
  constexpr int NG = 64 / G;                       // rows per load instruction
  constexpr uint32_t RW = NLD * NG;                 // rows per wave
  const int lane = threadIdx.x, g = lane / G, gl = lane % G;
  const uint64_t w0 = windows ? (uint64_t)(blockIdx.x % windows) * stride_rows : 0;
  uint32_t r[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    uint32_t z = (blockIdx.x * RW + (uint32_t)(NG * u + g)) * 0x9E3779B1u + seed;
    z = (z ^ (z >> 16)) * 0x85EBCA6Bu;
    z = (z ^ (z >> 13)) * 0xC2B2AE35u;
    z ^= z >> 16;
    if (sorted) {
      const uint32_t lo = (uint32_t)(((uint64_t)(NG * u + g) * rows) / RW), hi = (uint32_t)(((uint64_t)(NG * u + g + 1) * rows) / RW);
      r[u] = lo + (uint32_t)(((uint64_t)z * (hi > lo ? hi - lo : 1)) >> 32);
    } else {
      r[u] = (uint32_t)(((uint64_t)z * rows) >> 32);
    }
  }
  float4 v[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    if constexpr (NT) v[u] = ld_nt(base + (w0 + r[u]) * G + gl);
    else v[u] = base[(w0 + r[u]) * G + gl];
  }
  float4 acc = vzero4();
#pragma unroll
  for (int u = 0; u < NLD; ++u) vadd(acc, v[u]);
  if (acc.x == 1.2345e-30f) sink[lane] = acc;      // (keeps the loads alive; never true for table data)
}

// time `reps` launches of `waves` waves over [base, base + bytes); GB/s of row bytes in *gbs
hipError_t probe_rows(const void* base, size_t bytes, int waves, int reps, float* sink, hipStream_t s, double* gbs,
                      int windows, int sorted, int row_bytes, int nt, int loads) {
  *gbs = 0;
  if (row_bytes != 128 && row_bytes != 256 && row_bytes != 512) return hipErrorInvalidValue;
  uint64_t rows = bytes / (uint64_t)row_bytes, stride = 0;
  if (windows > 0) { stride = rows / (uint64_t)windows; rows = stride; }
  if (loads != 10 && loads != 20) return hipErrorInvalidValue;
  const int G = row_bytes / 16, rw = loads * (64 / G);
  if (rows < (uint64_t)rw || rows > 0xffffffffull) return hipErrorInvalidValue;
  auto launch = [&](uint32_t seed) {
#define DRS_PROBE(G_, NT_)                                                                                                               \
  do {                                                                                                                                     \
    if (loads == 10) hipLaunchKernelGGL((probe_rows_kernel<G_, NT_, 10>), dim3((unsigned)waves), dim3(64), 0, s, static_cast<const float4*>(base), \
                                        (uint32_t)rows, seed, reinterpret_cast<float4*>(sink), (uint32_t)windows, (uint32_t)stride, sorted); \
    else hipLaunchKernelGGL((probe_rows_kernel<G_, NT_, 20>), dim3((unsigned)waves), dim3(64), 0, s, static_cast<const float4*>(base),       \
                            (uint32_t)rows, seed, reinterpret_cast<float4*>(sink), (uint32_t)windows, (uint32_t)stride, sorted);            \
  } while (0)
    if (G == 8) { if (nt) DRS_PROBE(8, true); else DRS_PROBE(8, false); }
    else if (G == 16) { if (nt) DRS_PROBE(16, true); else DRS_PROBE(16, false); }
    else { if (nt) DRS_PROBE(32, true); else DRS_PROBE(32, false); }
#undef DRS_PROBE
  };
  hipEvent_t e0, e1;
  hipError_t r = hipEventCreate(&e0);
  if (r != hipSuccess) return r;
  r = hipEventCreate(&e1);
  if (r != hipSuccess) { (void)hipEventDestroy(e0); return r; }
  launch(1u);
  r = hipEventRecord(e0, s);
  for (int i = 0; i < reps && r == hipSuccess; ++i) {
    launch(0x51ED27u * (uint32_t)(i + 2));
    r = hipGetLastError();
  }
  if (r == hipSuccess) r = hipEventRecord(e1, s);
  if (r == hipSuccess) r = hipEventSynchronize(e1);
  float ms = 0.f;
  if (r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
  if (r == hipSuccess && ms > 0.f) *gbs = (double)waves * (double)rw * (double)row_bytes * reps / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return r;
}

// Latency probe: workgroup k walks `steps` DEPENDENT 16-byte loads through random 256-byte rows of chunk k
// ([base + k * chunk_bytes, + chunk_bytes)); out[k] = device-clock ticks (100 MHz) for the walk.  One lane per
// chunk, all chunks at once: the walks do not disturb each other, and a launch is steps x ~1 us long.
__global__ __launch_bounds__(64) void probe_latency_kernel(const float4* __restrict__ base, uint64_t chunk_rows, uint32_t rows,
                                                           int steps, uint32_t seed, uint64_t* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const float4* b = base + (uint64_t)blockIdx.x * chunk_rows * 16;
  uint32_t z = seed + blockIdx.x * 0x9E3779B1u;
  float sink = 0.f;
  const uint64_t t0 = wall_clock64();
  for (int i = 0; i < steps; ++i) {
    z = (z ^ (z >> 16)) * 0x85EBCA6Bu;
    z = (z ^ (z >> 13)) * 0xC2B2AE35u;
    z ^= z >> 16;
    const uint32_t r = (uint32_t)(((uint64_t)z * rows) >> 32);
    const float4 v = ld_nt(b + (uint64_t)r * 16 + (z & 15));
    z += __float_as_uint(v.x) | 1u;                  // the next address depends on the loaded value
    sink += v.y;
  }
  const uint64_t t1 = wall_clock64();
  out[blockIdx.x] = t1 - t0 + (sink == 1.2345e-30f ? 1 : 0);
}

hipError_t probe_latency(const void* base, size_t chunk_bytes, int n_chunks, int steps, uint64_t* d_ticks, hipStream_t s) {
  const uint64_t rows = chunk_bytes / 256;
  if (rows < 1 || rows > 0xffffffffull || n_chunks < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(probe_latency_kernel, dim3((unsigned)n_chunks), dim3(64), 0, s, static_cast<const float4*>(base), rows,
                     (uint32_t)rows, steps, 12345u, d_ticks);
  return hipGetLastError();
}

#endif  // DRS_LAB

hipError_t launch_fill_uniform(float* W, int64_t n, int32_t t, float lo, float hi, uint64_t seed,
                               hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t want = (n + 255) / 256;
  const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid), dim3(256), 0, s, W, n, t, lo, hi, seed);
  return hipGetLastError();
}

}  // namespace drs

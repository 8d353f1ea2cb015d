// Deep Interest Network (models/din.py:247-330): the attention units over the pooled behaviour
// embeddings.  Tables = [user profile | U behaviour tables | candidate ad | context]; per behaviour
// table i ONE attention unit with its own weights,
//     y_i = relu(W1_i . Concat(u_i, ad, u_i + ad) + b1_i)      [h]        (models/din.py:262-277)
//     o_i = relu(W2_i . y_i + b2_i)                             [D]
//     atten_out = Sum_i o_i                                                (:280)
//     top MLP input = Concat(profile, atten_out, ad, context)  [4 D]      (:311-318)
// Not MFMA work: every unit has different weights (K = 3 D, N = h = 1 in the shipped config), the
// arithmetic is 64 k FLOP per sample next to 97 KB of gathered rows -- the path is HBM-bound on the
// row gather, so the default launch FUSES gather, units and Concat: the [rows, T * D] pooled
// tensor (32 KB per sample, a third of the gathered bytes) is never written or re-read.
#include <hip/hip_ext.h>
#include <string.h>

#include "drs_internal.h"
#include "mlp_dev.h"
#include "rnn_dev.h"

namespace drs {
namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// a table row piece: read once per launch -- non-temporal when NT ("din_nt"; a compile-time property: a
// run-time select between the two loads can be merged into one plain load, sls.hip)
template <bool NT>
__device__ __forceinline__ float4 ld4row(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  if constexpr (NT) {
    const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
  return acc;
}
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float4 shfl_xor4(const float4& a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}

}  // namespace

// Packed weights of one unit, every piece 16-B aligned (D % 4 == 0):
//   [ W1 : h x 3D | W2 : D x h | b2 : D | b1 : h, padded to 4 ]
int64_t din_unit_stride(int D, int h) { return (int64_t)3 * D * h + (int64_t)D * h + D + (h + 3) / 4 * 4; }

namespace {

__global__ __launch_bounds__(256) void din_pack_kernel(const float* const* __restrict__ att, float* __restrict__ packed,
                                                       int D, int h, int64_t stride) {
  const int i = blockIdx.x;
  float* o = packed + (int64_t)i * stride;
  const float* W1 = att[4 * i + 0];
  const float* b1 = att[4 * i + 1];
  const float* W2 = att[4 * i + 2];
  const float* b2 = att[4 * i + 3];
  const int n1 = 3 * D * h, n2 = D * h;
  for (int k = threadIdx.x; k < n1; k += blockDim.x) o[k] = W1[k];
  for (int k = threadIdx.x; k < n2; k += blockDim.x) o[n1 + k] = W2[k];
  for (int k = threadIdx.x; k < D; k += blockDim.x) o[n1 + n2 + k] = b2[k];
  for (int k = threadIdx.x; k < (h + 3) / 4 * 4; k += blockDim.x) o[n1 + n2 + D + k] = k < h ? b1[k] : 0.f;
}

// The two-launch form (sequential-order mode, and shapes the fused kernel does not cover): T is the
// gather's pooled tensor [M, Tn * D].  One wave per sample.  Phase 1: lane l evaluates the first
// layer of units l, l + 64, ... as the oracle's FC does (k-ordered fmaf chain over the Concat,
// bias after, ReLU); phase 2: lane j < D owns output column j and walks the units IN ORDER (the
// reference's Sum over fc_outs): second layer = chain over the h hidden values, bias, ReLU.
// Bit-identical to oracle/drs_oracle.c's DIN branch.
__global__ __launch_bounds__(256) void din_attention_kernel(const float* __restrict__ T, int64_t ldt, int64_t M,
                                                            int Tn, int D, int h, const float* __restrict__ packed,
                                                            int64_t stride, float* __restrict__ R, int64_t ldr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int U = Tn - 3;
  float* sy = smem + (size_t)wave * U * h;          // [U][h] hidden values of this sample
  if (row >= M) return;                             // whole wave exits together
  const float* e = T + row * ldt;
  const float* ad = e + (int64_t)(Tn - 2) * D;
  const int n1 = 3 * D * h, n2 = D * h;
  for (int i0 = 0; i0 < U; i0 += 64) {
    const int i = min(i0 + lane, U - 1);
    const float* u = e + (int64_t)(1 + i) * D;
    const float* W1 = packed + (int64_t)i * stride;
    const float* b1 = W1 + n1 + n2 + D;
    for (int hh = 0; hh < h; ++hh) {
      const float* w = W1 + (int64_t)hh * 3 * D;
      float acc = 0.f;
      for (int k = 0; k < D; ++k) acc = fmaf(u[k], w[k], acc);
      for (int k = 0; k < D; ++k) acc = fmaf(ad[k], w[D + k], acc);
      for (int k = 0; k < D; ++k) acc = fmaf(u[k] + ad[k], w[2 * D + k], acc);
      const float y = acc + b1[hh];
      if (i0 + lane < U) sy[i * h + hh] = y > 0.f ? y : 0.f;
    }
  }
  __builtin_amdgcn_wave_barrier();
  float* out = R + row * ldr;
  for (int j = lane; j < D; j += 64) {
    float z = 0.f;
    for (int i = 0; i < U; ++i) {
      const float* W2 = packed + (int64_t)i * stride + n1;
      const float* b2 = W2 + n2;
      float acc = 0.f;
      for (int hh = 0; hh < h; ++hh) acc = fmaf(sy[i * h + hh], W2[(int64_t)j * h + hh], acc);
      float o = acc + b2[j];
      o = o > 0.f ? o : 0.f;
      z = i == 0 ? o : z + o;
    }
    out[D + j] = z;
    out[j] = e[j];
    out[2 * D + j] = ad[j];
    out[3 * D + j] = e[(int64_t)(Tn - 1) * D + j];
  }
}

// Who owns valid-sample number `smp` of a coalesced launch set (select chain over <= DRS_MAX_COALESCE entries,
// wave-uniform: no dynamic indexing of the kernel-argument arrays).
struct Owner {
  int b, vrow, ulen;
  const int32_t* idx;
  const int32_t* off;
};
__device__ __forceinline__ Owner owner_of(const SlsArgs& a, int smp) {
  Owner o = {smp, a.q.vstart[0] + smp, a.uniform_len[0], a.idx[0], a.off[0]};
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < a.q.n_q && smp >= a.q.cum[i];
    o.b = in ? smp - a.q.cum[i] : o.b;
    o.vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : o.vrow;
    o.ulen = in ? a.uniform_len[i] : o.ulen;
    o.idx = in ? a.idx[i] : o.idx;
    o.off = in ? a.off[i] : o.off;
  }
  if (a.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < a.q.n_q && smp >= a.q.cum[i];
      o.b = in ? smp - a.q.cum[i] : o.b;
      o.vrow = in ? a.q.vstart[i] + smp - a.q.cum[i] : o.vrow;
      o.ulen = in ? a.uniform_len[i] : o.ulen;
      o.idx = in ? a.idx[i] : o.idx;
      o.off = in ? a.off[i] : o.off;
    }
  }
  return o;
}

// FUSED gather + attention units + Concat.  A workgroup of NW waves serves S samples; its waves
// split into lane groups of G = D / 4 lanes (one 16-B piece of a row each), and lane group gg
// owns units gg, gg + NGB, ...: it pools the unit's bag for each of the S samples (the bag is
// summed in index order, like the sequential SparseLengthsSum), applies the unit (weights held in
// registers across the S samples; first layer = in-lane fmaf chains + a G-lane butterfly, so its
// summation order differs from the oracle's k-ordered chain -- the default-mode tolerance,
// tests/test_gpu_parity.py), and keeps a partial Sum.  The partial sums meet in LDS (wave order,
// then group order: fixed, independent of the launch size).
// Schedule: the indices (and bag bounds) of a group's NEXT unit are fetched while the rows of the
// current one are in flight, so a round costs one dependent HBM round trip instead of two; up to
// S x C row loads per lane are in flight (C = rows per bag covered per round: 3 when no bag of the
// launch is longer, else 4); loads past a bag's end read the zero page instead of branching.
// Bytes per sample: T bags of rows + indices in, 4 D floats out.
template <int G, int S, int H, int C, int NW, bool NT>
__global__ __launch_bounds__(64 * NW) void din_fused_kernel(SlsArgs a, const float* __restrict__ packed,
                                                            int64_t stride, const float* __restrict__ zero,
                                                            float* __restrict__ R, int64_t ldr) {
  constexpr int NG = 64 / G, NGB = NW * NG, D = 4 * G;
  // units a lane group has in flight per iteration: with few samples per workgroup it takes several
  // of its units at once (S x UU x C row loads per lane either way), so a small launch -- one
  // query: S = 1 -- needs a quarter of the dependent round trips.  The units are still applied and
  // summed in ascending order: the bits do not depend on S.
  // (Round 4: S = 4 with TWO units in flight -- 24 row loads per lane, 256 VGPRs -- through the generic loop
  // below: 78 us instead of 48 for the 2 048-sample launch; not kept.)
  constexpr int UU = S >= 4 ? 1 : 4 / S;
  static_assert(S <= NW, "wave s finishes sample s");
  __shared__ float4 s_z[S][NW][G];
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / G, gl = lane - g * G, gg = wave * NG + g;
  const int n_smp = a.q.cum[a.q.n_q];
  const int U = a.T - 3;
  const int col = gl * 4;

  // Sample groups are dealt to the XCDs in CONTIGUOUS runs (block b runs on XCD b % 8 -- a speed hint
  // only): neighbouring groups read the same 128-B lines of every table's index row (a line holds the
  // indices of ~10 samples, a group takes S of them), and with the round-robin order each such line
  // was fetched by three XCDs' L2s -- 6.5 % of the launch's HBM traffic (r02 PMC: 222.8 MB against
  // 209.1 MB algorithmic).  A pure renumbering: which workgroup serves which samples changes, no
  // result does.
  const unsigned nb_ = gridDim.x, xcd_ = blockIdx.x & 7u, per_ = nb_ >> 3, rem_ = nb_ & 7u;
  const unsigned grp = xcd_ * per_ + (xcd_ < rem_ ? xcd_ : rem_) + (blockIdx.x >> 3);
  Owner ow[S];
  bool live[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int smp = (int)grp * S + s;
    live[s] = smp < n_smp;
    ow[s] = owner_of(a, live[s] ? smp : 0);
  }
  bool bad = false;

  // a bag per sample of table t: bounds and the first C indices (fetched one unit ahead)
  struct Pre {
    int t;
    const float* W;          // the table's rows, at this lane's columns
    uint32_t rows;
    int beg[S], len[S];
    uint32_t r[S][C];
  };
  bool all_uniform = true;   // (wave-uniform: one scalar branch per unit)
#pragma unroll
  for (int s = 0; s < S; ++s) all_uniform = all_uniform && ow[s].ulen >= 0;
  auto bounds = [&](int t, Pre& p) {
    p.t = t;
    // (the table's base and row count travel with the indices: fetched a unit ahead, not in front
    // of the row loads that need them)
    p.W = a.tables + a.tab_off[t] + col;
    p.rows = (uint32_t)a.tab_rows[t];
    if (all_uniform) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        p.beg[s] = ow[s].b * ow[s].ulen;
        p.len[s] = live[s] ? ow[s].ulen : 0;
      }
    } else {
      // staged prefix sums: all 2 S loads in flight together
      int o0[S], o1[S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int32_t* offp = ow[s].off + (int64_t)t * a.off_stride + ow[s].b;
        o0[s] = offp[0];
        o1[s] = offp[1];
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        // (a fixed-length query coalesced with a ragged one: its prefix sums may never have been
        // uploaded -- the loads above touch mapped memory, their values are not used)
        const bool uni = ow[s].ulen >= 0;
        p.beg[s] = uni ? ow[s].b * ow[s].ulen : o0[s];
        p.len[s] = !live[s] ? 0 : uni ? ow[s].ulen : o1[s] - o0[s];
      }
    }
  };
  auto fetch_idx = [&](const Pre& p, int j0, uint32_t (&r)[S][C]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int32_t* ip = ow[s].idx + (int64_t)p.t * a.idx_stride;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = j0 + c;
        r[s][c] = (uint32_t)ip[j < p.len[s] ? p.beg[s] + j : 0];     // (slot 0 of the table's index block: always mapped)
      }
    }
  };
  auto prefetch = [&](int t, Pre& p) {
    bounds(t, p);
    fetch_idx(p, 0, p.r);
  };
  // rows [j0, j0 + C) of the S bags, added in index order
  const float* zcol = zero + col;
  auto add_rows = [&](const Pre& p, int j0, const uint32_t (&r)[S][C], float4 (&acc)[S]) {
    const float* __restrict__ W = p.W;
    const uint32_t rows = p.rows;
    float4 v[S][C];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const bool in = j0 + c < p.len[s];
        uint32_t rr = r[s][c];
        bad |= in && rr >= rows;
        rr = rr < rows ? rr : 0u;
        v[s][c] = ld4row<NT>(in ? W + ((uint64_t)(rr * ((uint32_t)D >> 2)) << 2) : zcol);
      }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < C; ++c) add4(acc[s], v[s][c]);
  };
  // everything of a bag beyond its first C rows (only bags longer than C: not the shipped config)
  auto add_tail = [&](const Pre& p, float4 (&acc)[S]) {
    int len_max = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) len_max = max(len_max, p.len[s]);
    for (int j0 = C; j0 < len_max; j0 += C) {
      uint32_t r[S][C];
      fetch_idx(p, j0, r);
      add_rows(p, j0, r, acc);
    }
  };
  auto zero_acc = [&](float4 (&acc)[S]) {
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  };

  // candidate ad (every lane group needs it) and this group's first unit: indices in flight together
  Pre pa, pn[UU];
  prefetch(a.T - 2, pa);
#pragma unroll
  for (int uu = 0; uu < UU; ++uu) {
    const int i = gg + uu * NGB;
    prefetch(i < U ? 1 + i : a.T - 2, pn[uu]);
  }
  float4 ad[S];
  zero_acc(ad);
  add_rows(pa, 0, pa.r, ad);
  add_tail(pa, ad);

  // pass-through features of the top MLP's input row: lane group 0 of waves 0..2 takes one each
  // (profile, candidate ad, context)
  if (wave < 3 && g == 0) {
    const int dst = wave == 0 ? 0 : wave == 1 ? 2 * D : 3 * D;
    float4 pv[S];
    if (wave == 1) {
#pragma unroll
      for (int s = 0; s < S; ++s) pv[s] = ad[s];
    } else {
      Pre pp;
      prefetch(wave == 0 ? 0 : a.T - 1, pp);
      zero_acc(pv);
      add_rows(pp, 0, pp.r, pv);
      add_tail(pp, pv);
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (live[s]) *reinterpret_cast<float4*>(R + (int64_t)ow[s].vrow * ldr + dst + col) = pv[s];
  }

  float4 z[S];
  zero_acc(z);
  // (UU == 1 keeps the single-unit loop exactly as it was tuned: the generic form below, with UU = 1,
  // compiles to a schedule that is 13 % slower on the 2 048-sample launch -- 54.5 vs 48 us, same box)
  if constexpr (UU == 1) {
  for (int i = gg; i < U; i += NGB) {
    const Pre p = pn[0];
    float4 u[S];
    zero_acc(u);
    // rows of this unit go out first ...
    const float* __restrict__ W = p.W;
    const uint32_t rows = p.rows;
    float4 v[S][C];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const bool in = c < p.len[s];
        uint32_t rr = p.r[s][c];
        bad |= in && rr >= rows;
        rr = rr < rows ? rr : 0u;
        v[s][c] = ld4row<NT>(in ? W + ((uint64_t)(rr * ((uint32_t)D >> 2)) << 2) : zcol);
      }
    // ... then the next unit's indices and this unit's weights (this lane's pieces, kept across
    // the S samples)
    if (i + NGB < U) prefetch(1 + i + NGB, pn[0]);
    const float* __restrict__ wp = packed + (int64_t)i * stride;
    float4 w1u[H], w1a[H], w1s[H];
    float w2[H][4];
    float b1[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      w1u[hh] = ld4(wp + hh * 3 * D + col);
      w1a[hh] = ld4(wp + hh * 3 * D + D + col);
      w1s[hh] = ld4(wp + hh * 3 * D + 2 * D + col);
      b1[hh] = wp[3 * D * H + D * H + D + hh];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int hh = 0; hh < H; ++hh) w2[hh][j] = wp[3 * D * H + (col + j) * H + hh];
    const float4 b2 = ld4(wp + 3 * D * H + D * H + col);
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < C; ++c) add4(u[s], v[s][c]);
    add_tail(p, u);

#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float4 sum = make_float4(u[s].x + ad[s].x, u[s].y + ad[s].y, u[s].z + ad[s].z, u[s].w + ad[s].w);
      float y[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float pd = dot4(u[s], w1u[hh], 0.f);
        pd = dot4(ad[s], w1a[hh], pd);
        pd = dot4(sum, w1s[hh], pd);
#pragma unroll
        for (int m = 1; m < G; m <<= 1) pd += __shfl_xor(pd, m);
        y[hh] = fmaxf(pd + b1[hh], 0.f);
      }
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        o.x = fmaf(y[hh], w2[hh][0], o.x); o.y = fmaf(y[hh], w2[hh][1], o.y);
        o.z = fmaf(y[hh], w2[hh][2], o.z); o.w = fmaf(y[hh], w2[hh][3], o.w);
      }
      add4(o, b2);
      add4(z[s], relu4(o));
    }
  }
  } else {
  for (int i0 = gg; i0 < U; i0 += UU * NGB) {
    Pre pc[UU];
    float4 v[UU][S][C];
    // rows of this iteration's units go out first ...
#pragma unroll
    for (int uu = 0; uu < UU; ++uu) {
      pc[uu] = pn[uu];
      const Pre& p = pc[uu];
      const bool have = i0 + uu * NGB < U;
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const bool in = have && c < p.len[s];
          uint32_t rr = p.r[s][c];
          bad |= in && rr >= p.rows;
          rr = rr < p.rows ? rr : 0u;
          v[uu][s][c] = ld4row<NT>(in ? p.W + ((uint64_t)(rr * ((uint32_t)D >> 2)) << 2) : zcol);
        }
    }
    // ... then the indices of the next iteration's units ...
#pragma unroll
    for (int uu = 0; uu < UU; ++uu) {
      const int in_ = i0 + (UU + uu) * NGB;
      if (in_ < U) prefetch(1 + in_, pn[uu]);
    }
    // ... and per unit, in ascending order: its weights (this lane's pieces, kept across the S
    // samples), the pooled rows, the unit, the partial Sum
#pragma unroll
    for (int uu = 0; uu < UU; ++uu) {
      const int i = i0 + uu * NGB;
      if (i >= U) break;                         // (uniform within a lane group)
      const Pre& p = pc[uu];
      const float* __restrict__ wp = packed + (int64_t)i * stride;
      float4 w1u[H], w1a[H], w1s[H];
      float w2[H][4];
      float b1[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        w1u[hh] = ld4(wp + hh * 3 * D + col);
        w1a[hh] = ld4(wp + hh * 3 * D + D + col);
        w1s[hh] = ld4(wp + hh * 3 * D + 2 * D + col);
        b1[hh] = wp[3 * D * H + D * H + D + hh];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < H; ++hh) w2[hh][j] = wp[3 * D * H + (col + j) * H + hh];
      const float4 b2 = ld4(wp + 3 * D * H + D * H + col);
      float4 u[S];
      zero_acc(u);
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < C; ++c) add4(u[s], v[uu][s][c]);
      add_tail(p, u);

#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float4 sum = make_float4(u[s].x + ad[s].x, u[s].y + ad[s].y, u[s].z + ad[s].z, u[s].w + ad[s].w);
        float y[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          float pd = dot4(u[s], w1u[hh], 0.f);
          pd = dot4(ad[s], w1a[hh], pd);
          pd = dot4(sum, w1s[hh], pd);
#pragma unroll
          for (int m = 1; m < G; m <<= 1) pd += __shfl_xor(pd, m);
          y[hh] = fmaxf(pd + b1[hh], 0.f);
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          o.x = fmaf(y[hh], w2[hh][0], o.x); o.y = fmaf(y[hh], w2[hh][1], o.y);
          o.z = fmaf(y[hh], w2[hh][2], o.z); o.w = fmaf(y[hh], w2[hh][3], o.w);
        }
        add4(o, b2);
        add4(z[s], relu4(o));
      }
    }
  }
  }
  if (bad) atomicOr(a.err, 1);
  // partial sums: the lane groups of a wave over the cross-lane network, the waves through LDS
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int m = G; m < 64; m <<= 1) add4(z[s], shfl_xor4(z[s], m));
    if (g == 0) s_z[s][wave][gl] = z[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (wave == s && g == 0 && live[s]) {          // wave s finishes sample s (S <= NW)
      float4 t = s_z[s][0][gl];
#pragma unroll
      for (int w = 1; w < NW; ++w) add4(t, s_z[s][w][gl]);
      *reinterpret_cast<float4*>(R + (int64_t)ow[s].vrow * ldr + D + col) = t;
    }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}

// Sum over the G lanes of a lane group, every lane ends with the total: the butterfly of __shfl_xor(., 1 / 2 / 4 / 8) --
// same pairs, same bits -- on the DPP network instead of ds_bpermute (xor 1 and 2 as quad permutes; after them the four
// lanes of a quad agree, so the mirror of 8 (16) lanes delivers the partner quad's (half-row's) sum).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 8 || G == 16, "lane groups of 8 or 16");
  v += dpp_f<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);                    // row_half_mirror
  if (G == 16) v += dpp_f<0x140>(v);       // row_mirror
  return v;
}

// PIPELINED form of the fused launch: hidden width 1 (the shipped attention unit 96-1-32) and launch sets whose
// bags all have one fixed length <= 3 (FusedShape::C == 3; din.json: 3 lookups).
// din_fused_kernel above walks a lane group's ~U / NGB units one dependent round trip after the other
// (indices -> rows -> unit), and the launch is a single wave of workgroups: its time is the length of
// that chain, not the bytes (profiles/r05_din/).  Here the workgroup first stages what the chain would
// fetch on the way -- the indices of every (table, sample) bag and the tables' bases -- in LDS with ONE
// round trip (thread t takes table t), after which a row load depends on nothing in HBM.  Each lane
// group then keeps P units in flight: slot j holds the row pieces AND the weights of one unit; a slot is
// refilled with the unit P places on as soon as its unit is applied.  Same work split, same summation
// order as din_fused_kernel: the two forms give the same bits (tests/test_gpu_parity.py).
template <int G, int S, int NW, int P, bool NT>
__global__ __launch_bounds__(64 * NW, 2) void din_pipe_kernel(SlsArgs a, const float* __restrict__ packed,
                                                              int64_t stride, const float* __restrict__ zero,
                                                              float* __restrict__ R, int64_t ldr) {
  constexpr int NG = 64 / G, NGB = NW * NG, D = 4 * G, C = 3;
  constexpr uint32_t kNone = 0xffffffffu;            // staged "no row here": the load reads the zero page
  static_assert(S <= NW, "wave s finishes sample s");
  __shared__ float4 s_z[S][NW][G];
  extern __shared__ int64_t s_dyn[];
  const int T = a.T;
  int64_t* s_off = s_dyn;                                         // [T] element offset of the table
  uint32_t* s_r = reinterpret_cast<uint32_t*>(s_off + T);         // [T][S * C] row numbers (kNone past the bag's end)
  if (a.ts && threadIdx.x == 0) a.ts[2 * blockIdx.x] = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / G, gl = lane - g * G, gg = wave * NG + g;
  const int n_smp = a.q.cum[a.q.n_q];
  const int U = T - 3;
  const int col = gl * 4;

  // (sample groups dealt to the XCDs in contiguous runs: din_fused_kernel)
  const unsigned nb_ = gridDim.x, xcd_ = blockIdx.x & 7u, per_ = nb_ >> 3, rem_ = nb_ & 7u;
  const unsigned grp = xcd_ * per_ + (xcd_ < rem_ ? xcd_ : rem_) + (blockIdx.x >> 3);
  bool live[S];
  int vrow[S];
  {
    // ---- stage: thread t takes table t (the owners of the S samples are only needed here) ----------
    Owner ow[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int smp = (int)grp * S + s;
      live[s] = smp < n_smp;
      ow[s] = owner_of(a, live[s] ? smp : 0);
      vrow[s] = ow[s].vrow;
    }
    bool bad = false;
    for (int t = threadIdx.x; t < T; t += 64 * NW) {
      const uint32_t rows = (uint32_t)a.tab_rows[t];
      s_off[t] = a.tab_off[t];
      uint32_t r[S][C];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int len = live[s] ? ow[s].ulen : 0;
        const int32_t* ip = ow[s].idx + (int64_t)t * a.idx_stride;
        if (len == C) {                        // (uniform per sample) the whole bag with one 12-byte load
          typedef int32_t I3 __attribute__((ext_vector_type(3), aligned(4)));
          const I3 i3 = *reinterpret_cast<const I3*>(ip + ow[s].b * C);
          r[s][0] = (uint32_t)i3.x; r[s][1] = (uint32_t)i3.y; r[s][2] = (uint32_t)i3.z;
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) r[s][c] = (uint32_t)ip[c < len ? ow[s].b * ow[s].ulen + c : 0];
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int len = live[s] ? ow[s].ulen : 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const bool in = c < len;
          bad |= in && r[s][c] >= rows;
          s_r[t * (S * C) + s * C + c] = !in ? kNone : r[s][c] < rows ? r[s][c] : 0u;
        }
      }
    }
    if (bad) atomicOr(a.err, 1);
  }
  __syncthreads();

  const float* zcol = zero + col;
  // the staged rows of the S bags of table t -> v
  auto issue_rows = [&](int t, bool have, float4 (&v)[S][C]) {
    const float* __restrict__ W = a.tables + s_off[t] + col;
    uint32_t r[S * C];
#pragma unroll
    for (int x = 0; x < S * C; ++x) r[x] = s_r[t * (S * C) + x];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const uint32_t rr = r[s * C + c];
        v[s][c] = ld4row<NT>(have && rr != kNone ? W + ((uint64_t)(rr * ((uint32_t)D >> 2)) << 2) : zcol);
      }
  };
  auto pool = [&](const float4 (&v)[S][C], float4 (&acc)[S]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < C; ++c) add4(acc[s], v[s][c]);
    }
  };
  // one unit's weights, this lane's pieces: [ W1 : 3D | W2 : D | b2 : D | b1 ] (h = 1)
  struct Wt {
    float4 w1u, w1a, w1s, w2, b2;
    float b1;
  };
  auto issue_w = [&](int i, Wt& w) {
    const float* __restrict__ wp = packed + (int64_t)i * stride;
    w.w1u = ld4(wp + col);
    w.w1a = ld4(wp + D + col);
    w.w1s = ld4(wp + 2 * D + col);
    w.w2 = ld4(wp + 3 * D + col);
    w.b2 = ld4(wp + 4 * D + col);
    w.b1 = wp[5 * D];
  };

  // slots: the candidate ad goes out together with the group's first P units
  const int K = (U + NGB - 1) / NGB;        // units per lane group (the last one may be missing: `have`)
  float4 v[P][S][C];
  Wt w[P];
  float4 ad[S];
  {
    float4 vad[S][C];
    issue_rows(T - 2, true, vad);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int i = gg + j * NGB;
      const bool have = j < K && i < U;
      issue_rows(have ? 1 + i : T - 2, have, v[j]);
      issue_w(have ? i : 0, w[j]);
    }
    pool(vad, ad);
  }

  // pass-through features of the top MLP's input row: lane group 0 of waves 0..2 takes one each
  // (profile, candidate ad, context)
  if (wave < 3 && g == 0) {
    const int dst = wave == 0 ? 0 : wave == 1 ? 2 * D : 3 * D;
    float4 pv[S];
    if (wave == 1) {
#pragma unroll
      for (int s = 0; s < S; ++s) pv[s] = ad[s];
    } else {
      float4 vp[S][C];
      issue_rows(wave == 0 ? 0 : T - 1, true, vp);
      pool(vp, pv);
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (live[s]) *reinterpret_cast<float4*>(R + (int64_t)vrow[s] * ldr + dst + col) = pv[s];
  }

  float4 z[S];
#pragma unroll
  for (int s = 0; s < S; ++s) z[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  // unit k of this lane group out of slot j; then the slot takes unit k + P (into the SAME registers, after the
  // unit's arithmetic: refilled before it, the compiler lands the loads in fresh registers and copies them over
  // at the end of the iteration, which waits for every load in flight)
  auto step = [&](int j, int k, bool refill) {
    // (the steps stay in program order: the scheduler otherwise pools the next slot's rows ahead of this slot's
    // refill -- every load consumed, then every load reissued, nothing in flight in between)
    __builtin_amdgcn_sched_barrier(0);
    const bool have = gg + k * NGB < U;
    float4 u[S];
    pool(v[j], u);
    const Wt& wt = w[j];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float4 sum = make_float4(u[s].x + ad[s].x, u[s].y + ad[s].y, u[s].z + ad[s].z, u[s].w + ad[s].w);
      float pd = dot4(u[s], wt.w1u, 0.f);
      pd = dot4(ad[s], wt.w1a, pd);
      pd = dot4(sum, wt.w1s, pd);
      pd = group_sum<G>(pd);
      const float y = fmaxf(pd + wt.b1, 0.f);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      o.x = fmaf(y, wt.w2.x, o.x); o.y = fmaf(y, wt.w2.y, o.y);
      o.z = fmaf(y, wt.w2.z, o.z); o.w = fmaf(y, wt.w2.w, o.w);
      add4(o, wt.b2);
      o = relu4(o);
      z[s].x += have ? o.x : 0.f; z[s].y += have ? o.y : 0.f;
      z[s].z += have ? o.z : 0.f; z[s].w += have ? o.w : 0.f;
    }
    if (refill) {
      const int in_ = gg + (k + P) * NGB;
      const bool hn = in_ < U;
      issue_rows(hn ? 1 + in_ : T - 2, hn, v[j]);
      issue_w(hn ? in_ : 0, w[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // rounds in which every slot is refilled (no branch around the loads: with one, the wait counts after it must
  // assume the loads were skipped, and wait for everything) ...
  int k0 = 0;
  for (; k0 + 2 * P <= K; k0 += P) {
#pragma unroll
    for (int j = 0; j < P; ++j) step(j, k0 + j, true);
  }
  // ... and the last ones
  for (; k0 < K; k0 += P) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int k = k0 + j;
      if (k >= K) break;                       // (uniform)
      step(j, k, k + P < K);
    }
  }
  // partial sums: the lane groups of a wave over the cross-lane network, the waves through LDS
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int m = G; m < 64; m <<= 1) add4(z[s], shfl_xor4(z[s], m));
    if (g == 0) s_z[s][wave][gl] = z[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (wave == s && g == 0 && live[s]) {          // wave s finishes sample s (S <= NW)
      float4 t = s_z[s][0][gl];
#pragma unroll
      for (int w_ = 1; w_ < NW; ++w_) add4(t, s_z[s][w_][gl]);
      *reinterpret_cast<float4*>(R + (int64_t)vrow[s] * ldr + D + col) = t;
    }
  if (a.ts) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) a.ts[2 * blockIdx.x + 1] = wall_clock64();
  }
}

struct FusedShape { int S, C; };
// Waves per workgroup.  FIXED for every launch size: it decides which units a lane group sums, i.e.
// the association of the fp32 Sum over the units -- a query's bits must not depend on how many
// queries were coalesced with it.  Measured on the din.json shape (254 tables, 3 lookups, D 32),
// 2048 samples per launch: 4 waves x 4 samples 49 us, 8 x 2 54 us, 8 x 4 62 us, 4 x 2 52 us,
// 4 x 1 63 us; one query (256 samples): 4 x 1 16 us, 8 x 1 12 us.
constexpr int kWaves = 4;

template <int G, int S, int H, int C, int NW>
void launch_fused_k(const SlsArgs& a, const float* packed, int64_t stride, const float* zero, float* R, int64_t ldr,
                    unsigned grid, hipStream_t s, hipEvent_t stop) {
  if (a.nt) {
    if (stop) hipExtLaunchKernelGGL((din_fused_kernel<G, S, H, C, NW, true>), dim3(grid), dim3(64 * NW), 0, s, nullptr, stop, 0, a, packed, stride, zero, R, ldr);
    else hipLaunchKernelGGL((din_fused_kernel<G, S, H, C, NW, true>), dim3(grid), dim3(64 * NW), 0, s, a, packed, stride, zero, R, ldr);
  } else {
    if (stop) hipExtLaunchKernelGGL((din_fused_kernel<G, S, H, C, NW, false>), dim3(grid), dim3(64 * NW), 0, s, nullptr, stop, 0, a, packed, stride, zero, R, ldr);
    else hipLaunchKernelGGL((din_fused_kernel<G, S, H, C, NW, false>), dim3(grid), dim3(64 * NW), 0, s, a, packed, stride, zero, R, ldr);
  }
}
template <int G, int S, int H>
void launch_fused_c(const FusedShape& f, const SlsArgs& a, const float* packed, int64_t stride, const float* zero,
                    float* R, int64_t ldr, unsigned grid, hipStream_t s, hipEvent_t stop) {
  if (f.C == 3) launch_fused_k<G, S, H, 3, kWaves>(a, packed, stride, zero, R, ldr, grid, s, stop);
  else launch_fused_k<G, S, H, 4, kWaves>(a, packed, stride, zero, R, ldr, grid, s, stop);
}
template <int G, int S>
void launch_fused_h(const FusedShape& f, const SlsArgs& a, int h, const float* packed, int64_t stride, const float* zero,
                    float* R, int64_t ldr, unsigned grid, hipStream_t s, hipEvent_t stop) {
  if (h == 1) launch_fused_c<G, S, 1>(f, a, packed, stride, zero, R, ldr, grid, s, stop);
  else if (h == 2) launch_fused_c<G, S, 2>(f, a, packed, stride, zero, R, ldr, grid, s, stop);
  else launch_fused_c<G, S, 4>(f, a, packed, stride, zero, R, ldr, grid, s, stop);
}
template <int G>
void launch_fused_s(const FusedShape& f, const SlsArgs& a, int h, const float* packed, int64_t stride, const float* zero,
                    float* R, int64_t ldr, unsigned grid, hipStream_t s, hipEvent_t stop) {
  if (f.S == 4) launch_fused_h<G, 4>(f, a, h, packed, stride, zero, R, ldr, grid, s, stop);
  else if (f.S == 2) launch_fused_h<G, 2>(f, a, h, packed, stride, zero, R, ldr, grid, s, stop);
  else launch_fused_h<G, 1>(f, a, h, packed, stride, zero, R, ldr, grid, s, stop);
}

// The pipelined form keeps kPipe units in flight per lane group.  Measured on din.json (12-query sets; one query):
// S = 4: depth 2 -> 40.5 us alone; S = 2: depth 2 / 3 / 4 -> 46.5 / 45.8 / 46.5 us alone, 147.8 / 138.9 / 138.3 k queries/s
// beside the MLP launches; S = 1 (one query): 2 / 4 / 6 -> 10.15 / 10.18 / 12.1 us.  The main phase already runs at
// ~0.77 of the HBM peak: deeper does not help, more registers hurt (profiles/r05_din/).
constexpr int kPipe = 2;
size_t din_pipe_lds(int T, int S) { return (size_t)T * (8 + (size_t)S * 3 * 4); }
template <int G, int S>
void launch_pipe_k(const SlsArgs& a, const float* packed, int64_t stride, const float* zero, float* R, int64_t ldr,
                   unsigned grid, hipStream_t s, hipEvent_t stop) {
  const size_t lds = din_pipe_lds(a.T, S);
  if (a.nt) {
    if (stop) hipExtLaunchKernelGGL((din_pipe_kernel<G, S, kWaves, kPipe, true>), dim3(grid), dim3(64 * kWaves), lds, s, nullptr, stop, 0, a, packed, stride, zero, R, ldr);
    else hipLaunchKernelGGL((din_pipe_kernel<G, S, kWaves, kPipe, true>), dim3(grid), dim3(64 * kWaves), lds, s, a, packed, stride, zero, R, ldr);
  } else {
    if (stop) hipExtLaunchKernelGGL((din_pipe_kernel<G, S, kWaves, kPipe, false>), dim3(grid), dim3(64 * kWaves), lds, s, nullptr, stop, 0, a, packed, stride, zero, R, ldr);
    else hipLaunchKernelGGL((din_pipe_kernel<G, S, kWaves, kPipe, false>), dim3(grid), dim3(64 * kWaves), lds, s, a, packed, stride, zero, R, ldr);
  }
}
template <int G>
void launch_pipe_s(const FusedShape& f, const SlsArgs& a, const float* packed, int64_t stride, const float* zero,
                   float* R, int64_t ldr, unsigned grid, hipStream_t s, hipEvent_t stop) {
  if (f.S == 4) launch_pipe_k<G, 4>(a, packed, stride, zero, R, ldr, grid, s, stop);
  else if (f.S == 2) launch_pipe_k<G, 2>(a, packed, stride, zero, R, ldr, grid, s, stop);
  else launch_pipe_k<G, 1>(a, packed, stride, zero, R, ldr, grid, s, stop);
}

// Samples per workgroup (the units' weights are read once per workgroup; fewer for small launches
// so that a single query still fills the chip) and rows per round of a launch.
FusedShape fused_shape(const SlsArgs& a, const Tune& tune) {
  FusedShape f;
  const int64_t n_smp = a.q.cum[a.q.n_q];
  f.S = tune.din_s > 0 ? tune.din_s : (n_smp >= 1024 ? 4 : n_smp >= 512 ? 2 : 1);
  f.C = 3;
  for (int i = 0; i < a.q.n_q; ++i)
    if (a.uniform_len[i] < 0 || a.uniform_len[i] > 3) f.C = 4;
  return f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// DIEN (models/dien.py:308-432): two caffe2 rnn_cell.BasicRNN layers over the behaviour
// embeddings of a query.  One wave per sample, lane j = hidden unit j (H <= 64): the lane keeps
// row j of the four weight matrices in registers for all U steps (D + 3 H values), the step's
// input and the two states are broadcast through 3 x 64 floats of LDS per wave (ds_read_b128 of one
// address for all lanes; no VALU cross-lane traffic), every product-sum is the oracle's k-ordered
// fmaf chain with the bias added after it, and the next step's input row is fetched while the
// current step computes.  VALU work: the recurrence is U = 40 dependent steps of 64-wide
// mat-vecs per sample -- an MFMA form would tile 16 samples x 16 hidden units per wave and
// exchange states through LDS with a barrier per step; at 1.1 MFLOP per sample the VALU form
// already runs a launch set in tens of microseconds, next to ~10 us of gather.
// Packed weights (dien_pack_kernel), transposed so that lane j's loads coalesce:
//   [ i2h_0^T : D x H | gates_0^T : H x H | i2h_1^T : H x H | gates_1^T : H x H | 4 biases : 4 x H ]
namespace {

__global__ __launch_bounds__(256) void dien_pack_kernel(const float* const* __restrict__ w, float* __restrict__ packed,
                                                        int D, int H) {
  // w: {i2h_w, i2h_b, gates_w, gates_b} of layer 1, then of layer 2 (row-major [out, in])
  const int K[4] = {D, H, H, H};
  const int src[4] = {0, 2, 4, 6};
  int64_t off = 0;
  for (int m = 0; m < 4; ++m) {
    const float* W = w[src[m]];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)K[m] * H; i += (int64_t)gridDim.x * blockDim.x) {
      const int k = (int)(i / H), j = (int)(i - (int64_t)k * H);
      packed[off + i] = W[(int64_t)j * K[m] + k];
    }
    off += (int64_t)K[m] * H;
  }
  const int bsrc[4] = {1, 3, 5, 7};
  for (int m = 0; m < 4; ++m)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < H; j += gridDim.x * blockDim.x) packed[off + (int64_t)m * H + j] = w[bsrc[m]][j];
}

template <int K>
__device__ __forceinline__ float chain_lds(const float* __restrict__ sv, const float (&w)[K]) {
  float acc = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < K / 4; ++k4) {
    const float4 v = *reinterpret_cast<const float4*>(sv + 4 * k4);
    acc = fmaf(v.x, w[4 * k4 + 0], acc); acc = fmaf(v.y, w[4 * k4 + 1], acc);
    acc = fmaf(v.z, w[4 * k4 + 2], acc); acc = fmaf(v.w, w[4 * k4 + 3], acc);
  }
  return acc;
}

template <int D, int H>
__global__ __launch_bounds__(256) void dien_rnn_kernel(const float* __restrict__ T, int64_t ldt, QTable q, int Tn,
                                                       const float* __restrict__ packed, float* __restrict__ R,
                                                       int64_t ldr) {
  static_assert(D % 4 == 0 && H % 4 == 0 && D <= 64 && H <= 64, "lane j = hidden unit j");
  __shared__ __attribute__((aligned(16))) float sx[4][64], sh0[4][64], sh1[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int smp = blockIdx.x * 4 + wave;
  if (smp >= q.cum[q.n_q]) return;                       // (no workgroup barrier below)
  int b = smp, bs = q.bs[0], v0 = q.vstart[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < q.n_q && smp >= q.cum[i];
    b = in ? smp - q.cum[i] : b;
    bs = in ? q.bs[i] : bs;
    v0 = in ? q.vstart[i] : v0;
  }
  if (q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < q.n_q && smp >= q.cum[i];
      b = in ? smp - q.cum[i] : b;
      bs = in ? q.bs[i] : bs;
      v0 = in ? q.vstart[i] : v0;
    }
  }
  const int U = Tn - 3;
  const int j = min(lane, H - 1);
  float wi0[D], wg0[H], wi1[H], wg1[H];
  const float* p = packed;
#pragma unroll
  for (int k = 0; k < D; ++k) wi0[k] = p[k * H + j];
  p += D * H;
#pragma unroll
  for (int k = 0; k < H; ++k) wg0[k] = p[k * H + j];
  p += H * H;
#pragma unroll
  for (int k = 0; k < H; ++k) wi1[k] = p[k * H + j];
  p += H * H;
#pragma unroll
  for (int k = 0; k < H; ++k) wg1[k] = p[k * H + j];
  p += H * H;
  const float bi0 = p[j], bg0 = p[H + j], bi1 = p[2 * H + j], bg1 = p[3 * H + j];

  // step t of "sample" b reads embedding n % U of sample n / U, n = t * bs + b (the reference's
  // Reshape of [bs, U*D] to [U, bs, D], models/dien.py:316-320)
  auto x_of = [&](int t) {
    const int n = t * bs + b;
    const int src = n / U, unit = n - src * U;
    return T[(int64_t)(v0 + src) * ldt + (int64_t)(1 + unit) * D + min(lane, D - 1)];
  };
  float* mx = sx[wave];
  float* m0 = sh0[wave];
  float* m1 = sh1[wave];
  m0[lane] = 0.f;                                        // initial_h = 0 (:498-499)
  m1[lane] = 0.f;
  float xv = x_of(0), h1 = 0.f;
  // every weight load has landed before the loop: inside it only the input prefetch is in flight,
  // and nothing waits for it before the next step's LDS write (with the weights still pending at
  // the loop header the waitcnt pass put a vmcnt(0) right behind the prefetch: 3 800 cycles a step)
  __builtin_amdgcn_s_waitcnt(0);
  for (int t = 0; t < U; ++t) {
    mx[lane] = xv;
    __builtin_amdgcn_wave_barrier();
    xv = x_of(min(t + 1, U - 1));                       // (unconditional: one more read of the last row)
    // layer 1: Tanh(Sum(FC(h_prev, gates_t), FC(x_t, i2h)))
    const float a0 = chain_lds<D>(mx, wi0) + bi0;
    const float g0 = chain_lds<H>(m0, wg0) + bg0;
    const float h0 = tanh_rnn(g0 + a0);
    __builtin_amdgcn_wave_barrier();
    m0[lane] = h0;
    __builtin_amdgcn_wave_barrier();
    // layer 2 on layer 1's new state
    const float a1 = chain_lds<H>(m0, wi1) + bi1;
    const float g1 = chain_lds<H>(m1, wg1) + bg1;
    h1 = tanh_rnn(g1 + a1);
    __builtin_amdgcn_wave_barrier();
    m1[lane] = h1;
    __builtin_amdgcn_wave_barrier();
  }
  // top MLP input row: [ last state | user profile | candidate ad | context ] (:411-421)
  float* out = R + (int64_t)(v0 + b) * ldr;
  const float* e = T + (int64_t)(v0 + b) * ldt;
  if (lane < H) out[lane] = h1;
  if (lane < D) {
    out[H + lane] = e[lane];
    out[H + D + lane] = e[(int64_t)(Tn - 2) * D + lane];
    out[H + 2 * D + lane] = e[(int64_t)(Tn - 1) * D + lane];
  }
}

// The MFMA form (H a multiple of 16).  A workgroup of H / 16 waves serves 16 samples; wave w owns
// hidden units [16 w, 16 w + 16) of BOTH layers.  Per step and layer the pre-activations are two
// v_mfma_f32_16x16x4_f32 chains (bit-for-bit k-ordered fp32 fma chains, as in mlp.hip): A = the
// wave's 16 weight rows (registers for the whole launch: (D + 3 H) / 4 VGPRs), B = the step's input
// [k][sample] -- the embeddings through LDS (one coalesced fetch of the 16 rows per workgroup and step,
// four steps ahead: round 5), the states from LDS ([hidden][sample], double-buffered so ONE workgroup
// barrier per step orders everything) -- D[m = hidden 4 g + q][n = sample r].  Same bits as dien_rnn_kernel.
// Cost: (D + 3 H) / 4 = 56 MFMAs of 32 cycles per wave and step at D 32 / H 64, 16 samples at a
// time, against 2 x 112 dependent VALU fmas per SAMPLE in the one-wave-per-sample form.
// Measured on dien.json's shape (40 steps, 2048 samples per launch = 128 workgroups): every wave
// running both layers 76 us per launch (the VALU form: 112 us); per step 1.9 us = 0.8 MFMA + 0.2 tanh
// + 0.2 barrier + 0.7 LDS / issue latency that one wave per SIMD cannot hide.  One wave set per
// layer (SPLIT, the default: 24 / 32 MFMAs per wave and step, two waves per SIMD): 66 us.  The engine
// also lets the launches of consecutive sets overlap on separate streams (each covers half the
// chip): 51 k -> 127 k (both layers per wave) -> 141 k queries/s (SPLIT).
typedef float f32x4_ __attribute__((ext_vector_type(4)));
struct DienW { const float* w[8]; };   // {i2h_w, i2h_b, gates_t_w, gates_t_b} x 2 layers, row-major [out, in]

template <int D, int H, int SPLIT>
__global__ __launch_bounds__(64 * (SPLIT ? 2 : 1) * (H / 16)) void dien_rnn_mfma_kernel(const float* __restrict__ T, int64_t ldt,
                                                                                     QTable q, int Tn, DienW W,
                                                                                     float* __restrict__ R, int64_t ldr,
                                                                                     DienTop top, Done done) {
  static_assert(D % 4 == 0 && H % 16 == 0 && H <= 64, "16 hidden units per wave");
  // SPLIT = 1: 2 x H / 16 waves; the first H / 16 run layer 1 (of step t + 1), the others layer 2
  // (of step t) -- the two layers of an iteration are independent, so the per-step critical path of
  // a wave is 24 or 32 MFMAs instead of 56, at two waves per SIMD.  SPLIT = 0: every wave runs both
  // layers of its 16 hidden units (four interleaved chains).  Same bits.
  // (Two independent 16-sample groups per workgroup, the other way to put two waves on a SIMD, was
  // measured as well: 130 us on half as many CUs instead of 76 us, no gain in queries/s.)
  constexpr int NW = H / 16, NT = 64 * (SPLIT ? 2 : 1) * NW;
  __shared__ float s0[2][H][16], s1[2][H][16];
  // x_t of the 16 samples, [sample][k] with rows 4 floats apart from a multiple of 64: the B operand read
  // sx[.][r][4 s + g] touches 64 different banks, the loaders' 16-byte writes are aligned
  constexpr int XLD = D + 4;
  __shared__ __attribute__((aligned(16))) float sx[2][16][XLD];
  extern __shared__ __attribute__((aligned(16))) float dien_top_lds[];   // fused top MLP: 2 x [kmax][16] (none otherwise)
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % NW, role = (threadIdx.x >> 6) / NW;
  const bool do1 = !SPLIT || role == 0, do2 = !SPLIT || role == 1;   // (wave-uniform)
  const int r = lane & 15, g = lane >> 4;
  const int n_smp = q.cum[q.n_q];
  const int smp_base = (int)blockIdx.x * 16;
  const int smp = min(smp_base + r, n_smp - 1);
  const bool live = smp_base + r < n_smp;
  int b = smp, bs = q.bs[0], v0 = q.vstart[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const bool in = i < q.n_q && smp >= q.cum[i];
    b = in ? smp - q.cum[i] : b;
    bs = in ? q.bs[i] : bs;
    v0 = in ? q.vstart[i] : v0;
  }
  if (q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
    for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
      const bool in = i < q.n_q && smp >= q.cum[i];
      b = in ? smp - q.cum[i] : b;
      bs = in ? q.bs[i] : bs;
      v0 = in ? q.vstart[i] : v0;
    }
  }
  const int U = Tn - 3;
  // A operands: lane (r, g) holds W[16 w + r][4 s + g] of every MFMA step s -- the i2h and gates_t
  // rows of the layer(s) this wave runs
  const int row = 16 * wave + r;
  float wia[SPLIT ? (D > H ? D : H) / 4 : D / 4], wga[H / 4];      // layer 1 (or, SPLIT role 1, layer 2)
  float wib[SPLIT ? 1 : H / 4], wgb[SPLIT ? 1 : H / 4];            // layer 2 when one wave runs both
  float bia[4], bga[4], bib[4], bgb[4];
  const int la = (SPLIT && role == 1) ? 1 : 0;                       // layer held in the "a" set
  const int Ka = la == 0 ? D : H;
#pragma unroll
  for (int s = 0; s < (int)(sizeof(wia) / sizeof(float)); ++s) wia[s] = 4 * s < Ka ? W.w[4 * la + 0][row * Ka + 4 * s + g] : 0.f;
#pragma unroll
  for (int s = 0; s < H / 4; ++s) wga[s] = W.w[4 * la + 2][row * H + 4 * s + g];
  if (!SPLIT) {
#pragma unroll
    for (int s = 0; s < H / 4; ++s) {
      wib[s] = W.w[4][row * H + 4 * s + g];
      wgb[s] = W.w[6][row * H + 4 * s + g];
    }
  }
  // biases of the 4 output rows this lane holds: hidden 16 w + 4 g + qd
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int hid = 16 * wave + 4 * g + qd;
    bia[qd] = W.w[4 * la + 1][hid]; bga[qd] = W.w[4 * la + 3][hid];
    bib[qd] = SPLIT ? 0.f : W.w[5][hid]; bgb[qd] = SPLIT ? 0.f : W.w[7][hid];
  }
  for (int i = threadIdx.x; i < 2 * H * 16; i += NT) {         // initial_h = 0 (models/dien.py:498-499)
    (&s0[0][0][0])[i] = 0.f;
    (&s1[0][0][0])[i] = 0.f;
  }
  // B operand of the input product: x_t[sample r][k = 4 s + g]; step t of "sample" b is embedding
  // n % U of sample n / U, n = t * bs + b (the reference's Reshape, models/dien.py:316-320).
  // Round 5: the 16 rows of a step (D floats each, contiguous in the gather's buffer) are fetched ONCE per
  // workgroup, 16 bytes per lane (loader lane f takes piece f % (D/4) of sample f / (D/4): a row per 128-byte
  // line), NB steps ahead into a register ring, and handed to the layer-1 waves through sx -- until then every
  // layer-1 wave fetched the operand itself, a dword per lane and MFMA step: 8 requests of 16 lines each per wave
  // and step, 32 per workgroup, and the texture path of a CU with two workgroups was the bound (the recurrence
  // with the fetch compiled out: 179 k -> 203 k queries/s; with a second workgroup on the CU a launch took twice
  // as long).  (Rounds 3-4 on the wait counters across the loop's back edge: docs/DESIGN_rounds_1-4.md.)
  constexpr int NB = 4;
  constexpr int NF = 16 * D / 4;                                  // float4 pieces of a step
  constexpr int NLD = 64 * NW;                                    // loader lanes: the waves that run layer 1
  constexpr int NL = (NF + NLD - 1) / NLD;
  f32x4_ xr[NB][NL];
  const float* xsrc[NL];                                          // this loader lane's sample: row 0 of its query ...
  int xb_[NL], xbs[NL];                                           // ... its number in the query, the query's size
  if (do1) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int f = min(wave * 64 + lane + i * NLD, NF - 1);
      const int ls = f / (D / 4), piece = f - ls * (D / 4);
      const int sm = min(smp_base + ls, n_smp - 1);
      int bb = sm, bsz = q.bs[0], vv = q.vstart[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const bool in = k < q.n_q && sm >= q.cum[k];
        bb = in ? sm - q.cum[k] : bb;
        bsz = in ? q.bs[k] : bsz;
        vv = in ? q.vstart[k] : vv;
      }
      if (q.n_q > 8) {
#pragma unroll
        for (int k = 8; k < DRS_MAX_COALESCE; ++k) {
          const bool in = k < q.n_q && sm >= q.cum[k];
          bb = in ? sm - q.cum[k] : bb;
          bsz = in ? q.bs[k] : bsz;
          vv = in ? q.vstart[k] : vv;
        }
      }
      xb_[i] = bb; xbs[i] = bsz;
      xsrc[i] = T + (int64_t)vv * ldt + D + 4 * piece;
    }
  }
  auto fetch_x = [&](int t, f32x4_ (&xq)[NL]) {
    if (!do1) return;
    const int tt = min(t, U - 1);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int n = tt * xbs[i] + xb_[i];
      const int src = n / U, unit = n - src * U;
      xq[i] = *reinterpret_cast<const f32x4_*>(xsrc[i] + (int64_t)src * ldt + (int64_t)unit * D);
    }
  };
  auto stash_x = [&](int buf, const f32x4_ (&xq)[NL]) {            // ring slot -> sx[buf]
    if (!do1) return;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int f = wave * 64 + lane + i * NLD;
      if (f < NF) {
        const int ls = f / (D / 4), piece = f - ls * (D / 4);
        *reinterpret_cast<f32x4_*>(&sx[buf][ls][4 * piece]) = xq[i];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < NB; ++j) fetch_x(j, xr[j]);              // slot j % NB holds x_j
  stash_x(0, xr[0]);
  stash_x(1, xr[1 % NB]);
  fetch_x(NB, xr[0]);
  fetch_x(NB + 1, xr[1 % NB]);
  __syncthreads();
  // Layer 2 runs one step behind layer 1: an iteration holds layer 2 of step t and layer 1 of step
  // t + 1, which do not depend on each other, and ONE barrier per iteration orders the
  // double-buffered state exchange.
  //   s0[t & 1] = layer-1 state after step t,  s1[t & 1] = layer-2 state after step t
  auto layer1 = [&](int rd, int wr, int xbuf) {   // x_t in sx[xbuf], state s0[rd] -> s0[wr]
    f32x4_ aa = {0.f, 0.f, 0.f, 0.f}, ag = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < H / 4; ++s) {
      ag = __builtin_amdgcn_mfma_f32_16x16x4f32(wga[s], s0[rd][4 * s + g][r], ag, 0, 0, 0);
      if (s < D / 4) aa = __builtin_amdgcn_mfma_f32_16x16x4f32(wia[s], sx[xbuf][r][4 * s + g], aa, 0, 0, 0);
    }
#pragma unroll
    for (int s = H / 4; s < D / 4; ++s) aa = __builtin_amdgcn_mfma_f32_16x16x4f32(wia[s], sx[xbuf][r][4 * s + g], aa, 0, 0, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      s0[wr][16 * wave + 4 * g + qd][r] = tanh_rnn((ag[qd] + bga[qd]) + (aa[qd] + bia[qd]));
  };
  f32x4_ h1v = {0.f, 0.f, 0.f, 0.f};
  if (do1) layer1(1, 0, 0);         // step 0 of layer 1: its previous state is the zero buffer s0[1]
  __syncthreads();
  for (int t0 = 0; t0 < U; t0 += NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = t0 + j;
      if (t >= U) break;                                 // (uniform)
      const int cur = t & 1, prv = cur ^ 1;
      f32x4_ (&xn)[NL] = xr[(j + 2) % NB];               // x_{t+2} (t0 is a multiple of NB): into sx[t & 1], whose x_t
                                                         // was read an iteration ago; x_{t+1} sits in sx[prv]
      // layer 2, step t: input = layer-1 state of step t (s0[cur]), previous own state s1[prv];
      // layer 1, step t + 1: input x_{t+1}, previous state s0[cur]; writes s0[prv].  (After the last
      // step layer 1 computes one step too many into the unused buffer: cheaper than a divergent tail.)
      if (SPLIT) {
        if (role == 0) {
          layer1(cur, prv, prv);
          stash_x(cur, xn);
          fetch_x(t + 2 + NB, xn);
        } else {
          f32x4_ ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < H / 4; ++s) {
            ba = __builtin_amdgcn_mfma_f32_16x16x4f32(wia[s], s0[cur][4 * s + g][r], ba, 0, 0, 0);
            bg = __builtin_amdgcn_mfma_f32_16x16x4f32(wga[s], s1[prv][4 * s + g][r], bg, 0, 0, 0);
          }
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            h1v[qd] = tanh_rnn((bg[qd] + bga[qd]) + (ba[qd] + bia[qd]));
            s1[cur][16 * wave + 4 * g + qd][r] = h1v[qd];
          }
        }
      } else {
        // one wave, four chains issued round-robin (independent MFMAs back to back)
        f32x4_ ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
        f32x4_ aa = {0.f, 0.f, 0.f, 0.f}, ag = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < H / 4; ++s) {
          const float h0k = s0[cur][4 * s + g][r];
          ba = __builtin_amdgcn_mfma_f32_16x16x4f32(wib[s], h0k, ba, 0, 0, 0);
          bg = __builtin_amdgcn_mfma_f32_16x16x4f32(wgb[s], s1[prv][4 * s + g][r], bg, 0, 0, 0);
          ag = __builtin_amdgcn_mfma_f32_16x16x4f32(wga[s], h0k, ag, 0, 0, 0);
          if (s < D / 4) aa = __builtin_amdgcn_mfma_f32_16x16x4f32(wia[s], sx[prv][r][4 * s + g], aa, 0, 0, 0);
        }
#pragma unroll
        for (int s = H / 4; s < D / 4; ++s) aa = __builtin_amdgcn_mfma_f32_16x16x4f32(wia[s], sx[prv][r][4 * s + g], aa, 0, 0, 0);
        stash_x(cur, xn);
        fetch_x(t + 2 + NB, xn);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          h1v[qd] = tanh_rnn((bg[qd] + bgb[qd]) + (ba[qd] + bib[qd]));
          s1[cur][16 * wave + 4 * g + qd][r] = h1v[qd];
          s0[prv][16 * wave + 4 * g + qd][r] = tanh_rnn((ag[qd] + bga[qd]) + (aa[qd] + bia[qd]));
        }
      }
      __syncthreads();
    }
  }
  // top MLP input rows: [ last state | user profile | candidate ad | context ] (:411-421)
  if (live && do2) {
    float* out = R + (int64_t)(v0 + b) * ldr;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) out[16 * wave + 4 * g + qd] = h1v[qd];
  }
  for (int i = threadIdx.x; i < 16 * 3 * D; i += NT) {
    const int smp_i = smp_base + i / (3 * D), c = i % (3 * D);
    if (smp_i >= n_smp) break;
    int bi = smp_i, vi = q.vstart[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const bool in = k < q.n_q && smp_i >= q.cum[k];
      bi = in ? smp_i - q.cum[k] : bi;
      vi = in ? q.vstart[k] : vi;
    }
    if (q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
      for (int k = 8; k < DRS_MAX_COALESCE; ++k) {
        const bool in = k < q.n_q && smp_i >= q.cum[k];
        bi = in ? smp_i - q.cum[k] : bi;
        vi = in ? q.vstart[k] : vi;
      }
    }
    const int tab = c < D ? 0 : c < 2 * D ? Tn - 2 : Tn - 1;
    const float ev = T[(int64_t)(vi + bi) * ldt + (int64_t)tab * D + c % D];
    R[(int64_t)(vi + bi) * ldr + H + c] = ev;
    if (top.n > 0) dien_top_lds[(H + c) * 16 + i / (3 * D)] = ev;
  }
  if (top.n <= 0) return;                              // (uniform: a kernel argument)

  // ---- the top MLP of the workgroup's 16 samples, in the same launch (round 4) ----------------------------
  // Every CU holds two workgroups of this model at a time (the recurrence: 2 x 120 registers per SIMD; the
  // stream kernel that ran the top MLP: 256), so a set cost each CU t_rnn + t_top of workgroup time at two
  // in flight -- and the top launch, 19 us alone, took 60-110 us beside recurrences.  Here its three small
  // layers (160-200-80-2: 0.1 MFLOP per sample) follow the last step as MFMA chains of the same form:
  // A = 16 rows of W [N, K] (lane (r, g): W[16 t + r][4 s + g], every 64-k chunk of them requested up front),
  // B = the layer's input [k][sample] in LDS, D[m = unit 4 g + qd][n = sample r]; k ascending from a zero
  // accumulator (through the zero-padded end of the last 64-k chunk, like theirs), bias, activation: the bits of
  // the stream kernels' chains (mlp.hip).  Tiles of 16 units go
  // round the workgroup's waves; the last layer stores to the output buffer and the workgroup signs off the
  // launch set itself (signal_done).
  {
    constexpr int NWV = NT / 64;
    const int wv = threadIdx.x >> 6;
    float* in = dien_top_lds;
    float* nxt = dien_top_lds + top.kmax * 16;           // (kmax: a multiple of 64)
    const int lastb = (U - 1) & 1;
    for (int i = threadIdx.x; i < H * 16; i += NT) in[i] = (&s1[lastb][0][0])[i];
    // k beyond a layer's K up to the next multiple of 64 meets zero weights in the twin: the inputs there must
    // be finite -- zeros
    for (int i = (H + 3 * D) * 16 + threadIdx.x; i < ((top.K[0] + 63) & ~63) * 16; i += NT) in[i] = 0.f;
    __syncthreads();
    for (int l = 0; l < top.n; ++l) {
      const int K = top.K[l], N = top.N[l], nch = (K + 63) >> 6;
      const bool fin = l + 1 == top.n;
      if (!fin)
        for (int i = N * 16 + threadIdx.x; i < ((N + 63) & ~63) * 16; i += NT) nxt[i] = 0.f;
      for (int t = wv; 16 * t < N; t += NWV) {
        // A operands from the layer's PACKED twin (mlp.hip pack_stream_kernel: per 128 units and 64 k a block of
        // 8192 floats, 1024 per 16 units, float4 q of lane (r, g) = W[unit r][64 c + 16 q + 4 j + g], j = 0..3 --
        // element j is the operand of MFMA step 16 c + 4 q + j): one coalesced 1-KB request per four steps.
        // (The first version read W [N, K] itself, a dword per lane and step: 16 cache lines per request, and
        // the launch took 100 us alone against 69 + 22 for the two it replaced.)
        const float* wp = top.Wp[l] + ((size_t)(t >> 3) * nch * 8192 + (t & 7) * 1024 + lane * 4);
        f32x4_ wq[4][4];
        float bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nch) {                                  // (uniform)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) wq[c][qq] = *reinterpret_cast<const f32x4_*>(wp + c * 8192 + qq * 256);
          }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bv[qd] = top.b[l][min(16 * t + 4 * g + qd, N - 1)];
        f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);               // (all of the tile's requests before its first MFMA)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nch) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c][qq][j], in[(64 * c + 16 * qq + 4 * j + g) * 16 + r], acc, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);         // (keeps the LDS operands next to their MFMAs)
            }
          }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int unit = 16 * t + 4 * g + qd;
          if (unit < N) {
            const float v = act_apply(acc[qd] + bv[qd], top.act[l]);
            if (!fin) nxt[unit * 16 + r] = v;
            else if (live) {
              float* dst = top.out + (int64_t)(v0 + b) * top.ldo + unit;
              if (top.sc1) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else *dst = v;
            }
          }
        }
      }
      __syncthreads();
      float* sw = in; in = nxt; nxt = sw;
    }
  }
  signal_done(done, gridDim.x, dien_top_lds);
}

template <int D>
bool launch_dien_mfma_h(const float* T, int64_t ldt, const QTable& q, int Tn, int H, const DienW& W, float* R,
                        int64_t ldr, unsigned grid, int split, hipStream_t s, const DienTop& top, const Done& done) {
  const size_t lds = top.n > 0 ? sizeof(float) * 2 * 16 * (size_t)top.kmax : 0;
  if (split) {
    switch (H) {
      case 16: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 16, 1>), dim3(grid), dim3(128), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
      case 32: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 32, 1>), dim3(grid), dim3(256), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
      case 64: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 64, 1>), dim3(grid), dim3(512), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
      default: return false;
    }
  }
  switch (H) {
    case 16: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 16, 0>), dim3(grid), dim3(64), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
    case 32: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 32, 0>), dim3(grid), dim3(128), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
    case 64: hipLaunchKernelGGL((dien_rnn_mfma_kernel<D, 64, 0>), dim3(grid), dim3(256), lds, s, T, ldt, q, Tn, W, R, ldr, top, done); return true;
    default: return false;
  }
}

template <int D>
bool launch_dien_h(const float* T, int64_t ldt, const QTable& q, int Tn, int H, const float* packed, float* R,
                   int64_t ldr, unsigned grid, hipStream_t s) {
  switch (H) {
    case 8: hipLaunchKernelGGL((dien_rnn_kernel<D, 8>), dim3(grid), dim3(256), 0, s, T, ldt, q, Tn, packed, R, ldr); return true;
    case 16: hipLaunchKernelGGL((dien_rnn_kernel<D, 16>), dim3(grid), dim3(256), 0, s, T, ldt, q, Tn, packed, R, ldr); return true;
    case 32: hipLaunchKernelGGL((dien_rnn_kernel<D, 32>), dim3(grid), dim3(256), 0, s, T, ldt, q, Tn, packed, R, ldr); return true;
    case 64: hipLaunchKernelGGL((dien_rnn_kernel<D, 64>), dim3(grid), dim3(256), 0, s, T, ldt, q, Tn, packed, R, ldr); return true;
    default: return false;
  }
}

}  // namespace

// Shapes the recurrent kernel is instantiated for.
bool dien_applicable(int32_t D, int32_t H) { return (D == 16 || D == 32 || D == 64) && (H == 8 || H == 16 || H == 32 || H == 64); }
int64_t dien_packed_floats(int32_t D, int32_t H) { return (int64_t)D * H + 3ll * H * H + 4ll * H; }

hipError_t launch_dien_pack(const float* const* w, float* packed, int32_t D, int32_t H, hipStream_t s) {
  const int64_t blocks = ((int64_t)(D > H ? D : H) * H + 255) / 256;
  hipLaunchKernelGGL(dien_pack_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, w, packed, D, H);
  return hipGetLastError();
}

// The top MLP rides in the recurrence's launch when its layers fit the in-kernel form: at most 4 of them, every
// input width a multiple of 4 (whole MFMA steps) and <= 256 (a tile's A operands live in 64 registers), the two
// activation buffers inside the default 64 KB of LDS next to the state buffers.
bool dien_top_fusable(int32_t n_layers, const int32_t* widths, int32_t H) {
  if (n_layers < 1 || n_layers > 4 || H % 16 != 0) return false;
  for (int l = 0; l < n_layers; ++l)
    if (widths[l] <= 0 || widths[l] > 256 || (widths[l] & 3) || widths[l + 1] <= 0) return false;
  return sizeof(float) * (2 * 16 * (size_t)dien_top_kmax(n_layers, widths) + 4 * 16 * (size_t)H) <= 60 * 1024;
}
// rows of one LDS activation buffer: the widest layer input, up to the end of its last 64-k chunk
int32_t dien_top_kmax(int32_t n_layers, const int32_t* widths) {
  int kmax = 0;
  for (int l = 0; l < n_layers; ++l) kmax = widths[l] > kmax ? widths[l] : kmax;
  return (kmax + 63) & ~63;
}

hipError_t launch_dien_rnn(const float* T, int64_t ldt, const QTable& q, int32_t Tn, int32_t D, int32_t H,
                           const float* packed, const float* const* w, int mfma, float* R, int64_t ldr,
                           hipStream_t s, const DienTop* top, const Done* done) {
  const int64_t n = q.cum[q.n_q];
  if (n <= 0) return hipSuccess;
  bool ok = false;
  // shapes without an instance of their own (and "dien_mfma" 3, which the parity tests use): the any-shape form
  if (!dien_applicable(D, H) || mfma == 3)
    return top && top->n > 0 ? hipErrorInvalidValue : launch_dien_rnn_any(T, ldt, q, Tn, D, H, packed, R, ldr, s);
  if (top && top->n > 0 && !(mfma && H % 16 == 0)) return hipErrorInvalidValue;   // (the engine asks dien_top_fusable first)
  if (mfma && H % 16 == 0) {
    DienW W;
    for (int i = 0; i < 8; ++i) W.w[i] = w[i];
    DienTop tp;
    memset(&tp, 0, sizeof tp);
    if (top) tp = *top;
    Done dn;
    memset(&dn, 0, sizeof dn);
    if (done && tp.n > 0) dn = *done;
    const unsigned g16 = (unsigned)((n + 15) / 16);
    if (D == 16) ok = launch_dien_mfma_h<16>(T, ldt, q, Tn, H, W, R, ldr, g16, mfma == 2, s, tp, dn);
    else if (D == 32) ok = launch_dien_mfma_h<32>(T, ldt, q, Tn, H, W, R, ldr, g16, mfma == 2, s, tp, dn);
    else if (D == 64) ok = launch_dien_mfma_h<64>(T, ldt, q, Tn, H, W, R, ldr, g16, mfma == 2, s, tp, dn);
    return ok ? hipGetLastError() : hipErrorInvalidValue;
  }
  const unsigned grid = (unsigned)((n + 3) / 4);
  if (D == 16) ok = launch_dien_h<16>(T, ldt, q, Tn, H, packed, R, ldr, grid, s);
  else if (D == 32) ok = launch_dien_h<32>(T, ldt, q, Tn, H, packed, R, ldr, grid, s);
  else if (D == 64) ok = launch_dien_h<64>(T, ldt, q, Tn, H, packed, R, ldr, grid, s);
  if (!ok) return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_din_pack(const float* const* att, float* packed, int32_t U, int32_t D, int32_t h, hipStream_t s) {
  hipLaunchKernelGGL(din_pack_kernel, dim3((unsigned)U), dim3(256), 0, s, att, packed, D, h, din_unit_stride(D, h));
  return hipGetLastError();
}

hipError_t launch_din_attention(const float* T, int64_t ldt, int64_t M, int32_t Tn, int32_t D, int32_t h,
                                const float* packed, float* R, int64_t ldr, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  const size_t lds = sizeof(float) * 4 * (size_t)(Tn - 3) * h;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(din_attention_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), lds, s, T, ldt, M, Tn, D, h,
                     packed, din_unit_stride(D, h), R, ldr);
  return hipGetLastError();
}

// Shapes the fused kernel is instantiated for: D = 32 or 64 (8 / 16 lanes x 16 B per row), hidden
// width 1, 2 or 4.
bool din_fused_applicable(int32_t D, int32_t h) { return (D == 32 || D == 64) && (h == 1 || h == 2 || h == 4); }
int64_t din_fused_grid(const SlsArgs& a, const Tune& tune) {
  const FusedShape f = fused_shape(a, tune);
  return (a.q.cum[a.q.n_q] + f.S - 1) / f.S;
}

hipError_t launch_din_fused(const SlsArgs& a_in, int32_t h, const float* packed, float* R, int64_t ldr, const Tune& tune,
                            hipStream_t s, hipEvent_t stop) {
  SlsArgs a = a_in;
  a.nt = tune.din_nt;
  const int64_t n_smp = a.q.cum[a.q.n_q];
  if (n_smp <= 0) return hipSuccess;
  const int64_t stride = din_unit_stride(a.D, h);
  const FusedShape f = fused_shape(a, tune);
  const unsigned grid = (unsigned)din_fused_grid(a, tune);
  // hidden width 1, fixed bag length <= 3, the staged indices fit LDS: the pipelined form (same bits)
  if (tune.din_pipe && h == 1 && f.C == 3 && din_pipe_lds(a.T, f.S) <= 48 * 1024) {
    log_launch(tune.log, "din_pipe_kernel<%d,S%d,P%d%s>[%u wg]", a.D == 32 ? 8 : 16, f.S, kPipe, a.nt ? ",nt" : "", grid);
    if (a.D == 32) launch_pipe_s<8>(f, a, packed, stride, tune.zero, R, ldr, grid, s, stop);
    else launch_pipe_s<16>(f, a, packed, stride, tune.zero, R, ldr, grid, s, stop);
    return hipGetLastError();
  }
  log_launch(tune.log, "din_fused_kernel<%d,S%d,h%d,C%d%s>[%u wg]", a.D == 32 ? 8 : 16, f.S, h, f.C, a.nt ? ",nt" : "", grid);
  if (a.D == 32) launch_fused_s<8>(f, a, h, packed, stride, tune.zero, R, ldr, grid, s, stop);
  else launch_fused_s<16>(f, a, h, packed, stride, tune.zero, R, ldr, grid, s, stop);
  return hipGetLastError();
}

}  // namespace drs

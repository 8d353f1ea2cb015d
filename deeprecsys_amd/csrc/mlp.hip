// FC / MLP / dot-interaction kernels for gfx950 on the fp32 matrix cores.
//
// Replaces the FC + Relu|Sigmoid operator pairs of create_mlp (reference
// models/dlrm_s_caffe2.py:223-279) and the Concat/BatchMatMul/Flatten/BatchGather/
// Concat chain of create_interactions (:331-365).
//
// Arithmetic contract (see oracle/drs_oracle.c): every output element is
//     act( fma-chain over k = 0..K-1 in order, starting from 0 )  + bias
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fp32 fma chain, and the
// operands are fed so that MFMA step s carries k = 4s .. 4s+3, so GPU and oracle
// agree bitwise up to the final expf of the sigmoid.
//
// Tiling: a workgroup (8 waves) owns a 16-row slab of the batch and 128 output
// columns at a time (one 16x16 tile per wave).  W (stored [N, K], K contiguous --
// already the "B^T" layout MFMA wants) streams through LDS in 64-deep K chunks; rows
// are padded to 68 floats and rows 8..15 stored at k^2 so the per-lane ds_read_b32
// operand fetches are conflict free.  The activations of a slab never leave LDS
// between layers.  Three kernels share this contract:
//   stream_kernel  all layers of one or two chains as ONE prefetched sequence of weight
//                  tiles (the default)
//   chain_kernel   per-layer passes (fallback: widths not a multiple of 4, inputs too
//                  wide for an LDS slab)
//   fc_kernel      one layer on a 2-D grid (fallback of gemm.hip's gemm_kernel)
#include <string.h>

#include "mlp_stream.h"

namespace drs {
namespace {


// K chunk staged per step is a template parameter KC in {64, 128, 192, 256}: a dependent
// global-load round costs ~1 us on this chip (Infinity-Cache latency; per-XCD L2s start
// cold every launch), far more than the MFMAs it feeds, so layers are cut into as few
// rounds as LDS allows.  Rows of a staged chunk are padded to KC+4 floats.


struct LayerIo {
  const float* a_glb;   // A operand in global memory (first layer) or nullptr
  int64_t lda_glb;
  int64_t a_row0;       // first row of this slab inside a_glb ...
  int64_t a_rows;       // ... which has this many valid rows
  const float* a_lds;   // A operand: activation slab in LDS (later layers) or nullptr
  int lda_lds;
  float* o_glb;         // output to global (last layer) or nullptr
  int64_t ldo_glb;
  float* o_lds;         // output slab in LDS or nullptr
  int ldo_lds;
  bool o_sc1;           // outputs of the query's LAST layer: write-through (agent-scope) stores,
                        // so publishing them to the last-arriving workgroup needs no L2 write-back fence
};

// One layer for the block's 16 rows [m0, m0+16) and the columns [n_begin, n_end).
// The workgroup is 8 waves (512 threads): a pass covers 128 columns, wave w owns the
// 16-column tile at n0 + 16w (on layers narrower than 128 the upper waves only help with
// the staging).  Two waves per SIMD is the point: with one wave per SIMD the ~390
// instructions of a K-chunk round (address math, selects, LDS traffic around only 32
// MFMAs) issue back to back with nothing to hide their latencies -- the in-kernel
// timeline showed 3.6 k cycles per round against 1 k cycles of MFMA.  Eight waves split the
// same round into streams half as long that interleave on each SIMD.
// Every output element is one k-ordered fma chain.
// sA: [nbuf][16][KC+4] (used only when A comes from global), sB: [nbuf][128][KC+4];
// nbuf = 2 (double buffered) when the layer needs more than one K chunk, else 1.
constexpr int kThreads = 512;
constexpr int PN = 128;                  // columns per pass (8 waves x 16)

template <bool A_LDS, bool O_LDS, bool VEC, int KC>
__device__ __forceinline__ void layer_pass(const LayerIo io, int64_t m0, int64_t M, int K,
                                           const float* __restrict__ W, int64_t ldw,
                                           const float* __restrict__ bias, int N, int n_begin,
                                           int n_end, int act, int nbuf, float* sA, float* sB TL_PARAM) {
  constexpr int BMK = 16;
  constexpr int LD = KC + 4;
  constexpr int QPR = KC / 4;                       // float4 per staged row
  constexpr int NA = (16 * QPR + kThreads - 1) / kThreads;   // float4 of A per thread per chunk (KC=64: half the threads)
  constexpr int NB = PN * QPR / kThreads;           // float4 of W per thread per chunk
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 15;   // row of A / column of the tile
  const int g = lane >> 4;   // k within an MFMA step
  const int n_chunks = (K + KC - 1) / KC;

  for (int n0 = n_begin; n0 < n_end; n0 += PN) {
    const bool my_tile = n0 + wave * 16 < n_end;    // wave-uniform: is there a tile for me?
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 ra[NA], rb[NB];
    auto fetch = [&](int kc) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int idx = min(tid + i * kThreads, 16 * QPR - 1);
        if (!A_LDS) ra[i] = load4_raw<VEC>(io.a_glb, io.lda_glb, io.a_row0 + idx / QPR, io.a_rows, kc + (idx % QPR) * 4, K);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * kThreads;
        rb[i] = load4_raw<VEC>(W, ldw, n0 + idx / QPR, N, kc + (idx % QPR) * 4, K);
      }
    };
    auto stash = [&](int buf, int kc) {
      const bool tail = kc + KC > K;              // uniform: only the last chunk needs the k mask
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * kThreads;
        const int row = idx / QPR, k = (idx % QPR) * 4;
        if (!A_LDS && idx < 16 * QPR)
          *reinterpret_cast<float4*>(sA + (buf * BMK + row) * LD + k) =
              swz4(tail ? mask4(ra[i], kc + k, K) : ra[i], row);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * kThreads;
        const int row = idx / QPR, k = (idx % QPR) * 4;
        *reinterpret_cast<float4*>(sB + (buf * PN + row) * LD + k) =
            swz4(tail ? mask4(rb[i], kc + k, K) : rb[i], row);
      }
    };
    TL(1);
    fetch(0);
    TL(2);
    stash(0, 0);
    TL(3);
    __syncthreads();
    TL(4);

    for (int c = 0; c < n_chunks; ++c) {
      const int buf = c & (nbuf - 1);
      const bool more = c + 1 < n_chunks;
      TL(10);
      if (more) fetch((c + 1) * KC);   // next chunk's global loads fly during the MFMAs
      TL(11);

      if (my_tile) {
        const int gs = swz(g, r);                         // see swz4: rows 8..15 live at k^2
        const float* pa = A_LDS ? io.a_lds + r * io.lda_lds + c * KC + gs
                                : sA + (buf * BMK + r) * LD + gs;
        const float* pb = sB + (buf * PN + wave * 16 + r) * LD + gs;
        const int ksteps = min(KC, K - c * KC + 3) / 4;   // steps that carry real k
        // Only the last chunk of an LDS activation slab can hold stale columns past K (staged
        // chunks are zero filled there): keep the select out of the steady state.
        const bool a_tail = A_LDS && (c + 1) * KC > K;
        // operands of 16 steps (64 k) are read together (one counted lgkmcnt stream), then
        // their 16 MFMAs; the SIMD's second wave covers the LDS latency in between
#pragma unroll
        for (int sg = 0; sg < KC / 64; ++sg) {
          if (16 * sg < ksteps) {                           // uniform
            float av[16], bv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
              av[s] = pa[4 * (16 * sg + s)];
              bv[s] = pb[4 * (16 * sg + s)];
            }
            if (a_tail) {
#pragma unroll
              for (int s = 0; s < 16; ++s) av[s] = (c * KC + 4 * (16 * sg + s) + g < K) ? av[s] : 0.f;
            }
            // fma(0, 0, acc) == acc, so a padded step is exact; skip groups of 4 uniformly
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (16 * sg + 4 * q < ksteps) {
#pragma unroll
                for (int s = 4 * q; s < 4 * q + 4; ++s)
                  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
              }
            }
          }
        }
      }
      TL(12);
      if (more) stash(buf ^ 1, (c + 1) * KC);
      TL(13);
      __syncthreads();
      TL(14);
    }

    // epilogue: bias + activation; lane holds rows g*4+i of its tile, column r
    const int col = n0 + wave * 16 + r;
    if (my_tile && col < N) {
      const float bcol = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = g * 4 + i;
        const float v = act_apply(acc[i] + bcol, act);
        if (O_LDS) {
          io.o_lds[row * io.ldo_lds + swz(col, row)] = v;
        } else if (m0 + row < M) {
          float* dst = io.o_glb + (m0 + row) * io.ldo_glb + col;
          if (io.o_sc1) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dst = v;
        }
      }
    }
  }
}


// Single layer, 2-D grid: blockIdx.x = 16-row slab, blockIdx.y = 128-column group.
template <bool VEC, int KC>
__global__ __launch_bounds__(512) void fc_kernel(const float* __restrict__ x, int64_t ldx, int64_t M,
                                                 int K, const float* __restrict__ W, int64_t ldw,
                                                 const float* __restrict__ b, int N, int act,
                                                 float* __restrict__ y, int64_t ldy, int nbuf,
                                                 Done done, XSrc xs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                              // [nbuf][16][KC+4]
  float* sB = sA + nbuf * 16 * (KC + 4);         // [nbuf][128][KC+4]
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(sB + nbuf * PN * (KC + 4));
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
#endif
  LayerIo io = {x, ldx, 0, 0, nullptr, 0, y, ldy, nullptr, 0, done.counter != nullptr};
  resolve_src(xs, x, M, (int64_t)blockIdx.x * 16, &io.a_glb, &io.a_row0, &io.a_rows);
  const int n0 = blockIdx.y * PN;
  layer_pass<false, false, VEC, KC>(io, (int64_t)blockIdx.x * 16, M, K, W, ldw, b, N, n0,
                                    min(n0 + PN, N), act, nbuf, sA, sB TL_ARG);
#ifdef DRS_TIMELINE
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
  signal_done(done, gridDim.x * gridDim.y, smem);
}

// One chain of layers on the block's 16 rows; activations ping-pong between two LDS slabs.
//
// The chain's input rows are streamed exactly once by exactly one workgroup, so every
// chunk of them is a compulsory miss all the way to HBM / Infinity Cache (~2 us) that a
// one-chunk-ahead prefetch cannot hide.  When they fit (slabA != nullptr) all 16 x K0
// inputs are therefore pulled into LDS with ONE round of loads up front and the first
// layer reads its A operand from LDS like every later layer; only the weights (shared by
// all workgroups, L2 resident after the warm-up) keep streaming per K chunk.
template <bool VEC, int KC>
__device__ __forceinline__ void run_chain(const ChainArgs& a, const XSrc& xs, int64_t m0, int slab_ld,
                                          int nbuf, float* sA, float* sB, float* slab0, float* slab1,
                                          float* slabA, int ldA, bool publish TL_PARAM) {
  float* cur = nullptr;
  const int K0 = a.width[0];
  const bool pre = slabA != nullptr && K0 <= 640;
  if (pre) {
    const float* base; int64_t row0, rows;
    resolve_src(xs, a.x, a.M, m0, &base, &row0, &rows);
    const int qpr = (K0 + 3) / 4;                 // float4 per row
    const int total = 16 * qpr;
    float4 v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = min((int)threadIdx.x + i * kThreads, total - 1);
      v[i] = load4_raw<VEC>(base, a.ldx, row0 + idx / qpr, rows, (idx % qpr) * 4, K0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      if (idx < total)
        *reinterpret_cast<float4*>(slabA + (idx / qpr) * ldA + (idx % qpr) * 4) =
            swz4(mask4(v[i], (idx % qpr) * 4, K0), idx / qpr);
    }
    __syncthreads();
  }
  for (int l = 0; l < a.n_layers; ++l) {
    const bool first = l == 0, last = l == a.n_layers - 1;
    const bool a_lds = !first || pre;
    float* nxt = (l & 1) ? slab1 : slab0;
    LayerIo io;
    io.a_glb = nullptr;
    io.a_row0 = io.a_rows = 0;
    if (first && !pre) resolve_src(xs, a.x, a.M, m0, &io.a_glb, &io.a_row0, &io.a_rows);
    io.lda_glb = a.ldx;
    io.a_lds = first ? (pre ? slabA : nullptr) : cur;
    io.lda_lds = first ? ldA : slab_ld;
    io.o_glb = last ? a.y : nullptr;
    io.ldo_glb = a.ldy;
    io.o_lds = last ? nullptr : nxt;
    io.ldo_lds = slab_ld;
    io.o_sc1 = last && publish;
    const int K = a.width[l], N = a.width[l + 1];
#define DRS_PASS(AL, OL)                                                                          \
  layer_pass<AL, OL, VEC, KC>(io, m0, a.M, K, a.W[l], K, a.b[l], N, 0, N, a.act[l], nbuf, sA, sB TL_ARG)
    if (!a_lds && last) { DRS_PASS(false, false); }
    else if (!a_lds) { DRS_PASS(false, true); }
    else if (last) { DRS_PASS(true, false); }
    else { DRS_PASS(true, true); }
#undef DRS_PASS
    __syncthreads();
    cur = nxt;
  }
}

// Up to two chains back to back in ONE launch on the same 16 rows: the bottom MLP
// (dense features -> dense_out slot of the interaction buffer) and, for the "cat"
// interaction, the top MLP that reads that buffer.  The second chain re-reads rows this
// very workgroup wrote: a workgroup-scope fence + barrier orders that.
template <bool VEC, int KC>
__global__ __launch_bounds__(512) void chain_kernel(ChainArgs a0, ChainArgs a1, int slab_ld, int nbuf,
                                                    int ldA, Done done, XSrc xs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                              // [nbuf][16][KC+4]
  float* sB = sA + nbuf * 16 * (KC + 4);         // [nbuf][128][KC+4]
  float* slab0 = sB + nbuf * PN * (KC + 4);      // [16][slab_ld]
  float* slab1 = slab0 + 16 * slab_ld;
  float* slabA = ldA > 0 ? slab1 + 16 * slab_ld : nullptr;   // [16][ldA] preloaded chain input
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(slab1 + 16 * slab_ld + (ldA > 0 ? 16 * ldA : 0));
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
#endif
  const int64_t m0 = (int64_t)blockIdx.x * 16;

  // L2 warm-up.  All workgroups walk the same weights in lock step, so without help every
  // K chunk is a compulsory miss in each XCD's L2 (the gather before us has flushed it) and
  // every staging round pays a full Infinity-Cache/HBM latency (~2 us, measured: waves 53%
  // in s_waitcnt).  Here each workgroup touches ONE slice of all the weights, one load per
  // 128-B line, fire-and-forget: the XCD's 16 or so resident workgroups together pull the
  // whole set into their L2 during the first layer's prologue.  Purely a hint: a different
  // workgroup->XCD placement changes speed, not results.
  float warm[2 * DRS_MAX_CHAIN];   // consumed only at the very end: never waited for early
  {
    const unsigned part = (blockIdx.x >> 3) & 15;          // my rank among the XCD's workgroups
    auto touch = [&](const ChainArgs& c, int l) -> float {
      if (l >= c.n_layers) return 0.f;
      const int64_t lines = ((int64_t)c.width[l] * c.width[l + 1] + 31) / 32;   // 128-B lines
      // 16 parts x 512 threads x 1 line: covers 1 MB per layer (all of RM1/RM2's layers)
      const int64_t i = min(lines - 1, (int64_t)part * kThreads + threadIdx.x + (int64_t)(blockIdx.x >> 7) * 16 * kThreads);
      return c.W[l][i * 32];
    };
#pragma unroll
    for (int l = 0; l < DRS_MAX_CHAIN; ++l) {
      warm[l] = touch(a0, l);
      warm[DRS_MAX_CHAIN + l] = touch(a1, l);
    }
  }

  // zero both slabs once: padded K tails of later layers must read finite values
  for (int i = threadIdx.x; i < 2 * 16 * slab_ld; i += blockDim.x) slab0[i] = 0.f;
  __syncthreads();

  run_chain<VEC, KC>(a0, xs, m0, slab_ld, nbuf, sA, sB, slab0, slab1, slabA, ldA,
                     done.counter != nullptr && a1.n_layers == 0 TL_ARG);
  if (a1.n_layers > 0) {
    __threadfence_block();
    __syncthreads();
    XSrc none;
    none.q.n_q = 0;
    run_chain<VEC, KC>(a1, none, m0, slab_ld, nbuf, sA, sB, slab0, slab1, slabA, ldA,
                       done.counter != nullptr TL_ARG);
  }
#pragma unroll
  for (int l = 0; l < 2 * DRS_MAX_CHAIN; ++l) asm volatile("" ::"v"(warm[l]));
#ifdef DRS_TIMELINE
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
  signal_done(done, gridDim.x, smem);
}

// The packed twin of a layer's weights (stream_kernel<true>): tile (pass p, chunk c) = 8192 floats,
// wave w's block = 1024, float4 q of lane (r, g) = { W[128 p + 16 w + r][64 c + 16 q + 4 j + g] : j = 0..3 },
// i.e. element j of float4 q is the B operand of MFMA step s = 4 q + j (k = 64 c + 4 s + g: natural
// k order); rows beyond N and k beyond K are zeros.
__global__ __launch_bounds__(512) void pack_stream_kernel(const float* __restrict__ W, int K, int N,
                                                          float* __restrict__ Wp) {
  const int nch = (K + 63) >> 6;
  const int tile = blockIdx.x, p = tile / nch, c = tile - p * nch;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int row = 128 * p + 16 * wave + r;
  float* o = Wp + (size_t)tile * 8192 + wave * 1024 + lane * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 v;
    float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 64 * c + 16 * q + 4 * j + g;
      vv[j] = (row < N && k < K) ? W[(size_t)row * K + k] : 0.f;
    }
    *reinterpret_cast<float4*>(o + q * 256) = v;
  }
}

// ---------------------------------------------------------------------------
// dot interaction: one wave per sample.  T[b] is [F, D]; Z = T T^T is computed
// in 16x16 MFMA tiles (A and B operands are the same register: B[k][j] = T[j][k]),
// the strictly-lower (or lower, with `itself`) triangle is scattered in the
// row-major BatchGather order i*(i-1)/2 + j (resp. i*(i+1)/2 + j) behind a copy
// of the dense row T[b][0][:].
__global__ __launch_bounds__(256) void interact_dot_kernel(const float* __restrict__ T, int64_t ldt,
                                                           int64_t B, int F, int D, int itself,
                                                           float* __restrict__ R, int64_t ldr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  const int Fp = (F + 15) & ~15;
  const int ldl = D + 1;                       // odd stride: conflict-free column reads
  float* t = smem + (size_t)wave * Fp * ldl;
  if (b >= B) return;                          // whole wave exits together
  const float* src = T + b * ldt;
  for (int i = lane; i < Fp * D; i += 64) {
    const int f = i / D, d = i - f * D;
    t[f * ldl + d] = f < F ? src[(int64_t)f * D + d] : 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  float* out = R + b * ldr;
  for (int d = lane; d < D; d += 64) out[d] = t[d];
  const int r = lane & 15, g = lane >> 4;
  const int ksteps = (D + 3) / 4;
  for (int ti = 0; ti < Fp; ti += 16)
    for (int tj = 0; tj <= ti; tj += 16) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < ksteps; ++s) {
        const int k = 4 * s + g;
        const float av = k < D ? t[(ti + r) * ldl + k] : 0.f;
        const float bv = k < D ? t[(tj + r) * ldl + k] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
      const int j = tj + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ti + g * 4 + q;
        if (i < F && (itself ? j <= i : j < i)) {
          const int p = itself ? i * (i + 1) / 2 + j : i * (i - 1) / 2 + j;
          out[D + p] = acc[q];
        }
      }
    }
}

__global__ void add_rows_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                int64_t ldb, float* __restrict__ o, int64_t ldo, int64_t M, int D) {
  const int64_t n = M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D;
    const int d = (int)(i - m * D);
    const float v = b ? a[m * lda + d] + b[m * ldb + d] : a[m * lda + d];
    o[m * ldo + d] = v;
  }
}

// Dense rows of up to DRS_MAX_COALESCE coalesced queries (one staged array per query) -> their virtual rows
// of the concat buffer, in ONE launch (W&D has no bottom MLP: models/wide_and_deep.py:271-281).
// V = 4: 16 bytes per thread (m_den, ldo multiples of 4, every pointer 16-byte aligned) -- the owner lookup below
// is per THREAD, and at a dword per thread it made W&D's 4 096 x 512 copy a 23-us launch (0.7 TB/s).
template <int V>
__global__ void copy_rows_multi_kernel(XSrc xs, int m_den, float* __restrict__ o, int64_t ldo) {
  const int64_t Mv = xs.q.vstart[xs.q.n_q];
  const int mv = m_den / V;
  const int64_t n = Mv * mv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / mv;
    const int d = (int)(i - v * mv) * V;
    const float* p = xs.x[0];
    int lo = xs.q.vstart[0], nb = xs.q.bs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const bool in = k < xs.q.n_q && v >= xs.q.vstart[k];
      p = in ? xs.x[k] : p;
      lo = in ? xs.q.vstart[k] : lo;
      nb = in ? xs.q.bs[k] : nb;
    }
    if (xs.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
      for (int k = 8; k < DRS_MAX_COALESCE; ++k) {
        const bool in = k < xs.q.n_q && v >= xs.q.vstart[k];
        p = in ? xs.x[k] : p;
        lo = in ? xs.q.vstart[k] : lo;
        nb = in ? xs.q.bs[k] : nb;
      }
    }
    const int64_t r = v - lo;
    if (r < nb) {
      if (V == 4) *reinterpret_cast<float4*>(o + v * ldo + d) = *reinterpret_cast<const float4*>(p + r * m_den + d);
      else o[v * ldo + d] = p[r * m_den + d];
    }
  }
}

}  // namespace

int64_t stream_packed_floats(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return (int64_t)((N + 127) / 128) * ((K + 63) / 64) * 8192;
}

hipError_t launch_pack_stream_weights(const float* W, int32_t K, int32_t N, float* Wp, hipStream_t s) {
  const unsigned tiles = (unsigned)(((N + 127) / 128) * ((K + 63) / 64));
  if (!tiles) return hipSuccess;
  hipLaunchKernelGGL(pack_stream_kernel, dim3(tiles), dim3(512), 0, s, W, K, N, Wp);
  return hipGetLastError();
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

constexpr size_t kLdsBudget = 156 * 1024;

static size_t stage_bytes(int kc, int nbuf, int) {
  return sizeof(float) * (size_t)nbuf * (16 + PN) * (kc + 4);
}

// Fewest K rounds that fit the LDS budget next to `extra` bytes of slabs.
// force_kc: drs_set_option "mlp_kc" (0 = fewest rounds that fit)
static bool pick_kc(int maxK, size_t extra, int nt, int force_kc, int* kc_out, int* nbuf_out) {
  const int cands[4] = {256, 192, 128, 64};
  int best_kc = 0, best_nbuf = 0, best_rounds = 1 << 30;
  for (int kc : cands) {
    if (force_kc && kc != force_kc) continue;
    const int rounds = (maxK + kc - 1) / kc;
    const int nbuf = rounds > 1 ? 2 : 1;
    if (stage_bytes(kc, nbuf, nt) + extra > kLdsBudget) continue;
    if (rounds < best_rounds || (rounds == best_rounds && kc < best_kc)) {
      best_rounds = rounds; best_kc = kc; best_nbuf = nbuf;
    }
  }
  if (!best_kc) return false;
  *kc_out = best_kc; *nbuf_out = best_nbuf;
  return true;
}


#define DRS_FOR_EACH_KC(X) X(64) X(128) X(192) X(256)

// HIP function attributes are per device: device_init() (engine.hip) calls this once for
// every device an engine is created on.
hipError_t mlp_set_attrs() {
  hipError_t e = hipSuccess;
#define SET_ATTR(KC_)                                                               \
  if (e == hipSuccess) e = set_max_lds(fc_kernel<true, KC_>);                       \
  if (e == hipSuccess) e = set_max_lds(fc_kernel<false, KC_>);                      \
  if (e == hipSuccess) e = set_max_lds(chain_kernel<true, KC_>);                    \
  if (e == hipSuccess) e = set_max_lds(chain_kernel<false, KC_>);
  DRS_FOR_EACH_KC(SET_ATTR)
#undef SET_ATTR
  if (e == hipSuccess) e = stream8_set_attrs();
  if (e == hipSuccess) e = stream4_set_attrs();
  if (e == hipSuccess) e = set_max_lds(interact_dot_kernel);
  return e;
}

hipError_t launch_fc(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W,
                     const float* b, int32_t N, int32_t act, float* y, int64_t ldy,
                     const Tune& tune, hipStream_t s, const Done* done, const XSrc* xsrc) {
  if (M <= 0) return hipSuccess;
  Done d;
  memset(&d, 0, sizeof d);
  if (done) d = *done;
  XSrc xs;
  memset(&xs, 0, sizeof xs);
  if (xsrc) xs = *xsrc;
  if (N >= 64 && K >= 64) {
    hipError_t ge = hipSuccess;
    if (launch_gemm(x, ldx, M, K, W, b, N, act, y, ldy, tune, s, d, xs, &ge)) return ge;
  }
  int kc = 64, nbuf = 2;
  if (!pick_kc(K, 0, 2, tune.mlp_kc, &kc, &nbuf)) return hipErrorInvalidValue;
#ifdef DRS_TIMELINE
  const size_t lds = stage_bytes(kc, nbuf, 2) + 8192;
#else
  const size_t lds = stage_bytes(kc, nbuf, 2);
#endif
  dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + PN - 1) / PN));
  bool vec = aligned16(x) && aligned16(W) && (ldx & 3) == 0 && (K & 3) == 0;
  for (int i = 0; i < xs.q.n_q; ++i) vec = vec && aligned16(xs.x[i]);
  log_launch(tune.log, "fc_kernel<%s,%d>[%u x %u wg, %dx%d]", vec ? "vec" : "scalar", kc, grid.x, grid.y, K, N);
#define LAUNCH(KC_)                                                                               \
  if (kc == KC_) {                                                                                \
    if (vec)                                                                                      \
      hipLaunchKernelGGL((fc_kernel<true, KC_>), grid, dim3(kThreads), lds, s, x, ldx, M, K, W,   \
                         (int64_t)K, b, N, act, y, ldy, nbuf, d, xs);                             \
    else                                                                                          \
      hipLaunchKernelGGL((fc_kernel<false, KC_>), grid, dim3(kThreads), lds, s, x, ldx, M, K, W,  \
                         (int64_t)K, b, N, act, y, ldy, nbuf, d, xs);                             \
  }
  DRS_FOR_EACH_KC(LAUNCH)
#undef LAUNCH
  return hipGetLastError();
}

static int chain_slab_ld2(const ChainArgs& a, const ChainArgs* b) {
  int w = 4;
  for (int l = 1; l < a.n_layers; ++l) w = a.width[l] > w ? a.width[l] : w;  // slabs hold layer outputs
  if (b) for (int l = 1; l < b->n_layers; ++l) w = b->width[l] > w ? b->width[l] : w;
  return (w + 3) / 4 * 4 + 4;
}

// ldA > 0: the chains' input slab (16 x K0) is preloaded into LDS (see run_chain)
static bool chain_plan(const ChainArgs& a, const ChainArgs* b, const Tune& tune, int* kc, int* nbuf,
                       size_t* lds, int* ldA) {
  int maxK = 1, k0 = a.width[0];
  for (int l = 0; l < a.n_layers; ++l) maxK = a.width[l] > maxK ? a.width[l] : maxK;
  if (b) {
    for (int l = 0; l < b->n_layers; ++l) maxK = b->width[l] > maxK ? b->width[l] : maxK;
    k0 = b->width[0] > k0 ? b->width[0] : k0;
  }
  const size_t slabs = sizeof(float) * (size_t)2 * 16 * chain_slab_ld2(a, b);
  const int lda = (k0 + 3) / 4 * 4 + 4;
  const size_t pre = (tune.mlp_preload && k0 <= 640) ? sizeof(float) * (size_t)16 * lda : 0;
  if (pre && pick_kc(maxK, slabs + pre, 2, tune.mlp_kc, kc, nbuf)) {
    *lds = stage_bytes(*kc, *nbuf, 2) + slabs + pre;
    *ldA = lda;
    return true;
  }
  if (!pick_kc(maxK, slabs, 2, tune.mlp_kc, kc, nbuf)) return false;
  *lds = stage_bytes(*kc, *nbuf, 2) + slabs;
  *ldA = 0;
  return true;
}

size_t chain_lds_bytes(const ChainArgs& a, const Tune& tune) {
  int kc, nbuf, lda;
  size_t lds;
  return chain_plan(a, nullptr, tune, &kc, &nbuf, &lds, &lda) ? lds : (size_t)1 << 30;
}

size_t chain2_lds_bytes(const ChainArgs& a, const ChainArgs& b, const Tune& tune) {
  int kc, nbuf, lda;
  size_t lds;
  return chain_plan(a, &b, tune, &kc, &nbuf, &lds, &lda) ? lds : (size_t)1 << 30;
}

static inline int pad64(int n) { return (n + 63) & ~63; }

// Lay the chain(s) out for stream_kernel.  false = not applicable (caller uses chain_kernel).
static bool stream_plan(const ChainArgs& a, const ChainArgs* b, const Tune& tune, const XSrc& xs,
                        bool publish, SArgs* out, size_t* lds_bytes, const DotArgs* dot = nullptr,
                        const SumArgs* sum = nullptr, bool d_wait = false /* the launch polls Done::wait_flag */,
                        NSplit* nsp = nullptr) {
  SArgs& p = *out;
  if (nsp) memset(nsp, 0, sizeof *nsp);
  memset(&p, 0, sizeof p);
  const int na = a.n_layers, nb = b ? b->n_layers : 0;
  if (na + nb > DRS_MAX_STREAM_LAYERS) return false;
  // second chain must read the buffer the first one writes (dense_out slot in front)
  const int d_out = a.width[na];
  int dotP = 0;
  if (dot) {
    // bottom -> T (dense_out slot) -> interaction -> R -> top
    dotP = dot->F * (dot->F - 1) / 2 + (dot->itself ? dot->F : 0);
    if (!b || dot->T != a.y || dot->ldt != a.ldy || dot->R != b->x || dot->ldr != b->ldx ||
        dot->D != d_out || (d_out & 3) || b->width[0] != d_out + dotP || dot->F < 2)
      return false;
  } else if (sum) {
    // [ sum of two column blocks | first chain's output ] -> second chain
    if (!b || dot || sum->cols <= 0 || (sum->cols & 3) || (sum->col_a & 3) || (sum->col_b & 3) || (sum->ld & 3) ||
        (sum->ldd & 3) || !aligned16(sum->src) || !aligned16(sum->dst) || b->x != sum->dst ||
        b->ldx != sum->ldd || a.y != sum->dst + sum->cols || a.ldy != sum->ldd ||
        b->width[0] != sum->cols + d_out || (d_out & 3))
      return false;
  } else if (b && (b->x != a.y || b->ldx != a.ldy || d_out > b->width[0] || (d_out & 3))) {
    return false;
  }
  auto ok_ptr = [](const void* q) { return aligned16(q); };
  // every weight matrix must live inside the engine's arena (tile addresses are 32-bit byte
  // offsets from its base)
  auto in_arena = [&](const float* w, int64_t n) {
    return tune.w_arena && w >= tune.w_arena && w + n <= tune.w_arena + tune.w_arena_floats &&
           tune.w_arena_floats < (1ull << 30);
  };
  for (int l = 0; l < na; ++l) if (!in_arena(a.W[l], (int64_t)a.width[l] * a.width[l + 1])) return false;
  for (int l = 0; l < nb; ++l) if (!in_arena(b->W[l], (int64_t)b->width[l] * b->width[l + 1])) return false;
  if (!ok_ptr(a.x) || (a.ldx & 3)) return false;
  for (int i = 0; i < xs.q.n_q; ++i) if (!ok_ptr(xs.x[i])) return false;
  for (int l = 0; l < na; ++l) if (!ok_ptr(a.W[l]) || (a.width[l] & 3)) return false;
  if (b) {
    if (!ok_ptr(b->x) || (b->ldx & 3)) return false;
    for (int l = 0; l < nb; ++l) if (!ok_ptr(b->W[l]) || (b->width[l] & 3)) return false;
  }
  // biases back to back, each padded to 4 floats (how the engine's arena lays them out)
  {
    const float* expect = a.b[0];
    if (!expect) return false;
    for (int l = 0; l < na; ++l) { if (a.b[l] != expect) return false; expect += (a.width[l + 1] + 3) & ~3; }
    for (int l = 0; l < nb; ++l) { if (b->b[l] != expect) return false; expect += (b->width[l + 1] + 3) & ~3; }
  }
  // the packed form ("mlp_stream" 2): every layer must carry its packed twin (engine layers of the
  // bottom / top / final / task MLPs do: drs_set_fc)
  bool pk = tune.mlp_stream >= 2 && tune.w_packed_hi > tune.w_packed_lo;
  {
    auto has_twin = [&](const float* w) {
      const uint64_t o = (uint64_t)(w - tune.w_arena);
      return o >= tune.w_packed_lo && o < tune.w_packed_hi;
    };
    for (int l = 0; l < na; ++l) pk = pk && has_twin(a.W[l]);
    for (int l = 0; l < nb; ++l) pk = pk && has_twin(b->W[l]);
  }
  // "mlp_stream" 4: stream4_kernel -- four waves x up to four tiles, b128 activation operands, the step table below
  // run segment by segment; its steps must fit the descriptor table
  const bool f4 = tune.mlp_stream == 4;
  const int nt3 = 4, nw3 = 16 / nt3;   // tiles per wave at most; waves
  auto tpw3 = [&](int N, int out_pad) {       // tiles per wave of a layer: 1 / 2 (/ 4): a pass covers nw3 * tpw tiles
    const int etl = ((out_pad > N ? out_pad : N) + 15) / 16;
    int t = 1;
    while (t < nt3 && etl > nw3 * t) t *= 2;
    return t;
  };
  auto steps3 = [&](int K, int N, int out_pad) {
    const int etl = ((out_pad > N ? out_pad : N) + 15) / 16, tpp = nw3 * tpw3(N, out_pad);
    return ((etl + tpp - 1) / tpp) * ((K + 63) / 64);
  };
  bool f3 = pk && f4;
  // Column-split form ("mlp_nsplit"; SArgs::ns): the first layer of the second chain over ns workgroups per slab of rows.
  // A slice is ONE pass of the four waves: N / ns in {64, 128, 256} columns (1 / 2 / 4 tiles per wave), N a multiple of
  // 64 (no zero pad in the slab), and the layer must hand its outputs on through LDS (not the chain's last).
  int ns = 0;
  if (f3 && b && !sum && nb >= 2 && tune.mlp_nsplit >= 2 && tune.xbuf && tune.xcnt && !d_wait &&
      a.M <= tune.mlp_nsplit_rows && a.M <= tune.xbuf_rows && b->width[1] <= tune.xbuf_cols && !(b->width[1] & 63)) {
    for (int S = tune.mlp_nsplit >= 4 ? 4 : 2; S >= 2 && !ns; S >>= 1) {
      const int cw = b->width[1] / S;
      if (b->width[1] % S == 0 && (cw == 64 || cw == 128 || cw == 256)) ns = S;
    }
  }
  if (f3) {
    int st = 0;
    for (int l = 0; l < na; ++l) {
      int op = l == na - 1 ? a.width[l + 1] : pad64(a.width[l + 1]);
      if (l == na - 1 && b && sum) op = pad64(b->width[0]) - sum->cols;
      st += steps3(a.width[l], a.width[l + 1], op);
    }
    for (int l = 0; l < nb; ++l)
      st += l == 0 && ns ? (b->width[0] + 63) / 64
                         : steps3(b->width[l], b->width[l + 1], l == nb - 1 ? b->width[l + 1] : pad64(b->width[l + 1]));
    for (int l = 0; l < na; ++l) f3 = f3 && a.width[l + 1] <= 4080 && a.width[l] <= 4096;
    for (int l = 0; l < nb; ++l) f3 = f3 && b->width[l + 1] <= 4080 && b->width[l] <= 4096;
    f3 = f3 && st <= DRS_MAX_STREAM_TILES;
  }
  if (sum && !f3 && pad64(b->width[0]) - sum->cols > ((d_out + 127) / 128) * 128) return false;   // zero pad must fall in an existing pass
  const int nwv = 8;
  const int passw = 16 * nwv;
  p.packed = f3 ? 5 : pk ? 1 : 0;   // 1: stream_kernel on the packed twins | 5: stream4 (6: its 32-row form, set below)
  const int lpad = f3 ? 8 : 4;      // slab rows: 64 m + 8 floats apart in the b128 form, 64 m + 4 else
  // rows per workgroup: 16, or 32 for stream4_kernel's two-halves form ("mlp_rows32": launches of at
  // least that many rows, no summed input, slabs that still fit LDS)
  int SR = 16;
  if (f3 && f4 && !sum && tune.mlp_rows32 > 0 && a.M >= tune.mlp_rows32) {
    size_t fl = 32 * (size_t)(pad64(a.width[0]) + lpad);
    if (b) {
      const int rc = dot ? dot->F * dot->D : b->width[0];
      fl += 32 * (size_t)(pad64(rc) + lpad);
      if (dot) fl += 32 * (size_t)(pad64(b->width[0]) + lpad);
    }
    int w0 = 0, w1 = 0, wh = 0;
    auto note = [&](int n) { int& w = wh ? w1 : w0; w = pad64(n) > w ? pad64(n) : w; wh ^= 1; };
    for (int l = 0; l < na; ++l) if (!(l == na - 1)) note(a.width[l + 1]);
    for (int l = 0; l < nb; ++l) if (!(l == nb - 1)) note(b->width[l + 1]);
    const bool q_in_x0 = !b && w1 && w1 <= pad64(a.width[0]);     // (see the Q slab below)
    fl += (w0 ? 32 * (size_t)(w0 + lpad) : 0) + (w1 && !q_in_x0 ? 32 * (size_t)(w1 + lpad) : 0);
    for (int l = 0; l < na; ++l) fl += (a.width[l + 1] + 3) & ~3;
    for (int l = 0; l < nb; ++l) fl += (b->width[l + 1] + 3) & ~3;
    fl += 4 + 4 * DRS_MAX_STREAM_TILES + (sizeof(SLayer) / 4) * DRS_MAX_STREAM_LAYERS;
    if (sizeof(float) * fl <= kLdsBudget) SR = 32;
  }
  // LDS layout (floats): [sB 2x128x68 (LDS-staged form only)][X0][RS][P][Q][biases]
  int off = 0;
  p.sB_off = off; off += pk ? 0 : 2 * 128 * 68;
  const int x0_ld = pad64(a.width[0]) + lpad;
  const int x0_off = off; off += SR * x0_ld;
  // RS: what the first chain's last layer writes its dense_out slot into and the pooled rows
  // are pulled beside: the second chain's input (cat) or the interaction's T slab (dot)
  int rs_off = -1, rs_ld = 0, rs_cols = 0, ri_off = -1, ri_ld = 0;
  if (b) {
    rs_cols = dot ? dot->F * dot->D : b->width[0];
    rs_ld = pad64(rs_cols) + lpad; rs_off = off; off += SR * rs_ld;
    if (dot) { ri_ld = pad64(b->width[0]) + lpad; ri_off = off; off += SR * ri_ld; }
  }
  // ping-pong widths
  int wP = 0, wQ = 0;
  {
    int which = 0;   // next ping-pong slab to write: 0 = P, 1 = Q
    auto note = [&](int n) { int& w = which ? wQ : wP; w = pad64(n) > w ? pad64(n) : w; which ^= 1; };
    for (int l = 0; l < na; ++l) if (!(l == na - 1)) note(a.width[l + 1]);
    for (int l = 0; l < nb; ++l) if (!(l == nb - 1)) note(b->width[l + 1]);
  }
  const int p_ld = wP + lpad, q_ld = wQ + lpad;
  const int p_off = off; off += wP ? SR * p_ld : 0;
  // 32-row form, single chain: the input slab is dead once layer 0 has run (the barrier behind it), and Q
  // is first written by layer 1 -- Q lives in X0's space when it fits there (RM3's 416-512-256-1 top
  // chain: 164 KB -> 130 KB)
  const bool q_in_x0 = SR == 32 && !b && wQ && q_ld <= x0_ld;
  const int q_off = q_in_x0 ? x0_off : off; off += wQ && !q_in_x0 ? SR * q_ld : 0;
  const int bias_off = off;
  for (int l = 0; l < na; ++l) off += (a.width[l + 1] + 3) & ~3;
  for (int l = 0; l < nb; ++l) off += (b->width[l + 1] + 3) & ~3;
  off = (off + 3) & ~3;
  p.tab_off = off; off += pk ? 4 * DRS_MAX_STREAM_TILES : 0;
  p.lay_off = off; off += pk ? (int)(sizeof(SLayer) / 4) * DRS_MAX_STREAM_LAYERS : 0;
  if (sizeof(float) * (size_t)off > kLdsBudget) return false;
  *lds_bytes = sizeof(float) * (size_t)off;
  p.lds_floats = off;
  if (SR == 32) p.packed = 6;

  int which = 0, cur_off = x0_off, cur_ld = x0_ld, n = 0, tiles = 0, boff = bias_off;
  auto add = [&](const ChainArgs& c, int l, bool last_of_chain, bool last_of_all) {
    SLayer& L = p.L[n++];
    L.W = c.W[l]; L.w_off = (uint32_t)(c.W[l] - tune.w_arena);
    L.wp_off = L.w_off + (uint32_t)(((uint64_t)c.width[l] * c.width[l + 1] + 63) / 64 * 64);   // twin right behind W
    L.b = c.b[l]; L.K = c.width[l]; L.N = c.width[l + 1]; L.act = c.act[l];
    L.in_off = cur_off; L.in_ld = cur_ld;
    L.out_off = -1; L.out_ld = 0; L.out_pad = L.N; L.out_col0 = 0;
    L.g_out = nullptr; L.g_ld = 0; L.g_sc1 = 0;
    L.b_off = boff; boff += (L.N + 3) & ~3;
    if (last_of_chain) {
      L.g_out = c.y; L.g_ld = c.ldy;
      L.g_sc1 = last_of_all && publish;
      if (!last_of_all) {           // dense_out slot of the second chain's input slab
        L.out_off = rs_off; L.out_ld = rs_ld; L.out_pad = L.N;
        if (sum) { L.out_col0 = sum->cols; L.out_pad = pad64(b->width[0]) - sum->cols; }   // behind the summed block, zero tail
        cur_off = dot ? ri_off : rs_off; cur_ld = dot ? ri_ld : rs_ld;
      }
    } else {
      L.out_off = which ? q_off : p_off; L.out_ld = which ? q_ld : p_ld; L.out_pad = pad64(L.N);
      cur_off = L.out_off; cur_ld = L.out_ld;
      which ^= 1;
    }
    tiles += ((L.N + passw - 1) / passw) * ((L.K + 63) / 64);
  };
  for (int l = 0; l < na; ++l) add(a, l, l == na - 1, l == na - 1 && !b);
  for (int l = 0; l < nb; ++l) add(*b, l, l == nb - 1, l == nb - 1);
  p.n_layers = n;
  p.n_tiles = tiles;
  p.n_table = 0;
  p.wait_tile = -1; p.ns = 0;
  if (f3) {
    // one descriptor per STEP of stream3_kernel: (layer, pass of nw3 x TPW tiles, 64-k chunk).
    // wp_off: chunk c of the twin's first 128-column pass; in_ld: floats between two such passes;
    // a_off: low half = LDS offset of (row 0, k = 64 c) of the input slab, high half = its leading dimension
    int ti = 0, inter_at = -1;
    if (dot) inter_at = 0;
    for (int l = 0; l < n; ++l) {
      const SLayer& L = p.L[l];
      const int nch = (L.K + 63) / 64, ntl = (L.N + 15) / 16;
      const int opad = L.out_off >= 0 && L.out_pad > L.N ? L.out_pad : L.N;
      const bool split = ns && l == na;      // this launch's split layer: the table names slice 0's tiles (one pass)
      const int etl = (opad + 15) / 16, tpw = split ? L.N / ns / 64 : tpw3(L.N, opad);
      const int tpp = nw3 * tpw, npass = split ? 1 : (etl + tpp - 1) / tpp;
      if (dot && l < na) inter_at += npass * nch;
      if (b && !sum && l == na) p.wait_tile = ti;
      if (split) {
        p.ns = ns;
        if (nsp) { nsp->t0 = ti; nsp->t1 = ti + nch; nsp->tps = tpp; nsp->n = L.N; nsp->off = L.out_off; nsp->ld = L.out_ld;
                   nsp->xbuf = tune.xbuf; nsp->xcnt = tune.xcnt; }
      }
      for (int ps = 0; ps < npass; ++ps)
        for (int c = 0; c < nch; ++c) {
          STile& t = p.tiles[ti];
          t.wp_off = L.wp_off + (uint32_t)c * 8192u;
          t.a_off = (L.in_off + c * 64) | (L.in_ld << 16);
          t.in_ld = nch * 8192;
          const bool last_of_layer = c == nch - 1 && ps == npass - 1;
          t.info = (ps * tpp) | (ntl << 8) | (c == nch - 1 ? S3_LAST : 0) | (last_of_layer ? S3_BARRIER : 0) |
                   (last_of_layer ? 0 : S3_ANEXT) | (tpw << S3_TPW_SHIFT) | (c == 0 ? S3_FIRST : 0) | (l << 24);
          ++ti;
        }
    }
    if (inter_at >= 0 && inter_at < ti) p.tiles[inter_at].info |= S3_INTERACT;
    p.n_table = ti;
    p.n_tiles = ti;
    // the arena range that holds the packed twins of this launch's layers (L2 warm-up)
    uint64_t lo = ~0ull, hi = 0;
    for (int l = 0; l < n; ++l) {
      const uint64_t b = p.L[l].wp_off, e = b + (uint64_t)stream_packed_floats(p.L[l].K, p.L[l].N);
      lo = b < lo ? b : lo; hi = e > hi ? e : hi;
    }
    lo &= ~1023ull;                                          // 4-KB granules
    hi = (hi + 1023) & ~1023ull;
    if (hi > tune.w_arena_floats) hi = tune.w_arena_floats & ~1023ull;
    if (hi < lo + 1024) { lo = 0; hi = 1024; }
    p.warm_off = (uint32_t)lo;
    p.warm_bytes = (int32_t)((hi - lo) * 4);
  } else if (pk && nwv == 8 && tiles <= DRS_MAX_STREAM_TILES) {
    int ti = 0, inter_at = -1;
    if (dot) {
      inter_at = 0;
      for (int l = 0; l < na; ++l) inter_at += ((a.width[l + 1] + 127) / 128) * ((a.width[l] + 63) / 64);
    }
    for (int l = 0; l < n; ++l) {
      const SLayer& L = p.L[l];
      const int nch = (L.K + 63) / 64, npass = (L.N + 127) / 128;
      for (int ps = 0; ps < npass; ++ps)
        for (int c = 0; c < nch; ++c) {
          STile& t = p.tiles[ti];
          t.wp_off = L.wp_off + (uint32_t)(ps * nch + c) * 8192u;
          t.a_off = L.in_off + c * 64;
          t.in_ld = L.in_ld;
          const int ncols = L.N - ps * 128;
          t.info = (ncols > 0xffff ? 0xffff : ncols) | (c == nch - 1 ? 1 << 16 : 0) |
                   (c == nch - 1 && ps == npass - 1 ? 1 << 17 : 0) | (ti == inter_at ? 1 << 18 : 0) | (l << 24);
          ++ti;
        }
    }
    p.n_table = ti;
  }
  p.n_bias = boff - bias_off;
  p.bias_off = bias_off;
  p.bias = a.b[0];
  p.M = a.M;
  p.zero = tune.zero;
  p.wbase = tune.w_arena;
  p.zero_off = tune.w_zero_off;
  p.dbg = tune.mlp_debug;
  SInput& i0 = p.in[0];
  i0.src = a.x; i0.ld = a.ldx; i0.col0 = 0; i0.cols = a.width[0]; i0.cols_pad = pad64(a.width[0]);
  i0.lds_off = x0_off; i0.lds_ld = x0_ld; i0.lds_col0 = 0; i0.use_xs = xs.q.n_q > 0;
  p.in[0].col2 = p.in[1].col2 = -1;
  p.n_inputs = 1;
  if (b) {
    SInput& i1 = p.in[1];
    i1.src = dot ? dot->T : b->x; i1.ld = dot ? dot->ldt : b->ldx; i1.col0 = d_out; i1.cols = rs_cols - d_out;
    i1.cols_pad = pad64(rs_cols) - d_out;
    i1.lds_off = rs_off; i1.lds_ld = rs_ld; i1.lds_col0 = d_out; i1.use_xs = 0;
    if (sum) {
      i1.src = sum->src; i1.ld = sum->ld; i1.col0 = sum->col_a; i1.col2 = sum->col_b;
      i1.cols = i1.cols_pad = sum->cols; i1.lds_col0 = 0;
      i1.g_dst = sum->dst; i1.g_ldd = sum->ldd;
    }
    p.n_inputs = 2;
  }
  if (dot) {
    p.inter_on = 1; p.F = dot->F; p.D = dot->D; p.itself = dot->itself ? 1 : 0; p.P = dotP;
    p.t_off = rs_off; p.t_ld = rs_ld; p.r_off = ri_off; p.r_ld = ri_ld; p.r_pad = pad64(b->width[0]);
    p.g_R = dot->R; p.g_ldr = dot->ldr;
    p.inter_tile = 0;
    for (int l = 0; l < na; ++l) p.inter_tile += ((a.width[l + 1] + passw - 1) / passw) * ((a.width[l] + 63) / 64);
  }
  return true;
}

bool stream_applicable(const ChainArgs& a, const ChainArgs& b, const Tune& tune, const XSrc* xsrc,
                       const DotArgs* dot, const SumArgs* sum, bool* can_defer) {
  if (!tune.mlp_stream || !tune.zero) return false;
  XSrc xs;
  memset(&xs, 0, sizeof xs);
  if (xsrc) xs = *xsrc;
  SArgs sp;
  size_t lds = 0;
  const bool ok = stream_plan(a, &b, tune, xs, true, &sp, &lds, dot, sum);
  // (the 16-row one-workgroup-per-CU form only: the 2cu / 32-row builds have no registers to spare for the late fetch)
  if (can_defer) *can_defer = ok && sp.packed == 5 && sp.wait_tile > 0 && !(tune.mlp_stream == 4 && tune.mlp_stream_2cu);
  return ok;
}

hipError_t launch_chain2(const ChainArgs& a, const ChainArgs* b, const Tune& tune, hipStream_t s,
                         const Done* done, const XSrc* xsrc, const DotArgs* dot, const SumArgs* sum) {
  if (a.M <= 0) return hipSuccess;
  Done d;
  memset(&d, 0, sizeof d);
  if (done) d = *done;
  XSrc xs;
  memset(&xs, 0, sizeof xs);
  if (xsrc) xs = *xsrc;
  if (a.n_layers < 1 || a.n_layers > DRS_MAX_CHAIN || (b && (b->n_layers < 1 || b->n_layers > DRS_MAX_CHAIN)))
    return hipErrorInvalidValue;
  if (tune.mlp_stream && tune.zero) {
    SArgs sp;
    size_t slds = 0;
    NSplit nsp;
    if (stream_plan(a, b, tune, xs, d.counter != nullptr, &sp, &slds, dot, sum, d.wait_flag != nullptr, &nsp)) {
#ifdef DRS_TIMELINE
      slds += 8192;
#endif
      const dim3 g3((unsigned)((a.M + 15) / 16));
      {
        // which form serves this launch (drs_last_dispatch; DESIGN.md dispatch table)
        const char* form = sp.ns ? (sp.packed == 6 ? (sp.ns == 4 ? "stream4_kernel<rows32,nsplit4>" : "stream4_kernel<rows32,nsplit2>")
                                                    : (sp.ns == 4 ? "stream4_kernel<nsplit4>" : "stream4_kernel<nsplit2>")) :
            sp.packed == 6 ? "stream4_kernel<rows32>" :
            sp.packed == 5 ? (sp.in[1].col2 >= 0 ? "stream4_kernel<sum>" : (tune.mlp_stream == 4 && tune.mlp_stream_2cu) ? "stream4_kernel<2cu>" : "stream4_kernel") :
            sp.packed ? ((tune.mlp_stream_2cu && sp.n_table > 0) ? "stream_kernel<packed,2cu>" : "stream_kernel<packed>") : "stream_kernel<lds>";
        log_launch(tune.log, "%s[%u wg, %d layers%s, %zu B lds]", form, (sp.packed == 6 ? (unsigned)((a.M + 31) / 32) : g3.x) * (sp.ns ? sp.ns : 1),
                   sp.n_layers, dot ? ", dot" : "", slds);
      }
      if (d.wait_flag) {   // early start: only the plain stream4_kernel form has the late fetch (callers ask stream_applicable)
        const bool plain = sp.packed == 5 && sp.in[1].col2 < 0 && !(tune.mlp_stream == 4 && tune.mlp_stream_2cu) && sp.wait_tile > 0;
        if (!plain) return hipErrorInvalidValue;
      }
      if (sp.packed >= 5) {
        const bool rows32 = sp.packed == 6;
        const unsigned grid = (rows32 ? (unsigned)((a.M + 31) / 32) : g3.x) * (sp.ns ? (unsigned)sp.ns : 1u);
        return launch_stream4(sp.in[1].col2 >= 0, tune.mlp_stream == 4 && tune.mlp_stream_2cu, rows32 ? 32 : 16, sp.ns != 0, grid, slds, s,
                              sp, d, xs, nsp);
      }
      return launch_stream8(!sp.packed ? 0 : (tune.mlp_stream_2cu && sp.n_table > 0) ? 2 : 1, g3.x, slds, s, sp, d, xs);
    }
  }
  if (dot || sum || d.wait_flag) return hipErrorInvalidValue;   // only the stream kernel has these joins (callers check stream_applicable)
  int kc = 64, nbuf = 2, lda = 0;
  size_t lds = 0;
  if (!chain_plan(a, b, tune, &kc, &nbuf, &lds, &lda)) return hipErrorInvalidValue;
#ifdef DRS_TIMELINE
  lds += 8192;
#endif
  bool vec = aligned16(a.x) && (a.ldx & 3) == 0;
  for (int l = 0; l < a.n_layers; ++l) vec = vec && aligned16(a.W[l]) && (a.width[l] & 3) == 0;
  if (b) {
    vec = vec && aligned16(b->x) && (b->ldx & 3) == 0;
    for (int l = 0; l < b->n_layers; ++l) vec = vec && aligned16(b->W[l]) && (b->width[l] & 3) == 0;
  }
  for (int i = 0; i < xs.q.n_q; ++i) vec = vec && aligned16(xs.x[i]);
  ChainArgs second;
  memset(&second, 0, sizeof second);
  if (b) second = *b;
  const dim3 grid((unsigned)((a.M + 15) / 16));
  const int sld = chain_slab_ld2(a, b);
  log_launch(tune.log, "chain_kernel<%s,%d>[%u wg, %d layers]", vec ? "vec" : "scalar", kc, grid.x, a.n_layers + (b ? b->n_layers : 0));
#define LAUNCH(KC_)                                                                               \
  if (kc == KC_) {                                                                                \
    if (vec)                                                                                      \
      hipLaunchKernelGGL((chain_kernel<true, KC_>), grid, dim3(kThreads), lds, s, a, second, sld, nbuf, lda, d, xs);  \
    else                                                                                          \
      hipLaunchKernelGGL((chain_kernel<false, KC_>), grid, dim3(kThreads), lds, s, a, second, sld, nbuf, lda, d, xs); \
  }
  DRS_FOR_EACH_KC(LAUNCH)
#undef LAUNCH
  return hipGetLastError();
}

hipError_t launch_chain(const ChainArgs& a, const Tune& tune, hipStream_t s, const Done* done,
                        const XSrc* xsrc) {
  return launch_chain2(a, nullptr, tune, s, done, xsrc, nullptr, nullptr);
}

#ifdef DRS_TIMELINE
// (every MLP translation unit keeps its own stamp buffer: whichever kernel ran last has stamps to hand over)
extern "C" int drs_debug_timeline(unsigned long long* out, int cap, int reset) {
  int n = tl_fetch_stream4(out, cap, reset);
  if (n == 0) n = tl_fetch_stream8(out, cap, reset);
  if (n == 0) n = tl_fetch_here(out, cap, reset);
  return n;
}
#endif

hipError_t launch_interact_dot(const float* T, int64_t ldt, int64_t B, int32_t F, int32_t D,
                               int32_t itself, float* R, int64_t ldr, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  const int Fp = (F + 15) & ~15;
  const size_t lds = sizeof(float) * 4 * (size_t)Fp * (D + 1);
  if (lds > 160 * 1024) return hipErrorInvalidValue;   // (attribute: mlp_set_attrs, per device)
  hipLaunchKernelGGL(interact_dot_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, s, T, ldt,
                     B, F, D, itself, R, ldr);
  return hipGetLastError();
}

static unsigned ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

hipError_t launch_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                           int64_t ldo, int64_t M, int32_t D, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(add_rows_kernel, dim3(ew_grid(M * D)), dim3(256), 0, s, a, lda, b, ldb, out,
                     ldo, M, D);
  return hipGetLastError();
}

hipError_t launch_copy_rows(const float* a, int64_t lda, float* out, int64_t ldo, int64_t M,
                            int32_t D, hipStream_t s) {
  return launch_add_rows(a, lda, nullptr, 0, out, ldo, M, D, s);
}

hipError_t launch_copy_rows_multi(const XSrc& xs, int32_t m_den, float* out, int64_t ldo, hipStream_t s) {
  const int64_t Mv = xs.q.n_q > 0 ? xs.q.vstart[xs.q.n_q] : 0;
  if (Mv <= 0 || m_den <= 0) return hipSuccess;
  bool vec = !(m_den & 3) && !(ldo & 3) && aligned16(out);
  for (int i = 0; i < xs.q.n_q; ++i) vec = vec && aligned16(xs.x[i]);
  if (vec) hipLaunchKernelGGL(copy_rows_multi_kernel<4>, dim3(ew_grid(Mv * (m_den / 4))), dim3(256), 0, s, xs, m_den, out, ldo);
  else hipLaunchKernelGGL(copy_rows_multi_kernel<1>, dim3(ew_grid(Mv * m_den)), dim3(256), 0, s, xs, m_den, out, ldo);
  return hipGetLastError();
}

}  // namespace drs

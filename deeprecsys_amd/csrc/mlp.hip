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

#include "drs_internal.h"
#include "mlp_dev.h"

namespace drs {
namespace {

// Optional in-kernel timeline (tools/mlp_timeline.py, built with -DDRS_TIMELINE into a
// separate library): wave 0 of workgroup 0 stamps the shader clock at the phase
// boundaries of every K-chunk round.  Compiled out of the product build.
#ifdef DRS_TIMELINE
__device__ unsigned long long g_tl[16384];
__device__ unsigned g_tl_n;
// stamps go to a spare 8 KB at the very end of the dynamic LDS (no global traffic while the
// kernel runs); thread 0 of workgroup 0 flushes them at the end
#define TL_SLOTS 1000
__device__ __forceinline__ void tl_stamp(unsigned long long* tl, unsigned tag, bool on) {
  if (on && threadIdx.x == 0) {
    const unsigned i = (unsigned)tl[0];
    if (i + 1 < TL_SLOTS) {
      tl[i + 1] = ((unsigned long long)tag << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull);
      tl[0] = i + 1;
    }
  }
}
#define TL_ON (blockIdx.x == 0 && blockIdx.y == 0)      // (stream4_kernel: every workgroup of slab 0 stamps, the one that signs off flushes)
#define TL(tag) tl_stamp(g_tl_lds, tag, TL_ON)
#define TL_DECL unsigned long long* g_tl_lds
#define TL_ARG , g_tl_lds
#define TL_PARAM , unsigned long long* g_tl_lds
#else
#define TL(tag)
#define TL_ARG
#define TL_PARAM
#endif

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

// ---------------------------------------------------------------------------
// Stream kernel: the same chain(s) of layers as chain_kernel, organised around ONE flat
// stream of weight tiles instead of per-layer passes.
//
// chain_kernel's K-chunk round is fetch -> MFMA -> stash -> barrier with every phase
// exposed (in-kernel timeline: ~1.0 k cycles of MFMA in a 2.2 k cycle round) and every
// pass of every layer starts with a cold fetch (6 x ~2.3 k cycles on RM1).  Weights do not
// depend on activations, so here the tiles W[n0:n0+128, c*64:(c+1)*64] of ALL layers form
// one sequence that is requested SIX tiles ahead of its use, across pass and layer
// boundaries (a ring of six register sets per thread, 16 VGPRs each):
//     round i:  issue global loads of tile i+6          (register set i%6)
//               MFMAs of tile i from LDS buffer i&1, interleaved with
//               the LDS stash of tile i+1 (set (i+1)%6 -> buffer (i+1)&1)
//               [epilogue of the pass: bias + activation -> next layer's LDS slab]
//               barrier
// so a round is bounded by the MFMA pipe (16 dependent MFMAs x 2 waves per SIMD), the
// loads have five rounds to land and the only cold start is the kernel's first tile.
// All layer inputs live in LDS slabs: the chains' global inputs (dense features; the
// pooled-embedding columns of the interaction buffer) are pulled in once at kernel start,
// every later activation is written there by the previous layer's epilogue.  Slab columns
// between K and the next multiple of 64 are kept zero, weight tiles read zeros beyond K,
// so the MFMA body has no selects and no branches.
// With a DotArgs the DLRM dot interaction runs between the two chains, in LDS (interact()).
// Requires K % 4 == 0 and 16-B aligned operands on every layer and the slabs to fit in
// LDS; launch_chain2 falls back to chain_kernel otherwise.
// The pairwise dots of the fused dot interaction on the matrix cores (north_star: "the feature-interaction
// batched dot ... use MFMA"): for one sample Z = T T^t with T the sample's F x D feature block; a wave
// takes a sample, lane (r, g) feeds T[r][4 s + g] as BOTH operands of MFMA step s (rows r >= F feed
// zeros), D / 4 dependent steps = one k-ordered fma chain per pair from 0, like the oracle's and like
// interact_dot_kernel's.  Lane (r, g) then holds Z[4 g + i][r], i = 0..3, and writes the pairs of the
// (strictly) lower triangle in BatchGather order behind the D dense columns.  pos(c, row) maps a column
// of a slab row to its LDS position (the kernels keep different column permutations).  The accumulator
// is a VGPR quad through inline asm: stream4_kernel must not have the compiler allocate AGPRs.
template <typename POS>
__device__ __forceinline__ void interact_pairs_mfma(const float* Ts, int t_ld, float* Rs, int r_ld, int rows, int F,
                                                    int D, int itself, float* g_R, int64_t g_ldr, int64_t m0,
                                                    int64_t M, int n_waves, int wave, int lane, POS pos) {
  const int r = lane & 15, g = lane >> 4, off = itself ? 1 : 0;
  const int nblk = (F + 15) >> 4;       // F > 16 (RM2 in dot mode: 33 features): Z in 16 x 16 blocks, lower triangle of blocks
  for (int row = wave; row < rows; row += n_waves) {
    const float* t = Ts + row * t_ld;
    for (int bi = 0; bi < nblk; ++bi)
      for (int bj = 0; bj <= bi; ++bj) {
        // A operand: features 16 bi + r (output rows), B operand: features 16 bj + r (output columns)
        const int fa = 16 * bi + r, fb = 16 * bj + r;
        const int base_a = (fa < F ? fa : 0) * D + g, base_b = (fb < F ? fb : 0) * D + g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < D; k0 += 4) {
          float va = t[pos(base_a + k0, row)], vb = t[pos(base_b + k0, row)];
          va = fa < F ? va : 0.f;
          vb = fb < F ? vb : 0.f;
          asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(va), "v"(vb));   // (s_nop: the operands were just written by VALU ops the compiler cannot see the consumer of)
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));      // the last step's results
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int fi = 16 * bi + 4 * g + i, fj = fb;
          if (fi < F && fj < F && (fj < fi || (off && fj == fi))) {
            const int c = D + fi * (fi - 1 + 2 * off) / 2 + fj;
            const float v = acc[i];
            Rs[row * r_ld + pos(c, row)] = v;
            if (g_R && m0 + row < M) g_R[(m0 + row) * g_ldr + c] = v;
          }
        }
      }
  }
}

struct SLayer {
  const float* W;          // [N, K] row-major
  uint32_t w_off, wp_off;  // ... as a float offset from SArgs::wbase (the engine's weight arena);
                           // wp_off: the packed twin (stream_kernel<true>)
  const float* b;          // [N] or nullptr
  int32_t K, N, act;
  int32_t in_off, in_ld;   // input slab: float offset in LDS, leading dimension
  int32_t out_off, out_ld; // output slab (out_off < 0: none)
  int32_t out_pad;         // columns [N, out_pad) of the output slab are zero filled
  int32_t out_col0;        // first column of this layer's outputs inside the output slab
  int32_t b_off;           // LDS copy of the bias (zeros when b == nullptr), N floats
  float* g_out;            // global output or nullptr
  int64_t g_ld;
  int32_t g_sc1;           // write-through stores (final outputs handed over by signal_done)
};
struct SInput {            // 16 x cols block of a global matrix -> LDS slab, zero padded to cols_pad
  const float* src;
  int64_t ld;
  int32_t col0, cols, cols_pad;
  int32_t lds_off, lds_ld, lds_col0;
  int32_t use_xs;
  int32_t col2;            // >= 0: the block is src[:, col0..] + src[:, col2..] (NCF's Sum)
  float* g_dst;            // also store the block to global (row-major, ld g_ldd) or nullptr
  int64_t g_ldd;
};
#define DRS_MAX_STREAM_LAYERS (2 * DRS_MAX_CHAIN)
#define DRS_MAX_STREAM_TILES 96
// A round of the packed stream kernel, precomputed by the host (stream_plan): which packed tile,
// where the activation operands sit, how many of the pass's columns exist, what happens after it.
// The iterator form keeps ~40 scalars of layer / pass / chunk state alive across six unrolled
// rounds -- they did not fit the SGPR file: the compiled round re-read kernel arguments and
// shuffled 120 spilled scalars through VGPR lanes, and the bare control flow of RMC1's 26 rounds
// (MFMAs, loads, LDS reads and barriers removed) took 10.7 of the launch's 34 us.
struct STile {
  uint32_t wp_off;         // packed tile (wave 0's slice) as a float offset from SArgs::wbase
  int32_t a_off;           // activation operands: LDS float offset of (row 0, k = 64 c) in the layer's input slab
  int32_t in_ld;           // ... and the slab's leading dimension
  int32_t info;            // bits 0..15: columns of this pass that exist (N - n0, capped); 16: last chunk of the
                           // pass (epilogue); 17: last round of the layer (barrier); 18: the dot interaction
                           // runs before this round; 24..31: layer index
};
struct SArgs {
  int32_t n_layers, n_tiles, sB_off, n_inputs;
  int32_t dbg, lds_floats;
  int32_t wait_tile, ns;      // wait_tile: first step of the second chain (where a launch with Done::wait_flag waits
                              // for the gather and fetches its second input), -1: the form has no such point
                              // ns: column slices of the split layer (stream4_kernel<..., SPL>; 0: none), see below
  // dot interaction between the chains (DotArgs): at tile `inter_tile` the T slab (F x D per
  // row) becomes the R slab (D + P per row, zero padded to r_pad) the second chain reads
  int32_t inter_on, inter_tile, F, D, itself, P;
  int32_t t_off, t_ld, r_off, r_ld, r_pad, packed;
  float* g_R;
  int64_t g_ldr;
  int32_t n_bias, bias_off; // all biases: n_bias floats at `bias` -> LDS float offset bias_off
  const float* bias;
  int64_t M;
  const float* zero;       // 16 B of zeros in device memory: source of every out-of-range float4 load
                           // (an address select keeps the load unconditional; a value select would
                           // put it under divergent control flow and serialise the tile's loads)
  const float* wbase;      // the weight arena: every tile address is wbase + a 32-bit float offset, so
  uint32_t zero_off, warm_off;// a tile load is `global_load_dwordx4 v, v_off, s[wbase]` (four VALU per
                           // address); zero_off: zeros INSIDE the arena for k beyond a layer's K
  SLayer L[DRS_MAX_STREAM_LAYERS];
  SInput in[2];
  // packed form, 8 waves: one descriptor per round, read with ONE scalar load (n_table == n_tiles
  // when the launch has at most DRS_MAX_STREAM_TILES rounds, else 0: the iterator form below)
  int32_t n_table, warm_bytes;   // warm_off / warm_bytes: the arena range holding this launch's packed twins (stream3_kernel's L2 warm-up)
  int32_t tab_off, lay_off;   // LDS float offsets of the copies of tiles[] and L[] the loop reads
  STile tiles[DRS_MAX_STREAM_TILES];
};
// Column-split form of stream4_kernel (SArgs::ns = 2 | 4): `ns` workgroups share a slab of rows.  Each runs everything up
// to the split layer (steps [t0, t1) of the table: the first layer of the second chain, RMC1's 576 -> 256) for ALL of the
// slab's rows, but only `tps` of that layer's column tiles (tiles tps y ...: one pass, 4 waves x tps / 4 tiles); it
// publishes its [rows, 16 tps] piece of the layer's output slab (LDS offset `off`, leading dimension `ld`, `n` columns)
// write-through in xbuf and takes a ticket on xcnt[slab]; the last arriver fetches the other pieces and runs the
// remaining layers.  The split is over N: every output keeps its k-ordered chain -- same bits.
// A kernel argument of its own BEHIND the others: grown into SArgs, it moved tiles[], Done and XSrc inside the argument
// block, the compiler cut its scalar loads differently and the 32-row build -- 106 SGPRs, 17 more in VGPR lanes -- came
// out with a (never accessed) 36-byte private segment, i.e. a launch that needs scratch set up.
struct NSplit {
  int32_t t0, t1, tps, n, off, ld;
  float* xbuf;                // [launch rows, n] in the slab's column order
  uint32_t* xcnt;             // [slabs] arrival tickets, zero between launches
};


// PK = true ("mlp_stream" 2, the default): the weight tiles come from the layers' PACKED twins
// (pack_stream_kernel below: per pass, chunk and wave, four float4 per lane = the wave's 16 MFMA
// B operands of the round, k in natural order) straight into the registers the MFMAs read --
// no LDS staging of W, no stash, and a workgroup barrier only where one layer's outputs become
// the next layer's inputs (RMC1: 5 barriers instead of 26) instead of one per 64-k chunk.
// PK = false: the LDS-staged form described above.  Same fma chains, same bits.
// NWV = waves per workgroup: 8 (a pass covers 128 output columns).
static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) <= 4096, "kernel arguments: 4 KB");
// The argument block of the stream kernels is 3.1 KB = 50 cache lines that the host rewrites for every
// launch: each first touch is a miss of the scalar cache all the way to memory, and the compiler fetches
// fields where it first needs them -- the prologue of stream3_kernel spent 10 k cycles (4 us) in ~40
// dependent s_load / s_waitcnt pairs before its first input load (in-kernel timeline, round 3).  One
// burst of loads, one per line, all in flight together, brings the whole block into the scalar cache
// for the price of ONE miss.
__device__ __forceinline__ void kernarg_burst() {
  static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) >= 0xc40 + 4, "argument block shorter than the burst");
  const uint32_t* kp_ = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t0_, t1_;
#define S3_KL(O0, O1) "s_load_dword %0, %2, " #O0 "\n\ts_load_dword %1, %2, " #O1 "\n\t"
  asm volatile(
      S3_KL(0x0, 0x40) S3_KL(0x80, 0xc0) S3_KL(0x100, 0x140) S3_KL(0x180, 0x1c0) S3_KL(0x200, 0x240)
      S3_KL(0x280, 0x2c0) S3_KL(0x300, 0x340) S3_KL(0x380, 0x3c0) S3_KL(0x400, 0x440) S3_KL(0x480, 0x4c0)
      S3_KL(0x500, 0x540) S3_KL(0x580, 0x5c0) S3_KL(0x600, 0x640) S3_KL(0x680, 0x6c0) S3_KL(0x700, 0x740)
      S3_KL(0x780, 0x7c0) S3_KL(0x800, 0x840) S3_KL(0x880, 0x8c0) S3_KL(0x900, 0x940) S3_KL(0x980, 0x9c0)
      S3_KL(0xa00, 0xa40) S3_KL(0xa80, 0xac0) S3_KL(0xb00, 0xb40) S3_KL(0xb80, 0xbc0) S3_KL(0xc00, 0xc40)
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0_), "=&s"(t1_) : "s"(kp_));
#undef S3_KL
}

// RD3: the table-driven packed form with a ring of THREE register sets instead of six, compiled for 128
// VGPRs (four waves per SIMD): two of its workgroups share a CU, so the launches of two overlapping
// sets (MLP-bound models run one MLP stream per slot) interleave on the same SIMDs instead of
// queueing for whole CUs -- what gemm_kernel<2, 1, 2, 4> does for the wide layers.
template <bool PK, int NWV, bool RD3 = false>
__global__ __launch_bounds__(64 * NWV, RD3 ? 4 : 1) void stream_kernel(SArgs a, Done done, XSrc xs) {
  static_assert(NWV == 8, "eight waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kernarg_burst();
  constexpr int kThreads = 64 * NWV;           // (shadows the file-scope 512)
  constexpr int PASSW = 16 * NWV;              // output columns per pass
  constexpr int RD = RD3 ? 3 : 6;   // ring depth (register sets of weight tiles in flight)
  static_assert(!RD3 || (PK && NWV == 8), "the 3-deep ring: table-driven packed form only");
  constexpr int LD = 68;                       // staged W rows: 64 k + 4 pad
  const int tid = threadIdx.x;
  // `wave` as a SCALAR: everything derived from it (the wave's columns, "is my tile inside N",
  // the wave's slice of a packed tile) then runs on the scalar unit.  The per-round bookkeeping
  // was ~150 VALU instructions per wave (21 of them 32-bit multiplies), which two waves per SIMD
  // issue back to back: with MFMAs and weight loads removed the launch still took 25.6 of 34 us.
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int gs = swz(g, r);
  const uint32_t lane16 = (uint32_t)lane * 16u;   // byte offset of this lane's float4 inside a 1-KB operand block
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  float* sB = smem + a.sB_off;
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(smem + a.lds_floats);
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
#endif
  TL(1);
  const float* zero = a.zero;
  // staging role of this thread: row frow (+32 j) of the tile, floats fk..fk+3 of the chunk
  const int frow = tid >> 4, fk = (tid & 15) * 4;
  const int st_lo = (frow & 8) ? 2 : 0, st_hi = 2 - st_lo;   // swz4 of my rows (same for all j)
  float* const st_base = sB + frow * LD + fk;

  // ---- fetch iterator: six tiles ahead ----------------------------------------------------
  int f_l = 0, f_n0 = 0, f_c = 0, f_K = a.L[0].K, f_N = a.L[0].N;
  uint32_t f_woff = PK ? a.L[0].wp_off : a.L[0].w_off;
  // The tile loads are issued through inline asm and waited for with an explicit
  // s_waitcnt (DRS_WAIT_TILE): the compiler's own counter model drains the whole ring at
  // the loop header (vmcnt(0) once per trip), which costs a full miss latency every six
  // rounds.  vmcnt retires in order, so waiting for "at most 20 newer" is exact for the
  // set requested five rounds ago no matter how many stores came in between.
  // One of the four loads of a tile (rows frow + 32 j): scalar base + 32-bit offset.
  auto fetch_part = [&](f32x4 (&rb)[4], int j) {
    if (PK) {
      // tile (pass f_n0 / 128, chunk f_c) of the packed twin: 8192 floats; wave w's block of
      // 1024, float4 j of lane `lane` (always in range: the twin is padded with zeros)
      // (16 waves: waves 8..15 take the next 128-column pass of the twin, or -- beyond the layer's
      // last one, their columns do not exist -- re-read this one: the loads must be issued anyway)
      int p128 = (f_n0 >> 7) + (wave >> 3);
      p128 = p128 * 128 < f_N ? p128 : (f_n0 >> 7);
      const uint32_t tile = (uint32_t)p128 * (uint32_t)((f_K + 63) >> 6) + (uint32_t)f_c;
      // scalar base of the wave's 4-KB slice, constant per-lane offset, float4 j as the immediate
      const float* sb = a.wbase + (f_woff + tile * 8192u + (uint32_t)(wave & 7) * 1024u);
      if (j == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[0]) : "v"(lane16), "s"(sb));
      else if (j == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(rb[1]) : "v"(lane16), "s"(sb));
      else if (j == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(rb[2]) : "v"(lane16), "s"(sb));
      else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(rb[3]) : "v"(lane16), "s"(sb));
      return;
    }
    const int k = f_c * 64 + fk;
    const int row = min(f_n0 + frow + 32 * j, f_N - 1);
    uint32_t off = f_woff + (uint32_t)row * (uint32_t)f_K + (uint32_t)k;
    off = k < f_K ? off : a.zero_off;            // out-of-range k reads zeros
    const uint32_t boff = off << 2;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[j]) : "v"(boff), "s"(a.wbase));
  };
  auto fetch_advance = [&]() {                   // (uniform)
    ++f_c;
    if (f_c * 64 >= f_K) {
      f_c = 0;
      f_n0 += PASSW;
      if (f_n0 >= f_N) {
        f_n0 = 0;
        if (f_l + 1 < a.n_layers) {
          ++f_l;
          f_K = a.L[f_l].K; f_N = a.L[f_l].N; f_woff = PK ? a.L[f_l].wp_off : a.L[f_l].w_off;
        }
      }
    }
  };
  auto fetch = [&](f32x4 (&rb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch_part(rb, j);
    fetch_advance();
  };
  // swz4 by address instead of by value: the halves of a float4 go to swapped 8-B slots
  // on rows 8..15 (two ds_write_b64, no selects)
  auto stash = [&](int buf, const f32x4 (&rb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* q = st_base + (buf * 128 + 32 * j) * LD;
      *reinterpret_cast<float2*>(q + st_lo) = make_float2(rb[j][0], rb[j][1]);
      *reinterpret_cast<float2*>(q + st_hi) = make_float2(rb[j][2], rb[j][3]);
    }
  };
#define DRS_WAIT_TILE(RB, N) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]))

  auto stash_part = [&](int buf, const f32x4 (&rb)[4], int q) {
    float* p = st_base + (buf * 128 + 32 * (q >> 1)) * LD;
    if (q & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(rb[q >> 1][2], rb[q >> 1][3]);
    else *reinterpret_cast<float2*>(p + st_lo) = make_float2(rb[q >> 1][0], rb[q >> 1][1]);
  };

  // ring of 6 register sets: tile i+6 is requested in round i and stashed in round i+5, so a
  // weight tile has five rounds to arrive (the gather of the next launch set runs beside this
  // kernel and pushes L2 misses to several microseconds)
  f32x4 rb0[4], rb1[4], rb2[4], rb3[4], rb4[4], rb5[4];
  // (table form: the tile's packed offset comes from its descriptor)
  const bool use_table = RD3 || (PK && NWV == 8 && a.n_table > 0);          // uniform
  // The round descriptors and the layer records are COPIED from the kernel-argument segment into
  // LDS by the prologue and read from there: a scalar load of a kernel argument the wave has not
  // touched yet is a cold miss all the way to HBM (the segment is written by the host for every
  // launch), and the loop touched a new 64-B line of it every few rounds -- the bare control flow of
  // RMC1's 26 rounds cost 10 us of a 34 us launch that way (0.4 us per round with every MFMA, load,
  // LDS read and barrier removed; with the arguments in HOST memory, HIP_FORCE_DEV_KERNARG=0, 49 us).
  const int n_table = a.n_table;
  const uint32_t* s_tab = reinterpret_cast<const uint32_t*>(smem + a.tab_off);
  const uint32_t* s_lay = reinterpret_cast<const uint32_t*>(smem + a.lay_off);
  auto lds_tile = [&](int i) {
    const uint4 v = *reinterpret_cast<const uint4*>(s_tab + 4 * min(i, n_table - 1));
    STile t;
    t.wp_off = __builtin_amdgcn_readfirstlane(v.x); t.a_off = __builtin_amdgcn_readfirstlane(v.y);
    t.in_ld = __builtin_amdgcn_readfirstlane(v.z); t.info = __builtin_amdgcn_readfirstlane(v.w);
    return t;
  };
  auto lds_layer = [&](int l) {
    SLayer L;
    uint32_t* d = reinterpret_cast<uint32_t*>(&L);
    const uint32_t* src = s_lay + l * (int)(sizeof(SLayer) / 4);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SLayer) / 4); ++i) d[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return L;
  };
  auto fetch_tile_wp = [&](f32x4 (&rb)[4], uint32_t wp) {
    const float* sb = a.wbase + (wp + (uint32_t)wave * 1024u);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[0]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(rb[1]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(rb[2]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(rb[3]) : "v"(lane16), "s"(sb));
  };
  auto fetch_tile = [&](f32x4 (&rb)[4], int i) { fetch_tile_wp(rb, a.tiles[min(i, a.n_table - 1)].wp_off); };   // (prologue: straight from the arguments)
  if (use_table) {
    fetch_tile(rb0, 0); fetch_tile(rb1, 1); fetch_tile(rb2, 2);
    if constexpr (RD == 6) { fetch_tile(rb3, 3); fetch_tile(rb4, 4); fetch_tile(rb5, 5); }
  } else {
    fetch(rb0); fetch(rb1); fetch(rb2); fetch(rb3);   // tiles 0..RD-1 (repeats past the end)
    if constexpr (RD == 6) { fetch(rb4); fetch(rb5); }
  }
  TL(2);
  // ---- chain inputs and biases -> LDS ------------------------------------------------------
  // Every load of the prologue -- the six weight tiles above, the 16-row blocks of both chain
  // inputs, the biases -- is REQUESTED before the first one is waited for: one memory round
  // trip instead of four (dense block, pooled block in two batches, biases: the in-kernel
  // timeline showed 3.9 us here, cold HBM / Infinity-Cache misses each).  A slot is 512 float4
  // (one per thread); slots [0, n0s) belong to input 0, the rest to input 1, so which input a
  // slot reads is uniform.
  {
    constexpr int PRE = RD3 ? 4 : 8;   // slots per batch (512 threads: RMC1 needs 6, RM3's 1024-wide chain 8)
    const SInput in0 = a.in[0];
    const SInput in1 = a.in[a.n_inputs > 1 ? 1 : 0];
    const int n0s = (16 * (in0.cols_pad >> 2) + kThreads - 1) / kThreads;
    const int n1s = a.n_inputs > 1 ? (16 * (in1.cols_pad >> 2) + kThreads - 1) / kThreads : 0;
    const float* base0 = in0.src;
    int64_t row00 = m0, rows0 = a.M;
    if (in0.use_xs) resolve_src(xs, in0.src, a.M, m0, &base0, &row00, &rows0);
    // biases: requested first, stored last (the engine keeps the chains' biases back to back,
    // padded to 4 floats: one flat copy; a global load in the epilogue would put a vmcnt(0)
    // = the full latency of the weight tiles just requested at the end of every pass)
    float bias_v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_v[j] = a.bias[min(tid + j * kThreads, a.n_bias - 1)];
    for (int s0 = 0; s0 < n0s + n1s; s0 += PRE) {
      float4 v[PRE], w2[PRE];
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        const int sl = s0 + j;
        if (sl < n0s + n1s) {                    // uniform
          const bool second = sl >= n0s;         // uniform
          const SInput& in = second ? in1 : in0;
          const float* base = second ? in1.src : base0;
          const int64_t row0 = second ? m0 : row00, rows = second ? a.M : rows0;
          const int qpr = in.cols_pad >> 2, total = 16 * qpr;
          const int idx = min((sl - (second ? n0s : 0)) * kThreads + tid, total - 1);
          const int row = idx / qpr, k = (idx - row * qpr) * 4;
          const int64_t grow = min(row0 + row, rows - 1);
          int64_t off = grow * in.ld + in.col0 + k;
          off = k < in.cols ? off : (int64_t)(zero - base);      // out-of-range k reads the zero page
          asm("" : "+v"(off));
          v[j] = *reinterpret_cast<const float4*>(base + off);
          if (in.col2 >= 0) {                    // uniform: the block is the SUM of two column blocks (NCF)
            int64_t off2 = grow * in.ld + in.col2 + k;
            off2 = k < in.cols ? off2 : (int64_t)(zero - base);
            asm("" : "+v"(off2));
            w2[j] = *reinterpret_cast<const float4*>(base + off2);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        const int sl = s0 + j;
        if (sl < n0s + n1s) {
          const bool second = sl >= n0s;
          const SInput& in = second ? in1 : in0;
          const int qpr = in.cols_pad >> 2, total = 16 * qpr;
          const int idx = (sl - (second ? n0s : 0)) * kThreads + tid;
          const int row = idx / qpr, k = (idx - row * qpr) * 4;
          float4 x = v[j];
          if (in.col2 >= 0) x = make_float4(x.x + w2[j].x, x.y + w2[j].y, x.z + w2[j].z, x.w + w2[j].w);
          float* dst = smem + in.lds_off + in.lds_col0;
          if (idx < total) *reinterpret_cast<float4*>(dst + row * in.lds_ld + k) = swz4(x, row);
          if (in.g_dst && idx < total && k < in.cols && m0 + row < a.M)
            *reinterpret_cast<float4*>(in.g_dst + (m0 + row) * in.g_ldd + k) = x;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + j * kThreads < a.n_bias) smem[a.bias_off + tid + j * kThreads] = bias_v[j];
    for (int i0 = 2 * kThreads; i0 < a.n_bias; i0 += kThreads)     // (more than 1024 bias words: not on any shipped config)
      if (i0 + tid < a.n_bias) smem[a.bias_off + i0 + tid] = a.bias[i0 + tid];
  }
  if (use_table) {
    const uint32_t* kp = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();   // SArgs is argument 0 (constant -> generic address space)
    uint32_t* dt = reinterpret_cast<uint32_t*>(smem + a.tab_off);
    uint32_t* dl = reinterpret_cast<uint32_t*>(smem + a.lay_off);
    for (int i = tid; i < 4 * a.n_table; i += kThreads) dt[i] = kp[offsetof(SArgs, tiles) / 4 + i];
    for (int i = tid; i < a.n_layers * (int)(sizeof(SLayer) / 4); i += kThreads) dl[i] = kp[offsetof(SArgs, L) / 4 + i];
  }
  TL(3);
  if (!PK) {
    DRS_WAIT_TILE(rb0, 0);
    stash(0, rb0);
  }
  __syncthreads();
  TL(4);

  // ---- dot interaction between the chains (one d-ordered fma chain per pair, like the
  // oracle and interact_dot_kernel: bit-identical) -------------------------------------------
  auto interact = [&]() {
    const float* Ts = smem + a.t_off;
    float* Rs = smem + a.r_off;
    const int D = a.D, W = a.r_pad, off = a.itself ? 1 : 0;
    for (int o = tid; o < 16 * W; o += kThreads) {
      const int row = o / W, c = o - row * W;
      const float* t = Ts + row * a.t_ld;
      float v = 0.f;
      if (c < D) {
        v = t[swz(c, row)];
      } else if (c < D + a.P) {
        continue;                                 // the pairs: on the matrix cores, below
      }
      Rs[row * a.r_ld + swz(c, row)] = v;
      if (a.g_R && c < D + a.P && m0 + row < a.M) a.g_R[(m0 + row) * a.g_ldr + c] = v;
    }
    interact_pairs_mfma(Ts, a.t_ld, Rs, a.r_ld, 16, a.F, D, a.itself, a.g_R, a.g_ldr, m0, a.M, kThreads / 64,
                          tid >> 6, tid & 63, [](int c, int row) { return swz(c, row); });
    __syncthreads();
  };

  // ---- consume iterator ------------------------------------------------------------------
  int c_tile = 0;
  int c_l = 0, c_n0 = 0, c_c = 0;
  SLayer cl = a.L[0];
  int c_nch = (cl.K + 63) >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

// timing experiments ("mlp_debug") exist only in the timeline build
#ifdef DRS_TIMELINE
#define DRS_DBG_MFMA_ON (!(a.dbg & 2))
#else
#define DRS_DBG_MFMA_ON true
#endif
#define DRS_ROUND(BUF, RB_FETCH, RB_STASH)                                                        \
  {                                                                                               \
    TL(10);                                                                                       \
    if (a.inter_on && c_tile == a.inter_tile) interact();                                         \
    ++c_tile;                                                                                     \
    TL(11);                                                                                       \
    const int col = c_n0 + wave * 16 + r;                                                         \
    if (c_n0 + wave * 16 < cl.N) {                                                                \
      const float* pa = smem + cl.in_off + r * cl.in_ld + c_c * 64 + gs;                          \
      const float* pb = sB + ((BUF) * 128 + wave * 16 + r) * LD + gs;                             \
      float av[16], bv[16];                                                                       \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) { av[s] = pa[4 * s]; bv[s] = pb[4 * s]; }    \
      /* issue order, pinned: the requests of the tile six ahead; all operand reads; then the    */ \
      /* dependent MFMA chain with one LDS write of the stash in the shadow of every second MFMA.*/ \
      /* Tried and dropped (r2, each 3-5 % slower on RMC1 / W&D / NCF): the requests spread INTO  */ \
      /* the chain (anything between two MFMAs on one accumulator delays the dependent issue),   */ \
      /* and the two waves of a SIMD running request / multiply halves in opposite order.        */ \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB_FETCH, q);                      \
      DRS_WAIT_TILE(RB_STASH, 20);                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                             \
        if (DRS_DBG_MFMA_ON) {                                                                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * q], bv[2 * q], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * q + 1], bv[2 * q + 1], acc, 0, 0, 0);   \
        }                                                                                         \
        stash_part((BUF) ^ 1, RB_STASH, q);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB_FETCH, q);                      \
      DRS_WAIT_TILE(RB_STASH, 20);                                                                \
      stash((BUF) ^ 1, RB_STASH);                                                                 \
    }                                                                                             \
    fetch_advance();                                                                              \
    TL(12);                                                                                       \
    if (c_c == c_nch - 1) {                                                                       \
      if (col < (cl.out_off >= 0 ? cl.out_pad : cl.N)) {                                          \
        const float bias_v = smem[cl.b_off + min(col, cl.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < cl.N ? act_apply(acc[i] + bias_v, cl.act) : 0.f;                  \
          if (cl.out_off >= 0) smem[cl.out_off + row * cl.out_ld + swz(col + cl.out_col0, row)] = v; \
          if (cl.g_out && col < cl.N && m0 + row < a.M) {                                         \
            float* dstg = cl.g_out + (m0 + row) * cl.g_ld + col;                                  \
            if (cl.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
      c_c = 0;                                                                                    \
      c_n0 += 128;                                                                                \
      if (c_n0 >= cl.N) {                                                                         \
        c_n0 = 0;                                                                                 \
        if (c_l + 1 < a.n_layers) { ++c_l; cl = a.L[c_l]; c_nch = (cl.K + 63) >> 6; }             \
      }                                                                                           \
    } else {                                                                                      \
      ++c_c;                                                                                      \
    }                                                                                             \
    TL(13);                                                                                       \
    __syncthreads();                                                                              \
    TL(14);                                                                                       \
  }

// Decomposition of the packed launch by removal (RMC1, 2 048 rows, 34 us): no round loop at all
// (prologue + hand-off only) 14 us; rounds with MFMAs, weight loads, LDS reads and barriers removed
// +10 us; MFMAs + weight loads +9 us; LDS reads + barriers +1 us.  Tried against the +10 us, each
// with no change of the total: the wave index as a scalar and the operand row hoisted per layer
// (fewer VALU), the table-driven rounds above with the rare blocks out of line (68 instructions
// between two MFMA groups instead of 700), the descriptors and layer records in LDS instead of
// the kernel-argument segment (kept: with the arguments in HOST memory, HIP_FORCE_DEV_KERNARG=0,
// the launch takes 49 us, so argument reads are not free), sixteen waves, skewing, prefetching
// the activation operands.
// 32-row workgroups (two activation tiles per weight operand set, two accumulators per wave) were
// built and measured as well: bit-identical, but the launch takes 60 us on 64 CUs instead of 33 us
// on 128 -- a round's time follows its MFMA count, i.e. with two waves per SIMD the rounds run at
// ~37 cycles per MFMA and SIMD, close to the pipe's 32; RMC1 -19 %, NCF -17 %, only RM3 at batch
// 512 +2 %.  What bounds the launch is 16 rows per CU on half the CUs plus ~14 us of fixed cost,
// not the round.
// Where a packed round's time goes (in-kernel timeline, RMC1): the 16 MFMAs of the two waves of a
// SIMD run as one phase at the pipe's rate (32 MFMAs in ~1 100 cycles) and the per-round
// bookkeeping of both (~1 000 cycles: tile addresses, iterator state, epilogue tests) as another
// -- a lone wave issues an fp32 MFMA only every ~75 cycles (also measured in din.hip's
// recurrence), so skewing the two waves against each other buys nothing (tried: s_sleep on waves
// 4..7 after every barrier, 0..1 000 cycles: 33.3-33.7 us throughout), and prefetching the next
// round's activation operands under the MFMAs neither (34.4 us).  The lever left is more MFMAs
// per round and wave (two column tiles sharing the activation operands) or four waves per SIMD.
// Packed form: round i waits for ITS set (requested six rounds ago: at most the 5 x 4 loads of the
// newer sets may still be in flight), reads the 16 activation operands from LDS, runs the
// dependent chain on the set's registers, and only then re-requests into them (tile i + 6).
// Every wave issues its 4 loads every round, also when its 16 columns lie beyond the layer's N
// (zeros in the twin), so the in-order vmcnt arithmetic holds for all of them.
#define DRS_ROUND_PK(RB, NEWER)                                                                   \
  {                                                                                               \
    TL(10);                                                                                       \
    if (a.inter_on && c_tile == a.inter_tile) interact();                                         \
    ++c_tile;                                                                                     \
    const int col = c_n0 + wave * 16 + r;                                                         \
    DRS_WAIT_TILE(RB, NEWER);                                                                     \
    TL(11);                                                                                       \
    if (c_n0 + wave * 16 < cl.N) {                                                                \
      const float* pa = pa_layer + c_c * 64;                                                      \
      float av[16];                                                                               \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) av[s] = pa[4 * s];                           \
      _Pragma("unroll") for (int s = 0; s < 16; ++s)                                              \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    TL(12);                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB, q);                              \
    fetch_advance();                                                                              \
    TL(13);                                                                                       \
    bool layer_done = false;                                                                      \
    if (c_c == c_nch - 1) {                                                                       \
      if (col < (cl.out_off >= 0 ? cl.out_pad : cl.N)) {                                          \
        const float bias_v = smem[cl.b_off + min(col, cl.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < cl.N ? act_apply(acc[i] + bias_v, cl.act) : 0.f;                  \
          if (cl.out_off >= 0) smem[cl.out_off + row * cl.out_ld + swz(col + cl.out_col0, row)] = v; \
          if (cl.g_out && col < cl.N && m0 + row < a.M) {                                         \
            float* dstg = cl.g_out + (m0 + row) * cl.g_ld + col;                                  \
            if (cl.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
      c_c = 0;                                                                                    \
      c_n0 += PASSW;                                                                              \
      if (c_n0 >= cl.N) {                                                                         \
        c_n0 = 0;                                                                                 \
        layer_done = true;                                                                        \
        if (c_l + 1 < a.n_layers) { ++c_l; cl = a.L[c_l]; c_nch = (cl.K + 63) >> 6; }             \
        pa_layer = smem + cl.in_off + r * cl.in_ld + gs;                                          \
      }                                                                                           \
    } else {                                                                                      \
      ++c_c;                                                                                      \
    }                                                                                             \
    /* the only hand-off between waves: a layer's outputs become the next layer's inputs */       \
    if (layer_done) __syncthreads();                                                              \
    TL(14);                                                                                       \
  }

  const float* pa_layer = smem + cl.in_off + r * cl.in_ld + gs;   // (iterator form: per layer, not per round)
  // ---- packed form, table-driven: one scalar descriptor load per round ---------------------
#define DRS_ROUND_T(RB)                                                                           \
  {                                                                                               \
    /* The control chain of a round (descriptor of the next round, packed offset of the tile six  */ \
    /* ahead: LDS read -> readfirstlane -> scalar address) is issued INSIDE the MFMA chain, in     */ \
    /* the ~60 idle issue cycles between two dependent MFMAs: a wave issues in order, so behind    */ \
    /* the chain it costs its full latency every round.                                            */ \
    const uint4 tn_raw = *reinterpret_cast<const uint4*>(s_tab + 4 * min(ti + 1, n_table - 1));   \
    const uint32_t wp_raw = s_tab[4 * min(ti + RD, n_table - 1)];                                 \
    if (__builtin_expect((t.info & (1 << 18)) != 0, 0)) interact();                               \
    const int ncols = t.info & 0xffff;                                                            \
    const bool act_now = wave * 16 < ncols;                                                       \
    if constexpr (RD == 3) { DRS_WAIT_TILE(RB, 8); } else { DRS_WAIT_TILE(RB, 20); }            \
    float av[16];                                                                                 \
    if (__builtin_expect(act_now, 1)) {                                                           \
      const float* pa = smem + t.a_off + r * t.in_ld + gs;                                        \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) av[s] = pa[4 * s];                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                               \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    const uint32_t wp6 = (uint32_t)__builtin_amdgcn_readfirstlane(wp_raw);                        \
    STile tn;                                                                                     \
    tn.wp_off = __builtin_amdgcn_readfirstlane(tn_raw.x); tn.a_off = __builtin_amdgcn_readfirstlane(tn_raw.y); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (__builtin_expect(act_now, 1)) {                                                           \
      _Pragma("unroll") for (int s = 4; s < 8; ++s)                                               \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    tn.in_ld = __builtin_amdgcn_readfirstlane(tn_raw.z); tn.info = __builtin_amdgcn_readfirstlane(tn_raw.w); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (__builtin_expect(act_now, 1)) {                                                           \
      _Pragma("unroll") for (int s = 8; s < 16; ++s)                                              \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    fetch_tile_wp(RB, wp6);                                                                       \
    if (__builtin_expect((t.info & (1 << 16)) != 0, 0)) {      /* last chunk of the pass */        \
      const SLayer el = lds_layer((t.info >> 24) & 0xff);                                         \
      const int col = el.N - ncols + wave * 16 + r;                                               \
      if (col < (el.out_off >= 0 ? el.out_pad : el.N)) {                                          \
        const float bias_v = smem[el.b_off + min(col, el.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < el.N ? act_apply(acc[i] + bias_v, el.act) : 0.f;                  \
          if (el.out_off >= 0) smem[el.out_off + row * el.out_ld + swz(col + el.out_col0, row)] = v; \
          if (el.g_out && col < el.N && m0 + row < a.M) {                                         \
            float* dstg = el.g_out + (m0 + row) * el.g_ld + col;                                  \
            if (el.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
    }                                                                                             \
    if (__builtin_expect((t.info & (1 << 17)) != 0, 0)) __syncthreads();   /* a layer's outputs -> the next layer's inputs */ \
    t = tn;                                                                                       \
    ++ti;                                                                                         \
  }
  if (use_table) {
    int ti = 0;
    STile t = lds_tile(0);
    for (int i = 0; i < n_table; i += RD) {
      DRS_ROUND_T(rb0)
      if (i + 1 >= n_table) break;
      DRS_ROUND_T(rb1)
      if (i + 2 >= n_table) break;
      DRS_ROUND_T(rb2)
      if constexpr (RD == 3) continue;
      if (i + 3 >= n_table) break;
      DRS_ROUND_T(rb3)
      if (i + 4 >= n_table) break;
      DRS_ROUND_T(rb4)
      if (i + 5 >= n_table) break;
      DRS_ROUND_T(rb5)
    }
  } else if constexpr (RD3) {
    // (launched only with a table)
  } else
#undef DRS_ROUND_T
  // this lane's activation operand row inside the current layer's input slab (per layer, not per round)
  if (PK && RD == 4) {
    for (int i = 0; i < a.n_tiles; i += 4) {
      DRS_ROUND_PK(rb0, 12)
      if (i + 1 >= a.n_tiles) break;
      DRS_ROUND_PK(rb1, 12)
      if (i + 2 >= a.n_tiles) break;
      DRS_ROUND_PK(rb2, 12)
      if (i + 3 >= a.n_tiles) break;
      DRS_ROUND_PK(rb3, 12)
    }
  } else if (PK) {
    for (int i = 0; i < a.n_tiles; i += 6) {
      DRS_ROUND_PK(rb0, 20)
      if (i + 1 >= a.n_tiles) break;
      DRS_ROUND_PK(rb1, 20)
      if (i + 2 >= a.n_tiles) break;
      DRS_ROUND_PK(rb2, 20)
      if (i + 3 >= a.n_tiles) break;
      DRS_ROUND_PK(rb3, 20)
      if (i + 4 >= a.n_tiles) break;
      DRS_ROUND_PK(rb4, 20)
      if (i + 5 >= a.n_tiles) break;
      DRS_ROUND_PK(rb5, 20)
    }
  } else
  for (int i = 0; i < a.n_tiles; i += 6) {
    DRS_ROUND(0, rb0, rb1)
    if (i + 1 >= a.n_tiles) break;
    DRS_ROUND(1, rb1, rb2)
    if (i + 2 >= a.n_tiles) break;
    DRS_ROUND(0, rb2, rb3)
    if (i + 3 >= a.n_tiles) break;
    DRS_ROUND(1, rb3, rb4)
    if (i + 4 >= a.n_tiles) break;
    DRS_ROUND(0, rb4, rb5)
    if (i + 5 >= a.n_tiles) break;
    DRS_ROUND(1, rb5, rb0)
  }
#undef DRS_ROUND
#undef DRS_ROUND_PK
#undef DRS_WAIT_TILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing requests
  TL(20);
  signal_done(done, gridDim.x, smem);
#ifdef DRS_TIMELINE
  TL(21);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
}

// ---------------------------------------------------------------------------
// The four-wave forms (round 3's stream3_kernel -- removed in round 5, when stream4_kernel below had overtaken it on
// every launch size: 5-7 query sets +6-9 % queries/s, profiles/r05_stream3_vs_stream4/ -- and stream4_kernel, which
// runs the same step table): the packed form re-cut around what the round-3 microbenchmarks (tools/ubench/) say
// about the fp32 matrix pipe of a SIMD:
//   * ONE wave keeps it busy: a dependent v_mfma_f32_16x16x4_f32 chain issues every 35 cycles, two or
//     four independent chains every 33 (the pipe's rate is 32) -- a second wave per SIMD adds nothing;
//   * a global_load_dwordx4 every 4 MFMAs and a ds_read_b128 every 8, placed BETWEEN the MFMAs, cost
//     1-2 % -- whereas the same instructions issued as a block before or after a round's MFMAs (what
//     the 8-wave forms do) leave the pipe idle for their whole issue + wait time (the in-kernel
//     timeline of the 8-wave form: ~1 100 cycles of MFMA and ~1 000 cycles of everything else per round);
//   * loads issued with EXEC = 0 take part in vmcnt like any other (tools/ubench/masked_vmcnt.hip).
// So: a workgroup is FOUR waves (one per SIMD) and a wave's instruction stream is one unbroken run
// of MFMAs with everything else in their shadow.
//   * a step = (pass, 64-k chunk); a wave owns TPW = 1 / 2 / 4 adjacent 16-column tiles of the pass
//     (a pass covers 4 TPW tiles; TPW by the layer's width), each its own accumulator, all fed by the
//     SAME activation operands: four ds_read_b128 per step and wave, fetched one step ahead into a
//     second register set (slabs keep, inside every 16-column block, column k at position
//     4 (k mod 4) + (k div 4): lane (r, g) finds the operands of four consecutive MFMA steps side by
//     side; rows are 64 m + 8 floats apart, which makes the b128 reads conflict-free);
//   * the weights of step i + RD are requested while step i runs: a ring slot is 4 tiles x 4 float4;
//     the four float4 of k-group q (MFMA steps 4q .. 4q+3) are reloaded right after the q-th quarter
//     of the step has consumed them, so the loads are spread evenly over the step and
//     `s_waitcnt vmcnt(16 (RD-1) + 12)` in front of every quarter is exact (every wave issues exactly
//     16 loads per step: tiles it does not own are requested with EXEC = 0);
//   * one descriptor per step (STile), the next RD of them in scalar registers;
//   * no asm block with register outputs sits under a branch: the compiler then never has to merge
//     two versions of a ring register (it did so with copies -- of registers whose loads were still
//     in flight -- in the first version of this kernel).
// Same packed twins, same k-ordered fma chains, same bits as every other form.
__device__ __forceinline__ int lpos(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }

#define S3_LAST (1 << 16)
#define S3_BARRIER (1 << 17)
#define S3_INTERACT (1 << 18)
#define S3_ANEXT (1 << 19)        // the NEXT step reads the same layer's input slab: its operands may be prefetched
#define S3_TPW_SHIFT 20           // bits 20..22: tiles per wave of this step's layer (1 / 2 / 4)
#define S3_FIRST (1 << 23)        // first chunk of a pass: the accumulators start from zero
#define S3_OFF_0 "0"
#define S3_OFF_1 "1024"
#define S3_OFF_2 "2048"
#define S3_OFF_3 "3072"
#define S3_OFF(Q) S3_OFF_##Q

// stream4_kernel ("mlp_stream" 4): the 4-wave form with every SEGMENT -- all 64-k chunks of one
// (layer, pass) for the 1 / 2 / 4 tiles a wave owns -- run by ONE asm statement (seg_asm.inc, generated
// by tools/gen_seg_asm.py): an unbroken run of MFMAs with the weight reloads, the operand prefetch and
// the loop control between them, no per-step descriptor decode, no EXEC masks (a tile a wave does not
// own is requested from the address of one it owns and its results are dropped by the epilogue).  The
// ring, the operands and the accumulators live in AGPRs under fixed names; the C++ around the
// statements (prologue, epilogues, interaction, hand-off) never touches an AGPR -- the Makefile checks
// the generated ISA for that.  Chunk 0 of the NEXT segment is requested while a segment's last chunk
// runs, so a layer boundary costs an epilogue and a barrier, not a memory round trip.
#include "seg_asm.inc"
// SUM1: the second input is the sum of two column blocks (NCF) -- a template parameter because the third
// staging array costs 32 VGPRs, and at 280 registers per wave instead of 312 a SIMD that hosts one of
// this kernel's waves still has room for two of the gather's (104 each) instead of one.
// TWO: compiled for 256 registers per wave (the input staging arrays halved), so that two workgroups
// share a CU -- what the MLP-bound models want (see stream_kernel's RD3 form).
// R: 16-row slabs per workgroup (1 | 2).  R = 2: a workgroup owns 32 rows as two halves that share every
// weight operand -- twice the MFMAs per byte of weights streamed from L2 and per fixed cost of a
// workgroup; taken for launches of many rows whose slabs still fit LDS ("mlp_rows32").
// SPL: the column-split form (SArgs::ns): blockIdx.x = slab of rows * ns + column slice.  Consecutive workgroups go to
// consecutive XCDs, so slice y of every slab runs on the XCDs k with k % ns == y: an XCD's L2 holds only its slice of
// the split layer's weights (a speed matter only: nothing depends on the placement).
template <bool SUM1, bool TWO, int R = 1, bool SPL = false>
__global__ __launch_bounds__(256, TWO ? 2 : 1) void stream4_kernel(SArgs a, Done done, XSrc xs, NSplit sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kThreads = 256;
  if constexpr (SPL) {          // (NSplit's line of the argument block rides on the burst below)
    static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) == 0xd00, "offset of the NSplit argument");
    uint32_t t_;
    asm volatile("s_load_dword %0, %1, 0xd00" : "=&s"(t_) : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
  }
  kernarg_burst();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  static_assert(R == 1 || (R == 2 && !TWO && !SUM1), "32-row form: one workgroup per CU, no summed input");
  static_assert(!SPL || (!TWO && !SUM1), "column-split form: one workgroup per CU, no summed input");
  const int ns_y = SPL ? (int)(blockIdx.x % (unsigned)a.ns) : 0;            // my column slice of the split layer
  const unsigned slab = SPL ? blockIdx.x / (unsigned)a.ns : blockIdx.x;     // my slab of 16 R rows
  const int64_t m0 = (int64_t)slab * (16 * R);
  // stores of the chains' outputs to GLOBAL memory: one workgroup per slab makes them (slice 0 before the split layer,
  // the last arriver behind it)
  bool gw = !SPL || ns_y == 0;
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(smem + a.lds_floats);
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
  const bool tl_on = slab == 0;
#undef TL_ON
#define TL_ON tl_on
#endif
  TL(1);
  const float* zero = a.zero;
  const int n_table = a.n_table;
  const uint32_t* s_tab = reinterpret_cast<const uint32_t*>(smem + a.tab_off);
  const uint32_t* s_lay = reinterpret_cast<const uint32_t*>(smem + a.lay_off);
  const float* const wbase = a.wbase;
  // A wave's tiles in a segment: byte offsets (from the arena) of their 4-KB blocks in chunk 0, + 16 lane;
  // nex = how many of its tpw tiles exist in the twin (the others are requested from tile 0's address)
  struct Seg { uint32_t off[4]; int nex, tpw, nch; };
  // (tadd: the split layer's steps name slice 0's tiles; slice y works ns_tps y tiles further on)
  auto tadd_of = [&](int i) { return SPL && i >= sp.t0 && i < sp.t1 ? ns_y * sp.tps : 0; };
  auto seg_of = [&](uint32_t wp_off, int pstride, int info, int tadd) {
    Seg q;
    q.tpw = (info >> S3_TPW_SHIFT) & 7;
    q.nch = pstride >> 13;
    const int tile0 = (info & 0xff) + tadd, ntl = (info >> 8) & 0xff;
    const int t0 = tile0 + q.tpw * wave;
    q.nex = min(max(ntl - t0, 0), q.tpw);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = q.nex > 0 ? t0 + min(j, q.nex - 1) : 0;
      q.off[j] = (wp_off + (uint32_t)(t >> 3) * (uint32_t)pstride + (uint32_t)(t & 7) * 1024u) * 4u + (uint32_t)lane * 16u;
    }
    return q;
  };
  auto prefetch = [&](const Seg& q, int slot) {
    if constexpr (R == 2) {
      if (slot)
        asm volatile(SEG2_PREFETCH1_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                     : "memory", SEG2_AGPR_CLOBBER);
      else
        asm volatile(SEG2_PREFETCH0_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                     : "memory", SEG2_AGPR_CLOBBER);
    } else if (slot)
      asm volatile(SEG_PREFETCH1_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                   : "memory", SEG_AGPR_CLOBBER);
    else
      asm volatile(SEG_PREFETCH0_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                   : "memory", SEG_AGPR_CLOBBER);
  };
  // ---- prologue: ONE memory round trip.  The chain inputs (the critical path: cold misses all the
  // way to HBM), the biases and the descriptor table are requested first, the weights of the first
  // RD steps right behind them; nothing is waited for before all of it is in flight.
  // A thread's role in the input copies is fixed: row tid / TPR, columns 4 (tid % TPR) + CG j -- no
  // division, one 64-bit row pointer per input.
  constexpr int TPR = kThreads / (16 * R), CG = 4 * TPR;   // threads per row; columns one pass of them covers (64 | 128)
  constexpr int PB = (TWO ? 256 : 512) / CG;                   // column groups per input and batch (512 columns)
  const int prow = tid / TPR, pk0 = (tid % TPR) * 4;
  const SInput& in0 = a.in[0];
  const SInput& in1 = a.in[a.n_inputs > 1 ? 1 : 0];
  const int nj0 = (in0.cols_pad + CG - 1) / CG, nj1 = a.n_inputs > 1 ? (in1.cols_pad + CG - 1) / CG : 0;
  const float* base0 = in0.src;
  int64_t row00 = m0, rows0 = a.M;
  if (in0.use_xs) resolve_src(xs, in0.src, a.M, m0, &base0, &row00, &rows0);
  const float* const rp0 = base0 + min(row00 + prow, rows0 - 1) * in0.ld + in0.col0;
  const float* const rp1 = in1.src + min(m0 + prow, a.M - 1) * in1.ld + in1.col0;
  const float* const rp2 = in1.src + min(m0 + prow, a.M - 1) * in1.ld + (in1.col2 >= 0 ? in1.col2 : in1.col0);
  const int cols0 = in0.cols, cols1 = in1.cols, cpad0 = in0.cols_pad, cpad1 = in1.cols_pad;
  constexpr bool sum1 = SUM1;
  float* const ld0 = smem + in0.lds_off + prow * in0.lds_ld;
  float* const ld1 = smem + in1.lds_off + prow * in1.lds_ld;
  const int lc0 = in0.lds_col0 + pk0, lc1 = in1.lds_col0 + pk0;
  float* const gd1 = in1.g_dst && m0 + prow < a.M && !SPL ? in1.g_dst + (m0 + prow) * in1.g_ldd : nullptr;
  // (a load beyond the block's real columns reads the zero page: an address select keeps it unconditional)
  auto issue = [&](const float* rp, int cols, int jb, int nj, float4 (&v)[PB]) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (jb + j < nj) {                         // uniform
        const int k = pk0 + CG * (jb + j);
        int64_t off = k < cols ? (int64_t)k : (int64_t)(zero - rp);   // (offset, not pointer, select: the load stays a global_load)
        asm("" : "+v"(off));
        v[j] = *reinterpret_cast<const float4*>(rp + off);
      }
  };
  auto store = [&](float* ld, int lc, int cpad, int jb, int nj, const float4 (&v)[PB]) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (jb + j < nj) {
        const int k = pk0 + CG * (jb + j);
        if (k < cpad) {
          // columns c .. c+3 (c a multiple of 4) sit 4 floats apart inside their 16-column block
          const int c = lc + CG * (jb + j);
          float* dst = ld + ((c & ~15) | ((c >> 2) & 3));
          dst[0] = v[j].x; dst[4] = v[j].y; dst[8] = v[j].z; dst[12] = v[j].w;
        }
      }
  };
  const uint32_t* kp = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();   // SArgs is argument 0
  uint32_t* dt = reinterpret_cast<uint32_t*>(smem + a.tab_off);
  uint32_t* dl = reinterpret_cast<uint32_t*>(smem + a.lay_off);
  const int n_tab_w = 4 * n_table, n_lay_w = a.n_layers * (int)(sizeof(SLayer) / 4);
  float4 pv0[PB], pv1[PB], pv2[PB];
  // early start ("mlp_early", plain 16-row form only): the second input -- the gather's pooled rows -- is fetched at the
  // first step of the second chain, once the gather's flag has been seen; everything before runs beside the gather
  constexpr bool kCanDefer = !SUM1 && !TWO && R == 1 && !SPL;
  const bool defer1 = kCanDefer && done.wait_flag != nullptr && a.wait_tile > 0;   // (uniform)
  issue(rp0, cols0, 0, nj0, pv0);
  if (!defer1) issue(rp1, cols1, 0, nj1, pv1);
  if constexpr (sum1) issue(rp2, cols1, 0, nj1, pv2);
  // biases, descriptors and layer records ride on the same round trip
  constexpr int NBV = 1024 / kThreads, NTV = 512 / kThreads;
  float bias_v[NBV];
  uint32_t tabv[NTV], layv[NTV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) bias_v[j] = a.bias[min(tid + j * kThreads, a.n_bias - 1)];
#pragma unroll
  for (int j = 0; j < NTV; ++j) {
    tabv[j] = kp[offsetof(SArgs, tiles) / 4 + min(tid + j * kThreads, n_tab_w - 1)];
    layv[j] = kp[offsetof(SArgs, L) / 4 + min(tid + j * kThreads, n_lay_w - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
  TL(2);
  // L2 warm-up (see stream3_kernel): one slice of the launch's packed weights per workgroup, fire and
  // forget, into the odd ring slot's registers (every later request retires after these)
  {
    const uint32_t nx = (gridDim.x + 7u) >> 3, rank = blockIdx.x >> 3;
    const uint32_t bytes = (uint32_t)a.warm_bytes;               // a multiple of 4096
    const uint32_t slice = ((bytes / nx) + 4095u) & ~4095u;
    const uint32_t o0 = rank * slice + (uint32_t)tid * 16u, last = bytes - 16u;
    const float* wb = wbase + a.warm_off;
#define S4_WARM(R, I)                                                                             \
    { const uint32_t o_ = min(o0 + (I) * 4096u, last);                                            \
      asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(o_), "s"(wb) : "memory", SEG_AGPR_CLOBBER); }
    S4_WARM("a[80:83]", 0) S4_WARM("a[84:87]", 1) S4_WARM("a[88:91]", 2) S4_WARM("a[92:95]", 3)
    S4_WARM("a[96:99]", 4) S4_WARM("a[100:103]", 5) S4_WARM("a[104:107]", 6) S4_WARM("a[108:111]", 7)
    S4_WARM("a[112:115]", 8) S4_WARM("a[116:119]", 9) S4_WARM("a[120:123]", 10) S4_WARM("a[124:127]", 11)
    S4_WARM("a[128:131]", 12) S4_WARM("a[132:135]", 13) S4_WARM("a[136:139]", 14) S4_WARM("a[140:143]", 15)
#undef S4_WARM
  }
  // chunk 0 of the first segment (descriptor straight from the arguments: its LDS copy is not there yet)
  {
    const STile e0 = a.tiles[0];
    prefetch(seg_of(e0.wp_off, e0.in_ld, e0.info, tadd_of(0)), 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  store(ld0, lc0, cpad0, 0, nj0, pv0);
  if constexpr (sum1) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      pv1[j] = make_float4(pv1[j].x + pv2[j].x, pv1[j].y + pv2[j].y, pv1[j].z + pv2[j].z, pv1[j].w + pv2[j].w);
  }
  if (!defer1) store(ld1, lc1, cpad1, 0, nj1, pv1);
  if (gd1) {                                     // NCF: the summed block is also kept in global memory
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (j < nj1 && pk0 + CG * j < cols1) *reinterpret_cast<float4*>(gd1 + pk0 + CG * j) = pv1[j];
  }
#pragma unroll
  for (int j = 0; j < NBV; ++j)
    if (tid + j * kThreads < a.n_bias) smem[a.bias_off + tid + j * kThreads] = bias_v[j];
#pragma unroll
  for (int j = 0; j < NTV; ++j) {
    if (tid + j * kThreads < n_tab_w) dt[tid + j * kThreads] = tabv[j];
    if (tid + j * kThreads < n_lay_w) dl[tid + j * kThreads] = layv[j];
  }
  // (inputs wider than 8 x 64 columns: further batches, one round trip each)
  for (int jb = PB; jb < nj0; jb += PB) { issue(rp0, cols0, jb, nj0, pv0); store(ld0, lc0, cpad0, jb, nj0, pv0); }
  for (int jb = PB; jb < (defer1 ? 0 : nj1); jb += PB) {
    issue(rp1, cols1, jb, nj1, pv1);
    if constexpr (sum1) {
      issue(rp2, cols1, jb, nj1, pv2);
#pragma unroll
      for (int j = 0; j < PB; ++j)
        pv1[j] = make_float4(pv1[j].x + pv2[j].x, pv1[j].y + pv2[j].y, pv1[j].z + pv2[j].z, pv1[j].w + pv2[j].w);
    }
    store(ld1, lc1, cpad1, jb, nj1, pv1);
    if (gd1) {
#pragma unroll
      for (int j = 0; j < PB; ++j)
        if (jb + j < nj1 && pk0 + CG * (jb + j) < cols1) *reinterpret_cast<float4*>(gd1 + pk0 + CG * (jb + j)) = pv1[j];
    }
  }
  for (int i0 = 1024; i0 < a.n_bias; i0 += kThreads)     // (more than 1024 bias words: not on any shipped config)
    if (i0 + tid < a.n_bias) smem[a.bias_off + i0 + tid] = a.bias[i0 + tid];
  for (int i = tid + 512; i < n_lay_w; i += kThreads) dl[i] = kp[offsetof(SArgs, L) / 4 + i];
  TL(3);
  __syncthreads();
  TL(4);

  // dot interaction between the chains: as stream_kernel's, on this form's slab layout
  auto interact = [&]() {
    const float* Ts = smem + a.t_off;
    float* Rs = smem + a.r_off;
    const int D = a.D, W = a.r_pad, off = a.itself ? 1 : 0;
    for (int o = tid; o < 16 * R * W; o += kThreads) {
      const int row = o / W, c = o - row * W;
      const float* t = Ts + row * a.t_ld;
      float v = 0.f;
      if (c < D) {
        v = t[lpos(c)];
      } else if (c < D + a.P) {
        continue;                                 // the pairs: on the matrix cores, below
      }
      Rs[row * a.r_ld + lpos(c)] = v;
      if (a.g_R && gw && c < D + a.P && m0 + row < a.M) a.g_R[(m0 + row) * a.g_ldr + c] = v;
    }
    interact_pairs_mfma(Ts, a.t_ld, Rs, a.r_ld, 16 * R, a.F, D, a.itself, gw ? a.g_R : nullptr, a.g_ldr, m0, a.M, kThreads / 64,
                          tid >> 6, tid & 63, [](int c, int) { return lpos(c); });
    __syncthreads();
  };

  // the fields of a layer record the epilogue needs, from its LDS copy
  struct Epi { int N, act, out_off, out_ld, out_pad, out_col0, b_off, g_sc1; float* g_out; int64_t g_ld; };
  auto lds_epi = [&](int l) {
    const uint32_t* src = s_lay + l * (int)(sizeof(SLayer) / 4);
    auto w = [&](size_t byte_off) { return (int)__builtin_amdgcn_readfirstlane(src[byte_off / 4]); };
    Epi e;
    e.N = w(offsetof(SLayer, N)); e.act = w(offsetof(SLayer, act));
    e.out_off = w(offsetof(SLayer, out_off)); e.out_ld = w(offsetof(SLayer, out_ld));
    e.out_pad = w(offsetof(SLayer, out_pad)); e.out_col0 = w(offsetof(SLayer, out_col0));
    e.b_off = w(offsetof(SLayer, b_off)); e.g_sc1 = w(offsetof(SLayer, g_sc1));
    const uint64_t glo = (uint32_t)w(offsetof(SLayer, g_out)), ghi = (uint32_t)w(offsetof(SLayer, g_out) + 4);
    e.g_out = reinterpret_cast<float*>(glo | (ghi << 32));
    const uint64_t llo = (uint32_t)w(offsetof(SLayer, g_ld)), lhi = (uint32_t)w(offsetof(SLayer, g_ld) + 4);
    e.g_ld = (int64_t)(llo | (lhi << 32));
    return e;
  };
  // Epilogue of one tile: bias + activation -> the next layer's slab (columns past N inside the pad
  // are zero filled) and / or global memory.  `lim`: columns that exist in the slab; `dst`: this lane's
  // slab address of (row 4 g, its column); the four rows of a lane are out_ld apart.
  auto epilogue = [&](const Epi& el, const float (&acc)[4], float bias_v, int col, int lim, float* dst, int rowoff = 0) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = acc[i] + bias_v;
    if (el.act == DRS_ACT_RELU) {                // (uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (el.act == DRS_ACT_SIGMOID) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], DRS_ACT_SIGMOID);
    }
    if (el.out_off >= 0 && col < lim) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i * el.out_ld] = col < el.N ? v[i] : 0.f;
    }
    if (el.g_out && gw && col < el.N) {          // the last layer of a chain
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + rowoff + g * 4 + i;
        if (row < a.M) {
          float* dstg = el.g_out + row * el.g_ld + col;
          if (el.g_sc1) __hip_atomic_store(dstg, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dstg = v[i];
        }
      }
    }
  };

#define S4_ACC_READ(DST, A0, A1, A2, A3)                                                          \
  asm volatile("v_accvgpr_read_b32 %0, " A0 "\n\tv_accvgpr_read_b32 %1, " A1 "\n\t"               \
               "v_accvgpr_read_b32 %2, " A2 "\n\tv_accvgpr_read_b32 %3, " A3                      \
               : "=v"(DST[0]), "=v"(DST[1]), "=v"(DST[2]), "=v"(DST[3]))
  auto desc = [&](int i) {
    const uint4 d = *reinterpret_cast<const uint4*>(s_tab + 4 * i);
    STile t;
    t.wp_off = __builtin_amdgcn_readfirstlane(d.x); t.a_off = __builtin_amdgcn_readfirstlane(d.y);
    t.in_ld = __builtin_amdgcn_readfirstlane(d.z); t.info = __builtin_amdgcn_readfirstlane(d.w);
    return t;
  };
  int ti = 0, par = 0;          // par: the ring slot this wave's chunk 0 of the segment was requested into
  STile cur = desc(0);
  int tadd = tadd_of(0);
  Seg sg = seg_of(cur.wp_off, cur.in_ld, cur.info, tadd);
  bool alive = true;            // (column-split form: false once another workgroup has taken my slab over)
  while (ti < n_table) {
    const int nti = ti + sg.nch;
    const int last_info = __builtin_amdgcn_readfirstlane(s_tab[4 * (nti - 1) + 3]);
    // the next segment's descriptor now (its chunk 0 is requested from inside this segment's statement),
    // and everything the epilogue needs from LDS -- layer record, biases -- BEFORE the statement: the
    // reads complete under its MFMAs instead of after them
    const STile nx = desc(min(nti, n_table - 1));
    const int tadd_n = tadd_of(min(nti, n_table - 1));
    const Seg sn = seg_of(nx.wp_off, nx.in_ld, nx.info, tadd_n);
    if constexpr (kCanDefer) {
      if (__builtin_expect(defer1 && ti == a.wait_tile, 0)) {
        // the gather's flag (a stream-ordered write queued behind it: its rows are in memory), then the rows
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(done.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != done.wait_val &&
                 ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(4);
          // (bounded: a flag that never comes must not hang the GPU -- bit 1 of the device error word makes the
          // host fail the set instead of handing out sums over rows that were not there yet)
          if (spins >= (1 << 22) && done.dev_err) atomicOr(const_cast<uint32_t*>(done.dev_err), 2u);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int jb = 0; jb < nj1; jb += PB) { issue(rp1, cols1, jb, nj1, pv1); store(ld1, lc1, cpad1, jb, nj1, pv1); }
        __syncthreads();
      }
    }
    if (__builtin_expect((cur.info & S3_INTERACT) != 0, 0)) interact();
    TL(10);
    const Epi el = lds_epi((last_info >> 24) & 0xff);
    const int tpw = sg.tpw;
    const int col0 = ((cur.info & 0xff) + tadd + tpw * wave) * 16 + r;
    const int lim = el.out_off >= 0 ? max(el.out_pad, el.N) : el.N;
    float* const dst = smem + el.out_off + (g * 4) * el.out_ld + lpos(col0 + el.out_col0);
    if (sg.nex > 0) {
      const float b0 = smem[el.b_off + min(col0, el.N - 1)], b1 = smem[el.b_off + min(col0 + 16, el.N - 1)];
      const float b2 = smem[el.b_off + min(col0 + 32, el.N - 1)], b3 = smem[el.b_off + min(col0 + 48, el.N - 1)];
      uint32_t aaddr = (uint32_t)(((cur.a_off & 0xffff) + r * (cur.a_off >> 16) + g * 4) * 4);
      int rem = sg.nch;
      uint32_t r0 = sg.off[0] + 32768u, r1 = sg.off[1] + 32768u, r2 = sg.off[2] + 32768u, r3 = sg.off[3] + 32768u;
      float c0[4], c1[4], c2[4], c3[4];
      if constexpr (R == 2) {
        uint32_t aaddr1 = aaddr + (uint32_t)(16 * (cur.a_off >> 16) * 4);       // rows 16 .. 31 of the slab
        float d0[4], d1[4], d2[4], d3[4];
        if (tpw == 4) {
          asm volatile(SEG2_ASM_T4 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        } else if (tpw == 2) {
          asm volatile(SEG2_ASM_T2 : "+v"(r0), "+v"(r1), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        } else {
          asm volatile(SEG2_ASM_T1 : "+v"(r0), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        }
        TL(12);
        // accumulators: tile j of half h at a[4 (j + tpw h) ...]
        float* const dsth = dst + 16 * el.out_ld;
        if (tpw == 4) {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
          S4_ACC_READ(c2, "a8", "a9", "a10", "a11"); S4_ACC_READ(c3, "a12", "a13", "a14", "a15");
          S4_ACC_READ(d0, "a16", "a17", "a18", "a19"); S4_ACC_READ(d1, "a20", "a21", "a22", "a23");
          S4_ACC_READ(d2, "a24", "a25", "a26", "a27"); S4_ACC_READ(d3, "a28", "a29", "a30", "a31");
          epilogue(el, c2, b2, col0 + 32, lim, dst + 32); epilogue(el, c3, b3, col0 + 48, lim, dst + 48);
          epilogue(el, d2, b2, col0 + 32, lim, dsth + 32, 16); epilogue(el, d3, b3, col0 + 48, lim, dsth + 48, 16);
          epilogue(el, c1, b1, col0 + 16, lim, dst + 16); epilogue(el, d0, b0, col0, lim, dsth, 16);
          epilogue(el, d1, b1, col0 + 16, lim, dsth + 16, 16);
        } else if (tpw == 2) {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
          S4_ACC_READ(d0, "a8", "a9", "a10", "a11"); S4_ACC_READ(d1, "a12", "a13", "a14", "a15");
          epilogue(el, c1, b1, col0 + 16, lim, dst + 16); epilogue(el, d0, b0, col0, lim, dsth, 16);
          epilogue(el, d1, b1, col0 + 16, lim, dsth + 16, 16);
        } else {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(d0, "a4", "a5", "a6", "a7");
          epilogue(el, d0, b0, col0, lim, dsth, 16);
        }
        epilogue(el, c0, b0, col0, lim, dst);
      } else {
      if (tpw == 4) {
        asm volatile(SEG_ASM_T4 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      } else if (tpw == 2) {
        asm volatile(SEG_ASM_T2 : "+v"(r0), "+v"(r1), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      } else {
        asm volatile(SEG_ASM_T1 : "+v"(r0), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      }
      TL(12);
      S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
      S4_ACC_READ(c2, "a8", "a9", "a10", "a11"); S4_ACC_READ(c3, "a12", "a13", "a14", "a15");
      if (tpw == 4) { epilogue(el, c2, b2, col0 + 32, lim, dst + 32); epilogue(el, c3, b3, col0 + 48, lim, dst + 48); }
      epilogue(el, c0, b0, col0, lim, dst);
      if (tpw >= 2) epilogue(el, c1, b1, col0 + 16, lim, dst + 16);
      }
      par = (par + sg.nch) & 1;
    } else {
      // this wave sits the segment out -- but it still has to request the next one's chunk 0, and
      // the columns of the pad that no twin tile covers want zeros in the slab
      if (el.out_off >= 0) {
        for (int h = 0; h < R; ++h)
          for (int t = 0; t < tpw; ++t)
            if (col0 + 16 * t < lim)
              for (int i = 0; i < 4; ++i) dst[16 * t + (16 * h + i) * el.out_ld] = 0.f;
      }
      prefetch(sn, par);
    }
    if (last_info & S3_BARRIER) { TL(13); __syncthreads(); TL(14); }
    if constexpr (SPL) {
      if (nti == sp.t1) {      // (uniform) the split layer is done: my piece of its output slab is in LDS
        // ---- the seam (cdna guide G16 R1, "splitk-seam"): piece -> exchange buffer by 16-byte write-through stores,
        // every wave drains, one lane takes the slab's ticket; whoever draws the last one has every piece visible.
        // The buffer keeps the slab's own column order (lpos permutes inside 16-column blocks; a piece is whole blocks).
        const int cw4 = sp.tps * 4;                       // float4 per row of a piece
        const int n4 = sp.n >> 2;                         // ... of the whole row
        float* const xrow = sp.xbuf + (size_t)m0 * sp.n;
        const float* const sl = smem + sp.off;
        for (int i = tid; i < 16 * R * cw4; i += kThreads) {
          const int row = i / cw4, c4 = ns_y * cw4 + (i - row * cw4);
          const f32x4 v = *reinterpret_cast<const f32x4*>(sl + row * sp.ld + 4 * c4);
          float* dstx = xrow + (size_t)row * sp.n + 4 * c4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstx), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory", SEG_AGPR_CLOBBER);
        __syncthreads();
        TL(15);
        // (the chains' first input slab sits at LDS offset 0 and is dead since layer 0: its first word carries the verdict)
        unsigned* const s_last = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
          const unsigned old = __hip_atomic_fetch_add(sp.xcnt + slab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == (unsigned)a.ns - 1) __hip_atomic_store(sp.xcnt + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_last[0] = old == (unsigned)a.ns - 1;
        }
        __syncthreads();
        alive = s_last[0] != 0;
        TL(16);
        if (!alive) break;
        // last arriver: the other pieces, device-coherent loads (the producers stored write-through), four in flight
        const int o4 = n4 - cw4;                            // float4 per row that are not mine
        for (int i0 = 0; i0 < 16 * R * o4; i0 += 4 * kThreads) {
          f32x4 v[4];
          int at[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = min(i0 + tid + j * kThreads, 16 * R * o4 - 1);
            const int row = i / o4, c = i - row * o4;
            const int c4 = c < ns_y * cw4 ? c : c + cw4;    // skip my own piece
            at[j] = row * sp.ld + 4 * c4;
            const float* src = xrow + (size_t)row * sp.n + 4 * c4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(src));
          }
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i0 + tid + j * kThreads < 16 * R * o4)
              *reinterpret_cast<float4*>(smem + sp.off + at[j]) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
        __syncthreads();
        TL(17);
        gw = true;
      }
    }
    ti = nti;
    cur = nx;
    tadd = tadd_n;
    sg = sn;
  }
#undef S4_ACC_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory", SEG_AGPR_CLOBBER);   // the trailing request
  TL(20);
  if constexpr (SPL) {
    if (alive) signal_done(done, gridDim.x / (unsigned)a.ns, smem, (int)slab);
  } else
  signal_done(done, gridDim.x, smem);
#ifdef DRS_TIMELINE
  TL(21);
  if (tl_on && alive && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
}

#ifdef DRS_TIMELINE
#undef TL_ON
#define TL_ON (blockIdx.x == 0 && blockIdx.y == 0)
#endif
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

template <typename F>
static hipError_t set_max_lds(F kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
  if (e == hipSuccess) e = set_max_lds(stream_kernel<false, 8>);
  if (e == hipSuccess) e = set_max_lds(stream_kernel<true, 8>);
  if (e == hipSuccess) e = set_max_lds(stream_kernel<true, 8, true>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<true, false>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, true>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 2>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 1, true>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 2, true>);
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
      if (sp.ns && sp.packed == 6) hipLaunchKernelGGL((stream4_kernel<false, false, 2, true>), dim3((unsigned)((a.M + 31) / 32) * sp.ns), dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.ns) hipLaunchKernelGGL((stream4_kernel<false, false, 1, true>), dim3(g3.x * sp.ns), dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.packed == 6) hipLaunchKernelGGL((stream4_kernel<false, false, 2>), dim3((unsigned)((a.M + 31) / 32)), dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.packed == 5 && sp.in[1].col2 >= 0) hipLaunchKernelGGL((stream4_kernel<true, false>), g3, dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.packed == 5 && tune.mlp_stream == 4 && tune.mlp_stream_2cu) hipLaunchKernelGGL((stream4_kernel<false, true>), g3, dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.packed == 5) hipLaunchKernelGGL((stream4_kernel<false, false>), g3, dim3(256), slds, s, sp, d, xs, nsp);
      else if (sp.packed && tune.mlp_stream_2cu && sp.n_table > 0) hipLaunchKernelGGL((stream_kernel<true, 8, true>), dim3((unsigned)((a.M + 15) / 16)), dim3(kThreads), slds, s, sp, d, xs);
      else if (sp.packed) hipLaunchKernelGGL((stream_kernel<true, 8>), dim3((unsigned)((a.M + 15) / 16)), dim3(kThreads), slds, s, sp, d, xs);
      else hipLaunchKernelGGL((stream_kernel<false, 8>), dim3((unsigned)((a.M + 15) / 16)), dim3(kThreads), slds, s, sp, d, xs);
      return hipGetLastError();
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
extern "C" int drs_debug_timeline(unsigned long long* out, int cap, int reset) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_n), sizeof n) != hipSuccess) return -1;
  if (n > 16384) n = 16384;
  if ((int)n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * n) != hipSuccess) return -1;
  if (reset) { unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_n), &z, sizeof z); }
  return (int)n;
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

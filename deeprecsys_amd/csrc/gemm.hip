// Wide FC layers (K*N >= "mlp_wide_kn" weights: RM3's 2560x1024, W&D's 1376x1024 and
// 1024x512) as a register-blocked GEMM on the fp32 matrix cores.
//
// Replaces the same FC + Relu|Sigmoid operator pair as mlp.hip (reference
// models/dlrm_s_caffe2.py:258-272) with the same arithmetic contract: every output is ONE
// k-ordered fp32 fma chain from 0, bias added after -- bit-identical to fc_kernel and to
// oracle/drs_oracle.c.
//
// Why a second kernel: fc_kernel gives a workgroup 16 rows x 128 columns, so every 16-row
// slab streams the whole weight panel (2 048 rows x 1376x1024: 720 MB through L2 for one
// layer) and each wave owns ONE accumulator -- 16 dependent MFMAs per K chunk.  Here a
// workgroup owns 64 rows x 128 columns and each of its 8 waves a 32 x 32 block held in
// FOUR independent accumulators (2 x 2 MFMA tiles): an operand read from LDS feeds two
// MFMAs, the MFMA pipe never waits on a dependent result, and the weights are read
// M/64 instead of M/16 times.
//
// Pipeline per 64-deep K chunk c (same scheme as stream_kernel, mlp.hip):
//   issue global loads of chunk c+4 (A: 64x64, W: 128x64; ring of 4 register sets,
//   6 float4 per thread each, inline asm + explicit vmcnt so the ring is never drained);
//   64 MFMAs on chunk c from LDS buffer c&1, operands read one 4-step group ahead;
//   the 12 ds_write_b64 that stash chunk c+1 into buffer (c+1)&1 ride in the MFMA shadow;
//   one barrier, placed three quarters into the chunk (see DRS_GROUND).
#include <string.h>

#include <type_traits>

#include "drs_internal.h"
#include "mlp_dev.h"

namespace drs {
namespace {

constexpr int kGThreads = 512;
constexpr int GKC = 64, GLD = 68;   // k per chunk, padded LDS row

struct GArgs {
  const float* x;
  int64_t ldx;
  int64_t M;
  const float* W;      // [N, K] row-major
  const float* b;
  float* y;
  int64_t ldy;
  const float* zero;   // 16 B of zeros: source of out-of-range float4 loads
  int32_t K, N, act, sc1;
};

// ---- epilogue shared by both kernels ---------------------------------------------------------------
// The MFMAs are issued with the WEIGHT rows as the A operand and the input rows as B (the product is
// commutative: same fma chains, same bits), so a lane ends up with FOUR CONSECUTIVE OUTPUT COLUMNS of one
// output row in four consecutive accumulator registers: bias + activation on a float4 and ONE 16-byte
// store.  (Round 4: the first form -- inputs as A, one dword store per accumulator register, bias loaded
// inside the column loop -- had the compiler put an s_waitcnt vmcnt(0) in front of every element, and
// vmcnt counts stores: 64 serialised write round trips per lane, 25 k - 78 k cycles = 5 % of a
// 2560 x 1024 x 8192 launch, tools/ubench/gemm_lab.hip.)  Stores go through inline asm so nothing waits
// on them before signal_done's own drain; SC1 = write-through (outputs the hand-off reads back).
template <bool SC1>
__device__ __forceinline__ void gst4(float* p, const f32x4 v) {
  // (s_nop 1: the two wait states gfx940+ wants before a VALU may overwrite the data registers of a store of
  // more than 64 bits -- the compiler's hazard recogniser does not look inside the statement)
  if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <bool SC1>
__device__ __forceinline__ void gst1(float* p, const float v) {
  if (SC1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
// bias of columns col .. col + 3 (zeros past N / without a bias)
__device__ __forceinline__ f32x4 bias4(const float* b, int col, int N, bool vec) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (b) {
    if (vec) { if (col < N) r = *reinterpret_cast<const f32x4*>(b + col); }
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) if (col + c < N) r[c] = b[col + c];
    }
  }
  return r;
}
template <bool SC1>
__device__ __forceinline__ void epi4(float* yrow, int col, int N, f32x4 v, const f32x4 b, int act) {
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = act_apply(v[c] + b[c], act);
  if (col < N) gst4<SC1>(yrow + col, v);
}
// the same without the alignment / N % 4 == 0 preconditions (drs_fc on arbitrary operands): one copy,
// activation and store form decided at run time, one dword store per element
__device__ __noinline__ void epi4_any(float* yrow, int col, int N, f32x4 v, const f32x4 b, int act, int sc1) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (col + c < N) {
      const float o = act_apply(v[c] + b[c], act);
      if (sc1) gst1<true>(yrow + col + c, o);
      else gst1<false>(yrow + col + c, o);
    }
}
// vec: float4 accesses of y and the bias are legal for every column group (N % 4 == 0: a group is inside or outside as a whole)
__device__ __forceinline__ bool epi_vec_ok(const float* b, const float* y, int64_t ldy, int N) {
  return !(N & 3) && !(ldy & 3) && !((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b)) & 15);
}
// MODE: 0 plain stores | 1 write-through | -1 the unaligned form
#define DRS_EPI_DISPATCH(CALL) \
  if (!vec) { CALL(-1) } else if (a.sc1) { CALL(1) } else { CALL(0) }

// TM x TN MFMA tiles (16 x 16) per wave; the 8 waves sit 2 x 4, so a workgroup owns
// 32 TM rows x 64 TN columns.  (2,2) is the full-rate shape; the smaller ones exist so a
// launch with few rows (one query: 256) or few columns still covers the chip.
// RING: register sets of chunks in flight (4, or 2 to fit 128 VGPRs); OCC: waves per SIMD the kernel is
// compiled for.  <2, 1, 2, 4> is the shape TWO workgroups of which share a CU (64 x 64 tiles, 70 KB of
// LDS, <= 128 VGPRs): their per-chunk barriers and bookkeeping are not in phase, so one workgroup's
// MFMAs run while the other's waves sit between two runs -- what the two waves of a SIMD inside ONE
// workgroup, locked to each other by the chunk barrier, cannot do for each other.
template <int TM, int TN, int RING = 4, int OCC = 2>
__global__ __launch_bounds__(512, OCC) void gemm_kernel(GArgs a, Done done, XSrc xs) {
  constexpr int GBM = 32 * TM, GBN = 64 * TN;
  constexpr int NA = TM, NB = 2 * TN;            // float4 per thread per chunk
  constexpr int NQ = 2 * (NA + NB);              // ds_write_b64 per stashed chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                        // [2][GBM][68]
  float* sB = smem + 2 * GBM * GLD;        // [2][GBN][68]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int gs = swz(g, r);
  const int wm = wave & 1, wn = wave >> 1;           // my (16 TM) x (16 TN) block inside the tile
  const int64_t m0 = (int64_t)blockIdx.x * GBM;
  const int n0 = blockIdx.y * GBN;
  const int K = a.K, N = a.N;
  const int nch = (K + GKC - 1) / GKC;

  const float* xb; int64_t row0, rows;
  resolve_src(xs, a.x, a.M, m0, &xb, &row0, &rows);

  // staging role: row frow (+32 j) of the A / W tile, floats fk..fk+3 of the chunk
  const int frow = tid >> 4, fk = (tid & 15) * 4;
  const int st_lo = (frow & 8) ? 2 : 0, st_hi = 2 - st_lo;    // swz4 by address (rows 8..15 at k^2)
  float* const stA = sA + frow * GLD + fk;
  float* const stB = sB + frow * GLD + fk;
  // per-thread row offsets (elements), clamped once: rows past the end only feed outputs
  // that are never stored
  int64_t offA[NA], offB[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) offA[j] = min(row0 + frow + 32 * j, rows - 1) * a.ldx + fk;
#pragma unroll
  for (int j = 0; j < NB; ++j) offB[j] = (int64_t)min(n0 + frow + 32 * j, N - 1) * K + fk;
  const int64_t zA = a.zero - xb, zB = a.zero - a.W;

  int f_c = 0;   // next chunk to request (stays on the last one past the end)
  auto fetch = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
    const int k0 = f_c * GKC;
    const bool in = k0 + fk < K;       // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int64_t off = in ? offA[j] + k0 : zA;
      asm("" : "+v"(off));
      const float* p = xb + off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(p));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int64_t off = in ? offB[j] + k0 : zB;
      asm("" : "+v"(off));
      const float* p = a.W + off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[j]) : "v"(p));
    }
    if (f_c + 1 < nch) ++f_c;
  };
  // one of the NQ ds_write_b64 of a chunk's stash
  auto stash_part = [&](int buf, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int q) {
    if (q < 2 * NA) {
      float* p = stA + (buf * GBM + 32 * (q >> 1)) * GLD;
      if (q & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(ra[q >> 1][2], ra[q >> 1][3]);
      else *reinterpret_cast<float2*>(p + st_lo) = make_float2(ra[q >> 1][0], ra[q >> 1][1]);
    } else {
      const int qq = q - 2 * NA;
      float* p = stB + (buf * GBN + 32 * (qq >> 1)) * GLD;
      if (qq & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(rb[qq >> 1][2], rb[qq >> 1][3]);
      else *reinterpret_cast<float2*>(p + st_lo) = make_float2(rb[qq >> 1][0], rb[qq >> 1][1]);
    }
  };
  // explicit wait for one register set: "at most 3 newer chunks outstanding" (vmcnt retires
  // in order); the registers are tied to the asm so no use can be scheduled above it
  auto gwait = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
    constexpr int n = (RING - 1) * (NA + NB);
    if constexpr (NA == 2 && NB == 4)
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) : "n"(n));
    else if constexpr (NA == 1 && NB == 4)
      asm volatile("s_waitcnt vmcnt(%5)" : "+v"(ra[0]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) : "n"(n));
    else if constexpr (NA == 2 && NB == 2)
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]) : "n"(n));
    else
      asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ra[0]), "+v"(rb[0]), "+v"(rb[1]) : "n"(n));
  };
#define DRS_GWAIT(RA, RB, N_) gwait(RA, RB)

  f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB], ra2[NA], rb2[NB], ra3[NA], rb3[NB];
  fetch(ra0, rb0); fetch(ra1, rb1);                                     // chunks 0 .. RING-1
  if constexpr (RING == 4) { fetch(ra2, rb2); fetch(ra3, rb3); }
  DRS_GWAIT(ra0, rb0, 18);
#pragma unroll
  for (int q = 0; q < NQ; ++q) stash_part(0, ra0, rb0, q);
  __syncthreads();

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* const pa0 = sA + (16 * TM * wm + r) * GLD + gs;
  const float* const pb0 = sB + (16 * TN * wn + r) * GLD + gs;

  // operand registers: [ping/pong][tile][k-step of a 4-step group]; set 0 always enters a round
  // holding group 0 of the round's chunk (loaded right after the previous round's barrier)
  float av[2][TM][4], bv[2][TN][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i][s] = pa0[16 * i * GLD + 4 * s];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j][s] = pb0[16 * j * GLD + 4 * s];
  }

  // One K chunk: MFMAs on buffer BUF, stash of the NEXT chunk (sets RAS/RBS) into BUF^1,
  // request of chunk +4 into the sets this chunk came from (RAF/RBF).  The barrier sits after
  // the THIRD of the four operand groups: by then every stash write of the next chunk has been
  // issued and every operand of this chunk has been read, so the last 16 (TM x TN x 4) MFMAs
  // run behind the barrier together with the first operand reads of the next chunk -- the MFMA
  // pipe does not drain while the workgroup synchronises and LDS answers.
#define DRS_GROUND(BUF, RAF, RBF, RAS, RBS)                                                       \
  {                                                                                               \
    fetch(RAF, RBF);                                                                              \
    const float* pa = pa0 + (BUF) * GBM * GLD;                                                    \
    const float* pb = pb0 + (BUF) * GBN * GLD;                                                    \
    DRS_GWAIT(RAS, RBS, 18);                                                                      \
    int wq = 0;   /* stash writes issued so far (compile-time after unrolling) */                  \
    _Pragma("unroll") for (int gq = 0; gq < 3; ++gq) {                                            \
      const int cur = gq & 1, nxt = cur ^ 1;                                                      \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) av[nxt][i][s] = pa[16 * i * GLD + 16 * (gq + 1) + 4 * s]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[nxt][j][s] = pb[16 * j * GLD + 16 * (gq + 1) + 4 * s]; \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                            \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cur][j][s], av[cur][i][s], acc[i][j], 0, 0, 0); \
        if (wq < NQ) { stash_part((BUF) ^ 1, RAS, RBS, wq); ++wq; }                               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
    }                                                                                             \
    static_assert(NQ <= 12, "all stash writes must precede the barrier");                         \
    __syncthreads();                                                                              \
    {                                                                                             \
      const float* pan = pa0 + ((BUF) ^ 1) * GBM * GLD;                                           \
      const float* pbn = pb0 + ((BUF) ^ 1) * GBN * GLD;                                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) av[0][i][s] = pan[16 * i * GLD + 4 * s];   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[0][j][s] = pbn[16 * j * GLD + 4 * s];   \
      }                                                                                           \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                            \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1][j][s], av[1][i][s], acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }

  if constexpr (RING == 4) {
    for (int c = 0; c < nch; c += 4) {
      DRS_GROUND(0, ra0, rb0, ra1, rb1)
      if (c + 1 >= nch) break;
      DRS_GROUND(1, ra1, rb1, ra2, rb2)
      if (c + 2 >= nch) break;
      DRS_GROUND(0, ra2, rb2, ra3, rb3)
      if (c + 3 >= nch) break;
      DRS_GROUND(1, ra3, rb3, ra0, rb0)
    }
  } else {
    for (int c = 0; c < nch; c += 2) {
      DRS_GROUND(0, ra0, rb0, ra1, rb1)
      if (c + 1 >= nch) break;
      DRS_GROUND(1, ra1, rb1, ra0, rb0)
    }
  }
#undef DRS_GROUND
#undef DRS_GWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing requests

  // epilogue: bias + activation.  (Weights are the A operand:) lane (r, g) holds, of each tile, output row r
  // and the four columns 4 g .. 4 g + 3 -- a float4 per tile
  {
    const bool vec = epi_vec_ok(a.b, a.y, a.ldy, N);
    f32x4 b4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b4[j] = bias4(a.b, n0 + 16 * TN * wn + 16 * j + 4 * g, N, vec);
    // every bias load has landed before the first store is issued (an empty asm that "reads" the registers
    // makes the COMPILER wait here, so its scoreboard is clean): the stores below are invisible to its vmcnt
    // bookkeeping, and a wait it placed later for a bias register would count them as well
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(b4[j]));
#define DRS_GEPI(MODE_)                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                      \
      const int64_t row = m0 + 16 * TM * wm + 16 * i + r;                                 \
      if (row < a.M) {                                                                    \
        float* yrow = a.y + row * a.ldy;                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                    \
          { if (MODE_ < 0) epi4_any(yrow, n0 + 16 * TN * wn + 16 * j + 4 * g, N, acc[i][j], b4[j], a.act, a.sc1); \
          else epi4<MODE_ == 1>(yrow, n0 + 16 * TN * wn + 16 * j + 4 * g, N, acc[i][j], b4[j], a.act); } \
      }                                                                                   \
    }
    DRS_EPI_DISPATCH(DRS_GEPI)
#undef DRS_GEPI
  }
  signal_done(done, gridDim.x * gridDim.y, smem);
}

// ---- the same GEMM on v_mfma_f32_32x32x2_f32 (round 4) -------------------------------------------
// A wave owns WTM x WTN tiles of 32 x 32 (the 2 x 2 form: a 64 x 64 block in four independent
// accumulators of 16 registers), a workgroup is FOUR waves sitting 2 x 2 (128 x 128 outputs at
// WTM = WTN = 2), 256 threads, <= 256 registers: two workgroups per CU.  Against the 16x16x4 forms
// above: an MFMA holds the pipe for 64 cycles instead of 32 (half the issues per FLOP), a lane's
// operand register of step s is ONE float for 32 output rows (an operand read from LDS feeds
// 64 x 64 / (64 + 64) = 32 MACs per float instead of 16 x 32 / 48 = 10.7), and the operands of FOUR
// consecutive steps come from one ds_read_b128.
//
// Same arithmetic contract: MFMA step s carries k = 2 s (lanes 0..31) and 2 s + 1 (lanes 32..63),
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)): ONE k-ordered fma chain per output from 0, bias after --
// the bits of gemm_kernel, fc_kernel and oracle/drs_oracle.c (test_gemm_kernel_every_tile_shape_is_bitwise).
//
// LDS image of a 32-deep K chunk, per row 36 floats (32 + 4: the sixteen rows a ds_read_b128 lane
// group touches land on sixteen different 16-B slots of the 256-B bank row).  Lane (i, h) of a wave
// needs, for the four steps of the 8-k group t, k = 8 t + 2 j + h (j = 0..3): those four floats are
// stored NEXT to each other -- element k = 8 t + 2 j + h sits at position 8 t + 4 h + j -- so the lane
// reads them with one ds_read_b128 at 8 t + 4 h.  The staging thread that holds k = 8 t + 4 u' ..
// + 3 (a float4 from global memory) writes (x0, x2) to 8 t + 2 u' and (x1, x3) to 8 t + 4 + 2 u':
// two ds_write_b64, conflict-free (16 consecutive lanes = two rows x eight float4 = 32 banks).
//
// Pipeline per chunk c (LDS buffer c & 1), the scheme of gemm_kernel with a ring of two register sets:
// request chunk c + 2; the four 8-k groups' MFMAs with the ds_write_b64 of chunk c + 1 in their shadow
// and every group's operands read one group ahead; ONE barrier, placed before the last group so its 16
// MFMAs (1024 cycles) cover the barrier and the first operand reads of chunk c + 1.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int G3KC = 32, G3LD = 36;
// tools/ubench/gemm_lab.hip (-DDRS_GEMM_TL): per-workgroup shader-clock / wall-clock stamps of the
// kernel's phases.  Compiled out of the product build.
#ifdef DRS_GEMM_TL
__device__ unsigned long long g_gtl[8 * 8192];
#define GTL(slot) if (tid == 0) { g_gtl[8 * (blockIdx.y * gridDim.x + blockIdx.x) + (slot)] = __builtin_readcyclecounter(); }
#define GTLW(slot) if (tid == 0) { g_gtl[8 * (blockIdx.y * gridDim.x + blockIdx.x) + (slot)] = wall_clock64(); }
#define GTLID(slot) if (tid == 0) { unsigned xcc_, hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc_), "=s"(hw_)); \
    g_gtl[8 * (blockIdx.y * gridDim.x + blockIdx.x) + (slot)] = ((unsigned long long)hw_ << 32) | xcc_; }
#else
#define GTLID(slot)
#define GTL(slot)
#define GTLW(slot)
#endif

// SPREAD (default): the request of chunk c + 2 goes out one load per MFMA step of chunk c's first two groups
// instead of as one block of ~95 instructions between two rounds (tools/ubench/gemm_lab.hip, 2560 x 1024:
// 8 192 rows 377 -> 367 us, 4 096 rows -- one workgroup per CU, nobody to fill the gap -- 198 -> 185 us)
// (DBG, tools/ubench/gemm_lab.hip only: timing experiments that drop a part of the loop -- 2 no LDS stash, 4 no
// barrier, 8 no global requests; the results are then not the GEMM's)
// XSPLIT: the input row is split between two sources (XSrc::ksplit): chunks below ksplit from the query's own dense
// array, the others from a.x at the virtual row -- a second set of row offsets (and, SPREAD == 2, a second descriptor
// and lane offsets that replace the first at the one chunk where the source changes).
template <int WTM, int WTN, int SPREAD = 1, int DBG = 0, bool XSPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm32_kernel(GArgs a, Done done, XSrc xs) {
  constexpr int BM = 64 * WTM, BN = 64 * WTN;
  constexpr int NA = 2 * WTM, NB = 2 * WTN;      // float4 per thread per chunk (rows frow + 32 j)
  constexpr int NQ = 2 * (NA + NB);              // ds_write_b64 per stashed chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                        // [2][BM][36]
  float* sB = smem + 2 * BM * G3LD;        // [2][BN][36]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = a.K, N = a.N;
  const int nch = (K + G3KC - 1) / G3KC;
  GTL(0) GTLW(4) GTLID(6)

  // staging role: row frow (+32 j), floats 4 u .. 4 u + 3 of the chunk
  const int frow = tid >> 3, u = tid & 7, fk = 4 * u;
  const int st = 8 * (u >> 1) + 2 * (u & 1);                    // (x0, x2) here, (x1, x3) at + 4
  float* const stA = sA + frow * G3LD + st;
  float* const stB = sB + frow * G3LD + st;
  // input rows: a 128-row tile may hold two queries' 64-row blocks (first layer: each query's own
  // dense array, XSrc), so every 64-row half resolves its source on its own
  const float* xb[WTM];
  int64_t offA[NA], offB[NB];
  const int ksp = XSPLIT ? xs.ksplit : 0;                 // first column that comes from a.x (XSPLIT)
  int64_t offT[XSPLIT ? NA : 1];                          // ... and the offsets of the virtual rows there
#pragma unroll
  for (int hm = 0; hm < WTM; ++hm) {
    int64_t row0, rows;
    resolve_src(xs, a.x, a.M, m0 + 64 * hm, &xb[hm], &row0, &rows);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      // rows past the end are clamped: they only feed outputs that are never stored
      const int64_t rr = row0 + frow + 32 * jj;
      // (an empty source -- rows == 0, which the engine's query table never holds -- reads its row 0 slot: never row -1)
      offA[2 * hm + jj] = (rr < rows ? rr : (rows > 0 ? rows - 1 : 0)) * (XSPLIT ? (int64_t)ksp : a.ldx) + fk;
      if constexpr (XSPLIT) {
        const int64_t vr = m0 + 64 * hm + frow + 32 * jj;
        offT[2 * hm + jj] = (vr < a.M ? vr : a.M - 1) * a.ldx + fk;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) offB[j] = (int64_t)min(n0 + frow + 32 * j, N - 1) * K + fk;

  int f_c = 0;   // next chunk to request (chunks past the end deliver zeros)
  auto fetch = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
    const int k0 = f_c * G3KC;
    const bool in = k0 + fk < K;       // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const float* p = in ? xb[j >> 1] + offA[j] + k0 : a.zero;
      if constexpr (XSPLIT) { if (in && k0 >= ksp) p = a.x + offT[j] + k0; }
      asm("" : "+v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(p));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* p = in ? a.W + offB[j] + k0 : a.zero;
      asm("" : "+v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[j]) : "v"(p));
    }
    ++f_c;
  };
  auto fetch_one = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], int q) {   // load q of the request `fetch` issues as a whole
    const int k0 = f_c * G3KC;
    const bool in = k0 + fk < K;
    if (q < NA) {
      const float* p = in ? xb[q >> 1] + offA[q] + k0 : a.zero;
      if constexpr (XSPLIT) { if (in && k0 >= ksp) p = a.x + offT[q] + k0; }
      asm("" : "+v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[q]) : "v"(p));
    } else {
      const float* p = in ? a.W + offB[q - NA] + k0 : a.zero;
      asm("" : "+v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[q - NA]) : "v"(p));
    }
    if (q == NA + NB - 1) ++f_c;
  };
  // SPREAD == 2 (the launcher takes it when K % 32 == 0 and every row offset fits 32 bits): the loop's
  // requests as BUFFER loads -- buffer_load_dwordx4 v, v_off32, s[rsrc:rsrc+3], s_k0 offen -- with the lane's
  // byte offset of its row CONSTANT, the chunk's k0 in the scalar offset and the matrix base in the resource
  // descriptor: no vector arithmetic per load at all (the flat form above spends two 64-bit adds, two selects
  // and a nop on each of a round's eight loads).  Requests for chunks past K -- the pad chunk of an odd chunk
  // count, whose MFMAs do run, and the ring's two trailing requests -- go through a descriptor with
  // num_records = 0: every lane is out of range, the hardware returns zeros and touches no memory.
  typedef int rsrc_t __attribute__((ext_vector_type(4)));
  uint32_t voA[NA], voB[NB];
  rsrc_t rsA[WTM], rsB;
  uint32_t s_k0 = 0;              // byte offset of the next request's chunk inside a row
  uint32_t s_nrec = 0xfffff000u;  // ... and the descriptors' num_records for it (0 past K)
  auto mk_rsrc = [](const float* p) {   // raw buffer (stride 0) over a workgroup-uniform base; words held in SGPRs
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    rsrc_t r = {(int)lo, (int)(hi & 0xffffu), (int)0xfffff000u, 0x00020000};
    return r;
  };
  uint32_t voT[XSPLIT ? NA : 1];
  rsrc_t rsT = {0, 0, 0, 0};
  bool switched = false;
  auto set_chunk = [&]() {
    s_k0 = (uint32_t)f_c * (G3KC * 4u);
    s_nrec = f_c < nch ? 0xfffff000u : 0u;
    if constexpr (XSPLIT) {
      if (!switched && f_c * G3KC >= ksp) {      // (uniform) the next request is the first one past the dense columns
        switched = true;
#pragma unroll
        for (int j2 = 0; j2 < NA; ++j2) voA[j2] = voT[j2];
#pragma unroll
        for (int hm = 0; hm < WTM; ++hm) rsA[hm] = rsT;
      }
    }
  };
  if constexpr (SPREAD == 2) {
#pragma unroll
    for (int j2 = 0; j2 < NA; ++j2) voA[j2] = (uint32_t)(offA[j2] * 4);
#pragma unroll
    for (int j2 = 0; j2 < NB; ++j2) voB[j2] = (uint32_t)(offB[j2] * 4);
#pragma unroll
    for (int hm = 0; hm < WTM; ++hm)
      rsA[hm] = mk_rsrc(xb[hm]);
    rsB = mk_rsrc(a.W);
    if constexpr (XSPLIT) {
#pragma unroll
      for (int j2 = 0; j2 < NA; ++j2) voT[j2] = (uint32_t)(offT[j2] * 4);
      rsT = mk_rsrc(a.x);
    }
  }
  auto fetch_one_s = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], int q) {
    rsrc_t r = q < NA ? rsA[q >> 1] : rsB;
    r[2] = (int)s_nrec;
    if (q < NA) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[q]) : "v"(voA[q]), "s"(r), "s"(s_k0));
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[q - NA]) : "v"(voB[q - NA]), "s"(r), "s"(s_k0));
    if (q == NA + NB - 1) { ++f_c; set_chunk(); }
  };
  auto stash_part = [&](int buf, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int q) {
    if (q < 2 * NA) {
      const f32x4 v = ra[q >> 1];
      float* p = stA + (buf * BM + 32 * (q >> 1)) * G3LD + 4 * (q & 1);
      *reinterpret_cast<float2*>(p) = (q & 1) ? make_float2(v[1], v[3]) : make_float2(v[0], v[2]);
    } else {
      const int qq = q - 2 * NA;
      const f32x4 v = rb[qq >> 1];
      float* p = stB + (buf * BN + 32 * (qq >> 1)) * G3LD + 4 * (qq & 1);
      *reinterpret_cast<float2*>(p) = (qq & 1) ? make_float2(v[1], v[3]) : make_float2(v[0], v[2]);
    }
  };
  // wait for one register set: at most the NA + NB loads of the newer request outstanding
  auto gwait = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], auto newer) {
    constexpr int n = decltype(newer)::value;   // loads of the newer request that may stay outstanding
    if constexpr (NA == 4 && NB == 4)
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) : "n"(n));
    else if constexpr (NA == 2 && NB == 4)
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) : "n"(n));
    else if constexpr (NA == 4 && NB == 2)
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]) : "n"(n));
    else
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]) : "n"(n));
  };

  f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
  using AllNewer = std::integral_constant<int, NA + NB>;
  using NoneNewer = std::integral_constant<int, 0>;
  fetch(ra0, rb0);
  fetch(ra1, rb1);
  if constexpr (SPREAD == 2) set_chunk();
  gwait(ra0, rb0, AllNewer{});
#pragma unroll
  for (int q = 0; q < NQ; ++q) stash_part(0, ra0, rb0, q);
  __syncthreads();

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const float* const pa0 = sA + (32 * WTM * wm + i32) * G3LD + 4 * h;
  const float* const pb0 = sB + (32 * WTN * wn + i32) * G3LD + 4 * h;
  // operand registers [ping/pong][tile]: the four floats are the lane's operands of the group's four steps
  f32x4 av[2][WTM], bv[2][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(pa0 + 32 * i * G3LD);
#pragma unroll
  for (int j = 0; j < WTN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(pb0 + 32 * j * G3LD);

  // (Dealing the shadow work out one item per MFMA with a sched_barrier after each was measured too: the same
  // at two workgroups per CU, 6 % SLOWER at one -- the scheduler's own placement inside a step stays.)
#define DRS_G3MFMA(SET, RAS, RBS, RAF, RBF)                                                       \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                              \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i)                                               \
      _Pragma("unroll") for (int j = 0; j < WTN; ++j)                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[SET][j][s_], av[SET][i][s_], acc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int w_ = 0; w_ < (NQ + 11) / 12; ++w_)                                 \
      if (wq < NQ) { if (!(DBG & 2)) stash_part(nbuf, RAS, RBS, wq); ++wq; }                      \
    if (SPREAD && fq < NA + NB) {                                                                 \
      if (DBG & 8) {} else if (SPREAD == 2) fetch_one_s(RAF, RBF, fq); else fetch_one(RAF, RBF, fq); \
      ++fq;                                                                                       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }
  // One K chunk: MFMAs on buffer BUF, stash of the NEXT chunk (sets RAS/RBS) into BUF^1, request of
  // chunk + 2 into the sets this chunk came from (RAF/RBF)
#define DRS_G3ROUND(BUF, RAF, RBF, RAS, RBS)                                                      \
  {                                                                                               \
    if (!SPREAD) fetch(RAF, RBF);                                                                 \
    const float* pa = pa0 + (BUF) * BM * G3LD;                                                    \
    const float* pb = pb0 + (BUF) * BN * G3LD;                                                    \
    if (DBG & 8) {} else if (SPREAD) gwait(RAS, RBS, NoneNewer{}); else gwait(RAS, RBS, AllNewer{}); \
    const int nbuf = (BUF) ^ 1;                                                                   \
    int wq = 0;   /* stash writes issued so far (compile-time after unrolling) */                  \
    int fq = 0;   /* SPREAD: loads of the next request issued so far */                            \
    _Pragma("unroll") for (int gq = 0; gq < 3; ++gq) {                                            \
      const int cur = gq & 1, nxt = cur ^ 1;                                                      \
      _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[nxt][i] = *reinterpret_cast<const f32x4*>(pa + 32 * i * G3LD + 8 * (gq + 1)); \
      _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[nxt][j] = *reinterpret_cast<const f32x4*>(pb + 32 * j * G3LD + 8 * (gq + 1)); \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if (cur == 0) { DRS_G3MFMA(0, RAS, RBS, RAF, RBF) } else { DRS_G3MFMA(1, RAS, RBS, RAF, RBF) }                  \
    }                                                                                             \
    static_assert(NQ <= 12 * ((NQ + 11) / 12), "all stash writes must precede the barrier");      \
    if (!(DBG & 4)) __syncthreads();                                                              \
    {                                                                                             \
      const float* pan = pa0 + nbuf * BM * G3LD;                                                  \
      const float* pbn = pb0 + nbuf * BN * G3LD;                                                  \
      _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(pan + 32 * i * G3LD); \
      _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(pbn + 32 * j * G3LD); \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    { DRS_G3MFMA(1, RAS, RBS, RAF, RBF) }                                                                \
  }

  // Always whole PAIRS of rounds: an exit between the two rounds made the compiler copy all 64
  // accumulator registers at the join (32 v_mov_b64 behind an s_nop 14 on every iteration).  For an
  // odd chunk count the second round of the last pair runs on a chunk past K, which `fetch` delivers
  // as zeros: fma(0, 0, c) = c, the chain's bits do not change (c is never -0: the chain starts at +0).
  GTL(1)
  for (int c = 0; c < nch; c += 2) {
    DRS_G3ROUND(0, ra0, rb0, ra1, rb1)
    DRS_G3ROUND(1, ra1, rb1, ra0, rb0)
  }
#undef DRS_G3ROUND
#undef DRS_G3MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing requests
  GTL(2)

  // epilogue: bias + activation.  (Weights are the A operand:) C/D of a 32 x 32 tile: lane (i32, h) holds output
  // row i32 and, in registers 4 qq .. 4 qq + 3, the four columns 8 qq + 4 h .. + 3 -- four float4 per tile
  {
    const bool vec = epi_vec_ok(a.b, a.y, a.ldy, N);
    f32x4 b4[WTN][4];
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) b4[j][qq] = bias4(a.b, n0 + 32 * WTN * wn + 32 * j + 8 * qq + 4 * h, N, vec);
#pragma unroll
    for (int j = 0; j < WTN; ++j)     // (as in gemm_kernel: bias loads landed before the first asm store)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) asm volatile("" : "+v"(b4[j][qq]));
#define DRS_G3EPI(MODE_)                                                                 \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) {                                     \
      const int64_t row = m0 + 32 * WTM * wm + 32 * i + i32;                              \
      if (row < a.M) {                                                                    \
        float* yrow = a.y + row * a.ldy;                                                  \
        _Pragma("unroll") for (int j = 0; j < WTN; ++j)                                   \
          _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                              \
            const f32x4 v = {acc[i][j][4 * qq], acc[i][j][4 * qq + 1], acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]}; \
            if (MODE_ < 0) epi4_any(yrow, n0 + 32 * WTN * wn + 32 * j + 8 * qq + 4 * h, N, v, b4[j][qq], a.act, a.sc1); \
            else epi4<MODE_ == 1>(yrow, n0 + 32 * WTN * wn + 32 * j + 8 * qq + 4 * h, N, v, b4[j][qq], a.act); \
          }                                                                               \
      }                                                                                   \
    }
    DRS_EPI_DISPATCH(DRS_G3EPI)
#undef DRS_G3EPI
  }
  GTL(3) GTLW(5)
  signal_done(done, gridDim.x * gridDim.y, smem);
}

}  // namespace

// per device (device_init, engine.hip)
hipError_t gemm_set_attrs() {
  for (const void* k : {reinterpret_cast<const void*>(gemm_kernel<2, 2>), reinterpret_cast<const void*>(gemm_kernel<1, 2>),
                        reinterpret_cast<const void*>(gemm_kernel<2, 1>), reinterpret_cast<const void*>(gemm_kernel<1, 1>),
                        reinterpret_cast<const void*>(gemm_kernel<2, 1, 2, 4>)}) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  for (const void* k : {reinterpret_cast<const void*>(gemm32_kernel<2, 2>), reinterpret_cast<const void*>(gemm32_kernel<1, 2>),
                        reinterpret_cast<const void*>(gemm32_kernel<2, 1>), reinterpret_cast<const void*>(gemm32_kernel<1, 1>),
                        reinterpret_cast<const void*>(gemm32_kernel<2, 2, 2>), reinterpret_cast<const void*>(gemm32_kernel<1, 2, 2>),
                        reinterpret_cast<const void*>(gemm32_kernel<2, 1, 2>), reinterpret_cast<const void*>(gemm32_kernel<1, 1, 2>),
                        reinterpret_cast<const void*>(gemm32_kernel<2, 2, 2, 0, true>), reinterpret_cast<const void*>(gemm32_kernel<1, 2, 2, 0, true>)}) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// tune.mlp_gemm ("mlp_gemm"): wide layers through gemm_kernel; tune.gemm_tile ("mlp_gemm_tile"):
// force TM*10+TN (22 | 12 | 21 | 11), 0 = by block count
// false = not applicable (caller falls back to fc_kernel)
static bool launch_gemm_impl(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, const float* b,
                             int32_t N, int32_t act, float* y, int64_t ldy, const Tune& tune,
                             hipStream_t s, const Done& d, const XSrc& xs, hipError_t* err, bool dry) {
  *err = hipSuccess;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  const float* zero_page = tune.zero;
  if (!tune.mlp_gemm || !zero_page || (K & 3) || (ldx & 3) || !al(x) || !al(W)) return false;
  for (int i = 0; i < xs.q.n_q; ++i) if (!al(xs.x[i])) return false;
  GArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.ldx = ldx; a.M = M; a.W = W; a.b = b; a.y = y; a.ldy = ldy; a.zero = zero_page;
  a.K = K; a.N = N; a.act = act; a.sc1 = d.counter != nullptr;
  // largest tile that still gives every CU a workgroup (256 CUs); rows are halved first (the
  // weight panel is the shared operand, re-read once per row block)
  auto blocks = [&](int tm, int tn) { return ((M + 32 * tm - 1) / (32 * tm)) * (int64_t)((N + 64 * tn - 1) / (64 * tn)); };
  int tm = 2, tn = 2;
  bool two_per_cu = false;       // "mlp_gemm_tile" 214: the 64 x 64 shape compiled for two workgroups per CU
  if (tune.gemm_tile == 214) { tm = 2; tn = 1; two_per_cu = true; }
  else if (tune.gemm_tile) { tm = tune.gemm_tile / 10; tn = tune.gemm_tile % 10; }
  // two 64 x 64 workgroups per CU once they fill the chip twice over ("mlp_gemm_2cu", set per model by
  // drs_create): W&D 84.8 k -> 89.2 k queries/s, RM3 config 3 29.5 k -> 30.9 k (= 65 % of the fp32-MFMA
  // peak), RM3 reference JSON 59.2 k -> 62.5 k; MT-WnD loses 5 % with it and keeps the 2 x 2 shape
  else if (tune.gemm_2cu && blocks(2, 1) >= 512) { tm = 2; tn = 1; two_per_cu = true; }
  // (2,2) already when it covers half the chip: in the pipelined engine other launches (the next
  // set's gather, the other MLP streams' GEMMs and chains) fill the remaining CUs, and a 2 x 2
  // wave tile does twice the MFMAs per operand read (measured: RM3 +4 %, W&D +2 % queries/s)
  else if (blocks(2, 2) >= 128) { tm = 2; tn = 2; }
  else if (blocks(1, 2) >= 256) { tm = 1; tn = 2; }
  else if (blocks(2, 1) >= 256) { tm = 2; tn = 1; }
  else { tm = 1; tn = 1; }
  // The 32x32x2 forms (gemm32_kernel; "mlp_gemm_tile" 322 | 321 | 312 | 311 forces one, "mlp_gemm32" 1 =
  // by block count)
  {
    auto b32 = [&](int wm_, int wn_) { return ((M + 64 * wm_ - 1) / (64 * wm_)) * (int64_t)((N + 64 * wn_ - 1) / (64 * wn_)); };
    int w = 0;
    if (tune.gemm_tile >= 300) w = tune.gemm_tile - 300;
    // by block count: the 128 x 128 form when it gives every CU its two workgroups (RM3 config 3's
    // 2560 x 1024 layer at 8 192 rows: 512 of them, 372 us against 391 us for the 16x16x4 kernel, the model
    // +4.5 % queries/s); smaller launches (W&D's 4 096 rows, the 1024 x 256 layer) stay with gemm_kernel, whose
    // 64 x 64 workgroups cover the chip where 128 x 128 ones would leave CUs idle (measured: W&D -5 % otherwise)
    else if (tune.gemm_tile == 0 && tune.gemm32 && b32(2, 2) >= tune.gemm32_blocks) w = 22;
    // ... and MT-WnD and MLP-bound DLRM take the 64 x 128 form for them instead ("mlp_gemm32_small" 12, set per
    // model by drs_create)
    else if (tune.gemm_tile == 0 && tune.gemm32 && tune.gemm32_small &&
             b32(tune.gemm32_small / 10, tune.gemm32_small % 10) >= tune.gemm32_small_blocks) w = tune.gemm32_small;
    if (w) {
      const int wm_ = w / 10, wn_ = w % 10;
      const dim3 grid((unsigned)((M + 64 * wm_ - 1) / (64 * wm_)), (unsigned)((N + 64 * wn_ - 1) / (64 * wn_)));
      const size_t lds = sizeof(float) * 2 * (64 * wm_ + 64 * wn_) * G3LD;
      // scalar-base requests when K is whole 32-k chunks and the row offsets fit 32 bits (gemm32_kernel, SPREAD == 2)
      // (offsets + a row's bytes stay below the descriptors' num_records of 0xfffff000)
      const bool sbase = !(K & 31) && ((uint64_t)M * (uint64_t)ldx + (uint64_t)K) * 4u < 0xfffff000ull && ((uint64_t)N + 1) * (uint64_t)K * 4u < 0xfffff000ull;
      if (xs.ksplit > 0) {
        // a split input row: the scalar-base forms 2 x 2 and 1 x 2 only (what W&D's and MT-WnD's first layer takes at
        // full launch sets); the dense arrays' row offsets fit 32 bits like the others
        const bool ok = sbase && (w == 22 || w == 12) && !(xs.ksplit & 31) && xs.ksplit >= 64 && xs.ksplit < K;
        if (dry) return ok;
        if (!ok) { *err = hipErrorInvalidValue; return true; }
        log_launch(tune.log, "gemm32_kernel<%d,%d,sbase,split%d>[%u x %u wg, %dx%d]", wm_, wn_, xs.ksplit, grid.x, grid.y, K, N);
        if (w == 22) hipLaunchKernelGGL((gemm32_kernel<2, 2, 2, 0, true>), grid, dim3(256), lds, s, a, d, xs);
        else hipLaunchKernelGGL((gemm32_kernel<1, 2, 2, 0, true>), grid, dim3(256), lds, s, a, d, xs);
        *err = hipGetLastError();
        return true;
      }
      if (dry) return false;
      log_launch(tune.log, "gemm32_kernel<%d,%d%s>[%u x %u wg, %dx%d]", wm_, wn_, sbase ? ",sbase" : "", grid.x, grid.y, K, N);
#define DRS_G3LAUNCH(WM_, WN_)                                                                              \
      if (sbase) hipLaunchKernelGGL((gemm32_kernel<WM_, WN_, 2>), grid, dim3(256), lds, s, a, d, xs);          \
      else hipLaunchKernelGGL((gemm32_kernel<WM_, WN_>), grid, dim3(256), lds, s, a, d, xs);
      if (w == 22) { DRS_G3LAUNCH(2, 2) } else if (w == 21) { DRS_G3LAUNCH(2, 1) } else if (w == 12) { DRS_G3LAUNCH(1, 2) } else { DRS_G3LAUNCH(1, 1) }
#undef DRS_G3LAUNCH
      *err = hipGetLastError();
      return true;
    }
  }
  if (dry) return false;
  if (xs.ksplit > 0) { *err = hipErrorInvalidValue; return true; }   // (callers ask gemm_split_applicable first)
  const dim3 grid((unsigned)((M + 32 * tm - 1) / (32 * tm)), (unsigned)((N + 64 * tn - 1) / (64 * tn)));
  const size_t lds = sizeof(float) * 2 * (32 * tm + 64 * tn) * GLD;
  log_launch(tune.log, "gemm_kernel<%d,%d%s>[%u x %u wg, %dx%d]", tm, tn, two_per_cu ? ",2cu" : "", grid.x, grid.y, K, N);
#define DRS_GLAUNCH(TM_, TN_) \
  if (tm == TM_ && tn == TN_) hipLaunchKernelGGL((gemm_kernel<TM_, TN_>), grid, dim3(kGThreads), lds, s, a, d, xs);
  if (two_per_cu) hipLaunchKernelGGL((gemm_kernel<2, 1, 2, 4>), grid, dim3(kGThreads), lds, s, a, d, xs);
  else { DRS_GLAUNCH(2, 2) DRS_GLAUNCH(1, 2) DRS_GLAUNCH(2, 1) DRS_GLAUNCH(1, 1) }
#undef DRS_GLAUNCH
  *err = hipGetLastError();
  return true;
}

bool launch_gemm(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, const float* b,
                 int32_t N, int32_t act, float* y, int64_t ldy, const Tune& tune,
                 hipStream_t s, const Done& d, const XSrc& xs, hipError_t* err) {
  return launch_gemm_impl(x, ldx, M, K, W, b, N, act, y, ldy, tune, s, d, xs, err, false);
}

// the same decisions without a launch: true when the layer would go to a gemm32_kernel form that reads a split row
bool gemm_split_applicable(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, int32_t N, const XSrc& xs,
                           const Tune& tune) {
  if (xs.ksplit <= 0 || N < 64 || K < 64) return false;
  hipError_t err = hipSuccess;
  Done d;
  memset(&d, 0, sizeof d);
  DispatchLog* keep = tune.log;
  (void)keep;
  return launch_gemm_impl(x, ldx, M, K, W, nullptr, N, 0, nullptr, 0, tune, nullptr, d, xs, &err, true) && err == hipSuccess;
}

}  // namespace drs

// Wide FC layers (K*N >= "mlp_wide_kn" weights: RM3's 2560x1024, W&D's 896x1024 and
// 1024x512) as a register-blocked GEMM on the fp32 matrix cores.
//
// Replaces the same FC + Relu|Sigmoid operator pair as mlp.hip (reference
// models/dlrm_s_caffe2.py:258-272) with the same arithmetic contract: every output is ONE
// k-ordered fp32 fma chain from 0, bias added after -- bit-identical to fc_kernel and to
// oracle/drs_oracle.c.
//
// Why a second kernel: fc_kernel gives a workgroup 16 rows x 128 columns, so every 16-row
// slab streams the whole weight panel (2 048 rows x 896x1024: 470 MB through L2 for one
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
//   one barrier.
#include <string.h>

#include "drs_internal.h"
#include "mlp_dev.h"

namespace drs {
namespace {

constexpr int kGThreads = 512;
constexpr int GBM = 64, GBN = 128, GKC = 64, GLD = 68;   // rows / columns / k per chunk, padded row

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

__global__ __launch_bounds__(512) void gemm_kernel(GArgs a, Done done, XSrc xs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                        // [2][64][68]
  float* sB = smem + 2 * GBM * GLD;        // [2][128][68]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int gs = swz(g, r);
  const int wm = wave & 1, wn = wave >> 1;           // my 32 x 32 block inside the 64 x 128 tile
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
  int64_t offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) offA[j] = min(row0 + frow + 32 * j, rows - 1) * a.ldx + fk;
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = (int64_t)min(n0 + frow + 32 * j, N - 1) * K + fk;
  const int64_t zA = a.zero - xb, zB = a.zero - a.W;

  int f_c = 0;   // next chunk to request (stays on the last one past the end)
  auto fetch = [&](f32x4 (&ra)[2], f32x4 (&rb)[4]) {
    const int k0 = f_c * GKC;
    const bool in = k0 + fk < K;       // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t off = in ? offA[j] + k0 : zA;
      asm("" : "+v"(off));
      const float* p = xb + off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(p));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t off = in ? offB[j] + k0 : zB;
      asm("" : "+v"(off));
      const float* p = a.W + off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[j]) : "v"(p));
    }
    if (f_c + 1 < nch) ++f_c;
  };
  // one of the 12 ds_write_b64 of a chunk's stash
  auto stash_part = [&](int buf, const f32x4 (&ra)[2], const f32x4 (&rb)[4], int q) {
    if (q < 4) {
      float* p = stA + (buf * GBM + 32 * (q >> 1)) * GLD;
      if (q & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(ra[q >> 1][2], ra[q >> 1][3]);
      else *reinterpret_cast<float2*>(p + st_lo) = make_float2(ra[q >> 1][0], ra[q >> 1][1]);
    } else {
      const int qq = q - 4;
      float* p = stB + (buf * GBN + 32 * (qq >> 1)) * GLD;
      if (qq & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(rb[qq >> 1][2], rb[qq >> 1][3]);
      else *reinterpret_cast<float2*>(p + st_lo) = make_float2(rb[qq >> 1][0], rb[qq >> 1][1]);
    }
  };
#define DRS_GWAIT(RA, RB, N_)                                                                    \
  asm volatile("s_waitcnt vmcnt(" #N_ ")"                                                        \
               : "+v"(RA[0]), "+v"(RA[1]), "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]))

  f32x4 ra0[2], rb0[4], ra1[2], rb1[4], ra2[2], rb2[4], ra3[2], rb3[4];
  fetch(ra0, rb0); fetch(ra1, rb1); fetch(ra2, rb2); fetch(ra3, rb3);   // chunks 0..3
  DRS_GWAIT(ra0, rb0, 18);
#pragma unroll
  for (int q = 0; q < 12; ++q) stash_part(0, ra0, rb0, q);
  __syncthreads();

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* const pa0 = sA + (32 * wm + r) * GLD + gs;
  const float* const pb0 = sB + (32 * wn + r) * GLD + gs;

  // One K chunk: MFMAs on buffer BUF, stash of the NEXT chunk (sets RAS/RBS) into BUF^1,
  // request of chunk +4 into the sets this chunk came from (RAF/RBF).
#define DRS_GROUND(BUF, RAF, RBF, RAS, RBS)                                                       \
  {                                                                                               \
    fetch(RAF, RBF);                                                                              \
    const float* pa = pa0 + (BUF) * GBM * GLD;                                                    \
    const float* pb = pb0 + (BUF) * GBN * GLD;                                                    \
    float av[2][2][4], bv[2][2][4];   /* [ping/pong][tile][step of the group] */                  \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                               \
      av[0][0][s] = pa[4 * s]; av[0][1][s] = pa[16 * GLD + 4 * s];                                \
      bv[0][0][s] = pb[4 * s]; bv[0][1][s] = pb[16 * GLD + 4 * s];                                \
    }                                                                                             \
    DRS_GWAIT(RAS, RBS, 18);                                                                      \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                            \
      const int cur = gq & 1, nxt = cur ^ 1;                                                      \
      if (gq < 3) {                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                           \
          av[nxt][0][s] = pa[16 * (gq + 1) + 4 * s]; av[nxt][1][s] = pa[16 * GLD + 16 * (gq + 1) + 4 * s]; \
          bv[nxt][0][s] = pb[16 * (gq + 1) + 4 * s]; bv[nxt][1][s] = pb[16 * GLD + 16 * (gq + 1) + 4 * s]; \
        }                                                                                         \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][0][s], bv[cur][0][s], acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][0][s], bv[cur][1][s], acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][1][s], bv[cur][0][s], acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][1][s], bv[cur][1][s], acc[1][1], 0, 0, 0); \
        if (4 * gq + s < 12) stash_part((BUF) ^ 1, RAS, RBS, 4 * gq + s);                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
    }                                                                                             \
    __syncthreads();                                                                              \
  }

  for (int c = 0; c < nch; c += 4) {
    DRS_GROUND(0, ra0, rb0, ra1, rb1)
    if (c + 1 >= nch) break;
    DRS_GROUND(1, ra1, rb1, ra2, rb2)
    if (c + 2 >= nch) break;
    DRS_GROUND(0, ra2, rb2, ra3, rb3)
    if (c + 3 >= nch) break;
    DRS_GROUND(1, ra3, rb3, ra0, rb0)
  }
#undef DRS_GROUND
#undef DRS_GWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing requests

  // epilogue: bias + activation; lane holds rows 4g+q of each tile, column r
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 32 * wn + 16 * j + r;
    if (col < N) {
      const float bcol = a.b ? a.b[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t row = m0 + 32 * wm + 16 * i + 4 * g + q;
          if (row < a.M) {
            const float v = act_apply(acc[i][j][q] + bcol, a.act);
            float* dst = a.y + row * a.ldy + col;
            if (a.sc1) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
          }
        }
    }
  }
  signal_done(done, gridDim.x * gridDim.y, smem);
}

}  // namespace

int g_mlp_gemm = 1;   // drs_set_option "mlp_gemm": wide layers through gemm_kernel

// false = not applicable (caller falls back to fc_kernel)
bool launch_gemm(const float* x, int64_t ldx, int64_t M, int32_t K, const float* W, const float* b,
                 int32_t N, int32_t act, float* y, int64_t ldy, const float* zero_page,
                 hipStream_t s, const Done& d, const XSrc& xs, hipError_t* err) {
  *err = hipSuccess;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!g_mlp_gemm || !zero_page || (K & 3) || (ldx & 3) || !al(x) || !al(W)) return false;
  for (int i = 0; i < xs.q.n_q; ++i) if (!al(xs.x[i])) return false;
  static bool attr = false;
  const size_t lds = sizeof(float) * 2 * (GBM + GBN) * GLD;
  if (!attr) {
    *err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (*err != hipSuccess) return true;
    attr = true;
  }
  GArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.ldx = ldx; a.M = M; a.W = W; a.b = b; a.y = y; a.ldy = ldy; a.zero = zero_page;
  a.K = K; a.N = N; a.act = act; a.sc1 = d.counter != nullptr;
  const dim3 grid((unsigned)((M + GBM - 1) / GBM), (unsigned)((N + GBN - 1) / GBN));
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(kGThreads), lds, s, a, d, xs);
  *err = hipGetLastError();
  return true;
}

}  // namespace drs

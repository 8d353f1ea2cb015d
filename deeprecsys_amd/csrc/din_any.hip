// The any-shape forms of DIN's attention units and DIEN's recurrence.
//
// The reference builds both models from whatever widths its command line names: an attention unit is
// create_mlp over "3*D - <arch_mlp_bot> - D" with any number of hidden layers of any width
// (models/din.py:255-277), the recurrence is rnn_cell.BasicRNN(dim_in = arch_sparse_feature_size,
// dim_out = hidden_size) for any two integers (models/dien.py:308-380).  din.hip's kernels are instantiated for the
// shapes the shipped configs use (one hidden layer of <= 64 units; D in {16, 32, 64}, hidden_size in {8, 16, 32,
// 64}); every other shape takes the kernels below, so that the boundary accepts what the reference accepts.
// They are plain: one workgroup per sample, activations in LDS, each output its own k-ordered fmaf chain with the
// bias added behind it -- the oracle's order (oracle/drs_oracle.c fc_impl), hence the attention output bit-identical
// to it after a sequential-order gather, and the recurrence bit-identical to din.hip's two forms on the shapes all
// three serve (tests/test_gpu_parity.py).  Not tuned: weights stream from L2 per sample.
#include <hip/hip_runtime.h>
#include <string.h>

#include "drs_internal.h"
#include "rnn_dev.h"

namespace drs {
namespace {

// Attention units of one sample.  ln: the unit's n_ln widths (3 D, hidden ..., D) on the device; att: per unit and
// layer {W [out, in] row-major, b}.  LDS: x [3 D] | z [D] | two activation buffers [maxw].
__global__ __launch_bounds__(256) void din_attention_any_kernel(const float* __restrict__ T, int64_t ldt, int Tn, int D,
                                                                int n_ln, const int32_t* __restrict__ ln,
                                                                const float* const* __restrict__ att, int maxw,
                                                                float* __restrict__ R, int64_t ldr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* x = smem;
  float* z = x + 3 * (size_t)D;
  float* act[2] = {z + D, z + D + maxw};
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* e = T + row * ldt;
  const float* ad = e + (int64_t)(Tn - 2) * D;
  const int U = Tn - 3, per = n_ln - 1;
  for (int i = 0; i < U; ++i) {
    const float* u = e + (int64_t)(1 + i) * D;
    for (int k = tid; k < D; k += 256) {               // Concat(u_i, ad, Sum(u_i, ad)) (models/din.py:262-271)
      const float uv = u[k], av = ad[k];
      x[k] = uv; x[D + k] = av; x[2 * D + k] = uv + av;
    }
    __syncthreads();
    const float* in = x;
    for (int l = 0; l < per; ++l) {
      const int K = ln[l], N = ln[l + 1];
      const float* W = att[((size_t)i * per + l) * 2];
      const float* b = att[((size_t)i * per + l) * 2 + 1];
      float* out = act[l & 1];
      const bool last = l == per - 1;
      for (int j = tid; j < N; j += 256) {
        const float* w = W + (int64_t)j * K;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(in[k], w[k], acc);
        float y = acc + b[j];
        y = y > 0.f ? y : 0.f;                          // every layer of a unit is FC + Relu (:186)
        if (last) z[j] = i == 0 ? y : z[j] + y;         // atten_out = Sum(fc_outs), in unit order (:283); j -> thread is fixed
        else out[j] = y;
      }
      __syncthreads();
      in = out;
    }
  }
  float* o = R + row * ldr;                             // Concat(profile, atten_out, ad, context) (:311-318)
  for (int j = tid; j < D; j += 256) {
    o[j] = e[j];
    o[D + j] = z[j];
    o[2 * D + j] = ad[j];
    o[3 * D + j] = e[(int64_t)(Tn - 1) * D + j];
  }
}

// The two BasicRNN layers of one sample; packed as dien_pack_kernel lays the weights out (transposed: consecutive
// threads read consecutive floats).  LDS: x [D] | layer-1 state, old and new [2 H] | layer-2 state [2 H].
__global__ __launch_bounds__(256) void dien_rnn_any_kernel(const float* __restrict__ T, int64_t ldt, QTable q, int Tn, int D,
                                                           int H, const float* __restrict__ packed, float* __restrict__ R,
                                                           int64_t ldr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* x = smem;
  float* s0 = x + D;
  float* s1 = s0 + 2 * (size_t)H;
  const int tid = threadIdx.x;
  const int smp = blockIdx.x;
  int b = smp, bs = q.bs[0], v0 = q.vstart[0];
#pragma unroll
  for (int i = 1; i < DRS_MAX_COALESCE; ++i) {          // whose sample this is (select chain: the argument arrays are never indexed dynamically)
    const bool in = i < q.n_q && smp >= q.cum[i];
    b = in ? smp - q.cum[i] : b;
    bs = in ? q.bs[i] : bs;
    v0 = in ? q.vstart[i] : v0;
  }
  const int U = Tn - 3;
  const float* wi0 = packed;
  const float* wg0 = wi0 + (int64_t)D * H;
  const float* wi1 = wg0 + (int64_t)H * H;
  const float* wg1 = wi1 + (int64_t)H * H;
  const float* bias = wg1 + (int64_t)H * H;
  auto chain = [&](const float* v, const float* w, int K, int j) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(v[k], w[(int64_t)k * H + j], acc);
    return acc;
  };
  for (int j = tid; j < H; j += 256) { s0[j] = 0.f; s1[j] = 0.f; }      // initial_h = 0 (models/dien.py:498-499)
  int cur = 0;
  for (int t = 0; t < U; ++t) {
    // step t of "sample" b reads embedding n % U of sample n / U, n = t * bs + b (the Reshape, :316-320)
    const int n = t * bs + b;
    const int src = n / U, unit = n - src * U;
    const float* xr = T + (int64_t)(v0 + src) * ldt + (int64_t)(1 + unit) * D;
    for (int d = tid; d < D; d += 256) x[d] = xr[d];
    __syncthreads();
    const float* h0 = s0 + (size_t)cur * H;
    float* h0n = s0 + (size_t)(cur ^ 1) * H;
    for (int j = tid; j < H; j += 256) {                // layer 1: Tanh(Sum(FC(h_prev, gates_t), FC(x_t, i2h)))
      const float a0 = chain(x, wi0, D, j) + bias[j];
      const float g0 = chain(h0, wg0, H, j) + bias[H + j];
      h0n[j] = tanh_rnn(g0 + a0);
    }
    __syncthreads();
    const float* h1 = s1 + (size_t)cur * H;
    float* h1n = s1 + (size_t)(cur ^ 1) * H;
    for (int j = tid; j < H; j += 256) {                // layer 2 on layer 1's new state
      const float a1 = chain(h0n, wi1, H, j) + bias[2 * H + j];
      const float g1 = chain(h1, wg1, H, j) + bias[3 * H + j];
      h1n[j] = tanh_rnn(g1 + a1);
    }
    __syncthreads();
    cur ^= 1;
  }
  // top MLP input row: [ last state | user profile | candidate ad | context ] (:411-421)
  float* out = R + (int64_t)(v0 + b) * ldr;
  const float* e = T + (int64_t)(v0 + b) * ldt;
  const float* hl = s1 + (size_t)cur * H;
  for (int j = tid; j < H; j += 256) out[j] = hl[j];
  for (int d = tid; d < D; d += 256) {
    out[H + d] = e[d];
    out[H + D + d] = e[(int64_t)(Tn - 2) * D + d];
    out[H + 2 * D + d] = e[(int64_t)(Tn - 1) * D + d];
  }
}

constexpr size_t kMaxLds = 160 * 1024;

}  // namespace

size_t din_any_lds(int32_t D, int32_t maxw) { return sizeof(float) * (4 * (size_t)D + 2 * (size_t)maxw); }
size_t dien_any_lds(int32_t D, int32_t H) { return sizeof(float) * ((size_t)D + 4 * (size_t)H); }
bool din_any_fits(int32_t D, int32_t maxw) { return din_any_lds(D, maxw) <= kMaxLds; }
bool dien_any_fits(int32_t D, int32_t H) { return dien_any_lds(D, H) <= kMaxLds; }

hipError_t launch_din_attention_any(const float* T, int64_t ldt, int64_t M, int32_t Tn, int32_t D, int32_t n_ln,
                                    const int32_t* d_ln, const float* const* d_att, int32_t maxw, float* R, int64_t ldr,
                                    hipStream_t s) {
  if (M <= 0) return hipSuccess;
  const size_t lds = din_any_lds(D, maxw);
  if (lds > kMaxLds) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {   // (per device and cheap next to this launch: no cached flag)
    const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attention_any_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (r != hipSuccess) return r;
  }
  hipLaunchKernelGGL(din_attention_any_kernel, dim3((unsigned)M), dim3(256), lds, s, T, ldt, Tn, D, n_ln, d_ln, d_att, maxw,
                     R, ldr);
  return hipGetLastError();
}

hipError_t launch_dien_rnn_any(const float* T, int64_t ldt, const QTable& q, int32_t Tn, int32_t D, int32_t H,
                               const float* packed, float* R, int64_t ldr, hipStream_t s) {
  const int64_t n = q.cum[q.n_q];
  if (n <= 0) return hipSuccess;
  const size_t lds = dien_any_lds(D, H);
  if (lds > kMaxLds) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {   // (per device and cheap next to this launch: no cached flag)
    const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(dien_rnn_any_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (r != hipSuccess) return r;
  }
  hipLaunchKernelGGL(dien_rnn_any_kernel, dim3((unsigned)n), dim3(256), lds, s, T, ldt, q, Tn, D, H, packed, R, ldr);
  return hipGetLastError();
}

}  // namespace drs

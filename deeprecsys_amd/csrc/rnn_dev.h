// Device helper shared by the DIEN recurrence's translation units (din.hip, din_any.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace drs {
namespace {

// tanh for the recurrence: 8 values per lane and step in the matrix-core form, where the library
// tanhf (two divergent paths, ~50 instructions) cost more than the MFMAs.  |x| < 0.25: the odd
// Taylor polynomial through x^9 (next term < 2e-9 relative); else 1 - 2 / (e^{2|x|} + 1) on
// v_exp_f32 / v_rcp_f32.  Within ~8 ulp of libm's tanhf (worst near |x| = 0.25); both DIEN kernels
// use it, so they agree bitwise with each other and with the oracle to the tolerance in
// tests/test_gpu_parity.py.
__device__ __forceinline__ float tanh_rnn(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);          // e^{2|x|}
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float x2 = ax * ax;
  float p = fmaf(x2, 0.021869488536155203f, -0.053968253968253971f);        // 62/2835, -17/315
  p = fmaf(x2, p, 0.13333333333333333f);                                    // 2/15
  p = fmaf(x2, p, -0.33333333333333331f);
  p = fmaf(x2, p, 1.0f) * ax;
  return copysignf(ax < 0.25f ? p : big, x);
}

}  // namespace
}  // namespace drs

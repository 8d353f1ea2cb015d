// GPU-box lab for the wide-layer GEMM (deeprecsys_amd/csrc/gemm.hip): launches gemm32_kernel /
// gemm_kernel directly (no engine), times them with HIP events, and -- built with -DDRS_GEMM_TL --
// reads the per-workgroup phase stamps: where a launch's cycles go (prologue / K loop / epilogue),
// how far the workgroups' start and end times are spread, and what the shader clock really is under
// the fp32 matrix load (shader-clock ticks per 100 MHz wall tick).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDRS_GEMM_TL -I deeprecsys_amd/csrc \
//       tools/ubench/gemm_lab.hip -o tools/ubench/gemm_lab
//   tools/ubench/gemm_lab [M K N [variants]]     variants: how many of the forms below to run (default 3: the
//   kernel with block / spread / scalar-base requests; 4-6 the knock-out timings -- their
//   results are not the GEMM's; 7-8 read LDS never written and have faulted: do not run them)
#include "../../deeprecsys_amd/csrc/gemm.hip"

namespace drs { void log_launch(DispatchLog*, const char*, ...) {} }   // (engine.hip's dispatch log: not in this lab)

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace drs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// k-ordered fma chain per output (the arithmetic contract), one thread per output
__global__ void ref_kernel(const float* x, const float* W, const float* b, float* y, int M, int K, int N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = __fmaf_rn(x[(int64_t)m * K + k], W[(int64_t)n * K + k], acc);
  acc += b[n];
  y[i] = acc > 0.f ? acc : 0.f;
}

template <typename F>
static float time_launch(F launch, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 2560, N = argc > 3 ? atoi(argv[3]) : 1024;
  const int n_variants = argc > 4 ? atoi(argv[4]) : 3;
  CK(gemm_set_attrs());
  for (const void* k : {(const void*)gemm32_kernel<2, 2, 0>, (const void*)gemm32_kernel<2, 2, 1, 2>, (const void*)gemm32_kernel<2, 2, 1, 4>, (const void*)gemm32_kernel<2, 2, 1, 8>, (const void*)gemm32_kernel<2, 2, 1, 6>, (const void*)gemm32_kernel<2, 2, 1, 14>})
    CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  std::vector<float> hx((size_t)M * K), hW((size_t)N * K), hb(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hW) v = rnd() - 0.5f;
  for (auto& v : hb) v = rnd();
  float *x, *W, *b, *y, *yr, *zero;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4));
  CK(hipMalloc(&y, (size_t)M * N * 4)); CK(hipMalloc(&yr, (size_t)M * N * 4)); CK(hipMalloc(&zero, 256));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(zero, 0, 256));
  ref_kernel<<<(unsigned)(((int64_t)M * N + 255) / 256), 256>>>(x, W, b, yr, M, K, N);
  CK(hipDeviceSynchronize());

  GArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.ldx = K; a.M = M; a.W = W; a.b = b; a.y = y; a.ldy = N; a.zero = zero; a.K = K; a.N = N; a.act = DRS_ACT_RELU; a.sc1 = 0;
  Done d;
  memset(&d, 0, sizeof d);
  XSrc xs;
  memset(&xs, 0, sizeof xs);
  const double fl = 2.0 * M * K * N;
  std::vector<float> hy((size_t)M * N), hr((size_t)M * N);
  CK(hipMemcpy(hr.data(), yr, hr.size() * 4, hipMemcpyDeviceToHost));
  auto check = [&](const char* name) {
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < hy.size(); ++i) bad += memcmp(&hy[i], &hr[i], 4) != 0;
    printf("  %-28s bitwise vs k-ordered fma chain: %s (%zu of %zu differ)\n", name, bad ? "DIFFERENT" : "identical", bad, hy.size());
  };
  auto report = [&](const char* name, float us) {
    printf("%-28s M=%d K=%d N=%d  %.2f us  %.1f TFLOP/s  (%.3f of 157.3)\n", name, M, K, N, us, fl / us / 1e6, fl / us / 1e6 / 157.3);
  };

  for (int variant = 0; variant < n_variants && variant < 8; ++variant) {
    a.ldx = K;
    const dim3 grid((M + 127) / 128, (N + 127) / 128);
    const size_t lds = sizeof(float) * 2 * (128 + 128) * G3LD;
    static const char* names[8] = {"gemm32_kernel<2,2,0>", "gemm32_kernel<2,2,SPREAD>", "gemm32<2,2,2> scalar-base requests", "SPREAD, no stash", "SPREAD, no barrier",
                                   "SPREAD, no requests", "SPREAD, no stash/barrier", "SPREAD, none of the three"};
    const char* nm = names[variant];
    auto launch = [&]() {
      switch (variant) {
        case 0: hipLaunchKernelGGL((gemm32_kernel<2, 2, 0>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 2: hipLaunchKernelGGL((gemm32_kernel<2, 2, 2>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 3: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1, 2>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 4: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1, 4>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 5: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1, 8>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 6: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1, 6>), grid, dim3(256), lds, 0, a, d, xs); break;
        case 7: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1, 14>), grid, dim3(256), lds, 0, a, d, xs); break;
        default: hipLaunchKernelGGL((gemm32_kernel<2, 2, 1>), grid, dim3(256), lds, 0, a, d, xs); break;
      }
    };
    CK(hipMemset(y, 0, (size_t)M * N * 4));
    const float us = time_launch(launch, 20);
    report(nm, us);
    check(nm);
#ifdef DRS_GEMM_TL
    launch();
    CK(hipDeviceSynchronize());
    const int nwg = grid.x * grid.y;
    std::vector<unsigned long long> t(8 * nwg);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_gtl), sizeof(unsigned long long) * t.size()));
    std::vector<double> pro, loop, epi, tot, startw, endw, clk;
    unsigned long long w0 = ~0ull, w1 = 0;
    for (int i = 0; i < nwg; ++i) { w0 = std::min(w0, t[8 * i + 4]); w1 = std::max(w1, t[8 * i + 5]); }
    for (int i = 0; i < nwg; ++i) {
      const unsigned long long* q = &t[8 * i];
      pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
      tot.push_back((double)(q[3] - q[0]));
      startw.push_back((q[4] - w0) / 100.0); endw.push_back((q[5] - w0) / 100.0);
      clk.push_back((double)(q[3] - q[0]) / ((q[5] - q[4]) / 100.0) / 1e3);   // shader ticks per us -> GHz
    }
    auto stat = [&](const char* nm, std::vector<double> v, const char* unit) {
      std::sort(v.begin(), v.end());
      double sum = 0; for (double z : v) sum += z;
      printf("  %-22s min %10.1f  p10 %10.1f  med %10.1f  p90 %10.1f  max %10.1f  avg %10.1f %s\n", nm, v[0], v[v.size() / 10], v[v.size() / 2],
             v[v.size() * 9 / 10], v.back(), sum / v.size(), unit);
    };
    printf("  %d workgroups; launch span (first start -> last end) %.2f us\n", nwg, (w1 - w0) / 100.0);
    stat("prologue", pro, "cycles");
    stat("K loop", loop, "cycles");
    stat("epilogue", epi, "cycles");
    stat("whole workgroup", tot, "cycles");
    stat("start after first", startw, "us");
    stat("end after first start", endw, "us");
    stat("shader clock", clk, "GHz");
    {   // K loop by XCD, and by position of the workgroup's tile
      double sx[8] = {0}, nx[8] = {0};
      std::vector<double> by_y(grid.y, 0.0), by_xo(8, 0.0);
      for (int i = 0; i < nwg; ++i) {
        const unsigned xcc = (unsigned)(t[8 * i + 6] & 0xf);
        sx[xcc & 7] += loop[i]; nx[xcc & 7] += 1;
        by_y[i / grid.x] += loop[i] / grid.x;
        by_xo[(i % grid.x) * 8 / grid.x] += loop[i] / (nwg / 8.0);
      }
      printf("  K loop avg by XCD (workgroups):");
      for (int k = 0; k < 8; ++k) printf(" %d: %.0f (%.0f)", k, nx[k] ? sx[k] / nx[k] : 0.0, nx[k]);
      printf("\n  K loop avg by blockIdx.y:");
      for (unsigned k = 0; k < grid.y; ++k) printf(" %.0f", by_y[k]);
      printf("\n  K loop avg by eighth of blockIdx.x:");
      for (int k = 0; k < 8; ++k) printf(" %.0f", by_xo[k]);
      // pairs sharing a CU: same (xcc, se, cu) bits of HW_ID
      printf("\n  first 16 workgroups: (id xcc hw_id loop)");
      for (int i = 0; i < 16 && i < nwg; ++i) printf(" (%d %u %08x %.0f)", i, (unsigned)(t[8 * i + 6] & 0xf), (unsigned)(t[8 * i + 6] >> 32), loop[i]);
      printf("\n");
    }
    const double ideal = (double)((K + 31) / 32) * 64 * 64 * 2;   // two waves per SIMD x 64 MFMAs x 64 cycles per chunk
    printf("  K loop ideal at two workgroups per CU: %.0f cycles (MFMA pipe time of both waves of a SIMD)\n", ideal);
#endif
  }
  a.ldx = K;
  {
    const dim3 grid((M + 63) / 64, (N + 63) / 64);
    const size_t lds = sizeof(float) * 2 * (64 + 64) * GLD;
    auto launch = [&]() { hipLaunchKernelGGL((gemm_kernel<2, 1, 2, 4>), grid, dim3(kGThreads), lds, 0, a, d, xs); };
    CK(hipMemset(y, 0, (size_t)M * N * 4));
    const float us = time_launch(launch, 20);
    report("gemm_kernel<2,1,2,4>", us);
    check("gemm_kernel<2,1,2,4>");
  }
  return 0;
}

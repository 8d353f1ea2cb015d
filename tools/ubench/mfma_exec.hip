// Microbenchmark (GPU box): does v_mfma_f32_16x16x4_f32 honour EXEC?  stream3_kernel interleaves its
// EXEC-masked weight loads with MFMAs; if a cleared EXEC bit suppressed the matrix instruction's
// result rows / columns, that interleaving would be illegal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int masked) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + (threadIdx.x & 15), b = 2.0f;
  if (masked) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\t"
                 "s_mov_b64 exec, %1\n\ts_nop 15\n\ts_nop 15" : "+v"(acc), "=&s"(sv) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
  }
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = acc[i];
}
int main() {
  float* d; hipMalloc(&d, 1024);
  std::vector<float> h0(256), h1(256);
  k<<<1, 64>>>(d, 0); hipMemcpy(h0.data(), d, 1024, hipMemcpyDeviceToHost);
  k<<<1, 64>>>(d, 1); hipMemcpy(h1.data(), d, 1024, hipMemcpyDeviceToHost);
  int same = 0, zero = 0;
  for (int i = 0; i < 256; ++i) { same += h0[i] == h1[i]; zero += h1[i] == 0.f; }
  printf("MFMA with EXEC = 0: %d of 256 results equal the unmasked ones, %d are zero\n", same, zero);
  for (int i = 0; i < 12; ++i) printf("  [%d] unmasked %g masked %g\n", i, h0[i], h1[i]);
  return 0;
}

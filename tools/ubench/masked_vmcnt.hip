// Microbenchmark (GPU box): does a global_load issued with EXEC = 0 take part in vmcnt (incremented at
// issue, retired in order behind older loads)?  stream3_kernel's weight ring wants a constant
// `s_waitcnt vmcnt(N)`; waves without a tile in a step would issue their loads with EXEC = 0.
//   control:  1 slow load, NO masked loads, s_waitcnt vmcnt(4)  -> must read the sentinel (stale): the
//             test is sensitive
//   test:     1 slow load, 4 masked loads, s_waitcnt vmcnt(4)   -> loaded value  <=> masked loads count
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void fill(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

template <int MASKED>
__global__ void t(const float* cold, float* out) {
  f32x4 v = {-1.f, -1.f, -1.f, -1.f};
  f32x4 d0, d1, d2, d3;
  const unsigned off = threadIdx.x * 16u;
  const float* base = cold + (size_t)blockIdx.x * (1u << 20);   // 4 MB apart: every block its own cold lines
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base));
  if (MASKED) {
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %4, exec\n\t"
        "s_mov_b64 exec, 0\n\t"
        "global_load_dwordx4 %0, %5, %6\n\t"
        "global_load_dwordx4 %1, %5, %6 offset:1024\n\t"
        "global_load_dwordx4 %2, %5, %6 offset:2048\n\t"
        "global_load_dwordx4 %3, %5, %6 offset:3072\n\t"
        "s_mov_b64 exec, %4"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&s"(sv)
        : "v"(off), "s"(base));
  }
  asm volatile("s_waitcnt vmcnt(4)" : "+v"(v));
  out[blockIdx.x * blockDim.x + threadIdx.x] = v[0];      // the store reads the register when it issues
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
}

int main() {
  const size_t n = (size_t)1 << 30;   // 4 GB of floats? no: 1 Gi floats = 4 GB
  float *cold, *out, *junk;
  if (hipMalloc(&cold, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&junk, (size_t)1 << 30);
  hipMalloc(&out, 1 << 20);
  fill<<<4096, 256>>>(cold, n, 7.0f);
  for (int variant = 0; variant < 2; ++variant) {
    fill<<<4096, 256>>>(junk, (size_t)1 << 28, 1.0f);   // 1 GB through the caches: evicts `cold`
    hipDeviceSynchronize();
    const int blocks = 512;
    if (variant == 0) t<0><<<blocks, 64>>>(cold, out); else t<1><<<blocks, 64>>>(cold + 512u * (1u << 20) / 1, out);
    hipDeviceSynchronize();
    std::vector<float> h(blocks * 64);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int loaded = 0, stale = 0, other = 0;
    for (float x : h) { if (x == 7.0f) ++loaded; else if (x == -1.0f) ++stale; else ++other; }
    printf("%s: loaded %d, stale %d, other %d of %zu lanes\n", variant ? "1 slow load + 4 EXEC=0 loads, vmcnt(4)" : "control: 1 slow load, vmcnt(4)",
           loaded, stale, other, h.size());
  }
  return 0;
}

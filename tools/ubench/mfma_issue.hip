// Microbenchmark (GPU box): how fast does ONE wave / do TWO waves of a SIMD issue
// v_mfma_f32_16x16x4_f32 -- one dependent chain, or 2 / 4 independent chains issued alternately --
// and what do global_load_dwordx4 / ds_read_b128 instructions placed BETWEEN the MFMAs cost?
// Inputs of the stream3 / gemm kernel designs.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int LD_EVERY, int DS_EVERY>
__global__ __launch_bounds__(512) void k(float* out, const float* w, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  f32x4 ld[4] = {};
  f32x4 ds = {0.f, 0.f, 0.f, 0.f};
  const unsigned off = (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 4096u;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        const int n = s * CHAINS + c;
        if (LD_EVERY && n % LD_EVERY == 0)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[(n / LD_EVERY) & 3]) : "v"(off), "s"(w + (size_t)(it & 63) * 8192));
        if (DS_EVERY && n % DS_EVERY == 0) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(ds) : "v"((unsigned)(size_t)lp));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (LD_EVERY) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]));
    if (DS_EVERY) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ds));
    a += ld[0][0] * 1e-30f + ds[0] * 1e-30f;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS, int LD_EVERY, int DS_EVERY>
void run(const char* name, int waves, int blocks, float* out, const float* w, unsigned long long* cyc) {
  const int iters = 200;
  k<CHAINS, LD_EVERY, DS_EVERY><<<blocks, waves * 64>>>(out, w, cyc, iters);
  k<CHAINS, LD_EVERY, DS_EVERY><<<blocks, waves * 64>>>(out, w, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * waves);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  double mx = 0, sum = 0;
  for (auto v : h) { sum += v; mx = v > mx ? v : mx; }
  const double n = (double)iters * 16 * CHAINS;
  printf("%-44s waves/WG %d blocks %3d: %.1f cyc per MFMA per wave (avg), %.1f (slowest wave); per SIMD %.1f\n", name, waves,
         blocks, sum / h.size() / n, mx / n, sum / h.size() / n / (waves > 4 ? waves / 4 : 1));
}

int main() {
  float *out, *w;
  unsigned long long* cyc;
  hipMalloc(&out, 4 << 20);
  hipMalloc(&w, 64 * 8192 * 4 + (1 << 20));
  hipMemset(w, 0, 64 * 8192 * 4 + (1 << 20));
  hipMalloc(&cyc, 1 << 20);
  for (int blocks : {1, 256}) {
    for (int waves : {1, 4, 8}) {
      run<1, 0, 0>("1 chain", waves, blocks, out, w, cyc);
      run<2, 0, 0>("2 chains", waves, blocks, out, w, cyc);
      run<4, 0, 0>("4 chains", waves, blocks, out, w, cyc);
      run<1, 4, 0>("1 chain + global_load_dwordx4 every 4", waves, blocks, out, w, cyc);
      run<2, 4, 0>("2 chains + global_load_dwordx4 every 4", waves, blocks, out, w, cyc);
      run<4, 4, 0>("4 chains + global_load_dwordx4 every 4", waves, blocks, out, w, cyc);
      run<2, 4, 8>("2 chains + gload every 4 + ds_read_b128 every 8", waves, blocks, out, w, cyc);
      run<4, 4, 8>("4 chains + gload every 4 + ds_read_b128 every 8", waves, blocks, out, w, cyc);
      run<2, 0, 2>("2 chains + ds_read_b128 every 2", waves, blocks, out, w, cyc);
    }
  }
  return 0;
}

// GPU-box lab: is the gather's speed a property of WHERE in HBM its rows live?
// Allocates physical memory in 1 GiB handles with the virtual-memory API (one naturally aligned
// buddy block each), maps every handle at its own address and measures, per handle:
//   rand256: waves reading random 256-byte rows (16 lanes x 16 B, 4 rows per load instruction, 80
//            rows per wave -- the access shape of SparseLengthsSum on 64-wide fp32 rows)
//   stream : a plain coalesced read of the whole GiB
// then, for the fastest and the slowest handle: the same random-row probe on each 64 MiB sixteenth,
// and again after the two handles swapped their virtual addresses (is the property the memory's or
// the mapping's?).  One line of JSON per measurement on stdout.
//   hipcc -O3 --offload-arch=gfx950 hbm_regions.hip -o hbm_regions && ./hbm_regions [max_chunks]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline float4 ld_nt(const float4* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// one wave = one "bag": L random rows of 256 B out of `rows`, summed; 4 rows per load instruction
__global__ __launch_bounds__(64) void rand_rows(const float4* __restrict__ base, uint32_t rows, int L, uint64_t seed, float4* out) {
  const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int j = g; j < L; j += 4) {
    const uint32_t r = (uint32_t)(mix(seed + (uint64_t)blockIdx.x * 1024 + j) % rows);
    const float4 v = ld_nt(base + (uint64_t)r * 16 + gl);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (int m = 16; m < 64; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
  }
  if (g == 0) out[(uint64_t)blockIdx.x * 16 + gl] = acc;
}

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ base, uint64_t n4, float4* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
    const float4 v = ld_nt(base + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 123.456f) out[0] = acc;
}

static hipEvent_t e0, e1;
static float4* d_out;

static double time_rand(const void* base, size_t bytes, int reps) {
  const uint32_t rows = (uint32_t)(bytes / 256);
  const int waves = 24576, L = 80;
  hipLaunchKernelGGL(rand_rows, dim3(waves), dim3(64), 0, 0, (const float4*)base, rows, L, 1ull, d_out);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rand_rows, dim3(waves), dim3(64), 0, 0, (const float4*)base, rows, L, 77ull * (i + 2), d_out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)waves * L * 256.0 * reps / (ms * 1e-3) / 1e9;      // GB/s of row bytes
}
static double time_stream(const void* base, size_t bytes, int reps) {
  hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const float4*)base, bytes / 16, d_out);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const float4*)base, bytes / 16, d_out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  const int max_chunks = argc > 1 ? atoi(argv[1]) : 270;
  const size_t chunk = argc > 2 ? (size_t)atoll(argv[2]) << 20 : (size_t)1 << 30;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMalloc(&d_out, 24576 * 256));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc desc = {};
  desc.location = prop.location;
  desc.flags = hipMemAccessFlagsProtReadWrite;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("{\"granularity\": %zu, \"chunk\": %zu}\n", gran, chunk);
  std::vector<hipMemGenericAllocationHandle_t> h;
  std::vector<void*> va;
  std::vector<double> gr, gs;
  for (int i = 0; i < max_chunks; ++i) {
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    if (free_b < chunk + ((size_t)3 << 30)) break;
    hipMemGenericAllocationHandle_t hh;
    if (hipMemCreate(&hh, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    void* p = nullptr;
    CK(hipMemAddressReserve(&p, chunk, 0, nullptr, 0));
    CK(hipMemMap(p, chunk, 0, hh, 0));
    CK(hipMemSetAccess(p, chunk, &desc, 1));
    CK(hipMemset(p, 0, chunk));
    CK(hipDeviceSynchronize());
    h.push_back(hh); va.push_back(p);
    const double r = time_rand(p, chunk, 6), s = time_stream(p, chunk, 3);
    gr.push_back(r); gs.push_back(s);
    printf("{\"chunk_no\": %d, \"va\": \"%p\", \"rand256_GBs\": %.1f, \"stream_GBs\": %.1f}\n", i, p, r, s);
    fflush(stdout);
  }
  const int n = (int)h.size();
  if (n < 2) return 0;
  const int ib = (int)(std::max_element(gr.begin(), gr.end()) - gr.begin());
  const int iw = (int)(std::min_element(gr.begin(), gr.end()) - gr.begin());
  for (int which : {ib, iw})
    for (int k = 0; k < 16; ++k) {
      const size_t sub = chunk / 16;
      const double r = time_rand((const char*)va[which] + k * sub, sub, 6);
      printf("{\"sub_of\": %d, \"sixteenth\": %d, \"rand256_GBs\": %.1f}\n", which, k, r);
    }
  // swap the two handles' addresses
  CK(hipMemUnmap(va[ib], chunk));
  CK(hipMemUnmap(va[iw], chunk));
  CK(hipMemMap(va[ib], chunk, 0, h[iw], 0));
  CK(hipMemMap(va[iw], chunk, 0, h[ib], 0));
  CK(hipMemSetAccess(va[ib], chunk, &desc, 1));
  CK(hipMemSetAccess(va[iw], chunk, &desc, 1));
  printf("{\"swapped\": [%d, %d], \"fast_memory_at_slow_va_GBs\": %.1f, \"slow_memory_at_fast_va_GBs\": %.1f, \"before\": [%.1f, %.1f]}\n", ib, iw,
         time_rand(va[iw], chunk, 6), time_rand(va[ib], chunk, 6), gr[ib], gr[iw]);
  // an arena of 4 chunks mapped back to back: all fast, all slow, alternating
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return gr[a] > gr[b]; });
  auto arena = [&](const char* name, std::vector<int> pick) {
    for (int i : pick) (void)hipMemUnmap(va[i], chunk);
    void* p = nullptr;
    CK(hipMemAddressReserve(&p, chunk * pick.size(), 0, nullptr, 0));
    for (size_t k = 0; k < pick.size(); ++k) CK(hipMemMap((char*)p + k * chunk, chunk, 0, h[pick[k]], 0));
    CK(hipMemSetAccess(p, chunk * pick.size(), &desc, 1));
    printf("{\"arena\": \"%s\", \"chunks\": %zu, \"rand256_GBs\": %.1f, \"stream_GBs\": %.1f}\n", name, pick.size(),
           time_rand(p, chunk * pick.size(), 6), time_stream(p, chunk * pick.size(), 2));
    CK(hipMemUnmap(p, chunk * pick.size()));
    CK(hipMemAddressFree(p, chunk * pick.size()));
    for (int i : pick) { CK(hipMemMap(va[i], chunk, 0, h[i], 0)); CK(hipMemSetAccess(va[i], chunk, &desc, 1)); }
  };
  if (n >= 16) {
    // (ib / iw were swapped above: leave them out)
    std::vector<int> f, s;
    for (int i : order) if (i != ib && i != iw && f.size() < 4) f.push_back(i);
    for (auto it = order.rbegin(); it != order.rend(); ++it) if (*it != ib && *it != iw && s.size() < 4) s.push_back(*it);
    arena("4 fastest", f);
    arena("4 slowest", s);
    arena("2 fastest + 2 slowest", {f[0], s[0], f[1], s[1]});
  }
  return 0;
}

// Device helpers shared by the MLP translation units (mlp.hip, gemm.hip).
#pragma once
#include "drs_internal.h"

namespace drs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == DRS_ACT_RELU) return v > 0.0f ? v : 0.0f;
  if (act == DRS_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// 4 consecutive floats of row `row` starting at column k of a [rows, K] matrix with
// leading dimension ld; zero outside.  Branch-free on purpose: every load is issued
// unconditionally from a clamped (always valid) address and masked afterwards, so the
// compiler can keep all of a chunk's loads in flight behind ONE s_waitcnt (a load
// inside divergent control flow gets its own vmcnt(0) at the join, which serialised
// the five loads of a chunk and cost ~5x).  VEC: base 16-B aligned, ld % 4 == 0 and
// K % 4 == 0, so a float4 never straddles the end of a row.
template <bool VEC>
__device__ __forceinline__ float4 load4_raw(const float* __restrict__ p, int64_t ld, int64_t row,
                                            int64_t rows, int k, int K) {
  // rows past the end are clamped, not masked: they only feed output rows/columns that
  // are never stored
  const float* q = p + (row < rows ? row : rows - 1) * ld;
  if (VEC) {
    return *reinterpret_cast<const float4*>(q + (k < K ? k : 0));
  } else {
    return make_float4(q[min(k + 0, K - 1)], q[min(k + 1, K - 1)], q[min(k + 2, K - 1)],
                       q[min(k + 3, K - 1)]);
  }
}
// zero the k >= K tail; applied when the value is written to LDS (i.e. after the MFMA
// block of the previous chunk), so the loads stay in flight across that block
// LDS bank swizzle.  Operand reads are ds_read_b32 at float address row*LD + 4s + g with
// LD = KC+4 (== 4 mod 32), so rows r and r+8 of a 16-row tile hit the same banks (2-way
// conflict on every read; measured: 46 % of all LDS cycles).  Storing element k of rows
// 8..15 at column k^2 instead moves them onto the two banks rows 0..7 leave free.  Bit 1
// of k never leaves its 16-byte slot, so the staging writes stay one ds_write_b128 with
// the halves of the float4 swapped.
__device__ __forceinline__ float4 swz4(const float4 v, int row) {
  return (row & 8) ? make_float4(v.z, v.w, v.x, v.y) : v;
}
__device__ __forceinline__ int swz(int col, int row) { return col ^ ((row & 8) >> 2); }

__device__ __forceinline__ float4 mask4(const float4 v, int k, int K) {
  return make_float4(k + 0 < K ? v.x : 0.f, k + 1 < K ? v.y : 0.f, k + 2 < K ? v.z : 0.f,
                     k + 3 < K ? v.w : 0.f);
}

// see Done in drs_internal.h.  Every thread fences its own output stores at system
// scope, the workgroup joins, then one lane takes a ticket; the workgroup that draws
// the last ticket knows all outputs are visible and publishes the flag.
// bid_in >= 0: this workgroup's rank among the n_blocks that sign off (the column-split form: one per slab of rows).
__device__ __forceinline__ void signal_done(const Done d, unsigned n_blocks, void* lds_scratch, int bid_in = -1) {
  if (!d.counter) return;
  unsigned* s_u = reinterpret_cast<unsigned*>(lds_scratch);   // dynamic LDS is free by now
  if (d.ts) {
    // every workgroup folds its slice of the gather's clock stamps into (min start, max end)
    const unsigned bid = bid_in >= 0 ? (unsigned)bid_in : blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned per = (d.ts_blocks + n_blocks - 1) / n_blocks;
    const unsigned end = min(d.ts_blocks, (bid + 1) * per);
    unsigned long long lo = ~0ull, hi = 0ull;
    for (unsigned i = bid * per + threadIdx.x; i < end; i += blockDim.x) {
      const unsigned long long a = d.ts[2 * i], b = d.ts[2 * i + 1];
      lo = a < lo ? a : lo;
      hi = b > hi ? b : hi;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const unsigned long long lo2 = __shfl_xor(lo, m), hi2 = __shfl_xor(hi, m);
      lo = lo2 < lo ? lo2 : lo;
      hi = hi2 > hi ? hi2 : hi;
    }
    // one (min, max) pair per workgroup, stored write-through; the last arriver folds them.
    // (Atomics on one address from every wave serialise at ~12 ns each: 2048 of them cost
    // more than the whole MLP.)
    unsigned long long* s_q = reinterpret_cast<unsigned long long*>(s_u + 4);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_q[2 * (threadIdx.x >> 6)] = lo; s_q[2 * (threadIdx.x >> 6) + 1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (unsigned w = 1; w < blockDim.x / 64; ++w) {
        lo = s_q[2 * w] < lo ? s_q[2 * w] : lo;
        hi = s_q[2 * w + 1] > hi ? s_q[2 * w + 1] : hi;
      }
      unsigned long long* part = reinterpret_cast<unsigned long long*>(d.span_acc) + 2 * bid;
      __hip_atomic_store(part, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(part + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // publish this workgroup's outputs (device memory, stored write-through = sc1, see
  // LayerIo::o_sc1): every wave drains its stores, then ONE lane takes a ticket; no L2
  // write-back fence is needed for write-through data (cdna guide G16, form R1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(d.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_u[0] = old == n_blocks - 1;
  }
  __syncthreads();
  if (!s_u[0]) return;
  // last workgroup of the launch: everything the query produced is visible to it.  Stream the
  // outputs, the device error word and the gather's clock span to host-mapped pinned memory in
  // ONE batch of write-through system-scope stores, wait once for them, then publish the flag.
  // (16 bytes per lane and 4 loads in flight: NCF hands 1 MB per 16-query launch set over this way.  Round 4 tried
  // 12 loads in flight, and the last arriver of each of 16 runs of workgroups copying its run's rows so that 16 CUs
  // write to the host at once: neither moved NCF -- 404-424 k queries/s with three sets in flight either way, 558 k
  // with the copy left out: the 27-31 GB/s of 64-byte PCIe writes are the bound, not who issues them.)
  // Round trips on this path: [loads of outputs | error word | span partials, all in flight
  // together] -> [host stores, one PCIe acknowledgement wait] -> flag.  The first version loaded
  // the error word and stored it only after the outputs had been acknowledged: two more
  // dependent round trips (~3 us) on every launch set.
  unsigned err = 0;
  if (threadIdx.x == 0) err = __hip_atomic_load(d.dev_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long lo = ~0ull, hi = 0ull;
  if (d.ts) {
    // fold the per-workgroup (min start, max end) pairs: all threads, then waves, then one lane
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(d.span_acc);
    for (unsigned b = threadIdx.x; b < n_blocks; b += blockDim.x) {
      const unsigned long long a = __hip_atomic_load(acc + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long z = __hip_atomic_load(acc + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lo = a < lo ? a : lo;
      hi = z > hi ? z : hi;
    }
  }
  {
    const unsigned n4 = d.out_words >> 2;
    const bool al = ((reinterpret_cast<uintptr_t>(d.dev_out) | reinterpret_cast<uintptr_t>(d.host_out)) & 15) == 0;
    unsigned done4 = 0;
    if (al) {
      for (unsigned i0 = 0; i0 < n4; i0 += 4 * blockDim.x) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned i = min(i0 + threadIdx.x + j * blockDim.x, n4 - 1);
          const float* p = d.dev_out + 4 * (size_t)i;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(p));   // device-coherent
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned i = i0 + threadIdx.x + j * blockDim.x;
          float* q = d.host_out + 4 * (size_t)i;
          // system scope, write-through.  (s_nop 1: gfx940+ wants TWO wait states between a store of more than 64 bits
          // and a VALU write of its data registers, and the compiler's hazard recogniser does not look inside the
          // statement.  A 12-deep version of this loop had its batch registers reused one instruction after the
          // store: sporadic wrong first words of a float4 on the host -- round 4, found by the MT-WnD parity test.)
          if (i < n4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(q), "v"(v[j]) : "memory");
        }
      }
      done4 = n4 << 2;
    }
    for (unsigned i = done4 + threadIdx.x; i < d.out_words; i += blockDim.x)
      __hip_atomic_store(d.host_out + i,
                         __hip_atomic_load(d.dev_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sc1 load (bypasses my L1) -> write-through store
  }
  if (d.ts) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const unsigned long long lo2 = __shfl_xor(lo, m), hi2 = __shfl_xor(hi, m);
      lo = lo2 < lo ? lo2 : lo;
      hi = hi2 > hi ? hi2 : hi;
    }
    unsigned long long* s_q = reinterpret_cast<unsigned long long*>(s_u + 4);
    if ((threadIdx.x & 63) == 0) { s_q[2 * (threadIdx.x >> 6)] = lo; s_q[2 * (threadIdx.x >> 6) + 1] = hi; }
    __syncthreads();            // (uniform: d.ts is a kernel argument)
    if (threadIdx.x == 0) {
      for (unsigned w = 1; w < blockDim.x / 64; ++w) {
        lo = s_q[2 * w] < lo ? s_q[2 * w] : lo;
        hi = s_q[2 * w + 1] > hi ? s_q[2 * w + 1] : hi;
      }
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(d.host_span), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(d.host_span) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.host_err, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // every payload store above is a write-through system-scope store; once each wave has seen
  // its own acknowledged (vmcnt(0)) and the workgroup has met, the flag can follow without an
  // L2 write-back fence (G16 R1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && d.host_flag)              // (null: the flag follows a DMA copy of the outputs, engine "out_dma")
    __hip_atomic_store(d.host_flag, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Where do the 16 input rows of the slab starting at virtual row m0 come from?  With
// coalesced queries the first layer reads each query's own staged dense array.
__device__ __forceinline__ void resolve_src(const XSrc& xs, const float* x, int64_t M, int64_t m0,
                                            const float** base, int64_t* row0, int64_t* rows) {
  *base = x; *row0 = m0; *rows = M;
  if (xs.q.n_q > 0) {
    const float* p = xs.x[0];
    int lo = xs.q.vstart[0], n = xs.q.bs[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const bool in = i < xs.q.n_q && m0 >= xs.q.vstart[i];
      p = in ? xs.x[i] : p;
      lo = in ? xs.q.vstart[i] : lo;
      n = in ? xs.q.bs[i] : n;
    }
    if (xs.q.n_q > 8) {   // (launch sets of 9 .. 16 queries only: smaller ones never touch the upper half of the argument arrays)
#pragma unroll
      for (int i = 8; i < DRS_MAX_COALESCE; ++i) {
        const bool in = i < xs.q.n_q && m0 >= xs.q.vstart[i];
        p = in ? xs.x[i] : p;
        lo = in ? xs.q.vstart[i] : lo;
        n = in ? xs.q.bs[i] : n;
      }
    }
    *base = p; *row0 = m0 - lo; *rows = n;
  }
}

}  // namespace
}  // namespace drs

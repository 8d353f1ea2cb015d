// Shared by the MLP translation units (mlp.hip: planning, chain / fc kernels; mlp_stream8.hip: stream_kernel;
// mlp_stream4.hip: stream4_kernel): the launch description a stream kernel receives, the step-table flags, device
// helpers, and the launch functions each kernel translation unit exports.
#pragma once
#include "drs_internal.h"
#include "mlp_dev.h"

namespace drs {

struct SLayer {
  const float* W;          // [N, K] row-major
  uint32_t w_off, wp_off;  // ... as a float offset from SArgs::wbase (the engine's weight arena);
                           // wp_off: the packed twin (stream_kernel<true>)
  const float* b;          // [N] or nullptr
  int32_t K, N, act;
  int32_t in_off, in_ld;   // input slab: float offset in LDS, leading dimension
  int32_t out_off, out_ld; // output slab (out_off < 0: none)
  int32_t out_pad;         // columns [N, out_pad) of the output slab are zero filled
  int32_t out_col0;        // first column of this layer's outputs inside the output slab
  int32_t b_off;           // LDS copy of the bias (zeros when b == nullptr), N floats
  float* g_out;            // global output or nullptr
  int64_t g_ld;
  int32_t g_sc1;           // write-through stores (final outputs handed over by signal_done)
};
struct SInput {            // 16 x cols block of a global matrix -> LDS slab, zero padded to cols_pad
  const float* src;
  int64_t ld;
  int32_t col0, cols, cols_pad;
  int32_t lds_off, lds_ld, lds_col0;
  int32_t use_xs;
  int32_t col2;            // >= 0: the block is src[:, col0..] + src[:, col2..] (NCF's Sum)
  float* g_dst;            // also store the block to global (row-major, ld g_ldd) or nullptr
  int64_t g_ldd;
};
#define DRS_MAX_STREAM_LAYERS (2 * DRS_MAX_CHAIN)
#define DRS_MAX_STREAM_TILES 96
// A round of the packed stream kernel, precomputed by the host (stream_plan): which packed tile,
// where the activation operands sit, how many of the pass's columns exist, what happens after it.
// The iterator form keeps ~40 scalars of layer / pass / chunk state alive across six unrolled
// rounds -- they did not fit the SGPR file: the compiled round re-read kernel arguments and
// shuffled 120 spilled scalars through VGPR lanes, and the bare control flow of RMC1's 26 rounds
// (MFMAs, loads, LDS reads and barriers removed) took 10.7 of the launch's 34 us.
struct STile {
  uint32_t wp_off;         // packed tile (wave 0's slice) as a float offset from SArgs::wbase
  int32_t a_off;           // activation operands: LDS float offset of (row 0, k = 64 c) in the layer's input slab
  int32_t in_ld;           // ... and the slab's leading dimension
  int32_t info;            // bits 0..15: columns of this pass that exist (N - n0, capped); 16: last chunk of the
                           // pass (epilogue); 17: last round of the layer (barrier); 18: the dot interaction
                           // runs before this round; 24..31: layer index
};
struct SArgs {
  int32_t n_layers, n_tiles, sB_off, n_inputs;
  int32_t dbg, lds_floats;
  int32_t wait_tile, ns;      // wait_tile: first step of the second chain (where a launch with Done::wait_flag waits
                              // for the gather and fetches its second input), -1: the form has no such point
                              // ns: column slices of the split layer (stream4_kernel<..., SPL>; 0: none), see below
  // dot interaction between the chains (DotArgs): at tile `inter_tile` the T slab (F x D per
  // row) becomes the R slab (D + P per row, zero padded to r_pad) the second chain reads
  int32_t inter_on, inter_tile, F, D, itself, P;
  int32_t t_off, t_ld, r_off, r_ld, r_pad, packed;
  float* g_R;
  int64_t g_ldr;
  int32_t n_bias, bias_off; // all biases: n_bias floats at `bias` -> LDS float offset bias_off
  const float* bias;
  int64_t M;
  const float* zero;       // 16 B of zeros in device memory: source of every out-of-range float4 load
                           // (an address select keeps the load unconditional; a value select would
                           // put it under divergent control flow and serialise the tile's loads)
  const float* wbase;      // the weight arena: every tile address is wbase + a 32-bit float offset, so
  uint32_t zero_off, warm_off;// a tile load is `global_load_dwordx4 v, v_off, s[wbase]` (four VALU per
                           // address); zero_off: zeros INSIDE the arena for k beyond a layer's K
  SLayer L[DRS_MAX_STREAM_LAYERS];
  SInput in[2];
  // packed form, 8 waves: one descriptor per round, read with ONE scalar load (n_table == n_tiles
  // when the launch has at most DRS_MAX_STREAM_TILES rounds, else 0: the iterator form below)
  int32_t n_table, warm_bytes;   // warm_off / warm_bytes: the arena range holding this launch's packed twins (stream3_kernel's L2 warm-up)
  int32_t tab_off, lay_off;   // LDS float offsets of the copies of tiles[] and L[] the loop reads
  STile tiles[DRS_MAX_STREAM_TILES];
};
// Column-split form of stream4_kernel (SArgs::ns = 2 | 4): `ns` workgroups share a slab of rows.  Each runs everything up
// to the split layer (steps [t0, t1) of the table: the first layer of the second chain, RMC1's 576 -> 256) for ALL of the
// slab's rows, but only `tps` of that layer's column tiles (tiles tps y ...: one pass, 4 waves x tps / 4 tiles); it
// publishes its [rows, 16 tps] piece of the layer's output slab (LDS offset `off`, leading dimension `ld`, `n` columns)
// write-through in xbuf and takes a ticket on xcnt[slab]; the last arriver fetches the other pieces and runs the
// remaining layers.  The split is over N: every output keeps its k-ordered chain -- same bits.
// A kernel argument of its own BEHIND the others: grown into SArgs, it moved tiles[], Done and XSrc inside the argument
// block, the compiler cut its scalar loads differently and the 32-row build -- 106 SGPRs, 17 more in VGPR lanes -- came
// out with a (never accessed) 36-byte private segment, i.e. a launch that needs scratch set up.
struct NSplit {
  int32_t t0, t1, tps, n, off, ld;
  float* xbuf;                // [launch rows, n] in the slab's column order
  uint32_t* xcnt;             // [slabs] arrival tickets, zero between launches
};

static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) <= 4096, "kernel arguments: 4 KB");

#define S3_LAST (1 << 16)
#define S3_BARRIER (1 << 17)
#define S3_INTERACT (1 << 18)
#define S3_ANEXT (1 << 19)        // the NEXT step reads the same layer's input slab: its operands may be prefetched
#define S3_TPW_SHIFT 20           // bits 20..22: tiles per wave of this step's layer (1 / 2 / 4)
#define S3_FIRST (1 << 23)        // first chunk of a pass: the accumulators start from zero
#define S3_OFF_0 "0"
#define S3_OFF_1 "1024"
#define S3_OFF_2 "2048"
#define S3_OFF_3 "3072"
#define S3_OFF(Q) S3_OFF_##Q

// mlp_stream8.hip -- form: 0 LDS-staged | 1 packed twins | 2 packed twins, two workgroups per CU
hipError_t launch_stream8(int form, unsigned grid, size_t lds, hipStream_t s, const SArgs& a, const Done& d, const XSrc& xs);
hipError_t stream8_set_attrs();
// mlp_stream4.hip -- the six builds: (summed input) | (two per CU) | rows per workgroup 16 / 32 | column split
hipError_t launch_stream4(bool sum1, bool two, int rows, bool split, unsigned grid, size_t lds, hipStream_t s, const SArgs& a,
                          const Done& d, const XSrc& xs, const NSplit& ns);
hipError_t stream4_set_attrs();
#ifdef DRS_TIMELINE
int tl_fetch_stream8(unsigned long long* out, int cap, int reset);
int tl_fetch_stream4(unsigned long long* out, int cap, int reset);
#endif

namespace {

// Optional in-kernel timeline (tools/mlp_timeline.py, built with -DDRS_TIMELINE into a
// separate library): wave 0 of workgroup 0 stamps the shader clock at the phase
// boundaries of every K-chunk round.  Compiled out of the product build.
#ifdef DRS_TIMELINE
__device__ unsigned long long g_tl[16384];
__device__ unsigned g_tl_n;
// stamps go to a spare 8 KB at the very end of the dynamic LDS (no global traffic while the
// kernel runs); thread 0 of workgroup 0 flushes them at the end
#define TL_SLOTS 1000
__device__ __forceinline__ void tl_stamp(unsigned long long* tl, unsigned tag, bool on) {
  if (on && threadIdx.x == 0) {
    const unsigned i = (unsigned)tl[0];
    if (i + 1 < TL_SLOTS) {
      tl[i + 1] = ((unsigned long long)tag << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull);
      tl[0] = i + 1;
    }
  }
}
#define TL_ON (blockIdx.x == 0 && blockIdx.y == 0)      // (stream4_kernel: every workgroup of slab 0 stamps, the one that signs off flushes)
#define TL(tag) tl_stamp(g_tl_lds, tag, TL_ON)
#define TL_DECL unsigned long long* g_tl_lds
#define TL_ARG , g_tl_lds
#define TL_PARAM , unsigned long long* g_tl_lds
#else
#define TL(tag)
#define TL_ARG
#define TL_PARAM
#endif

// The pairwise dots of the fused dot interaction on the matrix cores (north_star: "the feature-interaction
// batched dot ... use MFMA"): for one sample Z = T T^t with T the sample's F x D feature block; a wave
// takes a sample, lane (r, g) feeds T[r][4 s + g] as BOTH operands of MFMA step s (rows r >= F feed
// zeros), D / 4 dependent steps = one k-ordered fma chain per pair from 0, like the oracle's and like
// interact_dot_kernel's.  Lane (r, g) then holds Z[4 g + i][r], i = 0..3, and writes the pairs of the
// (strictly) lower triangle in BatchGather order behind the D dense columns.  pos(c, row) maps a column
// of a slab row to its LDS position (the kernels keep different column permutations).  The accumulator
// is a VGPR quad through inline asm: stream4_kernel must not have the compiler allocate AGPRs.
template <typename POS>
__device__ __forceinline__ void interact_pairs_mfma(const float* Ts, int t_ld, float* Rs, int r_ld, int rows, int F,
                                                    int D, int itself, float* g_R, int64_t g_ldr, int64_t m0,
                                                    int64_t M, int n_waves, int wave, int lane, POS pos) {
  const int r = lane & 15, g = lane >> 4, off = itself ? 1 : 0;
  const int nblk = (F + 15) >> 4;       // F > 16 (RM2 in dot mode: 33 features): Z in 16 x 16 blocks, lower triangle of blocks
  for (int row = wave; row < rows; row += n_waves) {
    const float* t = Ts + row * t_ld;
    for (int bi = 0; bi < nblk; ++bi)
      for (int bj = 0; bj <= bi; ++bj) {
        // A operand: features 16 bi + r (output rows), B operand: features 16 bj + r (output columns)
        const int fa = 16 * bi + r, fb = 16 * bj + r;
        const int base_a = (fa < F ? fa : 0) * D + g, base_b = (fb < F ? fb : 0) * D + g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < D; k0 += 4) {
          float va = t[pos(base_a + k0, row)], vb = t[pos(base_b + k0, row)];
          va = fa < F ? va : 0.f;
          vb = fb < F ? vb : 0.f;
          asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(va), "v"(vb));   // (s_nop: the operands were just written by VALU ops the compiler cannot see the consumer of)
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));      // the last step's results
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int fi = 16 * bi + 4 * g + i, fj = fb;
          if (fi < F && fj < F && (fj < fi || (off && fj == fi))) {
            const int c = D + fi * (fi - 1 + 2 * off) / 2 + fj;
            const float v = acc[i];
            Rs[row * r_ld + pos(c, row)] = v;
            if (g_R && m0 + row < M) g_R[(m0 + row) * g_ldr + c] = v;
          }
        }
      }
  }
}

// The argument block of the stream kernels is 3.1 KB = 50 cache lines that the host rewrites for every
// launch: each first touch is a miss of the scalar cache all the way to memory, and the compiler fetches
// fields where it first needs them -- the prologue of stream3_kernel spent 10 k cycles (4 us) in ~40
// dependent s_load / s_waitcnt pairs before its first input load (in-kernel timeline, round 3).  One
// burst of loads, one per line, all in flight together, brings the whole block into the scalar cache
// for the price of ONE miss.
__device__ __forceinline__ void kernarg_burst() {
  static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) >= 0xc40 + 4, "argument block shorter than the burst");
  const uint32_t* kp_ = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t0_, t1_;
#define S3_KL(O0, O1) "s_load_dword %0, %2, " #O0 "\n\ts_load_dword %1, %2, " #O1 "\n\t"
  asm volatile(
      S3_KL(0x0, 0x40) S3_KL(0x80, 0xc0) S3_KL(0x100, 0x140) S3_KL(0x180, 0x1c0) S3_KL(0x200, 0x240)
      S3_KL(0x280, 0x2c0) S3_KL(0x300, 0x340) S3_KL(0x380, 0x3c0) S3_KL(0x400, 0x440) S3_KL(0x480, 0x4c0)
      S3_KL(0x500, 0x540) S3_KL(0x580, 0x5c0) S3_KL(0x600, 0x640) S3_KL(0x680, 0x6c0) S3_KL(0x700, 0x740)
      S3_KL(0x780, 0x7c0) S3_KL(0x800, 0x840) S3_KL(0x880, 0x8c0) S3_KL(0x900, 0x940) S3_KL(0x980, 0x9c0)
      S3_KL(0xa00, 0xa40) S3_KL(0xa80, 0xac0) S3_KL(0xb00, 0xb40) S3_KL(0xb80, 0xbc0) S3_KL(0xc00, 0xc40)
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0_), "=&s"(t1_) : "s"(kp_));
#undef S3_KL
}

__device__ __forceinline__ int lpos(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }

template <typename F>
hipError_t set_max_lds(F kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
#ifdef DRS_TIMELINE
// (g_tl / g_tl_n exist once per translation unit)
inline int tl_fetch_here(unsigned long long* out, int cap, int reset) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_n), sizeof n) != hipSuccess) return -1;
  if (n > 16384) n = 16384;
  if ((int)n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * n) != hipSuccess) return -1;
  if (reset) { unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_n), &z, sizeof z); }
  return (int)n;
}
#endif

}  // namespace
}  // namespace drs

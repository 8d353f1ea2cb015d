// stream_kernel: the eight-wave stream kernel of the MLP chains (NCF, MT-WnD; the LDS-staged form in the lab build).
// Planning and the launch decision: mlp.hip (stream_plan, launch_chain2).
// Stream kernel: the same chain(s) of layers as chain_kernel, organised around ONE flat
// stream of weight tiles instead of per-layer passes.
//
// chain_kernel's K-chunk round is fetch -> MFMA -> stash -> barrier with every phase
// exposed (in-kernel timeline: ~1.0 k cycles of MFMA in a 2.2 k cycle round) and every
// pass of every layer starts with a cold fetch (6 x ~2.3 k cycles on RM1).  Weights do not
// depend on activations, so here the tiles W[n0:n0+128, c*64:(c+1)*64] of ALL layers form
// one sequence that is requested SIX tiles ahead of its use, across pass and layer
// boundaries (a ring of six register sets per thread, 16 VGPRs each):
//     round i:  issue global loads of tile i+6          (register set i%6)
//               MFMAs of tile i from LDS buffer i&1, interleaved with
//               the LDS stash of tile i+1 (set (i+1)%6 -> buffer (i+1)&1)
//               [epilogue of the pass: bias + activation -> next layer's LDS slab]
//               barrier
// so a round is bounded by the MFMA pipe (16 dependent MFMAs x 2 waves per SIMD), the
// loads have five rounds to land and the only cold start is the kernel's first tile.
// All layer inputs live in LDS slabs: the chains' global inputs (dense features; the
// pooled-embedding columns of the interaction buffer) are pulled in once at kernel start,
// every later activation is written there by the previous layer's epilogue.  Slab columns
// between K and the next multiple of 64 are kept zero, weight tiles read zeros beyond K,
// so the MFMA body has no selects and no branches.
// With a DotArgs the DLRM dot interaction runs between the two chains, in LDS (interact()).
// Requires K % 4 == 0 and 16-B aligned operands on every layer and the slabs to fit in
// LDS; launch_chain2 falls back to chain_kernel otherwise.
#include "mlp_stream.h"

namespace drs {
namespace {

// PK = true ("mlp_stream" 2, the default): the weight tiles come from the layers' PACKED twins
// (pack_stream_kernel below: per pass, chunk and wave, four float4 per lane = the wave's 16 MFMA
// B operands of the round, k in natural order) straight into the registers the MFMAs read --
// no LDS staging of W, no stash, and a workgroup barrier only where one layer's outputs become
// the next layer's inputs (RMC1: 5 barriers instead of 26) instead of one per 64-k chunk.
// PK = false: the LDS-staged form described above.  Same fma chains, same bits.
// NWV = waves per workgroup: 8 (a pass covers 128 output columns).
// RD3: the table-driven packed form with a ring of THREE register sets instead of six, compiled for 128
// VGPRs (four waves per SIMD): two of its workgroups share a CU, so the launches of two overlapping
// sets (MLP-bound models run one MLP stream per slot) interleave on the same SIMDs instead of
// queueing for whole CUs -- what gemm_kernel<2, 1, 2, 4> does for the wide layers.
template <bool PK, int NWV, bool RD3 = false>
__global__ __launch_bounds__(64 * NWV, RD3 ? 4 : 1) void stream_kernel(SArgs a, Done done, XSrc xs) {
  static_assert(NWV == 8, "eight waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kernarg_burst();
  constexpr int kThreads = 64 * NWV;           // (shadows the file-scope 512)
  constexpr int PASSW = 16 * NWV;              // output columns per pass
  constexpr int RD = RD3 ? 3 : 6;   // ring depth (register sets of weight tiles in flight)
  static_assert(!RD3 || (PK && NWV == 8), "the 3-deep ring: table-driven packed form only");
  constexpr int LD = 68;                       // staged W rows: 64 k + 4 pad
  const int tid = threadIdx.x;
  // `wave` as a SCALAR: everything derived from it (the wave's columns, "is my tile inside N",
  // the wave's slice of a packed tile) then runs on the scalar unit.  The per-round bookkeeping
  // was ~150 VALU instructions per wave (21 of them 32-bit multiplies), which two waves per SIMD
  // issue back to back: with MFMAs and weight loads removed the launch still took 25.6 of 34 us.
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int gs = swz(g, r);
  const uint32_t lane16 = (uint32_t)lane * 16u;   // byte offset of this lane's float4 inside a 1-KB operand block
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  float* sB = smem + a.sB_off;
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(smem + a.lds_floats);
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
#endif
  TL(1);
  const float* zero = a.zero;
  // staging role of this thread: row frow (+32 j) of the tile, floats fk..fk+3 of the chunk
  const int frow = tid >> 4, fk = (tid & 15) * 4;
  const int st_lo = (frow & 8) ? 2 : 0, st_hi = 2 - st_lo;   // swz4 of my rows (same for all j)
  float* const st_base = sB + frow * LD + fk;

  // ---- fetch iterator: six tiles ahead ----------------------------------------------------
  int f_l = 0, f_n0 = 0, f_c = 0, f_K = a.L[0].K, f_N = a.L[0].N;
  uint32_t f_woff = PK ? a.L[0].wp_off : a.L[0].w_off;
  // The tile loads are issued through inline asm and waited for with an explicit
  // s_waitcnt (DRS_WAIT_TILE): the compiler's own counter model drains the whole ring at
  // the loop header (vmcnt(0) once per trip), which costs a full miss latency every six
  // rounds.  vmcnt retires in order, so waiting for "at most 20 newer" is exact for the
  // set requested five rounds ago no matter how many stores came in between.
  // One of the four loads of a tile (rows frow + 32 j): scalar base + 32-bit offset.
  auto fetch_part = [&](f32x4 (&rb)[4], int j) {
    if (PK) {
      // tile (pass f_n0 / 128, chunk f_c) of the packed twin: 8192 floats; wave w's block of
      // 1024, float4 j of lane `lane` (always in range: the twin is padded with zeros)
      // (16 waves: waves 8..15 take the next 128-column pass of the twin, or -- beyond the layer's
      // last one, their columns do not exist -- re-read this one: the loads must be issued anyway)
      int p128 = (f_n0 >> 7) + (wave >> 3);
      p128 = p128 * 128 < f_N ? p128 : (f_n0 >> 7);
      const uint32_t tile = (uint32_t)p128 * (uint32_t)((f_K + 63) >> 6) + (uint32_t)f_c;
      // scalar base of the wave's 4-KB slice, constant per-lane offset, float4 j as the immediate
      const float* sb = a.wbase + (f_woff + tile * 8192u + (uint32_t)(wave & 7) * 1024u);
      if (j == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[0]) : "v"(lane16), "s"(sb));
      else if (j == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(rb[1]) : "v"(lane16), "s"(sb));
      else if (j == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(rb[2]) : "v"(lane16), "s"(sb));
      else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(rb[3]) : "v"(lane16), "s"(sb));
      return;
    }
    const int k = f_c * 64 + fk;
    const int row = min(f_n0 + frow + 32 * j, f_N - 1);
    uint32_t off = f_woff + (uint32_t)row * (uint32_t)f_K + (uint32_t)k;
    off = k < f_K ? off : a.zero_off;            // out-of-range k reads zeros
    const uint32_t boff = off << 2;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[j]) : "v"(boff), "s"(a.wbase));
  };
  auto fetch_advance = [&]() {                   // (uniform)
    ++f_c;
    if (f_c * 64 >= f_K) {
      f_c = 0;
      f_n0 += PASSW;
      if (f_n0 >= f_N) {
        f_n0 = 0;
        if (f_l + 1 < a.n_layers) {
          ++f_l;
          f_K = a.L[f_l].K; f_N = a.L[f_l].N; f_woff = PK ? a.L[f_l].wp_off : a.L[f_l].w_off;
        }
      }
    }
  };
  auto fetch = [&](f32x4 (&rb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch_part(rb, j);
    fetch_advance();
  };
  // swz4 by address instead of by value: the halves of a float4 go to swapped 8-B slots
  // on rows 8..15 (two ds_write_b64, no selects)
  auto stash = [&](int buf, const f32x4 (&rb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* q = st_base + (buf * 128 + 32 * j) * LD;
      *reinterpret_cast<float2*>(q + st_lo) = make_float2(rb[j][0], rb[j][1]);
      *reinterpret_cast<float2*>(q + st_hi) = make_float2(rb[j][2], rb[j][3]);
    }
  };
#define DRS_WAIT_TILE(RB, N) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]))

  auto stash_part = [&](int buf, const f32x4 (&rb)[4], int q) {
    float* p = st_base + (buf * 128 + 32 * (q >> 1)) * LD;
    if (q & 1) *reinterpret_cast<float2*>(p + st_hi) = make_float2(rb[q >> 1][2], rb[q >> 1][3]);
    else *reinterpret_cast<float2*>(p + st_lo) = make_float2(rb[q >> 1][0], rb[q >> 1][1]);
  };

  // ring of 6 register sets: tile i+6 is requested in round i and stashed in round i+5, so a
  // weight tile has five rounds to arrive (the gather of the next launch set runs beside this
  // kernel and pushes L2 misses to several microseconds)
  f32x4 rb0[4], rb1[4], rb2[4], rb3[4], rb4[4], rb5[4];
  // (table form: the tile's packed offset comes from its descriptor)
  const bool use_table = RD3 || (PK && NWV == 8 && a.n_table > 0);          // uniform
  // The round descriptors and the layer records are COPIED from the kernel-argument segment into
  // LDS by the prologue and read from there: a scalar load of a kernel argument the wave has not
  // touched yet is a cold miss all the way to HBM (the segment is written by the host for every
  // launch), and the loop touched a new 64-B line of it every few rounds -- the bare control flow of
  // RMC1's 26 rounds cost 10 us of a 34 us launch that way (0.4 us per round with every MFMA, load,
  // LDS read and barrier removed; with the arguments in HOST memory, HIP_FORCE_DEV_KERNARG=0, 49 us).
  const int n_table = a.n_table;
  const uint32_t* s_tab = reinterpret_cast<const uint32_t*>(smem + a.tab_off);
  const uint32_t* s_lay = reinterpret_cast<const uint32_t*>(smem + a.lay_off);
  auto lds_tile = [&](int i) {
    const uint4 v = *reinterpret_cast<const uint4*>(s_tab + 4 * min(i, n_table - 1));
    STile t;
    t.wp_off = __builtin_amdgcn_readfirstlane(v.x); t.a_off = __builtin_amdgcn_readfirstlane(v.y);
    t.in_ld = __builtin_amdgcn_readfirstlane(v.z); t.info = __builtin_amdgcn_readfirstlane(v.w);
    return t;
  };
  auto lds_layer = [&](int l) {
    SLayer L;
    uint32_t* d = reinterpret_cast<uint32_t*>(&L);
    const uint32_t* src = s_lay + l * (int)(sizeof(SLayer) / 4);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SLayer) / 4); ++i) d[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return L;
  };
  auto fetch_tile_wp = [&](f32x4 (&rb)[4], uint32_t wp) {
    const float* sb = a.wbase + (wp + (uint32_t)wave * 1024u);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[0]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(rb[1]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(rb[2]) : "v"(lane16), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(rb[3]) : "v"(lane16), "s"(sb));
  };
  auto fetch_tile = [&](f32x4 (&rb)[4], int i) { fetch_tile_wp(rb, a.tiles[min(i, a.n_table - 1)].wp_off); };   // (prologue: straight from the arguments)
  if (use_table) {
    fetch_tile(rb0, 0); fetch_tile(rb1, 1); fetch_tile(rb2, 2);
    if constexpr (RD == 6) { fetch_tile(rb3, 3); fetch_tile(rb4, 4); fetch_tile(rb5, 5); }
  } else {
    fetch(rb0); fetch(rb1); fetch(rb2); fetch(rb3);   // tiles 0..RD-1 (repeats past the end)
    if constexpr (RD == 6) { fetch(rb4); fetch(rb5); }
  }
  TL(2);
  // ---- chain inputs and biases -> LDS ------------------------------------------------------
  // Every load of the prologue -- the six weight tiles above, the 16-row blocks of both chain
  // inputs, the biases -- is REQUESTED before the first one is waited for: one memory round
  // trip instead of four (dense block, pooled block in two batches, biases: the in-kernel
  // timeline showed 3.9 us here, cold HBM / Infinity-Cache misses each).  A slot is 512 float4
  // (one per thread); slots [0, n0s) belong to input 0, the rest to input 1, so which input a
  // slot reads is uniform.
  {
    constexpr int PRE = RD3 ? 4 : 8;   // slots per batch (512 threads: RMC1 needs 6, RM3's 1024-wide chain 8)
    const SInput in0 = a.in[0];
    const SInput in1 = a.in[a.n_inputs > 1 ? 1 : 0];
    const int n0s = (16 * (in0.cols_pad >> 2) + kThreads - 1) / kThreads;
    const int n1s = a.n_inputs > 1 ? (16 * (in1.cols_pad >> 2) + kThreads - 1) / kThreads : 0;
    const float* base0 = in0.src;
    int64_t row00 = m0, rows0 = a.M;
    if (in0.use_xs) resolve_src(xs, in0.src, a.M, m0, &base0, &row00, &rows0);
    // biases: requested first, stored last (the engine keeps the chains' biases back to back,
    // padded to 4 floats: one flat copy; a global load in the epilogue would put a vmcnt(0)
    // = the full latency of the weight tiles just requested at the end of every pass)
    float bias_v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_v[j] = a.bias[min(tid + j * kThreads, a.n_bias - 1)];
    for (int s0 = 0; s0 < n0s + n1s; s0 += PRE) {
      float4 v[PRE], w2[PRE];
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        const int sl = s0 + j;
        if (sl < n0s + n1s) {                    // uniform
          const bool second = sl >= n0s;         // uniform
          const SInput& in = second ? in1 : in0;
          const float* base = second ? in1.src : base0;
          const int64_t row0 = second ? m0 : row00, rows = second ? a.M : rows0;
          const int qpr = in.cols_pad >> 2, total = 16 * qpr;
          const int idx = min((sl - (second ? n0s : 0)) * kThreads + tid, total - 1);
          const int row = idx / qpr, k = (idx - row * qpr) * 4;
          const int64_t grow = min(row0 + row, rows - 1);
          int64_t off = grow * in.ld + in.col0 + k;
          off = k < in.cols ? off : (int64_t)(zero - base);      // out-of-range k reads the zero page
          asm("" : "+v"(off));
          v[j] = *reinterpret_cast<const float4*>(base + off);
          if (in.col2 >= 0) {                    // uniform: the block is the SUM of two column blocks (NCF)
            int64_t off2 = grow * in.ld + in.col2 + k;
            off2 = k < in.cols ? off2 : (int64_t)(zero - base);
            asm("" : "+v"(off2));
            w2[j] = *reinterpret_cast<const float4*>(base + off2);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        const int sl = s0 + j;
        if (sl < n0s + n1s) {
          const bool second = sl >= n0s;
          const SInput& in = second ? in1 : in0;
          const int qpr = in.cols_pad >> 2, total = 16 * qpr;
          const int idx = (sl - (second ? n0s : 0)) * kThreads + tid;
          const int row = idx / qpr, k = (idx - row * qpr) * 4;
          float4 x = v[j];
          if (in.col2 >= 0) x = make_float4(x.x + w2[j].x, x.y + w2[j].y, x.z + w2[j].z, x.w + w2[j].w);
          float* dst = smem + in.lds_off + in.lds_col0;
          if (idx < total) *reinterpret_cast<float4*>(dst + row * in.lds_ld + k) = swz4(x, row);
          if (in.g_dst && idx < total && k < in.cols && m0 + row < a.M)
            *reinterpret_cast<float4*>(in.g_dst + (m0 + row) * in.g_ldd + k) = x;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + j * kThreads < a.n_bias) smem[a.bias_off + tid + j * kThreads] = bias_v[j];
    for (int i0 = 2 * kThreads; i0 < a.n_bias; i0 += kThreads)     // (more than 1024 bias words: not on any shipped config)
      if (i0 + tid < a.n_bias) smem[a.bias_off + i0 + tid] = a.bias[i0 + tid];
  }
  if (use_table) {
    const uint32_t* kp = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();   // SArgs is argument 0 (constant -> generic address space)
    uint32_t* dt = reinterpret_cast<uint32_t*>(smem + a.tab_off);
    uint32_t* dl = reinterpret_cast<uint32_t*>(smem + a.lay_off);
    for (int i = tid; i < 4 * a.n_table; i += kThreads) dt[i] = kp[offsetof(SArgs, tiles) / 4 + i];
    for (int i = tid; i < a.n_layers * (int)(sizeof(SLayer) / 4); i += kThreads) dl[i] = kp[offsetof(SArgs, L) / 4 + i];
  }
  TL(3);
  if (!PK) {
    DRS_WAIT_TILE(rb0, 0);
    stash(0, rb0);
  }
  __syncthreads();
  TL(4);

  // ---- dot interaction between the chains (one d-ordered fma chain per pair, like the
  // oracle and interact_dot_kernel: bit-identical) -------------------------------------------
  auto interact = [&]() {
    const float* Ts = smem + a.t_off;
    float* Rs = smem + a.r_off;
    const int D = a.D, W = a.r_pad;
    for (int o = tid; o < 16 * W; o += kThreads) {
      const int row = o / W, c = o - row * W;
      const float* t = Ts + row * a.t_ld;
      float v = 0.f;
      if (c < D) {
        v = t[swz(c, row)];
      } else if (c < D + a.P) {
        continue;                                 // the pairs: on the matrix cores, below
      }
      Rs[row * a.r_ld + swz(c, row)] = v;
      if (a.g_R && c < D + a.P && m0 + row < a.M) a.g_R[(m0 + row) * a.g_ldr + c] = v;
    }
    interact_pairs_mfma(Ts, a.t_ld, Rs, a.r_ld, 16, a.F, D, a.itself, a.g_R, a.g_ldr, m0, a.M, kThreads / 64,
                          tid >> 6, tid & 63, [](int c, int row) { return swz(c, row); });
    __syncthreads();
  };

  // ---- consume iterator ------------------------------------------------------------------
  int c_tile = 0;
  int c_l = 0, c_n0 = 0, c_c = 0;
  SLayer cl = a.L[0];
  int c_nch = (cl.K + 63) >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

// timing experiments ("mlp_debug") exist only in the timeline build
#ifdef DRS_TIMELINE
#define DRS_DBG_MFMA_ON (!(a.dbg & 2))
#else
#define DRS_DBG_MFMA_ON true
#endif
#define DRS_ROUND(BUF, RB_FETCH, RB_STASH)                                                        \
  {                                                                                               \
    TL(10);                                                                                       \
    if (a.inter_on && c_tile == a.inter_tile) interact();                                         \
    ++c_tile;                                                                                     \
    TL(11);                                                                                       \
    const int col = c_n0 + wave * 16 + r;                                                         \
    if (c_n0 + wave * 16 < cl.N) {                                                                \
      const float* pa = smem + cl.in_off + r * cl.in_ld + c_c * 64 + gs;                          \
      const float* pb = sB + ((BUF) * 128 + wave * 16 + r) * LD + gs;                             \
      float av[16], bv[16];                                                                       \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) { av[s] = pa[4 * s]; bv[s] = pb[4 * s]; }    \
      /* issue order, pinned: the requests of the tile six ahead; all operand reads; then the    */ \
      /* dependent MFMA chain with one LDS write of the stash in the shadow of every second MFMA.*/ \
      /* Tried and dropped (r2, each 3-5 % slower on RMC1 / W&D / NCF): the requests spread INTO  */ \
      /* the chain (anything between two MFMAs on one accumulator delays the dependent issue),   */ \
      /* and the two waves of a SIMD running request / multiply halves in opposite order.        */ \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB_FETCH, q);                      \
      DRS_WAIT_TILE(RB_STASH, 20);                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                             \
        if (DRS_DBG_MFMA_ON) {                                                                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * q], bv[2 * q], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * q + 1], bv[2 * q + 1], acc, 0, 0, 0);   \
        }                                                                                         \
        stash_part((BUF) ^ 1, RB_STASH, q);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB_FETCH, q);                      \
      DRS_WAIT_TILE(RB_STASH, 20);                                                                \
      stash((BUF) ^ 1, RB_STASH);                                                                 \
    }                                                                                             \
    fetch_advance();                                                                              \
    TL(12);                                                                                       \
    if (c_c == c_nch - 1) {                                                                       \
      if (col < (cl.out_off >= 0 ? cl.out_pad : cl.N)) {                                          \
        const float bias_v = smem[cl.b_off + min(col, cl.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < cl.N ? act_apply(acc[i] + bias_v, cl.act) : 0.f;                  \
          if (cl.out_off >= 0) smem[cl.out_off + row * cl.out_ld + swz(col + cl.out_col0, row)] = v; \
          if (cl.g_out && col < cl.N && m0 + row < a.M) {                                         \
            float* dstg = cl.g_out + (m0 + row) * cl.g_ld + col;                                  \
            if (cl.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
      c_c = 0;                                                                                    \
      c_n0 += 128;                                                                                \
      if (c_n0 >= cl.N) {                                                                         \
        c_n0 = 0;                                                                                 \
        if (c_l + 1 < a.n_layers) { ++c_l; cl = a.L[c_l]; c_nch = (cl.K + 63) >> 6; }             \
      }                                                                                           \
    } else {                                                                                      \
      ++c_c;                                                                                      \
    }                                                                                             \
    TL(13);                                                                                       \
    __syncthreads();                                                                              \
    TL(14);                                                                                       \
  }

// Decomposition of the packed launch by removal (RMC1, 2 048 rows, 34 us): no round loop at all
// (prologue + hand-off only) 14 us; rounds with MFMAs, weight loads, LDS reads and barriers removed
// +10 us; MFMAs + weight loads +9 us; LDS reads + barriers +1 us.  Tried against the +10 us, each
// with no change of the total: the wave index as a scalar and the operand row hoisted per layer
// (fewer VALU), the table-driven rounds above with the rare blocks out of line (68 instructions
// between two MFMA groups instead of 700), the descriptors and layer records in LDS instead of
// the kernel-argument segment (kept: with the arguments in HOST memory, HIP_FORCE_DEV_KERNARG=0,
// the launch takes 49 us, so argument reads are not free), sixteen waves, skewing, prefetching
// the activation operands.
// 32-row workgroups (two activation tiles per weight operand set, two accumulators per wave) were
// built and measured as well: bit-identical, but the launch takes 60 us on 64 CUs instead of 33 us
// on 128 -- a round's time follows its MFMA count, i.e. with two waves per SIMD the rounds run at
// ~37 cycles per MFMA and SIMD, close to the pipe's 32; RMC1 -19 %, NCF -17 %, only RM3 at batch
// 512 +2 %.  What bounds the launch is 16 rows per CU on half the CUs plus ~14 us of fixed cost,
// not the round.
// Where a packed round's time goes (in-kernel timeline, RMC1): the 16 MFMAs of the two waves of a
// SIMD run as one phase at the pipe's rate (32 MFMAs in ~1 100 cycles) and the per-round
// bookkeeping of both (~1 000 cycles: tile addresses, iterator state, epilogue tests) as another
// -- a lone wave issues an fp32 MFMA only every ~75 cycles (also measured in din.hip's
// recurrence), so skewing the two waves against each other buys nothing (tried: s_sleep on waves
// 4..7 after every barrier, 0..1 000 cycles: 33.3-33.7 us throughout), and prefetching the next
// round's activation operands under the MFMAs neither (34.4 us).  The lever left is more MFMAs
// per round and wave (two column tiles sharing the activation operands) or four waves per SIMD.
// Packed form: round i waits for ITS set (requested six rounds ago: at most the 5 x 4 loads of the
// newer sets may still be in flight), reads the 16 activation operands from LDS, runs the
// dependent chain on the set's registers, and only then re-requests into them (tile i + 6).
// Every wave issues its 4 loads every round, also when its 16 columns lie beyond the layer's N
// (zeros in the twin), so the in-order vmcnt arithmetic holds for all of them.
#define DRS_ROUND_PK(RB, NEWER)                                                                   \
  {                                                                                               \
    TL(10);                                                                                       \
    if (a.inter_on && c_tile == a.inter_tile) interact();                                         \
    ++c_tile;                                                                                     \
    const int col = c_n0 + wave * 16 + r;                                                         \
    DRS_WAIT_TILE(RB, NEWER);                                                                     \
    TL(11);                                                                                       \
    if (c_n0 + wave * 16 < cl.N) {                                                                \
      const float* pa = pa_layer + c_c * 64;                                                      \
      float av[16];                                                                               \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) av[s] = pa[4 * s];                           \
      _Pragma("unroll") for (int s = 0; s < 16; ++s)                                              \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    TL(12);                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) fetch_part(RB, q);                              \
    fetch_advance();                                                                              \
    TL(13);                                                                                       \
    bool layer_done = false;                                                                      \
    if (c_c == c_nch - 1) {                                                                       \
      if (col < (cl.out_off >= 0 ? cl.out_pad : cl.N)) {                                          \
        const float bias_v = smem[cl.b_off + min(col, cl.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < cl.N ? act_apply(acc[i] + bias_v, cl.act) : 0.f;                  \
          if (cl.out_off >= 0) smem[cl.out_off + row * cl.out_ld + swz(col + cl.out_col0, row)] = v; \
          if (cl.g_out && col < cl.N && m0 + row < a.M) {                                         \
            float* dstg = cl.g_out + (m0 + row) * cl.g_ld + col;                                  \
            if (cl.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
      c_c = 0;                                                                                    \
      c_n0 += PASSW;                                                                              \
      if (c_n0 >= cl.N) {                                                                         \
        c_n0 = 0;                                                                                 \
        layer_done = true;                                                                        \
        if (c_l + 1 < a.n_layers) { ++c_l; cl = a.L[c_l]; c_nch = (cl.K + 63) >> 6; }             \
        pa_layer = smem + cl.in_off + r * cl.in_ld + gs;                                          \
      }                                                                                           \
    } else {                                                                                      \
      ++c_c;                                                                                      \
    }                                                                                             \
    /* the only hand-off between waves: a layer's outputs become the next layer's inputs */       \
    if (layer_done) __syncthreads();                                                              \
    TL(14);                                                                                       \
  }

  const float* pa_layer = smem + cl.in_off + r * cl.in_ld + gs;   // (iterator form: per layer, not per round)
  // ---- packed form, table-driven: one scalar descriptor load per round ---------------------
#define DRS_ROUND_T(RB)                                                                           \
  {                                                                                               \
    /* The control chain of a round (descriptor of the next round, packed offset of the tile six  */ \
    /* ahead: LDS read -> readfirstlane -> scalar address) is issued INSIDE the MFMA chain, in     */ \
    /* the ~60 idle issue cycles between two dependent MFMAs: a wave issues in order, so behind    */ \
    /* the chain it costs its full latency every round.                                            */ \
    const uint4 tn_raw = *reinterpret_cast<const uint4*>(s_tab + 4 * min(ti + 1, n_table - 1));   \
    const uint32_t wp_raw = s_tab[4 * min(ti + RD, n_table - 1)];                                 \
    if (__builtin_expect((t.info & (1 << 18)) != 0, 0)) interact();                               \
    const int ncols = t.info & 0xffff;                                                            \
    const bool act_now = wave * 16 < ncols;                                                       \
    if constexpr (RD == 3) { DRS_WAIT_TILE(RB, 8); } else { DRS_WAIT_TILE(RB, 20); }            \
    float av[16];                                                                                 \
    if (__builtin_expect(act_now, 1)) {                                                           \
      const float* pa = smem + t.a_off + r * t.in_ld + gs;                                        \
      _Pragma("unroll") for (int s = 0; s < 16; ++s) av[s] = pa[4 * s];                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                               \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    const uint32_t wp6 = (uint32_t)__builtin_amdgcn_readfirstlane(wp_raw);                        \
    STile tn;                                                                                     \
    tn.wp_off = __builtin_amdgcn_readfirstlane(tn_raw.x); tn.a_off = __builtin_amdgcn_readfirstlane(tn_raw.y); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (__builtin_expect(act_now, 1)) {                                                           \
      _Pragma("unroll") for (int s = 4; s < 8; ++s)                                               \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    tn.in_ld = __builtin_amdgcn_readfirstlane(tn_raw.z); tn.info = __builtin_amdgcn_readfirstlane(tn_raw.w); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (__builtin_expect(act_now, 1)) {                                                           \
      _Pragma("unroll") for (int s = 8; s < 16; ++s)                                              \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], RB[s >> 2][s & 3], acc, 0, 0, 0);       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    fetch_tile_wp(RB, wp6);                                                                       \
    if (__builtin_expect((t.info & (1 << 16)) != 0, 0)) {      /* last chunk of the pass */        \
      const SLayer el = lds_layer((t.info >> 24) & 0xff);                                         \
      const int col = el.N - ncols + wave * 16 + r;                                               \
      if (col < (el.out_off >= 0 ? el.out_pad : el.N)) {                                          \
        const float bias_v = smem[el.b_off + min(col, el.N - 1)];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
          const int row = g * 4 + i;                                                              \
          const float v = col < el.N ? act_apply(acc[i] + bias_v, el.act) : 0.f;                  \
          if (el.out_off >= 0) smem[el.out_off + row * el.out_ld + swz(col + el.out_col0, row)] = v; \
          if (el.g_out && col < el.N && m0 + row < a.M) {                                         \
            float* dstg = el.g_out + (m0 + row) * el.g_ld + col;                                  \
            if (el.g_sc1) __hip_atomic_store(dstg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\
            else *dstg = v;                                                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
    }                                                                                             \
    if (__builtin_expect((t.info & (1 << 17)) != 0, 0)) __syncthreads();   /* a layer's outputs -> the next layer's inputs */ \
    t = tn;                                                                                       \
    ++ti;                                                                                         \
  }
  if (use_table) {
    int ti = 0;
    STile t = lds_tile(0);
    for (int i = 0; i < n_table; i += RD) {
      DRS_ROUND_T(rb0)
      if (i + 1 >= n_table) break;
      DRS_ROUND_T(rb1)
      if (i + 2 >= n_table) break;
      DRS_ROUND_T(rb2)
      if constexpr (RD == 3) continue;
      if (i + 3 >= n_table) break;
      DRS_ROUND_T(rb3)
      if (i + 4 >= n_table) break;
      DRS_ROUND_T(rb4)
      if (i + 5 >= n_table) break;
      DRS_ROUND_T(rb5)
    }
  } else if constexpr (RD3) {
    // (launched only with a table)
  } else
#undef DRS_ROUND_T
  // this lane's activation operand row inside the current layer's input slab (per layer, not per round)
  if (PK && RD == 4) {
    for (int i = 0; i < a.n_tiles; i += 4) {
      DRS_ROUND_PK(rb0, 12)
      if (i + 1 >= a.n_tiles) break;
      DRS_ROUND_PK(rb1, 12)
      if (i + 2 >= a.n_tiles) break;
      DRS_ROUND_PK(rb2, 12)
      if (i + 3 >= a.n_tiles) break;
      DRS_ROUND_PK(rb3, 12)
    }
  } else if (PK) {
    for (int i = 0; i < a.n_tiles; i += 6) {
      DRS_ROUND_PK(rb0, 20)
      if (i + 1 >= a.n_tiles) break;
      DRS_ROUND_PK(rb1, 20)
      if (i + 2 >= a.n_tiles) break;
      DRS_ROUND_PK(rb2, 20)
      if (i + 3 >= a.n_tiles) break;
      DRS_ROUND_PK(rb3, 20)
      if (i + 4 >= a.n_tiles) break;
      DRS_ROUND_PK(rb4, 20)
      if (i + 5 >= a.n_tiles) break;
      DRS_ROUND_PK(rb5, 20)
    }
  } else
  for (int i = 0; i < a.n_tiles; i += 6) {
    DRS_ROUND(0, rb0, rb1)
    if (i + 1 >= a.n_tiles) break;
    DRS_ROUND(1, rb1, rb2)
    if (i + 2 >= a.n_tiles) break;
    DRS_ROUND(0, rb2, rb3)
    if (i + 3 >= a.n_tiles) break;
    DRS_ROUND(1, rb3, rb4)
    if (i + 4 >= a.n_tiles) break;
    DRS_ROUND(0, rb4, rb5)
    if (i + 5 >= a.n_tiles) break;
    DRS_ROUND(1, rb5, rb0)
  }
#undef DRS_ROUND
#undef DRS_ROUND_PK
#undef DRS_WAIT_TILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing requests
  TL(20);
  signal_done(done, gridDim.x, smem);
#ifdef DRS_TIMELINE
  TL(21);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
}

}  // namespace

hipError_t launch_stream8(int form, unsigned grid, size_t lds, hipStream_t s, const SArgs& a, const Done& d, const XSrc& xs) {
  if (form == 2) hipLaunchKernelGGL((stream_kernel<true, 8, true>), dim3(grid), dim3(512), lds, s, a, d, xs);
  else if (form == 1) hipLaunchKernelGGL((stream_kernel<true, 8>), dim3(grid), dim3(512), lds, s, a, d, xs);
  else hipLaunchKernelGGL((stream_kernel<false, 8>), dim3(grid), dim3(512), lds, s, a, d, xs);
  return hipGetLastError();
}

hipError_t stream8_set_attrs() {
  hipError_t e = set_max_lds(stream_kernel<false, 8>);
  if (e == hipSuccess) e = set_max_lds(stream_kernel<true, 8>);
  if (e == hipSuccess) e = set_max_lds(stream_kernel<true, 8, true>);
  return e;
}

#ifdef DRS_TIMELINE
int tl_fetch_stream8(unsigned long long* out, int cap, int reset) { return tl_fetch_here(out, cap, reset); }
#endif

}  // namespace drs

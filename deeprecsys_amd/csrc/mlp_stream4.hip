// stream4_kernel: the four-wave stream kernel of the MLP chains (DLRM, W&D, DIEN, DIN) -- every (layer, pass) ONE hand-laid
// instruction stream (seg_asm.inc), ring and accumulators in AGPRs; 16 / 32 rows per workgroup, two per CU, column split.
// Planning and the launch decision: mlp.hip (stream_plan, launch_chain2).
#include "mlp_stream.h"

namespace drs {
namespace {

// ---------------------------------------------------------------------------
// The four-wave forms (round 3's stream3_kernel -- removed in round 5, when stream4_kernel below had overtaken it on
// every launch size: 5-7 query sets +6-9 % queries/s, profiles/r05_stream3_vs_stream4/ -- and stream4_kernel, which
// runs the same step table): the packed form re-cut around what the round-3 microbenchmarks (tools/ubench/) say
// about the fp32 matrix pipe of a SIMD:
//   * ONE wave keeps it busy: a dependent v_mfma_f32_16x16x4_f32 chain issues every 35 cycles, two or
//     four independent chains every 33 (the pipe's rate is 32) -- a second wave per SIMD adds nothing;
//   * a global_load_dwordx4 every 4 MFMAs and a ds_read_b128 every 8, placed BETWEEN the MFMAs, cost
//     1-2 % -- whereas the same instructions issued as a block before or after a round's MFMAs (what
//     the 8-wave forms do) leave the pipe idle for their whole issue + wait time (the in-kernel
//     timeline of the 8-wave form: ~1 100 cycles of MFMA and ~1 000 cycles of everything else per round);
//   * loads issued with EXEC = 0 take part in vmcnt like any other (tools/ubench/masked_vmcnt.hip).
// So: a workgroup is FOUR waves (one per SIMD) and a wave's instruction stream is one unbroken run
// of MFMAs with everything else in their shadow.
//   * a step = (pass, 64-k chunk); a wave owns TPW = 1 / 2 / 4 adjacent 16-column tiles of the pass
//     (a pass covers 4 TPW tiles; TPW by the layer's width), each its own accumulator, all fed by the
//     SAME activation operands: four ds_read_b128 per step and wave, fetched one step ahead into a
//     second register set (slabs keep, inside every 16-column block, column k at position
//     4 (k mod 4) + (k div 4): lane (r, g) finds the operands of four consecutive MFMA steps side by
//     side; rows are 64 m + 8 floats apart, which makes the b128 reads conflict-free);
//   * the weights of step i + RD are requested while step i runs: a ring slot is 4 tiles x 4 float4;
//     the four float4 of k-group q (MFMA steps 4q .. 4q+3) are reloaded right after the q-th quarter
//     of the step has consumed them, so the loads are spread evenly over the step and
//     `s_waitcnt vmcnt(16 (RD-1) + 12)` in front of every quarter is exact (every wave issues exactly
//     16 loads per step: tiles it does not own are requested with EXEC = 0);
//   * one descriptor per step (STile), the next RD of them in scalar registers;
//   * no asm block with register outputs sits under a branch: the compiler then never has to merge
//     two versions of a ring register (it did so with copies -- of registers whose loads were still
//     in flight -- in the first version of this kernel).
// Same packed twins, same k-ordered fma chains, same bits as every other form.

// stream4_kernel ("mlp_stream" 4): the 4-wave form with every SEGMENT -- all 64-k chunks of one
// (layer, pass) for the 1 / 2 / 4 tiles a wave owns -- run by ONE asm statement (seg_asm.inc, generated
// by tools/gen_seg_asm.py): an unbroken run of MFMAs with the weight reloads, the operand prefetch and
// the loop control between them, no per-step descriptor decode, no EXEC masks (a tile a wave does not
// own is requested from the address of one it owns and its results are dropped by the epilogue).  The
// ring, the operands and the accumulators live in AGPRs under fixed names; the C++ around the
// statements (prologue, epilogues, interaction, hand-off) never touches an AGPR -- the Makefile checks
// the generated ISA for that.  Chunk 0 of the NEXT segment is requested while a segment's last chunk
// runs, so a layer boundary costs an epilogue and a barrier, not a memory round trip.
#include "seg_asm.inc"
// SUM1: the second input is the sum of two column blocks (NCF) -- a template parameter because the third
// staging array costs 32 VGPRs, and at 280 registers per wave instead of 312 a SIMD that hosts one of
// this kernel's waves still has room for two of the gather's (104 each) instead of one.
// TWO: compiled for 256 registers per wave (the input staging arrays halved), so that two workgroups
// share a CU -- what the MLP-bound models want (see stream_kernel's RD3 form).
// R: 16-row slabs per workgroup (1 | 2).  R = 2: a workgroup owns 32 rows as two halves that share every
// weight operand -- twice the MFMAs per byte of weights streamed from L2 and per fixed cost of a
// workgroup; taken for launches of many rows whose slabs still fit LDS ("mlp_rows32").
// SPL: the column-split form (SArgs::ns): blockIdx.x = slab of rows * ns + column slice.  Consecutive workgroups go to
// consecutive XCDs, so slice y of every slab runs on the XCDs k with k % ns == y: an XCD's L2 holds only its slice of
// the split layer's weights (a speed matter only: nothing depends on the placement).
template <bool SUM1, bool TWO, int R = 1, bool SPL = false>
__global__ __launch_bounds__(256, TWO ? 2 : 1) void stream4_kernel(SArgs a, Done done, XSrc xs, NSplit sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kThreads = 256;
  if constexpr (SPL) {          // (NSplit's line of the argument block rides on the burst below)
    static_assert(sizeof(SArgs) + sizeof(Done) + sizeof(XSrc) == 0xd00, "offset of the NSplit argument");
    uint32_t t_;
    asm volatile("s_load_dword %0, %1, 0xd00" : "=&s"(t_) : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
  }
  kernarg_burst();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  static_assert(R == 1 || (R == 2 && !TWO && !SUM1), "32-row form: one workgroup per CU, no summed input");
  static_assert(!SPL || (!TWO && !SUM1), "column-split form: one workgroup per CU, no summed input");
  const int ns_y = SPL ? (int)(blockIdx.x % (unsigned)a.ns) : 0;            // my column slice of the split layer
  const unsigned slab = SPL ? blockIdx.x / (unsigned)a.ns : blockIdx.x;     // my slab of 16 R rows
  const int64_t m0 = (int64_t)slab * (16 * R);
  // stores of the chains' outputs to GLOBAL memory: one workgroup per slab makes them (slice 0 before the split layer,
  // the last arriver behind it)
  bool gw = !SPL || ns_y == 0;
#ifdef DRS_TIMELINE
  unsigned long long* g_tl_lds = reinterpret_cast<unsigned long long*>(smem + a.lds_floats);
  if (threadIdx.x == 0) g_tl_lds[0] = 0;
  const bool tl_on = slab == 0;
#undef TL_ON
#define TL_ON tl_on
#endif
  TL(1);
  const float* zero = a.zero;
  const int n_table = a.n_table;
  const uint32_t* s_tab = reinterpret_cast<const uint32_t*>(smem + a.tab_off);
  const uint32_t* s_lay = reinterpret_cast<const uint32_t*>(smem + a.lay_off);
  const float* const wbase = a.wbase;
  // A wave's tiles in a segment: byte offsets (from the arena) of their 4-KB blocks in chunk 0, + 16 lane;
  // nex = how many of its tpw tiles exist in the twin (the others are requested from tile 0's address)
  struct Seg { uint32_t off[4]; int nex, tpw, nch; };
  // (tadd: the split layer's steps name slice 0's tiles; slice y works ns_tps y tiles further on)
  auto tadd_of = [&](int i) { return SPL && i >= sp.t0 && i < sp.t1 ? ns_y * sp.tps : 0; };
  auto seg_of = [&](uint32_t wp_off, int pstride, int info, int tadd) {
    Seg q;
    q.tpw = (info >> S3_TPW_SHIFT) & 7;
    q.nch = pstride >> 13;
    const int tile0 = (info & 0xff) + tadd, ntl = (info >> 8) & 0xff;
    const int t0 = tile0 + q.tpw * wave;
    q.nex = min(max(ntl - t0, 0), q.tpw);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = q.nex > 0 ? t0 + min(j, q.nex - 1) : 0;
      q.off[j] = (wp_off + (uint32_t)(t >> 3) * (uint32_t)pstride + (uint32_t)(t & 7) * 1024u) * 4u + (uint32_t)lane * 16u;
    }
    return q;
  };
  auto prefetch = [&](const Seg& q, int slot) {
    if constexpr (R == 2) {
      if (slot)
        asm volatile(SEG2_PREFETCH1_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                     : "memory", SEG2_AGPR_CLOBBER);
      else
        asm volatile(SEG2_PREFETCH0_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                     : "memory", SEG2_AGPR_CLOBBER);
    } else if (slot)
      asm volatile(SEG_PREFETCH1_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                   : "memory", SEG_AGPR_CLOBBER);
    else
      asm volatile(SEG_PREFETCH0_ASM :: "v"(q.off[0]), "v"(q.off[1]), "v"(q.off[2]), "v"(q.off[3]), "s"(wbase)
                   : "memory", SEG_AGPR_CLOBBER);
  };
  // ---- prologue: ONE memory round trip.  The chain inputs (the critical path: cold misses all the
  // way to HBM), the biases and the descriptor table are requested first, the weights of the first
  // RD steps right behind them; nothing is waited for before all of it is in flight.
  // A thread's role in the input copies is fixed: row tid / TPR, columns 4 (tid % TPR) + CG j -- no
  // division, one 64-bit row pointer per input.
  constexpr int TPR = kThreads / (16 * R), CG = 4 * TPR;   // threads per row; columns one pass of them covers (64 | 128)
  constexpr int PB = (TWO ? 256 : 512) / CG;                   // column groups per input and batch (512 columns)
  const int prow = tid / TPR, pk0 = (tid % TPR) * 4;
  const SInput& in0 = a.in[0];
  const SInput& in1 = a.in[a.n_inputs > 1 ? 1 : 0];
  const int nj0 = (in0.cols_pad + CG - 1) / CG, nj1 = a.n_inputs > 1 ? (in1.cols_pad + CG - 1) / CG : 0;
  const float* base0 = in0.src;
  int64_t row00 = m0, rows0 = a.M;
  if (in0.use_xs) resolve_src(xs, in0.src, a.M, m0, &base0, &row00, &rows0);
  const float* const rp0 = base0 + min(row00 + prow, rows0 - 1) * in0.ld + in0.col0;
  const float* const rp1 = in1.src + min(m0 + prow, a.M - 1) * in1.ld + in1.col0;
  const float* const rp2 = in1.src + min(m0 + prow, a.M - 1) * in1.ld + (in1.col2 >= 0 ? in1.col2 : in1.col0);
  const int cols0 = in0.cols, cols1 = in1.cols, cpad0 = in0.cols_pad, cpad1 = in1.cols_pad;
  constexpr bool sum1 = SUM1;
  float* const ld0 = smem + in0.lds_off + prow * in0.lds_ld;
  float* const ld1 = smem + in1.lds_off + prow * in1.lds_ld;
  const int lc0 = in0.lds_col0 + pk0, lc1 = in1.lds_col0 + pk0;
  float* const gd1 = in1.g_dst && m0 + prow < a.M && !SPL ? in1.g_dst + (m0 + prow) * in1.g_ldd : nullptr;
  // (a load beyond the block's real columns reads the zero page: an address select keeps it unconditional)
  auto issue = [&](const float* rp, int cols, int jb, int nj, float4 (&v)[PB]) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (jb + j < nj) {                         // uniform
        const int k = pk0 + CG * (jb + j);
        int64_t off = k < cols ? (int64_t)k : (int64_t)(zero - rp);   // (offset, not pointer, select: the load stays a global_load)
        asm("" : "+v"(off));
        v[j] = *reinterpret_cast<const float4*>(rp + off);
      }
  };
  auto store = [&](float* ld, int lc, int cpad, int jb, int nj, const float4 (&v)[PB]) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (jb + j < nj) {
        const int k = pk0 + CG * (jb + j);
        if (k < cpad) {
          // columns c .. c+3 (c a multiple of 4) sit 4 floats apart inside their 16-column block
          const int c = lc + CG * (jb + j);
          float* dst = ld + ((c & ~15) | ((c >> 2) & 3));
          dst[0] = v[j].x; dst[4] = v[j].y; dst[8] = v[j].z; dst[12] = v[j].w;
        }
      }
  };
  const uint32_t* kp = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();   // SArgs is argument 0
  uint32_t* dt = reinterpret_cast<uint32_t*>(smem + a.tab_off);
  uint32_t* dl = reinterpret_cast<uint32_t*>(smem + a.lay_off);
  const int n_tab_w = 4 * n_table, n_lay_w = a.n_layers * (int)(sizeof(SLayer) / 4);
  float4 pv0[PB], pv1[PB], pv2[PB];
  // early start ("mlp_early", plain 16-row form only): the second input -- the gather's pooled rows -- is fetched at the
  // first step of the second chain, once the gather's flag has been seen; everything before runs beside the gather
#ifdef DRS_LAB
  constexpr bool kCanDefer = !SUM1 && !TWO && R == 1 && !SPL;
#else
  constexpr bool kCanDefer = false;          // ("mlp_early" lost its measurement: the late fetch exists in the lab build only)
#endif
  const bool defer1 = kCanDefer && done.wait_flag != nullptr && a.wait_tile > 0;   // (uniform)
  issue(rp0, cols0, 0, nj0, pv0);
  if (!defer1) issue(rp1, cols1, 0, nj1, pv1);
  if constexpr (sum1) issue(rp2, cols1, 0, nj1, pv2);
  // biases, descriptors and layer records ride on the same round trip
  constexpr int NBV = 1024 / kThreads, NTV = 512 / kThreads;
  float bias_v[NBV];
  uint32_t tabv[NTV], layv[NTV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) bias_v[j] = a.bias[min(tid + j * kThreads, a.n_bias - 1)];
#pragma unroll
  for (int j = 0; j < NTV; ++j) {
    tabv[j] = kp[offsetof(SArgs, tiles) / 4 + min(tid + j * kThreads, n_tab_w - 1)];
    layv[j] = kp[offsetof(SArgs, L) / 4 + min(tid + j * kThreads, n_lay_w - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
  TL(2);
  // L2 warm-up (see stream3_kernel): one slice of the launch's packed weights per workgroup, fire and
  // forget, into the odd ring slot's registers (every later request retires after these)
  {
    const uint32_t nx = (gridDim.x + 7u) >> 3, rank = blockIdx.x >> 3;
    const uint32_t bytes = (uint32_t)a.warm_bytes;               // a multiple of 4096
    const uint32_t slice = ((bytes / nx) + 4095u) & ~4095u;
    const uint32_t o0 = rank * slice + (uint32_t)tid * 16u, last = bytes - 16u;
    const float* wb = wbase + a.warm_off;
#define S4_WARM(R, I)                                                                             \
    { const uint32_t o_ = min(o0 + (I) * 4096u, last);                                            \
      asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(o_), "s"(wb) : "memory", SEG_AGPR_CLOBBER); }
    S4_WARM("a[80:83]", 0) S4_WARM("a[84:87]", 1) S4_WARM("a[88:91]", 2) S4_WARM("a[92:95]", 3)
    S4_WARM("a[96:99]", 4) S4_WARM("a[100:103]", 5) S4_WARM("a[104:107]", 6) S4_WARM("a[108:111]", 7)
    S4_WARM("a[112:115]", 8) S4_WARM("a[116:119]", 9) S4_WARM("a[120:123]", 10) S4_WARM("a[124:127]", 11)
    S4_WARM("a[128:131]", 12) S4_WARM("a[132:135]", 13) S4_WARM("a[136:139]", 14) S4_WARM("a[140:143]", 15)
#undef S4_WARM
  }
  // chunk 0 of the first segment (descriptor straight from the arguments: its LDS copy is not there yet)
  {
    const STile e0 = a.tiles[0];
    prefetch(seg_of(e0.wp_off, e0.in_ld, e0.info, tadd_of(0)), 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  store(ld0, lc0, cpad0, 0, nj0, pv0);
  if constexpr (sum1) {
#pragma unroll
    for (int j = 0; j < PB; ++j)
      pv1[j] = make_float4(pv1[j].x + pv2[j].x, pv1[j].y + pv2[j].y, pv1[j].z + pv2[j].z, pv1[j].w + pv2[j].w);
  }
  if (!defer1) store(ld1, lc1, cpad1, 0, nj1, pv1);
  if (gd1) {                                     // NCF: the summed block is also kept in global memory
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (j < nj1 && pk0 + CG * j < cols1) *reinterpret_cast<float4*>(gd1 + pk0 + CG * j) = pv1[j];
  }
#pragma unroll
  for (int j = 0; j < NBV; ++j)
    if (tid + j * kThreads < a.n_bias) smem[a.bias_off + tid + j * kThreads] = bias_v[j];
#pragma unroll
  for (int j = 0; j < NTV; ++j) {
    if (tid + j * kThreads < n_tab_w) dt[tid + j * kThreads] = tabv[j];
    if (tid + j * kThreads < n_lay_w) dl[tid + j * kThreads] = layv[j];
  }
  // (inputs wider than 8 x 64 columns: further batches, one round trip each)
  for (int jb = PB; jb < nj0; jb += PB) { issue(rp0, cols0, jb, nj0, pv0); store(ld0, lc0, cpad0, jb, nj0, pv0); }
  for (int jb = PB; jb < (defer1 ? 0 : nj1); jb += PB) {
    issue(rp1, cols1, jb, nj1, pv1);
    if constexpr (sum1) {
      issue(rp2, cols1, jb, nj1, pv2);
#pragma unroll
      for (int j = 0; j < PB; ++j)
        pv1[j] = make_float4(pv1[j].x + pv2[j].x, pv1[j].y + pv2[j].y, pv1[j].z + pv2[j].z, pv1[j].w + pv2[j].w);
    }
    store(ld1, lc1, cpad1, jb, nj1, pv1);
    if (gd1) {
#pragma unroll
      for (int j = 0; j < PB; ++j)
        if (jb + j < nj1 && pk0 + CG * (jb + j) < cols1) *reinterpret_cast<float4*>(gd1 + pk0 + CG * (jb + j)) = pv1[j];
    }
  }
  for (int i0 = 1024; i0 < a.n_bias; i0 += kThreads)     // (more than 1024 bias words: not on any shipped config)
    if (i0 + tid < a.n_bias) smem[a.bias_off + i0 + tid] = a.bias[i0 + tid];
  for (int i = tid + 512; i < n_lay_w; i += kThreads) dl[i] = kp[offsetof(SArgs, L) / 4 + i];
  TL(3);
  __syncthreads();
  TL(4);

  // dot interaction between the chains: as stream_kernel's, on this form's slab layout
  auto interact = [&]() {
    const float* Ts = smem + a.t_off;
    float* Rs = smem + a.r_off;
    const int D = a.D, W = a.r_pad;
    for (int o = tid; o < 16 * R * W; o += kThreads) {
      const int row = o / W, c = o - row * W;
      const float* t = Ts + row * a.t_ld;
      float v = 0.f;
      if (c < D) {
        v = t[lpos(c)];
      } else if (c < D + a.P) {
        continue;                                 // the pairs: on the matrix cores, below
      }
      Rs[row * a.r_ld + lpos(c)] = v;
      if (a.g_R && gw && c < D + a.P && m0 + row < a.M) a.g_R[(m0 + row) * a.g_ldr + c] = v;
    }
    interact_pairs_mfma(Ts, a.t_ld, Rs, a.r_ld, 16 * R, a.F, D, a.itself, gw ? a.g_R : nullptr, a.g_ldr, m0, a.M, kThreads / 64,
                          tid >> 6, tid & 63, [](int c, int) { return lpos(c); });
    __syncthreads();
  };

  // the fields of a layer record the epilogue needs, from its LDS copy
  struct Epi { int N, act, out_off, out_ld, out_pad, out_col0, b_off, g_sc1; float* g_out; int64_t g_ld; };
  auto lds_epi = [&](int l) {
    const uint32_t* src = s_lay + l * (int)(sizeof(SLayer) / 4);
    auto w = [&](size_t byte_off) { return (int)__builtin_amdgcn_readfirstlane(src[byte_off / 4]); };
    Epi e;
    e.N = w(offsetof(SLayer, N)); e.act = w(offsetof(SLayer, act));
    e.out_off = w(offsetof(SLayer, out_off)); e.out_ld = w(offsetof(SLayer, out_ld));
    e.out_pad = w(offsetof(SLayer, out_pad)); e.out_col0 = w(offsetof(SLayer, out_col0));
    e.b_off = w(offsetof(SLayer, b_off)); e.g_sc1 = w(offsetof(SLayer, g_sc1));
    const uint64_t glo = (uint32_t)w(offsetof(SLayer, g_out)), ghi = (uint32_t)w(offsetof(SLayer, g_out) + 4);
    e.g_out = reinterpret_cast<float*>(glo | (ghi << 32));
    const uint64_t llo = (uint32_t)w(offsetof(SLayer, g_ld)), lhi = (uint32_t)w(offsetof(SLayer, g_ld) + 4);
    e.g_ld = (int64_t)(llo | (lhi << 32));
    return e;
  };
  // Epilogue of one tile: bias + activation -> the next layer's slab (columns past N inside the pad
  // are zero filled) and / or global memory.  `lim`: columns that exist in the slab; `dst`: this lane's
  // slab address of (row 4 g, its column); the four rows of a lane are out_ld apart.
  auto epilogue = [&](const Epi& el, const float (&acc)[4], float bias_v, int col, int lim, float* dst, int rowoff = 0) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = acc[i] + bias_v;
    if (el.act == DRS_ACT_RELU) {                // (uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (el.act == DRS_ACT_SIGMOID) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], DRS_ACT_SIGMOID);
    }
    if (el.out_off >= 0 && col < lim) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i * el.out_ld] = col < el.N ? v[i] : 0.f;
    }
    if (el.g_out && gw && col < el.N) {          // the last layer of a chain
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + rowoff + g * 4 + i;
        if (row < a.M) {
          float* dstg = el.g_out + row * el.g_ld + col;
          if (el.g_sc1) __hip_atomic_store(dstg, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dstg = v[i];
        }
      }
    }
  };

#define S4_ACC_READ(DST, A0, A1, A2, A3)                                                          \
  asm volatile("v_accvgpr_read_b32 %0, " A0 "\n\tv_accvgpr_read_b32 %1, " A1 "\n\t"               \
               "v_accvgpr_read_b32 %2, " A2 "\n\tv_accvgpr_read_b32 %3, " A3                      \
               : "=v"(DST[0]), "=v"(DST[1]), "=v"(DST[2]), "=v"(DST[3]))
  auto desc = [&](int i) {
    const uint4 d = *reinterpret_cast<const uint4*>(s_tab + 4 * i);
    STile t;
    t.wp_off = __builtin_amdgcn_readfirstlane(d.x); t.a_off = __builtin_amdgcn_readfirstlane(d.y);
    t.in_ld = __builtin_amdgcn_readfirstlane(d.z); t.info = __builtin_amdgcn_readfirstlane(d.w);
    return t;
  };
  int ti = 0, par = 0;          // par: the ring slot this wave's chunk 0 of the segment was requested into
  STile cur = desc(0);
  int tadd = tadd_of(0);
  Seg sg = seg_of(cur.wp_off, cur.in_ld, cur.info, tadd);
  bool alive = true;            // (column-split form: false once another workgroup has taken my slab over)
  while (ti < n_table) {
    const int nti = ti + sg.nch;
    const int last_info = __builtin_amdgcn_readfirstlane(s_tab[4 * (nti - 1) + 3]);
    // the next segment's descriptor now (its chunk 0 is requested from inside this segment's statement),
    // and everything the epilogue needs from LDS -- layer record, biases -- BEFORE the statement: the
    // reads complete under its MFMAs instead of after them
    const STile nx = desc(min(nti, n_table - 1));
    const int tadd_n = tadd_of(min(nti, n_table - 1));
    const Seg sn = seg_of(nx.wp_off, nx.in_ld, nx.info, tadd_n);
    if constexpr (kCanDefer) {
      if (__builtin_expect(defer1 && ti == a.wait_tile, 0)) {
        // the gather's flag (a stream-ordered write queued behind it: its rows are in memory), then the rows
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(done.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != done.wait_val &&
                 ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(4);
          // (bounded: a flag that never comes must not hang the GPU -- bit 1 of the device error word makes the
          // host fail the set instead of handing out sums over rows that were not there yet)
          if (spins >= (1 << 22) && done.dev_err) atomicOr(const_cast<uint32_t*>(done.dev_err), 2u);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int jb = 0; jb < nj1; jb += PB) { issue(rp1, cols1, jb, nj1, pv1); store(ld1, lc1, cpad1, jb, nj1, pv1); }
        __syncthreads();
      }
    }
    if (__builtin_expect((cur.info & S3_INTERACT) != 0, 0)) interact();
    TL(10);
    const Epi el = lds_epi((last_info >> 24) & 0xff);
    const int tpw = sg.tpw;
    const int col0 = ((cur.info & 0xff) + tadd + tpw * wave) * 16 + r;
    const int lim = el.out_off >= 0 ? max(el.out_pad, el.N) : el.N;
    float* const dst = smem + el.out_off + (g * 4) * el.out_ld + lpos(col0 + el.out_col0);
    if (sg.nex > 0) {
      const float b0 = smem[el.b_off + min(col0, el.N - 1)], b1 = smem[el.b_off + min(col0 + 16, el.N - 1)];
      const float b2 = smem[el.b_off + min(col0 + 32, el.N - 1)], b3 = smem[el.b_off + min(col0 + 48, el.N - 1)];
      uint32_t aaddr = (uint32_t)(((cur.a_off & 0xffff) + r * (cur.a_off >> 16) + g * 4) * 4);
      int rem = sg.nch;
      uint32_t r0 = sg.off[0] + 32768u, r1 = sg.off[1] + 32768u, r2 = sg.off[2] + 32768u, r3 = sg.off[3] + 32768u;
      float c0[4], c1[4], c2[4], c3[4];
      if constexpr (R == 2) {
        uint32_t aaddr1 = aaddr + (uint32_t)(16 * (cur.a_off >> 16) * 4);       // rows 16 .. 31 of the slab
        float d0[4], d1[4], d2[4], d3[4];
        if (tpw == 4) {
          asm volatile(SEG2_ASM_T4 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        } else if (tpw == 2) {
          asm volatile(SEG2_ASM_T2 : "+v"(r0), "+v"(r1), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        } else {
          asm volatile(SEG2_ASM_T1 : "+v"(r0), "+v"(aaddr), "+v"(aaddr1), "+s"(rem)
                       : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                       : "memory", "scc", SEG2_AGPR_CLOBBER);
        }
        TL(12);
        // accumulators: tile j of half h at a[4 (j + tpw h) ...]
        float* const dsth = dst + 16 * el.out_ld;
        if (tpw == 4) {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
          S4_ACC_READ(c2, "a8", "a9", "a10", "a11"); S4_ACC_READ(c3, "a12", "a13", "a14", "a15");
          S4_ACC_READ(d0, "a16", "a17", "a18", "a19"); S4_ACC_READ(d1, "a20", "a21", "a22", "a23");
          S4_ACC_READ(d2, "a24", "a25", "a26", "a27"); S4_ACC_READ(d3, "a28", "a29", "a30", "a31");
          epilogue(el, c2, b2, col0 + 32, lim, dst + 32); epilogue(el, c3, b3, col0 + 48, lim, dst + 48);
          epilogue(el, d2, b2, col0 + 32, lim, dsth + 32, 16); epilogue(el, d3, b3, col0 + 48, lim, dsth + 48, 16);
          epilogue(el, c1, b1, col0 + 16, lim, dst + 16); epilogue(el, d0, b0, col0, lim, dsth, 16);
          epilogue(el, d1, b1, col0 + 16, lim, dsth + 16, 16);
        } else if (tpw == 2) {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
          S4_ACC_READ(d0, "a8", "a9", "a10", "a11"); S4_ACC_READ(d1, "a12", "a13", "a14", "a15");
          epilogue(el, c1, b1, col0 + 16, lim, dst + 16); epilogue(el, d0, b0, col0, lim, dsth, 16);
          epilogue(el, d1, b1, col0 + 16, lim, dsth + 16, 16);
        } else {
          S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(d0, "a4", "a5", "a6", "a7");
          epilogue(el, d0, b0, col0, lim, dsth, 16);
        }
        epilogue(el, c0, b0, col0, lim, dst);
      } else {
      if (tpw == 4) {
        asm volatile(SEG_ASM_T4 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      } else if (tpw == 2) {
        asm volatile(SEG_ASM_T2 : "+v"(r0), "+v"(r1), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      } else {
        asm volatile(SEG_ASM_T1 : "+v"(r0), "+v"(aaddr), "+s"(rem)
                     : "v"(sn.off[0]), "v"(sn.off[1]), "v"(sn.off[2]), "v"(sn.off[3]), "s"(wbase), "s"(par)
                     : "memory", "scc", SEG_AGPR_CLOBBER);
      }
      TL(12);
      S4_ACC_READ(c0, "a0", "a1", "a2", "a3"); S4_ACC_READ(c1, "a4", "a5", "a6", "a7");
      S4_ACC_READ(c2, "a8", "a9", "a10", "a11"); S4_ACC_READ(c3, "a12", "a13", "a14", "a15");
      if (tpw == 4) { epilogue(el, c2, b2, col0 + 32, lim, dst + 32); epilogue(el, c3, b3, col0 + 48, lim, dst + 48); }
      epilogue(el, c0, b0, col0, lim, dst);
      if (tpw >= 2) epilogue(el, c1, b1, col0 + 16, lim, dst + 16);
      }
      par = (par + sg.nch) & 1;
    } else {
      // this wave sits the segment out -- but it still has to request the next one's chunk 0, and
      // the columns of the pad that no twin tile covers want zeros in the slab
      if (el.out_off >= 0) {
        for (int h = 0; h < R; ++h)
          for (int t = 0; t < tpw; ++t)
            if (col0 + 16 * t < lim)
              for (int i = 0; i < 4; ++i) dst[16 * t + (16 * h + i) * el.out_ld] = 0.f;
      }
      prefetch(sn, par);
    }
    if (last_info & S3_BARRIER) { TL(13); __syncthreads(); TL(14); }
    if constexpr (SPL) {
      if (nti == sp.t1) {      // (uniform) the split layer is done: my piece of its output slab is in LDS
        // ---- the seam (cdna guide G16 R1, "splitk-seam"): piece -> exchange buffer by 16-byte write-through stores,
        // every wave drains, one lane takes the slab's ticket; whoever draws the last one has every piece visible.
        // The buffer keeps the slab's own column order (lpos permutes inside 16-column blocks; a piece is whole blocks).
        const int cw4 = sp.tps * 4;                       // float4 per row of a piece
        const int n4 = sp.n >> 2;                         // ... of the whole row
        float* const xrow = sp.xbuf + (size_t)m0 * sp.n;
        const float* const sl = smem + sp.off;
        for (int i = tid; i < 16 * R * cw4; i += kThreads) {
          const int row = i / cw4, c4 = ns_y * cw4 + (i - row * cw4);
          const f32x4 v = *reinterpret_cast<const f32x4*>(sl + row * sp.ld + 4 * c4);
          float* dstx = xrow + (size_t)row * sp.n + 4 * c4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstx), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory", SEG_AGPR_CLOBBER);
        __syncthreads();
        TL(15);
        // (the chains' first input slab sits at LDS offset 0 and is dead since layer 0: its first word carries the verdict)
        unsigned* const s_last = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
          const unsigned old = __hip_atomic_fetch_add(sp.xcnt + slab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == (unsigned)a.ns - 1) __hip_atomic_store(sp.xcnt + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_last[0] = old == (unsigned)a.ns - 1;
        }
        __syncthreads();
        alive = s_last[0] != 0;
        TL(16);
        if (!alive) break;
        // last arriver: the other pieces, device-coherent loads (the producers stored write-through), four in flight
        const int o4 = n4 - cw4;                            // float4 per row that are not mine
        for (int i0 = 0; i0 < 16 * R * o4; i0 += 4 * kThreads) {
          f32x4 v[4];
          int at[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = min(i0 + tid + j * kThreads, 16 * R * o4 - 1);
            const int row = i / o4, c = i - row * o4;
            const int c4 = c < ns_y * cw4 ? c : c + cw4;    // skip my own piece
            at[j] = row * sp.ld + 4 * c4;
            const float* src = xrow + (size_t)row * sp.n + 4 * c4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(src));
          }
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i0 + tid + j * kThreads < 16 * R * o4)
              *reinterpret_cast<float4*>(smem + sp.off + at[j]) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
        __syncthreads();
        TL(17);
        gw = true;
      }
    }
    ti = nti;
    cur = nx;
    tadd = tadd_n;
    sg = sn;
  }
#undef S4_ACC_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory", SEG_AGPR_CLOBBER);   // the trailing request
  TL(20);
  if constexpr (SPL) {
    if (alive) signal_done(done, gridDim.x / (unsigned)a.ns, smem, (int)slab);
  } else
  signal_done(done, gridDim.x, smem);
#ifdef DRS_TIMELINE
  TL(21);
  if (tl_on && alive && threadIdx.x == 0) {
    const unsigned n = (unsigned)g_tl_lds[0];
    unsigned base = g_tl_n;
    for (unsigned i = 0; i < n && base + i < 16384; ++i) g_tl[base + i] = g_tl_lds[i + 1];
    g_tl_n = base + n;
  }
#endif
}
}  // namespace

hipError_t launch_stream4(bool sum1, bool two, int rows, bool split, unsigned grid, size_t lds, hipStream_t s, const SArgs& a,
                          const Done& d, const XSrc& xs, const NSplit& ns) {
  const dim3 g(grid), b(256);
  if (split && rows == 32) hipLaunchKernelGGL((stream4_kernel<false, false, 2, true>), g, b, lds, s, a, d, xs, ns);
  else if (split) hipLaunchKernelGGL((stream4_kernel<false, false, 1, true>), g, b, lds, s, a, d, xs, ns);
  else if (rows == 32) hipLaunchKernelGGL((stream4_kernel<false, false, 2>), g, b, lds, s, a, d, xs, ns);
  else if (sum1) hipLaunchKernelGGL((stream4_kernel<true, false>), g, b, lds, s, a, d, xs, ns);
  else if (two) hipLaunchKernelGGL((stream4_kernel<false, true>), g, b, lds, s, a, d, xs, ns);
  else hipLaunchKernelGGL((stream4_kernel<false, false>), g, b, lds, s, a, d, xs, ns);
  return hipGetLastError();
}

hipError_t stream4_set_attrs() {
  hipError_t e = set_max_lds(stream4_kernel<false, false>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<true, false>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, true>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 2>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 1, true>);
  if (e == hipSuccess) e = set_max_lds(stream4_kernel<false, false, 2, true>);
  return e;
}

#ifdef DRS_TIMELINE
int tl_fetch_stream4(unsigned long long* out, int cap, int reset) { return tl_fetch_here(out, cap, reset); }
#endif

}  // namespace drs

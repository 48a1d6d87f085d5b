// Weight-stationary bf16 MFMA GEMM for the K = 512 linear layers of the CLIP text tower
// (qkv 512->1536, fc1 512->2048 over M = 10^5..10^6 candidate-caption rows):
//
//   C[M,N] = act(A[M,512] . W[N,512]^T + bias)        bf16 operands, fp32 accumulate, bf16 out
//
// Why another GEMM.  Measured on MI355X (tools/probes/ingest_probe.hip): one CU cannot pull more
// than ~20 B/clk from L2 and the chip ~10-12 TB/s, whatever the wave count, depth or pattern.  A
// 256x256-tile GEMM moves 1/128 byte per flop into the CU, so it tops out near 1.4 PFLOP/s before
// any epilogue, and at K = 512 the tile epilogue and the A/W ring coupling cost another factor 2
// (gemm256q: 0.67 ms for the qkv shape, 730 TFLOP/s).  Here the weights never move again:
//   * a work-group owns 256 output columns; each of its 8 waves keeps its 32 columns x 512 k of W
//     in REGISTERS as 32 MFMA operand fragments (128 VGPRs) for the whole kernel.  The register
//     file (512 KiB/CU) is the only on-CU memory that holds a 256 x 512 bf16 weight panel.
//   * only activations stream: 32-row blocks (32 KiB, contiguous in HBM because lda == K) land in a
//     4-deep LDS ring by LDS-DMA, one full 1 KiB row per DMA instruction.  CU ingest is 1/256 byte
//     per flop -- half of the tiled kernel -- and there is no weight traffic in steady state.
//   * every wave multiplies every block against its own columns: one ds_read_b128 + one
//     v_mfma_f32_32x32x16_bf16 per k16 step (LDS read pipe 50 % busy), two accumulator chains.
//   * the memory work of a block (ring refill, epilogue of the previous block, its two stores) is issued one
//     instruction at a time between groups of four MFMAs: a VMEM instruction blocks its wave until the CU's
//     vector-memory path takes it, so a burst keeps the wave out of the matrix pipe for a whole block period.
//   * one barrier per block.  Loads, LDS-DMA and stores retire in order on gfx9's vmcnt, and every
//     VMEM instruction of the loop is issued from inline asm in a fixed number per block, so the
//     wait for block i is an exact count (two blocks of DMA + three epilogues of stores stay in
//     flight).  Out-of-range rows / columns are dropped by the buffer descriptors' bounds check.
//   * LDS row r, 16-byte chunk c lives at chunk c ^ (r & 15): conflict-free for the ds_read_b128 lane
//     groups, and a pure permutation inside each DMA'd row (the source address carries it).
//   * the column groups of one row range sit on neighbouring CUs of one XCD and march in step, so
//     each activation block comes from HBM once and from that XCD's L2 for the other groups.
#include <type_traits>

#include "kernels.h"

namespace czc {

namespace {

// compile-time timing ablations of the folded-LayerNorm form (make ABL=n LIB=...; tools/ab_gemm.py out_mode 7; results are
// garbage): 1 no statistics DMA, 2 plain bias epilogue, 4 bf16 MFMA opcode on the same bytes
#ifndef CZC_LNF_ABL
#define CZC_LNF_ABL 0
#endif
constexpr int LNF_ABL = CZC_LNF_ABL;
constexpr int WR_K = 512;
constexpr int WR_ROWB = WR_K * 2;              // 1 KiB per activation row
constexpr int WR_BLK = 32;                     // rows per block
constexpr int WR_STAGE = WR_BLK * WR_ROWB;     // 32 KiB
constexpr int WR_D = 4;                        // ring depth
constexpr int WR_PATCH = 2048;                 // per-wave epilogue patch (32 rows x 64 bytes)
constexpr int WR_BIAS = 128;                   // per-wave bias slice (32 floats)
constexpr int WR_LDS = WR_D * WR_STAGE + 8 * (WR_PATCH + WR_BIAS);
constexpr int WR_TS = 8;                                         // LNF: row blocks of (mean, rstd) pairs the work-group's statistics area holds (2 x 4)
constexpr int WR_LDS_LNF = WR_LDS + WR_TS * 256;                 // + the statistics area
static_assert(WR_LDS_LNF <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// Scalar fp32 on purpose.  The same epilogue on pairs (v_pk_mul / v_pk_add / v_pk_fma_f32: 130 -> 103 VALU instructions per
// block) made the quick-GELU kernels ~10 % SLOWER inside the engine (fc1: 586 -> 646 us per launch, the activation-free q/k/v
// launches unchanged: profiles/r05_tower_ab.txt) -- the packed fp32 forms do not slide under the MFMA stream the way the
// scalar ones do.
template <int ACT>
__device__ __forceinline__ float wr_act(float v) {
  if (ACT == ACT_QUICK_GELU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v));
  return v;
}

// LNF (round 5: LayerNorm folded into the GEMM, GemmArgs::ln_stat): A is the raw 2-byte residual stream x (fp16 rows), W the
// weights with the LayerNorm gain folded in and every weight row CENTRED, W" = W * diag(gamma) - its row mean, as fp16 whose
// stored rows sum to zero (rowops.hip fold_ln_kernel), so that x . W"^T = (x - mean_m) . (W * diag(gamma))^T and
//   C = act( rstd_m * (x . W"^T) + b' ),   b' = b + W . beta,
// with rstd_m per row from the producer GEMM's partials (rowops.hip ln_finalize_kernel): no LayerNorm kernel, no normalised
// copy y of the rows, and an epilogue of the plain kernel's size (one FMA where it has an add, one 4-byte LDS read per
// block).  The multiply runs on the fp16 MFMA whatever the engine's operand type (x is fp16); the
// OUTPUT keeps the engine's type (F16 ? fp16 : bf16).  Per four blocks one more LDS-DMA (T: 128 rows x 8 bytes of statistics, by
// wave 0 only, into an area shared by the work-group), VMEM order of wave 0 in such a stream D0 D1 T D2 S0 D3 S1.
template <int ACT, bool F16 = false, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm_wreg_kernel(GemmArgs g, int ncg, int nsets, int nblk) {
  using OT = std::conditional_t<F16, f16_t, bf16_t>;  // output element type (the engine's activation type)
  using HT = std::conditional_t<LNF, f16_t, OT>;      // operand element type: MFMA opcode (common.h Half<>)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // XCD-major linear index: work-group b runs on XCD b % 8
  const int q = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int set = q / ncg, cg = q - set * ncg;
  if (set >= nsets) return;
  const int per = (nblk + nsets - 1) / nsets;
  const int b0 = set * per;
  const int nb = min(per, nblk - b0);
  if (nb <= 0) return;

  // ---- this wave's weight panel: 32 columns x 512 k as 32 MFMA fragments ----
  const int col0 = cg * 256 + wave * 32;
  u32x4_t wreg[32];
  {
    const int wrow = min(col0 + l31, g.N - 1);
    const unsigned char* wp = (const unsigned char*)g.W + (long)wrow * g.ldw * 2 + half * 16;
#pragma unroll
    for (int t = 0; t < 32; ++t) wreg[t] = *(const u32x4_t*)(wp + t * 32);
  }
  // consume the panel here: the compiler then waits for these loads once, before the loop, instead of
  // placing a vmcnt wait (which would drain the LDS-DMA ring) in front of the first MFMA of every block
#pragma unroll
  for (int t = 0; t < 32; t += 8)
    asm volatile("" ::"v"(wreg[t]), "v"(wreg[t + 1]), "v"(wreg[t + 2]), "v"(wreg[t + 3]), "v"(wreg[t + 4]), "v"(wreg[t + 5]),
                 "v"(wreg[t + 6]), "v"(wreg[t + 7]));
  unsigned char* patch = smem + WR_D * WR_STAGE + wave * WR_PATCH;
  float* bias_s = (float*)(smem + WR_D * WR_STAGE + 8 * WR_PATCH + wave * WR_BIAS);
  if (lane < 32) bias_s[lane] = (g.bias && col0 + lane < g.N) ? g.bias[col0 + lane] : 0.f;
  unsigned char* stat_s = smem + WR_LDS;                                       // LNF: the work-group's (mean, rstd) pairs
  // LNF, small launches (GemmArgs::ln_part; the launcher guarantees nb <= 8 = the area's 256 rows): the statistics of all of
  // this work-group's rows up front, from the producer's partials -- ln_finalize_kernel's arithmetic to the letter (sums in
  // block order, var = E[x^2] - mean^2 clamped), so the two routes give the same bits -- and no statistics request in the loop.
  // Saves a 5 us launch in front of every folded GEMM where that is a third of the GEMM (one or two images).
  const bool stats_here = LNF && g.ln_part != nullptr;
  if constexpr (LNF) {
    if (stats_here) {
      const int rows = min(nb * WR_BLK, g.M - b0 * WR_BLK);
      const float2* part = (const float2*)g.ln_part + (long)b0 * WR_BLK;
      for (int r = threadIdx.x; r < rows; r += 512) {
        float2 a = part[r];
        for (int b = 1; b < WR_K / 32; ++b) {
          const float2 p2 = part[b * g.ln_part_ld + r];
          a.x = __fadd_rn(a.x, p2.x);
          a.y = __fadd_rn(a.y, p2.y);
        }
        const float n = (float)WR_K;
        const float mean = a.x / n;
        const float var = fmaxf(__fsub_rn(a.y / n, __fmul_rn(mean, mean)), 0.f);
        *(float2*)(stat_s + r * 8) = make_float2(mean, rsqrtf(var + g.ln_eps));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // published by the first block barrier
    }
  }

  // ---- DMA side: this wave lands rows wave*4 + ii (ii 0..3) of every block ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  const int pitch = g.lda * 2;
  int voff[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int r = wave * 4 + ii;
    voff[ii] = r * pitch + ((lane ^ (r & 15)) << 4);
  }
  u32x4_t rsA;
  rsA.w = 0x00020000u;
  // LNF: the (mean, rstd) pairs of FOUR blocks (128 rows x 8 bytes = one 64-lane x 16-byte LDS-DMA) -> half (j / 4) % 2 of a
  // 2 KiB area shared by the work-group, requested by wave 0 only and only with every fourth block: every wave needs the same
  // pairs, and what an extra request costs is not its bytes.  Measured on the fc1 shape against the plain kernel (tools/ab_gemm.py
  // out_mode 7): a seventh VMEM instruction per wave and block +14 %; one per block on wave 0 behind a uniform branch, with the
  // ring wait chosen per wave, +4..10 % (every TAKEN branch in the block's stream is ~1 %: the compiler laid the skip and the
  // two waits out as three taken branches for waves 1..7); the same instruction in every wave with EXEC = 0 for waves 1..7
  // +8..12 % (the hardware still issues it).  Here waves 1..7 fall through an untaken branch, wave 0 leaves the stream once per
  // four blocks, and the ring wait is the plain kernel's for everybody (see `step`).  Wave 0's pieces are published like its
  // operand rows (its counted wait + the block barrier).  Two halves suffice: the pairs of blocks 4G..4G+3 are read in the
  // streams of blocks 4G+1..4G+4, the request that overwrites them (group G+2) is issued in the stream of block 4G+5.
  const unsigned stat0 = lds0 + WR_LDS;
  constexpr bool T4 = !(LNF_ABL & 8);  // ablation 8: one dword request per block (ring of WR_TS slots of 256 bytes)
  const unsigned voffT = (unsigned)(lane * (T4 ? 16 : 4));
  auto stat_dma = [&](int j) {
    if (LNF_ABL & 1) return;
    const bool mine = wave == 0 && (!T4 || (j & 3) == 0) && !stats_here;
    if (__builtin_expect(!mine, 1)) return;
    const long row0 = (long)(b0 + j) * WR_BLK;
    const unsigned long long pt = (unsigned long long)g.ln_stat + (unsigned long long)row0 * 8;
    u32x4_t rsT;
    rsT.x = __builtin_amdgcn_readfirstlane((unsigned)pt); rsT.y = __builtin_amdgcn_readfirstlane((unsigned)(pt >> 32) & 0xffffu);
    rsT.z = __builtin_amdgcn_readfirstlane((unsigned)(min((long)(T4 ? 4 * WR_BLK : WR_BLK), (long)g.M - row0) * 8)); rsT.w = 0x00020000u;
    const unsigned dst = stat0 + (T4 ? ((j >> 2) & 1) * 1024 : (j & (WR_TS - 1)) * 256);
    unsigned keep;
    if constexpr (T4)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "s"(dst), "v"(voffT), "s"(rsT)
                   : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "s"(dst), "v"(voffT), "s"(rsT)
                   : "memory");
  };
  auto issue = [&](int j) {  // block j of this work-group -> ring slot j % WR_D
    const long row0 = (long)(b0 + j) * WR_BLK;
    const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)row0 * pitch;
    rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu;
    rsA.z = (unsigned)(min((long)WR_BLK, (long)g.M - row0) * pitch);
    const unsigned dst = lds0 + (j & (WR_D - 1)) * WR_STAGE + wave * (4 * WR_ROWB);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(rsA)
        : "memory", "scc");
    if constexpr (LNF) stat_dma(j);
  };

  // ---- MFMA side ----
  // fragment t of row l31: logical chunk 2t+half -> physical (2t+half) ^ (l31 & 15); the low four
  // bits repeat with period 8 in t, the rest is an immediate offset
  int va[8];
#pragma unroll
  for (int tl = 0; tl < 8; ++tl) va[tl] = l31 * WR_ROWB + ((((2 * tl + half) ^ (l31 & 15)) & 15) << 4);
  f32x16_t acc0, acc1;

  // ---- the memory work of a block is spread through its MFMA stream ----
  // A VMEM instruction blocks its wave until the CU's vector-memory path accepts it, and that path is
  // the busy resource here (HBM-bound output stream).  Issued in a burst, six VMEM per wave per block
  // keep the wave out of the matrix pipe for about a whole block period; issued one at a time between
  // groups of four MFMAs, the wave only stalls while the queue is actually full and the partner wave's
  // MFMAs fill those holes.  The epilogue of block i-1 (kept in accP) rides in block i's stream too.
  unsigned dma_dst = 0;
  auto dma_piece = [&](int ii) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dma_dst + ii * WR_ROWB), "v"(voff[ii]), "s"(rsA)
                 : "memory");
  };
  auto dma_rebase = [&](int j) {
    const long row0 = (long)(b0 + j) * WR_BLK;
    const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)row0 * pitch;
    rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu;
    rsA.z = (unsigned)(min((long)WR_BLK, (long)g.M - row0) * pitch);
    dma_dst = lds0 + (j & (WR_D - 1)) * WR_STAGE + wave * (4 * WR_ROWB);
  };
  f32x16_t accP;  // finished accumulators of the previous block, waiting for their epilogue

  // output descriptor is rebased per block; columns past N get an offset no descriptor admits
  u32x4_t rsC;
  rsC.w = 0x00020000u;
  const int rrow = lane >> 2, rs = lane & 3;
  const int ccol = col0 + rs * 8;
  const int cpitch = g.ldc * 2;
  const unsigned coff0 = ccol < g.N ? (unsigned)(rrow * cpitch + ccol * 2) : 0x7ffffff0u;
  const unsigned coff1 = ccol < g.N ? (unsigned)((16 + rrow) * cpitch + ccol * 2) : 0x7ffffff0u;

  auto store_rebase = [&](int j) {
    const long row0 = (long)(b0 + j) * WR_BLK;
    const unsigned long long pc = (unsigned long long)g.out_act + (unsigned long long)row0 * cpitch;
    rsC.x = (unsigned)pc; rsC.y = (unsigned)(pc >> 32) & 0xffffu;
    rsC.z = (unsigned)(min((long)WR_BLK, (long)g.M - row0) * cpitch);
  };
  float ln_rstd = 0.f;  // LNF: rstd of this lane's row of the block whose epilogue is running
  auto epi_quad = [&](int qd, int js) {  // accP quad qd -> (LayerNorm correction,) bias, activation, 2-byte -> patch
    const float4 b4 = *(const float4*)(bias_s + 8 * qd + 4 * half);
    float4 v;
    if constexpr (LNF && !(LNF_ABL & 2)) {
      if (qd == 0)  // read once per block: the area is refilled (statistics of a later group of blocks) in this same stream
        ln_rstd = *(const float*)(stat_s + (LNF_ABL & 8 ? (js & (WR_TS - 1)) * 256 : ((js >> 2) & 1) * 1024 + (js & 3) * 256) + l31 * 8 + 4);
      v = make_float4(__fmaf_rn(accP[4 * qd], ln_rstd, b4.x), __fmaf_rn(accP[4 * qd + 1], ln_rstd, b4.y),
                      __fmaf_rn(accP[4 * qd + 2], ln_rstd, b4.z), __fmaf_rn(accP[4 * qd + 3], ln_rstd, b4.w));
    } else {
      v = make_float4(accP[4 * qd] + b4.x, accP[4 * qd + 1] + b4.y, accP[4 * qd + 2] + b4.z, accP[4 * qd + 3] + b4.w);
    }
    v.x = wr_act<ACT>(v.x); v.y = wr_act<ACT>(v.y); v.z = wr_act<ACT>(v.z); v.w = wr_act<ACT>(v.w);
    const int slot = (2 * qd + half) ^ ((l31 >> 1) & 7);
    *(uint2*)(patch + l31 * 64 + slot * 8) = make_uint2(Half<OT>::pack2(v.x, v.y), Half<OT>::pack2(v.z, v.w));
  };
  auto epi_store = [&](int pass) {  // 16 patch rows -> one buffer store
    const int r = pass * 16 + rrow;
    const int x = (r >> 1) & 7;
    const u32x4_t t4 = *(const u32x4_t*)(patch + r * 64 + ((rs ^ (x >> 1)) << 4));
    const u32x4_t sw = {t4.z, t4.w, t4.x, t4.y};
    const u32x4_t d = (x & 1) ? sw : t4;
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(pass ? coff1 : coff0), "s"(rsC) : "memory");
  };
  // block in ring slot `slot`: 8 groups of 4 MFMAs (fragments of group s+1 are read during group s), one
  // auxiliary step pinned after each group.  VMEM order per block: D0 D1 D2 S0 D3 S1.
  auto mfma_block_ilv = [&](int slot, auto steady_c, bool refill_rt, bool prev_rt, int jd, int js) {
    constexpr bool STEADY = decltype(steady_c)::value;  // steady state: refill and epilogue unconditional
    const bool refill = STEADY || refill_rt, prev = STEADY || prev_rt;
    const unsigned char* sA = smem + slot * WR_STAGE;
    u32x4_t fr[2][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) fr[0][k] = *(const u32x4_t*)(sA + va[k & 7] + (k >> 3) * 256);
#pragma unroll
    for (int sgm = 0; sgm < 8; ++sgm) {
      if (sgm < 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = 4 * (sgm + 1) + k;
          fr[(sgm + 1) & 1][k] = *(const u32x4_t*)(sA + va[t & 7] + (t >> 3) * 256);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        const int t = 4 * sgm + k;
        if constexpr (LNF && (LNF_ABL & 4)) {  // timing ablation: the bf16 opcode on the same bytes
          acc0 = Half<bf16_t>::mfma(wreg[t], fr[sgm & 1][k], acc0);
          acc1 = Half<bf16_t>::mfma(wreg[t + 1], fr[sgm & 1][k + 1], acc1);
        } else {
          acc0 = Half<HT>::mfma(wreg[t], fr[sgm & 1][k], acc0);
          acc1 = Half<HT>::mfma(wreg[t + 1], fr[sgm & 1][k + 1], acc1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // descriptor arithmetic (a few dozen SALU) rides in the slots too: nothing but the wait and the barrier
      // stands between two blocks' MFMA streams
      if (sgm == 0 && refill) { dma_rebase(jd); dma_piece(0); }
      if (sgm == 1 && prev) epi_quad(0, js);
      if (sgm == 2 && refill) dma_piece(1);
      if (sgm == 2 && prev) epi_quad(1, js);
      if (sgm == 3 && prev) epi_quad(2, js);
      if constexpr (LNF) { if (sgm == 3 && refill) stat_dma(jd); }
      if (sgm == 4 && refill) dma_piece(2);
      if (sgm == 4 && prev) { epi_quad(3, js); store_rebase(js); }
      if (sgm == 5 && prev) epi_store(0);
      if (sgm == 6 && refill) dma_piece(3);
      if (sgm == 7 && prev) epi_store(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int pro = nb < WR_D - 1 ? nb : WR_D - 1;
  for (int j = 0; j < pro; ++j) issue(j);
  {
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto step = [&](int i, auto steady_c) {
      // block i landed?  VMEM issued after its last DMA piece: S1 of that block period, then two full periods
      // of 4 DMA + 2 stores (stores start with the second block) -- all in order on vmcnt.
      constexpr bool STEADY = decltype(steady_c)::value;
      // LNF: wave 0 has a seventh instruction (T) in every fourth stream.  The same count stays valid: the thirteen newest
      // are at least the two previous streams' twelve and one more, so everything up to the last DMA piece of the stream
      // that requested block i (and its T) has landed -- at most one entry stricter than needed, on a request two block
      // periods old, and no per-wave choice of the wait (a branch in front of the barrier) as in the first form.
      if (STEADY) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      else if (i + 2 >= nb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (i < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      mfma_block_ilv(i & (WR_D - 1), steady_c, i + WR_D - 1 < nb, i > 0, i + WR_D - 1, i - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) accP[r] = acc0[r] + acc1[r];
    };
    // head (short store history: conservative waits), branch-free steady state, tail (no more refills)
    int i = 0;
    for (; i < nb && i < 4; ++i) step(i, F_());
    for (; i + WR_D - 1 < nb; ++i) step(i, T_());
    for (; i < nb; ++i) step(i, F_());
    store_rebase(nb - 1);
    epi_quad(0, nb - 1); epi_quad(1, nb - 1); epi_quad(2, nb - 1); epi_quad(3, nb - 1);
    epi_store(0); epi_store(1);
  }
}


// ================================================================================================
// The same weight-stationary structure with a RESIDUAL epilogue on a 2-byte residual stream (round 5; GemmArgs::x16):
//
//   x[M,N] <- fp16( x + A[M,512] . W[N,512]^T + bias )      N % 256 == 0 (the CLIP-text out-projection: N = 512)
//
// i.e. the out-projection without a 128 x 512 full-row tile that re-streams all of W per 128 rows (gemm_rowln: 10 KB per row
// through the CU's vector-memory path; here 2 x 1 KB of activations for the two column groups + 1 KB residual in + 1 KB x
// out).  Differences from gemm_wreg_kernel:
//   * every wave owns two 2 KiB LDS tiles (32 rows x 32 columns of fp16, slot = block parity: ring 128 KiB + 8 x 4 KiB = the
//     full 160 KiB; the bias lives in registers).  The residual rows of block i arrive in tile i & 1 by LDS-DMA (two
//     instructions: 16 rows x 64 B each, chunk-XOR swizzled in the source address), requested in block i's own MFMA stream
//     and consumed in block i+1's: a block period of latency cover and NO register with a load in flight across a branch or
//     a loop edge (a first version kept the rows in registers: hipcc copied such a register at a join before the load had
//     landed -- correct at 40 k rows, garbage at 64 k);
//   * the epilogue adds in the accumulator layout: lane (row, 4 columns) reads its four residual values from the tile
//     (ds_read_b64), forms (acc + bias) + residual in fp32, rounds once to fp16 and writes the result back IN PLACE; the
//     store pass then reads the tile in 16-byte pieces, 16 rows x 64 B per buffer store as in gemm_wreg_kernel;
//   * VMEM order per block: D0 R0 D1 R1 D2 D3 P S0 S1 (DMA pieces of block i+3, residual tiles of block i, LayerNorm partials
//     and stores of block i-1), all from inline asm, so the waits are exact counts: block i's operands have landed at
//     vmcnt(21), the residual tile of block i-1 at vmcnt(8) in front of the first epilogue quad;
//   * GemmArgs::row_part: (sum, sum of squares) of the stored fp16 values per row and 32-column block, for the LayerNorm that
//     the consumer GEMM folds into its epilogue;
//   * one accumulator chain (gfx950 forwards a dependent same-type MFMA's accumulator): 16 registers less.
// ================================================================================================
constexpr int WRR_TILE = 2048;                    // 32 rows x 64 B
constexpr int WRR_LDS = WR_D * WR_STAGE + 8 * 2 * WRR_TILE;
static_assert(WRR_LDS <= 160 * 1024, "LDS budget");

template <bool F16>
__global__ __launch_bounds__(512, 2) void gemm_wreg_resid_kernel(GemmArgs g, int ncg, int nsets, int nblk) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int q = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // integer division runs on the VALU: pin its (uniform) results to SGPRs
  const int set = __builtin_amdgcn_readfirstlane(q / ncg), cg = q - set * ncg;
  if (set >= nsets) return;
  const int per = __builtin_amdgcn_readfirstlane((nblk + nsets - 1) / nsets);
  const int b0 = set * per;
  const int nb = min(per, nblk - b0);
  if (nb <= 0) return;

  const int col0 = cg * 256 + wave * 32;
  u32x4_t wreg[32];
  {
    const unsigned char* wp = (const unsigned char*)g.W + (long)(col0 + l31) * g.ldw * 2 + half * 16;
#pragma unroll
    for (int t = 0; t < 32; ++t) wreg[t] = *(const u32x4_t*)(wp + t * 32);
  }
  // accumulator layout: lane (row l31, half) holds columns col0 + 8 qd + 4 half .. + 3 of quad qd
  f32x4_t bias4[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    bias4[qd] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bias4[qd] = *(const f32x4_t*)(g.bias + col0 + 8 * qd + 4 * half);
  }
#pragma unroll
  for (int t = 0; t < 32; t += 8)
    asm volatile("" ::"v"(wreg[t]), "v"(wreg[t + 1]), "v"(wreg[t + 2]), "v"(wreg[t + 3]), "v"(wreg[t + 4]), "v"(wreg[t + 5]),
                 "v"(wreg[t + 6]), "v"(wreg[t + 7]));
  asm volatile("" ::"v"(bias4[0]), "v"(bias4[1]), "v"(bias4[2]), "v"(bias4[3]));  // waited for here, once, like the weight panel
  unsigned char* tiles = smem + WR_D * WR_STAGE + wave * (2 * WRR_TILE);

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  const int pitch = g.lda * 2;
  int voff[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int r = wave * 4 + ii;
    voff[ii] = r * pitch + ((lane ^ (r & 15)) << 4);
  }
  // descriptors are rebuilt from scalars where they are used (a few SALU each): a descriptor carried around the block loop
  // ended up in VGPRs here, which an asm "s" operand cannot take
  auto desc_rows = [&](const void* base, int j, int row_pitch) __attribute__((always_inline)) {
    const long row0 = (long)(b0 + j) * WR_BLK;
    const unsigned long long pa = (unsigned long long)base + (unsigned long long)row0 * row_pitch;
    u32x4_t d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    d.z = __builtin_amdgcn_readfirstlane((unsigned)(min((long)WR_BLK, (long)g.M - row0) * row_pitch));
    d.w = 0x00020000u;
    return d;
  };
  auto issue = [&](int j) __attribute__((always_inline)) {
    const u32x4_t d = desc_rows(g.A, j, pitch);
    const unsigned dst = lds0 + (j & (WR_D - 1)) * WR_STAGE + wave * (4 * WR_ROWB);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(d)
        : "memory", "scc");
  };
  auto dma_piece = [&](int ii, int j) __attribute__((always_inline)) {
    const u32x4_t d = desc_rows(g.A, j, pitch);
    const unsigned dst = lds0 + (j & (WR_D - 1)) * WR_STAGE + wave * (4 * WR_ROWB) + ii * WR_ROWB;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "v"(voff[ii]), "s"(d)
                 : "memory");
  };
  int va[8];
#pragma unroll
  for (int tl = 0; tl < 8; ++tl) va[tl] = l31 * WR_ROWB + ((((2 * tl + half) ^ (l31 & 15)) & 15) << 4);
  f32x16_t acc0, accP;

  // residual tile of block j -> slot j & 1: DMA lane = 4 * row + physical chunk; logical chunk = physical ^ ((row >> 2) & 3)
  const int cpitch = g.ldc * 2, rpitch = g.ldr * 2;
  const int drow = lane >> 2, dsw = (lane >> 4) & 3;
  const unsigned roff0 = (unsigned)(drow * rpitch + (col0 + (((lane & 3) ^ dsw) << 3)) * 2);
  const unsigned roff1 = roff0 + (unsigned)(16 * rpitch);
  const unsigned tile0 = lds0 + WR_D * WR_STAGE + wave * (2 * WRR_TILE);
  auto resid_dma = [&](int pass, int j) __attribute__((always_inline)) {
    const u32x4_t rd = desc_rows(g.resid, j, rpitch);
    const unsigned dst = tile0 + (j & 1) * WRR_TILE + pass * 1024;
    const unsigned ro = pass ? roff1 : roff0;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "v"(ro), "s"(rd)
                 : "memory");
  };
  // store layout: lane = 4 * row + chunk; the lane owns columns col0 + 8 * chunk .. + 7 of rows pass * 16 + (lane >> 2)
  const int rs = lane & 3;
  const unsigned coff0 = (unsigned)(drow * cpitch + (col0 + rs * 8) * 2), coff1 = coff0 + (unsigned)(16 * cpitch);
  const int equad = l31 * 64 + half * 8;             // this lane's 8 bytes inside a 16-byte chunk of tile row l31
  const int esw = (l31 >> 2) & 3;
  float st_s = 0.f, st_q = 0.f;
  auto epi_quad = [&](int qd, int j) __attribute__((always_inline)) {  // (accP quad + bias) + residual -> fp16, in place in the tile
    unsigned char* pq = tiles + (j & 1) * WRR_TILE + equad + ((qd ^ esw) << 4);
    const uint2 rv = *(const uint2*)pq;
    const float r0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rv.x & 0xffffu));
    const float r1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rv.x >> 16));
    const float r2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rv.y & 0xffffu));
    const float r3 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rv.y >> 16));
    const float v0 = (accP[4 * qd] + bias4[qd][0]) + r0, v1 = (accP[4 * qd + 1] + bias4[qd][1]) + r1;
    const float v2 = (accP[4 * qd + 2] + bias4[qd][2]) + r2, v3 = (accP[4 * qd + 3] + bias4[qd][3]) + r3;
    const uint2 pk = make_uint2(pack2_f16(v0, v1), pack2_f16(v2, v3));
    *(uint2*)pq = pk;
    // LayerNorm partials of the stored values: this lane's four quads in order (columns 8 qd + 4 half .. + 3 of the 32-column block)
    const float w0 = f16lo(pk.x), w1 = f16hi(pk.x), w2 = f16lo(pk.y), w3 = f16hi(pk.y);
    const float ps = ln_sum4(w0, w1, w2, w3), pq2 = ln_sq4(w0, w1, w2, w3);
    st_s = qd == 0 ? ps : __fadd_rn(st_s, ps);
    st_q = qd == 0 ? pq2 : __fadd_rn(st_q, pq2);
  };
  // partials of block j's 32 rows over this wave's 32 columns: half 0 + half 1, written by lanes 0..31 (8 bytes per row,
  // 256 contiguous bytes per wave: row_part is [column block][row]); without a target the descriptor is empty
  // (no target: a valid base with every lane's offset past its few records -- the drop that serves the column edges; a null
  // base with zero records faulted)
  const long pblk = (long)(cg * 8 + wave) * g.part_ld * 8;
  const unsigned poff = (half || !g.row_part) ? 0x7ffffff0u : (unsigned)(l31 * 8);
  auto part_store = [&](int j) __attribute__((always_inline)) {
    const float os = __shfl_xor(st_s, 32, 64), oq = __shfl_xor(st_q, 32, 64);
    const float S = __fadd_rn(half ? os : st_s, half ? st_s : os), Q = __fadd_rn(half ? oq : st_q, half ? st_q : oq);
    const u32x4_t pd = desc_rows(g.row_part ? (const unsigned char*)g.row_part + pblk : (const unsigned char*)g.W, g.row_part ? j : -b0, 8);
    const uint2 sq = make_uint2(__float_as_uint(S), __float_as_uint(Q));
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(sq), "v"(poff), "s"(pd) : "memory");
  };
  auto epi_store = [&](int pass, int j) __attribute__((always_inline)) {  // 16 tile rows -> one buffer store
    const int r = pass * 16 + drow;
    const u32x4_t o = *(const u32x4_t*)(tiles + (j & 1) * WRR_TILE + r * 64 + ((rs ^ ((r >> 2) & 3)) << 4));
    const unsigned co = pass ? coff1 : coff0;
    const u32x4_t rc = desc_rows(g.out_f32, j, cpitch);
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(o), "v"(co), "s"(rc) : "memory");
  };
  auto wait_resid = [&](int n) __attribute__((always_inline)) {  // at most n younger VMEM instructions stay in flight
    switch (n) {
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    }
  };

  // block i in ring slot i & 3.  VMEM order: D0 R0 D1 R1 D2 D3 P S0 S1; the epilogue of block i-1 (accP) rides in the stream
  auto mfma_block = [&](auto steady_c, bool refill_rt, bool prev_rt, int rwait, int i) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool refill = STEADY || refill_rt, prev = STEADY || prev_rt;
    const unsigned char* sA = smem + (i & (WR_D - 1)) * WR_STAGE;
    const int jd = i + WR_D - 1, js = i - 1;
    u32x4_t fr[2][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) fr[0][k] = *(const u32x4_t*)(sA + va[k & 7] + (k >> 3) * 256);
#pragma unroll
    for (int sgm = 0; sgm < 8; ++sgm) {
      if (sgm < 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = 4 * (sgm + 1) + k;
          fr[(sgm + 1) & 1][k] = *(const u32x4_t*)(sA + va[t & 7] + (t >> 3) * 256);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc0 = Half<HT>::mfma(wreg[4 * sgm + k], fr[sgm & 1][k], acc0);
      __builtin_amdgcn_sched_barrier(0);
      if (sgm == 0 && refill) dma_piece(0, jd);
      if (sgm == 1) resid_dma(0, i);
      if (sgm == 2 && refill) dma_piece(1, jd);
      if (sgm == 2 && prev) { if (STEADY) wait_resid(8); else wait_resid(rwait); epi_quad(0, js); }
      if (sgm == 3) resid_dma(1, i);
      if (sgm == 3 && prev) epi_quad(1, js);
      if (sgm == 4 && refill) dma_piece(2, jd);
      if (sgm == 4 && prev) epi_quad(2, js);
      if (sgm == 5 && refill) dma_piece(3, jd);
      if (sgm == 5 && prev) epi_quad(3, js);
      if (sgm == 6 && prev) { part_store(js); epi_store(0, js); }
      if (sgm == 7 && prev) epi_store(1, js);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int pro = nb < WR_D - 1 ? nb : WR_D - 1;
  for (int j = 0; j < pro; ++j) issue(j);
  {
    using T_ = std::true_type;
    using F_ = std::false_type;
    // steady block: its operands were requested in block i-3's stream (last piece at position 6 of D0 R0 D1 R1 D2 D3 P S0 S1):
    // P S0 S1 + two full periods = 21 younger.  The residual tile of block i-1 (R1 at position 4 of its stream) is waited for
    // in front of quad 0, in slot 2 behind D0 R0 D1: D2 D3 P S0 S1 + D0 R0 D1 = 8 younger
    auto steady = [&](int i) __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      mfma_block(T_(), true, true, 8, i);
#pragma unroll
      for (int r = 0; r < 16; ++r) accP[r] = acc0[r];
    };
    // head / tail block: conservative counts (any count <= the number of younger instructions is safe)
    auto edge = [&](int i) __attribute__((always_inline)) {
      const bool refill = i + WR_D - 1 < nb, refill_prev = i + WR_D - 2 < nb;
      if (i + 2 >= nb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (i < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // younger than R1(i-1) at the wait: [D2 D3 if block i-1 refilled] [P S0 S1 if i-1 > 0] of its stream, then of this one
      // [D0] R0 [D1] (the D's if this block refills)
      const int rw = (refill_prev ? 2 : 0) + (i > 1 ? 3 : 0) + 1 + (refill ? 2 : 0);  // 1, 3, 4, 5, 6 or 8
      mfma_block(F_(), refill, i > 0, rw, i);
#pragma unroll
      for (int r = 0; r < 16; ++r) accP[r] = acc0[r];
    };
    int i = 0;
    for (; i < nb && i < 4; ++i) edge(i);
    for (; i + WR_D - 1 < nb; ++i) steady(i);
    for (; i < nb; ++i) edge(i);
    // last block's epilogue: its residual tile was requested in its own stream
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epi_quad(0, nb - 1); epi_quad(1, nb - 1); epi_quad(2, nb - 1); epi_quad(3, nb - 1);
    part_store(nb - 1);
    epi_store(0, nb - 1); epi_store(1, nb - 1);
  }
}

}  // namespace

int g_use_wreg = 1;        // 0: every K = 512 layer goes to the tiled kernels (tests pin kernel families with it)
// Every eligible layer, whatever its row count: the kernel's two accumulator chains and raw-exp quick-GELU give other
// last bits than the tiled kernels, so choosing by M would make an image's caption depend on its batch; measured at one
// image (M ~ 1-3 k rows) it is also the faster kernel (4.21 vs 4.10 captions/s), at 8 images equal
int g_wreg_min_m = 1;
int g_wreg_stats_in_kernel = 1;

// mirrors launch_gemm_wreg's split of the row blocks over the work-groups: at most WR_TS blocks (the statistics area) per work-group
bool gemm_wreg_stats_in_kernel(int M, int N) {
  if (!g_wreg_stats_in_kernel || M <= 0 || N <= 0) return false;
  static int cu_of[64];  // per device, filled on first use (racing host threads write the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!cu_of[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    cu_of[dev] = n;
  }
  const int n_cu = cu_of[dev] & ~7;
  const int ncg = cdiv(N, 256), nblk = cdiv(M, WR_BLK);
  int nsets = n_cu / ncg;
  if (nsets < 1) return false;
  if (nsets > nblk) nsets = nblk;
  return cdiv(nblk, nsets) <= WR_TS;
}

bool gemm_wreg_eligible(const GemmArgs& g) {
  if (g.ln_stat && !g.bias) return false;
  return (g_use_wreg || g.ln_stat) && g.K == WR_K && g.M >= g_wreg_min_m && g.N % 8 == 0 && g.ldc % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         g.out_act && !g.out_f32 && !g.resid && (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) &&
         (long)g.ldc * 2 * WR_BLK < (1L << 30) && (long)g.lda * 2 * WR_BLK < (1L << 30);
}

// below: the tiled kernel's x16 branch serves the layer (same k order, same epilogue arithmetic: bit-identical rows and partials,
// tests/test_kernels_gpu.py) -- a work-group of this kernel loads its 256 KB weight panel for as little as one 32-row block
int g_wreg_resid_min_m = 6144;
bool gemm_wreg_resid_eligible(const GemmArgs& g) {
  return g_use_wreg && g.M >= g_wreg_resid_min_m && g.x16 && g.K == WR_K && g.N % 256 == 0 && g.resid && g.out_f32 && !g.out_act && g.act == ACT_NONE &&
         g.ldc % 8 == 0 && g.ldr % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 && (long)g.ldc * 2 * WR_BLK < (1L << 30) &&
         (long)g.ldr * 2 * WR_BLK < (1L << 30) && (long)g.lda * 2 * WR_BLK < (1L << 30);
}

int launch_gemm_wreg_resid(const GemmArgs& g, hipStream_t st) {
  static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit& li) -> int {
    li.n_cu &= ~7;
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_wreg_resid_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WRR_LDS));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_wreg_resid_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRR_LDS));
    return 0;
  });
  if (init.rc) return launch_init_failed("gemm_wreg_resid");
  const int n_cu = init.n_cu;
  const int ncg = g.N / 256;
  const int nblk = cdiv(g.M, WR_BLK);
  int nsets = n_cu / ncg;
  if (nsets < 1) { snprintf(g_err, sizeof(g_err), "gemm_wreg_resid: N=%d needs more column groups than CUs", g.N); return 1; }
  if (nsets > nblk) nsets = nblk;
  dim3 grid(n_cu), block(512);
  if (g.f16) hipLaunchKernelGGL((gemm_wreg_resid_kernel<true>), grid, block, WRR_LDS, st, g, ncg, nsets, nblk);
  else hipLaunchKernelGGL((gemm_wreg_resid_kernel<false>), grid, block, WRR_LDS, st, g, ncg, nsets, nblk);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_gemm_wreg(const GemmArgs& g, hipStream_t st) {
  // one-time set-up behind a function-local static: engines on two host threads launch concurrently (EngineGroup)
  static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit& li) -> int {
    li.n_cu &= ~7;
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS))
    CZC_ATTR((gemm_wreg_kernel<ACT_NONE, false>));
    CZC_ATTR((gemm_wreg_kernel<ACT_QUICK_GELU, false>));
    CZC_ATTR((gemm_wreg_kernel<ACT_NONE, true>));
    CZC_ATTR((gemm_wreg_kernel<ACT_QUICK_GELU, true>));
#undef CZC_ATTR
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS_LNF))
    CZC_ATTR((gemm_wreg_kernel<ACT_NONE, false, true>));
    CZC_ATTR((gemm_wreg_kernel<ACT_QUICK_GELU, false, true>));
    CZC_ATTR((gemm_wreg_kernel<ACT_NONE, true, true>));
    CZC_ATTR((gemm_wreg_kernel<ACT_QUICK_GELU, true, true>));
#undef CZC_ATTR
    return 0;
  });
  if (init.rc) return launch_init_failed("gemm_wreg");
  const int n_cu = init.n_cu;
  const int ncg = cdiv(g.N, 256);
  const int nblk = cdiv(g.M, WR_BLK);
  int nsets = n_cu / ncg;
  if (nsets < 1) {
    snprintf(g_err, sizeof(g_err), "gemm_wreg: N=%d needs more column groups than CUs", g.N);
    return 1;
  }
  if (nsets > nblk) nsets = nblk;
  if (g.ln_part && (!g.ln_stat || cdiv(nblk, nsets) > WR_TS)) {
    snprintf(g_err, sizeof(g_err), "gemm_wreg: statistics in the kernel need at most %d row blocks per work-group (M=%d N=%d): gemm_wreg_stats_in_kernel()", WR_TS, g.M, g.N);
    return 1;
  }
  dim3 grid(n_cu), block(512);
#define CZC_WR_GO(A_, H_) do { if (g.ln_stat) hipLaunchKernelGGL((gemm_wreg_kernel<A_, H_, true>), grid, block, WR_LDS_LNF, st, g, ncg, nsets, nblk); \
                               else hipLaunchKernelGGL((gemm_wreg_kernel<A_, H_>), grid, block, WR_LDS, st, g, ncg, nsets, nblk); } while (0)
  if (g.f16) { if (g.act == ACT_QUICK_GELU) CZC_WR_GO(ACT_QUICK_GELU, true); else CZC_WR_GO(ACT_NONE, true); }
  else { if (g.act == ACT_QUICK_GELU) CZC_WR_GO(ACT_QUICK_GELU, false); else CZC_WR_GO(ACT_NONE, false); }
#undef CZC_WR_GO
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

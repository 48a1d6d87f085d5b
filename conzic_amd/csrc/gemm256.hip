// 256x256-tile bf16 MFMA GEMM with LDS-DMA staging, for the CLIP-text linear layers over the
// B*K candidate captions (M = 10^5..10^6 rows, N in {512,1536,2048}, K in {512,2048}).
//
//   C[M,N] = A[M,K] . W[N,K]^T (+bias)(quick-GELU)(+fp32 residual)      bf16 operands, fp32 accumulate
//
// Structure (MI355X_MICROARCH / cdna_hip_programming guides):
//  * 512 threads = 8 waves as 2(M) x 4(N); each wave owns 128x64 of C as 4x2 v_mfma_f32_32x32x16_bf16
//    tiles (128 accumulator registers).  One work-group per CU (128 KiB LDS, two 64 KiB stages).
//  * K step 64 (128-byte rows).  Global->LDS goes through `buffer_load_dwordx4 ... lds` (no VGPR
//    round trip, no ds_write): every wave instruction lands 1 KiB = 8 tile rows.  The LDS image is
//    lane-linear, so the bank-conflict swizzle (16-byte chunk ^ ((row>>1)&7)) is applied to the
//    per-lane SOURCE address and again on the ds_read_b128 side (guide rule 21).
//  * Buffer descriptors are rebased per tile with num_records = valid rows * pitch: rows past M / N
//    read as zero from the hardware bounds check, so ragged edges need no clamping and a tile never
//    needs more than 32-bit offsets even when the activation tensor exceeds 4 GiB.
//  * One raw s_barrier per K step: wait for tile kt (vmcnt(0)), barrier (which also proves every
//    wave left compute(kt-1), so that stage is free), issue the LDS-DMA of tile kt+1, then 24
//    ds_read_b128 + 32 MFMA per wave while the DMA flies.
//  * Work-group -> tile map is XCD-aware (block b runs on XCD b%8): the N tiles of one M tile are
//    consecutive inside one XCD's share, so an activation tile is fetched into one L2 only.
//  * Epilogue as in gemm.hip: weights are the MFMA A operand, so each lane owns 4 consecutive output
//    columns of one row per register quad -> 8-byte bf16 / 16-byte fp32 stores.
#include <type_traits>

#include "kernels.h"

namespace czc {

namespace {

constexpr int TM = 256, TN = 256, ROWB = 128;
constexpr int A_BYTES = TM * ROWB;            // 32 KiB
constexpr int STAGE = A_BYTES + TN * ROWB;    // 64 KiB
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  // x*sigmoid(1.702x) with the raw v_exp_f32 (2^x) and v_rcp_f32: ~1 ulp each, output is bf16 anyway
  if (ACT == ACT_QUICK_GELU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v));
  return v;
}

template <int ACT>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;

  const int nwg = tiles_m * tiles_n;
  int lin = blockIdx.x;
  {
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    lin = base + (lin >> 3);
  }
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int rows_a = min(TM, g.M - m0), rows_w = min(TN, g.N - n0);
  // buffer descriptors as plain SGPR quads (they are asm operands below): base, base_hi|stride 0,
  // num_records (bytes), flags (DATA_FORMAT=32 as make_buffer_rsrc would set)
  const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
  const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
  u32x4_t rsA, rsW;
  rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(rows_a * lda_b); rsA.w = 0x00020000u;
  rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(rows_w * ldw_b); rsW.w = 0x00020000u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);

  // staging: instruction ii of wave w lands tile rows w*32 + ii*8 + (lane>>3); lane's physical
  // chunk is lane&7, so it fetches logical chunk (lane&7) ^ ((row>>1)&7) of that row.
  // (named scalars, not arrays: indexed pointer/offset arrays ended up in scratch memory)
#define CZC_VO(ii, ld) ((wave * 32 + (ii) * 8 + (lane >> 3)) * (ld) + (((lane & 7) ^ (((ii) * 4 + (lane >> 4)) & 7)) << 4))
  const int voA0 = CZC_VO(0, lda_b), voA1 = CZC_VO(1, lda_b), voA2 = CZC_VO(2, lda_b), voA3 = CZC_VO(3, lda_b);
  const int voW0 = CZC_VO(0, ldw_b), voW1 = CZC_VO(1, ldw_b), voW2 = CZC_VO(2, ldw_b), voW3 = CZC_VO(3, ldw_b);
#undef CZC_VO

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K >> 6;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);

  // LDS-DMA issued from inline asm: hipcc would otherwise drain every DMA (vmcnt(0)) in front of the
  // first ds_read of the compute phase, because it cannot prove the two stages disjoint.  M0 = LDS
  // destination of lane 0 (saved/restored inside the statement; `s_nop 0` covers the M0 write ->
  // LDS-DMA hazard).  Completion is counted by hand: s_waitcnt vmcnt(0) at the top of each K step.
#define CZC_STAGE_TILE(KT)                                                                             \
  {                                                                                                    \
    const unsigned dstA = lds0 + ((KT) & 1) * STAGE + wave * (32 * ROWB);                              \
    const unsigned dstW = dstA + A_BYTES;                                                              \
    const unsigned so = (KT) * ROWB;                                                                   \
    unsigned keep;                                                                                     \
    asm volatile(                                                                                      \
        "s_mov_b32 %0, m0\n\t"                                                                         \
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"                \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"         \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"         \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"         \
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"                \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"         \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"         \
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"        \
        "s_mov_b32 m0, %0"                                                                             \
        : "=&s"(keep)                                                                                  \
        : "s"(dstA), "s"(dstW), "v"(voA0), "v"(voA1), "v"(voA2), "v"(voA3), "v"(voW0), "v"(voW1),      \
          "v"(voW2), "v"(voW3), "s"(rsA), "s"(rsW), "s"(so)                                            \
        : "memory", "scc");                                                                            \
  }

  CZC_STAGE_TILE(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) CZC_STAGE_TILE(kt + 1);
    const unsigned char* sA = smem + (kt & 1) * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = 2 * ks + half;
      uint4 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch));
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *(const uint4*)(sB + swz(brow + 32 * j, ch));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b[j]),
                                                              __builtin_bit_cast(bf16x8_t, a[i]), acc[i][j], 0, 0, 0);
    }
  }
#undef CZC_STAGE_TILE

  // epilogue: lane owns output row (lane&31) of each 32-row block and 4 consecutive columns per quad
  bf16_t* oa = (bf16_t*)g.out_act;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 128 + i * 32 + (lane & 31);
    if (row < g.M) {
      const long ro = (long)row * g.ldc;
      const long rr = (long)row * g.ldr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = n0 + wn * 64 + j * 32 + 8 * q + 4 * half;
          if (col < g.N) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            if (g.bias) {
              const float4 b4 = *(const float4*)(g.bias + col);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            v.x = act_fn<ACT>(v.x); v.y = act_fn<ACT>(v.y); v.z = act_fn<ACT>(v.z); v.w = act_fn<ACT>(v.w);
            if (g.resid) {
              const float4 r4 = *(const float4*)(g.resid + rr + col);
              v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            if (g.out_f32) *(float4*)(g.out_f32 + ro + col) = v;
            if (oa) {
              uint2 o;
              o.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
              o.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
              *(uint2*)(oa + ro + col) = o;
            }
          }
        }
      }
    }
  }
}

// ================================================================================================
// gemm256p: persistent, wave-specialised variant.
//   * 12 waves per work-group: waves 0-7 are MFMA waves (never issue a global load), waves 8-11 are
//     LOADER waves that only issue the LDS-DMA of the next K step (16 x 1 KiB each) -- PMC on the
//     plain kernel showed the 8 DMA issues per K step costing the MFMA waves about as many cycles as
//     their 32 MFMAs, and 53 % of wave cycles parked in s_waitcnt/s_barrier.
//   * one work-group per CU walks tiles wg, wg+grid, ...; the K-step sequence is continuous across
//     tiles (stage = step & 1), so while the MFMA waves run a tile's epilogue the loaders already
//     fetch the next tile's first K step: the pipeline prologue disappears behind the epilogue.
//   * hand-off per K step: loader `s_waitcnt vmcnt(0)` -> one s_barrier shared by all 12 waves ->
//     loaders issue step+1 into the stage the MFMA waves just left, MFMA waves compute step.
// ================================================================================================
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  // XCD-aware: work-group b runs on XCD b%8 and (persistent, grid = #CUs) handles virtual tiles
  // t = b, b+grid, ...; map t so that one XCD sweeps all N tiles of an M tile back to back.
  const int nt = tiles_m * tiles_n;
  const int xcd = t & 7, q = nt >> 3, r = nt & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int lin = base + (t >> 3);
  tm = lin / tiles_n;
  tn = lin - tm * tiles_n;
}

// Accumulators leave through a wave-private 4 KiB LDS patch that turns the MFMA layout (lane = one
// row, 4 columns per quad -> 8-byte pieces scattered over 32 rows) into row-major 16-byte pieces:
// 8 lanes cover one 128-byte line, so every global store / residual load is a full cache line
// instead of 64 partial ones.
// sum over the 8 lanes that share (lane >> 3): two quad-permute DPP steps, then the mirrored half row
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  return v;
}

// STATS (fp32 path only): besides the fp32 result and its bf16 copy, every wave leaves the per-row (sum, sum of
// squares) of its 64 result columns in g.row_stats[row][n/64][2] -- the LayerNorm statistics of the next layer in
// eight fixed-order partials per 512-wide row, so the consumer GEMM can apply the LayerNorm in its epilogue.
template <int ACT, bool OUT_F32, bool STATS = false, typename HT = bf16_t>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, f32x16_t (&acc)[4][2], unsigned char* patch, int m0,
                                              int n0, int wm, int wn, int lane) {
  const int half = lane >> 5;
  bf16_t* oa = (bf16_t*)g.out_act;
    const int l31 = lane & 31;
    const int rrow = lane >> 3, rslot = lane & 7;  // read side: 8 rows x 8 sixteen-byte slots per pass
    if (!OUT_F32) {
      // bf16 output: 32 rows x 64 columns per pass (128-byte rows), 8-byte slots XOR (row & 15)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = j * 32 + 8 * q + 4 * half;  // column inside the wave's 64
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            if (g.bias) {
              const int col = min(n0 + wn * 64 + cl, g.N - 4);
              const float4 b4 = *(const float4*)(g.bias + col);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            v.x = act_fn<ACT>(v.x); v.y = act_fn<ACT>(v.y); v.z = act_fn<ACT>(v.z); v.w = act_fn<ACT>(v.w);
            const int slot = (cl >> 2) ^ (l31 & 15);
            *(uint2*)(patch + l31 * 128 + slot * 8) = make_uint2(Half<HT>::pack2(v.x, v.y), Half<HT>::pack2(v.z, v.w));
          }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = pass * 8 + rrow;
          const int x = r & 15;
          // logical 8-byte slots 2*rslot, 2*rslot+1 live in the physical pair ((2*rslot)^x)>>1, swapped if x&1
          uint4 d = *(const uint4*)(patch + r * 128 + ((((2 * rslot) ^ x) >> 1) << 4));
          if (x & 1) d = make_uint4(d.z, d.w, d.x, d.y);
          const int row = m0 + wm * 128 + i * 32 + r;
          const int col = n0 + wn * 64 + rslot * 8;
          if (row < g.M && col < g.N) *(uint4*)(oa + (long)row * g.ldc + col) = d;
        }
      }
    } else {
      // fp32 output (+bias, +fp32 residual, optional bf16 copy): 32 rows x 32 columns per pass
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + rslot * 4;
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.bias && col < g.N) b4 = *(const float4*)(g.bias + col);
          // residual rows first: four independent 16-byte loads in flight across the LDS round trip
          float4 r4[4];
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int row = m0 + wm * 128 + i * 32 + pass * 8 + rrow;
            r4[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.resid && row < g.M && col < g.N) r4[pass] = *(const float4*)(g.resid + (long)row * g.ldr + col);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = (2 * q + half) ^ (l31 & 7);
            *(float4*)(patch + l31 * 128 + slot * 16) =
                make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          }
          uint2 pk[4];
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + rrow;
            float4 v = *(const float4*)(patch + r * 128 + ((rslot ^ (r & 7)) << 4));
            const int row = m0 + wm * 128 + i * 32 + r;
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            v.x = act_fn<ACT>(v.x); v.y = act_fn<ACT>(v.y); v.z = act_fn<ACT>(v.z); v.w = act_fn<ACT>(v.w);
            v.x += r4[pass].x; v.y += r4[pass].y; v.z += r4[pass].z; v.w += r4[pass].w;
            if (row < g.M && col < g.N) {
              if (g.out_f32) *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v;
              if (STATS) {
                ps[pass] += (v.x + v.y) + (v.z + v.w);
                pq[pass] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
              }
            }
            pk[pass] = make_uint2(Half<HT>::pack2(v.x, v.y), Half<HT>::pack2(v.z, v.w));
          }
          if (oa && !(STATS && (g.ln_groups & 2))) {
            // bf16 copy in 16-byte stores (the epilogue is store-ISSUE bound): lanes rslot, rslot^1 hold adjacent
            // 4-column pieces of the same row; swapping one piece per pass pair leaves the even lane with 8
            // columns of the first pass's row and the odd lane with 8 columns of the second pass's row
            const bool odd = rslot & 1;
#pragma unroll
            for (int pp = 0; pp < 4; pp += 2) {
              const uint2 send = odd ? pk[pp] : pk[pp + 1];
              uint2 recv;
              recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xf, 0xf, true);
              recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xf, 0xf, true);
              const uint4 d = odd ? make_uint4(recv.x, recv.y, pk[pp + 1].x, pk[pp + 1].y)
                                  : make_uint4(pk[pp].x, pk[pp].y, recv.x, recv.y);
              const int row = m0 + wm * 128 + i * 32 + (pp + (odd ? 1 : 0)) * 8 + rrow;
              const int c8 = n0 + wn * 64 + j * 32 + (rslot & 6) * 4;
              if (row < g.M && c8 < g.N) *(uint4*)(oa + (long)row * g.ldc + c8) = d;
            }
          }
        }
        if (STATS && !(g.ln_groups & 4)) {
          // the 8 lanes of a row (rslot 0..7) hold 8 columns each: reduce, then lane rslot == pass keeps pass's
          // row, so that one 8-byte store per lane covers the 32 rows of this block
          float ks = 0.f, kq = 0.f;
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const float s8 = sum8_dpp(ps[pass]), q8 = sum8_dpp(pq[pass]);
            if (rslot == pass) { ks = s8; kq = q8; }
          }
          const int row = m0 + wm * 128 + i * 32 + rslot * 8 + rrow;
          const int grp = (n0 >> 6) + wn;
          if (rslot < 4 && row < g.M && n0 + wn * 64 < g.N && !(g.ln_groups & 1))
            *(float2*)(g.row_stats + ((long)row * (g.N >> 6) + grp) * 2) = make_float2(ks, kq);
        }
      }
    }
}


// Split-fp16 activation output (qkv / fc1 of the split engine): same 32x32 fp32 round trip through the patch as the
// fp32 path, then lanes rslot / rslot^1 swap one 4-column piece per pass pair so that every lane holds the 8
// consecutive columns of one split_t group (16 bytes of fp16 hi parts followed by 16 bytes of lo parts).
template <int ACT>
__device__ __forceinline__ void tile_epilogue_split(const GemmArgs& g, f32x16_t (&acc)[4][2], unsigned char* patch, int m0,
                                                    int n0, int wm, int wn, int lane) {
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int rrow = lane >> 3, rslot = lane & 7;
  unsigned char* oa = (unsigned char*)g.out_act;
  const bool odd = rslot & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + rslot * 4;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias && col < g.N) b4 = *(const float4*)(g.bias + col);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int slot = (2 * q + half) ^ (l31 & 7);
        *(float4*)(patch + l31 * 128 + slot * 16) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
      float4 v[4];
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 8 + rrow;
        float4 t = *(const float4*)(patch + r * 128 + ((rslot ^ (r & 7)) << 4));
        t.x += b4.x; t.y += b4.y; t.z += b4.z; t.w += b4.w;
        t.x = act_fn<ACT>(t.x); t.y = act_fn<ACT>(t.y); t.z = act_fn<ACT>(t.z); t.w = act_fn<ACT>(t.w);
        v[pass] = t;
      }
#pragma unroll
      for (int pp = 0; pp < 4; pp += 2) {
        const float4 send = odd ? v[pp] : v[pp + 1];
        float4 recv;
        recv.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.x), 0xB1, 0xf, 0xf, true));
        recv.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.y), 0xB1, 0xf, 0xf, true));
        recv.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.z), 0xB1, 0xf, 0xf, true));
        recv.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.w), 0xB1, 0xf, 0xf, true));
        const float4 lo4 = odd ? recv : v[pp];      // columns c8 .. c8+3
        const float4 hi4 = odd ? v[pp + 1] : recv;  // columns c8+4 .. c8+7
        const float e[8] = {pin(lo4.x), pin(lo4.y), pin(lo4.z), pin(lo4.w), pin(hi4.x), pin(hi4.y), pin(hi4.z), pin(hi4.w)};
        typedef __attribute__((ext_vector_type(8))) _Float16 h8;
        h8 hh, ll;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          hh[t] = (_Float16)e[t];
          ll[t] = (_Float16)(e[t] - (float)hh[t]);
        }
        const int row = m0 + wm * 128 + i * 32 + (pp + (odd ? 1 : 0)) * 8 + rrow;
        const int c8 = n0 + wn * 64 + j * 32 + (rslot & 6) * 4;
        if (row < g.M && c8 < g.N) {
          unsigned char* dst = oa + ((long)row * g.ldc + c8) * 4;  // group of 8 elements = 32 bytes: hi plane, lo plane
          *(h8*)dst = hh;
          *(h8*)(dst + 16) = ll;
        }
      }
    }
  }
}

// Visit order of a persistent work-group.  Default ("legacy"): virtual tiles wg, wg+grid, ... through
// the XCD-aware map, i.e. the N tiles of an M tile run CONCURRENTLY on neighbouring CUs of one XCD.
// Alternative (debug bit5): one work-group walks all N tiles of an M tile back to back (A from HBM
// once, then L2) -- measured 9 % slower (0.684 vs 0.621 ms on the qkv shape), kept for A/B runs.
__device__ __forceinline__ int tile_count(int tiles_m, int tiles_n, bool legacy) {
  const int grid = gridDim.x, wg = blockIdx.x;
  if (legacy) {
    const int nt = tiles_m * tiles_n;
    return wg < nt ? (nt - 1 - wg) / grid + 1 : 0;
  }
  return wg < tiles_m ? ((tiles_m - 1 - wg) / grid + 1) * tiles_n : 0;
}
__device__ __forceinline__ void tile_at(int i, int tiles_m, int tiles_n, bool legacy, int& tm, int& tn) {
  const int grid = gridDim.x, wg = blockIdx.x;
  if (legacy) { tile_of(wg + i * grid, tiles_m, tiles_n, tm, tn); return; }
  const int round = i / tiles_n;
  tn = i - round * tiles_n;
  tm = wg + round * grid;
}

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(768) void gemm256p_kernel(GemmArgs g, int tiles_m, int tiles_n, int g_krot_enable) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 6;
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;

  if (wave >= 8) {
    // ------------------------------- loader waves -------------------------------------------
    const int lw = wave - 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
    // instruction ii (0..7) lands tile rows lw*64 + ii*8 + (lane>>3); swizzle phase depends on ii&1
    const int rbase = lw * 64 + (lane >> 3);
    const int ce = ((lane & 7) ^ ((lane >> 4) & 7)) << 4;        // ii even: (row>>1)&7 = lane>>4
    const int co = ((lane & 7) ^ (((lane >> 4) + 4) & 7)) << 4;  // ii odd : +4
    const int a_e = rbase * lda_b + ce, a_o = (rbase + 8) * lda_b + co;
    const int w_e = rbase * ldw_b + ce, w_o = (rbase + 8) * ldw_b + co;
    const int a16 = 16 * lda_b, w16 = 16 * ldw_b;
    unsigned step = 0;
    bool first = true;
    const int krot = (g_krot_enable & 1) ? (int)((blockIdx.x >> 3) % (unsigned)nk) : 0;
    const bool dbg_no_dma = g_krot_enable & 4;
    const bool legacy = (g_krot_enable & 32) == 0;  // bit5 selects the (slower, measured) M-major walk
    const int my_tiles = tile_count(tiles_m, tiles_n, legacy);
    for (int ti = 0; ti < my_tiles; ++ti) {
      int tm, tn;
      tile_at(ti, tiles_m, tiles_n, legacy, tm, tn);
      const int m0 = tm * TM, n0 = tn * TN;
      const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
      const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
      u32x4_t rsA, rsW;
      rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu;
      rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b); rsA.w = 0x00020000u;
      rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu;
      rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b); rsW.w = 0x00020000u;
      if (g_krot_enable & 8) rsA.z = 0;   // debug: out-of-range descriptor -> zeros, no A traffic
      if (g_krot_enable & 16) rsW.z = 0;  // debug: no W traffic
      for (int kt = 0; kt < nk; ++kt, ++step) {
        if (!first) {
          // previous step's DMA has landed -> publish it; the same barrier proves the stage this
          // step is written to has been left by every MFMA wave
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        first = false;
        const unsigned dstA = lds0 + (step & 1) * STAGE + lw * (64 * ROWB);
        const unsigned dstW = dstA + A_BYTES;
        // K steps are visited in a per-work-group rotated order (a sum is order-free): work-groups
        // that share a W tile would otherwise all request the same lines of it at the same moment
        int kr = kt + krot;
        if (kr >= nk) kr -= nk;
        const unsigned so = kr * ROWB;
        unsigned keep;
        if (!dbg_no_dma) asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(dstA), "s"(dstW), "v"(a_e), "v"(a_o), "v"(a_e + a16), "v"(a_o + a16), "v"(w_e), "v"(w_o),
              "v"(w_e + w16), "v"(w_o + w16), "s"(rsA), "s"(rsW), "s"(so)
            : "memory", "scc");
        if (!dbg_no_dma) asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(dstA + 32 * ROWB), "s"(dstW + 32 * ROWB), "v"(a_e + 2 * a16), "v"(a_o + 2 * a16), "v"(a_e + 3 * a16),
              "v"(a_o + 3 * a16), "v"(w_e + 2 * w16), "v"(w_o + 2 * w16), "v"(w_e + 3 * w16), "v"(w_o + 3 * w16),
              "s"(rsA), "s"(rsW), "s"(so)
            : "memory", "scc");
      }
    }
    if (!first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // publishes the very last step
    }
    return;
  }

  // --------------------------------- MFMA waves ---------------------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  unsigned step = 0;
  const bool legacy = (g_krot_enable & 32) == 0;  // bit5 selects the (slower, measured) M-major walk
  const int my_tiles = tile_count(tiles_m, tiles_n, legacy);
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, legacy, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      __builtin_amdgcn_s_barrier();  // step's tile is in LDS (loaders waited for it before arriving)
      asm volatile("" ::: "memory");
      const unsigned char* sA = smem + (step & 1) * STAGE;
      const unsigned char* sB = sA + A_BYTES;
      if (g_krot_enable & 2) continue;  // debug: fill-rate measurement without MFMA work
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + half;
        uint4 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const uint4*)(sB + swz(brow + 32 * j, ch));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b[j]),
                                                                __builtin_bit_cast(bf16x8_t, a[i]), acc[i][j], 0, 0, 0);
      }
    }
    // epilogue (no barrier inside: the loaders are already fetching the next tile's first K step)
    tile_epilogue<ACT, OUT_F32>(g, acc, smem + 2 * STAGE + wave * 4096, m0, n0, wm, wn, lane);
  }
}

// ================================================================================================
// gemm256q: gemm256p with a 4-deep ring of 32-wide K steps (4 x 32 KiB stages) instead of two
// 64-wide stages.  The loaders run up to three steps ahead and wait with a COUNTED vmcnt, so a
// step costs max(DMA issue, DMA latency, MFMA) instead of issue + latency: in gemm256p the 16 DMA
// issues of a step (~1000 cycles) and the landing of the last one are serialised in front of every
// barrier.
//   LDS rows are 64 bytes (32 bf16): 16-byte chunk c of row r sits at chunk c ^ ((r>>2)&3)
//   (16 distinct rows of a ds_read_b128 lane group -> 16 distinct slots of the 256-byte bank row);
//   one DMA instruction lands 16 rows.
// ================================================================================================
constexpr int QROWB = 64;                      // bytes per tile row per K step (32 bf16)
constexpr int QA_BYTES = TM * QROWB;           // 16 KiB
constexpr int QSTAGE = QA_BYTES + TN * QROWB;  // 32 KiB
constexpr int QS = 4;                          // ring depth

__device__ __forceinline__ int swzq(int row, int chunk) { return row * QROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int ACT, bool OUT_F32, bool STATS = false, bool F16 = false>
__global__ __launch_bounds__(768) void gemm256q_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;  // operand element type (common.h Half<>)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;  // 32-wide K steps
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int my_tiles = tile_count(tiles_m, tiles_n, true);
  const int total = my_tiles * nk;

  if (wave >= 8) {
    // ------------------------------- loader waves -------------------------------------------
    const int lw = wave - 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
    // instruction ii (0..3) lands tile rows lw*64 + ii*16 + (lane>>2); physical chunk lane&3
    const int rbase = lw * 64 + (lane >> 2);
    const int cq = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const int a0 = rbase * lda_b + cq, w0 = rbase * ldw_b + cq;
    const int a16 = 16 * lda_b, w16 = 16 * ldw_b;
    int cur_ti = -1;
    u32x4_t rsA, rsW;
    rsA.x = rsA.y = rsA.z = 0; rsA.w = 0x00020000u;
    rsW = rsA;
    auto issue = [&](int s) {
      const int ti = s / nk, kt = s - ti * nk;
      if (ti != cur_ti) {
        cur_ti = ti;
        int tm, tn;
        tile_at(ti, tiles_m, tiles_n, true, tm, tn);
        const int m0 = tm * TM, n0 = tn * TN;
        const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
        const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
        rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b);
        rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b);
      }
      const unsigned dstA = lds0 + (s & (QS - 1)) * QSTAGE + lw * (64 * QROWB);
      const unsigned dstW = dstA + QA_BYTES;
      const unsigned so = kt * QROWB;
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "s"(dstA), "s"(dstW), "v"(a0), "v"(a0 + a16), "v"(a0 + 2 * a16), "v"(a0 + 3 * a16), "v"(w0), "v"(w0 + w16),
            "v"(w0 + 2 * w16), "v"(w0 + 3 * w16), "s"(rsA), "s"(rsW), "s"(so)
          : "memory", "scc");
    };
    const int pro = total < QS ? total : QS;
    for (int s = 0; s < pro; ++s) issue(s);
    for (int s = 0; s < total; ++s) {
      // steps issued so far: 0 .. min(total, QS + max(s-1,0)) - 1; wait until step s has landed, i.e.
      // at most (issued - 1 - s) later steps (8 DMAs each) may still be in flight
      const int issued = (s == 0) ? pro : (QS + s - 1 < total ? QS + s - 1 : total);
      const int later = issued - 1 - s;
      if (later >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // publishes step s; proves stage (s-1)%QS has been left
      if (s >= 1 && s - 1 + QS < total) issue(s - 1 + QS);
    }
    return;
  }

  // --------------------------------- MFMA waves ---------------------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  int step = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, true, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* sA = smem + (step & (QS - 1)) * QSTAGE;
      const unsigned char* sB = sA + QA_BYTES;
      // Fragment stream written in software-pipelined order (weights b0,b1 / c0,c1 held, activation
      // fragments through a two-deep ring).  hipcc re-serialises it to one activation quad + lgkmcnt(0)
      // per MFMA pair because the kernel sits at the 168-VGPR cap of a 12-wave work-group; pinning the
      // order with sched_group_barrier produced the intended stream but 19 spills and no net gain
      // (measured 656 vs 658 TF/s over the four layer shapes), so the order is left to the compiler.
#define CZC_RA(ks_, i_) (*(const uint4*)(sA + swzq(arow + 32 * (i_), 2 * (ks_) + half)))
#define CZC_RB(ks_, j_) (*(const uint4*)(sB + swzq(brow + 32 * (j_), 2 * (ks_) + half)))
#define CZC_MM(b_, a_, i_, j_)                                                                                  \
  acc[i_][j_] = Half<HT>::mfma(b_, a_, acc[i_][j_])
      uint4 b0 = CZC_RB(0, 0), b1 = CZC_RB(0, 1), a0 = CZC_RA(0, 0), a1;
      a1 = CZC_RA(0, 1);
      CZC_MM(b0, a0, 0, 0); CZC_MM(b1, a0, 0, 1);
      a0 = CZC_RA(0, 2);
      CZC_MM(b0, a1, 1, 0); CZC_MM(b1, a1, 1, 1);
      a1 = CZC_RA(0, 3);
      CZC_MM(b0, a0, 2, 0); CZC_MM(b1, a0, 2, 1);
      uint4 c0 = CZC_RB(1, 0), c1 = CZC_RB(1, 1);
      a0 = CZC_RA(1, 0);
      CZC_MM(b0, a1, 3, 0); CZC_MM(b1, a1, 3, 1);
      a1 = CZC_RA(1, 1);
      CZC_MM(c0, a0, 0, 0); CZC_MM(c1, a0, 0, 1);
      a0 = CZC_RA(1, 2);
      CZC_MM(c0, a1, 1, 0); CZC_MM(c1, a1, 1, 1);
      a1 = CZC_RA(1, 3);
      CZC_MM(c0, a0, 2, 0); CZC_MM(c1, a0, 2, 1);
      CZC_MM(c0, a1, 3, 0); CZC_MM(c1, a1, 3, 1);
#undef CZC_RA
#undef CZC_RB
#undef CZC_MM
    }
    tile_epilogue<ACT, OUT_F32, STATS, HT>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, wm, wn, lane);
  }
}



// ================================================================================================
// gemm256w: the 4-deep ring of gemm256q WITHOUT loader waves.  Ablations of the loader-wave kernels
// show that their inner loop cannot be software-pipelined at the 168-VGPR cap of a 12-wave work-group and tops out near
// 1.0 PFLOP/s even with free operands.  Here the work-group is 8 waves with the full 256 registers each: every wave
// multiplies AND lands its share of the next stages -- rows 32w .. 32w+31 of the A and of the W tile, two 1 KiB LDS-DMA
// pieces each per stage -- three stages ahead, counted `vmcnt` for its own pieces at the top of a step, one barrier per
// step.  The two waves of a SIMD (w, w+4) issue their pieces half a step apart (start / middle of the step), so one
// of them always has MFMAs for the matrix pipe while the other sits in the vector-memory issue path.
// ================================================================================================
// DBG (timing ablations only, results are garbage): 1 no MFMA, 2 no MFMA + pieces of 8 rows x 128 B, 3 no DMA.
template <int ACT, bool OUT_F32, bool F16 = false, int DBG = 0>
__global__ __launch_bounds__(512) void gemm256w_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;  // 32-wide K steps
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int my_tiles = tile_count(tiles_m, tiles_n, true);
  const int total = my_tiles * nk;
  const bool late = DBG == 8 ? false : wave >= 4;

  // ---- DMA side: pieces ii = 0, 1 land tile rows wave*32 + ii*16 + (lane>>2); physical chunk lane&3 ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  const int rbase = wave * 32 + (lane >> 2);
  const int cq = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const int a0 = DBG == 2 ? (wave * 16 + (lane >> 3)) * lda_b + (lane & 7) * 16 : rbase * lda_b + cq;
  const int w0 = DBG == 2 ? (wave * 16 + (lane >> 3)) * ldw_b + (lane & 7) * 16 : rbase * ldw_b + cq;
  const int a16 = (DBG == 2 ? 8 : 16) * lda_b, w16 = (DBG == 2 ? 8 : 16) * ldw_b;
  int cur_ti = -1;
  u32x4_t rsA, rsW;
  rsA.x = rsA.y = rsA.z = 0; rsA.w = 0x00020000u;
  rsW = rsA;
  auto issue = [&](int s) {
    const int ti = s / nk, kt = s - ti * nk;
    if (ti != cur_ti) {
      cur_ti = ti;
      int tm, tn;
      tile_at(ti, tiles_m, tiles_n, true, tm, tn);
      const int m0 = tm * TM, n0 = tn * TN;
      const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
      const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
      rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b);
      rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b);
    }
    const unsigned dstA = lds0 + (s & (QS - 1)) * QSTAGE + wave * (32 * QROWB);
    const unsigned dstW = dstA + QA_BYTES;
    const unsigned soA = DBG == 2 ? (kt >> 1) * 128 + (kt & 1) * 128 * lda_b : kt * QROWB;
    const unsigned soW = DBG == 2 ? (kt >> 1) * 128 + (kt & 1) * 128 * ldw_b : kt * QROWB;
    if (DBG == 3) return;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %9 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %9 offen lds\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %10 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %10 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dstA), "s"(dstW), "v"(a0), "v"(a0 + a16), "v"(w0), "v"(w0 + w16), "s"(rsA), "s"(rsW), "s"(soA), "s"(soW)
        : "memory", "scc");
  };
  for (int s = 0; s < QS - 1 && s < total; ++s) issue(s);

  // ---- MFMA side: fragments run half a step ahead of the MFMAs, across the barrier ----
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
#define CZC_RFRAG(st_, ks_, a_, b_)                                                                                      \
  do {                                                                                                                   \
    const unsigned char* sA_ = smem + ((st_) & (QS - 1)) * QSTAGE;                                                       \
    const unsigned char* sB_ = sA_ + QA_BYTES;                                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_[j] = *(const uint4*)(sB_ + swzq(brow + 32 * j, 2 * (ks_) + half));  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) a_[i] = *(const uint4*)(sA_ + swzq(arow + 32 * i, 2 * (ks_) + half));  \
  } while (0)
#define CZC_MMA8(a_, b_)                                                                                                 \
  do {                                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(b_[j], a_[i], acc[i][j]);                 \
  } while (0)
  uint4 ca[4], cb[2];  // k16 #0 of the step about to run, read during the step before
  if (total > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (total == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // stage 0 published
  asm volatile("" ::: "memory");
  CZC_RFRAG(0, 0, ca, cb);
  int step = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, true, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      // own pieces of stage step+1 landed?  The only pieces issued after them are stage step+2's four (step+3 goes out
      // below); epilogue traffic of a tile boundary in between only makes the count conservative.
      if (step + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // publishes stage step+1; every wave has left stage step-1, whose slot is refilled now
      asm volatile("" ::: "memory");
      const bool refill = step + QS - 1 < total;
      if (refill && !late) issue(step + QS - 1);
      uint4 na[4], nb[2];
      if (DBG != 1 && DBG != 2) CZC_RFRAG(step, 1, na, nb);
      __builtin_amdgcn_sched_barrier(0);  // the reads stay ahead of the MFMAs that hide them
      if (DBG == 9) __builtin_amdgcn_s_setprio(1);
      if (DBG != 1 && DBG != 2) CZC_MMA8(ca, cb);
      if (DBG == 9) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (refill && late) issue(step + QS - 1);
      if (DBG != 1 && DBG != 2) CZC_RFRAG(step + 1, 0, ca, cb);  // past the last stage: a stale slot, never multiplied
      __builtin_amdgcn_sched_barrier(0);
      if (DBG == 9) __builtin_amdgcn_s_setprio(1);
      if (DBG != 1 && DBG != 2) CZC_MMA8(na, nb);
      if (DBG == 9) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    tile_epilogue<ACT, OUT_F32, false, HT>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, wm, wn, lane);
  }
#undef CZC_RFRAG
#undef CZC_MMA8
}

// ================================================================================================
// gemm256sq: gemm256q's 4-deep ring of 32 KiB stages for the SPLIT-fp16 precision.  A 64-byte tile row holds 16
// elements ([8 hi | 8 lo] x 2 groups) = one k16 MFMA step, three fp16 passes per product: 24 MFMAs per stage and
// wave on 12 ds_read_b128, loaders up to three stages ahead with counted vmcnt.  (gemm256s below, the two-stage
// form, waits for every stage's DMA latency in front of its barrier: 0.37 of MFMA peak; kept for A/B.)
// ================================================================================================
template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(768) void gemm256sq_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 4;  // one k16 MFMA step per stage: a 64-byte row = two split_t groups = 16 elements
  const int lda_b = g.lda * 4, ldw_b = g.ldw * 4;
  const int my_tiles = tile_count(tiles_m, tiles_n, true);
  const int total = my_tiles * nk;

  if (wave >= 8) {
    // ------------------------------- loader waves -------------------------------------------
    const int lw = wave - 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
    // instruction ii (0..3) lands tile rows lw*64 + ii*16 + (lane>>2); physical chunk lane&3
    const int rbase = lw * 64 + (lane >> 2);
    const int cq = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const int a0 = rbase * lda_b + cq, w0 = rbase * ldw_b + cq;
    const int a16 = 16 * lda_b, w16 = 16 * ldw_b;
    int cur_ti = -1;
    u32x4_t rsA, rsW;
    rsA.x = rsA.y = rsA.z = 0; rsA.w = 0x00020000u;
    rsW = rsA;
    auto issue = [&](int s) {
      const int ti = s / nk, kt = s - ti * nk;
      if (ti != cur_ti) {
        cur_ti = ti;
        int tm, tn;
        tile_at(ti, tiles_m, tiles_n, true, tm, tn);
        const int m0 = tm * TM, n0 = tn * TN;
        const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
        const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
        rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b);
        rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b);
      }
      const unsigned dstA = lds0 + (s & (QS - 1)) * QSTAGE + lw * (64 * QROWB);
      const unsigned dstW = dstA + QA_BYTES;
      const unsigned so = kt * QROWB;
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "s"(dstA), "s"(dstW), "v"(a0), "v"(a0 + a16), "v"(a0 + 2 * a16), "v"(a0 + 3 * a16), "v"(w0), "v"(w0 + w16),
            "v"(w0 + 2 * w16), "v"(w0 + 3 * w16), "s"(rsA), "s"(rsW), "s"(so)
          : "memory", "scc");
    };
    const int pro = total < QS ? total : QS;
    for (int s = 0; s < pro; ++s) issue(s);
    for (int s = 0; s < total; ++s) {
      // steps issued so far: 0 .. min(total, QS + max(s-1,0)) - 1; wait until step s has landed, i.e.
      // at most (issued - 1 - s) later steps (8 DMAs each) may still be in flight
      const int issued = (s == 0) ? pro : (QS + s - 1 < total ? QS + s - 1 : total);
      const int later = issued - 1 - s;
      if (later >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // publishes step s; proves stage (s-1)%QS has been left
      if (s >= 1 && s - 1 + QS < total) issue(s - 1 + QS);
    }
    return;
  }

  // --------------------------------- MFMA waves ---------------------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  int step = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, true, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* sA = smem + (step & (QS - 1)) * QSTAGE;
      const unsigned char* sB = sA + QA_BYTES;
      // row = [hi g0 | lo g0 | hi g1 | lo g1] (16-byte chunks); lane half h contracts group h: chunks 2h (hi), 2h+1 (lo)
#define CZC_F16(v_) __builtin_bit_cast(f16x8_t, v_)
      uint4 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *(const uint4*)(sB + swzq(brow + 32 * j, 2 * half));
        bl[j] = *(const uint4*)(sB + swzq(brow + 32 * j, 2 * half + 1));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 ah = *(const uint4*)(sA + swzq(arow + 32 * i, 2 * half));
        const uint4 al = *(const uint4*)(sA + swzq(arow + 32 * i, 2 * half + 1));
        // pass-major over the two column tiles: consecutive MFMAs never share an accumulator
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bl[j]), CZC_F16(ah), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bh[j]), CZC_F16(al), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bh[j]), CZC_F16(ah), acc[i][j], 0, 0, 0);
      }
#undef CZC_F16
    }
    unsigned char* patch = smem + QS * QSTAGE + wave * 4096;
    if (OUT_F32) tile_epilogue<ACT, true>(g, acc, patch, m0, n0, wm, wn, lane);
    else tile_epilogue_split<ACT>(g, acc, patch, m0, n0, wm, wn, lane);
  }
}




// ================================================================================================
// gemm256s: the persistent wave-specialised 256x256 kernel (gemm256p structure: 8 MFMA waves + 4 LDS-DMA loader
// waves, two 64 KiB stages of 128-byte rows) for the SPLIT-fp16 engine precision.  A 128-byte tile row is 32
// elements as four groups of [8 fp16 hi | 8 fp16 lo] (common.h split_t), so one stage feeds two k16 MFMA steps, and
// every product is three v_mfma_f32_32x32x16_f16 passes (hi*hi + lo*hi + hi*lo): 48 MFMAs per stage and wave on
// 24 ds_read_b128 -- three times the matrix work of the bf16 kernel on twice the bytes, which moves these layers
// from the HBM / CU-ingest bound of the bf16 tower towards the MFMA bound.
// ================================================================================================
template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(768) void gemm256s_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;  // 32 elements (128 bytes of split_t) per stage
  const int lda_b = g.lda * 4, ldw_b = g.ldw * 4;
  const int my_tiles = tile_count(tiles_m, tiles_n, true);

  if (wave >= 8) {
    // ------------------------------- loader waves (as gemm256p) -----------------------------
    const int lw = wave - 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
    const int rbase = lw * 64 + (lane >> 3);
    const int ce = ((lane & 7) ^ ((lane >> 4) & 7)) << 4;
    const int co = ((lane & 7) ^ (((lane >> 4) + 4) & 7)) << 4;
    const int a_e = rbase * lda_b + ce, a_o = (rbase + 8) * lda_b + co;
    const int w_e = rbase * ldw_b + ce, w_o = (rbase + 8) * ldw_b + co;
    const int a16 = 16 * lda_b, w16 = 16 * ldw_b;
    unsigned step = 0;
    bool first = true;
    for (int ti = 0; ti < my_tiles; ++ti) {
      int tm, tn;
      tile_at(ti, tiles_m, tiles_n, true, tm, tn);
      const int m0 = tm * TM, n0 = tn * TN;
      const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
      const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
      u32x4_t rsA, rsW;
      rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu;
      rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b); rsA.w = 0x00020000u;
      rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu;
      rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b); rsW.w = 0x00020000u;
      for (int kt = 0; kt < nk; ++kt, ++step) {
        if (!first) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        first = false;
        const unsigned dstA = lds0 + (step & 1) * STAGE + lw * (64 * ROWB);
        const unsigned dstW = dstA + A_BYTES;
        const unsigned so = kt * ROWB;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(dstA), "s"(dstW), "v"(a_e), "v"(a_o), "v"(a_e + a16), "v"(a_o + a16), "v"(w_e), "v"(w_o),
              "v"(w_e + w16), "v"(w_o + w16), "s"(rsA), "s"(rsW), "s"(so)
            : "memory", "scc");
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %11, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %11, %13 offen lds\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %12, %13 offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(dstA + 32 * ROWB), "s"(dstW + 32 * ROWB), "v"(a_e + 2 * a16), "v"(a_o + 2 * a16), "v"(a_e + 3 * a16),
              "v"(a_o + 3 * a16), "v"(w_e + 2 * w16), "v"(w_o + 2 * w16), "v"(w_e + 3 * w16), "v"(w_o + 3 * w16),
              "s"(rsA), "s"(rsW), "s"(so)
            : "memory", "scc");
      }
    }
    if (!first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // --------------------------------- MFMA waves ---------------------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  unsigned step = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, true, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* sA = smem + (step & 1) * STAGE;
      const unsigned char* sB = sA + A_BYTES;
#define CZC_F16(v_) __builtin_bit_cast(f16x8_t, v_)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ch = 2 * (2 * s2 + half);  // chunk 2g = hi plane, 2g+1 = lo plane of k-group g = 2*s2 + half
        uint4 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[j] = *(const uint4*)(sB + swz(brow + 32 * j, ch));
          bl[j] = *(const uint4*)(sB + swz(brow + 32 * j, ch + 1));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint4 ah = *(const uint4*)(sA + swz(arow + 32 * i, ch));
          const uint4 al = *(const uint4*)(sA + swz(arow + 32 * i, ch + 1));
          // (a_hi + a_lo)(w_hi + w_lo) ~ a_hi w_hi + a_lo w_hi + a_hi w_lo   (lo*lo ~ 2^-22, dropped); small terms first,
          // pass-major over the two column tiles so that consecutive MFMAs never share an accumulator
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bl[j]), CZC_F16(ah), acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bh[j]), CZC_F16(al), acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(bh[j]), CZC_F16(ah), acc[i][j], 0, 0, 0);
        }
      }
#undef CZC_F16
    }
    unsigned char* patch = smem + 2 * STAGE + wave * 4096;
    if (OUT_F32) tile_epilogue<ACT, true>(g, acc, patch, m0, n0, wm, wn, lane);
    else tile_epilogue_split<ACT>(g, acc, patch, m0, n0, wm, wn, lane);
  }
}

// ================================================================================================
// gemm_rowln: x <- x + A.W^T + b over FULL 512-wide rows, with the LayerNorm that follows it in the pre-LN block
// finished in the epilogue (HF:clip/modeling_clip.py:368-383: out-proj -> LN2 -> fc1, fc2 -> next layer's LN1).
// A 128 x 512 tile per work-group: 8 waves, wave w owns columns 64w .. 64w+63 of all 128 rows (the per-wave 128 x 64
// block of the 256 x 256 kernels), lands W rows 64w .. 64w+63 (which only it reads) and 16 of the 128 A rows.  Two
// LDS rings -- A 4 x 8 KiB, W 3 x 32 KiB -- plus the eight 4 KiB epilogue patches fill the 160 KiB.  Per step a wave
// issues A(step+3) then W(step+2) x 4, so `vmcnt(5)` at the top of a step says its W(step) pieces (and the older
// A(step)) have landed.  The epilogue writes the fp32 row AND keeps it in the registers the accumulators leave, takes
// the exact two-pass statistics of layernorm_kernel (mean, then centred squares) through two 512-byte exchanges in
// the patches, and stores y = LN(x) in the activation type: the LayerNorm kernel and its 2 KiB-per-row read go away.
// ================================================================================================
constexpr int RL_TM = 128, RL_N = 512;
constexpr int RL_AS = 4, RL_WS = 3;
constexpr int RL_A_BYTES = RL_TM * QROWB;  // 8 KiB
constexpr int RL_W_BYTES = RL_N * QROWB;   // 32 KiB
constexpr int RL_W_BASE = RL_AS * RL_A_BYTES;
constexpr int RL_PATCH = RL_W_BASE + RL_WS * RL_W_BYTES;
constexpr int RL_LDS = RL_PATCH + 8 * 4096;  // 160 KiB

template <bool F16>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(GemmArgs g, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int tiles_m) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int my_tiles = (tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  const bool late = wave >= 4;

  // ---- DMA side ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  const int cq = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const int a0 = (wave * 16 + (lane >> 2)) * lda_b + cq;
  const int w0 = (wave * 64 + (lane >> 2)) * ldw_b + cq;
  const int w16 = 16 * ldw_b;
  u32x4_t rsA, rsW;
  rsA.x = rsA.y = rsA.z = 0; rsA.w = 0x00020000u;
  {
    const unsigned long long pw = (unsigned long long)g.W;
    rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(RL_N * ldw_b); rsW.w = 0x00020000u;
  }
  int a_ti = -1;
  auto issueA = [&](int s) {
    const int ti = s / nk, kt = s - ti * nk;
    if (ti != a_ti) {
      a_ti = ti;
      const int m0 = ((int)blockIdx.x + ti * (int)gridDim.x) * RL_TM;
      const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
      rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(min(RL_TM, g.M - m0) * lda_b);
    }
    const unsigned dst = lds0 + (s & (RL_AS - 1)) * RL_A_BYTES + wave * (16 * QROWB);
    const unsigned so = kt * QROWB;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst), "v"(a0), "s"(rsA), "s"(so)
        : "memory", "scc");
  };
  auto issueW = [&](int s, int slot) {
    const int kt = s % nk;
    const unsigned dst = lds0 + RL_W_BASE + slot * RL_W_BYTES + wave * (64 * QROWB);
    const unsigned so = kt * QROWB;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst), "v"(w0), "v"(w0 + w16), "v"(w0 + 2 * w16), "v"(w0 + 3 * w16), "s"(rsW), "s"(so)
        : "memory", "scc");
  };
  // virtual steps -3, -2, -1 of the steady-state order: A0 | A1 W0 | A2 W1
  if (0 < total) issueA(0);
  if (1 < total) issueA(1);
  if (0 < total) issueW(0, 0);
  if (2 < total) issueA(2);
  if (1 < total) issueW(1, 1);

  // ---- MFMA side ----
  const int half = lane >> 5, l31 = lane & 31;
  const int brow = wave * 64 + l31;
  const int rrow = lane >> 3, rslot = lane & 7;
  unsigned char* patch = smem + RL_PATCH + wave * 4096;
  int step = 0, wslot = 0;  // wslot = step % 3
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int m0 = ((int)blockIdx.x + ti * (int)gridDim.x) * RL_TM;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++step) {
      if (step + 2 < total) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (step + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // publishes A(step); every wave has left step-1, whose two slots are refilled now
      asm volatile("" ::: "memory");
      const int wnext = wslot == 0 ? 2 : wslot - 1;  // (step + 2) % 3
      if (!late) {
        if (step + 3 < total) issueA(step + 3);
        if (step + 2 < total) issueW(step + 2, wnext);
      }
      const unsigned char* sA = smem + (step & (RL_AS - 1)) * RL_A_BYTES;
      const unsigned char* sB = smem + RL_W_BASE + wslot * RL_W_BYTES;
      uint4 b0[2], a0f[4], b1[2], a1f[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) b0[j] = *(const uint4*)(sB + swzq(brow + 32 * j, half));
#pragma unroll
      for (int i = 0; i < 4; ++i) a0f[i] = *(const uint4*)(sA + swzq(l31 + 32 * i, half));
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[j] = *(const uint4*)(sB + swzq(brow + 32 * j, 2 + half));
#pragma unroll
      for (int i = 0; i < 4; ++i) a1f[i] = *(const uint4*)(sA + swzq(l31 + 32 * i, 2 + half));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(b0[j], a0f[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      if (late) {
        if (step + 3 < total) issueA(step + 3);
        if (step + 2 < total) issueW(step + 2, wnext);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(b1[j], a1f[i], acc[i][j]);
      wslot = wslot == 2 ? 0 : wslot + 1;
    }

    // ---- epilogue: x (fp32) out, exact LayerNorm statistics across the eight waves, y out ----
    float4 vx[4][2][4];  // [i][j][pass]: row i*32 + pass*8 + rrow, columns wave*64 + j*32 + rslot*4 .. +3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wave * 64 + j * 32 + rslot * 4;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) b4 = *(const float4*)(g.bias + col);
        float4 r4[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = m0 + i * 32 + pass * 8 + rrow;
          r4[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.resid && row < g.M) r4[pass] = *(const float4*)(g.resid + (long)row * g.ldr + col);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = (2 * q + half) ^ (l31 & 7);
          *(float4*)(patch + l31 * 128 + slot * 16) =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = pass * 8 + rrow;
          float4 v = *(const float4*)(patch + r * 128 + ((rslot ^ (r & 7)) << 4));
          const int row = m0 + i * 32 + r;
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          v.x += r4[pass].x; v.y += r4[pass].y; v.z += r4[pass].z; v.w += r4[pass].w;
          if (row < g.M && g.out_f32) *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v;
          vx[i][j][pass] = v;
        }
      }
    }
    // mean: this wave's 64-column partial per row -> its patch [0, 512); then every row group sums the eight partials
    float mean[4][4], rstd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const float4 u = vx[i][0][pass], w = vx[i][1][pass];
        const float s8 = sum8_dpp(((u.x + u.y) + (u.z + u.w)) + ((w.x + w.y) + (w.z + w.w)));
        if (rslot == 0) *(float*)(patch + (i * 32 + pass * 8 + rrow) * 4) = s8;
      }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const float t = *(const float*)(smem + RL_PATCH + rslot * 4096 + (i * 32 + pass * 8 + rrow) * 4);
        mean[i][pass] = sum8_dpp(t) / (float)RL_N;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const float mu = mean[i][pass];
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float4 u = vx[i][j][pass];
          const float a = u.x - mu, b = u.y - mu, c = u.z - mu, d = u.w - mu;
          q += (a * a + b * b) + (c * c + d * d);
        }
        const float q8 = sum8_dpp(q);
        if (rslot == 0) *(float*)(patch + 512 + (i * 32 + pass * 8 + rrow) * 4) = q8;
      }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const float t = *(const float*)(smem + RL_PATCH + rslot * 4096 + 512 + (i * 32 + pass * 8 + rrow) * 4);
        rstd[i][pass] = rsqrtf(sum8_dpp(t) / (float)RL_N + g.ln_eps);
      }
    HT* oa = (HT*)g.out_act;
    const bool odd = rslot & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 64 + j * 32 + rslot * 4;
      const float4 gm = *(const float4*)(gamma + col);
      const float4 bt = *(const float4*)(beta + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 pk[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const float4 u = vx[i][j][pass];
          const float mu = mean[i][pass], rs = rstd[i][pass];
          pk[pass] = make_uint2(Half<HT>::pack2((u.x - mu) * rs * gm.x + bt.x, (u.y - mu) * rs * gm.y + bt.y),
                                Half<HT>::pack2((u.z - mu) * rs * gm.z + bt.z, (u.w - mu) * rs * gm.w + bt.w));
        }
        // 16-byte stores: lanes rslot, rslot^1 hold adjacent 4-column pieces of the same rows; the even lane ends
        // up with 8 columns of the first pass's row of a pair, the odd lane with 8 columns of the second's
#pragma unroll
        for (int pp = 0; pp < 4; pp += 2) {
          const uint2 send = odd ? pk[pp] : pk[pp + 1];
          uint2 recv;
          recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xf, 0xf, true);
          recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xf, 0xf, true);
          const uint4 d = odd ? make_uint4(recv.x, recv.y, pk[pp + 1].x, pk[pp + 1].y)
                              : make_uint4(pk[pp].x, pk[pp].y, recv.x, recv.y);
          const int row = m0 + i * 32 + (pp + (odd ? 1 : 0)) * 8 + rrow;
          const int c8 = wave * 64 + j * 32 + (rslot & 6) * 4;
          if (row < g.M) *(uint4*)(oa + (long)row * RL_N + c8) = d;
        }
      }
    }
  }
}

}  // namespace

int g_gemm256_min_m = 2048;
int g_gemm_krot = 0;  // bit0: rotate K order per work-group (no gain measured); bits1-2: debug (skip MFMA / skip DMA)

// x <- x + A.W^T + b with y = LayerNorm(x) from the same launch (gemm_rowln_kernel): N = 512 rows only.
int g_rowln_min_m = 4096;
bool gemm_rowln_eligible(const GemmArgs& g) {
  return g.M >= g_rowln_min_m && g.N == RL_N && g.K % 32 == 0 && g.K >= 64 && g.ldc == RL_N && g.act == ACT_NONE && g.out_act &&
         g.out_f32 && g.ln_gamma && g.ln_beta && !g.row_stats && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         (!g.resid || g.ldr % 4 == 0) && (long)RL_TM * g.lda * 2 < (1L << 31) && (long)RL_N * g.ldw * 2 < (1L << 31);
}
int launch_gemm_rowln(const GemmArgs& g, hipStream_t st) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    CZC_HIP_CHECK(hipGetDevice(&dev));
    CZC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
  }
  if (!gemm_rowln_eligible(g)) {
    snprintf(g_err, sizeof(g_err), "gemm_rowln: shape not eligible (M=%d N=%d K=%d)", g.M, g.N, g.K);
    return 1;
  }
  const int tiles_m = cdiv(g.M, RL_TM);
  dim3 grid(tiles_m < n_cu ? tiles_m : n_cu), block(512);
  if (g.f16) hipLaunchKernelGGL(gemm_rowln_kernel<true>, grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m);
  else hipLaunchKernelGGL(gemm_rowln_kernel<false>, grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

bool gemm256_eligible(const GemmArgs& g) {
  if (g.f16 && (g_use_gemm256 < 3 || g.row_stats || g.N % 8 || g.ldc % 8)) return false;  // fp16 operands: ring kernels only
  return g.M >= g_gemm256_min_m && g.N % 4 == 0 && g.K % 64 == 0 && g.ldc % 4 == 0 && (!g.resid || g.ldr % 4 == 0) &&
         (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) && (long)256 * g.lda * 2 < (1L << 31) &&
         (long)256 * g.ldw * 2 < (1L << 31);
}

int launch_gemm256(const GemmArgs& g, hipStream_t st) {
  static bool attr_set = false;
  static int n_cu = 0;
  const int shmem = 2 * STAGE;
  if (g_use_gemm256 >= 2 && g.N % 8 == 0 && g.ldc % 8 == 0 && g.K % 64 == 0) {
    const int shp = shmem + 8 * 4096;  // + one 4 KiB epilogue patch per MFMA wave = the full 160 KiB
    if (!n_cu) {
      int dev = 0;
      hipDeviceProp_t prop;
      CZC_HIP_CHECK(hipGetDevice(&dev));
      CZC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      n_cu = prop.multiProcessorCount;
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, shp))
      CZC_ATTR((gemm256p_kernel<ACT_NONE, false>));
      CZC_ATTR((gemm256p_kernel<ACT_NONE, true>));
      CZC_ATTR((gemm256p_kernel<ACT_QUICK_GELU, false>));
      CZC_ATTR((gemm256p_kernel<ACT_QUICK_GELU, true>));
      CZC_ATTR((gemm256q_kernel<ACT_NONE, false>));
      CZC_ATTR((gemm256q_kernel<ACT_NONE, true>));
      CZC_ATTR((gemm256q_kernel<ACT_QUICK_GELU, false>));
      CZC_ATTR((gemm256q_kernel<ACT_QUICK_GELU, true>));
      CZC_ATTR((gemm256q_kernel<ACT_NONE, true, true>));
      CZC_ATTR((gemm256q_kernel<ACT_NONE, false, false, true>));
      CZC_ATTR((gemm256q_kernel<ACT_NONE, true, false, true>));
      CZC_ATTR((gemm256q_kernel<ACT_QUICK_GELU, false, false, true>));
      CZC_ATTR((gemm256q_kernel<ACT_QUICK_GELU, true, false, true>));
#undef CZC_ATTR
    }
    const int tiles_m = cdiv(g.M, TM), tiles_n = cdiv(g.N, TN);
    const int nt = (g_gemm_krot & 32) ? tiles_m : tiles_m * tiles_n;  // work units: tiles, or M tiles for the M-major walk
    dim3 grid(nt < n_cu ? nt : n_cu), block(768);
    const bool f32 = g.out_f32 != nullptr || g.resid != nullptr;
    if (g.row_stats && !(g_use_gemm256 >= 3 && f32 && g.act == ACT_NONE && g.N % 64 == 0)) {
      snprintf(g_err, sizeof(g_err), "gemm256: row_stats needs the fp32-output ring kernel, no activation, N %% 64 == 0");
      return 1;
    }
    if (g_use_gemm256 == 4 && !g.row_stats && g_w_dbg && f32 && g.act == ACT_NONE && !g.f16) {
      dim3 gq(tiles_m * tiles_n < n_cu ? tiles_m * tiles_n : n_cu), bw(512);
#define CZC_GOWD(D_)                                                                                                     \
  do {                                                                                                                   \
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256w_kernel<ACT_NONE, true, false, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, shp)); \
    hipLaunchKernelGGL((gemm256w_kernel<ACT_NONE, true, false, D_>), gq, bw, shp, st, g, tiles_m, tiles_n);             \
  } while (0)
      if (g_w_dbg == 1) CZC_GOWD(1); else if (g_w_dbg == 2) CZC_GOWD(2); else if (g_w_dbg == 3) CZC_GOWD(3);
      else if (g_w_dbg == 8) CZC_GOWD(8); else CZC_GOWD(9);
#undef CZC_GOWD
      CZC_HIP_CHECK(hipGetLastError());
      return 0;
    }
    if (g_use_gemm256 == 4 && !g.row_stats) {
      dim3 gq(tiles_m * tiles_n < n_cu ? tiles_m * tiles_n : n_cu), bw(512);
#define CZC_GOW(A_, F_, H_)                                                                                              \
  do {                                                                                                                   \
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256w_kernel<A_, F_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, shp)); \
    hipLaunchKernelGGL((gemm256w_kernel<A_, F_, H_>), gq, bw, shp, st, g, tiles_m, tiles_n);                            \
  } while (0)
      if (g.f16) {
        if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOW(ACT_QUICK_GELU, true, true); else CZC_GOW(ACT_QUICK_GELU, false, true); }
        else { if (f32) CZC_GOW(ACT_NONE, true, true); else CZC_GOW(ACT_NONE, false, true); }
      } else {
        if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOW(ACT_QUICK_GELU, true, false); else CZC_GOW(ACT_QUICK_GELU, false, false); }
        else { if (f32) CZC_GOW(ACT_NONE, true, false); else CZC_GOW(ACT_NONE, false, false); }
      }
#undef CZC_GOW
      CZC_HIP_CHECK(hipGetLastError());
      return 0;
    }
    if (g_use_gemm256 >= 3) {
      dim3 gq(tiles_m * tiles_n < n_cu ? tiles_m * tiles_n : n_cu);
      if (g.row_stats) {
        hipLaunchKernelGGL((gemm256q_kernel<ACT_NONE, true, true>), gq, block, shp, st, g, tiles_m, tiles_n);
        CZC_HIP_CHECK(hipGetLastError());
        return 0;
      }
#define CZC_GOQ(A_, F_, H_) hipLaunchKernelGGL((gemm256q_kernel<A_, F_, false, H_>), gq, block, shp, st, g, tiles_m, tiles_n)
      if (g.f16) {
        if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOQ(ACT_QUICK_GELU, true, true); else CZC_GOQ(ACT_QUICK_GELU, false, true); }
        else { if (f32) CZC_GOQ(ACT_NONE, true, true); else CZC_GOQ(ACT_NONE, false, true); }
      } else {
        if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOQ(ACT_QUICK_GELU, true, false); else CZC_GOQ(ACT_QUICK_GELU, false, false); }
        else { if (f32) CZC_GOQ(ACT_NONE, true, false); else CZC_GOQ(ACT_NONE, false, false); }
      }
#undef CZC_GOQ
      CZC_HIP_CHECK(hipGetLastError());
      return 0;
    }
#define CZC_GO(A_, F_) hipLaunchKernelGGL((gemm256p_kernel<A_, F_>), grid, block, shp, st, g, tiles_m, tiles_n, g_gemm_krot)
    if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GO(ACT_QUICK_GELU, true); else CZC_GO(ACT_QUICK_GELU, false); }
    else { if (f32) CZC_GO(ACT_NONE, true); else CZC_GO(ACT_NONE, false); }
#undef CZC_GO
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (!attr_set) {
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel<ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      shmem));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel<ACT_QUICK_GELU>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
    attr_set = true;
  }
  const int tiles_m = cdiv(g.M, TM), tiles_n = cdiv(g.N, TN);
  dim3 grid(tiles_m * tiles_n), block(512);
  if (g.act == ACT_QUICK_GELU)
    hipLaunchKernelGGL(gemm256_kernel<ACT_QUICK_GELU>, grid, block, shmem, st, g, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL(gemm256_kernel<ACT_NONE>, grid, block, shmem, st, g, tiles_m, tiles_n);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int g_w_dbg = 0;  // timing ablations of gemm256w (test option w_dbg)
int g_use_gemm256s = 1;  // 1: four-stage ring (gemm256sq), 2: two-stage form (gemm256s), 0: 128x128 kernel

// split-fp16 operands; big-M layers only (BERT at a few thousand rows stays on the 128x128 + split-K path)
bool gemm256s_eligible(const GemmArgs& g) {
  return g_use_gemm256s && g.M >= 16384 && g.N % 8 == 0 && g.K % 32 == 0 && g.ldc % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         (!g.resid || g.ldr % 4 == 0) && (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) && !g.row_stats && !g.ln_stats &&
         !(g.out_act && g.out_f32) && (long)256 * g.lda * 4 < (1L << 31) && (long)256 * g.ldw * 4 < (1L << 31);
}

int launch_gemm256s(const GemmArgs& g, hipStream_t st) {
  static int n_cu = 0;
  const int shp = 2 * STAGE + 8 * 4096;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    CZC_HIP_CHECK(hipGetDevice(&dev));
    CZC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, shp))
    CZC_ATTR((gemm256s_kernel<ACT_NONE, false>));
    CZC_ATTR((gemm256s_kernel<ACT_NONE, true>));
    CZC_ATTR((gemm256s_kernel<ACT_QUICK_GELU, false>));
    CZC_ATTR((gemm256s_kernel<ACT_QUICK_GELU, true>));
    CZC_ATTR((gemm256sq_kernel<ACT_NONE, false>));
    CZC_ATTR((gemm256sq_kernel<ACT_NONE, true>));
    CZC_ATTR((gemm256sq_kernel<ACT_QUICK_GELU, false>));
    CZC_ATTR((gemm256sq_kernel<ACT_QUICK_GELU, true>));
#undef CZC_ATTR
  }
  const int tiles_m = cdiv(g.M, TM), tiles_n = cdiv(g.N, TN);
  dim3 grid(tiles_m * tiles_n < n_cu ? tiles_m * tiles_n : n_cu), block(768);
  const bool f32 = g.out_f32 != nullptr || g.resid != nullptr;
  if (g_use_gemm256s == 1 && g.K % 16 == 0) {
#define CZC_GOSQ(A_, F_) hipLaunchKernelGGL((gemm256sq_kernel<A_, F_>), grid, block, shp, st, g, tiles_m, tiles_n)
    if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOSQ(ACT_QUICK_GELU, true); else CZC_GOSQ(ACT_QUICK_GELU, false); }
    else { if (f32) CZC_GOSQ(ACT_NONE, true); else CZC_GOSQ(ACT_NONE, false); }
#undef CZC_GOSQ
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
#define CZC_GOS(A_, F_) hipLaunchKernelGGL((gemm256s_kernel<A_, F_>), grid, block, shp, st, g, tiles_m, tiles_n)
  if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOS(ACT_QUICK_GELU, true); else CZC_GOS(ACT_QUICK_GELU, false); }
  else { if (f32) CZC_GOS(ACT_NONE, true); else CZC_GOS(ACT_NONE, false); }
#undef CZC_GOS
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

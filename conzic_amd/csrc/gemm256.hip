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
  if (ACT == ACT_QUICK_GELU) return v * __frcp_rn(1.0f + __expf(-1.702f * v));
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

}  // namespace

bool gemm256_eligible(const GemmArgs& g) {
  return g.M >= 2048 && g.N % 4 == 0 && g.K % 64 == 0 && g.ldc % 4 == 0 && (!g.resid || g.ldr % 4 == 0) &&
         (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) && (long)256 * g.lda * 2 < (1L << 31) &&
         (long)256 * g.ldw * 2 < (1L << 31);
}

int launch_gemm256(const GemmArgs& g, hipStream_t st) {
  static bool attr_set = false;
  const int shmem = 2 * STAGE;
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

}  // namespace czc

// 256x256-tile MFMA GEMMs with LDS-DMA staging for the big-batch linear layers of the CLIP towers
// (M = 10^4..10^6 rows of candidate captions / image patches), bf16 or fp16 operands, fp32 accumulate:
//
//   C[M,N] = A[M,K] . W[N,K]^T (+bias)(quick-GELU)(+fp32 residual)
//
// Common structure (MI355X_MICROARCH / cdna_hip_programming guides):
//  * persistent work-groups, one per CU (160 KiB LDS), walking 256x256 output tiles through an XCD-aware map
//    (work-group b runs on XCD b%8: the N tiles of one M tile sit on neighbouring CUs of one XCD, so an activation
//    tile is fetched into one L2 only);
//  * a 4-deep LDS ring of 32-wide K steps (32 KiB per stage: 256 activation rows + 256 weight rows of 64 bytes),
//    filled by `buffer_load_dwordx4 ... lds` (no VGPR round trip, 1 KiB = 16 tile rows per instruction) from inline
//    asm with COUNTED s_waitcnt vmcnt; the LDS image is lane-linear, so the bank-conflict swizzle
//    (16-byte chunk ^ ((row>>2)&3)) sits in the per-lane SOURCE address and again on the ds_read_b128 side;
//  * buffer descriptors rebased per tile with num_records = valid rows * pitch: rows past M / N read as zero from the
//    hardware bounds check, so ragged edges need no clamping and offsets stay 32-bit beyond 4 GiB tensors;
//  * 8 MFMA waves as 2(M) x 4(N), each 128x64 of C = 4x2 v_mfma_f32_32x32x16 tiles (128 accumulator registers);
//    weights are the MFMA A operand, so a lane owns 4 consecutive output columns of one row per register quad, and
//    the accumulators leave through a wave-private 4 KiB LDS patch as full 128-byte lines.
// Three kernels on that base:
//   gemm256q  8 MFMA waves + 4 LOADER waves that only issue the LDS-DMA (activation-typed outputs: vision qkv / fc1)
//   gemm256x  no loader waves: two wave groups one phase apart, one in MFMAs while the other is in memory
//             instructions (fp32-residual outputs: out-proj / fc2 -- 5 % faster than gemm256q there, A/B in DESIGN.md)
//   gemm256sq gemm256q for split-fp16 operands (three fp16 passes per product)
// plus gemm_rowln: 128 x 512 full-row tiles with the following LayerNorm finished in the epilogue.
#include <type_traits>

#include "kernels.h"

namespace czc {

namespace {

constexpr int TM = 256, TN = 256;

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  // x*sigmoid(1.702x) with the raw v_exp_f32 (2^x) and v_rcp_f32: ~1 ulp each, output is bf16 anyway
  if (ACT == ACT_QUICK_GELU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v));
  return v;
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

// EPI (A/B experiments of the fp32 path): bit 0 = request the residual rows of block n+1 before block n is processed
// (two residual register sets), bit 1 = non-temporal residual loads and output stores
template <int ACT, bool OUT_F32, typename HT = bf16_t, int EPI = 0>
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
      auto load_resid = [&](int i, int j, float4 (&r)[4]) {
        const int col = n0 + wn * 64 + j * 32 + rslot * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = m0 + wm * 128 + i * 32 + pass * 8 + rrow;
          r[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.resid && row < g.M && col < g.N) {
            const float* p = g.resid + (long)row * g.ldr + col;
            if (EPI & 2) {
              const f32x4_t t = __builtin_nontemporal_load((const f32x4_t*)p);
              r[pass] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
              r[pass] = *(const float4*)p;
            }
          }
        }
      };
      float4 rnext[4];
      if (EPI & 1) load_resid(0, 0, rnext);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + rslot * 4;
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.bias && col < g.N) b4 = *(const float4*)(g.bias + col);
          // residual rows first: four independent 16-byte loads in flight across the LDS round trip
          float4 r4[4];
          if (EPI & 1) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) r4[pass] = rnext[pass];
            if (i * 2 + j + 1 < 8) load_resid((i * 2 + j + 1) >> 1, (i * 2 + j + 1) & 1, rnext);
          } else {
            load_resid(i, j, r4);
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
              if (g.out_f32) {
                if (EPI & 2) __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, (f32x4_t*)(g.out_f32 + (long)row * g.ldc + col));
                else *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v;
              }
            }
            pk[pass] = make_uint2(Half<HT>::pack2(v.x, v.y), Half<HT>::pack2(v.z, v.w));
          }
          if (oa) {
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
      }
    }
}


// fp32-residual epilogue with every VMEM instruction in inline asm and COUNTED waits.  tile_epilogue's compiler-scheduled
// form waits `vmcnt(0)` for each block's residual rows (the loads sit behind bounds branches, hipcc cannot count across
// them), and vmcnt retires loads and stores in order: every block therefore also waits until the PREVIOUS block's stores
// have reached memory -- eight load-latency + store-latency round trips per tile and wave, ~20 us of a 95 us fc2 tile.
// Here the residual rows of block n+1 are requested before block n's stores are issued, so the wait for block n
// (vmcnt = stores of n-1 + loads of n+1 still allowed in flight) never includes a store; rows / columns past the edge
// are handled by the buffer descriptors (reads return 0, writes are dropped), so there are no branches.
// Same arithmetic, same order (acc + bias, + residual) as tile_epilogue: bit-identical results.  NJ = 32-column blocks per wave.
// ABL (timing ablations, EXPERIMENTS build): 1 no residual loads, 2 no output stores
template <int NJ, int ABL = 0>
__device__ __forceinline__ void tile_epilogue_f32_asm(const GemmArgs& g, f32x16_t (&acc)[4][NJ], unsigned char* patch, int m0, int n0,
                                                      int wm, int wcol0, int lane) {
  const int half = lane >> 5, l31 = lane & 31, rrow = lane >> 3, rslot = lane & 7;
  const int rows = min(TM, g.M - m0);
  auto desc = [&](const void* base, int ld, int n_rows) {
    const unsigned long long pa = (unsigned long long)base + (unsigned long long)m0 * ld * 4;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((unsigned)(n_rows * ld * 4));
    r.w = 0x00020000u;
    return r;
  };
  const u32x4_t rsR = desc(g.resid, g.ldr, rows), rsO = desc(g.out_f32, g.ldc, rows);
  u32x4_t rsB;
  {
    const unsigned long long pb = (unsigned long long)g.bias;
    rsB.x = __builtin_amdgcn_readfirstlane((unsigned)pb);
    rsB.y = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32) & 0xffffu);
    rsB.z = __builtin_amdgcn_readfirstlane((unsigned)(g.bias ? g.N * 4 : 0));  // no bias: every read returns 0
    rsB.w = 0x00020000u;
  }
  asm volatile("s_nop 4" ::: "memory");  // descriptors fresh from v_readfirstlane -> buffer_* inside asm strings
  constexpr int NB = 4 * NJ;
  auto colof = [&](int blk) { return n0 + wcol0 + (blk % NJ) * 32 + rslot * 4; };
  auto off = [&](int blk, int pass, int ld) -> unsigned {
    const int col = colof(blk);
    const int row = wm * 128 + (blk / NJ) * 32 + pass * 8 + rrow;
    return col < g.N ? (unsigned)((row * ld + col) * 4) : 0x7ffffff0u;
  };
  f32x4_t bias4[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = colof(j);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bias4[j]) : "v"(col < g.N ? (unsigned)(col * 4) : 0x7ffffff0u), "s"(rsB) : "memory");
  }
  f32x4_t rr[2][4];
  auto load_resid = [&](int blk, f32x4_t (&r)[4]) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      if (ABL & 1) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(r[pass][0]), "=v"(r[pass][1]), "=v"(r[pass][2]), "=v"(r[pass][3]));
      else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[pass]) : "v"(off(blk, pass, g.ldr)), "s"(rsR) : "memory");
    }
  };
  load_resid(0, rr[0]);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int i = blk / NJ, j = blk % NJ;
    if (blk + 1 < NB) load_resid(blk + 1, rr[(blk + 1) & 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = (2 * q + half) ^ (l31 & 7);
      *(float4*)(patch + l31 * 128 + slot * 16) =
          make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
    f32x4_t(&r)[4] = rr[blk & 1];
    // residual rows of this block (and, first time round, the bias) have landed: younger = stores of block blk-1, loads of blk+1
    if (ABL) {  // the counts below assume four loads and four stores per block
      if (blk == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
      else if (ABL == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
      else if (ABL == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
    } else if (blk == 0 || blk + 1 == NB)
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rw = pass * 8 + rrow;
      const float4 t = *(const float4*)(patch + rw * 128 + ((rslot ^ (rw & 7)) << 4));
      f32x4_t v;
      v[0] = t.x + bias4[j][0]; v[1] = t.y + bias4[j][1]; v[2] = t.z + bias4[j][2]; v[3] = t.w + bias4[j][3];
      v[0] += r[pass][0]; v[1] += r[pass][1]; v[2] += r[pass][2]; v[3] += r[pass][3];
      // s_nop 1: a > 64-bit asm store must not be followed at once by a write of its data registers (hipcc pads its own
      // stores, not an asm string: without it some lanes stored the next instruction's operands)
      if (ABL & 2) asm volatile("" : : "v"(v), "v"(off(blk, pass, g.ldc)));
      else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(off(blk, pass, g.ldc)), "s"(rsO) : "memory");
    }
  }
}

// The same epilogue for a 2-BYTE residual stream (GemmArgs::x16, round 5: the bf16 engine's CLIP-text tower keeps x as IEEE
// fp16 rows -- 11 significand bits, fp32 accumulate / bias / residual add before the one rounding): resid and out_f32 point
// to fp16 rows (ldr / ldc in elements).  A lane owns 8 consecutive columns (16 bytes) of a row, four lanes cover a 32-column
// block row, 16 rows per pass and two passes per 32 x 32 block: every residual load and output store is again a full
// 16 bytes per lane and 64 bytes per row, at half the bytes of the fp32 form (2 + 2 instead of 4 + 4 VMEM instructions per
// block).  Same order of operations per element (patch value + bias, + residual), then one round-to-nearest-even to fp16
// from the pinned fp32 value -- what the tiled kernel's epilogue (gemm.hip) does for the same layer at small row counts.
template <int NJ>
__device__ __forceinline__ void tile_epilogue_x16_asm(const GemmArgs& g, f32x16_t (&acc)[4][NJ], unsigned char* patch, int m0, int n0,
                                                      int wm, int wcol0, int lane) {
  const int half = lane >> 5, l31 = lane & 31, rrow = lane >> 2, rs = lane & 3;
  const int rows = min(TM, g.M - m0);
  auto desc = [&](const void* base, int ld, int n_rows) {
    const unsigned long long pa = (unsigned long long)base + (unsigned long long)m0 * ld * 2;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((unsigned)(n_rows * ld * 2));
    r.w = 0x00020000u;
    return r;
  };
  u32x4_t rsR = desc(g.resid, g.ldr, rows), rsO = desc(g.out_f32, g.ldc, rows);
  u32x4_t rsB;
  {
    const unsigned long long pb = (unsigned long long)g.bias;
    rsB.x = __builtin_amdgcn_readfirstlane((unsigned)pb);
    rsB.y = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32) & 0xffffu);
    rsB.z = __builtin_amdgcn_readfirstlane((unsigned)(g.bias ? g.N * 4 : 0));  // no bias: every read returns 0
    rsB.w = 0x00020000u;
  }
  u32x4_t rsP;  // LayerNorm partials (GemmArgs::row_part, [column block][row] float2); empty without a target
  {
    // (no target: a valid base and four records; every lane's offset is then past the end, the drop that serves the edges)
    const unsigned long long pp = (unsigned long long)(g.row_part ? (const void*)g.row_part : (const void*)g.W);
    rsP.x = __builtin_amdgcn_readfirstlane((unsigned)pp);
    rsP.y = __builtin_amdgcn_readfirstlane((unsigned)(pp >> 32) & 0xffffu);
    rsP.z = __builtin_amdgcn_readfirstlane((unsigned)(g.row_part ? (long)(g.N / 32) * g.part_ld * 8 : 4));
    rsP.w = 0x00020000u;
  }
  // descriptors fresh from v_readfirstlane -> buffer_* inside asm strings: the wait states hipcc cannot see.  The descriptors
  // are operands of the nop so that none of their v_readfirstlane can sink below it (the partials' descriptor, first used
  // late in the first block, did: that block's first store went out with a stale descriptor and was lost)
  asm volatile("s_nop 4" : "+s"(rsR), "+s"(rsO), "+s"(rsB), "+s"(rsP) : : "memory");
  constexpr int NB = 4 * NJ;
  auto colof = [&](int blk) { return n0 + wcol0 + (blk % NJ) * 32 + rs * 8; };
  auto off = [&](int blk, int pass, int ld) -> unsigned {
    const int col = colof(blk);
    const int row = wm * 128 + (blk / NJ) * 32 + pass * 16 + rrow;
    return col < g.N ? (unsigned)((row * ld + col) * 2) : 0x7ffffff0u;
  };
  f32x4_t bias8[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = colof(j);
    const unsigned bo = col < g.N ? (unsigned)(col * 4) : 0x7ffffff0u;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bias8[j][0]) : "v"(bo), "s"(rsB) : "memory");
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(bias8[j][1]) : "v"(bo), "s"(rsB) : "memory");
  }
  u32x4_t rr[2][2];
  auto load_resid = [&](int blk, u32x4_t (&r)[2]) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[pass]) : "v"(off(blk, pass, g.ldr)), "s"(rsR) : "memory");
  };
  load_resid(0, rr[0]);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int i = blk / NJ, j = blk % NJ;
    if (blk + 1 < NB) load_resid(blk + 1, rr[(blk + 1) & 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = (2 * q + half) ^ (l31 & 7);
      *(float4*)(patch + l31 * 128 + slot * 16) =
          make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
    u32x4_t(&r)[2] = rr[blk & 1];
    // this block's residual rows (and, first time round, the bias) have landed: younger = the three stores (x, x, partials) of
    // block blk-1 and the two loads of blk+1
    if (blk == 0)
      asm volatile("s_waitcnt vmcnt(2)" : "+v"(r[0]), "+v"(r[1]), "+v"(bias8[j][0]), "+v"(bias8[j][1]) : : "memory");
    else if (blk + 1 == NB)
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(r[0]), "+v"(r[1]), "+v"(bias8[j][0]), "+v"(bias8[j][1]) : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(5)" : "+v"(r[0]), "+v"(r[1]), "+v"(bias8[j][0]), "+v"(bias8[j][1]) : : "memory");
    uint2 sq_keep = make_uint2(0u, 0u);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int rw = pass * 16 + rrow;
      const float4 t0 = *(const float4*)(patch + rw * 128 + (((2 * rs) ^ (rw & 7)) << 4));
      const float4 t1 = *(const float4*)(patch + rw * 128 + (((2 * rs + 1) ^ (rw & 7)) << 4));
      float v[8] = {t0.x + bias8[j][0][0], t0.y + bias8[j][0][1], t0.z + bias8[j][0][2], t0.w + bias8[j][0][3],
                    t1.x + bias8[j][1][0], t1.y + bias8[j][1][1], t1.z + bias8[j][1][2], t1.w + bias8[j][1][3]};
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned pr = r[pass][e];
        const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr & 0xffffu));
        const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr >> 16));
        o[e] = pack2_f16(v[2 * e] + lo, v[2 * e + 1] + hi);
      }
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(o), "v"(off(blk, pass, g.ldc)), "s"(rsO) : "memory");
      // LayerNorm partials of the stored values over this 32-column block, in the association order of the weight-stationary
      // residual kernel (lane (row, half) there sums its quads 0..3 in order, then half 0 + half 1): lane rs holds quad rs of
      // both halves, so the four lanes of a row are summed in lane order by quad-broadcast DPP moves
      const float w0 = f16lo(o[0]), w1 = f16hi(o[0]), w2 = f16lo(o[1]), w3 = f16hi(o[1]);
      const float w4 = f16lo(o[2]), w5 = f16hi(o[2]), w6 = f16lo(o[3]), w7 = f16hi(o[3]);
      float part[4] = {ln_sum4(w0, w1, w2, w3), ln_sum4(w4, w5, w6, w7), ln_sq4(w0, w1, w2, w3), ln_sq4(w4, w5, w6, w7)};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int x = __builtin_bit_cast(int, part[t]);
        const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true));
        const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, true));
        const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xf, 0xf, true));
        const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xf, 0xf, true));
        part[t] = __fadd_rn(__fadd_rn(__fadd_rn(q0, q1), q2), q3);
      }
      // one store per block for both passes: lane rs = 0 keeps pass 0's pair (row rrow), lane rs = 1 pass 1's (row 16 + rrow)
      const uint2 sq = make_uint2(__float_as_uint(__fadd_rn(part[0], part[1])), __float_as_uint(__fadd_rn(part[2], part[3])));
      if (pass == 0) sq_keep = sq;
      else {
        const uint2 out = rs == 1 ? sq : sq_keep;
        const int prow = m0 + wm * 128 + (blk / NJ) * 32 + (rs == 1 ? 16 : 0) + rrow;
        const long pcol = (n0 + wcol0 + (blk % NJ) * 32) / 32;
        const unsigned po = (g.row_part && rs < 2 && prow < g.M && n0 + wcol0 + (blk % NJ) * 32 < g.N) ? (unsigned)((pcol * g.part_ld + prow) * 8) : 0x7ffffff0u;
        asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(out), "v"(po), "s"(rsP) : "memory");
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

// Visit order of a persistent work-group: virtual tiles wg, wg+grid, ... through the XCD-aware map, i.e. the N tiles
// of an M tile run CONCURRENTLY on neighbouring CUs of one XCD.  (One work-group walking all N tiles of an M tile back
// to back -- A from HBM once, then L2 -- measured 9 % slower on the qkv shape, round 2.)
__device__ __forceinline__ int tile_count(int tiles_m, int tiles_n) {
  const int grid = gridDim.x, wg = blockIdx.x, nt = tiles_m * tiles_n;
  return wg < nt ? (nt - 1 - wg) / grid + 1 : 0;
}
__device__ __forceinline__ void tile_at(int i, int tiles_m, int tiles_n, int& tm, int& tn) {
  tile_of(blockIdx.x + i * gridDim.x, tiles_m, tiles_n, tm, tn);
}

// Start stagger of a persistent grid whose tiles end in an HBM burst (fp32 residual in, fp32 row out): left alone all
// work-groups run in lock-step -- every CU in its main loop (HBM nearly idle), then every CU in its epilogue (HBM the
// bound) -- and a work-group's epilogue takes as long as the whole chip's.  Work-group b starts `phase` x tau microseconds
// late (phase 0..7, the same for the work-groups that share an A tile), and the work-groups that have one tile less than
// the others (b >= nt % grid: they would idle at the end anyway) another `bonus` microseconds: the epilogues of different
// work-groups then fall on different moments and overlap other work-groups' main loops.  stagger = tau | bonus << 8.
// Only the start time changes: results are bit-identical.  (profiles/r03_rowln_stagger_ab.txt)
__device__ __forceinline__ void start_stagger(int stagger, int nt, int share) {
  if (!stagger) return;
  const int b = (int)blockIdx.x, tau = stagger & 255, bonus = stagger >> 8;
  const int rem = nt % (int)gridDim.x;
  unsigned long long dt = (unsigned long long)(((b >> 3) / share) & 7) * tau;
  if (rem && b >= rem) dt += bonus;
  dt *= 100ull;  // s_memrealtime counts at 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
}

// ================================================================================================
// gemm256q: 12 waves per work-group: waves 0-7 are MFMA waves (never issue a global load), waves 8-11 are LOADER
// waves that only issue the LDS-DMA of later K steps (8 x 1 KiB each per step).  The K-step sequence is continuous
// across tiles, so while the MFMA waves run a tile's epilogue the loaders already fetch the next tile's first steps.
// The loaders run up to three steps ahead and wait with a COUNTED vmcnt, so a step costs max(DMA issue, DMA latency,
// MFMA) instead of issue + latency; hand-off per K step = one s_barrier shared by all 12 waves.
//   LDS rows are 64 bytes (32 bf16): 16-byte chunk c of row r sits at chunk c ^ ((r>>2)&3)
//   (16 distinct rows of a ds_read_b128 lane group -> 16 distinct slots of the 256-byte bank row);
//   one DMA instruction lands 16 rows.
// ================================================================================================
constexpr int QROWB = 64;                      // bytes per tile row per K step (32 bf16)
constexpr int QA_BYTES = TM * QROWB;           // 16 KiB
constexpr int QSTAGE = QA_BYTES + TN * QROWB;  // 32 KiB
constexpr int QS = 4;                          // ring depth

__device__ __forceinline__ int swzq(int row, int chunk) { return row * QROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int ACT, bool OUT_F32, bool F16 = false>
__global__ __launch_bounds__(768) void gemm256q_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;  // operand element type (common.h Half<>)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;  // 32-wide K steps
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  const int my_tiles = tile_count(tiles_m, tiles_n);
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
        tile_at(ti, tiles_m, tiles_n, tm, tn);
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
    tile_at(ti, tiles_m, tiles_n, tm, tn);
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
    tile_epilogue<ACT, OUT_F32, HT>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, wm, wn, lane);
  }
}



// ================================================================================================
// gemm256x: the 4-deep ring of 32 KiB stages driven by two wave GROUPS that run one phase apart ("ping-pong").
// Counters and ablations of the ring kernels above say the same thing twice: with every wave of a SIMD in
// the same phase, the fragment reads + DMA issue of a stage and its MFMAs ADD instead of overlapping (0.70 + 0.68 ->
// 0.95 ms on the fc2 shape).  Here the two waves of a SIMD (w and w+4) are never in the same phase:
//
//     group 0 (waves 0-3, tile rows 0-127)  :  L(s) | M(s) | L(s+1) | M(s+1) | ...
//     group 1 (waves 4-7, tile rows 128-255):       | L(s) | M(s)   | L(s+1) | ...        ( | = s_barrier of all 8 waves )
//
//   L(s): issue this wave's four 1 KiB LDS-DMA pieces of a later stage, then read the stage's 12 MFMA fragments
//         (48 VGPRs: both k16 halves of 4 activation + 2 weight fragments) -- everything that blocks on the CU's
//         vector-memory / LDS paths;
//   M(s): 16 x v_mfma_f32_32x32x16 on those registers, nothing else (s_setprio 1 .. 0 around them).
// So in every barrier interval each SIMD has one wave that owns the matrix pipe and one that is in memory instructions.
// Ring protocol: stage t lives in slot t % 4.  Group 0 issues stage s+2 in L(s), group 1 stage s+3 in its L(s) (one
// interval later); both target slots whose last reader passed an lgkmcnt(0) at least one barrier earlier.  Stage s+1 is
// published by the barrier that ends group 0's M(s) = group 1's L(s): every wave waits for ITS pieces of s+1 with a
// counted vmcnt just before that barrier (4 resp. 8 younger pieces may stay in flight).  At a tile end group 0 waits
// one interval for group 1's last M, then both groups run the epilogue together; the ring runs on across tiles.
// ================================================================================================
// DBG (timing ablations only, results are garbage): 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no epilogue;
// 16 / 32: epilogue A/B forms (tile_epilogue EPI bits 0 / 1; results stay correct); 64 / 128: asm epilogue without its
// residual loads / without its stores
// Control flow (round 5, second half): the K loop is branch-free in its steady state.  The first form carried its conditions
// into every K step -- the `step + ahead < total` guards, the group-dependent choice of the counted waits in front of both
// barriers, run-time switches of the A/B arms, an integer division for the stage's tile -- which hipcc laid out as ~10 scalar
// branches per 16 MFMAs, several of them taken and two of them directly in front of a barrier the other seven waves wait at
// (the weight-stationary kernel showed what that costs: ~1 % per taken branch and 32 MFMAs, gemm_wreg.hip).  Now the two
// groups run their own copies of the loop (GRP is a compile-time constant in each), every tile but the last three K steps
// of a work-group runs the STEADY body (all guards true: one DMA issue, twelve fragment reads, one counted wait, two
// barriers, sixteen MFMAs, one loop branch) and the stage counter of the issue side advances incrementally.
template <int ACT, bool OUT_F32, bool F16 = false, int DBG = 0>
__global__ __launch_bounds__(512) void gemm256x_kernel(GemmArgs g, int tiles_m, int tiles_n, int var) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 5;  // 32-wide K steps
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  start_stagger(var >> 8, tiles_m * tiles_n, tiles_n);
  const int my_tiles = tile_count(tiles_m, tiles_n);
  const int total = my_tiles * nk;
#ifdef CZC_EXPERIMENTS
  const bool prio = !(var & 1), dma_late = var & 2;  // A/B arms (w_dbg bits 0 / 1)
#else
  constexpr bool prio = true, dma_late = false;
#endif

  // ---- DMA side: pieces ii = 0, 1 land tile rows wave*32 + ii*16 + (lane>>2); physical chunk lane&3 ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  const int rbase = wave * 32 + (lane >> 2);
  const int cq = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const int a0 = rbase * lda_b + cq, w0 = rbase * ldw_b + cq;
  const int a16 = 16 * lda_b, w16 = 16 * ldw_b;
  u32x4_t rsA, rsW;
  rsA.x = rsA.y = rsA.z = 0; rsA.w = 0x00020000u;
  rsW = rsA;
  int is = 0, is_ti = 0, is_kt = 0;  // next stage to issue: number, tile, K step (stages are issued in order)
  auto issue_rebase = [&]() {
    int tm, tn;
    tile_at(is_ti, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const unsigned long long pa = (unsigned long long)g.A + (unsigned long long)m0 * lda_b;
    const unsigned long long pw = (unsigned long long)g.W + (unsigned long long)n0 * ldw_b;
    rsA.x = (unsigned)pa; rsA.y = (unsigned)(pa >> 32) & 0xffffu; rsA.z = (unsigned)(min(TM, g.M - m0) * lda_b);
    rsW.x = (unsigned)pw; rsW.y = (unsigned)(pw >> 32) & 0xffffu; rsW.z = (unsigned)(min(TN, g.N - n0) * ldw_b);
  };
  auto issue_next = [&]() {  // stage `is` -> ring slot is % QS
    const unsigned dstA = lds0 + (is & (QS - 1)) * QSTAGE + wave * (32 * QROWB);
    const unsigned dstW = dstA + QA_BYTES;
    const unsigned so = is_kt * QROWB;
    if (!(DBG & 2)) {
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %9 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %9 offen lds\n\t"
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %9 offen lds\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %9 offen lds\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "s"(dstA), "s"(dstW), "v"(a0), "v"(a0 + a16), "v"(w0), "v"(w0 + w16), "s"(rsA), "s"(rsW), "s"(so)
          : "memory", "scc");
    }
    ++is;
    if (++is_kt == nk) {  // the next stage opens a tile (once per tile: the only branch of the issue side)
      is_kt = 0;
      ++is_ti;
      if (is < total) issue_rebase();
    }
  };

  auto run = [&](auto grp_c) {
    constexpr int GRP = decltype(grp_c)::value;
    constexpr int AHEAD = 2 + GRP;  // group 0 issues stage s+2 in L(s), group 1 stage s+3 in its L(s)
    if (total > 0) issue_rebase();
    for (int s = 0; s < AHEAD && s < total; ++s) issue_next();
    {
      const int later = (total < AHEAD ? total : AHEAD) - 1;  // stages issued behind stage 0
      if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // publishes stage 0
    asm volatile("" ::: "memory");

    // ---- MFMA side ----
    const int wn = wave & 3;
    const int half = lane >> 5;
    const int arow = GRP * 128 + (lane & 31);
    const int brow = wn * 64 + (lane & 31);
    int step = 0;
    const int steady_end = total - 3;  // steps below it: every guard of the generic body is true
    for (int ti = 0; ti < my_tiles; ++ti) {
      int tm, tn;
      tile_at(ti, tiles_m, tiles_n, tm, tn);
      const int m0 = tm * TM, n0 = tn * TN;
      f32x16_t acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      if (GRP) {  // group 1 runs one phase behind: this barrier pairs with the one that ends group 0's L of the tile's first stage
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      auto kstep = [&](auto steady_c) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        // ------------------------------- L(step) -------------------------------
        const unsigned char* sA = smem + (step & (QS - 1)) * QSTAGE;
        const unsigned char* sB = sA + QA_BYTES;
        if (!dma_late && (STEADY || step + AHEAD < total)) issue_next();
        uint4 fa[2][4], fb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DBG & 4) fb[ks][j] = make_uint4(lane, step, ks, j);
            else fb[ks][j] = *(const uint4*)(sB + swzq(brow + 32 * j, 2 * ks + half));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (DBG & 4) fa[ks][i] = make_uint4(lane, step, ks, i);
            else fa[ks][i] = *(const uint4*)(sA + swzq(arow + 32 * i, 2 * ks + half));
          }
        }
        if (dma_late && (STEADY || step + AHEAD < total)) issue_next();
        if (GRP) {  // stage step+1 is published by the barrier below: this wave's pieces of it must have landed
          if (STEADY || step + 3 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if (step + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------- M(step) -------------------------------
        if (prio) __builtin_amdgcn_s_setprio(1);
        if (DBG & 1) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {  // keep the fragments (and their LDS reads) alive without the matrix work
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, fb[ks][j])));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, fa[ks][i])));
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(fb[ks][j], fa[ks][i], acc[i][j]);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!GRP) {
          if (STEADY || step + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      int kt = 0;
      for (; kt < nk && step < steady_end; ++kt, ++step) kstep(std::true_type());
      for (; kt < nk; ++kt, ++step) kstep(std::false_type());
      if (!GRP) {  // group 1's last M of this tile
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (DBG & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
      } else {
        if constexpr (DBG == 256) {  // 2-byte residual stream (GemmArgs::x16; launch_gemm256 only instantiates it for ACT_NONE, OUT_F32)
          tile_epilogue_x16_asm<2>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, GRP, wn * 64, lane);
        } else if constexpr (OUT_F32 && ACT == ACT_NONE) {
          if (!(var & 8) && g.resid && g.out_f32 && !g.out_act)
            tile_epilogue_f32_asm<2, (DBG >> 6) & 3>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, GRP, wn * 64, lane);
          else
            tile_epilogue<ACT, OUT_F32, HT, (DBG >> 4) & 3>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, GRP, wn, lane);
        } else {
          tile_epilogue<ACT, OUT_F32, HT, (DBG >> 4) & 3>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, GRP, wn, lane);
        }
      }
    }
  };
  if (wave >> 2) run(std::integral_constant<int, 1>());
  else run(std::integral_constant<int, 0>());
}

#ifdef CZC_EXPERIMENTS  // an A/B arm that lost (profiles/r04_gemm256r_ablations.txt): `make EXPERIMENTS=1` builds only, not in the product library
// ================================================================================================
// gemm256r (round 4, A/B arm: test option gemm256 = 9): the vendor library's structure for the fp32-residual layers --
// FOUR waves, each 128 x 128 of C (16 v_mfma_f32_32x32x16 accumulators = 256 registers, one wave per SIMD), operands
// REGISTER-staged: every lane requests its 8 sixteen-byte pieces of a k32 stage (buffer loads, descriptor bounds) three
// stages before they are needed, holds them in VGPRs for two stages (two alternating sets, 64 registers) and writes
// them into the 4-slot LDS ring with ds_write_b128 two stages before they are read; one s_barrier per stage; the
// fragments of the next half stage are read while the current half's 16 MFMAs run.  Per half stage a wave issues 16
// MFMAs and 16 other instructions (8 ds_read_b128, 4 buffer loads, 4 ds_write_b128), pinned one to one behind the
// MFMAs with sched_group_barrier.  LDS reads per FLOP are a third lower than with eight 128 x 64 waves, and there is
// no M0 / s_nop traffic of the LDS-DMA form.  Same k order per accumulator and the same epilogue code as gemm256x:
// bit-identical results.  Ring slot protocol (QS = 4): stage s is read from slot s & 3 during stage s (its first
// fragments already during stage s - 1), written during stage s - 2; the barrier at the end of every stage orders
// both hand-overs.
// ================================================================================================
// DBG (timing ablations, results are garbage): 1 no MFMA, 2 no global loads, 4 no ring writes, 8 no epilogue, 16 no fragment reads
template <bool F16, int DBG = 0>
__global__ __launch_bounds__(256) void gemm256r_kernel(GemmArgs g, int tiles_m, int tiles_n, int var) {
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = g.K >> 5;
  const int lda_b = g.lda * 2, ldw_b = g.ldw * 2;
  start_stagger(var >> 8, tiles_m * tiles_n, tiles_n);
  const int my_tiles = tile_count(tiles_m, tiles_n);
  const int total = my_tiles * nk;

  // staging map: lane moves chunk (lane & 3) of rows wave * 64 + 16 * p + (lane >> 2), p = 0..3, of both operands
  const int srow = wave * 64 + (lane >> 2), sc = lane & 3;
  const int voA = srow * lda_b + sc * 16, voW = srow * ldw_b + sc * 16;
  const int soff = swzq(srow, sc);  // rows + 16 p: same swizzle phase, + 1024 p bytes
  const auto rsNone = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, (short)0, 0, 0x00020000);
  auto rsA = rsNone, rsW = rsNone;
  int cur_ti = -1;
  u32x4_t st0[8], st1[8];
  auto load_stage = [&](int s, u32x4_t (&r)[8]) {
    const int ti = s / nk, kt = s - ti * nk;
    if (s < total && ti != cur_ti) {
      cur_ti = ti;
      int tm, tn;
      tile_at(ti, tiles_m, tiles_n, tm, tn);
      const int m0 = tm * TM, n0 = tn * TN;
      rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.A + (long)m0 * lda_b), (short)0, min(TM, g.M - m0) * lda_b, 0x00020000);
      rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.W + (long)n0 * ldw_b), (short)0, min(TN, g.N - n0) * ldw_b, 0x00020000);
    }
    const auto ra = s < total ? rsA : rsNone;
    const auto rw = s < total ? rsW : rsNone;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      r[p] = __builtin_amdgcn_raw_buffer_load_b128(ra, voA + 16 * p * lda_b, kt * QROWB, 0);
      r[4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rw, voW + 16 * p * ldw_b, kt * QROWB, 0);
    }
  };
  auto store_stage = [&](int s, const u32x4_t (&r)[8]) {
    unsigned char* dst = smem + (s & (QS - 1)) * QSTAGE + soff;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(u32x4_t*)(dst + 1024 * p) = r[p];
      *(u32x4_t*)(dst + QA_BYTES + 1024 * p) = r[4 + p];
    }
  };
  const int half = lane >> 5;
  const int arow = wm * 128 + (lane & 31);
  const int brow = wn * 128 + (lane & 31);
  uint4 fa[2][4], fb[2][4];
  auto read_frags = [&](int s, int ks, uint4 (&a)[4], uint4 (&b)[4]) {
    const unsigned char* sA = smem + (s & (QS - 1)) * QSTAGE;
    const unsigned char* sB = sA + QA_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (DBG & 16) { b[i] = make_uint4(lane, s, ks, i); a[i] = make_uint4(lane, s, i, ks); continue; }
      b[i] = *(const uint4*)(sB + swzq(brow + 32 * i, 2 * ks + half));
      a[i] = *(const uint4*)(sA + swzq(arow + 32 * i, 2 * ks + half));
    }
  };
  // prologue: stages 0, 1 into the ring, 2 and 3 in flight in the two register sets
  load_stage(0, st0);
  load_stage(1, st1);
  store_stage(0, st0);
  load_stage(2, st0);
  store_stage(1, st1);
  load_stage(3, st1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(0, 0, fa[0], fb[0]);
  // steady state at stage s (even: set 0, odd: set 1): the set holds stage s + 2, is written to the ring and refilled with s + 4
  int step = 0;
  int ti4 = 3 / nk, kt4 = 3 - ti4 * nk;  // (tile, k step) of the last stage requested so far; rsA / rsW are that tile's
  if (ti4 >= my_tiles) { rsA = rsNone; rsW = rsNone; }
  for (int ti = 0; ti < my_tiles; ++ti) {
    int tm, tn;
    tile_at(ti, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define CZC_R_HALF(FA_, FB_)                                                                       \
  if (DBG & 1) {                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, FA_[i]))); asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, FB_[i]))); } \
  } else {                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = Half<HT>::mfma(FB_[j], FA_[i], acc[i][j]); \
  }
#define CZC_R_PIN()                                                                                 \
  _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
  }                                                                                                 \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
  }                                                                                                 \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                              \
  }
    // one stage: WR = the set that holds stage s + 2 (written to the ring, then refilled with stage s + 4).  The scalar
    // bookkeeping of the refill (descriptors of the tile stage s + 4 belongs to) sits in front, so that everything from
    // the first fragment read to the barrier is ONE basic block the interleave can be pinned in.
#define CZC_R_STAGE(WR_)                                                                            \
  {                                                                                                 \
    const int s = step;                                                                             \
    if (++kt4 == nk) {                                                                              \
      kt4 = 0;                                                                                      \
      ++ti4;                                                                                        \
      if (ti4 < my_tiles) {                                                                         \
        int tm4, tn4;                                                                               \
        tile_at(ti4, tiles_m, tiles_n, tm4, tn4);                                                   \
        const int mm = tm4 * TM, nn = tn4 * TN;                                                     \
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.A + (long)mm * lda_b), (short)0, min(TM, g.M - mm) * lda_b, 0x00020000); \
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.W + (long)nn * ldw_b), (short)0, min(TN, g.N - nn) * ldw_b, 0x00020000); \
      } else {                                                                                      \
        rsA = rsNone;                                                                               \
        rsW = rsNone;                                                                               \
      }                                                                                             \
    }                                                                                               \
    const int so4 = kt4 * QROWB;                                                                    \
    unsigned char* wdst = smem + ((s + 2) & (QS - 1)) * QSTAGE + soff;                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    /* first half: MFMAs of k16 step 0 | fragments of step 1, the A half of the ring write and of the refill */ \
    read_frags(s, 1, fa[1], fb[1]);                                                                 \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                 \
      if (DBG & 4) asm volatile("" ::"v"(WR_[p])); else *(u32x4_t*)(wdst + 1024 * p) = WR_[p];      \
      if (DBG & 2) WR_[p].x += 1; else WR_[p] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA + 16 * p * lda_b, so4, 0); \
    }                                                                                               \
    CZC_R_HALF(fa[0], fb[0])                                                                        \
    CZC_R_PIN()                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    /* second half: MFMAs of step 1 | first fragments of stage s + 1, the W half of the write and of the refill */ \
    read_frags(s + 1, 0, fa[0], fb[0]);                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                 \
      if (DBG & 4) asm volatile("" ::"v"(WR_[4 + p])); else *(u32x4_t*)(wdst + QA_BYTES + 1024 * p) = WR_[4 + p]; \
      if (DBG & 2) WR_[4 + p].x += 1; else WR_[4 + p] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW + 16 * p * ldw_b, so4, 0); \
    }                                                                                               \
    CZC_R_HALF(fa[1], fb[1])                                                                        \
    CZC_R_PIN()                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
    __builtin_amdgcn_s_barrier();                                                                   \
    asm volatile("" ::: "memory");                                                                  \
    ++step;                                                                                         \
  }
    for (int kt = 0; kt < nk; ++kt) {  // nk is even (K % 64 == 0): even stages use set 0, odd ones set 1
      CZC_R_STAGE(st0)
      ++kt;
      CZC_R_STAGE(st1)
    }
    if (DBG & 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      tile_epilogue_f32_asm<4, 0>(g, acc, smem + QS * QSTAGE + wave * 4096, m0, n0, wm, wn * 128, lane);
    }
  }
#undef CZC_R_STAGE
#undef CZC_R_PIN
#undef CZC_R_HALF
}

#endif  // CZC_EXPERIMENTS (gemm256r)

// ================================================================================================
// gemm256sq: gemm256q's 4-deep ring of 32 KiB stages for the SPLIT-fp16 precision.  A 64-byte tile row holds 16
// elements ([8 hi | 8 lo] x 2 groups) = one k16 MFMA step, three fp16 passes per product: 24 MFMAs per stage and
// wave on 12 ds_read_b128, loaders up to three stages ahead with counted vmcnt.
// ================================================================================================
template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(768) void gemm256sq_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K >> 4;  // one k16 MFMA step per stage: a 64-byte row = two split_t groups = 16 elements
  const int lda_b = g.lda * 4, ldw_b = g.ldw * 4;
  const int my_tiles = tile_count(tiles_m, tiles_n);
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
        tile_at(ti, tiles_m, tiles_n, tm, tn);
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
    tile_at(ti, tiles_m, tiles_n, tm, tn);
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

// DBG (timing ablations, EXPERIMENTS build, results are garbage): 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads, 8 no epilogue,
// 16 y not stored (-1 KB per row), 32 only the first 64 bytes of every x line stored, 64 only every other x line stored (-1 KB per row each)
// AE: the x phase of the epilogue with every VMEM instruction in inline asm and counted waits (below); false = the
// compiler-scheduled form it replaces (test option w_dbg bit 3), which waits vmcnt(0) in front of every 32 x 32 block --
// for the block's residual rows AND the previous block's stores: eight full memory round trips per tile.
template <bool F16, int DBG = 0, bool AE = true>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(GemmArgs g, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int tiles_m, int stagger_us = 0) {  // stagger_us: start_stagger()
  using HT = std::conditional_t<F16, f16_t, bf16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  start_stagger(stagger_us, tiles_m, 1);
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
    if (DBG & 2) return;
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
    if (DBG & 2) return;
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
      if (DBG & 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { b0[j] = make_uint4(lane, step, j, 0); b1[j] = make_uint4(lane, step, j, 1); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0f[i] = make_uint4(lane, step, i, 2); a1f[i] = make_uint4(lane, step, i, 3); }
      } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) b0[j] = *(const uint4*)(sB + swzq(brow + 32 * j, half));
#pragma unroll
      for (int i = 0; i < 4; ++i) a0f[i] = *(const uint4*)(sA + swzq(l31 + 32 * i, half));
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[j] = *(const uint4*)(sB + swzq(brow + 32 * j, 2 + half));
#pragma unroll
      for (int i = 0; i < 4; ++i) a1f[i] = *(const uint4*)(sA + swzq(l31 + 32 * i, 2 + half));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (DBG & 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, b0[j])));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, a0f[i])));
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(b0[j], a0f[i], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (late) {
        if (step + 3 < total) issueA(step + 3);
        if (step + 2 < total) issueW(step + 2, wnext);
      }
      if (DBG & 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, b1[j])));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4_t, a1f[i])));
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half<HT>::mfma(b1[j], a1f[i], acc[i][j]);
      }
      wslot = wslot == 2 ? 0 : wslot + 1;
    }

    if (DBG & 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
      continue;
    }
    // ---- epilogue: x (fp32) out, exact LayerNorm statistics across the eight waves, y out ----
    float4 vx[4][2][4];  // [i][j][pass]: row i*32 + pass*8 + rrow, columns wave*64 + j*32 + rslot*4 .. +3
    f32x4_t gmv[2], btv[2];  // AE: gamma / beta of this lane's columns, requested behind block 6 (they land under block 7 and the statistics)
    if constexpr (AE) {
      // Residual rows one block ahead of the block being finished (32 VGPRs), bias / gamma / beta through the same path,
      // bounds through the buffer descriptors (reads of rows past M return 0, writes are dropped): no branch, no
      // compiler-placed vmcnt(0).  VMEM order per tile: B B | L0 L1 | S0 L2 | S1 L3 | ... | S5 L7 | S6 G | S7 (L / S / G four
      // instructions each); gfx9 retires them in order, so "block b's rows have landed" is a count of the younger
      // instructions.  Same arithmetic in the same order as the other form (patch value + bias, + residual).
      const int rows = min(RL_TM, g.M - m0);
      auto desc = [&](const void* base, long first_row, int ld, int n_rows) {
        const unsigned long long pa = (unsigned long long)base + (unsigned long long)first_row * ld * 4;
        u32x4_t r;
        r.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
        r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
        r.z = __builtin_amdgcn_readfirstlane((unsigned)(base ? n_rows * ld * 4 : 0));
        r.w = 0x00020000u;
        return r;
      };
      const u32x4_t rsR = desc(g.resid, m0, g.ldr, rows), rsO = desc(g.out_f32, m0, g.ldc, rows);
      const u32x4_t rsB = desc(g.bias, 0, RL_N, 1), rsG = desc(gamma, 0, RL_N, 1), rsT = desc(beta, 0, RL_N, 1);
      asm volatile("s_nop 4" ::: "memory");  // descriptors fresh from v_readfirstlane -> buffer_* inside asm strings
      const int colb = (wave * 64 + rslot * 4) * 4;
      auto off = [&](int blk, int pass, int ld) -> unsigned {
        return (unsigned)((((blk >> 1) * 32 + pass * 8 + rrow) * ld + (blk & 1) * 32) * 4 + colb);
      };
      f32x4_t bias4[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bias4[j]) : "v"((unsigned)(colb + j * 128)), "s"(rsB) : "memory");
      f32x4_t rr[2][4];
      auto load_resid = [&](int blk, f32x4_t (&r)[4]) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[pass]) : "v"(off(blk, pass, g.ldr)), "s"(rsR) : "memory");
      };
      load_resid(0, rr[0]); load_resid(1, rr[1]);
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        const int i = blk >> 1, j = blk & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = (2 * q + half) ^ (l31 & 7);
          *(float4*)(patch + l31 * 128 + slot * 16) =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
        f32x4_t(&r)[4] = rr[blk & 1];
        if (blk == 0)  // younger: L1
          asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
        else  // younger: the previous block's stores and the next block's rows (block 7: S6 and G)
          asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(bias4[j]) : : "memory");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int rw = pass * 8 + rrow;
          const float4 t = *(const float4*)(patch + rw * 128 + ((rslot ^ (rw & 7)) << 4));
          f32x4_t v;
          v[0] = t.x + bias4[j][0]; v[1] = t.y + bias4[j][1]; v[2] = t.z + bias4[j][2]; v[3] = t.w + bias4[j][3];
          v[0] += r[pass][0]; v[1] += r[pass][1]; v[2] += r[pass][2]; v[3] += r[pass][3];
          // s_nop 1: a > 64-bit asm store must not be followed at once by a write of its data registers
          asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(off(blk, pass, g.ldc)), "s"(rsO) : "memory");
          vx[i][j][pass] = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (blk + 2 < 8) load_resid(blk + 2, rr[blk & 1]);
        if (blk == 6) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(gmv[jj]) : "v"((unsigned)(colb + jj * 128)), "s"(rsG) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(btv[jj]) : "v"((unsigned)(colb + jj * 128)), "s"(rsT) : "memory");
          }
        }
      }
    } else {
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
          if (DBG & 32) { if (row < g.M && g.out_f32 && rslot < 4) *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v; }  // ablation: first 64 B of every x line
          else if (DBG & 64) { if (row < g.M && g.out_f32 && j == 0) *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v; }  // ablation: every other x line
          else if (row < g.M && g.out_f32) *(float4*)(g.out_f32 + (long)row * g.ldc + col) = v;
          vx[i][j][pass] = v;
        }
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
    if constexpr (AE)  // gamma / beta have landed: the only younger VMEM instructions are the stores of block 7
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(gmv[0]), "+v"(gmv[1]), "+v"(btv[0]), "+v"(btv[1]) : : "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 64 + j * 32 + rslot * 4;
      float4 gm, bt;
      if constexpr (AE) {
        gm = make_float4(gmv[j][0], gmv[j][1], gmv[j][2], gmv[j][3]);
        bt = make_float4(btv[j][0], btv[j][1], btv[j][2], btv[j][3]);
      } else {
        gm = *(const float4*)(gamma + col);
        bt = *(const float4*)(beta + col);
      }
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
          if (DBG & 16) asm volatile("" ::"v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));  // ablation: y not stored
          else if (row < g.M) *(uint4*)(oa + (long)row * RL_N + c8) = d;
        }
      }
    }
  }
}

}  // namespace

int g_gemm256_min_m = 8192;  // below: the tiled kernel (64-wide tiles on small grids) is 1.6-2x faster at 2-5 k rows, equal at 9.6 k (tools/probes/mid_m_gemm.py)
int g_w_dbg = 0;  // gemm256x / gemm_rowln A/B switches (test option w_dbg): bit0 no s_setprio, bit1 DMA after the fragment reads, bit3 compiler-scheduled fp32 epilogue (gemm_rowln: x phase) instead of the asm-counted one; bits 8.. timing ablations

// x <- x + A.W^T + b with y = LayerNorm(x) from the same launch (gemm_rowln_kernel): N = 512 rows only.
int g_rowln_min_m = 8192;  // below: tiled GEMM + LayerNorm pass (19 vs 27 us at 4.8 k rows, equal at 9.6 k; tools/probes/mid_m_rowln.py)
bool gemm_rowln_eligible(const GemmArgs& g) {
  return g.M >= g_rowln_min_m && g.N == RL_N && g.K % 32 == 0 && g.K >= 64 && g.ldc == RL_N && g.act == ACT_NONE && g.out_act &&
         g.out_f32 && g.ln_gamma && g.ln_beta && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         (!g.resid || g.ldr % 4 == 0) && (long)RL_TM * g.lda * 2 < (1L << 31) && (long)RL_N * g.ldw * 2 < (1L << 31);
}
int launch_gemm_rowln(const GemmArgs& g, hipStream_t st) {
  static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit&) -> int {
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    return 0;
  });
  if (init.rc) return launch_init_failed("gemm_rowln");
  if (!gemm_rowln_eligible(g)) {
    snprintf(g_err, sizeof(g_err), "gemm_rowln: shape not eligible (M=%d N=%d K=%d)", g.M, g.N, g.K);
    return 1;
  }
  const int tiles_m = cdiv(g.M, RL_TM);
  dim3 grid(tiles_m < init.n_cu ? tiles_m : init.n_cu), block(512);
#ifdef CZC_EXPERIMENTS
  if ((g_w_dbg >> 8) && !g.f16) {  // timing ablations (tools/ab_gemm.py, out_mode 4)
#define CZC_RL_ABL(D_) case D_: \
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rowln_kernel<false, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS)); \
    hipLaunchKernelGGL((gemm_rowln_kernel<false, D_>), grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m); break;
    switch (g_w_dbg >> 8) {
      CZC_RL_ABL(1) CZC_RL_ABL(2) CZC_RL_ABL(3) CZC_RL_ABL(7) CZC_RL_ABL(8) CZC_RL_ABL(9) CZC_RL_ABL(10) CZC_RL_ABL(13) CZC_RL_ABL(15) CZC_RL_ABL(16) CZC_RL_ABL(32) CZC_RL_ABL(64)
      default: snprintf(g_err, sizeof(g_err), "gemm_rowln: ablation %d not built", g_w_dbg >> 8); return 1;
    }
#undef CZC_RL_ABL
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
#endif
  // start stagger (start_stagger): 2 us per phase + 16 us for the work-groups with a tile less, from four tiles per
  // work-group on (fewer: the delay costs more than the de-synchronised epilogues return); w_dbg bit 2 = off, bits 4-7 /
  // 8.. = tau / bonus of an A/B run
  int stagger = tiles_m / (int)grid.x >= 4 ? (2 | 16 << 8) : 0;
  if (g_w_dbg & 4) stagger = 0;
  if (g_w_dbg >> 4) stagger = ((g_w_dbg >> 4) & 15) | ((g_w_dbg >> 8) << 8);
  if (g_w_dbg & 8) {  // compiler-scheduled x phase (A/B, bit-identical)
    if (g.f16) hipLaunchKernelGGL((gemm_rowln_kernel<true, 0, false>), grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m);
    else hipLaunchKernelGGL((gemm_rowln_kernel<false, 0, false>), grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m);
  } else if (g.f16) hipLaunchKernelGGL(gemm_rowln_kernel<true>, grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m, stagger);
  else hipLaunchKernelGGL(gemm_rowln_kernel<false>, grid, block, RL_LDS, st, g, g.ln_gamma, g.ln_beta, tiles_m, stagger);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

bool gemm256_eligible(const GemmArgs& g) {
  if (g.x16 && !(g.resid && g.out_f32 && !g.out_act && g.act == ACT_NONE && g.ldr % 8 == 0)) return false;  // x16: residual-add layers only
  return g.M >= g_gemm256_min_m && g.N % 8 == 0 && g.K % 64 == 0 && g.ldc % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         (!g.resid || g.ldr % 4 == 0) && (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) && (long)256 * g.lda * 2 < (1L << 31) &&
         (long)256 * g.ldw * 2 < (1L << 31);
}

// g_use_gemm256: 0 = 128x128 kernel only; 1 (default) = gemm256x where the output is fp32 (+ residual), gemm256q where
// it is activation-typed; 3 / 5 pin gemm256q / gemm256x for every epilogue (A/B runs and kernel tests)
int launch_gemm256(const GemmArgs& g, hipStream_t st) {
  constexpr int shp = QS * QSTAGE + 8 * 4096;  // ring + one 4 KiB epilogue patch per MFMA wave = the full 160 KiB
  static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit&) -> int {
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, shp))
#define CZC_ATTR4(K_) CZC_ATTR((K_<ACT_NONE, false, false>)); CZC_ATTR((K_<ACT_NONE, true, false>)); \
                      CZC_ATTR((K_<ACT_QUICK_GELU, false, false>)); CZC_ATTR((K_<ACT_QUICK_GELU, true, false>)); \
                      CZC_ATTR((K_<ACT_NONE, false, true>)); CZC_ATTR((K_<ACT_NONE, true, true>)); \
                      CZC_ATTR((K_<ACT_QUICK_GELU, false, true>)); CZC_ATTR((K_<ACT_QUICK_GELU, true, true>))
    CZC_ATTR4(gemm256q_kernel);
    CZC_ATTR4(gemm256x_kernel);
#ifdef CZC_EXPERIMENTS
    CZC_ATTR((gemm256r_kernel<false>));
    CZC_ATTR((gemm256r_kernel<true>));
#endif
    CZC_ATTR((gemm256x_kernel<ACT_NONE, true, false, 256>));
    CZC_ATTR((gemm256x_kernel<ACT_NONE, true, true, 256>));
#undef CZC_ATTR4
#undef CZC_ATTR
    return 0;
  });
  if (init.rc) return launch_init_failed("gemm256");
  const int n_cu = init.n_cu;
  const int tiles_m = cdiv(g.M, TM), tiles_n = cdiv(g.N, TN);
  const bool f32 = g.out_f32 != nullptr || g.resid != nullptr;
  dim3 gq(tiles_m * tiles_n < n_cu ? tiles_m * tiles_n : n_cu);
  const bool pp = g_use_gemm256 == 5 || (g_use_gemm256 != 3 && f32);
#ifdef CZC_EXPERIMENTS
  if (pp && g_use_gemm256 != 9 && (g_w_dbg >> 8) && f32 && g.act == ACT_NONE && !g.f16) {  // timing ablations of the ping-pong kernel
#define CZC_GOXD(D_)                                                                                                     \
  case D_:                                                                                                               \
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256x_kernel<ACT_NONE, true, false, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, shp)); \
    hipLaunchKernelGGL((gemm256x_kernel<ACT_NONE, true, false, D_>), gq, dim3(512), shp, st, g, tiles_m, tiles_n, g_w_dbg & 255); \
    break
    switch (g_w_dbg >> 8) {
      CZC_GOXD(1); CZC_GOXD(2); CZC_GOXD(3); CZC_GOXD(4); CZC_GOXD(5); CZC_GOXD(6); CZC_GOXD(8); CZC_GOXD(9); CZC_GOXD(10); CZC_GOXD(12); CZC_GOXD(14);
      CZC_GOXD(16); CZC_GOXD(32); CZC_GOXD(48); CZC_GOXD(13); CZC_GOXD(7);
      CZC_GOXD(64); CZC_GOXD(128); CZC_GOXD(71); CZC_GOXD(135); CZC_GOXD(199);
      default: snprintf(g_err, sizeof(g_err), "gemm256x: ablation %d not built", g_w_dbg >> 8); return 1;
    }
#undef CZC_GOXD
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
#endif
#define CZC_GO(K_, A_, F_, H_, T_, ...) hipLaunchKernelGGL((K_<A_, F_, H_>), gq, dim3(T_), shp, st, g, tiles_m, tiles_n, ##__VA_ARGS__)
#define CZC_DISPATCH(K_, T_, ...)                                                                                          \
  do {                                                                                                                     \
    if (g.f16) {                                                                                                           \
      if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GO(K_, ACT_QUICK_GELU, true, true, T_, ##__VA_ARGS__); else CZC_GO(K_, ACT_QUICK_GELU, false, true, T_, ##__VA_ARGS__); } \
      else { if (f32) CZC_GO(K_, ACT_NONE, true, true, T_, ##__VA_ARGS__); else CZC_GO(K_, ACT_NONE, false, true, T_, ##__VA_ARGS__); }                               \
    } else {                                                                                                               \
      if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GO(K_, ACT_QUICK_GELU, true, false, T_, ##__VA_ARGS__); else CZC_GO(K_, ACT_QUICK_GELU, false, false, T_, ##__VA_ARGS__); } \
      else { if (f32) CZC_GO(K_, ACT_NONE, true, false, T_, ##__VA_ARGS__); else CZC_GO(K_, ACT_NONE, false, false, T_, ##__VA_ARGS__); }                             \
    }                                                                                                                      \
  } while (0)
  // gemm256x with the fp32-residual epilogue (the HBM burst at the end of every tile): start stagger (start_stagger) of 2 us
  // per phase + 32 us for the work-groups with a tile less, from four tiles per work-group on: fc2 -1.7 % at 312 k rows,
  // -1.9 % at 156 k (its epilogue is a fifth of a tile; gemm_rowln's is two thirds and gains 10 %).  w_dbg bit 2 = off,
  // bits 4-7 / 8.. = tau / bonus of an A/B run
  int stagger256 = (f32 && tiles_m * tiles_n / (int)gq.x >= 4) ? (2 | 32 << 8) : 0;
  if (g_w_dbg & 4) stagger256 = 0;
  if (f32 && (g_w_dbg >> 4)) stagger256 = ((g_w_dbg >> 4) & 15) | ((g_w_dbg >> 8) << 8);
  if (g.x16) {  // fp16 residual stream: ping-pong kernel with the 2-byte epilogue (eligibility: gemm256_eligible)
    if (g.f16) hipLaunchKernelGGL((gemm256x_kernel<ACT_NONE, true, true, 256>), gq, dim3(512), shp, st, g, tiles_m, tiles_n, (g_w_dbg & 7) | stagger256 << 8);
    else hipLaunchKernelGGL((gemm256x_kernel<ACT_NONE, true, false, 256>), gq, dim3(512), shp, st, g, tiles_m, tiles_n, (g_w_dbg & 7) | stagger256 << 8);
  } else if (g_use_gemm256 == 9 && f32 && g.act == ACT_NONE && g.resid && g.out_f32 && !g.out_act) {  // register-staged four-wave arm (A/B)
#ifdef CZC_EXPERIMENTS
    if ((g_w_dbg >> 8) && !g.f16) {
#define CZC_GORD(D_) case D_: \
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256r_kernel<false, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, shp)); \
      hipLaunchKernelGGL((gemm256r_kernel<false, D_>), gq, dim3(256), shp, st, g, tiles_m, tiles_n, stagger256 << 8); break
      switch (g_w_dbg >> 8) {
        CZC_GORD(1); CZC_GORD(2); CZC_GORD(4); CZC_GORD(8); CZC_GORD(16); CZC_GORD(3); CZC_GORD(6); CZC_GORD(9); CZC_GORD(10); CZC_GORD(14); CZC_GORD(22); CZC_GORD(30); CZC_GORD(29); CZC_GORD(17); CZC_GORD(24);
        default: snprintf(g_err, sizeof(g_err), "gemm256r: ablation %d not built", g_w_dbg >> 8); return 1;
      }
#undef CZC_GORD
      CZC_HIP_CHECK(hipGetLastError());
      return 0;
    }
    if (g.f16) hipLaunchKernelGGL((gemm256r_kernel<true>), gq, dim3(256), shp, st, g, tiles_m, tiles_n, stagger256 << 8);
    else hipLaunchKernelGGL((gemm256r_kernel<false>), gq, dim3(256), shp, st, g, tiles_m, tiles_n, stagger256 << 8);
#else
    snprintf(g_err, sizeof(g_err), "gemm256 = 9 (gemm256r, the register-staged A/B arm) exists in EXPERIMENTS=1 builds only");
    return 1;
#endif
  } else if (pp) CZC_DISPATCH(gemm256x_kernel, 512, (g_w_dbg & 15) | stagger256 << 8);
  else CZC_DISPATCH(gemm256q_kernel, 768);
#undef CZC_DISPATCH
#undef CZC_GO
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int g_use_gemm256s = 1;  // 0: split-fp16 layers stay on the 128x128 kernel
int g_gemm256s_min_m = 16384;  // rows from which a split-fp16 layer takes the 256 x 256 ring (test option gemm256s_min_m)

// split-fp16 operands; big-M layers only (BERT at a few thousand rows stays on the 128x128 + split-K path)
bool gemm256s_eligible(const GemmArgs& g) {
  return g_use_gemm256s && g.M >= g_gemm256s_min_m && g.N % 8 == 0 && g.K % 32 == 0 && g.ldc % 8 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         (!g.resid || g.ldr % 4 == 0) && (g.act == ACT_NONE || g.act == ACT_QUICK_GELU) &&
         !(g.out_act && g.out_f32) && (long)256 * g.lda * 4 < (1L << 31) && (long)256 * g.ldw * 4 < (1L << 31);
}

int launch_gemm256s(const GemmArgs& g, hipStream_t st) {
  constexpr int shp = QS * QSTAGE + 8 * 4096;
  static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit&) -> int {
#define CZC_ATTR(K_) CZC_HIP_CHECK(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, shp))
    CZC_ATTR((gemm256sq_kernel<ACT_NONE, false>));
    CZC_ATTR((gemm256sq_kernel<ACT_NONE, true>));
    CZC_ATTR((gemm256sq_kernel<ACT_QUICK_GELU, false>));
    CZC_ATTR((gemm256sq_kernel<ACT_QUICK_GELU, true>));
#undef CZC_ATTR
    return 0;
  });
  if (init.rc) return launch_init_failed("gemm256s");
  const int tiles_m = cdiv(g.M, TM), tiles_n = cdiv(g.N, TN);
  dim3 grid(tiles_m * tiles_n < init.n_cu ? tiles_m * tiles_n : init.n_cu), block(768);
  const bool f32 = g.out_f32 != nullptr || g.resid != nullptr;
#define CZC_GOSQ(A_, F_) hipLaunchKernelGGL((gemm256sq_kernel<A_, F_>), grid, block, shp, st, g, tiles_m, tiles_n)
  if (g.act == ACT_QUICK_GELU) { if (f32) CZC_GOSQ(ACT_QUICK_GELU, true); else CZC_GOSQ(ACT_QUICK_GELU, false); }
  else { if (f32) CZC_GOSQ(ACT_NONE, true); else CZC_GOSQ(ACT_NONE, false); }
#undef CZC_GOSQ
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

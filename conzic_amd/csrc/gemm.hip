// MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T (+bias)(act)(+resid)
//
// This is the kernel the polishing step spends >98% of its FLOPs in (SURVEY.md §8a row A8:
// the CLIP text tower over the B*K candidate captions; HF:clip/modeling_clip.py:309-350), and
// it also serves the BERT masked-LM tower (HF:bert/modeling_bert.py:175-203, :476-496) and the
// CLIP vision tower.  Both operands are K-contiguous (activations row-major, nn.Linear weights
// [out,in]), which is exactly the MFMA A/B fragment order, so no transposes are ever needed.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles),
// K step = 128 bytes per row (64 bf16 / 32 f32), LDS double-buffered (2 x 32 KiB), 16-byte
// chunks XOR-swizzled by ((row>>1)&7) so that a ds_read_b128 lane group (16 distinct rows,
// 128-byte row pitch) touches 16 distinct 16-byte slots of the 256-byte bank row.
// bf16 mode: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  f32 mode: v_mfma_f32_32x32x2_f32
// (exact fp32 fma chain) -- the verification path.
//
// Work-group -> tile map is XCD-aware: consecutive ids inside one XCD's share walk the N tiles
// of the same M tile, so the activation tile is fetched into one L2 only.
#include <map>
#include <mutex>

#include "kernels.h"

namespace czc {

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per tile row

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KPT = ROWB / 2;  // K elements per tile
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                0, 0);
  }
};
template <> struct Mma<f16_t> {
  static constexpr int KPT = ROWB / 2;
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<split_t> {
  static constexpr int KPT = ROWB / 4;  // 32 elements per 128-byte tile row (4 groups of 8 hi + 8 lo)
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KPT = ROWB / 4;
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    // lane half h holds k = 4h..4h+3 of this 8-wide k group; MFMA #e contracts k in {e, 4+e}
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- epilogue -----------------------------------------------------------------------------------
// The MFMA is issued with the WEIGHT fragment as its A operand and the ACTIVATION fragment as its
// B operand, so D[row][col] = C^T: col (lane&31) is the output ROW m and the accumulator
// registers walk the output COLUMNS n = (r&3) + 8*(r>>2) + 4*(lane>>5).  Every group of four
// registers is therefore four consecutive columns of one row: bias / residual / stores are
// 8-byte (bf16) or 16-byte (fp32) vector accesses, 4x fewer instructions than the natural layout.
template <typename T, int ACT>
__device__ __forceinline__ float apply_act(float v) {
  if (ACT == ACT_QUICK_GELU) {
    // half-precision engines: the same raw v_exp_f32 / v_rcp_f32 form as the ring and weight-stationary kernels
    // (gemm256.hip act_fn, gemm_wreg.hip wr_act) -- a layer must not change its last bits with the row count
    if (sizeof(T) == 2) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v));
    return v / (1.0f + expf(-1.702f * v));
  }
  if (ACT == ACT_GELU_ERF) return gelu_erf(v);
  return v;
}

// VEC: N % 4 == 0 and ldc/ldr % 4 == 0 (every call of the polishing step except the 30522-wide
// MLM decoder, which takes the scalar form).
template <typename T, int ACT, bool VEC, int NB = 2>
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x16_t (&acc)[NB][NB], int m0, int n0, int wm, int wn,
                                         int lane) {
  const int half = lane >> 5;
  T* oa = (T*)g.out_act;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = m0 + wm * (32 * NB) + i * 32 + (lane & 31);
    if (row >= g.M) continue;
    const long ro = (long)row * g.ldc;
    const long rr = (long)row * g.ldr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float st_s = 0.f, st_q = 0.f;  // LayerNorm partials of this lane's quads (x16 layers with GemmArgs::row_part)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * (32 * NB) + j * 32 + 8 * q + 4 * half;
        if (VEC) {
          if (col >= g.N) continue;
          float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          if (g.bias) {
            const float4 b = *(const float4*)(g.bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          v.x = apply_act<T, ACT>(v.x); v.y = apply_act<T, ACT>(v.y);
          v.z = apply_act<T, ACT>(v.z); v.w = apply_act<T, ACT>(v.w);
          if (sizeof(T) == 2 && g.x16) {
            // 2-byte residual stream (GemmArgs::x16): fp16 rows in, fp16 rows out, the sum formed in fp32 and rounded once --
            // the arithmetic of tile_epilogue_x16_asm (gemm256.hip), so the layer does not change its bits with the row count
            const f16_t* rh = (const f16_t*)g.resid;
            f16_t* oh = (f16_t*)g.out_f32;
            if (rh) {
              const uint2 r2 = *(const uint2*)(rh + rr + col);
              v.x += (float)__builtin_bit_cast(_Float16, (unsigned short)(r2.x & 0xffffu));
              v.y += (float)__builtin_bit_cast(_Float16, (unsigned short)(r2.x >> 16));
              v.z += (float)__builtin_bit_cast(_Float16, (unsigned short)(r2.y & 0xffffu));
              v.w += (float)__builtin_bit_cast(_Float16, (unsigned short)(r2.y >> 16));
            }
            const uint2 pk = make_uint2(pack2_f16(v.x, v.y), pack2_f16(v.z, v.w));
            if (oh) *(uint2*)(oh + ro + col) = pk;
            if (g.row_part) {  // the association order of the weight-stationary residual kernel: quads in order, then half 0 + half 1
              const float w0 = f16lo(pk.x), w1 = f16hi(pk.x), w2 = f16lo(pk.y), w3 = f16hi(pk.y);
              const float ps = ln_sum4(w0, w1, w2, w3), pq = ln_sq4(w0, w1, w2, w3);
              st_s = q == 0 ? ps : __fadd_rn(st_s, ps);
              st_q = q == 0 ? pq : __fadd_rn(st_q, pq);
              if (q == 3) {
                const float os = __shfl_xor(st_s, 32, 64), oq = __shfl_xor(st_q, 32, 64);
                if (!half) {
                  const long pcol = (n0 + wn * (32 * NB) + j * 32) / 32;
                  *(float2*)(g.row_part + (pcol * g.part_ld + row) * 2) = make_float2(__fadd_rn(st_s, os), __fadd_rn(st_q, oq));
                }
              }
            }
            continue;
          }
          if (g.resid) {
            const float4 r4 = *(const float4*)(g.resid + rr + col);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          if (g.out_f32) *(float4*)(g.out_f32 + ro + col) = v;
          if (oa) Act<T>::st4(oa, ro + col, v.x, v.y, v.z, v.w);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = col + e;
            if (c >= g.N) continue;
            float v = acc[i][j][4 * q + e] + (g.bias ? g.bias[c] : 0.f);
            v = apply_act<T, ACT>(v);
            if (g.resid) v += g.resid[rr + c];
            if (g.out_f32) g.out_f32[ro + c] = v;
            if (oa) Act<T>::st(oa, ro + c, v);
          }
        }
      }
    }
  }
}

// DEEP: operand requests run three K steps ahead (three rotating register sets) instead of one -- for launches of at most one
// work-group per CU (one or two images, the vision tower at small batches), which are chains of exposed L2 / HBM round
// trips; with several work-groups per CU (BERT at hundreds of images) the vector-memory path is the bound and the extra
// registers in flight cost 9 %, so the launcher picks by grid size.  Same summation order.
// TS: tile side, 128 or 64 (the same kernel with one 32x32 MFMA block per wave instead of 2x2): a launch that would put 128-wide
// tiles on less than a quarter of the CUs (one image: 40 tiles for fc2) runs four times as many work-groups, each with a
// quarter of the MFMA work per K step, on the same chain of operand round trips.  Same k order per output element.
template <typename T, int ACT, bool VEC, bool DEEP = false, int TS = 128>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, int tiles_m, int tiles_n, int kchunk) {
  constexpr int NB = TS / 64;                // 32x32 MFMA blocks per wave and dimension
  constexpr int TILE_B = TS * ROWB, STAGE_B = 2 * TILE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  // split-K launch (gridDim.y > 1): slice z multiplies k in [z*kchunk, (z+1)*kchunk) into its own fp32 slab
  // out_f32 + z*M*ldc (no bias / activation / residual: splitk_reduce_kernel adds them in a fixed order)
  int kshift = 0;  // bytes the operand bases are moved into their rows (split-K): the descriptors still end where the rows end
  if (gridDim.y > 1) {
    const int z = blockIdx.y;
    kshift = z * kchunk * (int)sizeof(T);
    g.A = (const unsigned char*)g.A + (size_t)kshift;
    g.W = (const unsigned char*)g.W + (size_t)kshift;
    g.out_f32 += (size_t)z * g.M * g.ldc;
    g.K = kchunk;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap of the linear work-group id (8 XCDs, block b runs on XCD b%8)
  const int nwg = tiles_m * tiles_n;
  int lin = blockIdx.x;
  {
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    lin = base + (lin >> 3);
  }
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * TS, n0 = tn * TS;

  // Rebased buffer descriptors: rows past M / N read as zero (hardware bounds check), offsets are
  // 32-bit and linear in the staging index (no per-row pointer arrays -> no scratch).
  const int lda_b = g.lda * (int)sizeof(T), ldw_b = g.ldw * (int)sizeof(T);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.A + (long)m0 * lda_b), (short)0,
                                                     min(TS, g.M - m0) * lda_b - kshift, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.W + (long)n0 * ldw_b), (short)0,
                                                     min(TS, g.N - n0) * ldw_b - kshift, 0x00020000);
  // staging map: thread t moves chunk (t&7) of rows (t>>3) + 32*i, i = 0..3, of both tiles
  const int sc = tid & 7, sr = tid >> 3;
  const int voA = sr * lda_b + sc * 16, voW = sr * ldw_b + sc * 16;
  const int soff = swz(sr, sc);  // rows +32: same swizzle phase, +4096 bytes

  f32x16_t acc[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / Mma<T>::KPT;
  // requests past the last K step go to an empty descriptor: zeros, no memory traffic, never stored (so that no VMEM
  // instruction sits behind a branch and hipcc keeps its counted vmcnt waits)
  const auto rsNone = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, (short)0, 0, 0x00020000);
  u32x4_t pa0, pa1, pa2, pa3, pw0, pw1, pw2, pw3, qa0, qa1, qa2, qa3, qw0, qw1, qw2, qw3, ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define CZC_LOAD_TILE(S, KT)                                                                       \
  {                                                                                                \
    const auto ra_ = (KT) < nk ? rsA : rsNone;                                                     \
    const auto rw_ = (KT) < nk ? rsW : rsNone;                                                     \
    S##a0 = __builtin_amdgcn_raw_buffer_load_b128(ra_, voA, (KT) * ROWB, 0);                       \
    S##a1 = __builtin_amdgcn_raw_buffer_load_b128(ra_, voA + 32 * lda_b, (KT) * ROWB, 0);          \
    S##w0 = __builtin_amdgcn_raw_buffer_load_b128(rw_, voW, (KT) * ROWB, 0);                       \
    S##w1 = __builtin_amdgcn_raw_buffer_load_b128(rw_, voW + 32 * ldw_b, (KT) * ROWB, 0);          \
    if constexpr (TS == 128) {                                                                     \
      S##a2 = __builtin_amdgcn_raw_buffer_load_b128(ra_, voA + 64 * lda_b, (KT) * ROWB, 0);        \
      S##a3 = __builtin_amdgcn_raw_buffer_load_b128(ra_, voA + 96 * lda_b, (KT) * ROWB, 0);        \
      S##w2 = __builtin_amdgcn_raw_buffer_load_b128(rw_, voW + 64 * ldw_b, (KT) * ROWB, 0);        \
      S##w3 = __builtin_amdgcn_raw_buffer_load_b128(rw_, voW + 96 * ldw_b, (KT) * ROWB, 0);        \
    }                                                                                              \
  }
#define CZC_STORE_TILE(S, dst)                                   \
  *(u32x4_t*)((dst) + soff) = S##a0;                             \
  *(u32x4_t*)((dst) + soff + 4096) = S##a1;                      \
  *(u32x4_t*)((dst) + TILE_B + soff) = S##w0;                    \
  *(u32x4_t*)((dst) + TILE_B + soff + 4096) = S##w1;             \
  if constexpr (TS == 128) {                                     \
    *(u32x4_t*)((dst) + soff + 8192) = S##a2;                    \
    *(u32x4_t*)((dst) + soff + 12288) = S##a3;                   \
    *(u32x4_t*)((dst) + TILE_B + soff + 8192) = S##w2;           \
    *(u32x4_t*)((dst) + TILE_B + soff + 12288) = S##w3;          \
  }
  CZC_LOAD_TILE(p, 0)
  if constexpr (DEEP) { CZC_LOAD_TILE(q, 1) CZC_LOAD_TILE(r, 2) }
  CZC_STORE_TILE(p, smem)
  __syncthreads();

  const int arow = wm * (32 * NB) + (lane & 31);
  const int brow = wn * (32 * NB) + (lane & 31);
  const int half = lane >> 5;

  auto compute = [&](int kt) {
    const unsigned char* sA = smem + (kt & 1) * STAGE_B;
    const unsigned char* sB = sA + TILE_B;
    if constexpr (sizeof(T) == 4 && !__is_same(T, float)) {
      // split_t: chunk 2g = hi plane, 2g+1 = lo plane of k-group g; MFMA step s takes group 2s+half.
      // (a_hi + a_lo)(w_hi + w_lo) ~ a_hi w_hi + a_lo w_hi + a_hi w_lo   (lo*lo ~ 2^-22, dropped)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ch = 2 * (2 * s2 + half);
        uint4 ah[NB], al[NB], bh[NB], bl[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          ah[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch));
          al[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch + 1));
          bh[i] = *(const uint4*)(sB + swz(brow + 32 * i, ch));
          bl[i] = *(const uint4*)(sB + swz(brow + 32 * i, ch + 1));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            Mma<T>::run(bl[j], ah[i], acc[i][j]);
            Mma<T>::run(bh[j], al[i], acc[i][j]);
            Mma<T>::run(bh[j], ah[i], acc[i][j]);
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + half;
        uint4 a[NB], b[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          a[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch));
          b[i] = *(const uint4*)(sB + swz(brow + 32 * i, ch));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);  // weight = A operand: D = C^T
      }
    }
  };

  if constexpr (DEEP) {
    // three register sets rotate: at step t the LDS buffer t & 1 holds step t, two sets hold steps t+1 and t+2 (in
    // flight), the third is free and requests step t+3; after the MFMAs step t+1 goes to the other LDS buffer
#define CZC_DEEP_STEP(T_, FREE_, NEXT_)                                                                   \
    {                                                                                                     \
      CZC_LOAD_TILE(FREE_, (T_) + 3)                                                                      \
      __builtin_amdgcn_sched_barrier(0); /* the LDS stores of the older set stay BEHIND the MFMAs: hipcc */ \
      compute(T_);                       /* would hoist them (and their vmcnt wait) in front             */ \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      if ((T_) + 1 < nk) { CZC_STORE_TILE(NEXT_, smem + (((T_) + 1) & 1) * STAGE_B) }                 \
      __syncthreads();                                                                                    \
    }
    for (int kt = 0; kt < nk; kt += 3) {
      CZC_DEEP_STEP(kt, p, q)
      if (kt + 1 >= nk) break;
      CZC_DEEP_STEP(kt + 1, q, r)
      if (kt + 2 >= nk) break;
      CZC_DEEP_STEP(kt + 2, r, p)
    }
#undef CZC_DEEP_STEP
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) { CZC_LOAD_TILE(p, kt + 1) }
      compute(kt);
      if (kt + 1 < nk) {
        unsigned char* dA = smem + ((kt + 1) & 1) * STAGE_B;
        CZC_STORE_TILE(p, dA)
      }
      __syncthreads();
    }
  }
#undef CZC_LOAD_TILE
#undef CZC_STORE_TILE
  epilogue<T, ACT, VEC, NB>(g, acc, m0, n0, wm, wn, lane);
}

int g_gemm_deep = 1;  // 0 never, 1 when the launch has at most one work-group per CU, 2 always (test option "gemm_deep")

// 64-wide tiles when the 128-wide tiles of a launch would number at most N/4 of the CUs (test option "gemm_small_tiles"; 0 never).
// N = 1 is where one to sixteen images gain (+8..35 %); 2 adds +5 % at 32 images, 4 (= every launch of at most one 128-wide
// tile per CU that is not split along K) +1.5 % at 128; 256 images unchanged (profiles/r03_small_tiles_ab.txt)
int g_gemm_small_tiles = 4;

static int gemm_n_cu() {  // CU count of the CURRENT device (engines may live on different GPUs of one process)
  static PerDeviceInit per_dev;
  const LaunchInit li = per_dev.get([](LaunchInit&) -> int { return 0; });
  return li.rc == 0 && li.n_cu > 0 ? li.n_cu : 256;
}

// split_small: a split-K launch whose K slices were sized for 64-wide tiles (try_splitk)
template <typename T>
static void launch_t(const GemmArgs& g, int tiles_m, int tiles_n, bool vec, hipStream_t st, int ksplit = 1, bool split_small = false) {
  const int kchunk = g.K / ksplit;
  const int n_cu = gemm_n_cu();
  const bool deep = split_small || g_gemm_deep == 2 || (g_gemm_deep == 1 && tiles_m * tiles_n * ksplit <= n_cu);  // at most one work-group per CU
  const bool small = split_small || (g_gemm_small_tiles && deep && ksplit == 1 && tiles_m * tiles_n * 4 <= n_cu * g_gemm_small_tiles);
  if (small) { tiles_m = cdiv(g.M, 64); tiles_n = cdiv(g.N, 64); }
  dim3 grid(tiles_m * tiles_n, ksplit), block(256);
#define CZC_GEMM_LAUNCH(ACT_, VEC_)                                                                                        \
  do {                                                                                                                     \
    if (small) hipLaunchKernelGGL((gemm_kernel<T, ACT_, VEC_, true, 64>), grid, block, 0, st, g, tiles_m, tiles_n, kchunk); \
    else if (deep) hipLaunchKernelGGL((gemm_kernel<T, ACT_, VEC_, true>), grid, block, 0, st, g, tiles_m, tiles_n, kchunk); \
    else hipLaunchKernelGGL((gemm_kernel<T, ACT_, VEC_, false>), grid, block, 0, st, g, tiles_m, tiles_n, kchunk);        \
  } while (0)
  if (vec) {
    if (g.act == ACT_QUICK_GELU) CZC_GEMM_LAUNCH(ACT_QUICK_GELU, true);
    else if (g.act == ACT_GELU_ERF) CZC_GEMM_LAUNCH(ACT_GELU_ERF, true);
    else CZC_GEMM_LAUNCH(ACT_NONE, true);
  } else {
    if (g.act == ACT_QUICK_GELU) CZC_GEMM_LAUNCH(ACT_QUICK_GELU, false);
    else if (g.act == ACT_GELU_ERF) CZC_GEMM_LAUNCH(ACT_GELU_ERF, false);
    else CZC_GEMM_LAUNCH(ACT_NONE, false);
  }
#undef CZC_GEMM_LAUNCH
}

// ---- skinny split-fp16 GEMM for M <= 32 (BERT at batch 1-2: weight streaming, latency bound) ------
// One work-group per 16 output columns; its 4 waves split K, each streaming its W rows straight
// from global memory into MFMA fragments (no LDS: every weight byte is used once), 16x16x32 f16
// MFMA with the weights as the A operand, so a lane ends up with 4 consecutive output columns of
// one row.  Partial sums meet in LDS; wave 0 applies bias / activation / residual and stores.
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

template <int ACT, bool TWO, int SK_U = 6>
__global__ __launch_bounds__(256) void gemm_skinny_split_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float red[3][2][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const int nl = lane & 15, kb = lane >> 4;
  const int nrow = min(n0 + nl, g.N - 1);
  const unsigned char* wp = (const unsigned char*)g.W + (long)nrow * g.ldw * 4 + kb * 32;
  const int m_a = min(nl, g.M - 1), m_b = min(16 + nl, g.M - 1);
  const unsigned char* xa = (const unsigned char*)g.A + (long)m_a * g.lda * 4 + kb * 32;
  const unsigned char* xb = (const unsigned char*)g.A + (long)m_b * g.lda * 4 + kb * 32;
  constexpr bool two = TWO;
  f32x4v_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int steps = g.K >> 5;  // 32 k per MFMA step = 4 groups of (16 B hi + 16 B lo)
#define CZC_F16(v_) __builtin_bit_cast(f16x8_t, v_)
  // The kernel is a chain of L2 / HBM round trips (a wave owns steps wave, wave+4, ...: 6 of them at K = 768), so the
  // operands of SK_U steps are requested together before their MFMAs run: one exposed latency per SK_U steps instead of
  // one per step (SK_U = 12 for the K = 3072 layer at <= 16 rows: two round trips instead of four).  The MFMAs keep their
  // order (same sums).  The epilogue's operands (bias, residual) are requested up front as well (round 4): they used to
  // cost wave 0 one more exposed round trip behind the LDS reduction.
  const int ep_col = n0 + 4 * kb;
  float ep_bias[4] = {0.f, 0.f, 0.f, 0.f}, ep_res[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = min(ep_col + r, g.N - 1);
      if (g.bias) ep_bias[r] = g.bias[c];
      if (g.resid) {
        ep_res[0][r] = g.resid[(long)m_a * g.ldr + c];
        if (two) ep_res[1][r] = g.resid[(long)m_b * g.ldr + c];
      }
    }
  }
  int s2 = wave;
  for (; s2 + 4 * (SK_U - 1) < steps; s2 += 4 * SK_U) {
    uint4 wh[SK_U], wl[SK_U], ah[SK_U], al[SK_U], bh[SK_U], bl[SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      const long o = (long)(s2 + 4 * u) * 128;
      wh[u] = *(const uint4*)(wp + o); wl[u] = *(const uint4*)(wp + o + 16);
      ah[u] = *(const uint4*)(xa + o); al[u] = *(const uint4*)(xa + o + 16);
      if (two) { bh[u] = *(const uint4*)(xb + o); bl[u] = *(const uint4*)(xb + o + 16); }
    }
    __builtin_amdgcn_sched_barrier(0);  // all requests first (hipcc would otherwise sink each load to its MFMA again)
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wl[u]), CZC_F16(ah[u]), acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh[u]), CZC_F16(al[u]), acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh[u]), CZC_F16(ah[u]), acc0, 0, 0, 0);
      if (two) {
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wl[u]), CZC_F16(bh[u]), acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh[u]), CZC_F16(bl[u]), acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh[u]), CZC_F16(bh[u]), acc1, 0, 0, 0);
      }
    }
  }
  for (; s2 < steps; s2 += 4) {
    const long o = (long)s2 * 128;
    const uint4 wh = *(const uint4*)(wp + o), wl = *(const uint4*)(wp + o + 16);
    const uint4 ah = *(const uint4*)(xa + o), al = *(const uint4*)(xa + o + 16);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wl), CZC_F16(ah), acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh), CZC_F16(al), acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh), CZC_F16(ah), acc0, 0, 0, 0);
    if (two) {
      const uint4 bh = *(const uint4*)(xb + o), bl = *(const uint4*)(xb + o + 16);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wl), CZC_F16(bh), acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh), CZC_F16(bl), acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(CZC_F16(wh), CZC_F16(bh), acc1, 0, 0, 0);
    }
  }
#undef CZC_F16
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[wave - 1][0][lane][r] = acc0[r]; red[wave - 1][1][lane][r] = acc1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0[r] += red[w][0][lane][r]; acc1[r] += red[w][1][lane][r]; }
  // D[row = n][col = m]: lane holds m = lane&15, n = n0 + 4*(lane>>4) + r
  split_t* oa = (split_t*)g.out_act;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = t * 16 + nl;
    if (m >= g.M) continue;
    const f32x4v_t a = t ? acc1 : acc0;
    const int col = ep_col;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = apply_act<float, ACT>(a[r] + ep_bias[r]);   // ep_bias is 0 without a bias: a + 0.f == a
      if (g.resid) v[r] += ep_res[t][r];
    }
    if (col + 3 < g.N && (g.ldc & 3) == 0) {
      if (g.out_f32) *(float4*)(g.out_f32 + (long)m * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
      if (oa) Act<split_t>::st4(oa, (long)m * g.ldc + col, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < g.N) {
          if (g.out_f32) g.out_f32[(long)m * g.ldc + col + r] = v[r];
          if (oa) Act<split_t>::st(oa, (long)m * g.ldc + col + r, v[r]);
        }
    }
  }
}

int g_use_skinny = 1;

static bool launch_skinny(const GemmArgs& g, hipStream_t st) {
  if (!g_use_skinny || g.M > 32 || (g.K & 31) || (g.lda & 7) || (g.ldw & 7)) return false;
  dim3 grid(cdiv(g.N, 16)), block(256);
#define CZC_SK(A_) do { if (g.M > 16) hipLaunchKernelGGL((gemm_skinny_split_kernel<A_, true>), grid, block, 0, st, g); \
                       else if (g.K >= 1536) hipLaunchKernelGGL((gemm_skinny_split_kernel<A_, false, 12>), grid, block, 0, st, g); \
                       else hipLaunchKernelGGL((gemm_skinny_split_kernel<A_, false>), grid, block, 0, st, g); } while (0)
  if (g.act == ACT_QUICK_GELU) CZC_SK(ACT_QUICK_GELU);
  else if (g.act == ACT_GELU_ERF) CZC_SK(ACT_GELU_ERF);
  else CZC_SK(ACT_NONE);
#undef CZC_SK
  return true;
}

// ---- deterministic split-K for small-M, long-K fp32-output layers (BERT out-proj / fc2 at a few thousand rows) ----
// 128x128 tiles give only M/128 * N/128 work-groups (180 for 3840 x 768) and each walks K through a two-stage
// pipeline whose step time is an L2/HBM round trip, not its 24 MFMAs: the kernel is latency-bound at 0.7
// work-groups per CU.  Splitting K over 2-4 work-groups per tile shortens the chain and fills the chip; the
// slabs are summed in slice order by one small kernel, so results do not depend on scheduling.
__global__ void splitk_reduce_kernel(const float* slabs, int nslab, long slab_stride, const float* bias, const float* resid,
                                     int ldr, float* out, int ldc, int M, int N) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the output
  const int n4 = N >> 2;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
  float4 v = *(const float4*)(slabs + (long)m * ldc + n);
  for (int z = 1; z < nslab; ++z) {
    const float4 p = *(const float4*)(slabs + z * slab_stride + (long)m * ldc + n);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  if (bias) { const float4 b = *(const float4*)(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (resid) { const float4 r = *(const float4*)(resid + (long)m * ldr + n); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  *(float4*)(out + (long)m * ldc + n) = v;
}

int g_use_splitk = 1;
// slab workspace per stream (two engines, or the two half-batch lanes of one engine, run split-K GEMMs concurrently)
struct SplitkWs { float* p = nullptr; size_t bytes = 0; };
static std::map<hipStream_t, SplitkWs> g_splitk_map;
static std::mutex g_splitk_mu;

// returns 0 when the shape does not want split-K, else the slice count it launched with
template <typename T>
static int try_splitk(const GemmArgs& g, int tiles_m, int tiles_n, hipStream_t st, int* rc) {
  *rc = 0;
  if (!g_use_splitk || g.out_act || !g.out_f32 || g.act != ACT_NONE || g.N % 4 || g.ldc % 4 || (g.resid && g.ldr % 4)) return 0;
  const int tiles = tiles_m * tiles_n;
  if (tiles >= 384) return 0;
  // K = 768 (out-projection): the unsplit kernel on 64-wide tiles is faster at every row count (26 vs 39 us at 3840 rows,
  // 14 vs 17 at 480; tools/probes/bert_gemm_forms.py); K = 3072 (fc2) gains from slices at every row count
  if (g_gemm_small_tiles && g.K < 2048) return 0;
  int ks = 0;
  bool split_small = false;
  if (g_gemm_small_tiles && tiles * 4 <= gemm_n_cu()) {
    // few tiles (up to ~64 images): 64-wide tiles, and as many slices (up to 8, at least 256 of K each) as it takes to
    // put a work-group on every CU
    const int t64 = cdiv(g.M, 64) * cdiv(g.N, 64);
    for (int c = 2; c <= 8; ++c)
      if (g.K % (c * Mma<T>::KPT) == 0 && g.K / c >= 256) { ks = c; if (t64 * c >= gemm_n_cu()) break; }
    split_small = ks != 0;
  } else {
    if (g.M < 256) return 0;
    for (int c = 4; c >= 2; --c)
      if (g.K % (c * Mma<T>::KPT) == 0 && g.K / c >= 256 && tiles * c <= 1024) { ks = c; break; }
  }
  if (!ks) return 0;
  const size_t need = (size_t)ks * g.M * g.ldc * 4;
  float* ws = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    SplitkWs& w = g_splitk_map[st];
    if (need > w.bytes) {
      if (w.p) { (void)hipStreamSynchronize(st); (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
      if (hipMalloc((void**)&w.p, need + need / 4) != hipSuccess) { *rc = 1; snprintf(g_err, sizeof(g_err), "split-K workspace allocation failed"); return ks; }
      w.bytes = need + need / 4;
    }
    ws = w.p;
  }
  GemmArgs p = g;
  p.bias = nullptr; p.resid = nullptr; p.out_f32 = ws; p.splitk_pending = nullptr;
  launch_t<T>(p, tiles_m, tiles_n, true, st, ks, split_small);
  if (g.splitk_pending) {  // the caller's next kernel sums the slabs itself (launch_layernorm_splitk)
    g.splitk_pending->slabs = ws; g.splitk_pending->nslab = ks; g.splitk_pending->slab_stride = (long)g.M * g.ldc; g.splitk_pending->ld = g.ldc;
    return ks;
  }
  const long n4 = (long)g.M * (g.N >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, ws, ks, (long)g.M * g.ldc,
                     g.bias, g.resid, g.ldr, g.out_f32, g.ldc, g.M, g.N);
  return ks;
}

int g_use_gemm256 = 1;  // 0: 128x128 only; 1: the 256x256 ring kernels of gemm256.hip (see launch_gemm256); 3 / 5 pin one of them

int launch_gemm(int prec, const GemmArgs& g_in, hipStream_t st) {
  GemmArgs g = g_in;
  g.f16 = prec == PREC_F16;
  if (g.splitk_pending) *g.splitk_pending = SplitkPending();
  if (g.M <= 0) return 0;
  const int kpt = prec_is_half(prec) ? Mma<bf16_t>::KPT : Mma<float>::KPT;  // split_t: 32 like float
  if (g.K % kpt != 0 || g.N <= 0) {
    snprintf(g_err, sizeof(g_err), "gemm: K=%d must be a multiple of %d", g.K, kpt);
    return 1;
  }
  if (g.x16) {  // 2-byte residual stream: half-precision engines, vectorisable shapes, no activation-typed second output
    const bool ok = prec_is_half(prec) && g.out_f32 && !g.out_act && g.N % 8 == 0 && g.ldc % 8 == 0 && (!g.resid || g.ldr % 8 == 0);
    if (!ok || (g.row_part && g.N % 32)) { snprintf(g_err, sizeof(g_err), "gemm: x16 (fp16 residual stream) needs a half-precision engine and N, ldc, ldr %% 8 == 0 (N %% 32 with row_part)"); return 1; }
    if (gemm_wreg_resid_eligible(g)) return launch_gemm_wreg_resid(g, st);
  }
  if (g.ln_stat && !(prec_is_half(prec) && gemm_wreg_eligible(g))) {
    snprintf(g_err, sizeof(g_err), "gemm: the folded-LayerNorm form (ln_stat) is served by the weight-stationary K = 512 kernel only");
    return 1;
  }
  if (prec_is_half(prec) && gemm_wreg_eligible(g)) return launch_gemm_wreg(g, st);
  if (prec_is_half(prec) && g_use_gemm256 && gemm256_eligible(g)) return launch_gemm256(g, st);
  if (prec == PREC_F16X3 && gemm256s_eligible(g)) return launch_gemm256s(g, st);
  if (prec == PREC_F16X3 && launch_skinny(g, st)) {
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (!g.resid || g.ldr % 4 == 0);
  int rc = 0;
  if (prec == PREC_F16X3 && try_splitk<split_t>(g, tiles_m, tiles_n, st, &rc)) {
    if (rc) return rc;
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (prec == PREC_BF16) launch_t<bf16_t>(g, tiles_m, tiles_n, vec, st);
  else if (prec == PREC_F16) launch_t<f16_t>(g, tiles_m, tiles_n, vec, st);
  else if (prec == PREC_F16X3) launch_t<split_t>(g, tiles_m, tiles_n, vec, st);
  else launch_t<float>(g, tiles_m, tiles_n, vec, st);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

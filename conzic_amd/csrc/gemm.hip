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
#include "kernels.h"

namespace czc {

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per tile row
constexpr int TILE_BYTES = BM * ROWB;           // 16 KiB (A) ; B identical
constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // A + B

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KPT = ROWB / 2;  // K elements per tile
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                0, 0);
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

template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
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
  const int m0 = tm * BM, n0 = tn * BN;

  const unsigned char* Ab = (const unsigned char*)g.A;
  const unsigned char* Wb = (const unsigned char*)g.W;
  const long lda_b = (long)g.lda * sizeof(T), ldw_b = (long)g.ldw * sizeof(T);

  // staging map: thread t moves chunk (t&7) of rows (t>>3) + 32*i, i = 0..3, of both tiles
  const int sc = tid & 7, sr = tid >> 3;
  const unsigned char* ap[4];
  const unsigned char* wp[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = sr + 32 * i;
    int ra = m0 + r; ra = ra < g.M ? ra : g.M - 1;
    int rw = n0 + r; rw = rw < g.N ? rw : g.N - 1;
    ap[i] = Ab + (long)ra * lda_b + sc * 16;
    wp[i] = Wb + (long)rw * ldw_b + sc * 16;
    soff[i] = swz(r, sc);
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / Mma<T>::KPT;
  uint4 ra[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *(const uint4*)(ap[i]);
    rw[i] = *(const uint4*)(wp[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *(uint4*)(smem + soff[i]) = ra[i];
    *(uint4*)(smem + TILE_BYTES + soff[i]) = rw[i];
  }
  __syncthreads();

  const int arow = wm * 64 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  const int half = lane >> 5;

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* sA = smem + (kt & 1) * STAGE_BYTES;
    const unsigned char* sB = sA + TILE_BYTES;
    if (kt + 1 < nk) {
      const long ko = (long)(kt + 1) * ROWB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *(const uint4*)(ap[i] + ko);
        rw[i] = *(const uint4*)(wp[i] + ko);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = 2 * ks + half;
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *(const uint4*)(sA + swz(arow + 32 * i, ch));
        b[i] = *(const uint4*)(sB + swz(brow + 32 * i, ch));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      unsigned char* dA = smem + ((kt + 1) & 1) * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(uint4*)(dA + soff[i]) = ra[i];
        *(uint4*)(dA + TILE_BYTES + soff[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  T* oa = (T*)g.out_act;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.act == ACT_QUICK_GELU) v = v / (1.0f + expf(-1.702f * v));
        else if (g.act == ACT_GELU_ERF) v = gelu_erf(v);
        if (g.resid) v += g.resid[(long)row * g.ldr + col];
        if (g.out_f32) g.out_f32[(long)row * g.ldc + col] = v;
        if (oa) Act<T>::st(oa + (long)row * g.ldc + col, v);
      }
    }
  }
}

int launch_gemm(int prec, const GemmArgs& g, hipStream_t st) {
  if (g.M <= 0) return 0;
  const int kpt = prec == PREC_BF16 ? Mma<bf16_t>::KPT : Mma<float>::KPT;
  if (g.K % kpt != 0 || g.N <= 0) {
    snprintf(g_err, sizeof(g_err), "gemm: K=%d must be a multiple of %d", g.K, kpt);
    return 1;
  }
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  dim3 grid(tiles_m * tiles_n), block(256);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(gemm_kernel<bf16_t>, grid, block, 0, st, g, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL(gemm_kernel<float>, grid, block, 0, st, g, tiles_m, tiles_n);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

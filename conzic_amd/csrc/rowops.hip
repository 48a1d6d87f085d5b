// Row-wise wavefront kernels: LayerNorm, embedding gathers, im2col, small index helpers.
// All are HBM-bound streaming kernels: one 64-lane wave per row, float4 accesses, shuffle
// reductions (no LDS), fp32 statistics regardless of the engine precision.
#include "kernels.h"

namespace czc {

constexpr int LN_MAXV = 4;  // float4 per lane -> H <= 1024

// one wave per row; returns this lane's normalised values in v[]
template <int NV>
__device__ __forceinline__ void ln_row(float4 (&v)[NV], int H, int lane, const float* gamma, const float* beta,
                                       float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const float4 gm = *(const float4*)(gamma + c);
      const float4 bt = *(const float4*)(beta + c);
      v[i].x = (v[i].x - mean) * rstd * gm.x + bt.x;
      v[i].y = (v[i].y - mean) * rstd * gm.y + bt.y;
      v[i].z = (v[i].z - mean) * rstd * gm.z + bt.z;
      v[i].w = (v[i].w - mean) * rstd * gm.w + bt.w;
    }
  }
}

// sum over the wave in the association order of gemm_rowln_kernel's epilogue (gemm256.hip): lane = 8*w + r sums its
// neighbours r^1, r^2, r^4 first (the 64-column partial of "wave" w there), then w^1, w^2, w^4
// LEAN: partners 1 .. 16 through ds_swizzle's bit-mask mode (no address registers; same partners in the same order, so the
// same bits): layernorm512_kernel then needs 30 VGPRs instead of 35, i.e. one 32-register allocation step -- a wave of it
// fits beside the two 240-register waves per SIMD of the weight-stationary GEMM (2 x 240 + 32 = 512), so with two image
// sub-batches on two streams one stream's LayerNorm pass can run on the CUs the other stream's qkv / fc1 GEMM occupies
// instead of waiting for its work-groups to leave.
template <bool LEAN>
__device__ __forceinline__ float wave_sum_rowln_order(float v) {
  if (LEAN) {
#define CZC_SWZ_XOR(M_) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1f | ((M_) << 10)))
    v += CZC_SWZ_XOR(1);
    v += CZC_SWZ_XOR(2);
    v += CZC_SWZ_XOR(4);
    v += CZC_SWZ_XOR(8);
    v += CZC_SWZ_XOR(16);
#undef CZC_SWZ_XOR
    v += __shfl_xor(v, 32, 64);
  } else {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  }
  return v;
}

// H = 512 rows of the half-precision CLIP-text tower: the SAME arithmetic, operation for operation, as the LayerNorm
// that gemm_rowln_kernel finishes in its epilogue (lane 8w + r holds columns 64w + 4r .. +3 and 64w + 32 + 4r .. +3;
// exact two-pass statistics; identical reduction tree), so that a row normalised by this kernel (few packed rows:
// the out-projection runs on the tiled GEMM) and by the full-row GEMM (many rows) come out bit-identical -- the
// engine's results must not depend on the batch size through the choice between the two.
template <typename T, bool LEAN>
__global__ __launch_bounds__(256) void layernorm512_kernel(const float* x, const int* row_idx, const float* gamma,
                                                           const float* beta, float eps, int M, T* y_act, float* y_f32) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long src = row_idx ? row_idx[m] : m;
  const float* xr = x + src * 512L;
  const int c0 = (lane >> 3) * 64 + (lane & 7) * 4;
  const float4 u = *(const float4*)(xr + c0), w = *(const float4*)(xr + c0 + 32);
  const float mean = wave_sum_rowln_order<LEAN>(((u.x + u.y) + (u.z + u.w)) + ((w.x + w.y) + (w.z + w.w))) / 512.0f;
  float q = 0.f;
  {
    const float a = u.x - mean, b = u.y - mean, c = u.z - mean, d = u.w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  {
    const float a = w.x - mean, b = w.y - mean, c = w.z - mean, d = w.w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum_rowln_order<LEAN>(q) / 512.0f + eps);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + 32 * j;
    const float4 v = j ? w : u;
    const float4 gm = *(const float4*)(gamma + c);
    const float4 bt = *(const float4*)(beta + c);
    const float ox = (v.x - mean) * rstd * gm.x + bt.x, oy = (v.y - mean) * rstd * gm.y + bt.y;
    const float oz = (v.z - mean) * rstd * gm.z + bt.z, ow = (v.w - mean) * rstd * gm.w + bt.w;
    if (y_f32) *(float4*)(y_f32 + (long)m * 512 + c) = make_float4(ox, oy, oz, ow);
    if (y_act) Act<T>::st4(y_act, (long)m * 512 + c, ox, oy, oz, ow);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const int* row_idx, const float* gamma,
                                                        const float* beta, float eps, int M, int H, T* y_act,
                                                        float* y_f32) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long src = row_idx ? row_idx[m] : m;
  const float* xr = x + src * (long)H;
  float4 v[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = c < H ? *(const float4*)(xr + c) : make_float4(0, 0, 0, 0);
  }
  ln_row<LN_MAXV>(v, H, lane, gamma, beta, eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      if (y_f32) *(float4*)(y_f32 + (long)m * H + c) = v[i];
      if (y_act) Act<T>::st4(y_act, (long)m * H + c, v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
}

// H = 512 rows of a 2-BYTE residual stream (round 5: the bf16 engine's CLIP-text tower keeps x as fp16 rows): a lane owns 8
// consecutive columns -- one 16-byte load, one 16-byte store -- exact two-pass statistics in fp32, partners 1 .. 16 through
// ds_swizzle (no address registers: the wave still fits beside the weight-stationary GEMM's two 240-register waves per SIMD).
// The one LayerNorm of this residual type at every row count, so there is no second kernel to agree with.
template <typename T>
__global__ __launch_bounds__(256) void layernorm512_x16_kernel(const f16_t* x, const int* row_idx, const float* gamma,
                                                               const float* beta, float eps, int M, T* y_act) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long src = row_idx ? row_idx[m] : m;
  const uint4 raw = *(const uint4*)(x + src * 512L + lane * 8);
  float v[8];
  const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[e] & 0xffffu));
    v[2 * e + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[e] >> 16));
  }
  const float mean = wave_sum_rowln_order<true>(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) / 512.0f;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float a = v[e] - mean, b = v[e + 1] - mean;
    q += a * a + b * b;
  }
  const float rstd = rsqrtf(wave_sum_rowln_order<true>(q) / 512.0f + eps);
  const float4 g0 = *(const float4*)(gamma + lane * 8), g1 = *(const float4*)(gamma + lane * 8 + 4);
  const float4 b0 = *(const float4*)(beta + lane * 8), b1 = *(const float4*)(beta + lane * 8 + 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  uint4 o;
  unsigned* op = (unsigned*)&o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    op[e] = Half<T>::pack2((v[2 * e] - mean) * rstd * gm[2 * e] + bt[2 * e], (v[2 * e + 1] - mean) * rstd * gm[2 * e + 1] + bt[2 * e + 1]);
  *(uint4*)((unsigned short*)y_act + (long)m * 512 + lane * 8) = o;
}

int launch_layernorm_x16(int prec, const void* x16, const int* row_idx, const float* gamma, const float* beta, float eps, int M,
                         int H, void* y_act, hipStream_t st) {
  if (M <= 0) return 0;
  if (H != 512 || !prec_is_half(prec) || !y_act) {
    snprintf(g_err, sizeof(g_err), "layernorm_x16: 512-wide rows of a half-precision engine only (H=%d)", H);
    return 1;
  }
  dim3 grid(cdiv(M, 4)), block(256);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(layernorm512_x16_kernel<bf16_t>, grid, block, 0, st, (const f16_t*)x16, row_idx, gamma, beta, eps, M, (bf16_t*)y_act);
  else
    hipLaunchKernelGGL(layernorm512_x16_kernel<f16_t>, grid, block, 0, st, (const f16_t*)x16, row_idx, gamma, beta, eps, M, (f16_t*)y_act);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// LayerNorm folded into the consumer GEMM (round 5): (mean, rstd) per row from the producers' partials -- row_part[b * ld + m] =
// (sum, sum of squares) of row m's stored fp16 values over 32-column block b -- summed over the blocks in order.  One kernel
// for every row count, so the statistics do not depend on the batch.  var = E[x^2] - mean^2 in fp32 (clamped at 0): the rows of
// these towers have |mean| of the order of their spread, so the subtraction costs a few ulps, not digits.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* part, long ld, int nblk, int M, float eps, float2* stat) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float2 a = part[m];
  for (int b = 1; b < nblk; ++b) {
    const float2 p = part[b * ld + m];
    a.x = __fadd_rn(a.x, p.x);
    a.y = __fadd_rn(a.y, p.y);
  }
  const float n = (float)(nblk * 32);
  const float mean = a.x / n;
  const float var = fmaxf(__fsub_rn(a.y / n, __fmul_rn(mean, mean)), 0.f);
  stat[m] = make_float2(mean, rsqrtf(var + eps));
}

int launch_ln_finalize(const float* part, long part_ld, int nblk, int M, float eps, float* stat, hipStream_t st) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, (const float2*)part, part_ld, nblk, M, eps, (float2*)stat);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// One-time weight preparation of the fold.  LN(x) . W^T + b = rstd * ((x - mean) . W'^T) + b' with W' = W * diag(gamma), b' = b +
// W . beta, and (x - mean 1) . W'^T = x . W"^T for W"[j,:] = W'[j,:] - rowmean(W'[j,:]): centring the WEIGHT rows once removes
// the mean from every product, so the consumer's epilogue is rstd * acc + b' and it reads neither the row mean nor a column
// sum (the first form subtracted mean * colsum(W') there: four more 16-byte LDS reads and eight FMAs per lane and block in a
// kernel whose LDS is the busiest unit: +5..9 % on the fc1 layer).  What the centring must not leave behind is the rounding:
// the stored fp16 row sums to eps_j != 0 and the product carries mean * eps_j.  So the row is rounded WITH ITS SUM IN VIEW:
// after round-to-nearest the row sum r (exact, in fp64) is driven towards zero one fp16 step at a time, each step the single
// entry whose neighbour in the right direction leaves the smallest |r| -- entries near zero have the finest steps, so a few
// dozen steps bring |r| from ~6 ulp of the typical entry down to the finest step of the row (< 1e-7 for these weights; an
// entry moves by at most a few of its own ulps).  One wave per output row; K = 512: eight entries per lane.
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* W, const float* gamma, const float* beta, const float* b, int N, int K,
                                                      f16_t* Wf, float* rowsum, float* bf) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  constexpr int PER = 8;  // K <= 512
  double wg[PER];
  double sum = 0.0;
  float wb = 0.f;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int k = lane + 64 * e;
    wg[e] = 0.0;
    if (k < K) {
      const float w = W[(long)j * K + k];
      wg[e] = (double)w * (double)gamma[k];
      wb += w * beta[k];
    }
    sum += wg[e];
  }
  auto wave_sum_f64 = [](double v) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  const double mean = wave_sum_f64(sum) / K;
  wb = wave_sum(wb);
  unsigned short h[PER];
  double r = 0.0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    h[e] = lane + 64 * e < K ? Half<f16_t>::from_f32((float)(wg[e] - mean)) : (unsigned short)0;
    r += (double)Half<f16_t>::to_f32(h[e]);
  }
  r = wave_sum_f64(r);
  for (int it = 0; it < 96; ++it) {
    // this lane's best single step: the neighbour (one step of the magnitude, normal numbers only) that leaves the smallest |r|
    double best = fabs(r), best_err = 0.0;
    int be = -1;
    unsigned short bh = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      if (lane + 64 * e >= K) continue;
      const unsigned short cur = h[e];
      const double v = (double)Half<f16_t>::to_f32(cur);
      for (int dir = -1; dir <= 1; dir += 2) {
        const int mag = (cur & 0x7fff) + dir;
        if (mag < 0x0400 || mag >= 0x7c00) continue;  // stay inside the normal numbers
        const unsigned short cand = (unsigned short)((cur & 0x8000) | mag);
        const double cv = (double)Half<f16_t>::to_f32(cand);
        const double rr = fabs(r + (cv - v)), err = fabs(cv - (wg[e] - mean));
        // equal |r| (the many entries of one binade): the entry that ends up closest to its exact value, so the steps spread
        if (rr < best || (be >= 0 && rr == best && err < best_err)) { best = rr; best_err = err; be = e; bh = cand; }
      }
    }
    double wbest = best;
    for (int o = 32; o; o >>= 1) wbest = fmin(wbest, __shfl_xor(wbest, o, 64));
    if (!(wbest < fabs(r))) break;  // no single step improves the sum
    const bool tied = be >= 0 && best == wbest;
    double werr = tied ? best_err : 1e300;
    for (int o = 32; o; o >>= 1) werr = fmin(werr, __shfl_xor(werr, o, 64));
    const unsigned long long who = __ballot(tied && best_err == werr);
    const int winner = __ffsll((long long)who) - 1;
    double delta = 0.0;
    if (lane == winner) {
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (e == be) { delta = (double)Half<f16_t>::to_f32(bh) - (double)Half<f16_t>::to_f32(h[e]); h[e] = bh; }
    }
    r += __shfl(delta, winner, 64);
  }
#pragma unroll
  for (int e = 0; e < PER; ++e)
    if (lane + 64 * e < K) Wf[(long)j * K + lane + 64 * e].v = h[e];
  if (lane == 0) {
    if (rowsum) rowsum[j] = (float)r;
    bf[j] = (b ? b[j] : 0.f) + wb;
  }
}

// rowsum (optional, [N]): what is left of each stored row's sum (test output)
int launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* b, int N, int K, void* Wf16, float* rowsum, float* bf,
                   hipStream_t st) {
  if (K > 512) { snprintf(g_err, sizeof(g_err), "fold_ln: K = %d > 512", K); return 1; }
  hipLaunchKernelGGL(fold_ln_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, W, gamma, beta, b, N, K, (f16_t*)Wf16, rowsum, bf);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int g_ln_lean = 1;  // test option ln_lean = 0: the 35-VGPR form of layernorm512_kernel (shuffles through ds_bpermute)

int launch_layernorm(int prec, const float* x, const int* row_idx, const float* gamma, const float* beta, float eps,
                     int M, int H, void* y_act, float* y_f32, hipStream_t st) {
  if (M <= 0) return 0;
  if (H % 4 || H > LN_MAXV * 256) {
    snprintf(g_err, sizeof(g_err), "layernorm: unsupported H=%d", H);
    return 1;
  }
  dim3 grid(cdiv(M, 4)), block(256);
  if (H == 512 && prec == PREC_BF16 && g_ln_lean)
    hipLaunchKernelGGL((layernorm512_kernel<bf16_t, true>), grid, block, 0, st, x, row_idx, gamma, beta, eps, M, (bf16_t*)y_act, y_f32);
  else if (H == 512 && prec == PREC_F16 && g_ln_lean)
    hipLaunchKernelGGL((layernorm512_kernel<f16_t, true>), grid, block, 0, st, x, row_idx, gamma, beta, eps, M, (f16_t*)y_act, y_f32);
  else if (H == 512 && prec == PREC_BF16)
    hipLaunchKernelGGL((layernorm512_kernel<bf16_t, false>), grid, block, 0, st, x, row_idx, gamma, beta, eps, M, (bf16_t*)y_act, y_f32);
  else if (H == 512 && prec == PREC_F16)
    hipLaunchKernelGGL((layernorm512_kernel<f16_t, false>), grid, block, 0, st, x, row_idx, gamma, beta, eps, M, (f16_t*)y_act, y_f32);
  else if (prec == PREC_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, st, x, row_idx, gamma, beta, eps, M, H,
                       (bf16_t*)y_act, y_f32);
  else if (prec == PREC_F16)
    hipLaunchKernelGGL(layernorm_kernel<f16_t>, grid, block, 0, st, x, row_idx, gamma, beta, eps, M, H,
                       (f16_t*)y_act, y_f32);
  else if (prec == PREC_F16X3)
    hipLaunchKernelGGL(layernorm_kernel<split_t>, grid, block, 0, st, x, row_idx, gamma, beta, eps, M, H,
                       (split_t*)y_act, y_f32);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, st, x, row_idx, gamma, beta, eps, M, H, (float*)y_act,
                       y_f32);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// LayerNorm of rows that still are split-K slabs (GemmArgs::splitk_pending; BERT fc2 -> LayerNorm): the row is the sum of the
// slice slabs in slice order, + bias, + residual -- splitk_reduce_kernel's arithmetic, element for element -- and is normalised
// in the same pass (layernorm_kernel's ln_row): one launch and no fp32 round trip of the row through HBM instead of two launches.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_splitk_kernel(const float* slabs, int nslab, long slab_stride, int ld, const float* bias,
                                                               const float* resid, int ldr, const float* gamma, const float* beta,
                                                               float eps, int M, int H, T* y_act, float* y_f32) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float4 v[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float4 a = *(const float4*)(slabs + (long)m * ld + c);
      for (int z = 1; z < nslab; ++z) {
        const float4 p = *(const float4*)(slabs + z * slab_stride + (long)m * ld + c);
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      }
      if (bias) { const float4 b = *(const float4*)(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      if (resid) { const float4 r = *(const float4*)(resid + (long)m * ldr + c); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
      v[i] = a;
    } else {
      v[i] = make_float4(0, 0, 0, 0);
    }
  }
  ln_row<LN_MAXV>(v, H, lane, gamma, beta, eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      if (y_f32) *(float4*)(y_f32 + (long)m * H + c) = v[i];
      if (y_act) Act<T>::st4(y_act, (long)m * H + c, v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
}

int launch_layernorm_splitk(int prec, const SplitkPending& sk, const float* bias, const float* resid, int ldr, const float* gamma,
                            const float* beta, float eps, int M, int H, void* y_act, float* y_f32, hipStream_t st) {
  if (M <= 0) return 0;
  if (H % 4 || H > LN_MAXV * 256 || sk.nslab <= 0 || !sk.slabs || sk.ld % 4 || (resid && ldr % 4) || (prec != PREC_F16X3 && prec != PREC_F32)) {
    snprintf(g_err, sizeof(g_err), "layernorm_splitk: unsupported shape / precision (H=%d, slabs=%d)", H, sk.nslab);
    return 1;
  }
  dim3 grid(cdiv(M, 4)), block(256);
  if (prec == PREC_F16X3)
    hipLaunchKernelGGL(layernorm_splitk_kernel<split_t>, grid, block, 0, st, sk.slabs, sk.nslab, sk.slab_stride, sk.ld, bias, resid, ldr, gamma,
                       beta, eps, M, H, (split_t*)y_act, y_f32);
  else
    hipLaunchKernelGGL(layernorm_splitk_kernel<float>, grid, block, 0, st, sk.slabs, sk.nslab, sk.slab_stride, sk.ld, bias, resid, ldr, gamma,
                       beta, eps, M, H, (float*)y_act, y_f32);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- BERT embeddings + LN ------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int* ids, int M, int T_, int H, const float* word,
                                                         const float* pos, const float* type0, const float* gamma,
                                                         const float* beta, float eps, T* y_act, float* y_f32) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int id = ids[m];
  const int p = m % T_;
  const float* wr = word + (long)id * H;
  const float* pr = pos + (long)p * H;
  float4 v[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const float4 a = *(const float4*)(wr + c), b = *(const float4*)(type0 + c), d = *(const float4*)(pr + c);
      // HF order: inputs_embeds + token_type_embeddings, then + position_embeddings
      v[i] = make_float4((a.x + b.x) + d.x, (a.y + b.y) + d.y, (a.z + b.z) + d.z, (a.w + b.w) + d.w);
    } else {
      v[i] = make_float4(0, 0, 0, 0);
    }
  }
  ln_row<LN_MAXV>(v, H, lane, gamma, beta, eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      if (y_f32) *(float4*)(y_f32 + (long)m * H + c) = v[i];
      if (y_act) Act<T>::st4(y_act, (long)m * H + c, v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
}

int launch_bert_embed(int prec, const int* ids, int B, int T_, int H, const float* word, const float* pos,
                      const float* type0, const float* gamma, const float* beta, float eps, void* y_act, float* y_f32,
                      hipStream_t st) {
  const int M = B * T_;
  if (M <= 0) return 0;
  dim3 grid(cdiv(M, 4)), block(256);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(bert_embed_kernel<bf16_t>, grid, block, 0, st, ids, M, T_, H, word, pos, type0, gamma, beta, eps,
                       (bf16_t*)y_act, y_f32);
  else if (prec == PREC_F16)
    hipLaunchKernelGGL(bert_embed_kernel<f16_t>, grid, block, 0, st, ids, M, T_, H, word, pos, type0, gamma, beta, eps,
                       (f16_t*)y_act, y_f32);
  else if (prec == PREC_F16X3)
    hipLaunchKernelGGL(bert_embed_kernel<split_t>, grid, block, 0, st, ids, M, T_, H, word, pos, type0, gamma, beta, eps,
                       (split_t*)y_act, y_f32);
  else
    hipLaunchKernelGGL(bert_embed_kernel<float>, grid, block, 0, st, ids, M, T_, H, word, pos, type0, gamma, beta, eps,
                       (float*)y_act, y_f32);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- CLIP text embeddings on packed segments ----------------------------------------------------
// one wave per (segment, own row).  The K candidates of an image share every token id but one,
// so the token-embedding rows they gather are the same cache lines (L2-resident after first touch).
__global__ __launch_bounds__(256) void clip_embed_kernel(const int* ids, int ids_stride, const int* seg_src,
                                                         const int* seg_pos0, const int* own_off, const int* own_len,
                                                         int n_seg, int max_len, int H, const float* tok,
                                                         const float* pos, float* x, int x16, float* stat, float eps) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int s = (int)(w / max_len), i = (int)(w % max_len);
  if (s >= n_seg || i >= own_len[s]) return;
  const int p = seg_pos0[s] + i;
  const int id = ids[(long)seg_src[s] * ids_stride + p];
  const float* tr = tok + (long)id * H;
  const float* pr = pos + (long)p * H;
  if (x16) {  // 2-byte residual stream: the same sums, rounded once to fp16
    f16_t* xh = (f16_t*)x + ((long)own_off[s] + i) * H;
    float sm = 0.f;
    float4 keep[LN_MAXV];  // compile-time indices only (H <= 1024)
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
      const int c = lane * 4 + t * 256;
      keep[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < H) {
        const float4 a = *(const float4*)(tr + c), b = *(const float4*)(pr + c);
        const uint2 pk = make_uint2(pack2_f16(a.x + b.x, a.y + b.y), pack2_f16(a.z + b.z, a.w + b.w));
        *(uint2*)(xh + c) = pk;
        keep[t] = make_float4(f16lo(pk.x), f16hi(pk.x), f16lo(pk.y), f16hi(pk.y));
        sm += (keep[t].x + keep[t].y) + (keep[t].z + keep[t].w);
      }
    }
    if (stat) {  // (mean, rstd) of the STORED row: layer 0's LayerNorm, folded into its q/k/v GEMM (two-pass, like the LayerNorm kernels)
      const float mean = wave_sum(sm) / (float)H;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < LN_MAXV; ++t) {
        if (lane * 4 + t * 256 < H) {
          const float a = keep[t].x - mean, b = keep[t].y - mean, c2 = keep[t].z - mean, d = keep[t].w - mean;
          q += (a * a + b * b) + (c2 * c2 + d * d);
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
      if (lane == 0) *(float2*)(stat + ((long)own_off[s] + i) * 2) = make_float2(mean, rstd);
    }
    return;
  }
  float* xr = x + ((long)own_off[s] + i) * H;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = *(const float4*)(tr + c), b = *(const float4*)(pr + c);
    *(float4*)(xr + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

int launch_clip_embed(const int* ids, int ids_stride, const int* seg_src, const int* seg_pos0, const int* own_off,
                      const int* own_len, int n_seg, int max_len, int H, const float* tok, const float* pos, float* x,
                      hipStream_t st, int x16, float* stat, float eps) {
  if (n_seg <= 0) return 0;
  dim3 grid(cdiv((long)n_seg * max_len, 4)), block(256);
  hipLaunchKernelGGL(clip_embed_kernel, grid, block, 0, st, ids, ids_stride, seg_src, seg_pos0, own_off, own_len, n_seg,
                     max_len, H, tok, pos, x, x16, stat, eps);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- vision helpers ----------------------------------------------------------------------------
// patches[b*P + py*G + px][c*p*p + iy*p + ix] = pixels[b][c][py*p+iy][px*p+ix]
// (conv2d stride=kernel=p, no bias == GEMM with the [hidden, 3*p*p] flattened conv weight)
template <typename T>
__global__ void im2col_kernel(const float* pix, int B, int S, int p, T* out) {
  const int G = S / p, P = G * G, Kc = 3 * p * p;
  const long total = (long)B * P * Kc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kc);
    const long bp = i / Kc;
    const int pp = (int)(bp % P), b = (int)(bp / P);
    const int c = k / (p * p), iy = (k / p) % p, ix = k % p;
    const int py = pp / G, px = pp % G;
    const float v = pix[(((long)b * 3 + c) * S + (py * p + iy)) * S + (px * p + ix)];
    Act<T>::st(out, i, v);
  }
}

int launch_im2col(int prec, const float* pixels, int B, int S, int p, void* out, hipStream_t st) {
  const long total = (long)B * (S / p) * (S / p) * 3 * p * p;
  dim3 grid((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), block(256);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(im2col_kernel<bf16_t>, grid, block, 0, st, pixels, B, S, p, (bf16_t*)out);
  else if (prec == PREC_F16)
    hipLaunchKernelGGL(im2col_kernel<f16_t>, grid, block, 0, st, pixels, B, S, p, (f16_t*)out);
  else if (prec == PREC_F16X3)
    hipLaunchKernelGGL(im2col_kernel<split_t>, grid, block, 0, st, pixels, B, S, p, (split_t*)out);
  else
    hipLaunchKernelGGL(im2col_kernel<float>, grid, block, 0, st, pixels, B, S, p, (float*)out);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// x[b, 0] = cls + pos[0];  x[b, 1+j] = patch_out[b*P+j] + pos[1+j]   (HF:clip/modeling_clip.py:202-218)
__global__ void vision_assemble_kernel(const float* patch_out, int B, int P, int H, const float* cls, const float* pos,
                                       float* x) {
  const long total = (long)B * (P + 1) * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % H);
    const long bt = i / H;
    const int t = (int)(bt % (P + 1)), b = (int)(bt / (P + 1));
    const float e = t == 0 ? cls[c] : patch_out[((long)b * P + (t - 1)) * H + c];
    x[i] = e + pos[(long)t * H + c];
  }
}

int launch_vision_assemble(const float* patch_out, int B, int P, int H, const float* cls, const float* pos, float* x,
                           hipStream_t st) {
  const long total = (long)B * (P + 1) * H;
  dim3 grid((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), block(256);
  hipLaunchKernelGGL(vision_assemble_kernel, grid, block, 0, st, patch_out, B, P, H, cls, pos, x);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
__global__ void convert_kernel(const float* src, T* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Act<T>::st(dst, i, src[i]);
}

int launch_convert(int prec, const float* src, void* dst, long n, hipStream_t st) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384)), block(256);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(convert_kernel<bf16_t>, grid, block, 0, st, src, (bf16_t*)dst, n);
  else if (prec == PREC_F16)
    hipLaunchKernelGGL(convert_kernel<f16_t>, grid, block, 0, st, src, (f16_t*)dst, n);
  else if (prec == PREC_F16X3)
    hipLaunchKernelGGL(convert_kernel<split_t>, grid, block, 0, st, src, (split_t*)dst, n);
  else
    hipLaunchKernelGGL(convert_kernel<float>, grid, block, 0, st, src, (float*)dst, n);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
__global__ void act_to_f32_kernel(const T* src, float* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = Act<T>::ld(src, i);
}

int launch_act_to_f32(int prec, const void* src, float* dst, long n, hipStream_t st) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384)), block(256);
  if (prec == PREC_BF16) hipLaunchKernelGGL(act_to_f32_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)src, dst, n);
  else if (prec == PREC_F16) hipLaunchKernelGGL(act_to_f32_kernel<f16_t>, grid, block, 0, st, (const f16_t*)src, dst, n);
  else if (prec == PREC_F16X3) hipLaunchKernelGGL(act_to_f32_kernel<split_t>, grid, block, 0, st, (const split_t*)src, dst, n);
  else hipLaunchKernelGGL(act_to_f32_kernel<float>, grid, block, 0, st, (const float*)src, dst, n);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void gather_rows_kernel(const float* src, const int* idx, int M, int H, float* dst) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* s = src + (long)idx[m] * H;
  float* d = dst + (long)m * H;
  for (int c = lane * 4; c < H; c += 256) *(float4*)(d + c) = *(const float4*)(s + c);
}

// generic row gather on raw bytes (row_bytes % 16 == 0): dst[m] = src[idx[m]]
__global__ void gather_rows_bytes_kernel(const unsigned char* src, const int* idx, int M, int row_bytes,
                                         unsigned char* dst) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const unsigned char* s = src + (long)idx[m] * row_bytes;
  unsigned char* d = dst + (long)m * row_bytes;
  for (int c = lane * 16; c < row_bytes; c += 1024) *(uint4*)(d + c) = *(const uint4*)(s + c);
}

int launch_gather_rows_bytes(const void* src, const int* idx, int M, int row_bytes, void* dst, hipStream_t st) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const unsigned char*)src, idx, M,
                     row_bytes, (unsigned char*)dst);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_gather_rows_f32(const float* src, const int* idx, int M, int H, float* dst, hipStream_t st) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, src, idx, M, H, dst);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void make_row_index_kernel(int* idx, int B, int T_, int gen_idx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) idx[b] = b * T_ + gen_idx;
}
int launch_make_row_index(int* idx, int B, int T_, int gen_idx, hipStream_t st) {
  hipLaunchKernelGGL(make_row_index_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, idx, B, T_, gen_idx);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void eos_index_kernel(const int* off, const int* len, int n, int* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = off[i] + len[i] - 1;
}
int launch_eos_index(const int* off, const int* len, int n, int* idx, hipStream_t st) {
  hipLaunchKernelGGL(eos_index_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, off, len, n, idx);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void l2_normalize_kernel(const float* x, int M, int D, float* y) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (long)m * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
  const float nrm = sqrtf(wave_sum(s));
  for (int c = lane; c < D; c += 64) y[(long)m * D + c] = xr[c] / nrm;
}
int launch_l2_normalize(const float* x, int M, int D, float* y, hipStream_t st) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, D, y);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void mask_positions_kernel(int* inp, int B, int T_, int gen_idx, int n_mask, int mask_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * n_mask) {
    const int b = i / n_mask, j = i % n_mask;
    if (gen_idx + j < T_) inp[b * T_ + gen_idx + j] = mask_id;
  }
}
int launch_mask_positions(int* inp, int B, int T_, int gen_idx, int n_mask, int mask_id, hipStream_t st) {
  if (n_mask <= 0) return 0;
  hipLaunchKernelGGL(mask_positions_kernel, dim3(cdiv(B * n_mask, 256)), dim3(256), 0, st, inp, B, T_, gen_idx, n_mask,
                     mask_id);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void broadcast_rows_kernel(const int* row, int T_, int B, int* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T_) dst[i] = row[i % T_];
}
int launch_broadcast_rows_i32(const int* row, int T_, int B, int* dst, hipStream_t st) {
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(cdiv(B * T_, 256)), dim3(256), 0, st, row, T_, B, dst);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

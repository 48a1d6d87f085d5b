// Attention over many tiny sequences (BERT T<=64 bidirectional, CLIP text Tc<=77 causal,
// CLIP vision T=50): softmax(q k^T * scale [+causal mask]) v, head_dim 64
// (HF:bert/modeling_bert.py:111-136, HF:clip/modeling_clip.py:259-277,:309-333).
//
// v1: exact-fp32 wavefront kernel, one wave per (sequence, head).  Whole K^T/V/Q of the head sit
// in LDS as fp32 (K transposed with an odd pitch, so both the transposing store and the
// lane-per-key reads are bank-conflict free); lane j owns key j (and j+64), lanes reduce with
// shuffles; lane d owns output dim d.  FLOPs here are <0.5% of the step; the cost is HBM
// streaming of qkv.  (An MFMA 16x16x32 variant for the bf16 engine replaces it in mfma_attention.)
#include "kernels.h"

namespace czc {

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* qkv, const int* seq_off, const int* seq_len,
                                                        int fixed_T, int heads, int causal, float scale, int Tcap,
                                                        T* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int s = blockIdx.x;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const int len = seq_len ? seq_len[s] : fixed_T;
  const long row0 = seq_off ? seq_off[s] : (long)s * fixed_T;
  const int Hd = heads * 64;
  const int Tp = Tcap | 1;
  float* Kt = lds + (size_t)wave * (64 * Tp + 2 * 64 * Tcap + 128);
  float* Vs = Kt + 64 * Tp;
  float* Qs = Vs + 64 * Tcap;
  float* Ps = Qs + 64 * Tcap;  // 128 floats

  for (int t = 0; t < len; ++t) {
    const T* r = qkv + (row0 + t) * (long)(3 * Hd) + h * 64 + lane;
    Qs[t * 64 + lane] = Act<T>::ld(r);
    Kt[lane * Tp + t] = Act<T>::ld(r + Hd);
    Vs[t * 64 + lane] = Act<T>::ld(r + 2 * Hd);
  }
  __syncthreads();

  for (int i = 0; i < len; ++i) {
    const int nk = causal ? i + 1 : len;
    float s0 = 0.f, s1 = 0.f;
    const int j0 = lane, j1 = lane + 64;
    const int j0c = j0 < len ? j0 : 0, j1c = j1 < len ? j1 : 0;
    const float* qi = Qs + i * 64;
#pragma unroll 4
    for (int d = 0; d < 64; d += 4) {
      const float4 q = *(const float4*)(qi + d);
      s0 += q.x * Kt[(d + 0) * Tp + j0c];
      s0 += q.y * Kt[(d + 1) * Tp + j0c];
      s0 += q.z * Kt[(d + 2) * Tp + j0c];
      s0 += q.w * Kt[(d + 3) * Tp + j0c];
      if (len > 64) {
        s1 += q.x * Kt[(d + 0) * Tp + j1c];
        s1 += q.y * Kt[(d + 1) * Tp + j1c];
        s1 += q.z * Kt[(d + 2) * Tp + j1c];
        s1 += q.w * Kt[(d + 3) * Tp + j1c];
      }
    }
    s0 = j0 < nk ? s0 * scale : -INFINITY;
    s1 = j1 < nk ? s1 * scale : -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = j0 < nk ? expf(s0 - m) : 0.f;
    const float e1 = j1 < nk ? expf(s1 - m) : 0.f;
    const float sum = wave_sum(e0 + e1);
    Ps[j0] = e0 / sum;
    Ps[j1] = e1 / sum;
    __syncthreads();
    float o = 0.f;
    for (int j = 0; j < nk; ++j) o += Ps[j] * Vs[j * 64 + lane];
    if (live) Act<T>::st(out + (row0 + i) * (long)Hd + h * 64 + lane, o);
    __syncthreads();
  }
}

int launch_attention(int prec, const void* qkv, const int* seq_off, const int* seq_len, int fixed_T, int n_seq,
                     int max_len, int heads, int causal, float scale, void* out, hipStream_t st) {
  if (n_seq <= 0) return 0;
  if (max_len > 128 || max_len <= 0) {
    snprintf(g_err, sizeof(g_err), "attention: max_len=%d unsupported (1..128)", max_len);
    return 1;
  }
  const int Tcap = (max_len + 3) & ~3;
  const size_t per_wave = (size_t)(64 * (Tcap | 1) + 2 * 64 * Tcap + 128) * sizeof(float);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 60 * 1024) wpb >>= 1;
  if (wpb > heads) wpb = heads >= 2 ? 2 : 1;
  const size_t shmem = per_wave * wpb;
  dim3 grid(n_seq, cdiv(heads, wpb)), block(64 * wpb);
  if (prec == PREC_BF16) {
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<bf16_t>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, block, shmem, st, (const bf16_t*)qkv, seq_off, seq_len, fixed_T,
                       heads, causal, scale, Tcap, (bf16_t*)out);
  } else {
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<float>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(attention_kernel<float>, grid, block, shmem, st, (const float*)qkv, seq_off, seq_len, fixed_T,
                       heads, causal, scale, Tcap, (float*)out);
  }
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

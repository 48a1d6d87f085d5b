// Fused score-combine: one workgroup per image.
//   cos_k   = <t_k/|t_k|, i/|i|>                                   (clip/clip.py:91-95)
//   clip_k  = softmax_K(cos * exp(logit_scale)), ref_k = cos        (clip/clip.py:96-98)
//   final_k = alpha*probs_k + beta*clip_k                           (gen_utils.py:77)
//             [+ gamma*softmax_K(senti)_k + 0.1*(1 - exp(repeats_k))]   (control_gen_utils.py:59)
//   best    = first argmax_K(final); inp[b, gen_idx] = cand[b, best]; best_cos = ref[best]
//                                                                    (gen_utils.py:78-80)
// Cosines: one wave per candidate row (cosine_kernel); the rest per image.  Wavefront reductions only; lanes stride the
// 512-wide feature row with coalesced reads; softmax/argmax over K run in LDS.
#include "kernels.h"

namespace czc {

constexpr int CB_THREADS = 256;
constexpr int CB_MAXK = 1024;

__device__ __forceinline__ float blk_reduce(float v, float* red, int op) {  // op 0 sum, 1 max
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = op ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < CB_THREADS / 64; ++i) r = op ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// cos[b,k] for all B*K candidates, one wave per candidate (the per-image kernel below used to walk its K
// candidates with four waves: 13x off the HBM rate at B = 256, K = 200).  Same arithmetic, same order.
__global__ __launch_bounds__(256) void cosine_kernel(const float* text_feat, const float* img_n, int B, int K, int D, float* cos_out,
                                                     int* nonfinite) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * K) return;
  const float* t = text_feat + row * D;
  const float* img = img_n + (row / K) * D;
  float nn = 0.f;
  for (int c = lane; c < D; c += 64) nn += t[c] * t[c];
  const float nrm = sqrtf(wave_sum(nn));
  float dot = 0.f;
  for (int c = lane; c < D; c += 64) dot += (t[c] / nrm) * img[c];
  dot = wave_sum(dot);
  if (lane == 0) {
    cos_out[row] = dot;
    // a non-finite cosine means a tower overflowed (fp16 operands beyond 65504) or was fed NaN weights: reported, not hidden
    if (nonfinite && !(fabsf(dot) <= 2.0f)) atomicOr(nonfinite, 1);
  }
}

__global__ __launch_bounds__(CB_THREADS) void combine_kernel(CombineArgs a) {
  __shared__ float s_cos[CB_MAXK];
  __shared__ float s_fin[CB_MAXK];
  __shared__ float red[8];
  __shared__ int s_best;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int K = a.K;

  // cosines come from cosine_kernel through clip_ref (overwritten below with the reference's logits/scale form)
  for (int k = tid; k < K; k += CB_THREADS) s_cos[k] = a.clip_ref[(long)b * K + k];
  __syncthreads();

  // softmax over K of cos * scale
  float mx = -INFINITY;
  for (int k = tid; k < K; k += CB_THREADS) mx = fmaxf(mx, s_cos[k] * a.logit_scale_exp);
  mx = blk_reduce(mx, red, 1);
  float sm = 0.f;
  for (int k = tid; k < K; k += CB_THREADS) sm += expf(s_cos[k] * a.logit_scale_exp - mx);
  sm = blk_reduce(sm, red, 0);
  float smx = -INFINITY, ssm = 1.f;
  const float ctemp = a.use_senti == 2 ? 0.1f : 1.0f;  // POS: softmax(acc/0.1); sentiment: softmax(score/1)
  if (a.use_senti) {
    for (int k = tid; k < K; k += CB_THREADS) smx = fmaxf(smx, a.senti_raw[(long)b * K + k] / ctemp);
    smx = blk_reduce(smx, red, 1);
    float s2 = 0.f;
    for (int k = tid; k < K; k += CB_THREADS) s2 += expf(a.senti_raw[(long)b * K + k] / ctemp - smx);
    ssm = blk_reduce(s2, red, 0);
  }
  for (int k = tid; k < K; k += CB_THREADS) {
    const long o = (long)b * K + k;
    const float lg = s_cos[k] * a.logit_scale_exp;
    const float cs = expf(lg - mx) / sm;
    const float ref = lg / a.logit_scale_exp;  // the reference divides the scaled logit back
    float f = a.alpha * a.probs[o] + a.beta * cs;
    if (a.use_senti) {
      const float sp = expf(a.senti_raw[o] / ctemp - smx) / ssm;
      f = f + a.gamma * sp;
      if (a.use_senti == 1) f = f + 0.1f * (1.0f - expf(a.repeats[o]));
    }
    a.clip_score[o] = cs;
    a.clip_ref[o] = ref;
    a.final_score[o] = f;
    s_fin[k] = f;
    s_cos[k] = ref;
  }
  __syncthreads();
  // first argmax (torch.argmax returns the first maximal index)
  float bm = -INFINITY;
  for (int k = tid; k < K; k += CB_THREADS) bm = fmaxf(bm, s_fin[k]);
  bm = blk_reduce(bm, red, 1);
  if (tid == 0) s_best = K;
  __syncthreads();
  for (int k = tid; k < K; k += CB_THREADS)
    if (s_fin[k] == bm) atomicMin(&s_best, k);
  __syncthreads();
  if (tid == 0) {
    const int bi = s_best < K ? s_best : 0;  // all-NaN scores: keep candidate 0
    a.best[b] = bi;
    a.best_cos[b] = s_cos[bi];
    if (a.inp) a.inp[(long)b * a.T + a.gen_idx] = a.cand[(long)b * K + bi];
  }
}

int launch_combine(const CombineArgs& a, hipStream_t st) {
  if (a.K > CB_MAXK) {
    snprintf(g_err, sizeof(g_err), "combine: K=%d > %d", a.K, CB_MAXK);
    return 1;
  }
  hipLaunchKernelGGL(cosine_kernel, dim3((unsigned)cdiv((long)a.B * a.K, 4)), dim3(256), 0, st, a.text_feat, a.img_n, a.B, a.K, a.D,
                     a.clip_ref, a.nonfinite);
  hipLaunchKernelGGL(combine_kernel, dim3(a.B), dim3(CB_THREADS), 0, st, a);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

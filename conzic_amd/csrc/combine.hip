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
  if (a.refine_kind) {
    // screen-then-refine: candidates marked in refine_kind carry an exact (split-fp16) cosine in
    // refine_cos; the others keep the screening (single-pass fp16) cosine minus the screening tower's mean error,
    // estimated on the mass-stratified SAMPLE among the re-encoded candidates (kind 2, weight = strata it stands for)
    float ds = 0.f, dw = 0.f;
    for (int k = tid; k < K; k += CB_THREADS) {
      const int kd = a.refine_kind[(long)b * K + k];
      if ((kd & 3) == 2) {
        const float w = (float)(kd >> 2);
        ds += w * (s_cos[k] - a.refine_cos[(long)b * K + k]);
        dw += w;
      }
    }
    ds = blk_reduce(ds, red, 0);
    dw = blk_reduce(dw, red, 0);
    const float mu = dw > 0.f ? ds / dw : 0.f;
    // Guard (runtime check of the bound the selection rests on): a candidate that keeps its screening cosine carries the
    // error (d_k - mu); its fused score moves by at most theta_x * |d_k - mu|.  Every re-encoded candidate -- the sample
    // and the mass carriers, ~16 per image -- shows its own |d_k - mu|: their maximum is recorded, and an image where it
    // exceeds the budget (checkpoints whose activations the single-pass fp16 tower does not carry as well as the
    // validated ones) is counted, for the host to re-run the call on the all-split engine (conzic_amd/runtime.py)
    float dev = 0.f;
    for (int k = tid; k < K; k += CB_THREADS) {
      const int kd = a.refine_kind[(long)b * K + k] & 3;
      if (kd == 1 || kd == 2) dev = fmaxf(dev, fabsf(s_cos[k] - a.refine_cos[(long)b * K + k] - mu));
    }
    dev = blk_reduce(dev, red, 1);
    if (tid == 0 && a.nonfinite && dev == dev) {
      atomicMax(a.nonfinite + 1, __float_as_int(dev));  // non-negative floats order like their bit patterns
      if (a.refine_guard > 0.f && dev > a.refine_guard) atomicAdd(a.nonfinite + 2, 1);
    }
    __syncthreads();
    // kind 3 (a margin-gated image's winner): exact cosine for the output only, its score keeps the screening cosine
    for (int k = tid; k < K; k += CB_THREADS) {
      const int kd = a.refine_kind[(long)b * K + k] & 3;
      s_cos[k] = (kd == 1 || kd == 2) ? a.refine_cos[(long)b * K + k] : s_cos[k] - mu;
    }
    __syncthreads();
  }

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
    a.best_cos[b] = (a.refine_kind && (a.refine_kind[(long)b * K + bi] & 3) == 3) ? a.refine_cos[(long)b * K + bi] : s_cos[bi];
    if (a.inp) a.inp[(long)b * a.T + a.gen_idx] = a.cand[(long)b * K + bi];
  }
}

// Screen-then-refine, selection.  After the screening pass (combine_kernel on the single-pass fp16 cosines) pick, per
// image, the candidates whose cosine is re-encoded by the split-fp16 tower:
//   kind 1: softmax_K mass above theta (their score error scales with beta * scale * p_k), plus the two best fused scores
//           (the winner's cosine is what the caller gets back: gen_utils.py:80-81);
//   kind 2: m_samples candidates of the rest, one per equal-MASS stratum in candidate order (where the cumulative
//           softmax mass of the rest crosses (j + 1/2) / m of its total); kind = 2 | hits << 2 when one candidate
//           covers several strata.  Their exact cosines give the mass-weighted mean error of the screening tower over
//           the candidates that are NOT re-encoded, which the final combine removes.
// list[b][0..count[b]) = the chosen candidate indices in ascending order.
//
// Margin gate (czc_generate only; gate_h = exp(logit_scale) * delta > 0): a whole *_generation call exposes the winner's id
// at every step and the winner's cosine at the snapshot steps (gen_utils.py:78-81, :92) -- not the K fused scores.  If the
// screening winner w stays the winner under EVERY assignment of cosine errors |d_k - common| <= delta, the exact scores
// would pick it too and the image needs no second pass: for each challenger r take the adversarial assignment (w's logit
// down by h, r's up by h, all others up or down by h -- f_r - f_w is monotone in their common factor, so the two extremes
// bound it) and require f_r < f_w.  Such an image re-encodes nothing (need_cos = 0) or only its winner, for the cosine the
// caller reads back (kind 3: exact cosine for the OUTPUT, screening cosine in the scores, so that every score of the
// image carries the same common error).  gated[0] counts those images, gated[1] all images.
__global__ __launch_bounds__(CB_THREADS) void refine_select_kernel(const float* clip_score, const float* final_score, int K,
                                                                   float theta, int m_samples, float gate_h, float beta, int need_cos,
                                                                   int* gated, int* kind, int* list, int* count) {
  __shared__ float s_p[CB_MAXK];
  __shared__ float s_f[CB_MAXK];
  __shared__ int s_kind[CB_MAXK];
  __shared__ float red[8];
  __shared__ int s_arg;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < K; k += CB_THREADS) {
    s_p[k] = clip_score[(long)b * K + k];
    s_f[k] = final_score[(long)b * K + k];
    s_kind[k] = s_p[k] > theta ? 1 : 0;
  }
  __syncthreads();
  if (gate_h > 0.f) {
    float bm = -INFINITY;
    for (int k = tid; k < K; k += CB_THREADS) bm = fmaxf(bm, s_f[k]);
    bm = blk_reduce(bm, red, 1);
    if (tid == 0) s_arg = K;
    __syncthreads();
    for (int k = tid; k < K; k += CB_THREADS)
      if (s_f[k] == bm) atomicMin(&s_arg, k);
    __syncthreads();
    const int w = s_arg < K ? s_arg : 0;
    const float pw = s_p[w], fw = s_f[w];
    const float up = expf(gate_h), dn = expf(-gate_h);
    const float ew = pw * dn, base_w = fw - beta * pw;
    float worst = -INFINITY;  // max over challengers and both extremes of (f_r - f_w) under the adversarial errors
    for (int k = tid; k < K; k += CB_THREADS) {
      if (k == w) continue;
      const float pr = s_p[k], er = pr * up, rest = fmaxf(1.0f - pw - pr, 0.f);
      const float db = (s_f[k] - beta * pr) - base_w;
      const float d_lo = db + beta * (er - ew) / (ew + er + rest * dn);
      const float d_hi = db + beta * (er - ew) / (ew + er + rest * up);
      const float d = fmaxf(d_lo, d_hi);
      worst = fmaxf(worst, d == d ? d : INFINITY);  // a NaN score never passes the gate
    }
    worst = blk_reduce(worst, red, 1);
    const bool pass = s_arg < K && worst < 0.f && fw == fw;
    if (tid == 0) {
      atomicAdd(gated + 1, 1);
      if (pass) atomicAdd(gated, 1);
    }
    if (pass) {
      if (tid == 0) {
        int n = 0;
        if (need_cos) list[(long)b * K + n++] = w;
        count[b] = n;
      }
      for (int k = tid; k < K; k += CB_THREADS) kind[(long)b * K + k] = (need_cos && k == w) ? 3 : 0;
      return;  // uniform over the work-group
    }
    __syncthreads();
  }
  for (int round = 0; round < 2; ++round) {  // first argmax of the fused score, then the runner-up
    float bm = -INFINITY;
    for (int k = tid; k < K; k += CB_THREADS) bm = fmaxf(bm, s_f[k]);
    bm = blk_reduce(bm, red, 1);
    if (tid == 0) s_arg = K;
    __syncthreads();
    for (int k = tid; k < K; k += CB_THREADS)
      if (s_f[k] == bm) atomicMin(&s_arg, k);
    __syncthreads();
    if (tid == 0 && s_arg < K) { s_kind[s_arg] = 1; s_f[s_arg] = -INFINITY; }
    __syncthreads();
  }
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < K; ++k) if (!s_kind[k]) tot += s_p[k];
    if (m_samples > 0 && tot > 0.f) {
      float cum = 0.f;
      int j = 0;
      for (int k = 0; k < K && j < m_samples; ++k) {
        if (s_kind[k]) continue;
        cum += s_p[k];
        int hits = 0;
        while (j < m_samples && cum >= ((float)j + 0.5f) * tot / (float)m_samples) { ++hits; ++j; }
        if (hits) s_kind[k] = 2 | (hits << 2);
      }
    }
    int n = 0;
    for (int k = 0; k < K; ++k) {
      kind[(long)b * K + k] = s_kind[k];
      if (s_kind[k]) list[(long)b * K + n++] = k;
    }
    count[b] = n;
  }
}

int launch_refine_select(const float* clip_score, const float* final_score, int B, int K, float theta, int m_samples, float gate_h,
                         float beta, int need_cos, int* gated, int* kind, int* list, int* count, hipStream_t st) {
  if (K > CB_MAXK) {
    snprintf(g_err, sizeof(g_err), "refine_select: K=%d > %d", K, CB_MAXK);
    return 1;
  }
  hipLaunchKernelGGL(refine_select_kernel, dim3(B), dim3(CB_THREADS), 0, st, clip_score, final_score, K, theta, m_samples, gate_h, beta,
                     need_cos, gated, kind, list, count);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// exact cosines of the re-encoded candidates: row r of text_feat belongs to flat candidate rlist[r] = b*K + k
__global__ __launch_bounds__(256) void refine_cosine_kernel(const float* text_feat, const float* img_n, const int* rlist, const int* n_rows,
                                                            int K, int D, float* cos_out, int* nonfinite) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= *n_rows) return;
  const int flat = rlist[row];
  const float* t = text_feat + row * D;
  const float* img = img_n + (long)(flat / K) * D;
  float nn = 0.f;
  for (int c = lane; c < D; c += 64) nn += t[c] * t[c];
  const float nrm = sqrtf(wave_sum(nn));
  float dot = 0.f;
  for (int c = lane; c < D; c += 64) dot += (t[c] / nrm) * img[c];
  dot = wave_sum(dot);
  if (lane == 0) {
    cos_out[flat] = dot;
    if (nonfinite && !(fabsf(dot) <= 2.0f)) atomicOr(nonfinite, 1);
  }
}

int launch_refine_cosine(const float* text_feat, const float* img_n, const int* rlist, const int* n_rows_dev, int n_rows_max, int K, int D,
                         float* cos_out, int* nonfinite, hipStream_t st) {
  if (n_rows_max <= 0) return 0;
  hipLaunchKernelGGL(refine_cosine_kernel, dim3((unsigned)cdiv(n_rows_max, 4)), dim3(256), 0, st, text_feat, img_n, rlist, n_rows_dev, K, D,
                     cos_out, nonfinite);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_combine(const CombineArgs& a, hipStream_t st) {
  if (a.K > CB_MAXK) {
    snprintf(g_err, sizeof(g_err), "combine: K=%d > %d", a.K, CB_MAXK);
    return 1;
  }
  if (a.text_feat)  // null: clip_ref already holds the cosines (second combine of the screen-then-refine engine)
    hipLaunchKernelGGL(cosine_kernel, dim3((unsigned)cdiv((long)a.B * a.K, 4)), dim3(256), 0, st, a.text_feat, a.img_n, a.B, a.K, a.D,
                       a.clip_ref, a.nonfinite);
  hipLaunchKernelGGL(combine_kernel, dim3(a.B), dim3(CB_THREADS), 0, st, a);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

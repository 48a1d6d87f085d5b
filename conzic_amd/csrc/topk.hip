// softmax(logits / T) * token_mask -> top-K sorted descending     (gen_utils.py:42-47)
//
// One 1024-thread workgroup per image row.  The whole vocabulary row (V <= 39k fp32 = 156 KB)
// is staged once in LDS (MI355X: 160 KB per CU), so HBM sees exactly one 122 KB read per row;
// everything after that (max, sum-exp, masking, 4-pass 8-bit radix select on the float bit
// patterns, ordered tie compaction, bitonic sort of the K winners) runs out of LDS.
// Probabilities are NOT renormalised after masking (reference quirk, gen_utils.py:45-46).
// Tie rule (torch.topk leaves it unspecified): equal probabilities are taken in ascending id.
// cand = idx * mask[idx] (gen_utils.py:72): a masked id becomes 0 = [PAD].
#include "kernels.h"

namespace czc {

constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 1024;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < TK_THREADS / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

__global__ __launch_bounds__(TK_THREADS) void softmax_mask_topk_kernel(const float* logits, int V, int K,
                                                                        const float* mask, float temperature, int dot_id,
                                                                        int dot_allowed, float* probs_out, int* idx_out,
                                                                        int* cand_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* p = (float*)smem_raw;                       // [V] probabilities (as float / uint bits)
  unsigned* hist = (unsigned*)(p + ((V + 3) & ~3));  // [256]
  float* red = (float*)(hist + 256);                 // [16]
  unsigned* cnt = (unsigned*)(red + 16);             // [TK_THREADS] per-thread tie counts -> scan
  unsigned long long* keys = (unsigned long long*)(cnt + TK_THREADS);  // [TK_MAXK] sort buffer
  unsigned* sh = (unsigned*)(keys + TK_MAXK);  // [4] broadcast words (all LDS in the dynamic region: G17)
#define s_prefix sh[0]
#define s_need sh[1]
#define s_nsel sh[2]

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float* lr = logits + (long)b * V;

  // 1. logits / temperature, row max
  float mx = -INFINITY;
  for (int i = tid; i < V; i += TK_THREADS) {
    const float x = lr[i] / temperature;
    p[i] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_reduce(mx, red, true);
  // 2. exp, sum
  float sm = 0.f;
  for (int i = tid; i < V; i += TK_THREADS) {
    const float e = expf(p[i] - mx);
    p[i] = e;
    sm += e;
  }
  sm = block_reduce(sm, red, false);
  // 3. normalise and mask ('.' follows the per-position rule of utils.py:53-59)
  for (int i = tid; i < V; i += TK_THREADS) {
    float mk = mask[i];
    if (i == dot_id) mk = dot_allowed ? 1.0f : 0.0f;
    p[i] = (p[i] / sm) * mk;
  }
  __syncthreads();

  // 4. radix select the K-th largest bit pattern (values are >= 0, so uint order == float order)
  const unsigned* pu = (const unsigned*)p;
  unsigned prefix = 0, need = (unsigned)K;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += TK_THREADS) hist[i] = 0;
    __syncthreads();
    const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < V; i += TK_THREADS) {
      const unsigned u = pu[i];
      if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int bin = 255;
      for (; bin > 0; --bin) {
        if (acc + hist[bin] >= need) break;
        acc += hist[bin];
      }
      s_prefix = prefix | ((unsigned)bin << shift);
      s_need = need - acc;  // how many we still need from inside this bin
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  const unsigned thr = prefix;  // K-th largest value; `need` of the elements equal to thr are taken

  // 5. collect: everything > thr (unordered), then the first `need` ties in ascending id
  if (tid == 0) s_nsel = 0;
  for (int i = tid; i < TK_MAXK; i += TK_THREADS) keys[i] = 0ull;
  __syncthreads();
  const int per = (V + TK_THREADS - 1) / TK_THREADS;
  const int lo = tid * per, hi = min(V, lo + per);
  unsigned ties = 0;
  for (int i = lo; i < hi; ++i) {
    const unsigned u = pu[i];
    if (u > thr) {
      const unsigned slot = atomicAdd(&s_nsel, 1u);
      keys[slot] = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
    } else if (u == thr) {
      ++ties;
    }
  }
  cnt[tid] = ties;
  __syncthreads();
  // exclusive scan of cnt (Hillis-Steele over 1024 entries, in place with double read)
  for (int off = 1; off < TK_THREADS; off <<= 1) {
    const unsigned v = tid >= off ? cnt[tid - off] : 0u;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  unsigned rank = cnt[tid] - ties;  // exclusive prefix
  const unsigned base = s_nsel;     // number of strictly-greater elements (= K - need)
  __syncthreads();
  for (int i = lo; i < hi && rank < need; ++i) {
    if (pu[i] == thr) {
      keys[base + rank] = ((unsigned long long)thr << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
      ++rank;
    }
  }
  __syncthreads();

  // 6. bitonic sort, descending, of TK_MAXK composite keys (value bits, ~id); padding = 0
  for (int k2 = 2; k2 <= TK_MAXK; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const int i = tid;  // TK_THREADS == TK_MAXK
      const int ixj = i ^ j;
      if (ixj > i) {
        const unsigned long long a = keys[i], c = keys[ixj];
        const bool desc = (i & k2) == 0;
        if (desc ? (a < c) : (a > c)) {
          keys[i] = c;
          keys[ixj] = a;
        }
      }
      __syncthreads();
    }
  }
  if (tid < K) {
    const unsigned long long kk = keys[tid];
    const unsigned u = (unsigned)(kk >> 32);
    const int id = (int)(0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFu));
    probs_out[(long)b * K + tid] = __uint_as_float(u);
    idx_out[(long)b * K + tid] = id;
    float mk = mask[id];
    if (id == dot_id) mk = dot_allowed ? 1.0f : 0.0f;
    cand_out[(long)b * K + tid] = (int)((float)id * mk);
  }
}

int launch_softmax_mask_topk(const float* logits, int B, int V, int K, const float* mask, float temperature, int dot_id,
                             int dot_allowed, float* probs, int* idxs, int* cand, hipStream_t st) {
  if (K > TK_MAXK || K <= 0 || K > V) {
    snprintf(g_err, sizeof(g_err), "topk: K=%d unsupported (1..%d, <= V)", K, TK_MAXK);
    return 1;
  }
  const size_t shmem = (size_t)((V + 3) & ~3) * 4 + 256 * 4 + 16 * 4 + TK_THREADS * 4 + TK_MAXK * 8 + 16;
  if (shmem > 160 * 1024) {
    snprintf(g_err, sizeof(g_err), "topk: V=%d does not fit the 160 KB LDS", V);
    return 1;
  }
  CZC_HIP_CHECK(hipFuncSetAttribute((const void*)softmax_mask_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)shmem));
  hipLaunchKernelGGL(softmax_mask_topk_kernel, dim3(B), dim3(TK_THREADS), shmem, st, logits, V, K, mask, temperature,
                     dot_id, dot_allowed, probs, idxs, cand);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

// Attention over many tiny segments (BERT T<=64 bidirectional, CLIP text Tc<=77 causal, CLIP
// vision T=50): softmax(q k^T * scale [+causal mask]) v, head_dim 64
// (HF:bert/modeling_bert.py:111-136, HF:clip/modeling_clip.py:259-277,:309-333).
//
// A *segment* is a run of `own` rows (queries + keys/values) preceded by an optional run of
// `pre` rows that contribute keys/values only.  That is how the K candidate captions of one image
// share their causal prefix (all hidden states before the first differing CLIP
// token are bit-identical across candidates): the prefix ("trunk") is a segment of its own,
// computed once per image, and each candidate ("branch") attends to trunk rows + its own rows.
//
// Two kernels, same contract:
//  * attention_mfma_kernel (bf16 engine): one wave per (segment, head).  S^T = K.Q^T with
//    v_mfma_f32_32x32x16_bf16 (K rows are the A operand, Q rows the B operand, both fetched as
//    16-byte fragments straight from the packed qkv rows -- no LDS).  In that layout a lane owns
//    one query column and 16 of every 32 keys, so the softmax is in-lane plus one cross-half
//    shuffle, and the probabilities already sit in the B-operand layout of O^T = V^T.P^T; only V
//    goes through LDS (transposed on the way in).  8 MFMAs per (segment, head) at Tc<=32.
//  * attention_valu_kernel (f32 engine): exact-fp32 wavefront version, K^T/V/Q in LDS.
#include <type_traits>

#include "kernels.h"

namespace czc {

// ------------------------------------------------------------------------------------------------
// common segment decoding
// ------------------------------------------------------------------------------------------------
struct Seg {
  int pre_off, pre_len, own_off, own_len;
};

__device__ __forceinline__ Seg load_seg(const SegTable& t, int s) {
  Seg g;
  if (t.own_len) {
    g.pre_off = t.pre_off ? t.pre_off[s] : 0;
    g.pre_len = t.pre_len ? t.pre_len[s] : 0;
    g.own_off = t.own_off[s];
    g.own_len = t.own_len[s];
  } else {
    g.pre_off = 0;
    g.pre_len = 0;
    g.own_off = s * t.fixed_T;
    g.own_len = t.fixed_T;
  }
  return g;
}

__device__ __forceinline__ long key_row(const Seg& g, int k) {
  return k < g.pre_len ? (long)g.pre_off + k : (long)g.own_off + (k - g.pre_len);
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int AT_MAXKT = 3;  // key tiles of 32 -> up to 96 keys

template <typename HT>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const bf16_t* qkv, SegTable tab, int heads, int causal,
                                                             float scale, int KP, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int s = blockIdx.x;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const Seg g = load_seg(tab, s);
  const int nk = g.pre_len + g.own_len, nq = g.own_len;
  if (nq <= 0) return;  // uniform per block
  const int Hd = heads * 64;
  const long pitch = 3L * Hd;  // elements per qkv row
  bf16_t* Vt = (bf16_t*)at_lds + (size_t)wave * 64 * KP;  // [64 d][KP keys]

  // ---- V^T -> LDS: 8 keys per pass, lane = (key, 16-byte chunk of the head row) ----
  for (int k0 = 0; k0 < nk; k0 += 8) {
    const int k = k0 + (lane >> 3);
    if (k < nk) {
      const int d0 = (lane & 7) * 8;
      const uint4 v = *(const uint4*)(qkv + key_row(g, k) * pitch + 2 * Hd + h * 64 + d0);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) Vt[(d0 + e) * KP + k] = (bf16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
  }
  // zero the padding keys this wave will read (nk..32*ceil(nk/32))
  const int nkt_all = (nk + 31) >> 5;
  for (int i = lane; i < 64 * (nkt_all * 32 - nk); i += 64) {
    const int d = i / (nkt_all * 32 - nk), k = nk + i % (nkt_all * 32 - nk);
    Vt[d * KP + k] = 0;
  }
  __syncthreads();

  const int half = lane >> 5, l31 = lane & 31;
  for (int q0 = 0; q0 < nq; q0 += 32) {
    const int q = min(q0 + l31, nq - 1);  // padded query lanes recompute the last real query
    const bf16_t* qp = qkv + ((long)g.own_off + q) * pitch + h * 64 + 8 * half;
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(qp + 16 * ks);
    const int kmax = causal ? min(nk, g.pre_len + min(q0 + 32, nq)) : nk;  // keys any lane of this tile may see
    const int nkt = (kmax + 31) >> 5;
    const int vis = causal ? g.pre_len + q : nk - 1;  // last visible key of this lane's query

    f32x16_t st[AT_MAXKT];
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
        const int kk = min(kt * 32 + l31, nk - 1);
        const bf16_t* kp = qkv + key_row(g, kk) * pitch + Hd + h * 64 + 8 * half;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 kf = *(const uint4*)(kp + 16 * ks);
          acc = Half<HT>::mfma(kf, qf[ks], acc);
        }
        st[kt] = acc;
      }
    }
    // softmax over keys: lane holds keys kt*32 + (r&3) + 8*(r>>2) + 4*half of query column l31
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = key <= vis && key < nk ? st[kt][r] * scale : -INFINITY;
          st[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    uint4 pf[AT_MAXKT][2];
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          e[r] = expf(st[kt][r] - mx);  // exp(-inf) = 0 for masked keys
          sum += e[r];
        }
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          pf[kt][sstep].x = Half<HT>::pack2(e[8 * sstep + 0], e[8 * sstep + 1]);
          pf[kt][sstep].y = Half<HT>::pack2(e[8 * sstep + 2], e[8 * sstep + 3]);
          pf[kt][sstep].z = Half<HT>::pack2(e[8 * sstep + 4], e[8 * sstep + 5]);
          pf[kt][sstep].w = Half<HT>::pack2(e[8 * sstep + 6], e[8 * sstep + 7]);
        }
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; k-slot j of step s, half h <-> key
    // 16s + 4h + (j&3) + 8*(j>>2): exactly the accumulator order the probabilities were produced in.
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16_t o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      const bf16_t* vrow = Vt + (dt * 32 + l31) * KP;
#pragma unroll
      for (int kt = 0; kt < AT_MAXKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int sstep = 0; sstep < 2; ++sstep) {
            const uint2 lo = *(const uint2*)(vrow + kt * 32 + 16 * sstep + 4 * half);
            const uint2 hi = *(const uint2*)(vrow + kt * 32 + 16 * sstep + 8 + 4 * half);
            const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
            o = Half<HT>::mfma(vf, pf[kt][sstep], o);
          }
        }
      }
      if (live && q0 + l31 < nq) {
        bf16_t* op = out + ((long)g.own_off + q0 + l31) * Hd + h * 64 + dt * 32 + 4 * half;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          uint2 w;
          w.x = Half<HT>::pack2(o[4 * qd] * inv, o[4 * qd + 1] * inv);
          w.y = Half<HT>::pack2(o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
          *(uint2*)(op + 8 * qd) = w;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// split-fp16 MFMA kernel (split engine precision): attention_mfma_kernel's structure with every product as three
// v_mfma_f32_32x32x16_f16 passes over fp16 hi/lo planes -- q, k, v arrive as split_t (common.h), the softmax
// probabilities are split into hi + lo before the PV product, V^T sits in LDS as two fp16 planes.  fp32-class
// accuracy (the split engine's bar is 1e-4 on the fused score) at MFMA speed instead of the VALU kernel's.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_mfma_split_kernel(const split_t* qkv, SegTable tab, int heads, int causal,
                                                                   float scale, int KP, split_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int s = blockIdx.x;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const Seg g = load_seg(tab, s);
  const int nk = g.pre_len + g.own_len, nq = g.own_len;
  if (nq <= 0) return;  // uniform per block
  const int Hd = heads * 64;
  const long pitchb = 3L * Hd * 4;  // bytes per qkv row (split_t: 4 bytes per element)
  const unsigned char* base = (const unsigned char*)qkv;
  _Float16* Vh = (_Float16*)at_lds + (size_t)wave * 2 * 64 * KP;  // [64 d][KP keys] hi plane, then lo plane
  _Float16* Vl = Vh + 64 * KP;
  // byte offset of the 8-element group that starts at element e0 (e0 % 8 == 0) of a row: hi 16 bytes, lo at +16
#define CZC_GRP(e0) ((long)((e0) >> 3) * 32)

  for (int k0 = 0; k0 < nk; k0 += 8) {
    const int k = k0 + (lane >> 3);
    if (k < nk) {
      const int d0 = (lane & 7) * 8;
      const unsigned char* vp = base + key_row(g, k) * pitchb + CZC_GRP(2 * Hd + h * 64 + d0);
      const uint4 vh = *(const uint4*)vp, vl = *(const uint4*)(vp + 16);
      const unsigned wh[4] = {vh.x, vh.y, vh.z, vh.w}, wl[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned short a = (unsigned short)((wh[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        const unsigned short b = (unsigned short)((wl[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        Vh[(d0 + e) * KP + k] = __builtin_bit_cast(_Float16, a);
        Vl[(d0 + e) * KP + k] = __builtin_bit_cast(_Float16, b);
      }
    }
  }
  const int nkt_all = (nk + 31) >> 5;
  for (int i = lane; i < 64 * (nkt_all * 32 - nk); i += 64) {
    const int d = i / (nkt_all * 32 - nk), k = nk + i % (nkt_all * 32 - nk);
    Vh[d * KP + k] = (_Float16)0.f;
    Vl[d * KP + k] = (_Float16)0.f;
  }
  __syncthreads();

#define CZC_F16(v_) __builtin_bit_cast(f16x8_t, v_)
  const int half = lane >> 5, l31 = lane & 31;
  for (int q0 = 0; q0 < nq; q0 += 32) {
    const int q = min(q0 + l31, nq - 1);
    const unsigned char* qp = base + ((long)g.own_off + q) * pitchb + CZC_GRP(h * 64 + 8 * half);
    uint4 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // step ks: elements 16ks + 8half .. +7 -> group 2ks + half
      qh[ks] = *(const uint4*)(qp + ks * 64);
      ql[ks] = *(const uint4*)(qp + ks * 64 + 16);
    }
    const int kmax = causal ? min(nk, g.pre_len + min(q0 + 32, nq)) : nk;
    const int nkt = (kmax + 31) >> 5;
    const int vis = causal ? g.pre_len + q : nk - 1;

    f32x16_t st[AT_MAXKT];
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
        const int kk = min(kt * 32 + l31, nk - 1);
        const unsigned char* kp = base + key_row(g, kk) * pitchb + CZC_GRP(Hd + h * 64 + 8 * half);
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 kh = *(const uint4*)(kp + ks * 64), kl = *(const uint4*)(kp + ks * 64 + 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kl), CZC_F16(qh[ks]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kh), CZC_F16(ql[ks]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kh), CZC_F16(qh[ks]), acc, 0, 0, 0);
        }
        st[kt] = acc;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = key <= vis && key < nk ? st[kt][r] * scale : -INFINITY;
          st[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    uint4 ph[AT_MAXKT][2], pl[AT_MAXKT][2];
#pragma unroll
    for (int kt = 0; kt < AT_MAXKT; ++kt) {
      if (kt < nkt) {
        float e[16], el[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          e[r] = pin(expf(st[kt][r] - mx));  // one fp32 value behind hi and lo (common.h pin())
          sum += e[r];
          el[r] = e[r] - (float)(_Float16)e[r];
        }
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          ph[kt][sstep] = make_uint4(pack2_f16(e[8 * sstep + 0], e[8 * sstep + 1]), pack2_f16(e[8 * sstep + 2], e[8 * sstep + 3]),
                                     pack2_f16(e[8 * sstep + 4], e[8 * sstep + 5]), pack2_f16(e[8 * sstep + 6], e[8 * sstep + 7]));
          pl[kt][sstep] = make_uint4(pack2_f16(el[8 * sstep + 0], el[8 * sstep + 1]), pack2_f16(el[8 * sstep + 2], el[8 * sstep + 3]),
                                     pack2_f16(el[8 * sstep + 4], el[8 * sstep + 5]), pack2_f16(el[8 * sstep + 6], el[8 * sstep + 7]));
        }
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16_t o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      const _Float16* vrh = Vh + (dt * 32 + l31) * KP;
      const _Float16* vrl = Vl + (dt * 32 + l31) * KP;
#pragma unroll
      for (int kt = 0; kt < AT_MAXKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int sstep = 0; sstep < 2; ++sstep) {
            const int ko = kt * 32 + 16 * sstep + 4 * half;
            const uint2 h0 = *(const uint2*)(vrh + ko), h1 = *(const uint2*)(vrh + ko + 8);
            const uint2 l0 = *(const uint2*)(vrl + ko), l1 = *(const uint2*)(vrl + ko + 8);
            const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vl), CZC_F16(ph[kt][sstep]), o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vh), CZC_F16(pl[kt][sstep]), o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vh), CZC_F16(ph[kt][sstep]), o, 0, 0, 0);
          }
        }
      }
      if (live && q0 + l31 < nq) {
        const long eo = ((long)g.own_off + q0 + l31) * Hd + h * 64 + dt * 32 + 4 * half;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          Act<split_t>::st4(out, eo + 8 * qd, o[4 * qd] * inv, o[4 * qd + 1] * inv, o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
      }
    }
  }
#undef CZC_F16
#undef CZC_GRP
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel for the BRANCH segments of a shared-prefix plan.
// The K candidates of an image own only a handful of rows each (the rows from the first differing
// token on) and are laid out back to back, so one wave packs G consecutive candidates of one image
// into a single 32-query MFMA tile: their trunk keys are shared (one S^T tile per 32 trunk keys for
// all G candidates at once), their own keys form one block-diagonal 32x32 tile (a query sees the
// own keys of its own candidate up to itself).  ~G times fewer waves and MFMAs than one wave per
// candidate; requires G * max(own_len) <= 32.
// ------------------------------------------------------------------------------------------------
constexpr int AB_MAXT = 3;  // trunk key tiles (<= 96 trunk keys)

template <typename HT>
__global__ __launch_bounds__(256) void attention_branch_kernel(const bf16_t* qkv, SegTable tab, int B, int K, int G,
                                                               int heads, float scale, int KP, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const int gpi = (K + G - 1) / G;  // groups per image the grid provides (G = the batch-wide packing factor)
  const int b = blockIdx.x / gpi;
  // this image's own packing factor: its arithmetic must not depend on the other images of the batch
  const int Gb = tab.img_max ? 32 / max(1, tab.img_max[b]) : G;
  const int k0 = (blockIdx.x - b * gpi) * Gb;
  if (k0 >= K) return;  // Gb >= G: an image with short branches needs fewer groups than the grid provides
  const int Gc = min(Gb, K - k0);
  const int s0 = B + b * K + k0;
  const int pre_off = tab.pre_off[s0], pre_len = tab.pre_len[s0];
  const int r0 = tab.own_off[s0];
  const int n_own = tab.own_off[s0 + Gc - 1] + tab.own_len[s0 + Gc - 1] - r0;  // <= 32 by construction
  const int nkt_t = (pre_len + 31) >> 5;
  const int Hd = heads * 64;
  const long pitch = 3L * Hd;
  // V image of this wave, ROW-major: 4 sub-tiles of 16 head dims, each [KP key slots][16 dims] (32-byte rows).
  // Filled with plain 16-byte stores and consumed with ds_read_b64_tr_b16, which hands every lane of a
  // 16-lane group one COLUMN of a [4 keys][16 dims] block -- the k-contiguous MFMA operand -- so the
  // transpose costs nothing (the first version scattered 2-byte stores: 64 ds_write_b16 per lane).
  // KP = 1 (mod 4): the four sub-tiles of one key land on different bank quarters.
  unsigned char* Vimg = at_lds + (size_t)wave * 128 * KP;
  const int SUB = KP * 32;
  const int own_slot0 = nkt_t * 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int q = min(l31, n_own - 1);
  // the query fragments first: their latency hides behind the V fill
  const bf16_t* qp = qkv + ((long)r0 + q) * pitch + h * 64 + 8 * half;
  uint4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(qp + 16 * ks);
  {
    // 32 key slots per pass, four independent 16-byte loads in flight per lane (a one-load-per-iteration loop
    // serialises 8-16 L2 round trips in front of everything else; this kernel is latency-bound, not byte-bound)
    const int sc = lane & 7;
    unsigned char* wp = Vimg + (sc >> 1) * SUB + (sc & 1) * 16;
    const bf16_t* vsrc = qkv + 2 * Hd + h * 64 + sc * 8;
    for (int kb = 0; kb < own_slot0 + 32; kb += 32) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + 8 * u + (lane >> 3);
        long row = -1;
        if (k < pre_len) row = (long)pre_off + k;
        else if (k >= own_slot0 && k - own_slot0 < n_own) row = (long)r0 + (k - own_slot0);
        // padding slots must be zero: their P is 0, and 0 * garbage is not
        v[u] = row >= 0 ? *(const uint4*)(vsrc + row * pitch) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) *(uint4*)(wp + (kb + 8 * u + (lane >> 3)) * 32) = v[u];
    }
  }
  // no barrier: the image is private to this wave and a wave's LDS operations execute in order
  asm volatile("" ::: "memory");

  // first slot of this query's own candidate
  int ss = 0;
  for (int j = 1; j < Gc; ++j) {
    const int o = tab.own_off[s0 + j] - r0;
    if (o <= q) ss = o;
  }

  f32x16_t st[AB_MAXT + 1];
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {  // tiles 0..nkt_t-1 trunk, tile nkt_t = own
      long krow;
      if (t < nkt_t) krow = (long)pre_off + min(t * 32 + l31, pre_len - 1);
      else krow = (long)r0 + min(l31, n_own - 1);
      const bf16_t* kp = qkv + krow * pitch + Hd + h * 64 + 8 * half;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *(const uint4*)(kp + 16 * ks);
        acc = Half<HT>::mfma(kf, qf[ks], acc);
      }
      st[t] = acc;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = t < nkt_t ? (t * 32 + idx < pre_len) : (idx >= ss && idx <= q);
        const float v = ok ? st[t][r] * scale : -INFINITY;
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  uint4 pf[AB_MAXT + 1][2];
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = __expf(st[t][r] - mx);
        sum += e[r];
      }
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        pf[t][sstep].x = Half<HT>::pack2(e[8 * sstep + 0], e[8 * sstep + 1]);
        pf[t][sstep].y = Half<HT>::pack2(e[8 * sstep + 2], e[8 * sstep + 3]);
        pf[t][sstep].z = Half<HT>::pack2(e[8 * sstep + 4], e[8 * sstep + 5]);
        pf[t][sstep].w = Half<HT>::pack2(e[8 * sstep + 6], e[8 * sstep + 7]);
      }
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16_t o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    // lane (dim l31 of this 32-dim half, half): keys 16*sstep + 4*half + {0..3} and + 8 + {0..3} of tile t --
    // the slots its P fragment holds.  Its 16-lane group reads the [4 keys][16 dims] block of sub-tile
    // 2*dt + (l31>>4); lane i of the group supplies row i>>2, dims 4*(i&3).. and receives column i.
    typedef __attribute__((ext_vector_type(4))) short tr4_t;
    typedef __attribute__((address_space(3))) tr4_t* tr4_lds_t;
    const unsigned char* vbase =
        Vimg + (dt * 2 + (l31 >> 4)) * SUB + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
    for (int t = 0; t <= AB_MAXT; ++t) {
      if (t <= nkt_t) {
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          const unsigned char* vp = vbase + (t * 32 + 16 * sstep) * 32;
          const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(vp)));
          const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(vp + 256)));
          const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
          o = Half<HT>::mfma(vf, pf[t][sstep], o);
        }
      }
    }
    if (live && l31 < n_own) {
      bf16_t* op = out + ((long)r0 + l31) * Hd + h * 64 + dt * 32 + 4 * half;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        uint2 w;
        w.x = Half<HT>::pack2(o[4 * qd] * inv, o[4 * qd + 1] * inv);
        w.y = Half<HT>::pack2(o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
        *(uint2*)(op + 8 * qd) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// split-fp16 form of attention_branch_kernel (split engine precision): G candidates of one image packed into one
// 32-query tile, shared trunk keys + block-diagonal own keys, every product three v_mfma_f32_32x32x16_f16 passes.
// q / k / v rows are split_t (16 bytes of fp16 hi parts + 16 bytes of lo parts per 8 elements); the V image in LDS
// is two fp16 planes in the [sub-tile][key slot][16 dims] layout ds_read_b64_tr_b16 reads as the k-contiguous operand.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_branch_split_kernel(const split_t* qkv, SegTable tab, int B, int K, int G,
                                                                     int heads, float scale, int KP, split_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const int gpi = (K + G - 1) / G;
  const int b = blockIdx.x / gpi;
  const int Gb = tab.img_max ? 32 / max(1, tab.img_max[b]) : G;  // per image, as in attention_branch_kernel
  const int k0 = (blockIdx.x - b * gpi) * Gb;
  if (k0 >= K) return;
  const int Gc = min(Gb, K - k0);
  const int s0 = B + b * K + k0;
  const int pre_off = tab.pre_off[s0], pre_len = tab.pre_len[s0];
  const int r0 = tab.own_off[s0];
  const int n_own = tab.own_off[s0 + Gc - 1] + tab.own_len[s0 + Gc - 1] - r0;  // <= 32 by construction
  if (n_own <= 0) return;  // a group of empty slots (refine plan: images with fewer chosen candidates than the widest one)
  const int nkt_t = (pre_len + 31) >> 5;
  const int Hd = heads * 64;
  const long pitchb = 3L * Hd * 4;
  const unsigned char* base = (const unsigned char*)qkv;
#define CZC_GRP(e0) ((long)((e0) >> 3) * 32)
#define CZC_F16(v_) __builtin_bit_cast(f16x8_t, v_)
  unsigned char* Vh = at_lds + (size_t)wave * 2 * 128 * KP;  // hi plane, then lo plane (128*KP bytes each)
  unsigned char* Vl = Vh + 128 * KP;
  const int SUB = KP * 32;
  const int own_slot0 = nkt_t * 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int q = min(l31, n_own - 1);
  const unsigned char* qp = base + ((long)r0 + q) * pitchb + CZC_GRP(h * 64 + 8 * half);
  uint4 qh[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qh[ks] = *(const uint4*)(qp + ks * 64);
    ql[ks] = *(const uint4*)(qp + ks * 64 + 16);
  }
  {
    const int sc = lane & 7;  // 8 head dims sc*8.. of one key: sub-tile sc>>1, half row (sc&1)*16 bytes
    const int wo = (sc >> 1) * SUB + (sc & 1) * 16;
    const unsigned char* vsrc = base + CZC_GRP(2 * Hd + h * 64 + sc * 8);
    for (int kb = 0; kb < own_slot0 + 32; kb += 32) {
      uint4 vh[4], vl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + 8 * u + (lane >> 3);
        long row = -1;
        if (k < pre_len) row = (long)pre_off + k;
        else if (k >= own_slot0 && k - own_slot0 < n_own) row = (long)r0 + (k - own_slot0);
        vh[u] = row >= 0 ? *(const uint4*)(vsrc + row * pitchb) : make_uint4(0, 0, 0, 0);
        vl[u] = row >= 0 ? *(const uint4*)(vsrc + row * pitchb + 16) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        *(uint4*)(Vh + wo + (kb + 8 * u + (lane >> 3)) * 32) = vh[u];
        *(uint4*)(Vl + wo + (kb + 8 * u + (lane >> 3)) * 32) = vl[u];
      }
    }
  }
  asm volatile("" ::: "memory");  // the image is private to this wave; a wave's LDS operations execute in order

  int ss = 0;
  for (int j = 1; j < Gc; ++j) {
    const int o = tab.own_off[s0 + j] - r0;
    if (o <= q) ss = o;
  }

  f32x16_t st[AB_MAXT + 1];
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {
      long krow;
      if (t < nkt_t) krow = (long)pre_off + min(t * 32 + l31, pre_len - 1);
      else krow = (long)r0 + min(l31, n_own - 1);
      const unsigned char* kp = base + krow * pitchb + CZC_GRP(Hd + h * 64 + 8 * half);
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kh = *(const uint4*)(kp + ks * 64), kl = *(const uint4*)(kp + ks * 64 + 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kl), CZC_F16(qh[ks]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kh), CZC_F16(ql[ks]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(kh), CZC_F16(qh[ks]), acc, 0, 0, 0);
      }
      st[t] = acc;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = t < nkt_t ? (t * 32 + idx < pre_len) : (idx >= ss && idx <= q);
        const float v = ok ? st[t][r] * scale : -INFINITY;
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  uint4 ph[AB_MAXT + 1][2], pl[AB_MAXT + 1][2];
#pragma unroll
  for (int t = 0; t <= AB_MAXT; ++t) {
    if (t <= nkt_t) {
      float e[16], el[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = pin(expf(st[t][r] - mx));
        sum += e[r];
        el[r] = e[r] - (float)(_Float16)e[r];
      }
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        ph[t][sstep] = make_uint4(pack2_f16(e[8 * sstep + 0], e[8 * sstep + 1]), pack2_f16(e[8 * sstep + 2], e[8 * sstep + 3]),
                                  pack2_f16(e[8 * sstep + 4], e[8 * sstep + 5]), pack2_f16(e[8 * sstep + 6], e[8 * sstep + 7]));
        pl[t][sstep] = make_uint4(pack2_f16(el[8 * sstep + 0], el[8 * sstep + 1]), pack2_f16(el[8 * sstep + 2], el[8 * sstep + 3]),
                                  pack2_f16(el[8 * sstep + 4], el[8 * sstep + 5]), pack2_f16(el[8 * sstep + 6], el[8 * sstep + 7]));
      }
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16_t o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    typedef __attribute__((ext_vector_type(4))) short tr4_t;
    typedef __attribute__((address_space(3))) tr4_t* tr4_lds_t;
    const int voff = (dt * 2 + (l31 >> 4)) * SUB + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
    for (int t = 0; t <= AB_MAXT; ++t) {
      if (t <= nkt_t) {
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          const int vo = voff + (t * 32 + 16 * sstep) * 32;
          const uint2 h0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(Vh + vo)));
          const uint2 h1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(Vh + vo + 256)));
          const uint2 l0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(Vl + vo)));
          const uint2 l1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(Vl + vo + 256)));
          const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vl), CZC_F16(ph[t][sstep]), o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vh), CZC_F16(pl[t][sstep]), o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(CZC_F16(vh), CZC_F16(ph[t][sstep]), o, 0, 0, 0);
        }
      }
    }
    if (live && l31 < n_own) {
      const long eo = ((long)r0 + l31) * Hd + h * 64 + dt * 32 + 4 * half;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        Act<split_t>::st4(out, eo + 8 * qd, o[4 * qd] * inv, o[4 * qd + 1] * inv, o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
    }
  }
#undef CZC_F16
#undef CZC_GRP
}

// ------------------------------------------------------------------------------------------------
// attention_image_kernel: the same packed-branch attention, organised so that HBM latency is never on a
// wave's critical path.  attention_branch_kernel gives every (group, head) its own wave, and each wave
// is one chain of dependent round trips (segment table -> Q/V rows -> K rows -> ... -> store): at 16
// waves per CU the kernel runs at 2.7 TB/s although it only moves 1.3 GB.  Here a work-group owns ONE
// image and four heads and walks the image's K/G groups itself:
//   * a group's 32 rows (q, k, v columns of the four heads: 48 KiB) are fetched by LDS-DMA into a two-deep
//     ring while the previous group is being computed, 12 DMA instructions per wave and group, exact
//     `vmcnt` accounting as in gemm_wreg.hip (loop VMEM only from inline asm).  (A three-deep ring with the
//     queries in registers measured the same 0.39 ms: the kernel is bound by the ~80 VMEM instructions a
//     work-group issues per group, not by how far ahead they are issued.);
//   * the image's trunk keys / values are loaded once, the K+1 segment offsets once (LDS);
//   * Q/K tiles are row-major [32][512 B] with 16-byte chunks XOR (row & 15) (conflict-free ds_read_b128),
//     V goes straight into the [16-dim sub-tile][key][16] image that ds_read_b64_tr_b16 wants: one DMA
//     instruction gathers one sub-tile (32 keys x 32 B);
//   * rows past the group (the next group's) are zero-filled by the descriptor and dropped on store.
// One wave = one head; math identical to attention_branch_kernel (S^T = K Q^T in-lane softmax, P V by MFMA).
// Needs heads % 4 == 0, trunk <= 32 keys, K <= 1024.
// ------------------------------------------------------------------------------------------------
constexpr int A2_ROWB = 512;             // 4 heads x 64 dims x bf16
constexpr int A2_TILE = 32 * A2_ROWB;    // 16 KiB
constexpr int A2_STAGE = 3 * A2_TILE;    // q, k, v of one group
constexpr int A2_RING = 2;               // the next group streams in while this one is computed
constexpr int A2_META = 4224;            // K + 1 segment offsets (K <= 1024), padded
constexpr int A2_LDS = A2_RING * A2_STAGE + 2 * A2_TILE + A2_META;

template <typename HT>
__global__ __launch_bounds__(256) void attention_image_kernel(const bf16_t* qkv, SegTable tab, int B, int K, int G, int heads,
                                                              float scale, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x, hq = blockIdx.y;
  const int h = hq * 4 + wave;
  const int Hd = heads * 64;
  const int pitch = 3 * Hd * 2;  // bytes per qkv row
  unsigned char* ring = at_lds;
  unsigned char* Kt = at_lds + A2_RING * A2_STAGE;  // trunk keys, same layout as a K tile
  unsigned char* Vt = Kt + A2_TILE;                 // trunk values, sub-tile images
  int* meta = (int*)(Vt + A2_TILE);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)at_lds);

  const int sb = B + b * K;
  for (int j = threadIdx.x; j <= K; j += 256)
    meta[j] = j < K ? tab.own_off[sb + j] : tab.own_off[sb + K - 1] + tab.own_len[sb + K - 1];
  const int pre_off = tab.pre_off[sb], pre_len = tab.pre_len[sb];
  __syncthreads();

  // ---- DMA plumbing: per-lane source offsets inside a 32-row block ----
  // row-major tiles: piece p (0..15) = rows 2p, 2p+1; lane -> row 2p + (lane>>5), physical chunk lane&31
  // sub-tile images: piece s (0..15) = sub-tile s; lane -> key lane>>1, 16-byte half lane&1
  int vo_row[4], vo_sub[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 2 * (wave * 4 + u) + (lane >> 5);
    vo_row[u] = r * pitch + hq * A2_ROWB + (((lane & 31) ^ (r & 15)) << 4);
    vo_sub[u] = (lane >> 1) * pitch + hq * A2_ROWB + (wave * 4 + u) * 32 + (lane & 1) * 16;
  }
  auto dma4 = [&](unsigned dst, const int (&vo)[4], int part, u32x4_t rs) {  // this wave's quarter of one tile
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst + wave * 4096), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "s"(rs), "s"(part * Hd * 2)
        : "memory", "scc");
  };
  auto rows_desc = [&](long row0, int nrows) {  // descriptor over rows [row0, row0 + nrows): the rest reads as zero
    const unsigned long long pa = (unsigned long long)qkv + (unsigned long long)row0 * pitch;
    u32x4_t rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    rs.z = __builtin_amdgcn_readfirstlane((unsigned)(nrows * pitch));
    rs.w = 0x00020000u;
    return rs;
  };
  const int Gb = tab.img_max ? 32 / max(1, __builtin_amdgcn_readfirstlane(tab.img_max[b])) : G;  // this image's own packing factor
  const int ngroups = (K + Gb - 1) / Gb;
  auto issue_group = [&](int gi) {
    const int k0 = gi * Gb, Gc = min(Gb, K - k0);
    const int r0 = __builtin_amdgcn_readfirstlane(meta[k0]);
    const int n_own = min(__builtin_amdgcn_readfirstlane(meta[k0 + Gc]) - r0, 32);
    const u32x4_t rs = rows_desc(r0, n_own);
    const unsigned st = lds0 + (gi & 1) * A2_STAGE;
    dma4(st, vo_row, 0, rs);
    dma4(st + A2_TILE, vo_row, 1, rs);
    dma4(st + 2 * A2_TILE, vo_sub, 2, rs);
  };
  // trunk (once), then the first group.  VMEM per wave: trunk 8, every group 12 DMA, 4 stores.
  {
    const u32x4_t rt = rows_desc(pre_off, min(pre_len, 32));
    dma4(lds0 + A2_RING * A2_STAGE, vo_row, 1, rt);
    dma4(lds0 + A2_RING * A2_STAGE + A2_TILE, vo_sub, 2, rt);
  }
  issue_group(0);

  const int opitch = Hd * 2;
  const int nkt_t = pre_len > 0 ? 1 : 0;

  typedef __attribute__((ext_vector_type(4))) short tr4_t;
  typedef __attribute__((address_space(3))) tr4_t* tr4_lds_t;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

  for (int gi = 0; gi < ngroups; ++gi) {
    // group gi landed?  issued after its DMA: the 4 stores of group gi-1 (in order on vmcnt)
    if (gi == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // publishes group gi; every wave has left the other stage
    asm volatile("" ::: "memory");
    if (gi + 1 < ngroups) issue_group(gi + 1);

    const int k0 = gi * Gb, Gc = min(Gb, K - k0);
    const int r0 = __builtin_amdgcn_readfirstlane(meta[k0]);
    const int n_own = min(__builtin_amdgcn_readfirstlane(meta[k0 + Gc]) - r0, 32);
    const unsigned char* Qs = ring + (gi & 1) * A2_STAGE;
    const unsigned char* Ks = Qs + A2_TILE;
    const unsigned char* Vs = Ks + A2_TILE;
    const int q = min(l31, n_own - 1);
    int ss = 0;  // first slot of this query's own candidate
    for (int j = 1; j < Gc; ++j) {
      const int o = meta[k0 + j] - r0;
      if (o <= q) ss = o;
    }
    // fragments: head `wave` owns chunks 8*wave .. 8*wave+7 of a row; step ks takes chunk 2ks + half
    const int cq = wave * 8 + half;
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(Qs + q * A2_ROWB + (((cq + 2 * ks) ^ (q & 15)) << 4));
    f32x16_t st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned char* kb = t == 0 ? Kt : Ks;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *(const uint4*)(kb + l31 * A2_ROWB + (((cq + 2 * ks) ^ (l31 & 15)) << 4));
        acc = Half<HT>::mfma(kf, qf[ks], acc);
      }
      st[t] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = t == 0 ? (idx < pre_len) : (idx >= ss && idx <= q);
        const float v = ok ? st[t][r] * scale : -INFINITY;
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    uint4 pf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = __expf(st[t][r] - mx);
        sum += e[r];
      }
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        pf[t][sstep].x = Half<HT>::pack2(e[8 * sstep + 0], e[8 * sstep + 1]);
        pf[t][sstep].y = Half<HT>::pack2(e[8 * sstep + 2], e[8 * sstep + 3]);
        pf[t][sstep].z = Half<HT>::pack2(e[8 * sstep + 4], e[8 * sstep + 5]);
        pf[t][sstep].w = Half<HT>::pack2(e[8 * sstep + 6], e[8 * sstep + 7]);
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    u32x4_t rc;  // output descriptor: rows of this group only (the rest is dropped)
    {
      const unsigned long long pc = (unsigned long long)out + (unsigned long long)r0 * opitch;
      rc.x = __builtin_amdgcn_readfirstlane((unsigned)pc);
      rc.y = __builtin_amdgcn_readfirstlane((unsigned)(pc >> 32) & 0xffffu);
      rc.z = __builtin_amdgcn_readfirstlane((unsigned)(n_own * opitch));
      rc.w = 0x00020000u;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16_t o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      // sub-tile (4*wave + 2*dt + (l31>>4)); lane i of a 16-lane group supplies row i>>2, dims 4*(i&3)..
      const int sub_off = (wave * 4 + dt * 2 + (l31 >> 4)) * 1024 + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= 1 - nkt_t) {  // the trunk tile is skipped for an empty trunk
          const unsigned char* vb = (t == 0 ? Vt : Vs) + sub_off;
#pragma unroll
          for (int sstep = 0; sstep < 2; ++sstep) {
            const unsigned char* vp = vb + 16 * sstep * 32;
            const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(vp)));
            const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t)(vp + 256)));
            const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
            o = Half<HT>::mfma(vf, pf[t][sstep], o);
          }
        }
      }
      // lane = query row l31, registers = dims 8qd + 4half + e of this 32-dim half.  v_permlane32_swap pairs
      // quad qd of the lower half-wave with quad qd+2 of the upper one, so that every lane ends up with 8
      // consecutive dims: two 16-byte stores per lane instead of four 8-byte ones (VMEM issue is what binds).
      u32x2_t w[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        w[qd].x = Half<HT>::pack2(o[4 * qd] * inv, o[4 * qd + 1] * inv);
        w[qd].y = Half<HT>::pack2(o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
      }
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) {
        const u32x2_t sx = __builtin_amdgcn_permlane32_swap(w[qa].x, w[qa + 2].x, false, false);
        const u32x2_t sy = __builtin_amdgcn_permlane32_swap(w[qa].y, w[qa + 2].y, false, false);
        const u32x4_t d = {sx.x, sy.x, sx.y, sy.y};  // lower half-wave: quad qa, dims 0..7; upper: quad qa+2
        const unsigned co = (unsigned)(l31 * opitch + (h * 64 + dt * 32 + 8 * (qa + 2 * half)) * 2);
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(co), "s"(rc) : "memory");
      }
    }
  }
}

int g_use_attention_image = 1;  // 0 off, 1 when the batch fills the chip, 2 always (tests)

// trunks through the generic kernel (n_seg = B), branches packed G per wave; f16: operands are IEEE fp16 (not bf16)
int launch_attention_shared(const void* qkv, const SegTable& tab, int B, int K, int max_own, int max_keys, int heads,
                            float scale, void* out, hipStream_t st, int f16) {
  if (max_keys > 96 || max_own > 32 || max_own <= 0) return -1;  // caller falls back to the generic path
  const int G = 32 / max_own;
  const int KPt = ((max_keys + 31) & ~31) + 4;
  const int wpb = heads >= 4 ? 4 : (heads >= 2 ? 2 : 1);
  {
    SegTable trunks = tab;
    trunks.n_seg = B;
    dim3 grid(B, cdiv(heads, wpb)), block(64 * wpb);
    if (f16) hipLaunchKernelGGL(attention_mfma_kernel<f16_t>, grid, block, (size_t)wpb * 64 * KPt * 2, st, (const bf16_t*)qkv,
                                trunks, heads, 1, scale, KPt, (bf16_t*)out);
    else hipLaunchKernelGGL(attention_mfma_kernel<bf16_t>, grid, block, (size_t)wpb * 64 * KPt * 2, st, (const bf16_t*)qkv, trunks,
                            heads, 1, scale, KPt, (bf16_t*)out);
  }
  // one work-group per (image, 4 heads) walking the image's groups in sequence: needs enough images to fill the
  // chip (a single image would leave 2 work-groups doing 40 groups each; the per-group kernel spreads those)
  if (g_use_attention_image && heads % 4 == 0 && max_keys <= 32 && K <= 1024 &&
      (B * (heads / 4) >= 128 || g_use_attention_image == 2)) {
    // one-time set-up behind a function-local static (two engines launch from two host threads)
    static PerDeviceInit per_dev;
  const LaunchInit init = per_dev.get([](LaunchInit&) -> int {
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_image_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS));
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_image_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS));
      return 0;
    });
    if (init.rc) return launch_init_failed("attention_image");
    if (f16) hipLaunchKernelGGL(attention_image_kernel<f16_t>, dim3(B, heads / 4), dim3(256), A2_LDS, st, (const bf16_t*)qkv, tab, B,
                                K, G, heads, scale, (bf16_t*)out);
    else hipLaunchKernelGGL(attention_image_kernel<bf16_t>, dim3(B, heads / 4), dim3(256), A2_LDS, st, (const bf16_t*)qkv, tab, B, K,
                            G, heads, scale, (bf16_t*)out);
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const int KP = ((max_keys + 31) & ~31) + 32 + 1;  // key slots: trunk tiles + the own tile, padded to 1 (mod 4)
  const int gpi = cdiv(K, G);
  dim3 grid(B * gpi, cdiv(heads, wpb)), block(64 * wpb);
  if (f16) hipLaunchKernelGGL(attention_branch_kernel<f16_t>, grid, block, (size_t)wpb * 128 * KP, st, (const bf16_t*)qkv, tab, B, K,
                              G, heads, scale, KP, (bf16_t*)out);
  else hipLaunchKernelGGL(attention_branch_kernel<bf16_t>, grid, block, (size_t)wpb * 128 * KP, st, (const bf16_t*)qkv, tab, B, K,
                          G, heads, scale, KP, (bf16_t*)out);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// split engine precision, shared-prefix plan: trunks through attention_mfma_split_kernel, branches packed G per tile
int launch_attention_shared_split(const void* qkv, const SegTable& tab, int B, int K, int max_own, int max_keys, int heads,
                                  float scale, void* out, hipStream_t st) {
  if (max_keys > 96 || max_own > 32 || max_own <= 0) return -1;  // caller falls back to the generic path
  const int G = 32 / max_own;
  const int KPt = ((max_keys + 31) & ~31) + 4;
  int wpb = heads >= 4 ? 4 : (heads >= 2 ? 2 : 1);
  {
    SegTable trunks = tab;
    trunks.n_seg = B;
    int wt = wpb;
    while (wt > 1 && (size_t)wt * 2 * 64 * KPt * 2 > 150 * 1024) wt >>= 1;
    const size_t shm = (size_t)wt * 2 * 64 * KPt * 2;
    if (shm > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_mfma_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    dim3 grid(B, cdiv(heads, wt)), block(64 * wt);
    hipLaunchKernelGGL(attention_mfma_split_kernel, grid, block, shm, st, (const split_t*)qkv, trunks, heads, 1, scale, KPt,
                       (split_t*)out);
  }
  const int KP = ((max_keys + 31) & ~31) + 32 + 1;
  const int gpi = cdiv(K, G);
  while (wpb > 1 && (size_t)wpb * 2 * 128 * KP > 150 * 1024) wpb >>= 1;
  const size_t shm = (size_t)wpb * 2 * 128 * KP;
  if (shm > 64 * 1024)
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_branch_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  dim3 grid(B * gpi, cdiv(heads, wpb)), block(64 * wpb);
  hipLaunchKernelGGL(attention_branch_split_kernel, grid, block, shm, st, (const split_t*)qkv, tab, B, K, G, heads, scale, KP,
                     (split_t*)out);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// exact-fp32 wavefront kernel (f32 engine; also the reference the MFMA kernel is tested against)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attention_valu_kernel(const T* qkv, SegTable tab, int heads, int causal,
                                                             float scale, int Tcap, T* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int s = blockIdx.x;
  int h = blockIdx.y * wpb + wave;
  const bool live = h < heads;
  if (!live) h = heads - 1;
  const Seg g = load_seg(tab, s);
  const int nk = g.pre_len + g.own_len, nq = g.own_len;
  const int Hd = heads * 64;
  const int Tp = Tcap | 1;
  float* Kt = lds + (size_t)wave * (64 * Tp + 2 * 64 * Tcap + 128);
  float* Vs = Kt + 64 * Tp;
  float* Qs = Vs + 64 * Tcap;
  float* Ps = Qs + 64 * Tcap;  // 128 floats

  for (int t = 0; t < nk; ++t) {
    const long r = key_row(g, t) * (long)(3 * Hd) + h * 64 + lane;
    Kt[lane * Tp + t] = Act<T>::ld(qkv, r + Hd);
    Vs[t * 64 + lane] = Act<T>::ld(qkv, r + 2 * Hd);
  }
  for (int t = 0; t < nq; ++t)
    Qs[t * 64 + lane] = Act<T>::ld(qkv, ((long)g.own_off + t) * (long)(3 * Hd) + h * 64 + lane);
  __syncthreads();

  for (int i = 0; i < nq; ++i) {
    const int vis = causal ? g.pre_len + i + 1 : nk;  // number of visible keys
    float s0 = 0.f, s1 = 0.f;
    const int j0 = lane, j1 = lane + 64;
    const int j0c = j0 < nk ? j0 : 0, j1c = j1 < nk ? j1 : 0;
    const float* qi = Qs + i * 64;
#pragma unroll 4
    for (int d = 0; d < 64; d += 4) {
      const float4 q = *(const float4*)(qi + d);
      s0 += q.x * Kt[(d + 0) * Tp + j0c];
      s0 += q.y * Kt[(d + 1) * Tp + j0c];
      s0 += q.z * Kt[(d + 2) * Tp + j0c];
      s0 += q.w * Kt[(d + 3) * Tp + j0c];
      if (nk > 64) {
        s1 += q.x * Kt[(d + 0) * Tp + j1c];
        s1 += q.y * Kt[(d + 1) * Tp + j1c];
        s1 += q.z * Kt[(d + 2) * Tp + j1c];
        s1 += q.w * Kt[(d + 3) * Tp + j1c];
      }
    }
    s0 = j0 < vis ? s0 * scale : -INFINITY;
    s1 = j1 < vis ? s1 * scale : -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = j0 < vis ? expf(s0 - m) : 0.f;
    const float e1 = j1 < vis ? expf(s1 - m) : 0.f;
    const float sum = wave_sum(e0 + e1);
    Ps[j0] = e0 / sum;
    Ps[j1] = e1 / sum;
    // every LDS region here is private to this wave and a wave's LDS operations execute in order: no barrier,
    // only keep the compiler from moving the reads above the writes (two block-wide barriers per query used
    // to serialise the four heads of a work-group: 30 barriers per sequence at T = 15)
    asm volatile("" ::: "memory");
    float o = 0.f;
    for (int j = 0; j < vis; ++j) o += Ps[j] * Vs[j * 64 + lane];
    if (live) Act<T>::st(out, ((long)g.own_off + i) * (long)Hd + h * 64 + lane, o);
    asm volatile("" ::: "memory");
  }
}

int g_use_mfma_attention = 1;

int launch_attention(int prec, const void* qkv, const SegTable& tab, int max_keys, int heads, int causal, float scale,
                     void* out, hipStream_t st) {
  if (tab.n_seg <= 0) return 0;
  if (max_keys > 96 || max_keys <= 0) {
    snprintf(g_err, sizeof(g_err), "attention: %d keys per segment unsupported (1..96)", max_keys);
    return 1;
  }
  if (prec_is_half(prec) && (g_use_mfma_attention || prec == PREC_F16)) {  // fp16 has no VALU form: MFMA kernel always
    const int KP = ((max_keys + 31) & ~31) + 4;
    int wpb = heads >= 4 ? 4 : (heads >= 2 ? 2 : 1);
    const size_t shmem = (size_t)wpb * 64 * KP * 2;
    dim3 grid(tab.n_seg, cdiv(heads, wpb)), block(64 * wpb);
    if (prec == PREC_F16) hipLaunchKernelGGL(attention_mfma_kernel<f16_t>, grid, block, shmem, st, (const bf16_t*)qkv, tab, heads,
                                             causal, scale, KP, (bf16_t*)out);
    else hipLaunchKernelGGL(attention_mfma_kernel<bf16_t>, grid, block, shmem, st, (const bf16_t*)qkv, tab, heads, causal, scale, KP,
                            (bf16_t*)out);
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (prec == PREC_F16X3 && g_use_mfma_attention) {
    const int KP = ((max_keys + 31) & ~31) + 4;
    int wpb = heads >= 4 ? 4 : (heads >= 2 ? 2 : 1);
    while (wpb > 1 && (size_t)wpb * 2 * 64 * KP * 2 > 150 * 1024) wpb >>= 1;
    const size_t shmem = (size_t)wpb * 2 * 64 * KP * 2;
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_mfma_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)shmem));
    dim3 grid(tab.n_seg, cdiv(heads, wpb)), block(64 * wpb);
    hipLaunchKernelGGL(attention_mfma_split_kernel, grid, block, shmem, st, (const split_t*)qkv, tab, heads, causal, scale, KP,
                       (split_t*)out);
    CZC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const int Tcap = (max_keys + 3) & ~3;
  const size_t per_wave = (size_t)(64 * (Tcap | 1) + 2 * 64 * Tcap + 128) * sizeof(float);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 60 * 1024) wpb >>= 1;
  if (wpb > heads) wpb = heads >= 2 ? 2 : 1;
  const size_t shmem = per_wave * wpb;
  dim3 grid(tab.n_seg, cdiv(heads, wpb)), block(64 * wpb);
  if (prec == PREC_F16X3) {
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_valu_kernel<split_t>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(attention_valu_kernel<split_t>, grid, block, shmem, st, (const split_t*)qkv, tab, heads, causal,
                       scale, Tcap, (split_t*)out);
  } else if (prec == PREC_BF16) {
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_valu_kernel<bf16_t>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(attention_valu_kernel<bf16_t>, grid, block, shmem, st, (const bf16_t*)qkv, tab, heads, causal,
                       scale, Tcap, (bf16_t*)out);
  } else {
    if (shmem > 64 * 1024)
      CZC_HIP_CHECK(hipFuncSetAttribute((const void*)attention_valu_kernel<float>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(attention_valu_kernel<float>, grid, block, shmem, st, (const float*)qkv, tab, heads, causal, scale,
                       Tcap, (float*)out);
  }
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

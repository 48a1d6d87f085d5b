// Fused q/k/v projection + packed-branch attention for the CLIP text tower (HF:clip/modeling_clip.py:309-350
// self-attention block, on the shared-prefix plan of SURVEY.md §3.4): the 3 KB/row q,k,v activations of the
// B*K candidate branches never go to HBM.
//
// Unfused, one layer writes qkv (0.96 GB) and reads it back in the attention kernel; both kernels are bound by
// how many vector-memory instructions a CU can issue (gemm_wreg.hip: 16 stores per 32-row block, attention: 80
// VMEM per group).  Here a work-group owns ONE head:
//   * waves 0-5 hold the head's projection panel in registers as in gemm_wreg.hip -- wave w: part w>>1 (q,k,v),
//     32 of its 64 columns, 512 k = 32 MFMA fragments -- and multiply every 32-row block against it;
//   * a block is one GROUP of an image's candidates (G consecutive branches, <= 32 rows, contiguous), fetched by
//     LDS-DMA into a three-deep ring two blocks ahead; rows past the group read as zero;
//   * their bf16 results go to LDS in the layouts the attention wants: q and k row-major [32][64] with chunks
//     XOR ((row>>1)&7), v as [4 sub-tiles][32 keys][16 dims] for ds_read_b64_tr_b16;
//   * waves 6 and 7 alternate over the blocks and do the attention of the block finished in the previous
//     iteration while the GEMM waves run the next two: S^T = K Q^T for the image's trunk keys and the group's
//     own keys, in-lane softmax, P V by MFMA (same arithmetic, same order as attention_image_kernel), then four
//     16-byte stores per lane of the context rows -- the only HBM writes of the kernel;
//   * one s_barrier per iteration for all eight waves; the attention of a block spans two iterations (scores +
//     softmax, then P V + store), tile buffers are three deep;
//   * the blocks of all images of a work-group form ONE continuous stream: the next image's segment offsets and
//     trunk k/v arrive by LDS-DMA into a second side buffer a few blocks into the current image, so there is no
//     per-image prologue or drain.
// The trunk rows (B*T of them) keep the ordinary path: their q,k,v are written by the GEMM, their attention runs
// in attention_mfma_kernel, and this kernel reads each image's trunk k,v from that buffer once per image.
// Requires head dim 64, hidden 512 (K = 512 panel), trunk <= 32 keys, groups <= 32 rows, K candidates <= 1024.
#include "kernels.h"

namespace czc {

namespace {

constexpr int QA_ROWB = 1024;              // bytes per input row (512 bf16)
constexpr int QA_STAGE = 32 * QA_ROWB;     // 32 KiB
constexpr int QA_RING = 3;
constexpr int QA_TB = 3;                   // q/k/v tile buffers
constexpr int QA_TILE = 4096;              // one 32 x 64 bf16 tile
constexpr int QA_TBS = 3 * QA_TILE;        // q, k, v
constexpr int QA_OFF_TILES = QA_RING * QA_STAGE;
constexpr int QA_META = 5120;               // K + 1 segment offsets (K <= 1024), whole 1 KiB DMA pieces
constexpr int QA_SIDE = 2 * QA_TILE + QA_META;  // per-image side data: trunk k tile, trunk v image, offsets
constexpr int QA_OFF_SIDE = QA_OFF_TILES + QA_TB * QA_TBS;  // two of them: the next image's arrive while this one runs
constexpr int QA_OFF_BIAS = QA_OFF_SIDE + 2 * QA_SIDE;
constexpr int QA_LDS = QA_OFF_BIAS + 6 * 128;
static_assert(QA_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned qa_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ unsigned short qa_bf(float v) { return (unsigned short)(qa_pk(v, 0.f) & 0xffffu); }

typedef __attribute__((ext_vector_type(4))) short qa_tr4_t;
typedef __attribute__((address_space(3))) qa_tr4_t* qa_tr4_lds_t;

__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(const bf16_t* y, int ldy, const bf16_t* W, const float* bias,
                                                          const bf16_t* qkv_trunk, SegTable tab, int B, int K, int G,
                                                          int heads, float scale, bf16_t* ctx, int nslices, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // work-group L runs on XCD L % 8: the eight heads of one image slice share an XCD (and its L2 copy of the rows)
  const int L = blockIdx.x;
  const int xin = L >> 3;
  const int h = xin % heads;
  const int slice = (L & 7) * (gridDim.x / (8 * heads)) + xin / heads;
  if (slice >= nslices) return;
  const int Hd = heads * 64;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem);
  unsigned char* tiles = smem + QA_OFF_TILES;
  auto side_kt = [&](int par) { return smem + QA_OFF_SIDE + par * QA_SIDE; };
  auto side_vt = [&](int par) { return smem + QA_OFF_SIDE + par * QA_SIDE + QA_TILE; };
  auto side_meta = [&](int par) { return (int*)(smem + QA_OFF_SIDE + par * QA_SIDE + 2 * QA_TILE); };
  const bool gemm_wave = wave < 6;
  const int part = wave >> 1, chalf = wave & 1;  // GEMM waves: q/k/v and which 32 of the head's 64 columns

  // ---- DMA plumbing (all eight waves land 4 rows of every block) ----
  const int pitch = ldy * 2;
  int voff[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int r = wave * 4 + ii;
    voff[ii] = r * pitch + ((lane ^ (r & 15)) << 4);
  }
  auto rows_desc = [&](const void* base, long row0, int nrows, int row_bytes) {
    const unsigned long long pa = (unsigned long long)base + (unsigned long long)row0 * row_bytes;
    u32x4_t rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    rs.z = __builtin_amdgcn_readfirstlane((unsigned)(nrows * row_bytes));
    rs.w = 0x00020000u;
    return rs;
  };
  auto dma4 = [&](unsigned dst, int v0, int v1, int v2, int v3, u32x4_t rs, int soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(dst), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(soff)
        : "memory", "scc");
  };
  auto dma1 = [&](unsigned dst, int v, u32x4_t rs, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "v"(v), "s"(rs), "s"(soff)
                 : "memory");
  };
  const int ngroups = (K + G - 1) / G;
  const int NI = slice < B ? (B - 1 - slice) / nslices + 1 : 0;  // images slice, slice + nslices, ...
  const int NB = NI * ngroups;                                   // blocks of this work-group, one continuous stream
  auto group_rows = [&](const int* meta, int gi, int& r0, int& n_own) {
    const int k0 = gi * G, Gc = min(G, K - k0);
    r0 = __builtin_amdgcn_readfirstlane(meta[k0]);
    n_own = min(__builtin_amdgcn_readfirstlane(meta[k0 + Gc]) - r0, 32);
  };

  // A-fragment addresses of the ring (gemm_wreg.hip): logical chunk 2t+half of row l31 -> physical ^ (l31 & 15)
  int va[8];
#pragma unroll
  for (int tl = 0; tl < 8; ++tl) va[tl] = l31 * QA_ROWB + ((((2 * tl + half) ^ (l31 & 15)) & 15) << 4);

  // ---- first image's side data, synchronously; later images' arrive by LDS-DMA while their predecessor runs ----
  int pre_len_par[2] = {0, 0};
  {
    const int sb = B + slice * K;
    int* meta = side_meta(0);
    for (int j = threadIdx.x; j <= K; j += 512) meta[j] = tab.own_off[sb + j];  // own_off has S + 1 entries
    const int pre_off = tab.pre_off[sb], pre_len = tab.pre_len[sb];
    pre_len_par[0] = pre_len;
    if (threadIdx.x < 256) {
      const int row = threadIdx.x >> 3, c8 = threadIdx.x & 7;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (row < pre_len) {
        const bf16_t* src = qkv_trunk + (long)(pre_off + row) * 3 * Hd + h * 64 + c8 * 8;
        kv = *(const uint4*)(src + Hd);
        vv = *(const uint4*)(src + 2 * Hd);
      }
      *(uint4*)(side_kt(0) + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = kv;
      *(uint4*)(side_vt(0) + (c8 >> 1) * 1024 + row * 32 + (c8 & 1) * 16) = vv;
    }
  }
  __syncthreads();
  // next image's side data (issued by wave 6 a few blocks into the current image): offsets = one contiguous
  // run of own_off, trunk k rows as a swizzled row-major tile, trunk v as sub-tile images
  auto side_prefetch = [&](int n_next) {
    const int par = n_next & 1;
    const int sb = B + (slice + n_next * nslices) * K;
    const int pre_off = tab.pre_off[sb], pre_len = tab.pre_len[sb];
    const unsigned base = lds0 + QA_OFF_SIDE + par * QA_SIDE;
    {  // offsets: (K + 1) ints from own_off + sb
      const u32x4_t rm = rows_desc(tab.own_off + sb, 0, K + 1, 4);
      for (int pc = 0; pc * 256 <= K; ++pc) dma1(base + 2 * QA_TILE + pc * 1024, lane * 16, rm, pc * 1024);
    }
    const u32x4_t rt = rows_desc(qkv_trunk, pre_off, min(pre_len, 32), 3 * Hd * 2);
    const int tp = 3 * Hd * 2;
    {  // k tile: piece p = rows 8p..8p+7, lane -> row 8p + (lane>>3), physical chunk lane&7
      int vk[4];
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int row = 8 * pp + (lane >> 3);
        vk[pp] = row * tp + Hd * 2 + h * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
      }
      dma4(base, vk[0], vk[1], vk[2], vk[3], rt, 0);
    }
    {  // v image: piece = sub-tile, lane -> key lane>>1, 16-byte half lane&1
      const int vv = (lane >> 1) * tp + 2 * Hd * 2 + h * 128 + (lane & 1) * 16;
      dma4(base + QA_TILE, vv, vv + 32, vv + 64, vv + 96, rt, 0);
    }
  };

  // block cursors: c = the block of this iteration, d = two ahead (the one whose DMA is issued now)
  int cn = 0, cg = 0, dn = 0, dg = 0, d_slot = 0;
  auto advance = [&](int& n, int& g) { if (++g == ngroups) { g = 0; ++n; } };
  auto issue_all4 = [&]() {  // DMA of block (dn, dg) -> ring slot, this wave's four rows at once
    int r0, nr;
    group_rows(side_meta(dn & 1), dg, r0, nr);
    const u32x4_t rs = rows_desc(y, r0, nr, pitch);
    dma4(lds0 + d_slot * QA_STAGE + wave * (4 * QA_ROWB), voff[0], voff[1], voff[2], voff[3], rs, 0);
  };
  // prologue: blocks 0 and 1
  for (int j = 0; j < 2 && j < NB; ++j) {
    issue_all4();
    advance(dn, dg);
    d_slot = d_slot == QA_RING - 1 ? 0 : d_slot + 1;
  }

  auto iter_wait_barrier = [&](int gb) {
    if (gb < NB) {
      // block gb landed?  behind it in order: block gb+1's four pieces (and, on the attention waves, side data
      // and stores -- ignoring them only makes the wait stricter)
      if (gb + 1 < NB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // block gb, tiles of block gb-1 and pending side data published; slot (gb+2)%3 free
    asm volatile("" ::: "memory");
  };

  if (gemm_wave) {
    // ---- GEMM waves: projection panel (32 output columns x 512 k) and bias slice ----
    u32x4_t wreg[32];
    float* bias_s = (float*)(smem + QA_OFF_BIAS) + wave * 32;
    {
      const int wrow = part * Hd + h * 64 + chalf * 32;
      const unsigned char* wp = (const unsigned char*)W + (long)(wrow + l31) * 512 * 2 + half * 16;
#pragma unroll
      for (int t = 0; t < 32; ++t) wreg[t] = *(const u32x4_t*)(wp + t * 32);
#pragma unroll
      for (int t = 0; t < 32; t += 8)
        asm volatile("" ::"v"(wreg[t]), "v"(wreg[t + 1]), "v"(wreg[t + 2]), "v"(wreg[t + 3]), "v"(wreg[t + 4]), "v"(wreg[t + 5]),
                     "v"(wreg[t + 6]), "v"(wreg[t + 7]));
      if (lane < 32) bias_s[lane] = bias ? bias[wrow + lane] : 0.f;
    }
    int c_slot = 0, c_tb = 0;
    for (int gb = 0; gb < NB + 2; ++gb) {
      iter_wait_barrier(gb);
      if (gb < NB && !(dbg & 2)) {
        // ---------------- projection of block gb: 32 MFMAs, two chains, one DMA piece per two MFMA groups ----------------
        const unsigned char* sA = smem + c_slot * QA_STAGE;
        f32x16_t acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const bool refill = gb + 2 < NB;
        u32x4_t rsn;
        unsigned dstn = 0;
        if (refill) {
          int r0n, nn;
          group_rows(side_meta(dn & 1), dg, r0n, nn);
          rsn = rows_desc(y, r0n, nn, pitch);
          dstn = lds0 + d_slot * QA_STAGE + wave * (4 * QA_ROWB);
        } else {
          rsn = rows_desc(y, 0, 0, pitch);
        }
        u32x4_t fr[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fr[0][k] = *(const u32x4_t*)(sA + va[k & 7] + (k >> 3) * 256);
#pragma unroll
        for (int sgm = 0; sgm < 8; ++sgm) {
          if (sgm < 7) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int t = 4 * (sgm + 1) + k;
              fr[(sgm + 1) & 1][k] = *(const u32x4_t*)(sA + va[t & 7] + (t >> 3) * 256);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            const int t = 4 * sgm + k;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wreg[t]),
                                                           __builtin_bit_cast(bf16x8_t, fr[sgm & 1][k]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wreg[t + 1]),
                                                           __builtin_bit_cast(bf16x8_t, fr[sgm & 1][k + 1]), acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (refill && (sgm & 1) == 0) dma1(dstn + (sgm >> 1) * QA_ROWB, voff[sgm >> 1], rsn, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // lane = row l31, registers = columns 8q + 4half + e of this wave's 32 -> bf16 -> the attention layouts
        unsigned char* tb = tiles + c_tb * QA_TBS + part * QA_TILE;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 b4 = *(const float4*)(bias_s + 8 * qd + 4 * half);
          const uint2 w2 = make_uint2(qa_pk(acc0[4 * qd] + acc1[4 * qd] + b4.x, acc0[4 * qd + 1] + acc1[4 * qd + 1] + b4.y),
                                      qa_pk(acc0[4 * qd + 2] + acc1[4 * qd + 2] + b4.z, acc0[4 * qd + 3] + acc1[4 * qd + 3] + b4.w));
          if (part < 2) {  // q, k: row-major, 16-byte chunk 4*chalf + qd, XOR ((row>>1)&7)
            *(uint2*)(tb + l31 * 128 + (((4 * chalf + qd) ^ ((l31 >> 1) & 7)) << 4) + half * 8) = w2;
          } else {         // v: sub-tile 2*chalf + (qd>>1), key l31, dims 8*(qd&1) + 4*half ..
            *(uint2*)(tb + (2 * chalf + (qd >> 1)) * 1024 + l31 * 32 + (8 * (qd & 1) + 4 * half) * 2) = w2;
          }
        }
      }
      if (gb + 2 < NB) { advance(dn, dg); d_slot = d_slot == QA_RING - 1 ? 0 : d_slot + 1; }
      c_slot = c_slot == QA_RING - 1 ? 0 : c_slot + 1;
      c_tb = c_tb == QA_TB - 1 ? 0 : c_tb + 1;
    }
  } else {
    const int a = wave - 6;
    // attention state carried from the score half to the P V half of a block
    uint4 pf[2][2];
    float inv = 0.f;
    int a_r0 = 0, a_n = 0, a_par = 0;
    int n1 = 0, g1 = -1, n2 = 0, g2 = -1;  // blocks gb-1 and gb-2 (image, group); g < 0: none yet
    for (int gb = 0; gb < NB + 2; ++gb) {
      iter_wait_barrier(gb);
      // a few blocks into image cn its successor's side data starts to arrive (its buffer's last reader was the
      // P V half of image cn-1's last block, two iterations after image cn began)
      if (gb < NB && cg == 3 && cn + 1 < NI) {
        pre_len_par[(cn + 1) & 1] = tab.pre_len[B + (slice + (cn + 1) * nslices) * K];
        if (a == 0) side_prefetch(cn + 1);
      }
      if (gb + 2 < NB) {
        issue_all4();
        advance(dn, dg);
        d_slot = d_slot == QA_RING - 1 ? 0 : d_slot + 1;
      }
      // ---------------- scores + softmax of block gb-1 ----------------
      const int j1 = gb - 1;
      if (j1 >= 0 && j1 < NB && (j1 & 1) == a && !(dbg & 1)) {
        const int* meta = side_meta(n1 & 1);
        const int pre_len = pre_len_par[n1 & 1];
        const int k0 = g1 * G, Gc = min(G, K - k0);
        group_rows(meta, g1, a_r0, a_n);
        a_par = n1 & 1;
        const unsigned char* Qs = tiles + (j1 % QA_TB) * QA_TBS;
        const unsigned char* Ks = Qs + QA_TILE;
        const unsigned char* Kt = side_kt(a_par);
        const int q = min(l31, a_n - 1);
        int ss = 0;
        for (int j = 1; j < Gc; ++j) {
          const int o = meta[k0 + j] - a_r0;
          if (o <= q) ss = o;
        }
        uint4 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(Qs + q * 128 + (((2 * ks + half) ^ ((q >> 1) & 7)) << 4));
        f32x16_t st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned char* kb = t == 0 ? Kt : Ks;
          f32x16_t acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint4 kf = *(const uint4*)(kb + l31 * 128 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]),
                                                          acc, 0, 0, 0);
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
            pf[t][sstep].x = pack2_bf16(e[8 * sstep + 0], e[8 * sstep + 1]);
            pf[t][sstep].y = pack2_bf16(e[8 * sstep + 2], e[8 * sstep + 3]);
            pf[t][sstep].z = pack2_bf16(e[8 * sstep + 4], e[8 * sstep + 5]);
            pf[t][sstep].w = pack2_bf16(e[8 * sstep + 6], e[8 * sstep + 7]);
          }
        }
        sum += __shfl_xor(sum, 32, 64);
        inv = 1.0f / sum;
      }
      // ---------------- P V + store of block gb-2 ----------------
      const int j2 = gb - 2;
      if (j2 >= 0 && j2 < NB && (j2 & 1) == a && !(dbg & 1)) {
        const unsigned char* Vs = tiles + (j2 % QA_TB) * QA_TBS + 2 * QA_TILE;
        const unsigned char* Vt = side_vt(a_par);
        const bool has_trunk = pre_len_par[a_par] > 0;
        const int opitch = Hd * 2;
        const u32x4_t rc = rows_desc(ctx, a_r0, a_n, opitch);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f32x16_t o;
#pragma unroll
          for (int r = 0; r < 16; ++r) o[r] = 0.f;
          const int sub_off = (dt * 2 + (l31 >> 4)) * 1024 + (4 * half + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t == 1 || has_trunk) {
              const unsigned char* vb = (t == 0 ? Vt : Vs) + sub_off;
#pragma unroll
              for (int sstep = 0; sstep < 2; ++sstep) {
                const unsigned char* vp = vb + 16 * sstep * 32;
                const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((qa_tr4_lds_t)(vp)));
                const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((qa_tr4_lds_t)(vp + 256)));
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                            __builtin_bit_cast(bf16x8_t, pf[t][sstep]), o, 0, 0, 0);
              }
            }
          }
          typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
          u32x2_t w[4];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            w[qd].x = pack2_bf16(o[4 * qd] * inv, o[4 * qd + 1] * inv);
            w[qd].y = pack2_bf16(o[4 * qd + 2] * inv, o[4 * qd + 3] * inv);
          }
#pragma unroll
          for (int qa = 0; qa < 2; ++qa) {
            const u32x2_t sx = __builtin_amdgcn_permlane32_swap(w[qa].x, w[qa + 2].x, false, false);
            const u32x2_t sy = __builtin_amdgcn_permlane32_swap(w[qa].y, w[qa + 2].y, false, false);
            const u32x4_t d = {sx.x, sy.x, sx.y, sy.y};
            const unsigned co = (unsigned)(l31 * opitch + (h * 64 + dt * 32 + 8 * (qa + 2 * half)) * 2);
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(d), "v"(co), "s"(rc) : "memory");
          }
        }
      }
      // cursors: blocks gb-1 / gb-2 of the next iteration
      n2 = n1; g2 = g1;
      if (gb < NB) { n1 = cn; g1 = cg; advance(cn, cg); }
    }
  }
}

}  // namespace

int g_use_qkv_attn = 1;
int g_qkv_attn_dbg = 0;  // timing ablations only (results invalid): 1 no attention, 2 no projection

bool qkv_attn_eligible(int H, int heads, int max_keys, int max_own, int K) {
  // K / G >= 8 groups per image: the next image's side data is fetched while the current one runs
  return g_use_qkv_attn && H == 512 && heads == 8 && max_keys <= 32 && max_own > 0 && max_own <= 32 && K <= 1024 &&
         (K + (32 / max_own) - 1) / (32 / max_own) >= 8;
}

// branch rows of a shared-prefix plan: ctx[rows of the B*K branches] from y (LN1 output, all rows); the trunk
// rows' k, v must already be in qkv_trunk (ordinary GEMM over the first n_trunk rows)
int launch_qkv_attn(const void* y, int ldy, const void* Wqkv, const float* bqkv, const void* qkv_trunk, const SegTable& tab, int B,
                    int K, int max_own, int heads, float scale, void* ctx, hipStream_t st) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    CZC_HIP_CHECK(hipGetDevice(&dev));
    CZC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount & ~63;  // whole (8 XCDs x 8 heads) groups
    if (n_cu < 64) n_cu = 64;
    CZC_HIP_CHECK(hipFuncSetAttribute((const void*)qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS));
  }
  const int G = 32 / max_own;
  int nslices = n_cu / heads;
  if (nslices > B) nslices = B;
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(n_cu), dim3(512), QA_LDS, st, (const bf16_t*)y, ldy, (const bf16_t*)Wqkv, bqkv,
                     (const bf16_t*)qkv_trunk, tab, B, K, G, heads, scale, (bf16_t*)ctx, nslices, g_qkv_attn_dbg);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc

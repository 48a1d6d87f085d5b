// Host-side launchers of the gfx950 kernels (one translation unit per kernel family).
#pragma once
#include "common.h"

namespace czc {

// ---- gemm.hip ---------------------------------------------------------------------------
// C[M,N] = A[M,K] . W[N,K]^T  (+bias[N]) (act) (+resid[M,N] fp32)
// A, W are row-major with K contiguous (nn.Linear layout), element type by `prec`
// (bf16_t or float).  out_act (same element type, may be null) and/or out_f32 (may be null)
// are written with leading dimension ldc.  resid may alias out_f32.
struct GemmArgs {
  const void* A = nullptr; int lda = 0;
  const void* W = nullptr; int ldw = 0;
  const float* bias = nullptr;
  const float* resid = nullptr; int ldr = 0;
  void* out_act = nullptr; float* out_f32 = nullptr; int ldc = 0;
  int M = 0, N = 0, K = 0;
  int act = 0;
  int f16 = 0;  // 2-byte operands are IEEE fp16 instead of bf16 (set by launch_gemm from the precision)
  // 2-byte residual stream (round 5; half-precision engines only): `resid` and `out_f32` point to IEEE fp16 rows (ldr / ldc in
  // elements) instead of fp32 ones -- x <- fp16(x + A.W^T + b), the sum formed in fp32 and rounded once.  Served by the tiled
  // kernel, the ping-pong ring kernel and the weight-stationary kernel's residual form; the others refuse it.
  int x16 = 0;
  // x16 layers may also leave per-row LayerNorm partials of the rows they write (round 5: LayerNorm folded into the consumer
  // GEMM): row_part[b * part_ld + m] = float2(sum, sum of squares) of the STORED fp16 values of row m over the 32-column
  // block b (N / 32 blocks), every producer in the same association order (common.h ln_part4 / the epilogues' comments), so
  // that the statistics do not depend on which kernel served the layer.  Null: not written.
  float* row_part = nullptr;
  long part_ld = 0;
  // LayerNorm folded into the weight-stationary K = 512 GEMM (gemm_wreg.hip LNF): A = the raw fp16 residual rows, W = fp16
  // weights with the gain folded in and their rows centred (rowops.hip fold_ln_kernel), bias = b + W.beta, ln_stat = float2
  // (mean, rstd) per row of A
  const float* ln_stat = nullptr;
  // small launches: the statistics are formed inside the consumer from the producer's partials ([16][ln_part_ld] float2, what
  // ln_finalize_kernel would read; same arithmetic, bit-identical) -- only where gemm_wreg_stats_in_kernel(M, N) says so;
  // ln_stat must still be non-null (it selects the folded kernel) but is not read
  const float* ln_part = nullptr;
  long ln_part_ld = 0;
  // Full-row kernel (gemm_rowln, N = 512): out_f32 <- resid + A.W^T + bias and out_act <- LayerNorm(out_f32; ln_gamma,
  // ln_beta, ln_eps) from one launch.
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  float ln_eps = 0.f;
  // Deferred split-K reduce (round 6; BERT fc2 -> LayerNorm): when the launcher splits K, it leaves the slice sums in its slab
  // workspace and describes them here INSTEAD of launching splitk_reduce_kernel -- the caller's next kernel
  // (launch_layernorm_splitk) sums the slabs, adds bias and residual in the reduce kernel's order and normalises the row in one
  // pass.  pending->nslab == 0 after the call: the launch was not split, out_f32 holds the finished rows as usual.
  struct SplitkPending* splitk_pending = nullptr;
};
struct SplitkPending { const float* slabs = nullptr; int nslab = 0; long slab_stride = 0; int ld = 0; };
int launch_gemm(int prec, const GemmArgs& g, hipStream_t st);
bool gemm256_eligible(const GemmArgs& g);
int launch_gemm256(const GemmArgs& g, hipStream_t st);  // gemm256.hip: 256x256 LDS-DMA ring kernels (bf16 / fp16 operands)
extern int g_use_gemm256;  // 0 off, 1 default (ping-pong kernel for fp32 outputs, loader-wave kernel otherwise), 3 / 5 pin one of them
bool gemm256s_eligible(const GemmArgs& g);  // split-fp16 operands, same 256x256 persistent structure
int launch_gemm256s(const GemmArgs& g, hipStream_t st);
extern int g_use_gemm256s, g_gemm256s_min_m;
extern int g_w_dbg;
extern int g_ln_lean;
bool gemm_rowln_eligible(const GemmArgs& g);  // gemm256.hip: 128 x 512 full-row kernel, LayerNorm in the epilogue
int launch_gemm_rowln(const GemmArgs& g, hipStream_t st);
extern int g_rowln_min_m;
extern int g_use_skinny;
extern int g_use_splitk;
extern int g_gemm_deep;
extern int g_gemm_small_tiles;  // 128x128 kernel: two-deep operand prefetch 0 never / 1 for launches of <= one work-group per CU / 2 always
bool gemm_wreg_eligible(const GemmArgs& g);
int launch_gemm_wreg(const GemmArgs& g, hipStream_t st);  // gemm_wreg.hip: weights in registers, K = 512
// the same kernel with a residual epilogue on a 2-byte residual stream (K = 512, N % 256 == 0, x16): x <- fp16(x + A.W^T + b)
bool gemm_wreg_resid_eligible(const GemmArgs& g);
int launch_gemm_wreg_resid(const GemmArgs& g, hipStream_t st);
extern int g_use_wreg;
extern int g_wreg_resid_min_m;
extern int g_wreg_stats_in_kernel;  // 0: the folded consumer always takes its statistics from ln_finalize_kernel
bool gemm_wreg_stats_in_kernel(int M, int N);  // a folded launch of this shape can form its rows' statistics itself (GemmArgs::ln_part)
extern int g_wreg_min_m, g_gemm256_min_m;  // row-count thresholds of the two big-batch GEMM families

// ---- imageproc.hip ------------------------------------------------------------------------
// CLIP image geometry on the device, bit-identical to the PIL path of the reference's CLIPProcessor
size_t imageproc_scratch_bytes(int H, int W, int S);
int launch_clip_preprocess(const unsigned char* rgb_dev, int H, int W, int S, const float* mean, const float* stdv,
                           unsigned char* scratch, float* out, hipStream_t st);

// ---- rowops.hip -------------------------------------------------------------------------
// y = LN(x[row_idx ? row_idx[m] : m]) ; x fp32 [*,H]; outputs optional
// LayerNorm of rows that are still split-K slabs: v = slab_0 + slab_1 + ... (+ bias) (+ resid), in splitk_reduce_kernel's order, then
// launch_layernorm's arithmetic on v: bit-identical to the reduce kernel followed by the LayerNorm kernel (split-fp16 / f32 towers)
int launch_layernorm_splitk(int prec, const SplitkPending& sk, const float* bias, const float* resid, int ldr, const float* gamma,
                            const float* beta, float eps, int M, int H, void* y_act, float* y_f32, hipStream_t st);
int launch_layernorm(int prec, const float* x, const int* row_idx, const float* gamma, const float* beta, float eps,
                     int M, int H, void* y_act, float* y_f32, hipStream_t st);
// the same on a 2-byte residual stream: x16 = fp16 rows [*,512] (H must be 512, half-precision engines), y_act only
int launch_layernorm_x16(int prec, const void* x16, const int* row_idx, const float* gamma, const float* beta, float eps, int M,
                         int H, void* y_act, hipStream_t st);
// BERT embeddings: word + position + token_type(0) -> LN  (HF:bert/modeling_bert.py:98-106)
int launch_bert_embed(int prec, const int* ids, int B, int T, int H, const float* word, const float* pos,
                      const float* type0, const float* gamma, const float* beta, float eps, void* y_act, float* y_f32,
                      hipStream_t st);
// CLIP text embeddings on packed segments: x[own_off[s]+i] = tok[ids[src[s], pos0[s]+i]] + pos[pos0[s]+i]
// (HF:clip/modeling_clip.py:250-254)
int launch_clip_embed(const int* ids, int ids_stride, const int* seg_src, const int* seg_pos0, const int* own_off,
                      const int* own_len, int n_seg, int max_len, int H, const float* tok, const float* pos, float* x,
                      hipStream_t st, int x16 = 0, float* stat = nullptr, float eps = 0.f);  // x16: x is a 2-byte (fp16) residual stream; stat: float2 (mean, rstd) per row
// LayerNorm folded into the consumer GEMM: statistics from the producers' partials, one-time weight preparation
int launch_ln_finalize(const float* part, long part_ld, int nblk, int M, float eps, float* stat, hipStream_t st);
int launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* b, int N, int K, void* Wf16, float* rowsum, float* bf,
                   hipStream_t st);
// vision: im2col of [B,3,S,S] into patches [B*P, 3*p*p] (act type), then assemble cls/pos
int launch_im2col(int prec, const float* pixels, int B, int S, int p, void* out, hipStream_t st);
int launch_vision_assemble(const float* patch_out, int B, int P, int H, const float* cls, const float* pos, float* x,
                           hipStream_t st);
int launch_convert(int prec, const float* src, void* dst, long n, hipStream_t st);  // fp32 -> act type
int launch_act_to_f32(int prec, const void* src, float* dst, long n, hipStream_t st);  // act type -> fp32
// gather rows: dst[m] = src[idx[m]]  (fp32 rows of width H)
int launch_gather_rows_f32(const float* src, const int* idx, int M, int H, float* dst, hipStream_t st);
int launch_gather_rows_bytes(const void* src, const int* idx, int M, int row_bytes, void* dst, hipStream_t st);
// rows b*T+gen_idx
int launch_make_row_index(int* idx, int B, int T, int gen_idx, hipStream_t st);
// idx[s] = off[s] + len[s] - 1
int launch_eos_index(const int* off, const int* len, int n, int* idx, hipStream_t st);
int launch_l2_normalize(const float* x, int M, int D, float* y, hipStream_t st);
// inp[b, gen_idx .. gen_idx+n_mask) = mask_id
int launch_mask_positions(int* inp, int B, int T, int gen_idx, int n_mask, int mask_id, hipStream_t st);
int launch_broadcast_rows_i32(const int* row, int T, int B, int* dst, hipStream_t st);

// ---- attention.hip ------------------------------------------------------------------------
// Segment s = `own_len[s]` rows starting at own_off[s] (queries and keys/values) preceded by
// `pre_len[s]` key/value-only rows starting at pre_off[s] (the shared causal prefix of an image's
// candidates).  own_len == null: fixed-length segments s*fixed_T.. (BERT, vision).
struct SegTable {
  const int* pre_off; const int* pre_len; const int* own_off; const int* own_len;
  int n_seg; int fixed_T;
  // shared-prefix plans: longest branch (own rows of one candidate) of every image [B], or null.  The packed-branch
  // kernels take their packing factor from it PER IMAGE (32 / img_max[b] candidates per 32-query tile), so that an
  // image's attention arithmetic does not depend on which other images share its batch.
  const int* img_max = nullptr;
};
// softmax(q k^T * scale [+causal]) v; qkv [M, 3*heads*64] act type; out [M, heads*64] (own rows only)
int launch_attention(int prec, const void* qkv, const SegTable& tab, int max_keys, int heads, int causal, float scale,
                     void* out, hipStream_t st);
extern int g_use_mfma_attention;
extern int g_use_attention_image;
// bf16 engine, shared-prefix plan (B trunk segments then B*K branch segments): returns -1 when the
// shapes do not fit the packed-branch kernel (caller then uses launch_attention)
int launch_attention_shared(const void* qkv, const SegTable& tab, int B, int K, int max_own, int max_keys, int heads,
                            float scale, void* out, hipStream_t st, int f16 = 0);

// the same for the split engine precision (qkv / out are split_t)
int launch_attention_shared_split(const void* qkv, const SegTable& tab, int B, int K, int max_own, int max_keys, int heads,
                                  float scale, void* out, hipStream_t st);

// ---- topk.hip -----------------------------------------------------------------------------
int launch_softmax_mask_topk(const float* logits, int B, int V, int K, const float* mask, float temperature, int dot_id,
                             int dot_allowed, float* probs, int* idxs, int* cand, hipStream_t st);

// ---- bridge.hip ---------------------------------------------------------------------------
struct BridgeDev {
  int bert_vocab;
  const uint32_t* piece_off;
  const uint8_t* piece_bytes;
  const uint8_t* piece_class;
  const uint8_t* piece_flags;
  const int* byte_sym;
  const int* byte_sym_eow;
  const unsigned long long* hkeys;  // open addressing, ~0 = empty
  const unsigned long long* hvals;  // rank<<32 | out
  unsigned hmask;
  int bos_id, eos_id;
  // per BERT token: the CLIP ids of the token standing alone as a word (all-letter pieces whose byte-level BPE yields at
  // most BR_TOKMAX ids; tok_bpe_len 0 = not tabulated), filled once on the device by launch_bridge_precompute with the
  // same BPE code the per-row kernel runs.  May be null (every chunk then takes the merge loop).
  const int* tok_bpe = nullptr;
  const uint8_t* tok_bpe_len = nullptr;
};
constexpr int BR_TOKMAX = 8;
// fills tok_ids [bert_vocab * BR_TOKMAX] / tok_len [bert_vocab] (device buffers) from the uploaded tables of `bd`
int launch_bridge_precompute(const BridgeDev& bd, int* tok_ids, uint8_t* tok_len, hipStream_t st);
// rows: inp[b,:] with column gen_idx replaced by cand[b,k] (cand==null: rows are taken verbatim,
// n_rows = B, K = 1).  Writes clip_ids [B*K,77], clip_len, and optionally senti/repeats.
// Sentiment score of a row: sum of lexicon[id] over its non-special pieces, or -- when lex_pos [V][5] and
// lex_cls [V] are given -- sum of lex_pos[id][lex_cls[id]] over its word-start pieces.
// POS template for the control score (null tags = no POS score)
struct PosDev {
  const uint8_t* tag_of_token;  // [V]
  const uint16_t* masks;        // [n] accepted-tag bit masks, 0xFFFF = wildcard
  int n;
};
int launch_bridge(const BridgeDev& bd, const int* inp, int B, int T, int gen_idx, const int* cand, int K,
                  const float* lexicon, const float* lex_pos, const uint8_t* lex_cls, int negative, const PosDev& pos, int* clip_ids, int* clip_len, float* senti_raw,
                  float* repeats, int* overflow_flag, hipStream_t st);
// exclusive scan of len[n] -> off[n+1]; totals[0] = sum, totals[1] = max
int launch_scan(const int* len, int n, int* off, int* totals, hipStream_t st);
// Shared-prefix plan for B images x K candidates (segments: B trunks, then B*K branches):
//   p_b = share ? min(min_k LCP(ids[b,k], ids[b,0]), min_k len[b,k] - 1) : 0
//   trunk b: own_len p_b, src row b*K, pos0 0, pre_len 0;  branch (b,k): own_len len-p_b, src b*K+k, pos0 p_b, pre_len p_b
// max_len_out (two device ints, pre-zeroed) receive max len and max branch own_len via atomicMax.
int launch_prefix_plan(const int* clip_ids, const int* clip_len, int B, int K, int share, int* own_len, int* pre_len,
                       int* seg_src, int* seg_pos0, int* max_len_out, int* img_max, hipStream_t st,  // img_max [B]: longest branch per image
                       int* rep = nullptr, int* n_dup_out = nullptr);
// rep (optional, [B*K]): exact de-duplication -- rep[b*K+k] = lowest k' with an identical CLIP id row in image b (k itself for a
// first occurrence); the others get own_len 0 and pool their representative's EOS row; *n_dup_out += their number
// after the scan of own_len: pre_off[trunk] = 0, pre_off[branch (b,k)] = own_off[b];
// eos_idx[b*K+k] = own_off[B+b*K+k] + own_len[B+b*K+k] - 1
int launch_prefix_finish(const int* own_off, const int* own_len, int B, int K, int* pre_off, int* eos_idx,
                         int* n_trunk_rows, hipStream_t st, const int* rep = nullptr);

// ---- combine.hip --------------------------------------------------------------------------
struct CombineArgs {
  const float* text_feat;  // [B*K, D]
  const float* img_n;      // [B, D] L2-normalised
  float logit_scale_exp;
  const float* probs;      // [B,K]
  const int* cand;         // [B,K]
  const float* senti_raw;  // [B,K] or null
  const float* repeats;    // [B,K] or null
  float alpha, beta, gamma;
  int use_senti;           // 0 none, 1 sentiment (softmax(raw/1) + repeat penalty), 2 POS (softmax(raw/0.1))
  int B, K, D;
  float* clip_score; float* clip_ref; float* final_score;  // [B,K] (non-null, engine scratch)
  int* best; float* best_cos;                                // [B]
  int* inp; int T; int gen_idx;                              // write-back target (may be null)
  int* nonfinite = nullptr;                                  // set to 1 when a cosine is not finite (may be null)
  // screen-then-refine: candidates with refine_kind != 0 take their cosine from refine_cos (see combine.hip)
  const int* refine_kind = nullptr;                          // [B,K]
  const float* refine_cos = nullptr;                         // [B,K]
  // guard of the screen-then-refine scores: nonfinite[1] <- max over the re-encoded candidates of |screening error - its
  // estimated mean| (float bits), nonfinite[2] += images where that exceeds refine_guard (0 = no guard)
  float refine_guard = 0.f;
};
// text_feat == null: clip_ref already holds the cosines
int launch_combine(const CombineArgs& a, hipStream_t st);
// screen-then-refine engine (combine.hip): choose the candidates to re-encode / cosines of the re-encoded rows
// gate_h > 0: margin gate of czc_generate (combine.hip); gated[0] += images that passed it, gated[1] += images
int launch_refine_select(const float* clip_score, const float* final_score, int B, int K, float theta, int m_samples, float gate_h,
                         float beta, int need_cos, int* gated, int* kind, int* list, int* count, hipStream_t st);
int launch_refine_cosine(const float* text_feat, const float* img_n, const int* rlist, const int* n_rows_dev, int n_rows_max, int K, int D,
                         float* cos_out, int* nonfinite, hipStream_t st);
// segment plan of the refine pass (bridge.hip): B trunks (prefix lengths of the screening plan) + B x Kr branch slots
// (Kr = *kr_dev = the largest per-image count; an image's chosen candidates first, its other slots empty), i.e. the
// regular shape the packed-branch attention kernels take; rlist / eos_idx are compact (row r = count_off[b] + i)
int launch_refine_plan(const int* clip_len, const int* trunk_len, const int* list, const int* count, const int* count_off,
                       const int* kr_dev, int B, int K, int* own_len, int* pre_len, int* seg_src, int* seg_pos0, int* rlist,
                       int* max_len_out, int* img_max, hipStream_t st);
int launch_refine_finish(const int* own_off, const int* own_len, const int* count, const int* count_off, const int* kr_dev, int B, int K,
                         int* pre_off, int* eos_idx, hipStream_t st);

// ---- czc_internal_hooks (declared below, private to the build): what libconzic_hip_test.so may reach inside this library -----------
// The product library has hidden visibility; the hook library (api_test.hip) gets the launchers it wraps and the
// process-wide kernel-family switches it flips through this table instead of through exported C++ symbols.
constexpr int HOOKS_ABI = 0x0601;
struct Hooks {
  char* (*err_buf)();  // the calling host thread's g_err [512]
  decltype(&launch_gemm) gemm;
  decltype(&launch_gemm_rowln) gemm_rowln;
  decltype(&launch_layernorm) layernorm;
  decltype(&launch_convert) convert;
  decltype(&launch_act_to_f32) act_to_f32;
  decltype(&launch_attention) attention;
  decltype(&launch_softmax_mask_topk) softmax_mask_topk;
  decltype(&launch_bridge_precompute) bridge_precompute;
  decltype(&launch_bridge) bridge;
  decltype(&launch_l2_normalize) l2_normalize;
  decltype(&launch_combine) combine;
  decltype(&launch_layernorm_x16) layernorm_x16;
  decltype(&launch_ln_finalize) ln_finalize;
  decltype(&launch_fold_ln) fold_ln;
  decltype(&gemm_wreg_stats_in_kernel) wreg_stats_ok;
  int *use_gemm256, *use_skinny, *use_splitk, *gemm_deep, *gemm_small_tiles, *use_wreg, *use_gemm256s, *w_dbg, *ln_lean,
      *rowln_min_m, *wreg_min_m, *gemm256_min_m, *use_mfma_attention, *use_attention_image, *wreg_resid_min_m, *wreg_stats_in_kernel,
      *gemm256s_min_m;
};

}  // namespace czc

// The one door through which libconzic_hip_test.so (include/conzic_hip_test.h: kernel-level parity hooks for tests/ and the GEMM
// microbenchmark for tools/) reaches this library's kernel launchers and process-wide kernel-family switches.  NOT part of the
// drop-in boundary and not in include/conzic_hip.h: the symbol is exported (the hook library links against it), its table layout
// is private to the build.  Returns NULL when `abi` is not this build's tag.  Nothing on the product path calls it.
extern "C" __attribute__((visibility("default"))) const void* czc_internal_hooks(int abi);


/*
 * conzic_hip.h -- C ABI of the MI355X-native ConZIC polishing engine (libconzic_hip.so).
 *
 * Drop-in boundary for ONE path of joeyz0z/ConZIC: the per-position polishing step
 *   mask a position -> BERT masked-LM forward -> softmax/top-K -> K candidate captions ->
 *   CLIP text encode -> cosine vs cached image embedding -> alpha/beta(/gamma) fusion -> argmax
 * i.e. the loop bodies of
 *   gen_utils.py:64-81   (sequential_generation), :114-130 (shuffle_generation),
 *   gen_utils.py:160-179 (span_generation),       :209-226 (random_generation),
 *   control_gen_utils.py:43-65, :98-120           (sentiment_*_generation)
 * plus the once-per-image CLIP vision encode (clip/clip.py:48-62).
 *
 * The reference is pure Python over torch/transformers, so it has no FFI of its own; the
 * Python modules gen_utils / control_gen_utils / clip.clip / utils at the repo root keep the
 * reference's call surface and bind these entry points through ctypes.
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every function returns an int
 * status (0 = CZC_OK); czc_last_error() gives the message.  One engine per GPU, one host thread
 * per engine, calls are synchronous on return unless noted.  Pointers named *_host must be host
 * memory; pointers named `src`/`dst`/`pixels` may be host OR device memory (hipMemcpyDefault).
 * The engine owns all device memory it allocates; caller buffers stay caller-owned.
 */
#ifndef CONZIC_HIP_H
#define CONZIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libconzic_hip.so is built with -fvisibility=hidden: the functions declared between this push and its pop are the
 * library's whole dynamic symbol table (tests/test_host_logic.py checks `nm -D`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define CZC_OK 0
#define CZC_ERR_ARG 1
#define CZC_ERR_HIP 2
#define CZC_ERR_STATE 3
#define CZC_ERR_OVERFLOW 4 /* text bridge scratch overflow (row text > CZC_BRIDGE_MAX_BYTES), or a non-finite CLIP cosine: an fp16
                              quantity overflowed in a tower (options resid16 / refine_rows16 = 0 keep fp32 rows; CZC_PREC_SPLIT) */

#define CZC_PREC_BF16 0 /* throughput mode: CLIP towers on bf16 MFMA operands (fp32 accumulate, residual, LN, */
                        /* softmax); the BERT tower runs on split-fp16 MFMA (hi+lo planes, 3 passes, ~22  */
                        /* mantissa bits) because softmax(logits/0.1) amplifies logit error tenfold and   */
                        /* BERT is only 1.4% of the step's FLOPs                                          */
#define CZC_PREC_F32 1  /* f32-input MFMA everywhere (verification mode, ~1e-6 of the CPU reference) */
#define CZC_PREC_ALL_BF16 2 /* bf16 MFMA in every tower (experiments only) */
#define CZC_PREC_SPLIT 3 /* every tower on split-fp16 MFMA (hi+lo fp16 planes, three passes, ~22 mantissa bits,  */
                         /* fp32 accumulate): fp32-class results at 3/16 of the f32-MFMA cost.  The mode for     */
                         /* checkpoints whose logit_scale.exp() is large (clip/clip.py:95-98: x100 for the       */
                         /* published clip-vit-base-patch32), where a bf16 cosine error would be multiplied by   */
                         /* 100 ahead of softmax_K and leave the 1e-3 fused-score budget                          */

#define CZC_PREC_FP16 4 /* CLIP towers on single-pass IEEE fp16 MFMA: the bf16 kernels with the fp16 opcode and converter */
                        /* (same bytes, same speed), 11 significand bits instead of 8 -> ~8x smaller cosine error: inside  */
                        /* the 1e-3 fused-score budget at the published logit scale (x100) where bf16 is out by 2-3x.  */
                        /* BERT on split-fp16 as in CZC_PREC_BF16.  Operands are bounded (LayerNorm outputs, attention    */
                        /* context, quick-GELU outputs); residual stream / accumulators / softmax / statistics stay fp32  */

#define CZC_BRIDGE_MAX_BYTES 512 /* decoded caption text per candidate row */
#define CZC_CLIP_MAX_LEN 77       /* clip/clip.py:71-72 (max_length = 77, truncation) */
#define CZC_PREC_REFINE 5 /* screen-then-refine, for checkpoints with a large logit scale (published CLIP: x100): all K candidates  */
                         /* through the single-pass fp16 text tower (CZC_PREC_FP16 speed), then the candidates that carry the     */
                         /* softmax_K mass (p_k > theta_x / (beta * exp(logit_scale)), theta_x = 2), the two best fused scores and */
                         /* a mass-stratified sample of the rest (it measures the screening tower's mean error) are re-encoded by */
                         /* the split-fp16 tower and the scores are formed from the mixed cosines: a candidate that keeps its     */
                         /* screening cosine moves its fused score by at most theta_x * |its error - the mean|.  Inside            */
                         /* czc_generate a margin gate skips the second pass where the winner is already certain                   */
                         /* (czc_refine_gate_stats).  Vision tower and BERT: split-fp16.  Options "refine_samples" (12) /           */
                         /* "refine_theta_x1000" (2000) / "refine_gate_x1e6" (400) tune it.  Measured figures: the block below.     */
/*
 * BEGIN GENERATED measured
 * Measured on one MI355X, round 6 (generated by tools/refresh_docs.py from profiles/r06_*):
 *   fused score vs the reference, worst over the full-size goldens: CZC_PREC_BF16 2.46e-04, CZC_PREC_REFINE 2.78e-04,
 *   CZC_PREC_SPLIT 4.2e-06, CZC_PREC_F32 7.3e-06 (bar 1e-3);
 *   CZC_PREC_REFINE against CZC_PREC_SPLIT over 2560 more image-steps: worst 3.80e-04, 99.9th percentile 1.3e-04, winners identical
 *   2560 / 2560; guard sample maximum 1.74e-04 against 2.33e-04 over all candidates;
 *   BASELINE configs[2]: 79.4 captions/s (CZC_PREC_BF16), 67.1 (CZC_PREC_REFINE through czc_generate, 76 % of the image-steps gated).
 * END GENERATED measured
 */
#define CZC_MAX_TOPK 1024
#define CZC_MAX_BERT_LEN 64

typedef struct czc_engine czc_engine;

/* Shapes of the three frozen towers the reference loads at demo.py:125-132 / clip/clip.py:11-16. */
typedef struct czc_config {
  /* BertForMaskedLM (HF:bert/modeling_bert.py) */
  int32_t bert_vocab, bert_hidden, bert_layers, bert_heads, bert_inter, bert_max_pos;
  float bert_eps;
  /* CLIP text tower + projection (HF:clip/modeling_clip.py:494-586, :675) */
  int32_t clip_vocab, clip_hidden, clip_layers, clip_heads, clip_inter, clip_max_pos, clip_proj;
  float clip_eps;
  int32_t clip_bos_id, clip_eos_id;
  /* CLIP vision tower + projection (HF:clip/modeling_clip.py:594-656, :674) */
  int32_t vis_hidden, vis_layers, vis_heads, vis_inter, vis_image, vis_patch;
  /* BERT special ids (tokenizer.mask_token_id gen_utils.py:67; vocab['.'] utils.py:55-58; skip set gen_utils.py:75) */
  int32_t pad_id, unk_id, cls_id, sep_id, mask_id, dot_id;
  int32_t precision; /* CZC_PREC_* */
} czc_config;

/* Tables that replace the host string round trip of gen_utils.py:75 (batch_decode) ->
 * clip/clip.py:71-74 (CLIPTokenizer) with an on-device BERT-id -> CLIP-id bridge.
 * Built once on the host from the two tokenizers (conzic_amd/bridge.py). */
typedef struct czc_bridge_tables {
  int32_t bert_vocab;
  const uint32_t* piece_off;  /* [bert_vocab+1] offsets into piece_bytes/piece_class            */
  const uint8_t* piece_bytes; /* UTF-8 of each piece, '##' stripped, NFC + lowercased            */
  const uint8_t* piece_class; /* per byte: bits0-1 class of its char (0 L,1 N,2 other,3 space),  */
                              /*           bit2 = first byte of a char                           */
  const uint8_t* piece_flags; /* [bert_vocab] bit0 special (skipped), bit1 '##' continuation,    */
                              /*              bit2 clean-up removes the space before it          */
  int32_t clip_vocab;
  const int32_t* byte_sym;     /* [256] CLIP id of a byte inside a word                          */
  const int32_t* byte_sym_eow; /* [256] CLIP id of a byte that ends a word ('</w>' suffix)       */
  int32_t n_merges;
  const int32_t* merge_left;  /* [n_merges] BPE merges in rank order: (left,right) -> out        */
  const int32_t* merge_right;
  const int32_t* merge_out;
  int32_t bos_id, eos_id;
} czc_bridge_tables;

/* Hyper-parameters of one generate call: demo.py:55-60 / gen_utils.py:289-292 keyword arguments. */
typedef struct czc_hyper {
  float alpha;       /* weight of BERT fluency probs          (gen_utils.py:77)              */
  float beta;        /* weight of CLIP softmax_K score        (gen_utils.py:77)              */
  float gamma;       /* weight of sentiment softmax_K         (control_gen_utils.py:59)      */
  float temperature; /* lm_temperature                         (gen_utils.py:43-44)           */
  int32_t control;   /* 0 caption path (gen_utils.py:77); 1 sentiment: + gamma*softmax_K(senti) + 0.1*(1-e^repeats) */
                     /* (control_gen_utils.py:59); 2 POS: + gamma*softmax_K(acc/0.1) (control_gen_utils.py:165-169)  */
  int32_t negative;      /* sentiment_ctl == "negative" (sentiments_classifer.py:31-32)         */
} czc_hyper;

/* Outputs of one position-step at parity granularity (every pointer optional / may be NULL).
 * Shapes use B images, K = top_k.  Host or device memory. */
typedef struct czc_step_out {
  float* probs;      /* [B,K] masked softmax probs, descending   (gen_utils.py:45-47) */
  int32_t* idxs;     /* [B,K] their vocab ids                     (gen_utils.py:47)    */
  int32_t* cand_ids; /* [B,K] idxs * token_mask[idxs]             (gen_utils.py:72)    */
  int32_t* clip_ids; /* [B*K, CZC_CLIP_MAX_LEN] bridged CLIP ids  (clip/clip.py:71-74) */
  int32_t* clip_len; /* [B*K] tokens incl. BOS/EOS                                     */
  float* clip_score; /* [B,K] softmax_K(cos * exp(logit_scale))   (clip/clip.py:97)    */
  float* clip_ref;   /* [B,K] cosine                              (clip/clip.py:98)    */
  float* senti_raw;  /* [B,K] control score: sentence sentiment (sentiments_classifer.py:46) or POS-template */
                     /*       match fraction (POS_classifier.py:17-29), by czc_hyper.control                  */
  float* repeats;    /* [B,K] repeat count - 1                    (control_gen_utils.py:53)    */
  float* final_score;/* [B,K] fused score                         (gen_utils.py:77 / control_gen_utils.py:59) */
  int32_t* best;     /* [B]   argmax_K (first max)                (gen_utils.py:78)    */
  float* best_cos;   /* [B]   cosine of the winner                (gen_utils.py:80)    */
  float* logits;     /* [B,V] BERT logits of the masked row       (gen_utils.py:42)    */
} czc_step_out;

/* ---- lifecycle ---------------------------------------------------------------------- */
int czc_create(const czc_config* cfg, int device_id, czc_engine** out_engine);
int czc_destroy(czc_engine* e);
/* A second engine on the same GPU that SHARES `parent`'s resident weights (no copy, no reload) and owns everything
 * else: stream, workspace, image embeddings, token mask / bridge / lexicon / POS tables and the control callback (set
 * them on the replica as on the parent), options (copied at creation), profile.  Images are independent (gen_utils.py:64-81 has no
 * cross-image term), so a host that drives parent and replica(s) from separate threads on disjoint sub-batches gets
 * the same captions image for image while their kernels overlap on the GPU.  `parent` must be finalized and must
 * outlive its replicas; czc_destroy(replica) frees only what the replica owns. */
int czc_replicate(czc_engine* parent, czc_engine** out_engine);
const char* czc_last_error(const czc_engine* e); /* e may be NULL: last create error */
int czc_version(void);

/* ---- frozen state (replaces from_pretrained at demo.py:125-126, clip/clip.py:12-16) ---- */
/* One call per state-dict entry (names and shapes: SURVEY.md §8b).  dtype must be 0 (fp32).
 * `src` may be a device pointer, which is how ranks > 0 hand over weights they received
 * through the RCCL broadcast (bench.py / conzic_amd/dist.py). */
int czc_load_tensor(czc_engine* e, const char* name, int dtype, int ndim, const int64_t* shape, const void* src);
/* After the last czc_load_tensor: checks completeness, fuses q/k/v, ties the MLM decoder to the
 * word embeddings (HF:bert/modeling_bert.py:910-913), converts GEMM operands to the engine precision. */
int czc_finalize_weights(czc_engine* e);
/* token_mask of demo.py:135-143: fp32 [V]; the '.' entry is overridden per step (utils.py:53-59). */
int czc_set_token_mask(czc_engine* e, const float* mask, int vocab);
int czc_set_bridge(czc_engine* e, const czc_bridge_tables* t);
/* Per-BERT-token sentiment score (stand-in for sentiments_classifer.py:9-33). */
int czc_set_lexicon(czc_engine* e, const float* lexicon, int vocab);

/* The same score keyed the way the reference scores (sentiments_classifer.py:14-30: per WORD and coarse POS class,
 * mean of pos_score - neg_score over the word's SentiWordNet synsets): table fp32 [V][5] over the classes
 * 0 '' | 1 n | 2 v | 3 a | 4 r, addressed by a word's FIRST piece ('##' continuations add nothing), and
 * class_of_token uint8 [V] = the class a context-free tagger gives that token (host pointer).  Where nltk and its
 * corpora exist, conzic_amd/sentiment.py fills both from SentiWordNet; table == NULL returns to czc_set_lexicon. */
int czc_set_lexicon_pos(czc_engine* e, const float* table, const uint8_t* class_of_token, int vocab);

/* POS control (control_gen_utils.py:136-195 / POS_classifier.py:6-31): per-BERT-token universal-tagset id
 * (stand-in for nltk.pos_tag) and the template as one bit mask of accepted tag ids per word
 * position (0xFFFF = the reference's "" wildcard; bit 15 = the "" tag that pads a too-short sentence matches as well,
 * POS_classifier.py:19-25 with a string entry); n_template <= 32. */
int czc_set_pos(czc_engine* e, const uint8_t* tag_of_token, int vocab, const uint16_t* template_masks, int n_template);

/* Control scores from the host instead of the tables above: the reference scores every candidate SENTENCE with nltk
 * (sentiments_classifer.py:9-33: word_tokenize -> context-dependent pos_tag -> SentiWordNet; POS_classifier.py:12-29:
 * pos_tag(tagset="universal") against the template), which the per-token tables can only approximate.  With a callback
 * set, every step with czc_hyper.control != 0 calls it once, AFTER the step's CLIP text tower has been queued on the
 * engine's stream and BEFORE the combine kernel that consumes the scores: the host scores while the GPU encodes the
 * candidates (the reference runs the two one after the other, control_gen_utils.py:56-58), so the scorer's wall time only
 * shows where it exceeds the tower's:
 *   inp   int32 [B,T]  the current rows, [MASK] at column gen_idx          (control_gen_utils.py:49-50)
 *   cand  int32 [B,K]  the candidate ids, idxs * token_mask[idxs]           (control_gen_utils.py:52)
 *   scores fp32 [B,K]  out: the raw control score of row b with cand[b][k] at gen_idx -- the sentence sentiment AFTER the
 *                      sign flip of "negative" (sentiments_classifer.py:30-32), or the template match fraction
 *                      (POS_classifier.py:17-29); softmax_K / gamma / the repeat penalty stay in the combine kernel
 * and a non-zero return fails the step with CZC_ERR_STATE.  All pointers are host memory owned by the engine and valid
 * during the call only; the call happens on the thread that called czc_step / czc_generate.  fn == NULL removes it. */
typedef int (*czc_control_fn)(void* user, const int32_t* inp, const int32_t* cand, int B, int T, int K, int gen_idx,
                              float* scores);
int czc_set_control_callback(czc_engine* e, czc_control_fn fn, void* user);

/* ---- once per image: clip/clip.py:48-62 after the image processor ------------------------ */
/* pixels fp32 [B,3,S,S] -> un-normalised image_embeds [B,proj] (out may be NULL).  The engine
 * keeps the L2-normalised embeds resident for the following step/generate calls (the north
 * star's "encode once per image and cache"). */
int czc_encode_images(czc_engine* e, const float* pixels, int B, float* out_embeds);
/* The image processor itself (clip/clip.py:55-56 -> CLIPProcessor -> HF CLIPImageProcessor, PIL backend): RGB uint8
 * [height][width][3] (host or device pointer) -> bicubic resize of the shorter side to S (Pillow's 8-bit resampler,
 * bit-exact) -> centre crop SxS -> /255 -> (x - mean[c]) / std[c] -> CHW fp32, written to slot `slot` of the engine's
 * staged pixel batch on the device (and to pixels_out [3,S,S], host or device, when not NULL).  mean/std: 3 floats. */
int czc_preprocess_u8(czc_engine* e, const uint8_t* rgb, int height, int width, const float* mean, const float* stdv,
                      int slot, float* pixels_out);
/* czc_encode_images over staged slots 0..B-1 (no host round trip of the pixels). */
int czc_encode_staged(czc_engine* e, int B, float* out_embeds);
/* Alternative: hand over image_embeds computed elsewhere (fp32 [B,proj], un-normalised). */
int czc_set_image_embeds(czc_engine* e, const float* embeds, int B);

/* clip/clip.py:64-84 after tokenisation: CLIP ids int32 [n, CZC_CLIP_MAX_LEN] (right-padded) and
 * lengths (tokens incl. BOS/EOS) -> un-normalised text_embeds fp32 [n, proj]. */
int czc_encode_text(czc_engine* e, const int32_t* clip_ids, const int32_t* clip_len, int n, float* out_embeds);
/* clip/clip.py:86-98 `compute_image_text_similarity_via_embeddings`: image_embeds [B, proj] and text_embeds [B*K, proj]
 * (both as returned by czc_encode_images / czc_encode_text, un-normalised) -> clip_score [B, K] = softmax over an image's K
 * texts of cos * exp(logit_scale of the loaded checkpoint), clip_ref [B, K] = the cosines.  K <= CZC_MAX_TOPK.  (czc_step /
 * czc_generate never call it: the same arithmetic runs fused inside their score-combine kernel.) */
int czc_similarity(czc_engine* e, const float* image_embeds, const float* text_embeds, int B, int K, float* clip_score,
                   float* clip_ref);

/* ---- the hot path ------------------------------------------------------------------------ */
/* Parity granularity: one position-step (gen_utils.py:66-81) on `inp` int32 [B,T] (in/out, host
 * or device).  gen_idx = seed_len + position; n_mask = how many consecutive positions starting
 * at gen_idx are overwritten with [MASK] before the BERT forward (1 normally, 2 for the first
 * step of a span, 0 = re-use the previous forward, gen_utils.py:164-166); dot_allowed = the
 * update_token_mask rule (utils.py:53-59).  n_mask = 0 needs a previous czc_step / czc_generate forward of the same
 * [B,T]; after an n_mask = 1 step only row gen_idx of that forward exists (option "bert_prune"), so n_mask = 0 at
 * another gen_idx returns CZC_ERR_STATE -- the reference only re-uses a forward after masking two positions
 * (gen_utils.py:164-166), which is n_mask = 2 here and keeps every row. */
int czc_step(czc_engine* e, int32_t* inp, int B, int T, int gen_idx, int n_mask, int dot_allowed, int top_k,
             const czc_hyper* hp, const czc_step_out* out);

/* Throughput granularity: a whole *_generation call with no host round trips except one 8-byte
 * size read per step.  init_ids int32 [T] = `[CLS] prompt [MASK]xL [SEP]` (utils.py:46-51),
 * seed_len = len(prompt.split())+1 (gen_utils.py:56), positions[n_steps] / n_mask[n_steps] as in
 * czc_step, snapshot_every = steps per bookkeeping snapshot (gen_utils.py:82-92).
 * out_ids int32 [n_steps/snapshot_every, B, T], out_cos fp32 [n_steps/snapshot_every, B]
 * (the winner cosine of the snapshot's last step, gen_utils.py:80-81,92). */
int czc_generate(czc_engine* e, int B, int T, int L, int seed_len, const int32_t* init_ids_host, int top_k,
                 int n_steps, const int32_t* positions_host, const int32_t* n_mask_host, int snapshot_every,
                 const czc_hyper* hp, int32_t* out_ids, float* out_cos);

/* Engine options (all are exact work reductions / kernel choices; results agree within the engine precision):
 *   "share_prefix"    (1) encode the causal prefix common to an image's K candidates once per step instead of K
 *                         times
 *   "dedup"           (1) with "share_prefix": candidates of one image whose CLIP id rows are identical (all the candidates the
 *                         token mask turned into [PAD], gen_utils.py:72-75) are encoded once and share one feature
 *                         (czc_dedup_stats)
 *   "bert_prune"      (1) n_mask == 1 steps: the last BERT layer behind its attention (out-projection, LayerNorms, MLP) on the
 *                         masked row of every sequence only -- the one row the MLM head reads (gen_utils.py:69)
 *   "bert_fuse_splitk_ln" (1) BERT fc2 (split along K at every row count above 32): the LayerNorm kernel behind it sums the slice slabs,
 *                         bias and residual itself instead of a reduce kernel followed by the LayerNorm kernel; bit-identical
 *   "pack_branches"   (1) attention of the branch rows with G candidates packed per 32-query MFMA tile
 *   "pool_last_layer" (1) last CLIP-text layer: out-projection + MLP on the EOS rows only
 *   "fold_ln"         (1) with "resid16": the LayerNorms inside the CLIP-text stack folded into the q/k/v and fc1 GEMMs (they
 *                         multiply x itself on the fp16 MFMA by weights whose rows carry the gain and are centred, and scale with
 *                         the row's rstd, which comes from partial sums the producer GEMMs leave): no LayerNorm kernel and no
 *                         normalised copy of the rows; 0 = LayerNorm kernels
 *   "fuse_ln"         (1) fp32-residual CLIP-text tower (fp16 / refine engines; bf16 with resid16 = 0) at >= 8192 packed rows: the out-projection runs as a full-row
 *                         kernel that also emits LN2 of its result (no LayerNorm pass for it); 2 = fc2 -> the next
 *                         layer's LN1 as well (measured slower), 0 = off
 *   "resid16"         (1) residual stream of the CLIP-text tower as IEEE fp16 rows in HBM (fp32 accumulate, bias and residual
 *                         add; one rounding per update): 1 = the bf16 engine (CZC_PREC_BF16), 2 = the single-pass fp16 tower too
 *                         (outside its validated error budget: experiments), 0 = fp32 rows everywhere.  With it the
 *                         out-projection runs on the weight-stationary kernel ("fuse_ln" then has nothing to fuse)
 *   "refine_theta_gen_x1000" (4000): the same threshold inside czc_generate (ids and winner cosines are its output, not the K scores)
 *   "refine_samples" (12) / "refine_samples_step" (24), "refine_theta_x1000" (2000): CZC_PREC_REFINE selection -- strata of the
 *                         mass-stratified sample inside czc_generate / in czc_step, and the softmax_K mass threshold
 *                         theta = value / 1000 / (beta * exp(logit_scale))
 *   "refine_guard_x1e6" (200): trip point of czc_refine_guard, in units of 1e-6 of cosine
 *   "refine_gate_x1e6" (400): cosine-error bound delta of the margin gate of czc_generate (czc_refine_gate_stats), 0 = off
 *   "refine_rows16"   (1) CZC_PREC_REFINE inside czc_generate: the screening pass on the 2-byte residual stream with the folded
 *                         LayerNorms (the "resid16" + "fold_ln" tower form on fp16 operands); gate bound and guard trip point
 *                         are multiplied by "refine_rows16_x1000" / 1000 (1750) while it is on and the selection's mass threshold
 *                         divided by it.  czc_step is not affected */
int czc_set_option(czc_engine* e, const char* name, int value);
/* Reads an option back (same names and units as czc_set_option), plus three read-only derived values:
 *   "refine_guard_generate_x1e6" / "refine_gate_generate_x1e6": the guard's trip point / the margin gate's bound in force inside
 *       czc_generate (the base values x "refine_rows16_x1000" / 1000 while its screening pass runs on fp16 rows);
 *   "has_folded_ln_weights": 1 when czc_finalize_weights built the folded-LayerNorm operands (it does where an option can use
 *       them: bf16 engine with resid16 >= 1, refine engine with refine_rows16, fp16 engine with resid16 = 2; turning such an
 *       option on afterwards returns CZC_ERR_STATE). */
int czc_get_option(czc_engine* e, const char* name, int* value);

/* ---- measurement ------------------------------------------------------------------------- */
/* HIP-event timing of kernel classes on the engine's own stream (bench.py roofline leg).  on: 0 off, 1 an event pair
 * around every kernel class (3 % slower at B = 256, 37 % at B = 1), 2 only around the CLIP-text tower's classes: its
 * linear layers "gemm_clip_text" / "gemm_clip_refine" (the roofline family), "attention_clip_text" (flops = 4 * hidden *
 * causal (query, key) pairs per layer) and "rowops_clip_text" (embedding, LayerNorm, gathers).
 * kind: those four | "gemm_bert" | "gemm_vision" | "attention" (BERT, vision) | "rowops" (BERT, vision, head) | "topk" |
 * "bridge" | "combine" */
int czc_profile_enable(czc_engine* e, int on);
int czc_profile_reset(czc_engine* e);
int czc_profile_get(czc_engine* e, const char* kind, double* total_ms, int64_t* launches, double* flops);
/* Start / end (ms) of every launch of class `kind` since czc_profile_reset(e), on the clock whose zero is
 * czc_profile_reset(ref) (ref == e for a single engine).  For engines that ran concurrently the host takes the union
 * of their intervals = the time the GPU spent on that class.  Fills at most `cap` pairs; *n = number available. */
int czc_profile_intervals(czc_engine* e, czc_engine* ref, const char* kind, double* start_ms, double* end_ms, int cap,
                          int* n);
int czc_sync(czc_engine* e);
/* counters of the last generate/step: rows pushed through the CLIP text tower etc. */
int czc_stats(czc_engine* e, int64_t* clip_rows, int64_t* clip_seqs, int64_t* bert_rows, int64_t* steps);
/* Exact de-duplication (option "dedup", default 1): of the *clip_seqs candidate sequences the text tower was asked for since
 * czc_profile_reset, *dedup_seqs had a CLIP id row identical to an earlier candidate of the same image (every candidate the
 * token mask turned into [PAD] decodes to the same caption without the word, gen_utils.py:72-75) and were not encoded
 * again: they take their representative's feature bit for bit (f32 engine: exactly what their own rows would have produced;
 * MFMA engines: within the attention tile's packing noise of it, tests/test_step_gpu.py::test_dedup_is_exact). */
int czc_dedup_stats(czc_engine* e, int64_t* dedup_seqs, int64_t* clip_seqs);
/* CZC_PREC_REFINE engines: candidate sequences / packed rows re-encoded by the split-fp16 tower since czc_profile_reset
 * (of the clip_seqs / clip_rows the screening pass saw); zero for the other precisions. */
int czc_refine_stats(czc_engine* e, int64_t* refine_seqs, int64_t* refine_rows);
/* CZC_PREC_REFINE engines, runtime guard of the 1e-3 bound: candidates that keep their screening (single-pass fp16) cosine
 * carry its error minus the estimated mean, and move their fused score by at most theta_x * |that| (theta_x = 2).  Every step
 * that runs the full selection records, over the ~20 candidates per image it re-encodes exactly, the largest
 * |screening error - mean| (*max_dev, since the last reset) and counts the image-steps where it exceeds option
 * "refine_guard_x1e6" * 1e-6 (default 200) in *tripped.  The candidates that are NOT re-encoded reach at most twice the sample
 * maximum on the validated towers (tools/refine_validate.py prints the ratio: 1.15-1.54 on ten plain weight draws, 2.02 on a x6
 * outlier tower), so below the trip point a KEPT candidate moves its own score by at most theta_x * 2 * 2.0e-4 = 8e-4.  That is
 * not the whole error of czc_step: the sample's estimate of the mean screening error is itself uncertain, what is left of the
 * mean scales the softmax denominator, and every RE-ENCODED candidate moves by beta * p_k * exp(logit_scale) * (kept mass) *
 * (error of the mean) -- the largest term on peaky images, measured not bounded: worst |d final_score| against the all-split
 * engine over eleven weight draws 4.5e-4 .. 9.2e-4 with 12 strata, 2.8e-4 .. 5.4e-4 with the 24 czc_step uses since round 6
 * ("refine_samples_step"; czc_generate keeps 12: only the winner matters there).  A checkpoint whose activations the fp16 tower
 * carries worse trips the guard, and
 * conzic_amd/runtime.py then repeats the call on the all-split engine (CZC_REFINE_GUARD=rerun | warn | off).  Inside czc_generate
 * gated image-steps re-encode nothing and are not measured: the audit steps (czc_refine_gate_stats) are. */
int czc_refine_guard(czc_engine* e, int reset, float* max_dev, int64_t* tripped);
/* CZC_PREC_REFINE engines, margin gate of czc_generate: a whole *_generation call returns the winner's id of every step and
 * the winner's cosine at the snapshot steps (gen_utils.py:78-81, :92), not the K fused scores.  An image-step whose screening
 * (single-pass fp16) winner stays the winner under EVERY assignment of cosine errors |d_k - common| <= delta (delta = option
 * "refine_gate_x1e6" * 1e-6, default 400 = 2x the largest deviation measured over 256 k candidates; the check is the
 * adversarial one: winner's logit down, challenger's up, the rest both ways) needs no second pass for its id; at a snapshot
 * step its winner alone is re-encoded, for the cosine the call returns.  Image-steps that fail the gate take the full selection,
 * and so does every image at the snapshot step of every fourth sweep (the first included) -- whether or not the caller passes
 * out_cos -- and, in a call too short to reach a snapshot step, at its last step: the audit steps on which the guard above keeps
 * measuring the screening tower.  The gate is the guard's dependant: with "refine_guard_x1e6" = 0 nothing polices the bound the
 * gate rests on and nothing is gated.
 * czc_step never gates: all K scores are its output and all of them are refined.  *gated of *image_steps since
 * czc_profile_reset. */
int czc_refine_gate_stats(czc_engine* e, int64_t* gated, int64_t* image_steps);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CONZIC_HIP_H */

// Engine: frozen-weight residency, workspace, the three tower forwards and the polishing
// step/generate loops behind the C ABI of include/conzic_hip.h.
//
// One engine = one GPU = one HIP stream.  Weights are resident for the engine's lifetime
// (bf16 GEMM operands + fp32 LayerNorm/bias/embedding tables: ~0.75 GB of the 288 GB HBM);
// activations live in a grow-only workspace sized by the packed CLIP row count of the step.
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>

#include "../../include/conzic_hip.h"
#include "kernels.h"
#include "bridge_hash.h"

namespace czc {
thread_local char g_err[512] = {0};
}
using namespace czc;

namespace {

struct Tensor {
  float* p = nullptr;
  std::vector<int64_t> shape;
  size_t numel = 0;
};

struct LayerW {
  void* qkv_w = nullptr; float* qkv_b = nullptr;
  void* o_w = nullptr; float* o_b = nullptr;
  float *ln1_g = nullptr, *ln1_b = nullptr;
  void* fc1_w = nullptr; float* fc1_b = nullptr;
  void* fc2_w = nullptr; float* fc2_b = nullptr;
  float *ln2_g = nullptr, *ln2_b = nullptr;
  // LayerNorm folded into the K = 512 GEMMs (bf16 engine's CLIP-text tower on the 2-byte residual stream): fp16 weights with
  // the gain folded in and their rows centred, bias + W.beta (rowops.hip fold_ln_kernel); null where the fold is not used
  void *qkv_wf = nullptr, *fc1_wf = nullptr;
  float *qkv_bf = nullptr, *fc1_bf = nullptr;
};

struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct ProfKind {
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;
  double flops = 0;
  int64_t launches = 0;
};

}  // namespace

struct czc_engine {
  czc_config cfg;
  int dev = 0;
  hipStream_t st = nullptr;
  bool shares_weights = false;  // czc_replicate: weights belong to the parent engine (which must outlive this one)
  hipEvent_t prof_ref = nullptr;  // recorded by czc_profile_reset: time zero of czc_profile_intervals
  char err[512] = {0};
  bool finalized = false;
  size_t esz = 2;  // bytes per CLIP activation element
  int pb = PREC_F32, pc = PREC_BF16;  // BERT / CLIP tower precisions
  int pv = PREC_BF16;                 // CLIP vision tower precision (= pc, except split-fp16 in the screen-then-refine engine)
  size_t eb = 4;   // bytes per BERT activation element
  // Screen-then-refine (CZC_PREC_REFINE): every candidate goes through the single-pass fp16 text tower; the candidates
  // that carry the softmax_K mass (+ a mass-stratified sample of the rest) are re-encoded by the split-fp16 tower
  // (ctext_x / tproj_wx) and the final scores are formed from the mixed cosines (combine.hip)
  bool refine = false;
  // mass threshold theta = refine_theta_x / (beta * exp(logit_scale)): 0.01 at beta 2, scale 100.  A candidate that keeps its
  // screening cosine moves its fused score by at most theta_x * |d_k - mean|: round 5 lowered theta_x from 4 to 2, so the
  // largest deviation measured (2.1e-4 over 256 k candidates) gives 4.2e-4 instead of 8.4e-4 of the 1e-3 bar
  float refine_theta_x = 2.0f;
  // czc_generate returns ids and winner cosines, not the K scores: its selection keeps round 3's threshold (4.0: fewer mass
  // carriers to re-encode; winners identical to the all-split engine on every validated image-step), the bound above is czc_step's
  float refine_theta_gen = 4.0f;
  bool in_generate = false;
  int refine_samples = 12;      // strata of the sample among the candidates below the threshold, inside czc_generate
  // czc_step (all K fused scores are its output): twice the strata.  The sample's job is the mass-weighted MEAN screening error of
  // the candidates that keep their screening cosine; what is left of it after the correction scales the softmax denominator, and
  // through it the score of every re-encoded candidate by beta * p_k * exp(logit_scale) * (kept mass) * (error of the mean) -- the
  // largest term of czc_step's error on peaky images, and one the guard does not see.  Round 6, eleven weight draws x 2560
  // image-steps: worst fused-score difference 4.5e-4 .. 9.2e-4 with 12 strata, 2.8e-4 .. 5.4e-4 with 24 (profiles/r06_refine_samples_sweep.jsonl)
  int refine_samples_step = 24;
  // guard: the ~20 candidates an image re-encodes show their own |screening error - mean|; the candidates that keep their
  // screening cosine reach at most GUARD_RATIO = 2 times that sample maximum (fitted: 1.75 worst over 1280 image-steps), so
  // theta_x * 2 * sample_dev <= 1e-3 holds while sample_dev <= 2.5e-4 at theta_x = 2; trip point 0.8 of that
  float refine_guard_dev = 2.0e-4f;
  float guard_max_dev = 0.f; int64_t guard_trips = 0;   // since the last czc_refine_guard(reset = 1)
  // margin gate (czc_generate only; combine.hip refine_select_kernel): an image whose screening winner survives every
  // assignment of cosine errors |d_k - common| <= refine_gate_delta skips the second pass (its winner alone is re-encoded at
  // the snapshot steps, for the cosine the caller reads).  0 = off.  Default 4e-4 = 2x the largest |error - mean| measured
  // over 256 k candidates (2.1e-4), 2.7x the guard's sample maximum (1.5e-4)
  float refine_gate_delta = 4.0e-4f;
  // czc_generate's screening pass on the 2-byte residual stream with the LayerNorms folded into its GEMMs (the bf16 engine's
  // tower form, fp16 operands).  fp16 rows move its cosines: over the validation's candidates the guard's sample maximum
  // inside czc_generate grows from 1.3-1.6e-4 to 2.3-2.8e-4 (x1.72-1.76; the true maximum over all candidates of the
  // czc_step series from 2.1e-4 to 3.0e-4), so while it is on the three quantities that are bounds on that deviation move
  // together by refine_rows16_factor = 1.75: gate bound 7e-4, guard trip point 3.5e-4 (= bound / 2, as before) and the
  // selection's mass threshold theta_gen / 1.75 (theta * deviation, what a kept screening cosine can move a score by, stays).
  // czc_step keeps fp32 rows: all K fused scores are its output and fp16 rows would take its worst one from 6.3e-4 to 1.3e-3.
  int refine_rows16 = 1;
  float refine_rows16_factor = 1.75f;
  bool gate_now = false, gate_need_cos = true;  // set per step by czc_generate; czc_step never gates (all K scores are its output)
  int64_t stat_gated = 0, stat_gate_images = 0;

  std::map<std::string, Tensor> w;
  std::vector<LayerW> bert, ctext, cvis, ctext_x;
  // extra processed weights
  void* mlm_dense_w = nullptr; void* decoder_w = nullptr; void* tproj_w = nullptr; void* vproj_w = nullptr;
  void* patch_w = nullptr; void* tproj_wx = nullptr;

  std::map<std::string, Buf> ws;
  float* d_mask = nullptr; int mask_vocab = 0;
  float* d_lex = nullptr;
  float* d_lex_pos = nullptr; uint8_t* d_lex_cls = nullptr;  // (word-start piece, coarse POS class) keyed table [V][5] + class per token
  uint8_t* d_pos_tags = nullptr; uint16_t* d_pos_masks = nullptr; int pos_n = 0;
  BridgeDev bd; bool has_bridge = false;
  std::vector<void*> bridge_allocs;
  float* d_img_n = nullptr; int img_B = 0;
  float logit_scale_exp = 1.f;
  double plan_pairs = 0;    // causal (query, key) pairs of the text-tower plan about to run (totals[7]): attention FLOPs = 4 * H * pairs per layer
  int* h_totals = nullptr;  // pinned, 64 ints: [0..7] plan totals, [8],[9] non-finite flags, [16..27] refine-plan totals
  int last_BT = 0, last_B = 0, last_T = 0;  // shape of the forward whose rows b_x / b_xg hold (n_mask = 0 re-use needs the same B AND T)
  int bert_prune = 1;       // last BERT layer behind the attention on the one row per sequence the MLM head reads (n_mask == 1 steps)
  int bert_pruned_idx = -1; // row the previous forward kept (-1: all rows of b_x are valid)
  int bert_fuse_splitk_ln = 1;  // BERT fc2: the LayerNorm kernel sums the split-K slabs itself (no reduce kernel); 0 = two kernels
  int share_prefix = 1;  // encode the candidates' common causal prefix once per image
  int dedup = 1;         // candidates of one image with identical CLIP id rows are encoded once (bridge.hip prefix_plan_kernel; exact)
  float* d_staged = nullptr; int staged_cap = 0, staged_n = 0;  // czc_preprocess_u8 output slots [cap][3][S][S]
  int pack_branches = 1; // pack several candidates' rows into one attention MFMA tile
  int pool_last_layer = 1; // last CLIP-text layer: out-proj + MLP on the EOS rows only
  int fuse_ln = 1;         // bf16 / fp16 CLIP-text tower: 1 = out-proj as a full-row kernel with LN2 in its epilogue
                           // (gemm_rowln_kernel; +1 % captions/s), 2 = fc2 -> next layer's LN1 as well (measured slower: the
                           // K = 2048 GEMM pays more for 128-row tiles than the LayerNorm pass costs), 0 = off
  // 2-byte residual stream of the CLIP-TEXT tower (round 5): x lives in HBM as IEEE fp16 rows (fp32 accumulate / bias /
  // residual add, one rounding per update), the out-projection runs on the weight-stationary kernel's residual form and the
  // LayerNorms read 1 KiB rows.  1 (default): the bf16 engine; 2: the single-pass fp16 tower too (CZC_PREC_FP16 and the
  // screening pass of CZC_PREC_REFINE: outside their validated error budget, experiments only); 0: fp32 residual everywhere.
  // The split-fp16 / f32 towers and the vision tower always keep the fp32 stream.
  int resid16 = 1;
  // with the 2-byte stream: LayerNorm folded into the q/k/v and fc1 GEMMs (they read x itself; statistics from the producer
  // GEMMs' partials; no LayerNorm kernel inside the stack).  0: LayerNorm kernels on the fp16 rows
  int fold_ln = 1;
  int prof = 0;  // 0 off, 1 every kernel class, 2 only the CLIP-text linear layers (the roofline kernel family)
  std::map<std::string, ProfKind> pk;
  int64_t stat_clip_rows = 0, stat_clip_seqs = 0, stat_bert_rows = 0, stat_steps = 0;
  int64_t stat_refine_rows = 0, stat_refine_seqs = 0;
  int64_t stat_dedup_seqs = 0;  // candidate sequences that rode on an identical one's rows (of stat_clip_seqs)
  // czc_set_control_callback: the host scores the K candidate sentences of every image itself (the reference's own
  // nltk scorer where it is installed) between the two halves of a step; replaces the table look-ups of the bridge kernel
  czc_control_fn ctl_fn = nullptr; void* ctl_user = nullptr;
  int32_t* h_ctl_ids = nullptr; float* h_ctl_scores = nullptr; size_t h_ctl_cap = 0;  // pinned: [B*T + B*K] ids, [B*K] scores
};

namespace {

#define E_CHECK(expr)                                                                         \
  do {                                                                                        \
    int _r = (expr);                                                                          \
    if (_r) {                                                                                 \
      if (!e->err[0]) snprintf(e->err, sizeof(e->err), "%s", czc::g_err);                     \
      return _r;                                                                              \
    }                                                                                         \
  } while (0)

#define E_HIP(expr)                                                                           \
  do {                                                                                        \
    hipError_t _h = (expr);                                                                   \
    if (_h != hipSuccess) {                                                                   \
      snprintf(e->err, sizeof(e->err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,           \
               hipGetErrorString(_h));                                                        \
      return CZC_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

int fail(czc_engine* e, int code, const char* fmt, const char* a = "") {
  snprintf(e->err, sizeof(e->err), fmt, a);
  return code;
}

int ensure(czc_engine* e, const char* name, size_t bytes, void** out) {
  Buf& b = e->ws[name];
  if (b.bytes < bytes) {
    if (b.p) {
      E_HIP(hipStreamSynchronize(e->st));
      E_HIP(hipFree(b.p));
    }
    size_t want = bytes + bytes / 4 + 256;
    E_HIP(hipMalloc(&b.p, want));
    b.bytes = want;
  }
  *out = b.p;
  return 0;
}

struct ProfScope {
  czc_engine* e;
  ProfKind* k = nullptr;
  ProfScope(czc_engine* e_, const char* kind, double flops) : e(e_) {
    if (!e->prof) return;
    // level 2: the CLIP-text tower only -- its linear layers (the roofline family) and its attention / row kernels
    if (e->prof == 2 && strncmp(kind, "gemm_clip_text", 14) != 0 && strcmp(kind, "gemm_clip_refine") != 0 &&
        !strstr(kind, "_clip_text")) return;
    k = &e->pk[kind];
    if (k->used + 2 > k->ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) { k = nullptr; return; }
        k->ev.push_back(ev);
      }
    }
    k->flops += flops;
    k->launches += 1;
    (void)hipEventRecord(k->ev[k->used], e->st);
  }
  ~ProfScope() {
    if (!k) return;
    (void)hipEventRecord(k->ev[k->used + 1], e->st);
    k->used += 2;
  }
};

const Tensor* find(czc_engine* e, const std::string& n) {
  auto it = e->w.find(n);
  return it == e->w.end() ? nullptr : &it->second;
}

int need(czc_engine* e, const std::string& n, size_t numel, float** out) {
  const Tensor* t = find(e, n);
  if (!t) return fail(e, CZC_ERR_STATE, "missing tensor %s", n.c_str());
  if (t->numel != numel) return fail(e, CZC_ERR_STATE, "tensor %s has the wrong size", n.c_str());
  *out = t->p;
  return 0;
}

// GEMM operand in engine precision (new allocation); frees nothing
int to_act(czc_engine* e, int prec, const float* src, size_t numel, void** out) {
  void* p = nullptr;
  E_HIP(hipMalloc(&p, numel * prec_bytes(prec)));
  E_CHECK(launch_convert(prec, src, p, (long)numel, e->st));
  *out = p;
  return 0;
}

int build_layers(czc_engine* e, std::vector<LayerW>& L, int n_layers, int H, int I, bool bert_style,
                 const std::string& prefix, int prec) {
  const size_t es = prec_bytes(prec);
  L.resize(n_layers);
  for (int n = 0; n < n_layers; ++n) {
    LayerW& l = L[n];
    std::string p = prefix + std::to_string(n);
    std::string q, k, v, o, ln1, fc1, fc2, ln2;
    if (bert_style) {
      q = p + ".attention.self.query"; k = p + ".attention.self.key"; v = p + ".attention.self.value";
      o = p + ".attention.output.dense"; ln1 = p + ".attention.output.LayerNorm";
      fc1 = p + ".intermediate.dense"; fc2 = p + ".output.dense"; ln2 = p + ".output.LayerNorm";
    } else {
      q = p + ".self_attn.q_proj"; k = p + ".self_attn.k_proj"; v = p + ".self_attn.v_proj";
      o = p + ".self_attn.out_proj"; ln1 = p + ".layer_norm1";
      fc1 = p + ".mlp.fc1"; fc2 = p + ".mlp.fc2"; ln2 = p + ".layer_norm2";
    }
    float *qw, *kw, *vw, *qb, *kb, *vb, *t;
    E_CHECK(need(e, q + ".weight", (size_t)H * H, &qw));
    E_CHECK(need(e, k + ".weight", (size_t)H * H, &kw));
    E_CHECK(need(e, v + ".weight", (size_t)H * H, &vw));
    E_CHECK(need(e, q + ".bias", H, &qb));
    E_CHECK(need(e, k + ".bias", H, &kb));
    E_CHECK(need(e, v + ".bias", H, &vb));
    E_HIP(hipMalloc(&l.qkv_w, (size_t)3 * H * H * es));
    char* base = (char*)l.qkv_w;
    E_CHECK(launch_convert(prec, qw, base, (long)H * H, e->st));
    E_CHECK(launch_convert(prec, kw, base + (size_t)H * H * es, (long)H * H, e->st));
    E_CHECK(launch_convert(prec, vw, base + (size_t)2 * H * H * es, (long)H * H, e->st));
    E_HIP(hipMalloc((void**)&l.qkv_b, (size_t)3 * H * 4));
    E_HIP(hipMemcpyAsync(l.qkv_b, qb, (size_t)H * 4, hipMemcpyDeviceToDevice, e->st));
    E_HIP(hipMemcpyAsync(l.qkv_b + H, kb, (size_t)H * 4, hipMemcpyDeviceToDevice, e->st));
    E_HIP(hipMemcpyAsync(l.qkv_b + 2 * H, vb, (size_t)H * 4, hipMemcpyDeviceToDevice, e->st));
    E_CHECK(need(e, o + ".weight", (size_t)H * H, &t)); E_CHECK(to_act(e, prec, t, (size_t)H * H, &l.o_w));
    E_CHECK(need(e, o + ".bias", H, &l.o_b));
    E_CHECK(need(e, ln1 + ".weight", H, &l.ln1_g)); E_CHECK(need(e, ln1 + ".bias", H, &l.ln1_b));
    E_CHECK(need(e, fc1 + ".weight", (size_t)I * H, &t)); E_CHECK(to_act(e, prec, t, (size_t)I * H, &l.fc1_w));
    E_CHECK(need(e, fc1 + ".bias", I, &l.fc1_b));
    E_CHECK(need(e, fc2 + ".weight", (size_t)H * I, &t)); E_CHECK(to_act(e, prec, t, (size_t)H * I, &l.fc2_w));
    E_CHECK(need(e, fc2 + ".bias", H, &l.fc2_b));
    E_CHECK(need(e, ln2 + ".weight", H, &l.ln2_g)); E_CHECK(need(e, ln2 + ".bias", H, &l.ln2_b));
  }
  return 0;
}

int gemm(czc_engine* e, int prec, const char* kind, const void* A, int lda, const void* W, int ldw, const float* bias,
         const float* resid, int ldr, void* out_act, float* out_f32, int ldc, int M, int N, int K, int act) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.resid = resid; g.ldr = ldr;
  g.out_act = out_act; g.out_f32 = out_f32; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act;
  ProfScope ps(e, kind, 2.0 * M * (double)N * K);
  E_CHECK(launch_gemm(prec, g, e->st));
  return 0;
}

// x <- fp16(x + A.W^T + b) on a 2-byte residual stream (GemmArgs::x16): x16 [M,N] fp16 rows, updated in place
// part (optional): LayerNorm partials of the rows written, [N / 32][part_ld] float2
int gemm_x16(czc_engine* e, int prec, const char* kind, const void* A, int lda, const void* W, const float* bias, void* x16, int M,
             int N, int K, float* part = nullptr, long part_ld = 0) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.resid = (const float*)x16; g.ldr = N;
  g.out_act = nullptr; g.out_f32 = (float*)x16; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = ACT_NONE; g.x16 = 1;
  g.row_part = part; g.part_ld = part_ld;
  ProfScope ps(e, kind, 2.0 * M * (double)N * K);
  E_CHECK(launch_gemm(prec, g, e->st));
  return 0;
}

int gemm_ex(czc_engine* e, int prec, const char* kind, const GemmArgs& g) {
  ProfScope ps(e, kind, 2.0 * g.M * (double)g.N * g.K);
  E_CHECK(launch_gemm(prec, g, e->st));
  return 0;
}

// ---- pre-LN transformer stack shared by the CLIP text and vision towers -----------------------
// x_f32 [M,H] residual stream (updated in place); packed sequences described by off/len or fixed_T
// `pool_idx` (optional): only these n_pool rows are needed after the stack (EOS rows of the CLIP text
// tower).  The last layer then runs its out-projection and MLP on those rows only (K/V of every row
// are still produced); the pooled residual rows are returned in *pooled (fp32 [n_pool, H]).
// P = precision of the layer weights in L (the engine's CLIP precision, or PREC_F16X3 for the refine pass).
// r16: x (and *pooled) are fp16 rows of a 2-byte residual stream (czc_engine::resid16; H = 512, half-precision P).
int clip_stack(czc_engine* e, int P, const char* gk, std::vector<LayerW>& L, float* x, int M, int H, int I, int heads,
               float eps, const SegTable& tab, int max_keys, int causal, int plan_B = 0, int plan_K = 0,
               int plan_max_own = 0, const int* pool_idx = nullptr, int n_pool = 0, float** pooled = nullptr, bool r16 = false,
               const float* ln_stat0 = nullptr) {  // ln_stat0: (mean, rstd) of the incoming rows (embedding kernel): enables the LayerNorm fold
  const size_t esz = prec_bytes(P);
  void *y, *qkv, *ctx, *hbuf;
  E_CHECK(ensure(e, "cs_y", (size_t)M * H * esz, &y));
  E_CHECK(ensure(e, "cs_qkv", (size_t)M * 3 * H * esz, &qkv));
  E_CHECK(ensure(e, "cs_ctx", (size_t)M * H * esz, &ctx));
  E_CHECK(ensure(e, "cs_h", (size_t)M * I * esz, &hbuf));
  const float scale = 1.0f / sqrtf(64.0f);
  // the text tower's attention / row kernels are timed under their own classes (bench.py: MFMA utilisation of the whole
  // CLIP-text K-candidate batch, not only of its linear layers)
  const bool text = !strcmp(gk, "gemm_clip_text") || !strcmp(gk, "gemm_clip_refine");
  const char* ak = text ? "attention_clip_text" : "attention";
  const char* rk = text ? "rowops_clip_text" : "rowops";
  const double attn_flops = text ? 4.0 * H * e->plan_pairs : 0.0;
  // LayerNorm fused into the producer: out-proj (fuse_ln >= 1) and fc2 (fuse_ln >= 2) run on full 512-wide rows and
  // leave y = LN(x) beside the new fp32 x; the LayerNorm kernel then only runs where no such producer exists.
  const bool rowln = !r16 && prec_is_half(P) && e->fuse_ln && H == 512 && M >= g_rowln_min_m && I % 32 == 0;
  // LayerNorm folded into the q/k/v and fc1 GEMMs: they multiply x itself and correct with the row statistics `stat`, which
  // come from the producer's partials `part` (out-projection -> LN2, fc2 -> the next layer's LN1, the embedding kernel -> layer 0)
  const bool fold = r16 && e->fold_ln && !L.empty() && L[0].qkv_wf && ln_stat0;
  float *part = nullptr, *stat = nullptr;
  if (fold) {
    E_CHECK(ensure(e, "cs_part", (size_t)(H / 32) * M * 8, (void**)&part));
    E_CHECK(ensure(e, "cs_stat", (size_t)M * 8 + 256, (void**)&stat));
  }
  // the statistics of `rows` rows for a folded consumer with N columns: stat <- (mean, rstd) from part -- unless the consumer is
  // small enough to form them itself (gemm_wreg_stats_in_kernel: one or two images), which saves this launch
  auto finalize = [&](int rows, int N) -> int {
    if (gemm_wreg_stats_in_kernel(rows, N)) return 0;
    ProfScope ps(e, rk, 0);
    E_CHECK(launch_ln_finalize(part, M, H / 32, rows, eps, stat, e->st));
    return 0;
  };
  // st_: the rows' statistics; nullptr = what finalize(rows, N) left (in `stat`, or still in `part` for the consumer to sum)
  auto folded_gemm = [&](const void* Ax, const void* Wf, const float* bf, const float* st_, void* out, int rows, int N,
                         int act) -> int {
    GemmArgs g;
    g.A = Ax; g.lda = H; g.W = Wf; g.ldw = H; g.bias = bf; g.resid = nullptr; g.ldr = 0; g.out_act = out; g.out_f32 = nullptr; g.ldc = N;
    g.M = rows; g.N = N; g.K = H; g.act = act;
    g.ln_stat = st_ ? st_ : stat;
    if (!st_ && gemm_wreg_stats_in_kernel(rows, N)) { g.ln_part = part; g.ln_part_ld = M; g.ln_eps = eps; }
    ProfScope ps(e, gk, 2.0 * rows * (double)N * H);
    E_CHECK(launch_gemm(P, g, e->st));
    return 0;
  };
  auto ln = [&](const float* gm, const float* bt, int rows, const void* src, void* dst) -> int {  // dst <- LN(src rows)
    ProfScope ps(e, rk, 0);
    if (r16) E_CHECK(launch_layernorm_x16(P, src, nullptr, gm, bt, eps, rows, H, dst, e->st));
    else E_CHECK(launch_layernorm(P, (const float*)src, nullptr, gm, bt, eps, rows, H, dst, nullptr, e->st));
    return 0;
  };
  auto resid_gemm = [&](const void* A, int lda, const void* W, const float* b, int K) -> int {  // x += A.W^T + b
    if (r16) return gemm_x16(e, P, gk, A, lda, W, b, x, M, H, K);
    return gemm(e, P, gk, A, lda, W, K, b, x, H, nullptr, x, H, M, H, K, ACT_NONE);
  };
  auto resid_ln_gemm = [&](const void* A, int lda, const void* W, const float* b, int K, const float* gm,
                           const float* bt) -> int {  // x += A.W^T + b; y = LN(x; gm, bt)
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = b; g.resid = x; g.ldr = H; g.out_act = y; g.out_f32 = x;
    g.ldc = H; g.M = M; g.N = H; g.K = K; g.act = ACT_NONE; g.f16 = P == PREC_F16;
    g.ln_gamma = gm; g.ln_beta = bt; g.ln_eps = eps;
    ProfScope ps(e, gk, 2.0 * M * (double)H * K);
    E_CHECK(launch_gemm_rowln(g, e->st));
    return 0;
  };
  bool have_y = false;  // y already holds this layer's LN1 output (left by the previous layer's fc2)
  for (size_t n = 0; n < L.size(); ++n) {
    LayerW& l = L[n];
    if (fold) {
      E_CHECK(folded_gemm(x, l.qkv_wf, l.qkv_bf, n == 0 ? ln_stat0 : nullptr, qkv, M, 3 * H, ACT_NONE));
    } else {
      if (!have_y) E_CHECK(ln(l.ln1_g, l.ln1_b, M, x, y));
      E_CHECK(gemm(e, P, gk, y, H, l.qkv_w, H, l.qkv_b, nullptr, 0, qkv, nullptr, 3 * H, M, 3 * H, H, ACT_NONE));
    }
    { ProfScope ps(e, ak, attn_flops);
      int rc = -1;
      if (plan_B > 0 && prec_is_half(P) && g_use_mfma_attention && e->pack_branches)
        rc = launch_attention_shared(qkv, tab, plan_B, plan_K, plan_max_own, max_keys, heads, scale, ctx, e->st, P == PREC_F16);
      else if (plan_B > 0 && P == PREC_F16X3 && g_use_mfma_attention && e->pack_branches)
        rc = launch_attention_shared_split(qkv, tab, plan_B, plan_K, plan_max_own, max_keys, heads, scale, ctx, e->st);
      if (rc > 0) E_CHECK(rc);
      if (rc < 0) E_CHECK(launch_attention(P, qkv, tab, max_keys, heads, causal, scale, ctx, e->st)); }
    if (pool_idx && n + 1 == L.size() && e->pool_last_layer) {
      void *ctx_e, *y_e, *h_e; float* x_e;
      E_CHECK(ensure(e, "cs_ctx_e", (size_t)n_pool * H * esz, &ctx_e));
      E_CHECK(ensure(e, "cs_y_e", (size_t)n_pool * H * esz, &y_e));
      E_CHECK(ensure(e, "cs_h_e", (size_t)n_pool * I * esz, &h_e));
      E_CHECK(ensure(e, "cs_x_e", (size_t)n_pool * H * 4, (void**)&x_e));
      { ProfScope ps(e, rk, 0);
        E_CHECK(launch_gather_rows_bytes(ctx, pool_idx, n_pool, H * (int)esz, ctx_e, e->st));
        if (r16) E_CHECK(launch_gather_rows_bytes(x, pool_idx, n_pool, H * 2, x_e, e->st));
        else E_CHECK(launch_gather_rows_f32(x, pool_idx, n_pool, H, x_e, e->st)); }
      if (fold) {
        E_CHECK(gemm_x16(e, P, gk, ctx_e, H, l.o_w, l.o_b, x_e, n_pool, H, H, part, M));
        E_CHECK(finalize(n_pool, I));
        E_CHECK(folded_gemm(x_e, l.fc1_wf, l.fc1_bf, nullptr, h_e, n_pool, I, ACT_QUICK_GELU));
      } else {
        if (r16) E_CHECK(gemm_x16(e, P, gk, ctx_e, H, l.o_w, l.o_b, x_e, n_pool, H, H));
        else E_CHECK(gemm(e, P, gk, ctx_e, H, l.o_w, H, l.o_b, x_e, H, nullptr, x_e, H, n_pool, H, H, ACT_NONE));
        E_CHECK(ln(l.ln2_g, l.ln2_b, n_pool, x_e, y_e));
        E_CHECK(gemm(e, P, gk, y_e, H, l.fc1_w, H, l.fc1_b, nullptr, 0, h_e, nullptr, I, n_pool, I, H, ACT_QUICK_GELU));
      }
      if (r16) E_CHECK(gemm_x16(e, P, gk, h_e, I, l.fc2_w, l.fc2_b, x_e, n_pool, H, I));
      else E_CHECK(gemm(e, P, gk, h_e, I, l.fc2_w, I, l.fc2_b, x_e, H, nullptr, x_e, H, n_pool, H, I, ACT_NONE));
      *pooled = x_e;
      return 0;
    }
    if (fold) {
      E_CHECK(gemm_x16(e, P, gk, ctx, H, l.o_w, l.o_b, x, M, H, H, part, M));
      E_CHECK(finalize(M, I));
      E_CHECK(folded_gemm(x, l.fc1_wf, l.fc1_bf, nullptr, hbuf, M, I, ACT_QUICK_GELU));
      const bool more = n + 1 < L.size();  // the last layer's rows only feed the final LayerNorm (on the pooled rows)
      E_CHECK(gemm_x16(e, P, gk, hbuf, I, l.fc2_w, l.fc2_b, x, M, H, I, more ? part : nullptr, M));
      if (more) E_CHECK(finalize(M, 3 * H));
      continue;
    }
    if (rowln) E_CHECK(resid_ln_gemm(ctx, H, l.o_w, l.o_b, H, l.ln2_g, l.ln2_b));
    else {
      E_CHECK(resid_gemm(ctx, H, l.o_w, l.o_b, H));
      E_CHECK(ln(l.ln2_g, l.ln2_b, M, x, y));
    }
    E_CHECK(gemm(e, P, gk, y, H, l.fc1_w, H, l.fc1_b, nullptr, 0, hbuf, nullptr, I, M, I, H, ACT_QUICK_GELU));
    have_y = rowln && e->fuse_ln >= 2 && n + 1 < L.size();
    if (have_y) E_CHECK(resid_ln_gemm(hbuf, I, l.fc2_w, l.fc2_b, I, L[n + 1].ln1_g, L[n + 1].ln1_b));
    else E_CHECK(resid_gemm(hbuf, I, l.fc2_w, l.fc2_b, I));
  }
  return 0;
}

// ---- BERT encoder (post-LN) on [B*T] rows: leaves the final hidden state in ws "b_x" ----------
// keep_idx >= 0: only row keep_idx of every sequence is read afterwards (the masked slot, gen_utils.py:69): the last layer
// still forms q/k/v and the attention for all rows, then gathers that row and runs out-proj / LN / MLP / LN on B rows
// instead of B*T; the result goes to ws "b_xg" [B,H].  Same kernels per row, so the row's values do not change.
int bert_forward(czc_engine* e, const int* d_inp, int B, int T, int keep_idx = -1) {
  const czc_config& c = e->cfg;
  const int P = e->pb, M = B * T, H = c.bert_hidden, I = c.bert_inter;
  void *xa, *qkv, *ctx, *hbuf; float *x, *tmp;
  E_CHECK(ensure(e, "b_x", (size_t)M * H * 4, (void**)&x));
  E_CHECK(ensure(e, "b_tmp", (size_t)M * H * 4, (void**)&tmp));
  E_CHECK(ensure(e, "b_xa", (size_t)M * H * e->eb, &xa));
  E_CHECK(ensure(e, "b_qkv", (size_t)M * 3 * H * e->eb, &qkv));
  E_CHECK(ensure(e, "b_ctx", (size_t)M * H * e->eb, &ctx));
  E_CHECK(ensure(e, "b_h", (size_t)M * I * e->eb, &hbuf));
  float *word, *pos, *typ, *g, *b;
  E_CHECK(need(e, "bert.embeddings.word_embeddings.weight", (size_t)c.bert_vocab * H, &word));
  E_CHECK(need(e, "bert.embeddings.position_embeddings.weight", (size_t)c.bert_max_pos * H, &pos));
  E_CHECK(need(e, "bert.embeddings.token_type_embeddings.weight", (size_t)2 * H, &typ));
  E_CHECK(need(e, "bert.embeddings.LayerNorm.weight", H, &g));
  E_CHECK(need(e, "bert.embeddings.LayerNorm.bias", H, &b));
  { ProfScope ps(e, "rowops", 0);
    E_CHECK(launch_bert_embed(P, d_inp, B, T, H, word, pos, typ, g, b, c.bert_eps, xa, x, e->st)); }
  const float scale = 1.0f / sqrtf(64.0f);
  e->bert_pruned_idx = -1;
  // fc2 + residual + LayerNorm (HF:bert/modeling_bert.py:488-496): xr <- LN(xr + h.W2^T + b), xa <- the same in the operand type.
  // Where the launcher splits K (fc2's K = 3072 at every row count above the skinny kernel's 32) the slice sums stay in its slab
  // workspace and the LayerNorm kernel sums them itself (GemmArgs::splitk_pending): one launch less per layer, no fp32 round trip
  auto fc2_ln = [&](const void* h, LayerW& l, float* xr, int rows) -> int {
    SplitkPending pend;
    GemmArgs g;
    g.A = h; g.lda = I; g.W = l.fc2_w; g.ldw = I; g.bias = l.fc2_b; g.resid = xr; g.ldr = H; g.out_act = nullptr; g.out_f32 = tmp; g.ldc = H;
    g.M = rows; g.N = H; g.K = I; g.act = ACT_NONE;
    g.splitk_pending = e->bert_fuse_splitk_ln ? &pend : nullptr;
    E_CHECK(gemm_ex(e, P, "gemm_bert", g));
    ProfScope ps(e, "rowops", 0);
    if (pend.nslab > 0) E_CHECK(launch_layernorm_splitk(P, pend, l.fc2_b, xr, H, l.ln2_g, l.ln2_b, c.bert_eps, rows, H, xa, xr, e->st));
    else E_CHECK(launch_layernorm(P, tmp, nullptr, l.ln2_g, l.ln2_b, c.bert_eps, rows, H, xa, xr, e->st));
    return 0;
  };
  for (int n = 0; n < c.bert_layers; ++n) {
    LayerW& l = e->bert[n];
    E_CHECK(gemm(e, P, "gemm_bert", xa, H, l.qkv_w, H, l.qkv_b, nullptr, 0, qkv, nullptr, 3 * H, M, 3 * H, H, ACT_NONE));
    { ProfScope ps(e, "attention", 0);
      SegTable tab{nullptr, nullptr, nullptr, nullptr, B, T};
      E_CHECK(launch_attention(P, qkv, tab, T, c.bert_heads, 0, scale, ctx, e->st)); }
    if (keep_idx >= 0 && n + 1 == c.bert_layers) {
      int* idx; void* ctx_g; float* x_g;
      E_CHECK(ensure(e, "b_pidx", (size_t)B * 4, (void**)&idx));
      E_CHECK(ensure(e, "b_cg", (size_t)B * H * e->eb, &ctx_g));
      E_CHECK(ensure(e, "b_xg", (size_t)B * H * 4, (void**)&x_g));
      { ProfScope ps(e, "rowops", 0);
        E_CHECK(launch_make_row_index(idx, B, T, keep_idx, e->st));
        E_CHECK(launch_gather_rows_bytes(ctx, idx, B, H * (int)e->eb, ctx_g, e->st));
        E_CHECK(launch_gather_rows_f32(x, idx, B, H, x_g, e->st)); }
      // tmp / xa / hbuf: their first B rows are free here (xa fed this layer's q/k/v, hbuf the previous layer's fc2)
      E_CHECK(gemm(e, P, "gemm_bert", ctx_g, H, l.o_w, H, l.o_b, x_g, H, nullptr, tmp, H, B, H, H, ACT_NONE));
      { ProfScope ps(e, "rowops", 0); E_CHECK(launch_layernorm(P, tmp, nullptr, l.ln1_g, l.ln1_b, c.bert_eps, B, H, xa, x_g, e->st)); }
      E_CHECK(gemm(e, P, "gemm_bert", xa, H, l.fc1_w, H, l.fc1_b, nullptr, 0, hbuf, nullptr, I, B, I, H, ACT_GELU_ERF));
      E_CHECK(fc2_ln(hbuf, l, x_g, B));
      e->bert_pruned_idx = keep_idx;
      break;
    }
    E_CHECK(gemm(e, P, "gemm_bert", ctx, H, l.o_w, H, l.o_b, x, H, nullptr, tmp, H, M, H, H, ACT_NONE));
    { ProfScope ps(e, "rowops", 0); E_CHECK(launch_layernorm(P, tmp, nullptr, l.ln1_g, l.ln1_b, c.bert_eps, M, H, xa, x, e->st)); }
    E_CHECK(gemm(e, P, "gemm_bert", xa, H, l.fc1_w, H, l.fc1_b, nullptr, 0, hbuf, nullptr, I, M, I, H, ACT_GELU_ERF));
    E_CHECK(fc2_ln(hbuf, l, x, M));
  }
  return 0;
}

// MLM head on row gen_idx of every sequence -> logits fp32 [B,V] in ws "b_logits"
int mlm_head(czc_engine* e, int B, int T, int gen_idx, float** logits_out) {
  const czc_config& c = e->cfg;
  const int P = e->pb, H = c.bert_hidden, V = c.bert_vocab;
  float* x = (float*)e->ws["b_x"].p;
  int* idx; float *gx, *t32, *logits; void *ga, *ta;
  E_CHECK(ensure(e, "h_idx", (size_t)B * 4, (void**)&idx));
  E_CHECK(ensure(e, "h_gx", (size_t)B * H * 4, (void**)&gx));
  E_CHECK(ensure(e, "h_ga", (size_t)B * H * e->eb, &ga));
  E_CHECK(ensure(e, "h_t32", (size_t)B * H * 4, (void**)&t32));
  E_CHECK(ensure(e, "h_ta", (size_t)B * H * e->eb, &ta));
  E_CHECK(ensure(e, "b_logits", (size_t)B * V * 4, (void**)&logits));
  float *db, *g, *b, *bias;
  E_CHECK(need(e, "cls.predictions.transform.dense.bias", H, &db));
  E_CHECK(need(e, "cls.predictions.transform.LayerNorm.weight", H, &g));
  E_CHECK(need(e, "cls.predictions.transform.LayerNorm.bias", H, &b));
  E_CHECK(need(e, "cls.predictions.bias", V, &bias));
  if (e->bert_pruned_idx >= 0 && e->bert_pruned_idx != gen_idx)
    return fail(e, CZC_ERR_STATE, "n_mask=0 re-use of a forward that kept one row only (it follows an n_mask >= 2 step, gen_utils.py:164-166)%s");
  { ProfScope ps(e, "rowops", 0);
    if (e->bert_pruned_idx >= 0) {
      gx = (float*)e->ws["b_xg"].p;  // the forward left exactly these rows
    } else {
      E_CHECK(launch_make_row_index(idx, B, T, gen_idx, e->st));
      E_CHECK(launch_gather_rows_f32(x, idx, B, H, gx, e->st));
    }
    E_CHECK(launch_convert(P, gx, ga, (long)B * H, e->st)); }
  E_CHECK(gemm(e, P, "gemm_bert", ga, H, e->mlm_dense_w, H, db, nullptr, 0, nullptr, t32, H, B, H, H, ACT_GELU_ERF));
  { ProfScope ps(e, "rowops", 0); E_CHECK(launch_layernorm(P, t32, nullptr, g, b, c.bert_eps, B, H, ta, nullptr, e->st)); }
  E_CHECK(gemm(e, P, "gemm_bert", ta, H, e->decoder_w, H, bias, nullptr, 0, nullptr, logits, V, B, V, H, ACT_NONE));
  *logits_out = logits;
  return 0;
}

// CLIP text tower (clip/clip.py:78-83) on B x K candidate sequences: ids [B*K,77] + len -> feat fp32
// [B*K, proj].  With `share` the causal prefix common to an image's K candidates is encoded once
// (trunk segment) and every candidate only carries the rows from its first differing token on.
// Two halves around the step's single host round trip (32 bytes of totals: rows, longest sequence, overflow):
// clip_plan builds the segment table on the device and starts the read, clip_tower runs on the sizes it returned.
struct PlanBufs { int *own_len, *pre_len, *src, *pos0, *own_off, *pre_off, *eidx, *img_max, *rep; };

// the refine engine inside czc_generate: screening pass on fp16 rows (czc_engine::refine_rows16)
static inline bool refine_rows16_now(const czc_engine* e) {
  return e->refine && e->in_generate && e->refine_rows16 && e->cfg.clip_hidden == 512;
}

// pfx "p": the plan of the screening / only pass; "r": the plan of the refine pass
int plan_bufs(czc_engine* e, int B, int K, PlanBufs* p, const char* pfx = "p") {
  const int n_seq = B * K, S = B + n_seq;
  const std::string q(pfx);
  E_CHECK(ensure(e, (q + "_own_len").c_str(), (size_t)S * 4, (void**)&p->own_len));
  E_CHECK(ensure(e, (q + "_pre_len").c_str(), (size_t)S * 4, (void**)&p->pre_len));
  E_CHECK(ensure(e, (q + "_src").c_str(), (size_t)S * 4, (void**)&p->src));
  E_CHECK(ensure(e, (q + "_pos0").c_str(), (size_t)S * 4, (void**)&p->pos0));
  E_CHECK(ensure(e, (q + "_own_off").c_str(), (size_t)(S + 1) * 4, (void**)&p->own_off));
  E_CHECK(ensure(e, (q + "_pre_off").c_str(), (size_t)S * 4, (void**)&p->pre_off));
  E_CHECK(ensure(e, (q + "_eidx").c_str(), (size_t)n_seq * 4, (void**)&p->eidx));
  E_CHECK(ensure(e, (q + "_img_max").c_str(), (size_t)B * 4, (void**)&p->img_max));
  E_CHECK(ensure(e, (q + "_rep").c_str(), (size_t)n_seq * 4, (void**)&p->rep));
  return 0;
}

int clip_plan(czc_engine* e, const int* cids, const int* clen, int B, int K, int share, int* totals) {
  PlanBufs p;
  E_CHECK(plan_bufs(e, B, K, &p));
  const int S = B + B * K;
  { ProfScope ps(e, "bridge", 0);
    // de-duplication rides on the shared-prefix plan (K > 1 candidates per image; czc_encode_text's independent sequences have none)
    int* rep = e->dedup && share && K > 1 ? p.rep : nullptr;
    E_CHECK(launch_prefix_plan(cids, clen, B, K, share, p.own_len, p.pre_len, p.src, p.pos0, totals + 3, p.img_max, e->st, rep, totals + 5));
    E_CHECK(launch_scan(p.own_len, S, p.own_off, totals, e->st));
    E_CHECK(launch_prefix_finish(p.own_off, p.own_len, B, K, p.pre_off, p.eidx, totals + 6, e->st, rep)); }
  E_HIP(hipMemcpyAsync(e->h_totals, totals, 32, hipMemcpyDeviceToHost, e->st));
  int* flag;
  E_CHECK(ensure(e, "s_nonfinite", 32, (void**)&flag));
  E_HIP(hipMemcpyAsync(e->h_totals + 8, flag, 4, hipMemcpyDeviceToHost, e->st));  // the previous steps' cosine check
  return 0;
}

// The text tower on a planned set of packed segments: n_seg segments (tab), M rows, n_pool sequences pooled at p.eidx.
// P / L / tproj: the tower's precision and weights; plan_B > 0: regular B x K plan (packed-branch attention kernels).
int clip_tower_on(czc_engine* e, int P, std::vector<LayerW>& L, const void* tproj, const char* gk, const int* cids,
                  const PlanBufs& p, int n_seg, int n_pool, int M, int max_len, int plan_B, int plan_K, int max_branch,
                  const char* feat_name, float** feat_out) {
  const czc_config& c = e->cfg;
  const int H = c.clip_hidden;
  float *x, *feat, *tok, *pos, *fg, *fb; void* pa;
  E_CHECK(ensure(e, "c_x", (size_t)M * H * 4, (void**)&x));
  E_CHECK(ensure(e, "c_pa", (size_t)n_pool * H * prec_bytes(P), &pa));
  E_CHECK(ensure(e, feat_name, (size_t)n_pool * c.clip_proj * 4, (void**)&feat));
  E_CHECK(need(e, "text_model.embeddings.token_embedding.weight", (size_t)c.clip_vocab * H, &tok));
  E_CHECK(need(e, "text_model.embeddings.position_embedding.weight", (size_t)c.clip_max_pos * H, &pos));
  E_CHECK(need(e, "text_model.final_layer_norm.weight", H, &fg));
  E_CHECK(need(e, "text_model.final_layer_norm.bias", H, &fb));
  const bool r16 = H == 512 && ((e->resid16 >= 1 && P == PREC_BF16) || (e->resid16 >= 2 && P == PREC_F16) ||
                                (P == PREC_F16 && refine_rows16_now(e)));
  float* stat0 = nullptr;
  if (r16 && e->fold_ln && !L.empty() && L[0].qkv_wf) E_CHECK(ensure(e, "c_stat0", (size_t)M * 8 + 256, (void**)&stat0));
  { ProfScope ps(e, "rowops_clip_text", 0);
    E_CHECK(launch_clip_embed(cids, CZC_CLIP_MAX_LEN, p.src, p.pos0, p.own_off, p.own_len, n_seg, max_len, H, tok, pos, x, e->st, r16 ? 1 : 0,
                              stat0, c.clip_eps)); }
  SegTable tab{p.pre_off, p.pre_len, p.own_off, p.own_len, n_seg, 0, plan_B > 0 ? p.img_max : nullptr};
  float* pooled = nullptr;
  E_CHECK(clip_stack(e, P, gk, L, x, M, H, c.clip_inter, c.clip_heads, c.clip_eps, tab, max_len, 1, plan_B,
                     plan_K, max_branch, p.eidx, n_pool, &pooled, r16, stat0));
  { ProfScope ps(e, "rowops_clip_text", 0);
    if (r16) {
      if (pooled) E_CHECK(launch_layernorm_x16(P, pooled, nullptr, fg, fb, c.clip_eps, n_pool, H, pa, e->st));
      else E_CHECK(launch_layernorm_x16(P, x, p.eidx, fg, fb, c.clip_eps, n_pool, H, pa, e->st));
    } else if (pooled) E_CHECK(launch_layernorm(P, pooled, nullptr, fg, fb, c.clip_eps, n_pool, H, pa, nullptr, e->st));
    else E_CHECK(launch_layernorm(P, x, p.eidx, fg, fb, c.clip_eps, n_pool, H, pa, nullptr, e->st)); }
  E_CHECK(gemm(e, P, gk, pa, H, tproj, H, nullptr, nullptr, 0, nullptr, feat, c.clip_proj, n_pool, c.clip_proj, H, ACT_NONE));
  *feat_out = feat;
  return 0;
}

// the engine's text tower on the regular B x K plan built by clip_plan; `exact`: the refine engine's split-fp16 weights
int clip_tower(czc_engine* e, const int* cids, int B, int K, int share, int M, int max_len, int max_branch, int n_trunk,
               float** feat_out, bool exact = false) {
  (void)share; (void)n_trunk;
  PlanBufs p;
  E_CHECK(plan_bufs(e, B, K, &p));
  const bool x = exact && e->refine;
  return clip_tower_on(e, x ? (int)PREC_F16X3 : e->pc, x ? e->ctext_x : e->ctext, x ? e->tproj_wx : e->tproj_w, "gemm_clip_text",
                       cids, p, B + B * K, B * K, M, max_len, B, K, max_branch, "c_feat", feat_out);
}

// sizes of the tower from the totals the plan read back (after the stream has been synchronised)
int read_totals(czc_engine* e, int* M, int* max_len, int* max_branch, int* n_trunk) {
  const czc_config& c = e->cfg;
  *M = e->h_totals[0]; *max_len = e->h_totals[3]; *max_branch = e->h_totals[4]; *n_trunk = e->h_totals[6];
  e->plan_pairs = (double)e->h_totals[7];
  if (e->h_totals[8]) return fail(e, CZC_ERR_OVERFLOW, "non-finite CLIP cosine: an fp16 quantity overflowed in a tower (fp16 residual rows: set option resid16 = 0 on the bf16 engine, refine_rows16 = 0 on the refine engine; fp16 operands: use CZC_PREC_SPLIT)%s");
  if (e->h_totals[2]) return fail(e, CZC_ERR_OVERFLOW, "text bridge overflow (row text > CZC_BRIDGE_MAX_BYTES)%s");
  if (*max_len > c.clip_max_pos || *max_len > CZC_CLIP_MAX_LEN)
    return fail(e, CZC_ERR_ARG, "CLIP sequence longer than max_position_embeddings%s");
  return 0;
}

int clip_text_forward(czc_engine* e, const int* cids, const int* clen, int B, int K, int share, int* totals,
                      float** feat_out, bool exact = false) {
  E_CHECK(clip_plan(e, cids, clen, B, K, share, totals));
  E_HIP(hipStreamSynchronize(e->st));  // the one host round trip per step (the reference has one too, gen_utils.py:81)
  int M, max_len, max_branch, n_trunk;
  E_CHECK(read_totals(e, &M, &max_len, &max_branch, &n_trunk));
  E_CHECK(clip_tower(e, cids, B, K, share, M, max_len, max_branch, n_trunk, feat_out, exact));
  e->stat_clip_rows += M;
  e->stat_clip_seqs += B * K;
  return 0;
}

// ---- one position-step on the device-resident d_inp (gen_utils.py:66-81), in two halves around the size read ----
struct StepArgs {
  int* d_inp; int B, T, gen_idx, n_mask, dot_allowed, K;
  czc_hyper hp;
};
struct StepBufs { float *probs, *senti, *reps; int *idxs, *cand, *cids, *clen, *totals; };

int step_bufs(czc_engine* e, int n_seq, StepBufs* b) {
  E_CHECK(ensure(e, "s_probs", (size_t)n_seq * 4, (void**)&b->probs));
  E_CHECK(ensure(e, "s_idxs", (size_t)n_seq * 4, (void**)&b->idxs));
  E_CHECK(ensure(e, "s_cand", (size_t)n_seq * 4, (void**)&b->cand));
  E_CHECK(ensure(e, "s_cids", (size_t)n_seq * CZC_CLIP_MAX_LEN * 4, (void**)&b->cids));
  E_CHECK(ensure(e, "s_clen", (size_t)n_seq * 4, (void**)&b->clen));
  E_CHECK(ensure(e, "s_tot", 32, (void**)&b->totals));
  E_CHECK(ensure(e, "s_senti", (size_t)n_seq * 4, (void**)&b->senti));
  E_CHECK(ensure(e, "s_reps", (size_t)n_seq * 4, (void**)&b->reps));
  return 0;
}

// mask -> BERT -> MLM head -> softmax/top-K -> text bridge -> segment plan -> start of the 32-byte size read
int step_phase_a(czc_engine* e, const StepArgs& a) {
  const czc_config& c = e->cfg;
  const czc_hyper* hp = &a.hp;
  if (a.n_mask > 0) {
    E_CHECK(launch_mask_positions(a.d_inp, a.B, a.T, a.gen_idx, a.n_mask, c.mask_id, e->st));
    E_CHECK(bert_forward(e, a.d_inp, a.B, a.T, e->bert_prune && a.n_mask == 1 ? a.gen_idx : -1));
  }
  float* logits;
  E_CHECK(mlm_head(e, a.B, a.T, a.gen_idx, &logits));
  StepBufs b;
  E_CHECK(step_bufs(e, a.B * a.K, &b));
  { ProfScope ps(e, "topk", 0);
    E_CHECK(launch_softmax_mask_topk(logits, a.B, c.bert_vocab, a.K, e->d_mask, hp->temperature, c.dot_id, a.dot_allowed,
                                     b.probs, b.idxs, b.cand, e->st)); }
  E_HIP(hipMemsetAsync(b.totals, 0, 32, e->st));
  { ProfScope ps(e, "bridge", 0);
    PosDev pos{hp->control == 2 ? e->d_pos_tags : nullptr, e->d_pos_masks, e->pos_n};
    E_CHECK(launch_bridge(e->bd, a.d_inp, a.B, a.T, a.gen_idx, b.cand, a.K, hp->control == 1 ? e->d_lex : nullptr,
                          hp->control == 1 ? e->d_lex_pos : nullptr, e->d_lex_cls, hp->negative,
                          pos, b.cids, b.clen, b.senti, b.reps, b.totals + 2, e->st)); }
  E_CHECK(clip_plan(e, b.cids, b.clen, a.B, a.K, e->share_prefix, b.totals));
  return 0;
}

// Control scores from the host (czc_set_control_callback): rows as the reference decodes them at
// control_gen_utils.py:54-57 / :158-160 -- `inp` with [MASK] at gen_idx, candidate k of image b replaces it by cand[b][k].
// The scores are only needed by the combine kernel, so the step fetches the ids with its size read (control_fetch, queued
// in front of the step's one host round trip), launches the CLIP tower, and calls the host WHILE the tower runs
// (control_score): the reference's Python scorer (control_gen_utils.py:56-57) disappears under the tower's GPU time wherever
// it is the shorter of the two.
int control_pinned(czc_engine* e, size_t n_inp, size_t n_seq) {
  const size_t want = (n_inp + n_seq) * 4 + n_seq * 4;
  if (e->h_ctl_cap < want) {
    E_HIP(hipStreamSynchronize(e->st));
    if (e->h_ctl_ids) (void)hipHostFree(e->h_ctl_ids);
    e->h_ctl_ids = nullptr; e->h_ctl_cap = 0;
    E_HIP(hipHostMalloc((void**)&e->h_ctl_ids, want + want / 4));
    e->h_ctl_cap = want + want / 4;
  }
  e->h_ctl_scores = (float*)(e->h_ctl_ids + n_inp + n_seq);
  return 0;
}

int control_fetch(czc_engine* e, const StepArgs& a) {
  StepBufs b;
  const size_t n_seq = (size_t)a.B * a.K, n_inp = (size_t)a.B * a.T;
  E_CHECK(step_bufs(e, (int)n_seq, &b));
  E_CHECK(control_pinned(e, n_inp, n_seq));
  E_HIP(hipMemcpyAsync(e->h_ctl_ids, a.d_inp, n_inp * 4, hipMemcpyDeviceToHost, e->st));
  E_HIP(hipMemcpyAsync(e->h_ctl_ids + n_inp, b.cand, n_seq * 4, hipMemcpyDeviceToHost, e->st));
  return 0;
}

// after the step's host round trip (the ids have landed) and after the tower's launches have been queued
int control_score(czc_engine* e, const StepArgs& a) {
  StepBufs b;
  const size_t n_seq = (size_t)a.B * a.K, n_inp = (size_t)a.B * a.T;
  E_CHECK(step_bufs(e, (int)n_seq, &b));
  memset(e->h_ctl_scores, 0, n_seq * 4);
  const int rc = e->ctl_fn(e->ctl_user, e->h_ctl_ids, e->h_ctl_ids + n_inp, a.B, a.T, a.K, a.gen_idx, e->h_ctl_scores);
  if (rc) return fail(e, CZC_ERR_STATE, "the control callback reported an error%s");
  // pinned source: the copy is queued behind the tower; the buffer is next written after the next step's round trip
  E_HIP(hipMemcpyAsync(b.senti, e->h_ctl_scores, n_seq * 4, hipMemcpyHostToDevice, e->st));
  return 0;
}

// CLIP text tower on the planned rows -> cosine / softmax_K / fusion / argmax / write-back
int step_phase_b(czc_engine* e, const StepArgs& a, int M, int max_len, int max_branch, int n_trunk) {
  const czc_config& c = e->cfg;
  const czc_hyper* hp = &a.hp;
  const int n_seq = a.B * a.K;
  StepBufs b;
  E_CHECK(step_bufs(e, n_seq, &b));
  float* feat;
  E_CHECK(clip_tower(e, b.cids, a.B, a.K, e->share_prefix, M, max_len, max_branch, n_trunk, &feat));
  if (hp->control && e->ctl_fn) E_CHECK(control_score(e, a));  // host scorer under the tower that was just queued
  float *cscore, *cref, *fin, *bcos; int* best;
  E_CHECK(ensure(e, "s_cscore", (size_t)n_seq * 4, (void**)&cscore));
  E_CHECK(ensure(e, "s_cref", (size_t)n_seq * 4, (void**)&cref));
  E_CHECK(ensure(e, "s_fin", (size_t)n_seq * 4, (void**)&fin));
  E_CHECK(ensure(e, "s_best", (size_t)a.B * 4, (void**)&best));
  E_CHECK(ensure(e, "s_bcos", (size_t)a.B * 4, (void**)&bcos));
  CombineArgs ca;
  ca.text_feat = feat; ca.img_n = e->d_img_n; ca.logit_scale_exp = e->logit_scale_exp; ca.probs = b.probs; ca.cand = b.cand;
  ca.senti_raw = b.senti; ca.repeats = b.reps; ca.alpha = hp->alpha; ca.beta = hp->beta; ca.gamma = hp->gamma;
  ca.use_senti = hp->control; ca.B = a.B; ca.K = a.K; ca.D = c.clip_proj; ca.clip_score = cscore; ca.clip_ref = cref;
  ca.final_score = fin; ca.best = best; ca.best_cos = bcos; ca.inp = a.d_inp; ca.T = a.T; ca.gen_idx = a.gen_idx;
  E_CHECK(ensure(e, "s_nonfinite", 32, (void**)&ca.nonfinite));
  if (!e->refine) {
    ProfScope ps(e, "combine", 0);
    E_CHECK(launch_combine(ca, e->st));
    return 0;
  }
  // ---- screen-then-refine: scores from the screening cosines (no write-back), selection, second pass, final scores ----
  int *kind, *list, *count, *count_off, *rtot, *rlist;
  float* rcos;
  E_CHECK(ensure(e, "r_kind", (size_t)n_seq * 4, (void**)&kind));
  E_CHECK(ensure(e, "r_list", (size_t)n_seq * 4, (void**)&list));
  E_CHECK(ensure(e, "r_count", (size_t)a.B * 4, (void**)&count));
  E_CHECK(ensure(e, "r_count_off", (size_t)(a.B + 1) * 4, (void**)&count_off));
  E_CHECK(ensure(e, "r_tot", 64, (void**)&rtot));
  E_CHECK(ensure(e, "r_rlist", (size_t)n_seq * 4, (void**)&rlist));
  E_CHECK(ensure(e, "r_cos", (size_t)n_seq * 4, (void**)&rcos));
  PlanBufs sp, rp;
  E_CHECK(plan_bufs(e, a.B, a.K, &sp));
  E_CHECK(plan_bufs(e, a.B, a.K, &rp, "r"));
  const int S = a.B + n_seq;
  // (on fp16 rows the deviation a kept screening cosine carries is refine_rows16_factor times larger: the mass threshold
  // shrinks by the same factor, so theta * deviation -- what such a candidate can move its score by -- stays what it was)
  const float theta = (e->in_generate ? e->refine_theta_gen / (refine_rows16_now(e) ? e->refine_rows16_factor : 1.f) : e->refine_theta_x) /
                      fmaxf(hp->beta * e->logit_scale_exp, 1e-6f);
  { ProfScope ps(e, "combine", 0);
    ca.inp = nullptr;
    E_CHECK(launch_combine(ca, e->st));
    const float gate_h = e->gate_now ? e->refine_gate_delta * (refine_rows16_now(e) ? e->refine_rows16_factor : 1.f) * e->logit_scale_exp : 0.f;
    E_CHECK(launch_refine_select(cscore, fin, a.B, a.K, theta, e->in_generate ? e->refine_samples : e->refine_samples_step, gate_h, hp->beta, e->gate_need_cos ? 1 : 0,
                                 ca.nonfinite + 4, kind, list, count, e->st)); }
  { ProfScope ps(e, "bridge", 0);
    E_HIP(hipMemsetAsync(rtot, 0, 64, e->st));
    E_HIP(hipMemsetAsync(rp.own_len, 0, (size_t)S * 4, e->st));
    E_CHECK(launch_scan(count, a.B, count_off, rtot + 8, e->st));  // rtot[8] = R, rtot[9] = Kr = largest per-image count
    E_CHECK(launch_refine_plan(b.clen, sp.own_len, list, count, count_off, rtot + 9, a.B, a.K, rp.own_len, rp.pre_len, rp.src, rp.pos0,
                               rlist, rtot + 3, rp.img_max, e->st));
    E_CHECK(launch_scan(rp.own_len, S, rp.own_off, rtot, e->st));   // rtot[0] = rows of the refine pass
    E_CHECK(launch_refine_finish(rp.own_off, rp.own_len, count, count_off, rtot + 9, a.B, a.K, rp.pre_off, rp.eidx, e->st)); }
  E_HIP(hipMemcpyAsync(e->h_totals + 16, rtot, 48, hipMemcpyDeviceToHost, e->st));
  E_HIP(hipStreamSynchronize(e->st));  // second (and last) host round trip of the step: the sizes of the refine pass
  const int M2 = e->h_totals[16], max_len2 = e->h_totals[19], max_branch2 = e->h_totals[20], R = e->h_totals[24], Kr = e->h_totals[25];
  if (R < 0 || R > n_seq || M2 < 0 || Kr < 0 || Kr > a.K) return fail(e, CZC_ERR_STATE, "refine plan returned impossible sizes%s");
  if (R > 0) {
    float* feat2;
    e->plan_pairs = (double)e->h_totals[16 + 7];
    // regular B x Kr plan (empty slots have no rows): the packed-branch split attention serves it like the screening plan
    E_CHECK(clip_tower_on(e, PREC_F16X3, e->ctext_x, e->tproj_wx, "gemm_clip_refine", b.cids, rp, a.B + a.B * Kr, R, M2, max_len2,
                          a.B, Kr, max_branch2, "c_feat2", &feat2));
    ProfScope ps(e, "combine", 0);
    E_CHECK(launch_refine_cosine(feat2, e->d_img_n, rlist, count_off + a.B, R, a.K, c.clip_proj, rcos, ca.nonfinite, e->st));
  }
  { ProfScope ps(e, "combine", 0);
    ca.text_feat = nullptr; ca.inp = a.d_inp; ca.refine_kind = kind; ca.refine_cos = rcos;
    ca.refine_guard = e->refine_guard_dev * (refine_rows16_now(e) ? e->refine_rows16_factor : 1.f);
    E_CHECK(launch_combine(ca, e->st)); }
  e->stat_refine_rows += M2;
  e->stat_refine_seqs += R;
  return 0;
}

int step_device(czc_engine* e, int* d_inp, int B, int T, int gen_idx, int n_mask, int dot_allowed, int K,
                const czc_hyper* hp) {
  const czc_config& c = e->cfg;
  if (!e->finalized) return fail(e, CZC_ERR_STATE, "weights not finalized%s");
  if (c.bert_layers <= 0) return fail(e, CZC_ERR_STATE, "this engine was created without the BERT tower%s");
  if (!e->d_mask) return fail(e, CZC_ERR_STATE, "token mask not set%s");
  if (!e->has_bridge) return fail(e, CZC_ERR_STATE, "bridge tables not set%s");
  if (!e->d_img_n || e->img_B != B) return fail(e, CZC_ERR_STATE, "image embeds not set for this batch size%s");
  if (T > CZC_MAX_BERT_LEN || gen_idx < 0 || gen_idx >= T || K > CZC_MAX_TOPK)
    return fail(e, CZC_ERR_ARG, "step: bad T/gen_idx/K%s");
  if (hp->control == 1 && !e->ctl_fn && !e->d_lex && !e->d_lex_pos)
    return fail(e, CZC_ERR_STATE, "sentiment path needs a lexicon (czc_set_lexicon / czc_set_lexicon_pos) or czc_set_control_callback%s");
  if (hp->control == 2 && !e->ctl_fn && !e->d_pos_tags)
    return fail(e, CZC_ERR_STATE, "POS path needs czc_set_pos or czc_set_control_callback%s");
  if (n_mask <= 0 && (e->last_B != B || e->last_T != T))
    return fail(e, CZC_ERR_STATE, "n_mask=0 needs a previous forward of the same [B,T] shape%s");
  StepArgs a{d_inp, B, T, gen_idx, n_mask, dot_allowed, K, *hp};
  E_CHECK(step_phase_a(e, a));
  if (n_mask > 0) { e->stat_bert_rows += B * T; e->last_BT = B * T; e->last_B = B; e->last_T = T; }
  if (hp->control && e->ctl_fn) E_CHECK(control_fetch(e, a));  // ids for the host scorer ride on the same round trip
  E_HIP(hipStreamSynchronize(e->st));  // the one host round trip per step (the reference has one too, gen_utils.py:81)
  int M, max_len, max_branch, n_trunk;
  E_CHECK(read_totals(e, &M, &max_len, &max_branch, &n_trunk));
  E_CHECK(step_phase_b(e, a, M, max_len, max_branch, n_trunk));
  e->stat_clip_rows += M;
  e->stat_clip_seqs += B * K;
  e->stat_dedup_seqs += e->h_totals[5];
  e->stat_steps += 1;
  return 0;
}

int copy_out(czc_engine* e, void* dst, const char* ws_name, size_t bytes) {
  if (!dst) return 0;
  E_HIP(hipMemcpyAsync(dst, e->ws[ws_name].p, bytes, hipMemcpyDefault, e->st));
  return 0;
}

// Does any option setting of this engine run the text tower on fp16 rows with the LayerNorms folded into its GEMMs?  The folded
// operands (44 MB for 12 layers) are built by czc_finalize_weights only then: the bf16 engine (resid16 >= 1, its default), the
// refine engine whose czc_generate screens on fp16 rows (refine_rows16, default), a single-pass fp16 engine asked for resid16 = 2.
bool wants_folded_ln(const czc_engine* e) {
  if (!prec_is_half(e->pc) || e->cfg.clip_hidden != 512 || !e->fold_ln) return false;
  if (e->pc == PREC_BF16) return e->resid16 >= 1;
  return e->resid16 >= 2 || (e->refine && e->refine_rows16);
}

void free_layer_set(std::vector<LayerW>& L) {
  for (auto& l : L) {
    (void)hipFree(l.qkv_w); (void)hipFree(l.qkv_b); (void)hipFree(l.o_w); (void)hipFree(l.fc1_w); (void)hipFree(l.fc2_w);
    (void)hipFree(l.qkv_wf); (void)hipFree(l.qkv_bf); (void)hipFree(l.fc1_wf); (void)hipFree(l.fc1_bf);
  }
  L.clear();
}

}  // namespace

// =================================================================================================
extern "C" {

int czc_version(void) { return 100; }

const char* czc_last_error(const czc_engine* e) { return e ? e->err : czc::g_err; }

int czc_create(const czc_config* cfg, int device_id, czc_engine** out) {
  if (!cfg || !out) { snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: null argument"); return CZC_ERR_ARG; }
  if (cfg->bert_hidden != cfg->bert_heads * 64 || cfg->clip_hidden != cfg->clip_heads * 64 ||
      cfg->vis_hidden != cfg->vis_heads * 64) {
    snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: head_dim must be 64 for all towers");
    return CZC_ERR_ARG;
  }
  if (cfg->precision != CZC_PREC_BF16 && cfg->precision != CZC_PREC_F32 && cfg->precision != CZC_PREC_ALL_BF16 &&
      cfg->precision != CZC_PREC_SPLIT && cfg->precision != CZC_PREC_FP16 && cfg->precision != CZC_PREC_REFINE) {
    snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: unknown precision");
    return CZC_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: no HIP device visible (the engine has no CPU fallback)");
    return CZC_ERR_HIP;
  }
  if (device_id < 0 || device_id >= ndev) {
    snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: device %d out of range", device_id);
    return CZC_ERR_ARG;
  }
  czc_engine* e = new czc_engine();
  e->cfg = *cfg;
  e->dev = device_id;
  // precision 0: CLIP towers on bf16 MFMA; BERT on split-fp16 MFMA (hi+lo planes, three fp16 passes,
  // ~22 mantissa bits: the tau=0.1 softmax amplifies logit error tenfold, and BERT is 1.4% of the
  // FLOPs); 1: everything on f32 MFMA; 2: everything bf16 (experiments); 3: every tower on split-fp16 MFMA
  // 4: CLIP towers on single-pass fp16 MFMA (the bf16 kernels on IEEE fp16 operands), BERT on split-fp16
  // 5: screen-then-refine: CLIP-text screening pass on single-pass fp16, refine pass / vision tower / BERT on split-fp16
  e->refine = cfg->precision == CZC_PREC_REFINE;
  e->pc = cfg->precision == CZC_PREC_F32 ? PREC_F32
          : (cfg->precision == CZC_PREC_SPLIT ? PREC_F16X3
             : ((cfg->precision == CZC_PREC_FP16 || e->refine) ? PREC_F16 : PREC_BF16));
  e->pb = cfg->precision == CZC_PREC_ALL_BF16 ? PREC_BF16
                                              : (cfg->precision == CZC_PREC_F32 ? PREC_F32 : PREC_F16X3);
  e->pv = e->refine ? (int)PREC_F16X3 : e->pc;
  e->esz = prec_bytes(e->pc);
  e->eb = prec_bytes(e->pb);
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&e->st) != hipSuccess ||
      hipHostMalloc((void**)&e->h_totals, 256) != hipSuccess) {
    snprintf(czc::g_err, sizeof(czc::g_err), "czc_create: stream/host allocation failed");
    delete e;
    return CZC_ERR_HIP;
  }
  memset(&e->bd, 0, sizeof(e->bd));
  *out = e;
  return CZC_OK;
}

int czc_destroy(czc_engine* e) {
  if (!e) return CZC_OK;
  (void)hipSetDevice(e->dev);
  (void)hipStreamSynchronize(e->st);
  if (e->shares_weights) { e->w.clear(); e->bert.clear(); e->ctext.clear(); e->cvis.clear(); e->ctext_x.clear();
                           e->mlm_dense_w = e->decoder_w = e->tproj_w = e->vproj_w = e->patch_w = e->tproj_wx = nullptr; }
  for (auto& kv : e->w) if (kv.second.p) (void)hipFree(kv.second.p);
  free_layer_set(e->bert); free_layer_set(e->ctext); free_layer_set(e->cvis); free_layer_set(e->ctext_x);
  (void)hipFree(e->mlm_dense_w); (void)hipFree(e->decoder_w); (void)hipFree(e->tproj_w); (void)hipFree(e->vproj_w);
  (void)hipFree(e->patch_w); (void)hipFree(e->tproj_wx);
  for (auto& kv : e->ws) if (kv.second.p) (void)hipFree(kv.second.p);
  for (void* p : e->bridge_allocs) (void)hipFree(p);
  (void)hipFree(e->d_mask); (void)hipFree(e->d_lex); (void)hipFree(e->d_lex_pos); (void)hipFree(e->d_lex_cls); (void)hipFree(e->d_img_n); (void)hipFree(e->d_staged);
  (void)hipFree(e->d_pos_tags); (void)hipFree(e->d_pos_masks);
  for (auto& kv : e->pk) for (hipEvent_t ev : kv.second.ev) (void)hipEventDestroy(ev);
  if (e->prof_ref) (void)hipEventDestroy(e->prof_ref);
  if (e->h_totals) (void)hipHostFree(e->h_totals);
  if (e->h_ctl_ids) (void)hipHostFree(e->h_ctl_ids);
  (void)hipStreamDestroy(e->st);
  delete e;
  return CZC_OK;
}

// A second engine on the same GPU over the SAME resident weights: its own stream, workspace, image embeddings,
// token mask / bridge / lexicon tables, options and profile.  Two (or three) of them driven from separate host
// threads polish disjoint image sub-batches concurrently: one sub-batch's small BERT / top-K / LayerNorm / attention
// launches and the tail rounds of its persistent GEMMs fill what the other's big launches leave idle.
int czc_replicate(czc_engine* p, czc_engine** out) {
  if (!p || !out) return CZC_ERR_ARG;
  if (!p->finalized) return fail(p, CZC_ERR_STATE, "czc_replicate: czc_finalize_weights first%s");
  czc_engine* e = new czc_engine();
  e->cfg = p->cfg; e->dev = p->dev; e->finalized = true; e->shares_weights = true;
  e->esz = p->esz; e->pb = p->pb; e->pc = p->pc; e->pv = p->pv; e->eb = p->eb;
  e->refine = p->refine; e->refine_theta_x = p->refine_theta_x; e->refine_samples = p->refine_samples; e->refine_samples_step = p->refine_samples_step;
  e->refine_guard_dev = p->refine_guard_dev; e->refine_gate_delta = p->refine_gate_delta; e->refine_theta_gen = p->refine_theta_gen;
  e->refine_rows16 = p->refine_rows16; e->refine_rows16_factor = p->refine_rows16_factor;
  e->w = p->w; e->bert = p->bert; e->ctext = p->ctext; e->cvis = p->cvis; e->ctext_x = p->ctext_x;
  e->mlm_dense_w = p->mlm_dense_w; e->decoder_w = p->decoder_w; e->tproj_w = p->tproj_w; e->vproj_w = p->vproj_w;
  e->patch_w = p->patch_w; e->tproj_wx = p->tproj_wx;
  e->logit_scale_exp = p->logit_scale_exp;
  e->share_prefix = p->share_prefix; e->dedup = p->dedup; e->pack_branches = p->pack_branches; e->pool_last_layer = p->pool_last_layer;
  e->fuse_ln = p->fuse_ln; e->bert_prune = p->bert_prune; e->bert_fuse_splitk_ln = p->bert_fuse_splitk_ln; e->resid16 = p->resid16; e->fold_ln = p->fold_ln;
  memset(&e->bd, 0, sizeof(e->bd));
  if (hipSetDevice(e->dev) != hipSuccess || hipStreamCreate(&e->st) != hipSuccess ||
      hipHostMalloc((void**)&e->h_totals, 256) != hipSuccess) {
    e->w.clear();
    delete e;
    return fail(p, CZC_ERR_HIP, "czc_replicate: stream/host allocation failed%s");
  }
  *out = e;
  return CZC_OK;
}

int czc_load_tensor(czc_engine* e, const char* name, int dtype, int ndim, const int64_t* shape, const void* src) {
  if (e && e->shares_weights) return fail(e, CZC_ERR_STATE, "czc_load_tensor: a replica shares its parent's weights%s");
  if (!e || !name || !src || ndim < 0 || ndim > 8) return CZC_ERR_ARG;
  if (dtype != 0) return fail(e, CZC_ERR_ARG, "czc_load_tensor(%s): only fp32 (dtype 0) is accepted", name);
  E_HIP(hipSetDevice(e->dev));
  Tensor t;
  t.numel = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= (size_t)shape[i]; }
  auto it = e->w.find(name);
  if (it != e->w.end() && it->second.p) { (void)hipFree(it->second.p); }
  E_HIP(hipMalloc((void**)&t.p, t.numel * 4 + 16));
  E_HIP(hipMemcpy(t.p, src, t.numel * 4, hipMemcpyDefault));
  e->w[name] = t;
  e->finalized = false;
  return CZC_OK;
}

int czc_finalize_weights(czc_engine* e) {
  if (!e) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  const czc_config& c = e->cfg;
  const bool has_bert = c.bert_layers > 0 && c.bert_vocab > 0;
  if (e->shares_weights) return fail(e, CZC_ERR_STATE, "czc_finalize_weights: a replica shares its parent's weights%s");
  if (e->finalized) return CZC_OK;  // nothing loaded since the last call (czc_load_tensor clears the flag)
  E_HIP(hipStreamSynchronize(e->st));
  free_layer_set(e->bert); free_layer_set(e->ctext); free_layer_set(e->cvis); free_layer_set(e->ctext_x);
  for (void** p : {&e->mlm_dense_w, &e->decoder_w, &e->tproj_w, &e->vproj_w, &e->patch_w, &e->tproj_wx}) { (void)hipFree(*p); *p = nullptr; }
  if (has_bert) E_CHECK(build_layers(e, e->bert, c.bert_layers, c.bert_hidden, c.bert_inter, true, "bert.encoder.layer.", e->pb));
  E_CHECK(build_layers(e, e->ctext, c.clip_layers, c.clip_hidden, c.clip_inter, false, "text_model.encoder.layers.", e->pc));
  if (wants_folded_ln(e)) {  // folded-LayerNorm operands of the text tower (used where resid16 + fold_ln apply)
    const int H = c.clip_hidden, I = c.clip_inter;
    for (int n = 0; n < c.clip_layers; ++n) {
      LayerW& l = e->ctext[n];
      const std::string p = "text_model.encoder.layers." + std::to_string(n);
      float *qw, *kw, *vw, *f1w;
      E_CHECK(need(e, p + ".self_attn.q_proj.weight", (size_t)H * H, &qw));
      E_CHECK(need(e, p + ".self_attn.k_proj.weight", (size_t)H * H, &kw));
      E_CHECK(need(e, p + ".self_attn.v_proj.weight", (size_t)H * H, &vw));
      E_CHECK(need(e, p + ".mlp.fc1.weight", (size_t)I * H, &f1w));
      // (each pointer lands in the LayerW the moment it exists: czc_destroy / the next czc_finalize_weights frees whatever an
      // early return leaves behind)
      E_HIP(hipMalloc(&l.qkv_wf, (size_t)3 * H * H * 2));
      E_HIP(hipMalloc((void**)&l.qkv_bf, (size_t)3 * H * 4));
      E_HIP(hipMalloc(&l.fc1_wf, (size_t)I * H * 2));
      E_HIP(hipMalloc((void**)&l.fc1_bf, (size_t)I * 4));
      const float* srcw[3] = {qw, kw, vw};
      for (int t = 0; t < 3; ++t)
        E_CHECK(launch_fold_ln(srcw[t], l.ln1_g, l.ln1_b, l.qkv_b + t * H, H, H, (char*)l.qkv_wf + (size_t)t * H * H * 2, nullptr,
                               l.qkv_bf + t * H, e->st));
      E_CHECK(launch_fold_ln(f1w, l.ln2_g, l.ln2_b, l.fc1_b, I, H, l.fc1_wf, nullptr, l.fc1_bf, e->st));
    }
  }
  if (e->refine)
    E_CHECK(build_layers(e, e->ctext_x, c.clip_layers, c.clip_hidden, c.clip_inter, false, "text_model.encoder.layers.", PREC_F16X3));
  E_CHECK(build_layers(e, e->cvis, c.vis_layers, c.vis_hidden, c.vis_inter, false, "vision_model.encoder.layers.", e->pv));
  float* t;
  if (has_bert) {
    E_CHECK(need(e, "cls.predictions.transform.dense.weight", (size_t)c.bert_hidden * c.bert_hidden, &t));
    E_CHECK(to_act(e, e->pb, t, (size_t)c.bert_hidden * c.bert_hidden, &e->mlm_dense_w));
    // decoder weight tied to the word embeddings (HF:bert/modeling_bert.py:910-913)
    E_CHECK(need(e, "bert.embeddings.word_embeddings.weight", (size_t)c.bert_vocab * c.bert_hidden, &t));
    E_CHECK(to_act(e, e->pb, t, (size_t)c.bert_vocab * c.bert_hidden, &e->decoder_w));
    if (!find(e, "cls.predictions.bias") && find(e, "cls.predictions.decoder.bias")) {
      e->w["cls.predictions.bias"] = e->w["cls.predictions.decoder.bias"];
      e->w.erase("cls.predictions.decoder.bias");
    }
  }
  E_CHECK(need(e, "text_projection.weight", (size_t)c.clip_proj * c.clip_hidden, &t));
  E_CHECK(to_act(e, e->pc, t, (size_t)c.clip_proj * c.clip_hidden, &e->tproj_w));
  if (e->refine) E_CHECK(to_act(e, PREC_F16X3, t, (size_t)c.clip_proj * c.clip_hidden, &e->tproj_wx));
  E_CHECK(need(e, "visual_projection.weight", (size_t)c.clip_proj * c.vis_hidden, &t));
  E_CHECK(to_act(e, e->pv, t, (size_t)c.clip_proj * c.vis_hidden, &e->vproj_w));
  const size_t pk = (size_t)3 * c.vis_patch * c.vis_patch;
  E_CHECK(need(e, "vision_model.embeddings.patch_embedding.weight", (size_t)c.vis_hidden * pk, &t));
  E_CHECK(to_act(e, e->pv, t, (size_t)c.vis_hidden * pk, &e->patch_w));
  const Tensor* ls = find(e, "logit_scale");
  if (!ls) return fail(e, CZC_ERR_STATE, "missing tensor %s", "logit_scale");
  float lsv = 0.f;
  E_HIP(hipStreamSynchronize(e->st));
  E_HIP(hipMemcpy(&lsv, ls->p, 4, hipMemcpyDeviceToHost));
  e->logit_scale_exp = expf(lsv);
  // GEMM-operand masters are no longer needed (LN/bias/embedding tables stay fp32)
  std::vector<std::string> drop;
  for (auto& kv : e->w) {
    const std::string& n = kv.first;
    const bool is_w = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    const bool keep = n.find("LayerNorm") != std::string::npos || n.find("layer_norm") != std::string::npos ||
                      n.find("layrnorm") != std::string::npos || n.find("layernorm") != std::string::npos ||
                      n.find("embedding") != std::string::npos;
    if (is_w && !keep && kv.second.shape.size() == 2) drop.push_back(n);
  }
  for (auto& n : drop) { (void)hipFree(e->w[n].p); e->w.erase(n); }
  e->finalized = true;
  return CZC_OK;
}

int czc_set_token_mask(czc_engine* e, const float* mask, int vocab) {
  if (!e || !mask || vocab != e->cfg.bert_vocab) return e ? fail(e, CZC_ERR_ARG, "token mask size != bert_vocab%s") : CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  if (!e->d_mask) { E_HIP(hipMalloc((void**)&e->d_mask, (size_t)vocab * 4)); }
  E_HIP(hipMemcpy(e->d_mask, mask, (size_t)vocab * 4, hipMemcpyDefault));
  e->mask_vocab = vocab;
  return CZC_OK;
}

int czc_set_lexicon(czc_engine* e, const float* lex, int vocab) {
  if (!e || !lex || vocab != e->cfg.bert_vocab) return e ? fail(e, CZC_ERR_ARG, "lexicon size != bert_vocab%s") : CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  if (!e->d_lex) { E_HIP(hipMalloc((void**)&e->d_lex, (size_t)vocab * 4)); }
  E_HIP(hipMemcpy(e->d_lex, lex, (size_t)vocab * 4, hipMemcpyDefault));
  return CZC_OK;
}

int czc_set_lexicon_pos(czc_engine* e, const float* table, const uint8_t* class_of_token, int vocab) {
  if (!e) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  if (!table) {  // back to the per-token lexicon of czc_set_lexicon
    (void)hipFree(e->d_lex_pos); (void)hipFree(e->d_lex_cls);
    e->d_lex_pos = nullptr; e->d_lex_cls = nullptr;
    return CZC_OK;
  }
  if (!class_of_token || vocab != e->cfg.bert_vocab) return fail(e, CZC_ERR_ARG, "lexicon table size != bert_vocab%s");
  std::vector<uint8_t> cls(class_of_token, class_of_token + vocab);
  for (uint8_t c : cls) if (c > 4) return fail(e, CZC_ERR_ARG, "coarse POS class must be 0..4 ('' n v a r)%s");
  if (!e->d_lex_pos) E_HIP(hipMalloc((void**)&e->d_lex_pos, (size_t)vocab * 5 * 4));
  if (!e->d_lex_cls) E_HIP(hipMalloc((void**)&e->d_lex_cls, (size_t)vocab));
  E_HIP(hipMemcpy(e->d_lex_pos, table, (size_t)vocab * 5 * 4, hipMemcpyDefault));
  E_HIP(hipMemcpy(e->d_lex_cls, cls.data(), (size_t)vocab, hipMemcpyHostToDevice));
  return CZC_OK;
}

int czc_set_pos(czc_engine* e, const uint8_t* tag_of_token, int vocab, const uint16_t* template_masks, int n_template) {
  if (!e || !tag_of_token || !template_masks) return CZC_ERR_ARG;
  if (vocab != e->cfg.bert_vocab || n_template <= 0 || n_template > 32) return fail(e, CZC_ERR_ARG, "bad POS tables%s");
  E_HIP(hipSetDevice(e->dev));
  if (!e->d_pos_tags) E_HIP(hipMalloc((void**)&e->d_pos_tags, (size_t)vocab));
  if (!e->d_pos_masks) E_HIP(hipMalloc((void**)&e->d_pos_masks, 64));
  E_HIP(hipMemcpy(e->d_pos_tags, tag_of_token, (size_t)vocab, hipMemcpyDefault));
  E_HIP(hipMemcpy(e->d_pos_masks, template_masks, (size_t)n_template * 2, hipMemcpyDefault));
  e->pos_n = n_template;
  return CZC_OK;
}

static int upload(czc_engine* e, const void* src, size_t bytes, void** out) {
  void* p = nullptr;
  E_HIP(hipMalloc(&p, bytes + 16));
  E_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  e->bridge_allocs.push_back(p);
  *out = p;
  return 0;
}

static int build_bridge(czc_engine* e, const czc_bridge_tables* t, BridgeDev* bd) {
  if (t->bert_vocab <= 0 || t->n_merges < 0) return fail(e, CZC_ERR_ARG, "bad bridge tables%s");
  const size_t nbytes = t->piece_off[t->bert_vocab];
  void* p;
  bd->bert_vocab = t->bert_vocab;
  E_CHECK(upload(e, t->piece_off, (size_t)(t->bert_vocab + 1) * 4, &p)); bd->piece_off = (const uint32_t*)p;
  E_CHECK(upload(e, t->piece_bytes, nbytes ? nbytes : 1, &p)); bd->piece_bytes = (const uint8_t*)p;
  E_CHECK(upload(e, t->piece_class, nbytes ? nbytes : 1, &p)); bd->piece_class = (const uint8_t*)p;
  E_CHECK(upload(e, t->piece_flags, (size_t)t->bert_vocab, &p)); bd->piece_flags = (const uint8_t*)p;
  E_CHECK(upload(e, t->byte_sym, 256 * 4, &p)); bd->byte_sym = (const int*)p;
  E_CHECK(upload(e, t->byte_sym_eow, 256 * 4, &p)); bd->byte_sym_eow = (const int*)p;
  size_t cap = 1024;
  while (cap < (size_t)t->n_merges * 2 + 2) cap <<= 1;
  std::vector<unsigned long long> keys(cap, ~0ull), vals(cap, 0ull);
  for (int r = 0; r < t->n_merges; ++r) {
    const unsigned long long key = ((unsigned long long)(unsigned)t->merge_left[r] << 32) | (unsigned)t->merge_right[r];
    unsigned h = bridge_hash(key) & (unsigned)(cap - 1);
    bool dup = false;
    while (keys[h] != ~0ull) {
      if (keys[h] == key) { dup = true; break; }  // first (lowest) rank wins
      h = (h + 1) & (unsigned)(cap - 1);
    }
    if (dup) continue;
    keys[h] = key;
    vals[h] = ((unsigned long long)(unsigned)r << 32) | (unsigned)t->merge_out[r];
  }
  E_CHECK(upload(e, keys.data(), cap * 8, &p)); bd->hkeys = (const unsigned long long*)p;
  E_CHECK(upload(e, vals.data(), cap * 8, &p)); bd->hvals = (const unsigned long long*)p;
  bd->hmask = (unsigned)(cap - 1);
  bd->bos_id = t->bos_id;
  bd->eos_id = t->eos_id;
  // ids of every all-letter token standing alone as a word, tabulated once by the same BPE code (bridge.hip)
  bd->tok_bpe = nullptr; bd->tok_bpe_len = nullptr;
  int* tok_ids = nullptr; uint8_t* tok_len = nullptr;
  E_HIP(hipMalloc((void**)&tok_ids, (size_t)t->bert_vocab * BR_TOKMAX * 4 + 16));
  e->bridge_allocs.push_back(tok_ids);
  E_HIP(hipMalloc((void**)&tok_len, (size_t)t->bert_vocab + 16));
  e->bridge_allocs.push_back(tok_len);
  E_CHECK(launch_bridge_precompute(*bd, tok_ids, tok_len, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  bd->tok_bpe = tok_ids; bd->tok_bpe_len = tok_len;
  return 0;
}

int czc_set_bridge(czc_engine* e, const czc_bridge_tables* t) {
  if (!e || !t) return CZC_ERR_ARG;
  if (t->bert_vocab != e->cfg.bert_vocab) return fail(e, CZC_ERR_ARG, "bridge bert_vocab != config%s");
  E_HIP(hipSetDevice(e->dev));
  for (void* p : e->bridge_allocs) (void)hipFree(p);
  e->bridge_allocs.clear();
  E_CHECK(build_bridge(e, t, &e->bd));
  e->has_bridge = true;
  return CZC_OK;
}

int czc_set_image_embeds(czc_engine* e, const float* embeds, int B) {
  if (!e || !embeds || B <= 0) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  const int D = e->cfg.clip_proj;
  float* raw;
  E_CHECK(ensure(e, "img_raw", (size_t)B * D * 4, (void**)&raw));
  E_HIP(hipMemcpyAsync(raw, embeds, (size_t)B * D * 4, hipMemcpyDefault, e->st));
  if (e->img_B < B) { if (e->d_img_n) (void)hipFree(e->d_img_n); e->d_img_n = nullptr; E_HIP(hipMalloc((void**)&e->d_img_n, (size_t)B * D * 4)); }
  E_CHECK(launch_l2_normalize(raw, B, D, e->d_img_n, e->st));
  e->img_B = B;
  E_HIP(hipStreamSynchronize(e->st));
  return CZC_OK;
}

int czc_encode_images(czc_engine* e, const float* pixels, int B, float* out_embeds) {
  if (!e || !pixels || B <= 0) return CZC_ERR_ARG;
  if (!e->finalized) return fail(e, CZC_ERR_STATE, "weights not finalized%s");
  E_HIP(hipSetDevice(e->dev));
  const czc_config& c = e->cfg;
  const int P = e->pv, S = c.vis_image, p = c.vis_patch, G = S / p, NP = G * G, H = c.vis_hidden;
  const size_t vsz = prec_bytes(P);
  const int Kc = 3 * p * p, T = NP + 1, M = B * T;
  float *pix, *pe, *x, *emb; void *patches, *ca; int* idx;
  E_CHECK(ensure(e, "v_pix", (size_t)B * 3 * S * S * 4, (void**)&pix));
  E_CHECK(ensure(e, "v_patches", (size_t)B * NP * Kc * vsz, &patches));
  E_CHECK(ensure(e, "v_pe", (size_t)B * NP * H * 4, (void**)&pe));
  E_CHECK(ensure(e, "v_x", (size_t)M * H * 4, (void**)&x));
  E_CHECK(ensure(e, "v_idx", (size_t)B * 4, (void**)&idx));
  E_CHECK(ensure(e, "v_ca", (size_t)B * H * vsz, &ca));
  E_CHECK(ensure(e, "img_raw", (size_t)B * c.clip_proj * 4, (void**)&emb));
  E_HIP(hipMemcpyAsync(pix, pixels, (size_t)B * 3 * S * S * 4, hipMemcpyDefault, e->st));
  float *cls, *pos, *g0, *b0, *g1, *b1;
  E_CHECK(need(e, "vision_model.embeddings.class_embedding", H, &cls));
  E_CHECK(need(e, "vision_model.embeddings.position_embedding.weight", (size_t)T * H, &pos));
  E_CHECK(need(e, "vision_model.pre_layrnorm.weight", H, &g0));
  E_CHECK(need(e, "vision_model.pre_layrnorm.bias", H, &b0));
  E_CHECK(need(e, "vision_model.post_layernorm.weight", H, &g1));
  E_CHECK(need(e, "vision_model.post_layernorm.bias", H, &b1));
  { ProfScope ps(e, "rowops", 0); E_CHECK(launch_im2col(P, pix, B, S, p, patches, e->st)); }
  E_CHECK(gemm(e, P, "gemm_vision", patches, Kc, e->patch_w, Kc, nullptr, nullptr, 0, nullptr, pe, H, B * NP, H, Kc, ACT_NONE));
  { ProfScope ps(e, "rowops", 0);
    E_CHECK(launch_vision_assemble(pe, B, NP, H, cls, pos, x, e->st));
    E_CHECK(launch_layernorm(P, x, nullptr, g0, b0, c.clip_eps, M, H, nullptr, x, e->st)); }
  SegTable vtab{nullptr, nullptr, nullptr, nullptr, B, T};
  E_CHECK(clip_stack(e, P, "gemm_vision", e->cvis, x, M, H, c.vis_inter, c.vis_heads, c.clip_eps, vtab, T, 0));
  { ProfScope ps(e, "rowops", 0);
    E_CHECK(launch_make_row_index(idx, B, T, 0, e->st));
    E_CHECK(launch_layernorm(P, x, idx, g1, b1, c.clip_eps, B, H, ca, nullptr, e->st)); }
  E_CHECK(gemm(e, P, "gemm_vision", ca, H, e->vproj_w, H, nullptr, nullptr, 0, nullptr, emb, c.clip_proj, B, c.clip_proj, H, ACT_NONE));
  if (e->img_B < B) { if (e->d_img_n) (void)hipFree(e->d_img_n); e->d_img_n = nullptr; E_HIP(hipMalloc((void**)&e->d_img_n, (size_t)B * c.clip_proj * 4)); }
  E_CHECK(launch_l2_normalize(emb, B, c.clip_proj, e->d_img_n, e->st));
  e->img_B = B;
  if (out_embeds) E_HIP(hipMemcpyAsync(out_embeds, emb, (size_t)B * c.clip_proj * 4, hipMemcpyDefault, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  return CZC_OK;
}

int czc_preprocess_u8(czc_engine* e, const uint8_t* rgb, int height, int width, const float* mean, const float* stdv,
                      int slot, float* pixels_out) {
  if (!e || !rgb || !mean || !stdv || slot < 0) return CZC_ERR_ARG;
  if (height <= 0 || width <= 0 || (long)height * width > (1L << 28)) return fail(e, CZC_ERR_ARG, "preprocess: bad image size%s");
  E_HIP(hipSetDevice(e->dev));
  const int S = e->cfg.vis_image;
  const size_t img = (size_t)3 * S * S;
  // staged batch grows by doubling and keeps its contents
  if (slot >= e->staged_cap) {
    int cap = e->staged_cap ? e->staged_cap : 8;
    while (cap <= slot) cap *= 2;
    float* nb = nullptr;
    E_HIP(hipMalloc((void**)&nb, (size_t)cap * img * 4));
    if (e->d_staged) {
      E_HIP(hipMemcpyAsync(nb, e->d_staged, (size_t)e->staged_cap * img * 4, hipMemcpyDeviceToDevice, e->st));
      E_HIP(hipStreamSynchronize(e->st));
      E_HIP(hipFree(e->d_staged));
    }
    e->d_staged = nb;
    e->staged_cap = cap;
  }
  unsigned char *d_rgb, *scratch;
  E_CHECK(ensure(e, "pp_rgb", (size_t)height * width * 3, (void**)&d_rgb));
  E_CHECK(ensure(e, "pp_scratch", imageproc_scratch_bytes(height, width, S), (void**)&scratch));
  E_HIP(hipMemcpyAsync(d_rgb, rgb, (size_t)height * width * 3, hipMemcpyDefault, e->st));
  float* dst = e->d_staged + (size_t)slot * img;
  {
    ProfScope ps(e, "rowops", 0);
    if (launch_clip_preprocess(d_rgb, height, width, S, mean, stdv, scratch, dst, e->st)) return fail(e, CZC_ERR_HIP, "%s", g_err);
  }
  if (pixels_out) E_HIP(hipMemcpyAsync(pixels_out, dst, img * 4, hipMemcpyDefault, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  if (slot + 1 > e->staged_n) e->staged_n = slot + 1;
  return CZC_OK;
}

int czc_encode_staged(czc_engine* e, int B, float* out_embeds) {
  if (!e || B <= 0) return CZC_ERR_ARG;
  if (B > e->staged_n) return fail(e, CZC_ERR_STATE, "encode_staged: fewer staged images than requested%s");
  return czc_encode_images(e, e->d_staged, B, out_embeds);
}

int czc_encode_text(czc_engine* e, const int32_t* clip_ids, const int32_t* clip_len, int n, float* out_embeds) {
  if (!e || !clip_ids || !clip_len || n <= 0 || !out_embeds) return CZC_ERR_ARG;
  if (!e->finalized) return fail(e, CZC_ERR_STATE, "weights not finalized%s");
  E_HIP(hipSetDevice(e->dev));
  e->err[0] = 0;
  int *cids, *clen, *totals;
  E_CHECK(ensure(e, "s_cids", (size_t)n * CZC_CLIP_MAX_LEN * 4, (void**)&cids));
  E_CHECK(ensure(e, "s_clen", (size_t)n * 4, (void**)&clen));
  E_CHECK(ensure(e, "s_tot", 32, (void**)&totals));
  E_HIP(hipMemcpyAsync(cids, clip_ids, (size_t)n * CZC_CLIP_MAX_LEN * 4, hipMemcpyDefault, e->st));
  E_HIP(hipMemcpyAsync(clen, clip_len, (size_t)n * 4, hipMemcpyDefault, e->st));
  E_HIP(hipMemsetAsync(totals, 0, 32, e->st));
  { int* flag; E_CHECK(ensure(e, "s_nonfinite", 32, (void**)&flag)); E_HIP(hipMemsetAsync(flag, 0, 32, e->st)); }
  float* feat;
  E_CHECK(clip_text_forward(e, cids, clen, n, 1, 0, totals, &feat, true));  // independent sequences: no sharing; exact tower
  E_HIP(hipMemcpyAsync(out_embeds, feat, (size_t)n * e->cfg.clip_proj * 4, hipMemcpyDefault, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  return CZC_OK;
}

// clip/clip.py:86-98 `compute_image_text_similarity_via_embeddings` as a call of its own (the per-step path never needs it:
// czc_step / czc_generate fuse it into the score-combine kernel; this is the drop-in CLIP wrapper's public method): both sides
// L2-normalised, cos = t . i, logits = cos * exp(logit_scale) of the loaded checkpoint, softmax over the K texts of an image.
int czc_similarity(czc_engine* e, const float* image_embeds, const float* text_embeds, int B, int K, float* clip_score,
                   float* clip_ref) {
  if (!e || !image_embeds || !text_embeds || B <= 0 || K <= 0 || K > CZC_MAX_TOPK) return e ? fail(e, CZC_ERR_ARG, "similarity: bad B / K%s") : CZC_ERR_ARG;
  if (!e->finalized) return fail(e, CZC_ERR_STATE, "weights not finalized%s");
  E_HIP(hipSetDevice(e->dev));
  e->err[0] = 0;
  const int D = e->cfg.clip_proj;
  const size_t bk = (size_t)B * K;
  float *tf, *ie, *in_, *zero, *cs, *cr, *fin, *bc; int *cand, *best, *flag;
  E_CHECK(ensure(e, "sim_tf", bk * D * 4, (void**)&tf));
  E_CHECK(ensure(e, "sim_ie", (size_t)B * D * 4, (void**)&ie));
  E_CHECK(ensure(e, "sim_in", (size_t)B * D * 4, (void**)&in_));
  E_CHECK(ensure(e, "sim_zero", bk * 4, (void**)&zero));
  E_CHECK(ensure(e, "sim_cand", bk * 4, (void**)&cand));
  E_CHECK(ensure(e, "sim_cs", bk * 4, (void**)&cs));
  E_CHECK(ensure(e, "sim_cr", bk * 4, (void**)&cr));
  E_CHECK(ensure(e, "sim_fin", bk * 4, (void**)&fin));
  E_CHECK(ensure(e, "sim_best", (size_t)B * 4, (void**)&best));
  E_CHECK(ensure(e, "sim_bc", (size_t)B * 4, (void**)&bc));
  E_CHECK(ensure(e, "sim_flag", 32, (void**)&flag));
  E_HIP(hipMemcpyAsync(tf, text_embeds, bk * D * 4, hipMemcpyDefault, e->st));
  E_HIP(hipMemcpyAsync(ie, image_embeds, (size_t)B * D * 4, hipMemcpyDefault, e->st));
  E_HIP(hipMemsetAsync(zero, 0, bk * 4, e->st));
  E_HIP(hipMemsetAsync(cand, 0, bk * 4, e->st));
  E_HIP(hipMemsetAsync(flag, 0, 32, e->st));
  E_CHECK(launch_l2_normalize(ie, B, D, in_, e->st));
  CombineArgs a;
  a.text_feat = tf; a.img_n = in_; a.logit_scale_exp = e->logit_scale_exp; a.probs = zero; a.cand = cand;
  a.senti_raw = nullptr; a.repeats = nullptr; a.alpha = 0.f; a.beta = 1.f; a.gamma = 0.f; a.use_senti = 0;
  a.B = B; a.K = K; a.D = D; a.clip_score = cs; a.clip_ref = cr; a.final_score = fin; a.best = best; a.best_cos = bc;
  a.inp = nullptr; a.T = 0; a.gen_idx = 0; a.nonfinite = flag;
  E_CHECK(launch_combine(a, e->st));
  if (clip_score) E_HIP(hipMemcpyAsync(clip_score, cs, bk * 4, hipMemcpyDefault, e->st));
  if (clip_ref) E_HIP(hipMemcpyAsync(clip_ref, cr, bk * 4, hipMemcpyDefault, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  return CZC_OK;
}

int czc_step(czc_engine* e, int32_t* inp, int B, int T, int gen_idx, int n_mask, int dot_allowed, int top_k,
             const czc_hyper* hp, const czc_step_out* out) {
  if (!e || !inp || !hp || B <= 0) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  e->err[0] = 0;
  int* d_inp;
  { int* flag; E_CHECK(ensure(e, "s_nonfinite", 32, (void**)&flag)); E_HIP(hipMemsetAsync(flag, 0, 32, e->st)); }
  E_CHECK(ensure(e, "g_inp", (size_t)B * T * 4, (void**)&d_inp));
  E_HIP(hipMemcpyAsync(d_inp, inp, (size_t)B * T * 4, hipMemcpyDefault, e->st));
  e->gate_now = false;  // parity granularity: every one of the K fused scores is an output, all of them are refined
  e->in_generate = false;
  E_CHECK(step_device(e, d_inp, B, T, gen_idx, n_mask, dot_allowed, top_k, hp));
  const size_t bk = (size_t)B * top_k;
  if (out) {
    E_CHECK(copy_out(e, out->probs, "s_probs", bk * 4));
    E_CHECK(copy_out(e, out->idxs, "s_idxs", bk * 4));
    E_CHECK(copy_out(e, out->cand_ids, "s_cand", bk * 4));
    E_CHECK(copy_out(e, out->clip_ids, "s_cids", bk * CZC_CLIP_MAX_LEN * 4));
    E_CHECK(copy_out(e, out->clip_len, "s_clen", bk * 4));
    E_CHECK(copy_out(e, out->clip_score, "s_cscore", bk * 4));
    E_CHECK(copy_out(e, out->clip_ref, "s_cref", bk * 4));
    E_CHECK(copy_out(e, out->senti_raw, "s_senti", bk * 4));
    E_CHECK(copy_out(e, out->repeats, "s_reps", bk * 4));
    E_CHECK(copy_out(e, out->final_score, "s_fin", bk * 4));
    E_CHECK(copy_out(e, out->best, "s_best", (size_t)B * 4));
    E_CHECK(copy_out(e, out->best_cos, "s_bcos", (size_t)B * 4));
    E_CHECK(copy_out(e, out->logits, "b_logits", (size_t)B * e->cfg.bert_vocab * 4));
  }
  E_HIP(hipMemcpyAsync(inp, d_inp, (size_t)B * T * 4, hipMemcpyDefault, e->st));
  E_HIP(hipMemcpyAsync(e->h_totals + 9, e->ws["s_nonfinite"].p, 24, hipMemcpyDeviceToHost, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  if (e->h_totals[9]) return fail(e, CZC_ERR_OVERFLOW, "non-finite CLIP cosine: an fp16 quantity overflowed in a tower (fp16 residual rows: set option resid16 = 0 on the bf16 engine, refine_rows16 = 0 on the refine engine; fp16 operands: use CZC_PREC_SPLIT)%s");
  { float dev; memcpy(&dev, e->h_totals + 10, 4); e->guard_max_dev = fmaxf(e->guard_max_dev, dev); e->guard_trips += e->h_totals[11];
    e->stat_gated += e->h_totals[13]; e->stat_gate_images += e->h_totals[14]; }
  return CZC_OK;
}

int czc_generate(czc_engine* e, int B, int T, int L, int seed_len, const int32_t* init_ids_host, int top_k,
                 int n_steps, const int32_t* positions_host, const int32_t* n_mask_host, int snapshot_every,
                 const czc_hyper* hp, int32_t* out_ids, float* out_cos) {
  if (!e || !init_ids_host || !positions_host || !hp || B <= 0 || n_steps < 0 || snapshot_every <= 0)
    return CZC_ERR_ARG;
  if (seed_len + L > T) return fail(e, CZC_ERR_ARG, "generate: seed_len + L > T%s");
  E_HIP(hipSetDevice(e->dev));
  e->err[0] = 0;
  int *d_inp, *d_row;
  { int* flag; E_CHECK(ensure(e, "s_nonfinite", 32, (void**)&flag)); E_HIP(hipMemsetAsync(flag, 0, 32, e->st)); }
  E_CHECK(ensure(e, "g_inp", (size_t)B * T * 4, (void**)&d_inp));
  E_CHECK(ensure(e, "g_row", (size_t)T * 4, (void**)&d_row));
  E_HIP(hipMemcpyAsync(d_row, init_ids_host, (size_t)T * 4, hipMemcpyHostToDevice, e->st));
  E_CHECK(launch_broadcast_rows_i32(d_row, T, B, d_inp, e->st));
  int snap = 0;
  bool audited = false;
  for (int s = 0; s < n_steps; ++s) {
    const int pos = positions_host[s];
    if (pos < 0 || pos >= L) return fail(e, CZC_ERR_ARG, "generate: position out of range%s");
    const int nm = n_mask_host ? n_mask_host[s] : 1;
    // what this call returns of a step: the ids it leaves in d_inp and, at the snapshot steps, the winner's cosine
    // AUDIT steps -- the snapshot step of every fourth sweep, starting with the first -- take the full selection for every
    // image: there the guard (czc_refine_guard) measures the screening tower on all images although most other steps are
    // gated (a checkpoint the fp16 tower carries badly trips it in the first sweep).  At the other snapshot steps a gated
    // image re-encodes its winner alone, for the cosine this call returns.  The gate is the guard's dependant: with the guard
    // switched off (refine_guard_x1e6 = 0) nothing polices the bound the gate rests on, so nothing is gated
    const bool snap_idx = (s + 1) % snapshot_every == 0;
    const bool snap_step = snap_idx && out_cos != nullptr;
    // (whether or not the caller reads cosines: the guard must see the screening tower on every call -- and a call too short
    // to reach a snapshot step is audited at its last step)
    const bool audit = (snap_idx && (s / snapshot_every) % 4 == 0) || (s + 1 == n_steps && !audited);
    audited = audited || audit;
    e->gate_now = e->refine && e->refine_gate_delta > 0.f && e->refine_guard_dev > 0.f && !audit;
    e->gate_need_cos = snap_step;
    e->in_generate = true;
    E_CHECK(step_device(e, d_inp, B, T, seed_len + pos, nm, pos == L - 1 ? 1 : 0, top_k, hp));
    if ((s + 1) % snapshot_every == 0) {
      if (out_ids)
        E_HIP(hipMemcpyAsync(out_ids + (size_t)snap * B * T, d_inp, (size_t)B * T * 4, hipMemcpyDefault, e->st));
      if (out_cos)
        E_HIP(hipMemcpyAsync(out_cos + (size_t)snap * B, e->ws["s_bcos"].p, (size_t)B * 4, hipMemcpyDefault, e->st));
      ++snap;
    }
  }
  E_HIP(hipMemcpyAsync(e->h_totals + 9, e->ws["s_nonfinite"].p, 24, hipMemcpyDeviceToHost, e->st));
  E_HIP(hipStreamSynchronize(e->st));
  if (e->h_totals[9]) return fail(e, CZC_ERR_OVERFLOW, "non-finite CLIP cosine: an fp16 quantity overflowed in a tower (fp16 residual rows: set option resid16 = 0 on the bf16 engine, refine_rows16 = 0 on the refine engine; fp16 operands: use CZC_PREC_SPLIT)%s");
  { float dev; memcpy(&dev, e->h_totals + 10, 4); e->guard_max_dev = fmaxf(e->guard_max_dev, dev); e->guard_trips += e->h_totals[11];
    e->stat_gated += e->h_totals[13]; e->stat_gate_images += e->h_totals[14]; }
  return CZC_OK;
}

int czc_set_control_callback(czc_engine* e, czc_control_fn fn, void* user) {
  if (!e) return CZC_ERR_ARG;
  e->ctl_fn = fn;
  e->ctl_user = fn ? user : nullptr;
  return CZC_OK;
}

int czc_set_option(czc_engine* e, const char* name, int value) {
  if (!e || !name) return CZC_ERR_ARG;
  if (!strcmp(name, "share_prefix")) { e->share_prefix = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "dedup")) { e->dedup = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "bert_prune")) { e->bert_prune = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "bert_fuse_splitk_ln")) { e->bert_fuse_splitk_ln = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "pack_branches")) { e->pack_branches = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "pool_last_layer")) { e->pool_last_layer = value ? 1 : 0; return CZC_OK; }
  if (!strcmp(name, "fuse_ln")) { e->fuse_ln = value; return CZC_OK; }  // 0 off, 1 out-proj -> LN2, 2 also fc2 -> next LN1
  const auto fold_ready = [&]() -> int {  // an option that turns the folded form on after the weights were finalized without it
    if (e->finalized && wants_folded_ln(e) && (e->ctext.empty() || !e->ctext[0].qkv_wf))
      return fail(e, CZC_ERR_STATE, "option %s: this engine was finalized without the folded-LayerNorm operands (set it before czc_finalize_weights)", name);
    return CZC_OK;
  };
  if (!strcmp(name, "resid16")) { const int old = e->resid16; e->resid16 = value < 0 ? 0 : value; const int rc = fold_ready(); if (rc) e->resid16 = old; return rc; }
  if (!strcmp(name, "fold_ln")) { const int old = e->fold_ln; e->fold_ln = value ? 1 : 0; const int rc = fold_ready(); if (rc) e->fold_ln = old; return rc; }
  if (!strcmp(name, "refine_samples")) { e->refine_samples = value < 0 ? 0 : value; return CZC_OK; }
  if (!strcmp(name, "refine_samples_step")) { e->refine_samples_step = value < 0 ? 0 : value; return CZC_OK; }
  if (!strcmp(name, "refine_theta_x1000")) { e->refine_theta_x = (float)value / 1000.f; return CZC_OK; }
  if (!strcmp(name, "refine_theta_gen_x1000")) { e->refine_theta_gen = (float)value / 1000.f; return CZC_OK; }
  if (!strcmp(name, "refine_guard_x1e6")) { e->refine_guard_dev = (float)value * 1e-6f; return CZC_OK; }
  if (!strcmp(name, "refine_gate_x1e6")) { e->refine_gate_delta = value < 0 ? 0.f : (float)value * 1e-6f; return CZC_OK; }
  if (!strcmp(name, "refine_rows16")) { const int old = e->refine_rows16; e->refine_rows16 = value ? 1 : 0; const int rc = fold_ready(); if (rc) e->refine_rows16 = old; return rc; }
  if (!strcmp(name, "refine_rows16_x1000")) { e->refine_rows16_factor = value < 1000 ? 1.f : (float)value / 1000.f; return CZC_OK; }
  return fail(e, CZC_ERR_ARG, "unknown option %s", name);
}

int czc_get_option(czc_engine* e, const char* name, int* value) {
  if (!e || !name || !value) return CZC_ERR_ARG;
  const bool r16 = e->refine && e->refine_rows16 && e->cfg.clip_hidden == 512;  // what czc_generate's screening pass runs on
  const float f16x = r16 ? e->refine_rows16_factor : 1.f;
  struct { const char* n; int v; } tab[] = {
      {"share_prefix", e->share_prefix}, {"bert_prune", e->bert_prune}, {"pack_branches", e->pack_branches},
      {"dedup", e->dedup}, {"pool_last_layer", e->pool_last_layer}, {"bert_fuse_splitk_ln", e->bert_fuse_splitk_ln}, {"fuse_ln", e->fuse_ln}, {"resid16", e->resid16}, {"fold_ln", e->fold_ln},
      {"refine_samples", e->refine_samples}, {"refine_samples_step", e->refine_samples_step}, {"refine_theta_x1000", (int)lrintf(e->refine_theta_x * 1000.f)},
      {"refine_theta_gen_x1000", (int)lrintf(e->refine_theta_gen * 1000.f)},
      {"refine_guard_x1e6", (int)lrintf(e->refine_guard_dev * 1e6f)}, {"refine_gate_x1e6", (int)lrintf(e->refine_gate_delta * 1e6f)},
      {"refine_rows16", e->refine_rows16}, {"refine_rows16_x1000", (int)lrintf(e->refine_rows16_factor * 1000.f)},
      // read-only, derived: the trip point / gate bound in force inside czc_generate (x refine_rows16_factor on fp16 rows)
      {"refine_guard_generate_x1e6", (int)lrintf(e->refine_guard_dev * f16x * 1e6f)},
      {"refine_gate_generate_x1e6", (int)lrintf(e->refine_gate_delta * f16x * 1e6f)},
      {"has_folded_ln_weights", !e->ctext.empty() && e->ctext[0].qkv_wf ? 1 : 0},
  };
  for (auto& t : tab) if (!strcmp(name, t.n)) { *value = t.v; return CZC_OK; }
  return fail(e, CZC_ERR_ARG, "unknown option %s", name);
}

int czc_sync(czc_engine* e) {
  if (!e) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  E_HIP(hipStreamSynchronize(e->st));
  return CZC_OK;
}

int czc_profile_enable(czc_engine* e, int on) {
  if (!e) return CZC_ERR_ARG;
  e->prof = on < 0 ? 0 : (on > 2 ? 1 : on);
  return CZC_OK;
}

int czc_profile_reset(czc_engine* e) {
  if (!e) return CZC_ERR_ARG;
  (void)hipSetDevice(e->dev);  // the current device is per host thread (events are created on it)
  (void)hipStreamSynchronize(e->st);
  if (!e->prof_ref) (void)hipEventCreate(&e->prof_ref);
  if (e->prof_ref) (void)hipEventRecord(e->prof_ref, e->st);
  for (auto& kv : e->pk) { kv.second.used = 0; kv.second.flops = 0; kv.second.launches = 0; }
  e->stat_clip_rows = e->stat_clip_seqs = e->stat_bert_rows = e->stat_steps = 0;
  e->stat_refine_rows = e->stat_refine_seqs = 0;
  e->stat_dedup_seqs = 0;
  e->stat_gated = e->stat_gate_images = 0;
  return CZC_OK;
}

int czc_profile_get(czc_engine* e, const char* kind, double* total_ms, int64_t* launches, double* flops) {
  if (!e || !kind) return CZC_ERR_ARG;
  E_HIP(hipSetDevice(e->dev));
  E_HIP(hipStreamSynchronize(e->st));
  double ms = 0;
  int64_t n = 0;
  double fl = 0;
  auto it = e->pk.find(kind);
  if (it != e->pk.end()) {
    ProfKind& k = it->second;
    for (size_t i = 0; i + 1 < k.used; i += 2) {
      float t = 0;
      if (hipEventElapsedTime(&t, k.ev[i], k.ev[i + 1]) == hipSuccess) ms += t;
    }
    n = k.launches;
    fl = k.flops;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  return CZC_OK;
}

// Start / end (ms) of every launch of kernel class `kind` since czc_profile_reset(e), on the clock whose zero is
// czc_profile_reset(ref) (ref = e for one engine; the same ref for all engines that ran concurrently, so that the
// host can take the union of their intervals: the time the GPU spent on that class with the launches overlapping).
int czc_profile_intervals(czc_engine* e, czc_engine* ref, const char* kind, double* start_ms, double* end_ms, int cap,
                          int* n_out) {
  if (!e || !ref || !kind || !n_out || cap < 0 || (cap > 0 && (!start_ms || !end_ms))) return CZC_ERR_ARG;
  if (!ref->prof_ref) return fail(e, CZC_ERR_STATE, "czc_profile_intervals: czc_profile_reset the reference engine first%s");
  if (ref->dev != e->dev) return fail(e, CZC_ERR_ARG, "czc_profile_intervals: engines on different devices%s");
  E_HIP(hipSetDevice(e->dev));
  E_HIP(hipStreamSynchronize(e->st));
  E_HIP(hipStreamSynchronize(ref->st));
  int n = 0;
  auto it = e->pk.find(kind);
  if (it != e->pk.end()) {
    ProfKind& k = it->second;
    for (size_t i = 0; i + 1 < k.used; i += 2, ++n) {
      if (n >= cap) continue;
      float a = 0, b = 0;
      E_HIP(hipEventElapsedTime(&a, ref->prof_ref, k.ev[i]));
      E_HIP(hipEventElapsedTime(&b, ref->prof_ref, k.ev[i + 1]));
      start_ms[n] = a; end_ms[n] = b;
    }
  }
  *n_out = n;
  return CZC_OK;
}

int czc_stats(czc_engine* e, int64_t* clip_rows, int64_t* clip_seqs, int64_t* bert_rows, int64_t* steps) {
  if (!e) return CZC_ERR_ARG;
  if (clip_rows) *clip_rows = e->stat_clip_rows;
  if (clip_seqs) *clip_seqs = e->stat_clip_seqs;
  if (bert_rows) *bert_rows = e->stat_bert_rows;
  if (steps) *steps = e->stat_steps;
  return CZC_OK;
}

int czc_dedup_stats(czc_engine* e, int64_t* dedup_seqs, int64_t* clip_seqs) {
  if (!e) return CZC_ERR_ARG;
  if (dedup_seqs) *dedup_seqs = e->stat_dedup_seqs;
  if (clip_seqs) *clip_seqs = e->stat_clip_seqs;
  return CZC_OK;
}

int czc_refine_stats(czc_engine* e, int64_t* refine_seqs, int64_t* refine_rows) {
  if (!e) return CZC_ERR_ARG;
  if (refine_seqs) *refine_seqs = e->stat_refine_seqs;
  if (refine_rows) *refine_rows = e->stat_refine_rows;
  return CZC_OK;
}

int czc_refine_gate_stats(czc_engine* e, int64_t* gated, int64_t* image_steps) {
  if (!e) return CZC_ERR_ARG;
  if (gated) *gated = e->stat_gated;
  if (image_steps) *image_steps = e->stat_gate_images;
  return CZC_OK;
}

int czc_refine_guard(czc_engine* e, int reset, float* max_dev, int64_t* tripped) {
  if (!e) return CZC_ERR_ARG;
  if (max_dev) *max_dev = e->guard_max_dev;
  if (tripped) *tripped = e->guard_trips;
  if (reset) { e->guard_max_dev = 0.f; e->guard_trips = 0; }
  return CZC_OK;
}

const void* czc_internal_hooks(int abi) {
  static const Hooks h = {
      []() -> char* { return czc::g_err; },
      &launch_gemm, &launch_gemm_rowln, &launch_layernorm, &launch_convert, &launch_act_to_f32, &launch_attention,
      &launch_softmax_mask_topk, &launch_bridge_precompute, &launch_bridge, &launch_l2_normalize, &launch_combine,
      &launch_layernorm_x16, &launch_ln_finalize, &launch_fold_ln, &gemm_wreg_stats_in_kernel,
      &g_use_gemm256, &g_use_skinny, &g_use_splitk, &g_gemm_deep, &g_gemm_small_tiles, &g_use_wreg, &g_use_gemm256s, &g_w_dbg,
      &g_ln_lean, &g_rowln_min_m, &g_wreg_min_m, &g_gemm256_min_m, &g_use_mfma_attention, &g_use_attention_image, &g_wreg_resid_min_m,
      &g_wreg_stats_in_kernel, &g_gemm256s_min_m};
  return abi == HOOKS_ABI ? &h : nullptr;
}

}  // extern "C"

// Kernel-level parity hooks of include/conzic_hip_test.h (czc_test_*): run ONE kernel on host data.
// Used only by tests/ (-m gpu) to compare each HIP kernel with the CPU oracle, and by tools/ for kernel A/B timing.
// Built into libconzic_hip_test.so, which links against the product library's C ABI only (czc_internal_hooks).
#include <vector>
#include <cstring>
#include <algorithm>
#include <cstdlib>

#include "../../include/conzic_hip_test.h"
#include "kernels.h"
#include "bridge_hash.h"

#include "../../include/conzic_hip.h"

using namespace czc;

// everything this file needs from inside libconzic_hip.so comes through its one internal door (the product library
// exports its C ABI and nothing else): launchers, the calling thread's error buffer, the kernel-family switches
static const Hooks& HK() {
  static const Hooks* h = (const Hooks*)czc_internal_hooks(HOOKS_ABI);
  if (!h) { fprintf(stderr, "libconzic_hip_test.so: libconzic_hip.so was built from another tree (hooks ABI)\n"); abort(); }
  return *h;
}
#define launch_gemm HK().gemm
#define launch_gemm_rowln HK().gemm_rowln
#define launch_layernorm HK().layernorm
#define launch_convert HK().convert
#define launch_act_to_f32 HK().act_to_f32
#define launch_attention HK().attention
#define launch_softmax_mask_topk HK().softmax_mask_topk
#define launch_bridge_precompute HK().bridge_precompute
#define launch_bridge HK().bridge
#define launch_l2_normalize HK().l2_normalize
#define launch_combine HK().combine
#define launch_layernorm_x16 HK().layernorm_x16
#define launch_ln_finalize HK().ln_finalize
#define launch_fold_ln HK().fold_ln
#define gemm_wreg_stats_in_kernel HK().wreg_stats_ok
#define g_use_gemm256 (*HK().use_gemm256)
#define g_use_skinny (*HK().use_skinny)
#define g_use_splitk (*HK().use_splitk)
#define g_gemm_deep (*HK().gemm_deep)
#define g_gemm_small_tiles (*HK().gemm_small_tiles)
#define g_use_wreg (*HK().use_wreg)
#define g_use_gemm256s (*HK().use_gemm256s)
#define g_w_dbg (*HK().w_dbg)
#define g_ln_lean (*HK().ln_lean)
#define g_rowln_min_m (*HK().rowln_min_m)
#define g_wreg_min_m (*HK().wreg_min_m)
#define g_gemm256_min_m (*HK().gemm256_min_m)
#define g_gemm256s_min_m (*HK().gemm256s_min_m)
#define g_use_mfma_attention (*HK().use_mfma_attention)
#define g_use_attention_image (*HK().use_attention_image)
#define g_wreg_resid_min_m (*HK().wreg_resid_min_m)
#define g_wreg_stats_in_kernel (*HK().wreg_stats_in_kernel)
#define TEST_ERR (HK().err_buf())

static int g_bench_pad = 0;
static int g_bridge_no_table = 0;  // czc_test_bridge: 1 = every chunk through the merge loop (option "bridge_no_table")  // czc_bench_gemm: extra elements per row of A and W (row pitch vs L2 channel experiments)

namespace {

struct DevPool {
  std::vector<void*> ptrs;
  ~DevPool() { for (void* p : ptrs) (void)hipFree(p); }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return p;
  }
  void* up(const void* src, size_t bytes) {
    void* p = alloc(bytes);
    if (p && src && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
  }
};

#define T_HIP(expr)                                                                                   \
  do {                                                                                                \
    hipError_t _h = (expr);                                                                           \
    if (_h != hipSuccess) {                                                                           \
      snprintf(TEST_ERR, 512, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,           \
               hipGetErrorString(_h));                                                                \
      return CZC_ERR_HIP;                                                                             \
    }                                                                                                 \
  } while (0)
#define T_CHECK(expr) do { int _r = (expr); if (_r) return _r; } while (0)
#define T_PTR(p) do { if (!(p)) { snprintf(TEST_ERR, 512, "device allocation/copy failed"); return CZC_ERR_HIP; } } while (0)

// fp32 host -> device buffer in precision `prec`
void* up_act(DevPool& pool, int prec, const float* src, size_t n) {
  float* f = (float*)pool.up(src, n * 4);
  if (!f) return nullptr;
  if (prec == PREC_F32) return f;
  void* a = pool.alloc(n * prec_bytes(prec));
  if (!a) return nullptr;
  if (launch_convert(prec, f, a, (long)n, nullptr)) return nullptr;
  return a;
}

int down_act(DevPool& pool, int prec, const void* src, size_t n, float* host) {
  const float* f = (const float*)src;
  if (prec != PREC_F32) {
    float* t = (float*)pool.alloc(n * 4);
    T_PTR(t);
    T_CHECK(launch_act_to_f32(prec, src, t, (long)n, nullptr));
    f = t;
  }
  T_HIP(hipMemcpy(host, f, n * 4, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace

extern "C" {

int czc_test_gemm(int precision, int M, int N, int K, const float* A, const float* W, const float* bias,
                  const float* resid, int act, float* C) {
  DevPool pool;
  void* dA = up_act(pool, precision, A, (size_t)M * K); T_PTR(dA);
  void* dW = up_act(pool, precision, W, (size_t)N * K); T_PTR(dW);
  float* dB = bias ? (float*)pool.up(bias, (size_t)N * 4) : nullptr;
  float* dR = resid ? (float*)pool.up(resid, (size_t)M * N * 4) : nullptr;
  const bool typed_out = (act & 0x100) != 0;  // store through the activation-typed epilogue (bf16 / split / f32)
  act &= 0xff;
  GemmArgs g;
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.bias = dB; g.resid = dR; g.ldr = N;
  g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act;
  if (typed_out) {
    void* dO = pool.alloc((size_t)M * N * prec_bytes(precision)); T_PTR(dO);
    g.out_act = dO; g.out_f32 = nullptr;
    T_CHECK(launch_gemm(precision, g, nullptr));
    T_HIP(hipDeviceSynchronize());
    return down_act(pool, precision, dO, (size_t)M * N, C);
  }
  float* dC = (float*)pool.alloc((size_t)M * N * 4); T_PTR(dC);
  g.out_act = nullptr; g.out_f32 = dC;
  T_CHECK(launch_gemm(precision, g, nullptr));
  T_HIP(hipDeviceSynchronize());
  T_HIP(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  return 0;
}

// The residual-add layers on a 2-byte residual stream (GemmArgs::x16): x_out = fp16(fp16(resid) + A.W^T + bias), in place on the
// fp16 rows as the engine runs it; which kernel serves it follows the shape and the czc_test_set_option switches (weight-stationary
// residual kernel at K = 512 / N % 256 == 0, ping-pong ring kernel from gemm256_min_m rows, tiled kernel otherwise).
int czc_test_gemm_x16(int precision, int M, int N, int K, const float* A, const float* W, const float* bias, const float* resid,
                      float* x_out, float* part_out) {
  if (precision != PREC_BF16 && precision != PREC_F16) { snprintf(TEST_ERR, 512, "gemm_x16: bf16 / fp16 operands only"); return CZC_ERR_ARG; }
  DevPool pool;
  void* dA = up_act(pool, precision, A, (size_t)M * K); T_PTR(dA);
  void* dW = up_act(pool, precision, W, (size_t)N * K); T_PTR(dW);
  float* dB = bias ? (float*)pool.up(bias, (size_t)N * 4) : nullptr;
  void* dx = up_act(pool, PREC_F16, resid, (size_t)M * N); T_PTR(dx);
  GemmArgs g;
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.bias = dB; g.resid = (const float*)dx; g.ldr = N; g.out_act = nullptr;
  g.out_f32 = (float*)dx; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = ACT_NONE; g.x16 = 1;
  float* dpart = nullptr;
  if (part_out) {  // LayerNorm partials [N / 32][M] float2 of the rows written
    dpart = (float*)pool.alloc((size_t)(N / 32) * M * 8); T_PTR(dpart);
    T_HIP(hipMemset(dpart, 0xff, (size_t)(N / 32) * M * 8));
    g.row_part = dpart; g.part_ld = M;
  }
  T_CHECK(launch_gemm(precision, g, nullptr));
  T_HIP(hipDeviceSynchronize());
  if (part_out) T_HIP(hipMemcpy(part_out, dpart, (size_t)(N / 32) * M * 8, hipMemcpyDeviceToHost));
  return down_act(pool, PREC_F16, dx, (size_t)M * N, x_out);
}

// LayerNorm folded into the weight-stationary K = 512 GEMM: out[M,N] = act( LN(fp16(x); gamma, beta, eps) . W^T + bias ) in the
// operand type of `precision`, computed as rstd * (x . W"^T) + b' from x itself: centred weights prepared by fold_ln_kernel
// (rowsum_out, optional [N]: what is left of each stored row's sum), statistics by ln_finalize_kernel from the partials `part`
// [16][M] float2 the caller supplies (a producer GEMM would have written them).
int czc_test_ln_fold_gemm(int precision, int M, int N, const float* x, const float* W, const float* gamma, const float* beta,
                          const float* bias, const float* part, float eps, int act, float* out, float* rowsum_out) {
  const int K = 512;
  if (precision != PREC_BF16 && precision != PREC_F16) { snprintf(TEST_ERR, 512, "ln_fold_gemm: bf16 / fp16 engines only"); return CZC_ERR_ARG; }
  DevPool pool;
  void* dx = up_act(pool, PREC_F16, x, (size_t)M * K); T_PTR(dx);
  float* dW = (float*)pool.up(W, (size_t)N * K * 4); T_PTR(dW);
  float* dg = (float*)pool.up(gamma, K * 4); T_PTR(dg);
  float* dbt = (float*)pool.up(beta, K * 4); T_PTR(dbt);
  float* db = bias ? (float*)pool.up(bias, (size_t)N * 4) : nullptr;
  float* dpart = (float*)pool.up(part, (size_t)16 * M * 8); T_PTR(dpart);
  float* dstat = (float*)pool.alloc((size_t)M * 8 + 256); T_PTR(dstat);
  void* dWf = pool.alloc((size_t)N * K * 2); T_PTR(dWf);
  float* dcs = (float*)pool.alloc((size_t)N * 4); T_PTR(dcs);
  float* dbf = (float*)pool.alloc((size_t)N * 4); T_PTR(dbf);
  void* dout = pool.alloc((size_t)M * N * 2); T_PTR(dout);
  const bool in_kernel = gemm_wreg_stats_in_kernel(M, N);  // small launches: the consumer sums the partials itself (option wreg_stats_in_kernel)
  if (!in_kernel) T_CHECK(launch_ln_finalize(dpart, M, 16, M, eps, dstat, nullptr));
  T_CHECK(launch_fold_ln(dW, dg, dbt, db, N, K, dWf, dcs, dbf, nullptr));
  GemmArgs g;
  g.A = dx; g.lda = K; g.W = dWf; g.ldw = K; g.bias = dbf; g.resid = nullptr; g.ldr = 0; g.out_act = dout; g.out_f32 = nullptr; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.act = act;
  g.ln_stat = dstat;
  if (in_kernel) { g.ln_part = dpart; g.ln_part_ld = M; g.ln_eps = eps; }
  T_CHECK(launch_gemm(precision, g, nullptr));
  T_HIP(hipDeviceSynchronize());
  if (rowsum_out) T_HIP(hipMemcpy(rowsum_out, dcs, (size_t)N * 4, hipMemcpyDeviceToHost));
  return down_act(pool, precision, dout, (size_t)M * N, out);
}

// LayerNorm of fp16 rows (512 wide) into the operand type: y = LN(fp16(x))
int czc_test_layernorm_x16(int precision, int M, const float* x, const float* gamma, const float* beta, float eps, float* y) {
  DevPool pool;
  void* dx = up_act(pool, PREC_F16, x, (size_t)M * 512); T_PTR(dx);
  float* dg = (float*)pool.up(gamma, 512 * 4); T_PTR(dg);
  float* db = (float*)pool.up(beta, 512 * 4); T_PTR(db);
  void* dy = pool.alloc((size_t)M * 512 * 2); T_PTR(dy);
  T_CHECK(launch_layernorm_x16(precision, dx, nullptr, dg, db, eps, M, 512, dy, nullptr));
  T_HIP(hipDeviceSynchronize());
  return down_act(pool, precision, dy, (size_t)M * 512, y);
}

// Microbenchmark of the GEMM kernels on device-resident random data (tools/bench_gemm.py).
// out_mode 0: bf16/act output (+bias, act); 1: fp32 output with in-place fp32 residual (+bias).
int czc_bench_gemm(int precision, int M, int N, int K, int act, int out_mode, int iters, int use256, double* ms_out) {
  DevPool pool;
  const size_t es = prec_bytes(precision);
  std::vector<float> ha((size_t)1 << 20), hw((size_t)N * K), hb(N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : ha) v = rnd();
  for (auto& v : hw) v = rnd() * 0.05f;
  for (auto& v : hb) v = rnd();
  const int pad = g_bench_pad;
  const size_t Kp = (size_t)K + pad;
  void* dA = pool.alloc((size_t)M * Kp * es); T_PTR(dA);
  float* tmp = (float*)pool.up(ha.data(), ha.size() * 4); T_PTR(tmp);
  for (size_t off = 0; off < (size_t)M * Kp; off += ha.size()) {
    const size_t n = std::min(ha.size(), (size_t)M * Kp - off);
    T_CHECK(launch_convert(precision, tmp, (char*)dA + off * es, (long)n, nullptr));
  }
  void* dW = nullptr;
  if (pad) {
    std::vector<float> hwp((size_t)N * Kp);
    for (auto& v : hwp) v = rnd() * 0.05f;
    dW = up_act(pool, precision, hwp.data(), hwp.size());
  } else {
    dW = up_act(pool, precision, hw.data(), hw.size());
  }
  T_PTR(dW);
  float* dB = (float*)pool.up(hb.data(), hb.size() * 4); T_PTR(dB);
  void* dOa = nullptr; float* dOf = nullptr;
  // out_mode: 0 activation-typed output; 1 fp32 output + fp32 residual (in place); 4 full-row kernel with the LayerNorm in
  // its epilogue; 5 the GEMM + LayerNorm pair it replaces
  if (out_mode == 0) { dOa = pool.alloc((size_t)M * N * es); T_PTR(dOa); }
  else { dOf = (float*)pool.alloc((size_t)M * N * 4); T_PTR(dOf); T_HIP(hipMemset(dOf, 0, (size_t)M * N * 4)); }
  GemmArgs g;
  g.A = dA; g.lda = (int)Kp; g.W = dW; g.ldw = (int)Kp; g.bias = dB; g.resid = dOf; g.ldr = N; g.out_act = dOa; g.out_f32 = dOf;
  g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act;
  if (out_mode == 6) { g.x16 = 1; }  // 6: fp16 residual stream in place (the fp32 buffer doubles as M x N fp16 rows)
  if (out_mode == 7) {  // 7: folded-LayerNorm consumer (K = 512): the operands' bytes read as fp16, unit statistics
    float* dstat = (float*)pool.alloc((size_t)M * 8 + 256); T_PTR(dstat);
    std::vector<float> hs((size_t)M * 2);
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 0.1f; hs[i + 1] = 1.0f; }
    T_HIP(hipMemcpy(dstat, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    g.ln_stat = dstat;
    g.out_act = pool.alloc((size_t)M * N * es); T_PTR(g.out_act);
    g.out_f32 = nullptr; g.resid = nullptr;
  }
  if (out_mode == 4 || out_mode == 5) {  // 4: full-row kernel with the LayerNorm in its epilogue; 5: the pair it replaces
    g.out_act = pool.alloc((size_t)M * N * 2); T_PTR(g.out_act);
    g.ln_gamma = dB; g.ln_beta = dB; g.ln_eps = 1e-5f; g.f16 = precision == PREC_F16;
  }
  const int saved = g_use_gemm256, saved_wreg = g_use_wreg;
  // use256: 0 128x128 kernel; 1 default choice of the ring kernels; 3 loader-wave ring kernel; 7 ping-pong ring kernel;
  // 6 weight-stationary kernel where eligible (else the default ring kernel)
  g_use_gemm256 = use256 == 7 ? 5 : use256 == 6 ? 1 : use256;
  g_use_wreg = use256 == 6 ? 1 : 0;
  hipEvent_t e0, e1;
  T_HIP(hipEventCreate(&e0)); T_HIP(hipEventCreate(&e1));
  auto run = [&]() -> int {
    if (out_mode == 4) return launch_gemm_rowln(g, nullptr);
    if (out_mode == 5) {
      GemmArgs p = g;
      p.out_act = nullptr; p.ln_gamma = p.ln_beta = nullptr;
      T_CHECK(launch_gemm(precision, p, nullptr));
      return launch_layernorm(precision, g.out_f32, nullptr, g.ln_gamma, g.ln_beta, g.ln_eps, M, N, g.out_act, nullptr, nullptr);
    }
    return launch_gemm(precision, g, nullptr);
  };
  for (int i = 0; i < 2; ++i) T_CHECK(run());
  T_HIP(hipDeviceSynchronize());
  T_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) T_CHECK(run());
  T_HIP(hipEventRecord(e1, nullptr));
  T_HIP(hipEventSynchronize(e1));
  float ms = 0;
  T_HIP(hipEventElapsedTime(&ms, e0, e1));
  g_use_gemm256 = saved;
  g_use_wreg = saved_wreg;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms_out = ms / iters;
  return 0;
}

// Full-row GEMM with the LayerNorm in its epilogue (gemm_rowln_kernel; bf16 or fp16 operands, N = 512):
//   x_out = resid + A[M,K].W[512,K]^T + bias;  y_out = LayerNorm(x_out; gamma, beta, eps) rounded to the operand type
int czc_test_gemm_rowln(int precision, int M, int K, const float* A, const float* W, const float* bias, const float* resid,
                        const float* gamma, const float* beta, float eps, float* x_out, float* y_out) {
  const int H = 512;
  if (precision != PREC_BF16 && precision != PREC_F16) { snprintf(TEST_ERR, 512, "gemm_rowln: bf16 / fp16 only"); return CZC_ERR_ARG; }
  DevPool pool;
  void* dA = up_act(pool, precision, A, (size_t)M * K); T_PTR(dA);
  void* dW = up_act(pool, precision, W, (size_t)H * K); T_PTR(dW);
  float* dB = bias ? (float*)pool.up(bias, (size_t)H * 4) : nullptr;
  float* dx = (float*)pool.up(resid, (size_t)M * H * 4); T_PTR(dx);
  float* dg = (float*)pool.up(gamma, (size_t)H * 4); T_PTR(dg);
  float* dbt = (float*)pool.up(beta, (size_t)H * 4); T_PTR(dbt);
  void* dy = pool.alloc((size_t)M * H * 2); T_PTR(dy);
  GemmArgs g;
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.bias = dB; g.resid = dx; g.ldr = H; g.out_act = dy; g.out_f32 = dx;
  g.ldc = H; g.M = M; g.N = H; g.K = K; g.act = ACT_NONE; g.f16 = precision == PREC_F16;
  g.ln_gamma = dg; g.ln_beta = dbt; g.ln_eps = eps;
  const int saved = g_rowln_min_m;
  g_rowln_min_m = 1;
  const int rc = launch_gemm_rowln(g, nullptr);
  g_rowln_min_m = saved;
  T_CHECK(rc);
  T_HIP(hipDeviceSynchronize());
  T_HIP(hipMemcpy(x_out, dx, (size_t)M * H * 4, hipMemcpyDeviceToHost));
  return down_act(pool, precision, dy, (size_t)M * H, y_out);
}

int czc_test_set_option(const char* name, int value) {
  if (!strcmp(name, "gemm256")) { g_use_gemm256 = value; return 0; }
  if (!strcmp(name, "skinny")) { g_use_skinny = value; return 0; }
  if (!strcmp(name, "splitk")) { g_use_splitk = value; return 0; }
  if (!strcmp(name, "gemm_deep")) { g_gemm_deep = value; return 0; }
  if (!strcmp(name, "gemm_small_tiles")) { g_gemm_small_tiles = value; return 0; }
  if (!strcmp(name, "bridge_no_table")) { g_bridge_no_table = value; return 0; }
  if (!strcmp(name, "wreg")) { g_use_wreg = value; return 0; }
  if (!strcmp(name, "gemm256s")) { g_use_gemm256s = value; return 0; }
  if (!strcmp(name, "bench_pad")) { g_bench_pad = value; return 0; }
  if (!strcmp(name, "w_dbg")) { g_w_dbg = value; return 0; }
  if (!strcmp(name, "ln_lean")) { g_ln_lean = value; return 0; }
  if (!strcmp(name, "rowln_min_m")) { g_rowln_min_m = value; return 0; }
  if (!strcmp(name, "wreg_min_m")) { g_wreg_min_m = value; return 0; }
  if (!strcmp(name, "gemm256_min_m")) { g_gemm256_min_m = value; return 0; }
  if (!strcmp(name, "gemm256s_min_m")) { g_gemm256s_min_m = value; return 0; }
  if (!strcmp(name, "mfma_attention")) { g_use_mfma_attention = value; return 0; }
  if (!strcmp(name, "attention_image")) { g_use_attention_image = value; return 0; }
  if (!strcmp(name, "wreg_resid_min_m")) { g_wreg_resid_min_m = value; return 0; }
  if (!strcmp(name, "wreg_stats_in_kernel")) { g_wreg_stats_in_kernel = value; return 0; }
  snprintf(TEST_ERR, 512, "unknown option %s", name);
  return CZC_ERR_ARG;
}

int czc_test_layernorm(int precision, int M, int H, const float* x, const float* gamma, const float* beta, float eps,
                       float* y) {
  DevPool pool;
  float* dx = (float*)pool.up(x, (size_t)M * H * 4); T_PTR(dx);
  float* dg = (float*)pool.up(gamma, (size_t)H * 4); T_PTR(dg);
  float* db = (float*)pool.up(beta, (size_t)H * 4); T_PTR(db);
  void* dy = pool.alloc((size_t)M * H * 4); T_PTR(dy);
  T_CHECK(launch_layernorm(precision, dx, nullptr, dg, db, eps, M, H, dy, nullptr, nullptr));
  T_HIP(hipDeviceSynchronize());
  T_CHECK(down_act(pool, precision, dy, (size_t)M * H, y));
  return 0;
}

int czc_test_attention(int precision, int n_seq, const int32_t* seq_len, int heads, int causal, float scale,
                       const float* qkv, float* out) {
  DevPool pool;
  std::vector<int> off(n_seq + 1, 0);
  int mx = 0;
  for (int i = 0; i < n_seq; ++i) { off[i + 1] = off[i] + seq_len[i]; mx = seq_len[i] > mx ? seq_len[i] : mx; }
  const size_t M = off[n_seq];
  const int Hd = heads * 64;
  void* dq = up_act(pool, precision, qkv, M * 3 * Hd); T_PTR(dq);
  int* doff = (int*)pool.up(off.data(), (n_seq + 1) * 4); T_PTR(doff);
  int* dlen = (int*)pool.up(seq_len, n_seq * 4); T_PTR(dlen);
  void* dout = pool.alloc(M * Hd * 4); T_PTR(dout);
  SegTable tab{nullptr, nullptr, doff, dlen, n_seq, 0};
  T_CHECK(launch_attention(precision, dq, tab, mx, heads, causal, scale, dout, nullptr));
  T_HIP(hipDeviceSynchronize());
  T_CHECK(down_act(pool, precision, dout, M * Hd, out));
  return 0;
}

int czc_test_topk(int B, int V, int K, const float* logits, const float* mask, float temperature, int dot_id,
                  int dot_allowed, float* probs, int32_t* idxs, int32_t* cand) {
  DevPool pool;
  float* dl = (float*)pool.up(logits, (size_t)B * V * 4); T_PTR(dl);
  float* dm = (float*)pool.up(mask, (size_t)V * 4); T_PTR(dm);
  float* dp = (float*)pool.alloc((size_t)B * K * 4); T_PTR(dp);
  int* di = (int*)pool.alloc((size_t)B * K * 4); T_PTR(di);
  int* dc = (int*)pool.alloc((size_t)B * K * 4); T_PTR(dc);
  T_CHECK(launch_softmax_mask_topk(dl, B, V, K, dm, temperature, dot_id, dot_allowed, dp, di, dc, nullptr));
  T_HIP(hipDeviceSynchronize());
  T_HIP(hipMemcpy(probs, dp, (size_t)B * K * 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(idxs, di, (size_t)B * K * 4, hipMemcpyDeviceToHost));
  if (cand) T_HIP(hipMemcpy(cand, dc, (size_t)B * K * 4, hipMemcpyDeviceToHost));
  return 0;
}

int czc_test_bridge(const czc_bridge_tables* t, const czc_config* cfg, int n_rows, int T, const int32_t* rows,
                    int32_t* clip_ids, int32_t* clip_len) {
  (void)cfg;
  DevPool pool;
  BridgeDev bd;
  const size_t nbytes = t->piece_off[t->bert_vocab];
  bd.bert_vocab = t->bert_vocab;
  bd.piece_off = (const uint32_t*)pool.up(t->piece_off, (size_t)(t->bert_vocab + 1) * 4); T_PTR(bd.piece_off);
  bd.piece_bytes = (const uint8_t*)pool.up(t->piece_bytes, nbytes ? nbytes : 1); T_PTR(bd.piece_bytes);
  bd.piece_class = (const uint8_t*)pool.up(t->piece_class, nbytes ? nbytes : 1); T_PTR(bd.piece_class);
  bd.piece_flags = (const uint8_t*)pool.up(t->piece_flags, (size_t)t->bert_vocab); T_PTR(bd.piece_flags);
  bd.byte_sym = (const int*)pool.up(t->byte_sym, 256 * 4); T_PTR(bd.byte_sym);
  bd.byte_sym_eow = (const int*)pool.up(t->byte_sym_eow, 256 * 4); T_PTR(bd.byte_sym_eow);
  size_t cap = 1024;
  while (cap < (size_t)t->n_merges * 2 + 2) cap <<= 1;
  std::vector<unsigned long long> keys(cap, ~0ull), vals(cap, 0ull);
  for (int r = 0; r < t->n_merges; ++r) {
    const unsigned long long key = ((unsigned long long)(unsigned)t->merge_left[r] << 32) | (unsigned)t->merge_right[r];
    unsigned h = bridge_hash(key) & (unsigned)(cap - 1);
    bool dup = false;
    while (keys[h] != ~0ull) {
      if (keys[h] == key) { dup = true; break; }
      h = (h + 1) & (unsigned)(cap - 1);
    }
    if (dup) continue;
    keys[h] = key;
    vals[h] = ((unsigned long long)(unsigned)r << 32) | (unsigned)t->merge_out[r];
  }
  bd.hkeys = (const unsigned long long*)pool.up(keys.data(), cap * 8); T_PTR(bd.hkeys);
  bd.hvals = (const unsigned long long*)pool.up(vals.data(), cap * 8); T_PTR(bd.hvals);
  bd.hmask = (unsigned)(cap - 1);
  bd.bos_id = t->bos_id;
  bd.eos_id = t->eos_id;
  if (!g_bridge_no_table) {  // the product's fast path: per-token ids tabulated on the device
    int* tok_ids = (int*)pool.alloc((size_t)t->bert_vocab * BR_TOKMAX * 4); T_PTR(tok_ids);
    uint8_t* tok_len = (uint8_t*)pool.alloc((size_t)t->bert_vocab); T_PTR(tok_len);
    T_CHECK(launch_bridge_precompute(bd, tok_ids, tok_len, nullptr));
    T_HIP(hipDeviceSynchronize());
    bd.tok_bpe = tok_ids; bd.tok_bpe_len = tok_len;
  }
  int* drows = (int*)pool.up(rows, (size_t)n_rows * T * 4); T_PTR(drows);
  int* dids = (int*)pool.alloc((size_t)n_rows * CZC_CLIP_MAX_LEN * 4); T_PTR(dids);
  int* dlen = (int*)pool.alloc((size_t)n_rows * 4); T_PTR(dlen);
  int* dovf = (int*)pool.alloc(4); T_PTR(dovf);
  T_HIP(hipMemset(dovf, 0, 4));
  PosDev nopos{nullptr, nullptr, 0};
  T_CHECK(launch_bridge(bd, drows, n_rows, T, -1, nullptr, 1, nullptr, nullptr, nullptr, 0, nopos, dids, dlen, nullptr, nullptr, dovf, nullptr));
  T_HIP(hipDeviceSynchronize());
  int ovf = 0;
  T_HIP(hipMemcpy(&ovf, dovf, 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(clip_ids, dids, (size_t)n_rows * CZC_CLIP_MAX_LEN * 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(clip_len, dlen, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
  if (ovf) { snprintf(TEST_ERR, 512, "bridge overflow on %d rows", ovf); return CZC_ERR_OVERFLOW; }
  return 0;
}

int czc_test_combine(int B, int K, int D, const float* text_feat, const float* img_embeds, float logit_scale,
                     const float* probs, const float* senti_raw, const float* repeats, const czc_hyper* hp,
                     float* clip_score, float* clip_ref, float* final_score, int32_t* best) {
  DevPool pool;
  const size_t bk = (size_t)B * K;
  float* dt = (float*)pool.up(text_feat, bk * D * 4); T_PTR(dt);
  float* di = (float*)pool.up(img_embeds, (size_t)B * D * 4); T_PTR(di);
  float* din = (float*)pool.alloc((size_t)B * D * 4); T_PTR(din);
  float* dp = (float*)pool.up(probs, bk * 4); T_PTR(dp);
  float* ds = senti_raw ? (float*)pool.up(senti_raw, bk * 4) : nullptr;
  float* dr = repeats ? (float*)pool.up(repeats, bk * 4) : nullptr;
  int* dcand = (int*)pool.alloc(bk * 4); T_PTR(dcand);
  T_HIP(hipMemset(dcand, 0, bk * 4));
  float* o1 = (float*)pool.alloc(bk * 4); float* o2 = (float*)pool.alloc(bk * 4); float* o3 = (float*)pool.alloc(bk * 4);
  int* ob = (int*)pool.alloc((size_t)B * 4); float* oc = (float*)pool.alloc((size_t)B * 4);
  T_PTR(o1); T_PTR(o2); T_PTR(o3); T_PTR(ob); T_PTR(oc);
  T_CHECK(launch_l2_normalize(di, B, D, din, nullptr));
  CombineArgs a;
  a.text_feat = dt; a.img_n = din; a.logit_scale_exp = expf(logit_scale); a.probs = dp; a.cand = dcand;
  a.senti_raw = ds; a.repeats = dr; a.alpha = hp->alpha; a.beta = hp->beta; a.gamma = hp->gamma;
  a.use_senti = (ds && (dr || hp->control == 2)) ? hp->control : 0; a.B = B; a.K = K; a.D = D; a.clip_score = o1; a.clip_ref = o2;
  a.final_score = o3; a.best = ob; a.best_cos = oc; a.inp = nullptr; a.T = 0; a.gen_idx = 0;
  T_CHECK(launch_combine(a, nullptr));
  T_HIP(hipDeviceSynchronize());
  T_HIP(hipMemcpy(clip_score, o1, bk * 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(clip_ref, o2, bk * 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(final_score, o3, bk * 4, hipMemcpyDeviceToHost));
  T_HIP(hipMemcpy(best, ob, (size_t)B * 4, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

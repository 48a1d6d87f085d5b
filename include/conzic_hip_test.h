/*
 * conzic_hip_test.h -- kernel-level parity hooks of the MI355X-native ConZIC engine (libconzic_hip_test.so).
 *
 * TEST INFRASTRUCTURE, not part of the drop-in boundary: each czc_test_* call runs ONE kernel family of
 * libconzic_hip.so on host data so that tests/ (-m gpu) can compare it with the CPU oracle, czc_bench_gemm is the GEMM
 * microbenchmark behind tools/ab_gemm.py / tools/bench_gemm.py, and czc_test_set_option flips process-wide kernel-family
 * switches for A/B runs.  The product library (include/conzic_hip.h) exports none of these; this library reaches its
 * launchers and switches through czc_internal_hooks, links against nothing but its C ABI, and is loaded next to it by tests and tools only (conzic_amd/native.py: load_test()).
 */
#ifndef CONZIC_HIP_TEST_H
#define CONZIC_HIP_TEST_H

#include "conzic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* C[M,N] = A[M,K] * W[N,K]^T (+bias) (+activation: 0 none, 1 quick_gelu, 2 gelu_erf) (+resid[M,N]).
 * act | 0x100: take the result through the activation-typed output path (bf16 / split-fp16 / f32 per
 * `precision`, resid must be NULL) instead of the fp32 one -- the path the tower-internal layers use. */
int czc_test_gemm(int precision, int M, int N, int K, const float* A, const float* W, const float* bias,
                  const float* resid, int act, float* C);
/* Residual-add layer on a 2-byte residual stream (the bf16 engine's CLIP-text tower, csrc/kernels.h GemmArgs::x16):
 * x_out[M,N] = fp16(fp16(resid) + A[M,K] * W[N,K]^T + bias), the sum formed in fp32; returned as fp32.  N, K % 8 == 0. */
int czc_test_gemm_x16(int precision, int M, int N, int K, const float* A, const float* W, const float* bias, const float* resid,
                      float* x_out, float* part_out /* NULL or [N/32][M][2]: (sum, sum of squares) per row and 32-column block */);
/* LayerNorm folded into the K = 512 GEMM: out[M,N] = act(LN(fp16(x); gamma, beta, eps) . W^T + bias), from x itself, the folded
 * (gain applied, rows centred) fp16 weights and the row statistics derived from `part` [16][M][2] (partials of fp16(x) as a
 * producer GEMM writes them).  rowsum_out: NULL or [N], what is left of the sum of each stored weight row. */
int czc_test_ln_fold_gemm(int precision, int M, int N, const float* x, const float* W, const float* gamma, const float* beta,
                          const float* bias, const float* part, float eps, int act, float* out, float* rowsum_out);
/* LayerNorm of fp16 rows, 512 wide: y[M,512] = LN(fp16(x)) rounded to the operand type of `precision` (bf16 / fp16). */
int czc_test_layernorm_x16(int precision, int M, const float* x, const float* gamma, const float* beta, float eps, float* y);

/* Full-row GEMM with the following LayerNorm in its epilogue (bf16 / fp16 operands, 512 columns, K % 32 == 0):
 *   x_out[M,512] = resid + A[M,K] * W[512,K]^T + bias;  y_out = LayerNorm(x_out; gamma, beta, eps) in the operand type */
int czc_test_gemm_rowln(int precision, int M, int K, const float* A, const float* W, const float* bias, const float* resid,
                        const float* gamma, const float* beta, float eps, float* x_out, float* y_out);
/* GEMM microbenchmark on device-resident data: ms per launch (tools/bench_gemm.py); use256: 0 128x128 kernel,
 * 1 the default choice among the 256x256 LDS-DMA ring kernels, 3 the loader-wave ring kernel, 7 the ping-pong ring
 * kernel, 6 the weight-stationary kernel where eligible. */
int czc_bench_gemm(int precision, int M, int N, int K, int act, int out_mode, int iters, int use256, double* ms_out);
/* Process-wide kernel-family switches for tests and tools: "gemm256" 0|1|3|5, "wreg" 0|1, "gemm256s" 0|1, "skinny",
 * "splitk", "gemm_deep" 0|1|2, "mfma_attention", "attention_image" 0|1|2 (2 = force), the "*_min_m" row-count thresholds, "w_dbg"
 * (ping-pong kernel A/B bits), "bench_pad" (row padding of czc_bench_gemm operands). */
int czc_test_set_option(const char* name, int value);
int czc_test_layernorm(int precision, int M, int H, const float* x, const float* gamma, const float* beta, float eps,
                       float* y);
/* qkv [sum(len), 3*heads*64] packed sequences; causal 0/1; scale; out [sum(len), heads*64] */
int czc_test_attention(int precision, int n_seq, const int32_t* seq_len, int heads, int causal, float scale,
                       const float* qkv, float* out);
int czc_test_topk(int B, int V, int K, const float* logits, const float* mask, float temperature, int dot_id,
                  int dot_allowed, float* probs, int32_t* idxs, int32_t* cand);
int czc_test_bridge(const czc_bridge_tables* t, const czc_config* cfg, int n_rows, int T, const int32_t* rows,
                    int32_t* clip_ids, int32_t* clip_len);
int czc_test_combine(int B, int K, int D, const float* text_feat, const float* img_embeds, float logit_scale,
                     const float* probs, const float* senti_raw, const float* repeats, const czc_hyper* hp,
                     float* clip_score, float* clip_ref, float* final_score, int32_t* best);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CONZIC_HIP_TEST_H */

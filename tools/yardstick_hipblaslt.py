#!/usr/bin/env python3
"""Vendor-library yardstick for the four CLIP-text linear-layer shapes (measurement only: nothing here is on the
product path).  torch.matmul on bf16 tensors dispatches to hipBLASLt / rocBLAS on ROCm; the numbers answer one
question: how fast does the vendor's tuned GEMM run C[M,N] = A[M,K] . W[N,K]^T (fp32 accumulate, bf16 out) at
M = 312 k rows on this chip, on uniform random data -- i.e. is ~1.4 PFLOP/s a ceiling of 256x256-tile GEMMs here
 or only of this repo's kernels.

    python tools/yardstick_hipblaslt.py [M]    ->  one JSON line per shape + a summary line
"""
import json
import os
import sys
import time

import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 312000
SHAPES = [("qkv", 1536, 512), ("out", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048), ("square4096", 4096, 4096)]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rows = []
for name, N, K in SHAPES:
    m = 4096 if name == "square4096" else M
    A = (torch.rand(m, K, device=dev) * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        C = A @ W.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * m * N * K / (ms * 1e-3) / 1e12
    byt = (m * K + N * K + m * N) * 2
    row = dict(shape=name, M=m, N=N, K=K, ms=round(ms, 4), tflops=round(tf, 1), frac_of_2500=round(tf / 2500, 3),
               min_hbm_gbps=round(byt / (ms * 1e-3) / 1e9, 1))
    rows.append(row)
    print(json.dumps(row), flush=True)
tot_ms = sum(r["ms"] for r in rows[:4])
tot_fl = sum(2.0 * r["M"] * r["N"] * r["K"] for r in rows[:4])
print(json.dumps(dict(summary="four layer shapes", ms=round(tot_ms, 3), tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                      torch=torch.__version__, hip=torch.version.hip,
                      prefer_hipblaslt=os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"))), flush=True)

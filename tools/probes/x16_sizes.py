#!/usr/bin/env python3
"""Debug: residual GEMM on fp16 rows at engine-like row counts, against fp64."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import kernel_hooks as KH
import torch
from conzic_amd import engine as E, native
def bf(a): return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()
def f16(a): return np.ascontiguousarray(a, np.float32).astype(np.float16).astype(np.float32)
rng = np.random.default_rng(1)
for K in (512, 2048):
    W = (rng.standard_normal((512, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    for M in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "12856,25712,28672,40000,64280,102848").split(",")]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        resid = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
        out = KH.gemm_x16(native.PREC_BF16, A, W, bias, resid)
        ref = bf(A).astype(np.float64) @ bf(W).astype(np.float64).T + bias + f16(resid)
        err = np.abs(out - ref)
        tol = np.maximum(np.abs(ref), 1.0) * 2.0 ** -10 + 2e-4 * np.sqrt(K / 64)
        bad = err > tol
        rows = np.unique(np.argwhere(bad)[:, 0])
        print(f"K={K} M={M}: max err {err.max():.3e} bad elements {int(bad.sum())} bad rows {len(rows)} first rows {rows[:12].tolist()} blocks {np.unique(rows // 32)[:12].tolist()}", flush=True)
sys.stdout.flush(); os._exit(0)

#!/usr/bin/env python3
"""GEMM microbenchmark over the CLIP-text layer shapes of BASELINE configs[2] (M = 256*200*15)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conzic_amd import native

lib = native.load_test()
if 'CZC_W_DBG' in os.environ:
    lib.czc_test_set_option(b'w_dbg', int(os.environ['CZC_W_DBG']))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 768000
shapes = [("qkv", 1536, 512, 0, 0), ("out", 512, 512, 0, 1), ("fc1", 2048, 512, 1, 0), ("fc2", 512, 2048, 0, 1)]
PREC = int(os.environ.get('CZC_GEMM_PREC', '0'))  # 0 bf16, 1 f32, 3 split-fp16 (use CZC_GEMM_VARIANTS=0)
VARIANTS = tuple(int(v) for v in os.environ.get('CZC_GEMM_VARIANTS', '3,7,6').split(','))  # czc_bench_gemm use256 codes
tot = {v: 0.0 for v in VARIANTS}
fl = 0.0
for name, N, K, act, mode in shapes:
    for use256 in VARIANTS:
        ms = C.c_double()
        native.check(lib.czc_bench_gemm(PREC, M, N, K, act, mode, 5, use256, C.byref(ms)), None, "bench")
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        tot[use256] += ms.value
        print(f"{name:4s} M={M} N={N:5d} K={K:5d} use256={use256}: {ms.value:8.3f} ms  {tf:7.1f} TF/s", flush=True)
    fl += 2.0 * M * N * K
for u in VARIANTS:
    print(f"layer GEMMs use256={u}: {tot[u]:.2f} ms -> {fl / (tot[u] * 1e-3) / 1e12:.1f} TF/s")
sys.stdout.flush()
if not os.environ.get('CZC_NORMAL_EXIT'):
    os._exit(0)  # skip interpreter/HIP teardown (it can hang on this image); rocprofv3 runs set CZC_NORMAL_EXIT=1

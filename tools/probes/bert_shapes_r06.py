"""Round 6: the BERT tower's four GEMM shapes (split-fp16, three MFMA passes per product) at the row counts of the headline run
(256 images x 15 rows = 3840 on one stream, 1920 per stream with two) under the launcher's switches: which tile form serves each
shape best.  usage: bert_shapes_r06.py [M ...]   ->   one line per (shape, M): median us per arm (8 alternated repeats of 5 launches)"""
import ctypes as C
import statistics
import sys

sys.path.insert(0, __file__.rsplit('/', 3)[0])
from conzic_amd import native  # noqa: E402

lib = native.load_test()
Ms = [int(v) for v in sys.argv[1:]] or [1920, 3840]
SHAPES = {"qkv": (2304, 768, 0, 0), "fc1": (3072, 768, 2, 0), "out": (768, 768, 0, 1), "fc2": (768, 3072, 0, 1)}
DEFAULTS = {"splitk": 1, "gemm_small_tiles": 4, "gemm_deep": 1, "gemm256s_min_m": 16384}
ARMS = {"product": {}, "ring256": {"gemm256s_min_m": 1024}, "deep": {"gemm_deep": 2}, "tiles64": {"gemm_small_tiles": 1024, "gemm_deep": 2},
        "nosplit": {"splitk": 0}}
for name, (N, K, act, out_mode) in SHAPES.items():
    for M in Ms:
        t = {a: [] for a in ARMS}
        for r in range(8):
            for a in (list(ARMS) if r % 2 == 0 else list(ARMS)[::-1]):
                for k, v in {**DEFAULTS, **ARMS[a]}.items():
                    lib.czc_test_set_option(k.encode(), v)
                ms = C.c_double()
                native.check(lib.czc_bench_gemm(3, M, N, K, act, out_mode, 5, 0, C.byref(ms)), None, "bench")
                t[a].append(ms.value * 1e3)
        fl = 2.0 * M * N * K * 3
        print(f"{name} N={N} K={K} M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f} us ({fl / statistics.median(v) / 1e6:.0f} TF/s-mfma)" for a, v in t.items()), flush=True)
for k, v in DEFAULTS.items():
    lib.czc_test_set_option(k.encode(), v)

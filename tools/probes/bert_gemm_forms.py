"""BERT's two fp32-residual layers (out-proj K = 768, fc2 K = 3072; N = 768; split-fp16) at 60-3840 rows: split-K (128-wide tiles,
2-4 K slices into fp32 slabs + splitk_reduce) against the unsplit tiled kernel with 64- or 128-wide tiles.
usage: bert_gemm_forms.py [M ...]"""
import ctypes as C
import statistics
import sys

sys.path.insert(0, __file__.rsplit('/', 3)[0])
from conzic_amd import native  # noqa: E402

lib = native.load_test()
Ms = [int(v) for v in sys.argv[1:]] or [60, 120, 240, 480, 960, 1920, 3840]
ARMS = {"splitk": (1, 4), "unsplit64": (0, 4), "unsplit128": (0, 0)}
for K in (768, 3072):
    for M in Ms:
        t = {a: [] for a in ARMS}
        for r in range(8):
            for a in (list(ARMS) if r % 2 == 0 else list(ARMS)[::-1]):
                sk, small = ARMS[a]
                lib.czc_test_set_option(b"splitk", sk)
                lib.czc_test_set_option(b"gemm_small_tiles", small)
                ms = C.c_double()
                native.check(lib.czc_bench_gemm(3, M, 768, K, 0, 1, 5, 0, C.byref(ms)), None, "bench")
                t[a].append(ms.value * 1e3)
        print(f"K={K} M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f} us" for a, v in t.items()), flush=True)
lib.czc_test_set_option(b"splitk", 1)
lib.czc_test_set_option(b"gemm_small_tiles", 4)

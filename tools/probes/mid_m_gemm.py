"""Which kernel should serve the fp32-residual layers (fc2: K = 2048, out-proj: K = 512; N = 512) at a few thousand packed rows
(4-32 images)?  The 256 x 256 ring kernel (use256 = 7) against the tiled kernel with 128- and 64-wide tiles, one step ahead
or three (test options gemm_deep, gemm_small_tiles).  Interleaved, medians.  usage: mid_m_gemm.py [M ...]"""
import ctypes as C
import statistics
import sys

sys.path.insert(0, __file__.rsplit('/', 3)[0])
from conzic_amd import native  # noqa: E402

lib = native.load_test()
Ms = [int(v) for v in sys.argv[1:]] or [2400, 4800, 9600, 19200, 38400, 76800]
ARMS = {"ring256": (7, 1, 4), "tile128": (0, 0, 0), "tile128_deep": (0, 2, 0), "tile64_deep": (0, 2, 1 << 20)}
for K in (2048, 512):
    for M in Ms:
        t = {a: [] for a in ARMS}
        for r in range(8):
            for a in (list(ARMS) if r % 2 == 0 else list(ARMS)[::-1]):
                use256, deep, small = ARMS[a]
                lib.czc_test_set_option(b"gemm_deep", deep)
                lib.czc_test_set_option(b"gemm_small_tiles", small)
                ms = C.c_double()
                native.check(lib.czc_bench_gemm(0, M, 512, K, 0, 1, 5, use256, C.byref(ms)), None, "bench")
                t[a].append(ms.value * 1e3)
        print(f"K={K} M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f} us" for a, v in t.items()), flush=True)
lib.czc_test_set_option(b"gemm_deep", 1)
lib.czc_test_set_option(b"gemm_small_tiles", 4)

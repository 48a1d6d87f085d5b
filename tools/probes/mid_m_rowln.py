"""out-proj + LN2 at a few thousand to a few hundred thousand packed rows: the full-row kernel (out_mode 4) against the GEMM +
LayerNorm pair (out_mode 5) with the GEMM on the ring kernel or on the tiled kernel (64-wide tiles on small grids).
usage: mid_m_rowln.py [M ...]"""
import ctypes as C
import statistics
import sys

sys.path.insert(0, __file__.rsplit('/', 3)[0])
from conzic_amd import native  # noqa: E402

lib = native.load_test()
Ms = [int(v) for v in sys.argv[1:]] or [4800, 9600, 19200, 38400, 76800, 156000]
ARMS = {"rowln": (4, 7, 2048), "ring256+LN": (5, 7, 2048), "tiled+LN": (5, 0, 1 << 30)}
for M in Ms:
    t = {a: [] for a in ARMS}
    for r in range(8):
        for a in (list(ARMS) if r % 2 == 0 else list(ARMS)[::-1]):
            mode, use256, min_m = ARMS[a]
            lib.czc_test_set_option(b"gemm256_min_m", min_m)
            ms = C.c_double()
            native.check(lib.czc_bench_gemm(0, M, 512, 512, 0, mode, 5, use256, C.byref(ms)), None, "bench")
            t[a].append(ms.value * 1e3)
    print(f"K=512 M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f} us" for a, v in t.items()), flush=True)
lib.czc_test_set_option(b"gemm256_min_m", 8192)

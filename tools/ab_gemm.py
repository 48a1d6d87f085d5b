"""Interleaved A/B of GEMM kernel variants on one shape (run-to-run noise on a box is several per cent: alternate the
arms and compare medians).  usage: ab_gemm.py M N K act out_mode arm[,arm...] [rounds]; arm = use256[:w_dbg[:out_mode[:row_pad]]]
(out_mode 4 = full-row kernel with the LayerNorm in its epilogue, 5 = the GEMM + LayerNorm pair it replaces)"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from conzic_amd import native  # noqa: E402

lib = native.load_test()
M, N, K, act, mode = (int(v) for v in sys.argv[1:6])
def _arm(text):
    f = [int(x) for x in text.split(':')]
    return (f[0], f[1] if len(f) > 1 else 0, f[2] if len(f) > 2 else mode, f[3] if len(f) > 3 else 0)


arms = [_arm(a) for a in sys.argv[6].split(',')]
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 12
ITERS = int(os.environ.get("AB_ITERS", "5"))  # launches per timing: 5 = a burst from a cool chip; hundreds = the sustained (power-limited) rate
t = {a: [] for a in arms}
for r in range(rounds):
    for a in (arms if r % 2 == 0 else arms[::-1]):
        ms = C.c_double()
        lib.czc_test_set_option(b'w_dbg', a[1])
        lib.czc_test_set_option(b'bench_pad', a[3])
        native.check(lib.czc_bench_gemm(0, M, N, K, act, a[2], ITERS, a[0], C.byref(ms)), None, "bench")
        t[a].append(ms.value)
for a in arms:
    med = statistics.median(t[a])
    print(f"M={M} N={N} K={K} use256={a[0]} w_dbg={a[1]} out_mode={a[2]} pad={a[3]}: median {med:.4f} ms  min {min(t[a]):.4f}  max {max(t[a]):.4f}  "
          f"{2.0 * M * N * K / (med * 1e-3) / 1e12:.1f} TF/s", flush=True)

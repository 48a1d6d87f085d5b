cd /root/repo
export TMPDIR=/tmp
timeout 300 python - 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from conzic_amd import native, engine as E
lib = native.load()
lib.czc_test_set_option(b"gemm256_min_m", 1)
M, N, K = 256, 256, 64
R = np.zeros((M, N), np.float32)
def run(A, W, g256):
    lib.czc_test_set_option(b"gemm256", g256)
    return E.test_gemm(0, A, W, bias=None, resid=R)
for rep in range(2):
  for kk in (0, 40):
    A = np.zeros((M, K), np.float32); A[:, kk] = 1.0
    W = np.zeros((N, K), np.float32); W[:, kk] = np.arange(N)
    c = run(A, W, 9)
    exp = np.tile(np.arange(N, dtype=np.float32), (M, 1))
    bad = np.argwhere(c != exp)
    rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
    print("W probe k", kk, "n bad", len(bad), "rows", rows[:40], "cols", cols[:70], flush=True)
    print("   sample", [(int(r), int(cc), float(c[r, cc])) for r, cc in bad[:12]], flush=True)
lib.czc_test_set_option(b"gemm256", 1); lib.czc_test_set_option(b"gemm256_min_m", 2048)
PY

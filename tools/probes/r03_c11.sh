set -x
mkdir -p gpurun_out/r03l
cd /root/repo
export TMPDIR=/tmp
timeout 300 python - > gpurun_out/r03l/check.log 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from conzic_amd import native, engine as E
lib = native.load()
rng = np.random.default_rng(1)
for (M, N, K) in ((2048 + 333, 512, 2048), (70000, 512, 512), (2500, 512, 128), (5000, 1024, 64), (300, 200, 192), (257, 512, 64)):
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32)
    outs = {}
    for name, g256, dbg in (("x", 5, 0), ("x_asm", 5, 8), ("k", 9, 0)):
        lib.czc_test_set_option(b"w_dbg", dbg); lib.czc_test_set_option(b"gemm256", g256); lib.czc_test_set_option(b"gemm256_min_m", 1)
        outs[name] = E.test_gemm(0, A, W, bias=b, resid=R)
        outs[name + "_nobias"] = E.test_gemm(0, A, W, bias=None, resid=R)
    lib.czc_test_set_option(b"w_dbg", 0); lib.czc_test_set_option(b"gemm256", 1); lib.czc_test_set_option(b"gemm256_min_m", 2048)
    ref = outs["x"]
    print(M, N, K, {k: (bool((v == outs["x" + ("_nobias" if k.endswith("_nobias") else "")]).all()), float(np.abs(v - outs["x" + ("_nobias" if k.endswith("_nobias") else "")]).max())) for k, v in outs.items()}, flush=True)
PY
cat gpurun_out/r03l/check.log
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:8,9 8 > gpurun_out/r03l/ab_fc2.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 7,7:8,9 8 > gpurun_out/r03l/ab_out.log 2>&1
timeout 300 python tools/ab_gemm.py 156000 512 2048 0 1 7,7:8,9 6 >> gpurun_out/r03l/ab_fc2.log 2>&1
cat gpurun_out/r03l/ab_*.log

set -x
mkdir -p gpurun_out/r03k
cd /root/repo
export TMPDIR=/tmp
timeout 300 python - > gpurun_out/r03k/dma_in_m_check.log 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from conzic_amd import native, engine as E
lib = native.load()
rng = np.random.default_rng(1)
for (M, N, K) in ((2048 + 333, 512, 2048), (70000, 512, 512), (2500, 512, 128), (5000, 1024, 64)):
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32); R = rng.standard_normal((M, N)).astype(np.float32)
    outs = []
    for v in (0, 4):
        lib.czc_test_set_option(b"w_dbg", v)
        lib.czc_test_set_option(b"gemm256", 5)
        outs.append(E.test_gemm(0, A, W, bias=b, resid=R))
    lib.czc_test_set_option(b"w_dbg", 0); lib.czc_test_set_option(b"gemm256", 1)
    print(M, N, K, "identical:", bool((outs[0] == outs[1]).all()), float(np.abs(outs[0] - outs[1]).max()))
PY
cat gpurun_out/r03k/dma_in_m_check.log
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:4,7:5 8 > gpurun_out/r03k/ab_fc2.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 7,7:4 8 > gpurun_out/r03k/ab_out.log 2>&1
timeout 300 python tools/ab_gemm.py 156000 512 2048 0 1 7,7:4 8 >> gpurun_out/r03k/ab_fc2.log 2>&1
cat gpurun_out/r03k/ab_*.log

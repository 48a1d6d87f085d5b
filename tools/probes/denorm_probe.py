import numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import kernel_hooks as KH
from conzic_amd import engine as E
rng = np.random.default_rng(0)
for sa, sw in ((2.0, 0.05), (0.01, 1.0), (1.0, 0.0005), (0.01, 0.01), (3e-4, 1.0)):
    A = (rng.standard_normal((300, 768)) * sa).astype(np.float32)
    W = (rng.standard_normal((256, 768)) * sw).astype(np.float32)
    C = KH.gemm(3, A, W)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    print(f"scale A={sa} W={sw}: max rel err {np.abs(C-ref).max()/np.abs(ref).max():.3e}")

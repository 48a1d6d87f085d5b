import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import kernel_hooks as KH
from conzic_amd import engine as E, native
rng = np.random.default_rng(1)
M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 9000, 2048
A = rng.standard_normal((M, K)).astype(np.float32)
W = (rng.standard_normal((512, K)) * 0.05).astype(np.float32)
resid = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
out, part = KH.gemm_x16(native.PREC_BF16, A, W, None, resid, want_part=True)
bad = ~np.isfinite(part[..., 0])
print("nan slots", int(bad.sum()), "of", bad.size)
idx = np.argwhere(bad)
print("blocks with nan:", np.unique(idx[:, 0]).tolist())
rows = np.unique(idx[:, 1])
print("rows with nan: count", len(rows), "first", rows[:40].tolist(), "rows%256:", np.unique(rows % 256)[:64].tolist())
blk = out.astype(np.float64).reshape(M, 16, 32)
ref = blk.sum(-1).T
good = ~bad
print("max err on written slots", float(np.abs(part[..., 0][good] - ref[good]).max()))
sys.stdout.flush(); os._exit(0)

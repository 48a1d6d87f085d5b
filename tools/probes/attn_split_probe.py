import numpy as np, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conzic_amd import engine as E, native
import kernel_hooks as KH
from test_kernels_gpu import _attn_ref
lib = native.load_test()
heads = 12
lens = [64]
for seed in range(6):
    rng = np.random.default_rng(seed)
    qkv = rng.standard_normal((sum(lens), 3 * heads * 64)).astype(np.float32)
    out = KH.attention(3, qkv, lens, heads, False, 0.125)
    lib.czc_test_set_option(b"mfma_attention", 0)
    out2 = KH.attention(3, qkv, lens, heads, False, 0.125)
    lib.czc_test_set_option(b"mfma_attention", 1)
    ref = _attn_ref(qkv, lens, heads, False, 0.125)
    err = np.abs(out - ref)
    r, c = np.unravel_index(err.argmax(), err.shape)
    h = c // 64
    Hd = heads * 64
    q = qkv[r, h*64:(h+1)*64]; k = qkv[:, Hd + h*64: Hd + (h+1)*64]
    s = (k @ q) * 0.125
    p = np.exp(s - s.max()); p /= p.sum()
    bad_rows = np.unique(np.argwhere(err > 1e-5)[:, 0])
    print(f"seed {seed}: max err {err.max():.2e} at row {r} col {c} (head {h}, d {c%64}); valu err {np.abs(out2-ref).max():.2e}; "
          f"n elems > 1e-5: {(err > 1e-5).sum()}; bad rows {bad_rows.tolist()}; s.max {s.max():.2f} p.max {p.max():.3f} p.min {p.min():.2e}")
    if err.max() > 1e-5:
        bad = np.argwhere(err > 1e-5)
        print("   bad cols (d within head):", sorted(set((bad[:, 1] % 64).tolist()))[:64], "heads", sorted(set((bad[:,1]//64).tolist())))
    x = ref[r, c]
    hi = np.float16(x)
    print(f"   ref {x:.9f} out {out[r,c]:.9f} valu {out2[r,c]:.9f} fp16(ref) {float(hi):.9f} ref-fp16 {x-float(hi):.3e} ulp {float(np.spacing(hi)):.3e}")

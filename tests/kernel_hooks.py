"""Kernel-level hooks for tests/ and tools/: typed numpy wrappers over libconzic_hip_test.so (include/conzic_hip_test.h).

TEST INFRASTRUCTURE.  Each function runs ONE kernel family of the product library on host data so that a `-m gpu` test can
compare it with the CPU oracle; nothing in conzic_amd/ imports this module and the product library exports none of the
symbols behind it.  (Until round 5 these lived in conzic_amd/engine.py as test_*; the names lost the prefix here so that
pytest does not collect them.)
"""
import ctypes as C

import numpy as np

from conzic_amd import native
from conzic_amd.bridge import BridgeArrays
from conzic_amd.engine import _ptr


def gemm(prec, A, W, bias=None, resid=None, act=0, typed_out=False):
    lib = native.load_test()
    if typed_out:
        act |= 0x100
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = A.shape
    N = W.shape[0]
    Cm = np.empty((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if resid is None else np.ascontiguousarray(resid, np.float32)
    native.check(lib.czc_test_gemm(prec, M, N, K, A.ctypes.data, W.ctypes.data, _ptr(b), _ptr(r), act, Cm.ctypes.data),
                 None, "czc_test_gemm")
    return Cm


def ln_fold_gemm(prec, x, W, gamma, beta, bias, part, eps, act=0, want_rowsum=False):
    """act(LN(fp16(x)) . W^T + bias) through the folded-LayerNorm weight-stationary GEMM; part [16, M, 2] = partials of fp16(x).
    want_rowsum: also the sums of the stored (centred, fp16) weight rows."""
    lib = native.load_test()
    x = np.ascontiguousarray(x, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, N = x.shape[0], W.shape[0]
    g = np.ascontiguousarray(gamma, np.float32)
    bt = np.ascontiguousarray(beta, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    pt = np.ascontiguousarray(part, np.float32)
    assert pt.shape == (16, M, 2) and x.shape[1] == 512 and W.shape[1] == 512
    out = np.empty((M, N), np.float32)
    rowsum = np.empty(N, np.float32)
    native.check(lib.czc_test_ln_fold_gemm(prec, M, N, x.ctypes.data, W.ctypes.data, g.ctypes.data, bt.ctypes.data, _ptr(b), pt.ctypes.data,
                                           float(eps), int(act), out.ctypes.data, rowsum.ctypes.data), None, "czc_test_ln_fold_gemm")
    return (out, rowsum) if want_rowsum else out


def gemm_x16(prec, A, W, bias, resid, want_part=False):
    """x = fp16(fp16(resid) + A.W^T + bias) on a 2-byte residual stream (GemmArgs::x16); returned as fp32
    (with want_part: also the LayerNorm partials [N/32, M, 2])."""
    lib = native.load_test()
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = A.shape
    N = W.shape[0]
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = np.ascontiguousarray(resid, np.float32)
    out = np.empty((M, N), np.float32)
    part = np.empty((N // 32, M, 2), np.float32) if want_part else None
    native.check(lib.czc_test_gemm_x16(prec, M, N, K, A.ctypes.data, W.ctypes.data, _ptr(b), r.ctypes.data, out.ctypes.data, _ptr(part)),
                 None, "czc_test_gemm_x16")
    return (out, part) if want_part else out


def layernorm_x16(prec, x, gamma, beta, eps):
    lib = native.load_test()
    x = np.ascontiguousarray(x, np.float32)
    M = x.shape[0]
    assert x.shape[1] == 512
    g = np.ascontiguousarray(gamma, np.float32)
    b = np.ascontiguousarray(beta, np.float32)
    y = np.empty_like(x)
    native.check(lib.czc_test_layernorm_x16(prec, M, x.ctypes.data, g.ctypes.data, b.ctypes.data, float(eps), y.ctypes.data),
                 None, "czc_test_layernorm_x16")
    return y


def gemm_rowln(prec, A, W, bias, resid, gamma, beta, eps):
    """x = resid + A.W^T + bias and y = LayerNorm(x) from the full-row kernel (W has 512 rows)."""
    lib = native.load_test()
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = A.shape
    assert W.shape == (512, K)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = np.ascontiguousarray(resid, np.float32)
    g = np.ascontiguousarray(gamma, np.float32)
    bt = np.ascontiguousarray(beta, np.float32)
    x = np.empty((M, 512), np.float32)
    y = np.empty((M, 512), np.float32)
    native.check(lib.czc_test_gemm_rowln(prec, M, K, A.ctypes.data, W.ctypes.data, _ptr(b), r.ctypes.data, g.ctypes.data,
                                         bt.ctypes.data, C.c_float(eps), x.ctypes.data, y.ctypes.data), None,
                 "czc_test_gemm_rowln")
    return x, y


def layernorm(prec, x, gamma, beta, eps):
    lib = native.load_test()
    x = np.ascontiguousarray(x, np.float32)
    g = np.ascontiguousarray(gamma, np.float32)
    b = np.ascontiguousarray(beta, np.float32)
    y = np.empty_like(x)
    native.check(lib.czc_test_layernorm(prec, x.shape[0], x.shape[1], x.ctypes.data, g.ctypes.data, b.ctypes.data,
                                        C.c_float(eps), y.ctypes.data), None, "czc_test_layernorm")
    return y


def attention(prec, qkv, seq_len, heads, causal, scale):
    lib = native.load_test()
    qkv = np.ascontiguousarray(qkv, np.float32)
    sl = np.ascontiguousarray(seq_len, np.int32)
    out = np.empty((qkv.shape[0], heads * 64), np.float32)
    native.check(lib.czc_test_attention(prec, sl.size, sl.ctypes.data, heads, 1 if causal else 0, C.c_float(scale),
                                        qkv.ctypes.data, out.ctypes.data), None, "czc_test_attention")
    return out


def topk(logits, mask, K, temperature, dot_id, dot_allowed):
    lib = native.load_test()
    lg = np.ascontiguousarray(logits, np.float32)
    mk = np.ascontiguousarray(np.asarray(mask, np.float32).reshape(-1))
    B, V = lg.shape
    p = np.empty((B, K), np.float32)
    i = np.empty((B, K), np.int32)
    c = np.empty((B, K), np.int32)
    native.check(lib.czc_test_topk(B, V, K, lg.ctypes.data, mk.ctypes.data, C.c_float(temperature), dot_id,
                                   1 if dot_allowed else 0, p.ctypes.data, i.ctypes.data, c.ctypes.data), None,
                 "czc_test_topk")
    return p, i, c


def bridge(tables: BridgeArrays, rows):
    lib = native.load_test()
    rows = np.ascontiguousarray(rows, np.int32)
    n, T = rows.shape
    ids = np.empty((n, native.CLIP_MAX_LEN), np.int32)
    ln = np.empty((n,), np.int32)
    st = tables.as_struct()
    native.check(lib.czc_test_bridge(C.byref(st), None, n, T, rows.ctypes.data, ids.ctypes.data, ln.ctypes.data), None,
                 "czc_test_bridge")
    return ids, ln


def combine(text_feat, img_embeds, logit_scale, probs, hyper, senti_raw=None, repeats=None):
    lib = native.load_test()
    tf = np.ascontiguousarray(text_feat, np.float32)
    ie = np.ascontiguousarray(img_embeds, np.float32)
    pr = np.ascontiguousarray(probs, np.float32)
    B, K = pr.shape
    D = ie.shape[1]
    sr = None if senti_raw is None else np.ascontiguousarray(senti_raw, np.float32)
    rp = None if repeats is None else np.ascontiguousarray(repeats, np.float32)
    cs, cr, fs = (np.empty((B, K), np.float32) for _ in range(3))
    best = np.empty((B,), np.int32)
    native.check(lib.czc_test_combine(B, K, D, tf.ctypes.data, ie.ctypes.data, C.c_float(logit_scale), pr.ctypes.data,
                                      _ptr(sr), _ptr(rp), C.byref(hyper), cs.ctypes.data, cr.ctypes.data,
                                      fs.ctypes.data, best.ctypes.data), None, "czc_test_combine")
    return cs, cr, fs, best

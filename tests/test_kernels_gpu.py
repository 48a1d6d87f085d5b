"""-m gpu: every HIP kernel, called through the C ABI (czc_test_*), against the CPU oracle /
a plain torch fp32 statement of the same op."""
import json
import os

import numpy as np
import pytest
import torch

from conzic_amd import engine as E
import kernel_hooks as KH
from conzic_amd import harness, native, synth
from conzic_amd.bridge import tables_from_tokenizers
from conzic_amd.text import tokenizers_from_vocab
from goldutil import GOLD

pytestmark = pytest.mark.gpu

BF16, F32 = native.PREC_BF16, native.PREC_F32
WREG_DEFAULT = 1  # czc_test_set_option("wreg"): weight-stationary GEMM on (0: K = 512 layers go to the tiled kernels)


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _act(v, act):
    t = torch.from_numpy(v)
    if act == 1:
        return (t * torch.sigmoid(1.702 * t)).numpy()
    if act == 2:
        return torch.nn.functional.gelu(t).numpy()
    return v


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 256, 512), (15, 768, 768), (300, 200, 192), (1, 30522, 128),
                                   (3000, 512, 2048), (129, 1536, 512), (77, 64, 3072)])
def test_gemm_asymmetric(prec, M, N, K):
    """Asymmetric random operands (catches operand/row-col transposes), ragged M/N edges."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[:, : K // 2] *= 3.0  # asymmetric along K too
    C = KH.gemm(prec, A, W)
    if prec == BF16:
        ref = _bf16_round(A).astype(np.float64) @ _bf16_round(W).astype(np.float64).T
        tol = 2e-3 * np.sqrt(K / 64)
    else:
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        tol = 2e-5 * np.sqrt(K / 64)
    err = np.abs(C - ref).max()
    assert err < tol, f"max err {err} (tol {tol})"


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_epilogues(prec, act):
    rng = np.random.default_rng(act + 10)
    M, N, K = 200, 320, 256
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    C = KH.gemm(prec, A, W, bias=bias, resid=resid, act=act)
    a, w = (A, W) if prec == F32 else (_bf16_round(A), _bf16_round(W))
    pre = (a.astype(np.float64) @ w.astype(np.float64).T + bias).astype(np.float32)
    ref = _act(pre, act) + resid
    assert np.abs(C - ref).max() < (3e-5 if prec == F32 else 2e-3)


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("H", [128, 512, 768])
def test_layernorm(prec, H):
    rng = np.random.default_rng(H)
    x = (rng.standard_normal((37, H)) * 3 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    b = (0.1 * rng.standard_normal(H)).astype(np.float32)
    for eps in (1e-5, 1e-12):
        y = KH.layernorm(prec, x, g, b, eps)
        ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (H,), torch.from_numpy(g), torch.from_numpy(b), eps).numpy()
        tol = 3e-6 * 10 if prec == F32 else 2e-2
        assert np.abs(y - ref).max() < tol


def _attn_ref(qkv, lens, heads, causal, scale):
    Hd = heads * 64
    out = np.zeros((qkv.shape[0], Hd), np.float32)
    o = 0
    for L in lens:
        blk = torch.from_numpy(qkv[o:o + L])
        q, k, v = blk[:, :Hd], blk[:, Hd:2 * Hd], blk[:, 2 * Hd:]
        q = q.view(L, heads, 64).transpose(0, 1)
        k = k.view(L, heads, 64).transpose(0, 1)
        v = v.view(L, heads, 64).transpose(0, 1)
        s = q @ k.transpose(-1, -2) * scale
        if causal:
            s = s + torch.full((L, L), float("-inf")).triu(1)
        out[o:o + L] = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, Hd).numpy()
        o += L
    return out


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_packed_sequences(prec, causal):
    rng = np.random.default_rng(3)
    heads = 8
    lens = [15, 1, 7, 16, 20, 77, 50, 64, 65, 2]
    M = sum(lens)
    qkv = rng.standard_normal((M, 3 * heads * 64)).astype(np.float32)
    if prec == BF16:
        qkv = _bf16_round(qkv)
    out = KH.attention(prec, qkv, lens, heads, causal, 0.125)
    ref = _attn_ref(qkv, lens, heads, causal, 0.125)
    assert np.abs(out - ref).max() < (2e-5 if prec == F32 else 1.5e-2)


@pytest.mark.parametrize("heads", [2, 12])
def test_attention_other_head_counts(heads):
    rng = np.random.default_rng(4)
    lens = [17, 17, 50]
    qkv = rng.standard_normal((sum(lens), 3 * heads * 64)).astype(np.float32)
    out = KH.attention(F32, qkv, lens, heads, False, 0.125)
    assert np.abs(out - _attn_ref(qkv, lens, heads, False, 0.125)).max() < 2e-5


@pytest.mark.parametrize("V,K", [(640, 12), (30522, 200), (30522, 512), (5000, 1024)])
def test_softmax_mask_topk(V, K):
    rng = np.random.default_rng(V + K)
    B = 5
    logits = (rng.standard_normal((B, V)) * 0.6).astype(np.float32)
    mask = (rng.random(V) > 0.1).astype(np.float32)
    dot_id = 17
    for dot_allowed in (False, True):
        p, i, c = KH.topk(logits, mask, K, 0.1, dot_id, dot_allowed)
        m = torch.from_numpy(mask.copy())
        m[dot_id] = 1.0 if dot_allowed else 0.0
        probs = torch.softmax(torch.from_numpy(logits) / 0.1, -1) * m
        rp, ri = probs.topk(K, dim=-1)
        np.testing.assert_allclose(p, rp.numpy(), rtol=2e-5, atol=1e-30)
        # ids must agree except where two neighbours are tied to within exp() rounding
        bad = np.argwhere(i != ri.numpy())
        for b_, k_ in bad:
            assert np.isclose(probs[b_, int(i[b_, k_])].item(), rp[b_, k_].item(), rtol=1e-5), (b_, k_)
        assert len(bad) <= 4
        mi = torch.from_numpy(i.astype(np.int64))
        np.testing.assert_array_equal(c, (mi * m[mi]).long().numpy())
        assert (np.diff(p, axis=1) <= 0).all()


@pytest.mark.parametrize("K", [200, 256, 300])
def test_topk_with_a_trained_models_logit_range(K):
    """A trained BERT's logits span 15-25 units, so `softmax(logits / 0.1)` underflows: beyond a gap of ~8.7 from the maximum
    the fp32 probabilities are DENORMAL, beyond ~10.3 exactly zero (the random-weight towers of the other tests never get
    there).  300 plausible tokens spread over a gap of 0..10.4 above a bulk that underflows: K = 200 ends in the normal
    range, 256 in the denormal range (kept, not flushed: torch keeps them and they decide the candidate SET), 300 beyond the
    non-zero count (zero-probability fill-ins from the lowest ids; `torch.topk`'s own choice among equal zeros is
    unspecified)."""
    V, B = 30522, 3
    rng = np.random.default_rng(K)
    logits = rng.standard_normal((B, V)).astype(np.float32)
    hot = np.stack([rng.choice(np.arange(1000, V), size=300, replace=False) for _ in range(B)])
    for b in range(B):
        logits[b, hot[b]] = np.linspace(16.0, 5.6, 300).astype(np.float32) + rng.uniform(-0.01, 0.01, 300).astype(np.float32)
    mask = np.ones(V, np.float32)
    mask[:999] = 0.0
    p, i, c = KH.topk(logits, mask, K, 0.1, 1012, False)
    m = torch.from_numpy(mask.copy())
    m[1012] = 0.0
    probs = (torch.softmax(torch.from_numpy(logits) / 0.1, -1) * m).numpy()
    for b in range(B):
        nz = int((probs[b] > 0).sum())
        assert 250 < nz < 300, nz                      # the construction: some of the 300 underflow to exactly zero
        order = np.argsort(-probs[b], kind="stable")
        n = min(K, nz)
        # non-zero part: same ids; values to fp32 rounding in the normal range, to a few denormal quanta below it
        ref_p = probs[b][order[:n]]
        got_p = p[b, :n]
        assert (got_p > 0).all() and (np.diff(p[b]) <= 0).all()
        normal = ref_p > 1.2e-38
        np.testing.assert_allclose(got_p[normal], ref_p[normal], rtol=3e-5)
        np.testing.assert_allclose(got_p[~normal], ref_p[~normal], rtol=0.02, atol=3e-45)
        assert set(i[b, :n].tolist()) == set(order[:n].tolist()) or \
            len(set(i[b, :n].tolist()) ^ set(order[:n].tolist())) <= 2      # a swap at the K-th place between near-equal denormals
        if K > nz:   # fill-ins: exact zeros, lowest ids first, all masked here -> [PAD]
            assert (p[b, nz:] == 0).all() and i[b, nz:].tolist() == list(range(K - nz)) and (c[b, nz:] == 0).all()


def test_topk_ties_and_all_masked():
    """Fewer than K non-zero probabilities: zeros are taken in ascending id, cand -> 0 ([PAD])."""
    V, K = 1000, 16
    logits = np.zeros((2, V), np.float32)
    logits[0, [5, 900, 33]] = [3.0, 2.0, 1.0]
    logits[1, :] = np.linspace(0, 1, V)
    mask = np.zeros(V, np.float32)
    mask[[5, 33, 900]] = 1
    p, i, c = KH.topk(logits, mask, K, 1.0, 0, False)
    assert i[0, :3].tolist() == [5, 900, 33]
    assert i[0, 3:].tolist() == [j for j in range(V) if j not in (5, 33, 900)][: K - 3]
    assert (p[0, 3:] == 0).all() and (c[0, 3:] == 0).all()
    assert i[1, :3].tolist() == [900, 33, 5]


@pytest.mark.parametrize("label", ["tiny", "full"])
def test_device_bridge_matches_hf_golden(label):
    g = json.load(open(os.path.join(GOLD, "text_bridge.json")))[label]
    sv = harness.cached_vocab(label == "tiny")
    bt, ct = tokenizers_from_vocab(sv)
    t = tables_from_tokenizers(bt, ct)
    by_len = {}
    for ids, c in zip(g["rows"], g["clip_ids"]):
        if len(ids) <= 64:
            by_len.setdefault(len(ids), []).append((ids, c))
    n = 0
    for T, items in by_len.items():
        rows = np.array([it[0] for it in items], np.int32)
        ids, ln = KH.bridge(t, rows)
        for r, (_, c) in enumerate(items):
            assert ln[r] == len(c)
            assert ids[r, : ln[r]].tolist() == c
            assert (ids[r, ln[r]:] == t.eos_id).all()
            n += 1
    assert n > 200


def test_device_bridge_overflow_fails_loudly():
    sv = harness.cached_vocab(True)
    bt, ct = tokenizers_from_vocab(sv)
    t = tables_from_tokenizers(bt, ct)
    longest = max(range(len(sv.bert_tokens)), key=lambda i: len(sv.bert_tokens[i].encode()))
    rows = np.full((1, 64), longest, np.int32)
    if 64 * (len(sv.bert_tokens[longest].encode()) + 1) > 512:
        with pytest.raises(native.NativeError, match="overflow"):
            KH.bridge(t, rows)


@pytest.mark.parametrize("senti", [False, True])
def test_fused_score_combine(senti):
    rng = np.random.default_rng(11)
    B, K, D = 3, 200, 512
    tf = rng.standard_normal((B * K, D)).astype(np.float32)
    ie = rng.standard_normal((B, D)).astype(np.float32)
    probs = np.sort(rng.random((B, K)).astype(np.float32), axis=1)[:, ::-1].copy()
    sraw = rng.standard_normal((B, K)).astype(np.float32)
    reps = rng.integers(0, 3, (B, K)).astype(np.float32)
    for ls in (2.6592, 4.6052):
        hp = E.Engine.hyper(0.02, 2.0, 0.1, gamma=5.0 if senti else None)
        cs, cr, fs, best = KH.combine(tf, ie, ls, probs, hp, sraw if senti else None, reps if senti else None)
        t = torch.from_numpy(tf).view(B, K, D)
        i = torch.from_numpy(ie)
        t = t / t.norm(dim=-1, keepdim=True)
        i = i / i.norm(dim=-1, keepdim=True)
        scale = torch.tensor(ls).exp()
        lg = torch.matmul(t, i.unsqueeze(-1)).squeeze(-1) * scale
        rcs, rcr = lg.softmax(1), lg / scale
        fin = 0.02 * torch.from_numpy(probs) + 2.0 * rcs
        if senti:
            fin = fin + 5.0 * torch.softmax(torch.from_numpy(sraw), 1) + 0.1 * (1 - torch.exp(torch.from_numpy(reps)))
        np.testing.assert_allclose(cr, rcr.numpy(), atol=2e-6)
        np.testing.assert_allclose(cs, rcs.numpy(), atol=2e-6, rtol=2e-4)
        np.testing.assert_allclose(fs, fin.numpy(), atol=1e-5, rtol=2e-4)
        np.testing.assert_array_equal(best, fin.argmax(1).numpy())


def test_combine_first_argmax_on_ties():
    B, K, D = 1, 8, 64
    tf = np.tile(np.arange(1, D + 1, dtype=np.float32), (K, 1))  # identical candidates
    ie = np.ones((B, D), np.float32)
    probs = np.zeros((B, K), np.float32)
    hp = E.Engine.hyper(0.02, 2.0, 0.1)
    _, _, fs, best = KH.combine(tf, ie, 2.0, probs, hp)
    assert best[0] == 0 and np.allclose(fs, fs[0, 0])


@pytest.mark.parametrize("variant", [1, 3, 5])
@pytest.mark.parametrize("M,N,K,act,resid", [(2048, 512, 512, 0, True), (3000, 1536, 512, 0, False),
                                             (5000, 2048, 512, 1, False), (2500, 512, 2048, 0, True),
                                             (2304, 320, 192, 1, True), (70000, 512, 512, 0, True)])
def test_gemm256_variants(variant, M, N, K, act, resid):
    """The 256x256 LDS-DMA ring kernels (3: loader-wave kernel, 5: ping-pong kernel, for every epilogue; 1: the
    product's choice between them): ragged M and N edges, many tiles per work-group, all epilogues."""
    lib = native.load_test()
    rng = np.random.default_rng(M + N + K + act)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if resid else None
    try:
        assert lib.czc_test_set_option(b"gemm256", variant) == 0
        C = KH.gemm(BF16, A, W, bias=bias, resid=R, act=act)
    finally:
        lib.czc_test_set_option(b"gemm256", 1)
    pre = (_bf16_round(A).astype(np.float64) @ _bf16_round(W).astype(np.float64).T + bias).astype(np.float32)
    ref = _act(pre, act) + (R if resid else 0)
    err = np.abs(C - ref).max()
    assert err < 3e-3 * np.sqrt(K / 64), err


@pytest.mark.parametrize("M,N,K", [(2500, 512, 2048), (70000, 512, 512), (2304, 320, 192), (12345, 512, 2048), (300, 512, 128),
                                   (700, 256, 64), (100, 40, 64), (80000, 512, 2048)])
def test_register_staged_four_wave_gemm_equals_the_ring_kernel_bitwise(M, N, K):
    """Round-4 A/B arm `gemm256 = 9` (gemm256r: four waves x 128 x 128 accumulators, operands staged through registers
    two stages ahead, ds_write into the ring, MFMAs interleaved one to one with the memory instructions) on the
    fp32-residual layers: same k order per accumulator and the same epilogue as the ping-pong ring kernel (gemm256 = 5), so
    the fp32 outputs are identical bit for bit -- ragged M / N edges, one to many tiles per work-group, two to 64 stages."""
    lib = native.load_test()
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    out = {}
    try:
        assert lib.czc_test_set_option(b"gemm256_min_m", 1) == 0
        for v in (5, 9):
            assert lib.czc_test_set_option(b"gemm256", v) == 0
            try:
                out[v] = KH.gemm(BF16, A, W, bias=bias, resid=R)
            except native.NativeError as exc:
                if v == 9 and "EXPERIMENTS=1 builds only" in str(exc):
                    pytest.skip("gemm256r is an A/B arm: built by `make EXPERIMENTS=1` only, not into the product library")
                raise
    finally:
        lib.czc_test_set_option(b"gemm256", 1)
        lib.czc_test_set_option(b"gemm256_min_m", 8192)
    ref = (_bf16_round(A).astype(np.float64) @ _bf16_round(W).astype(np.float64).T + bias).astype(np.float32) + R
    assert np.abs(out[5] - ref).max() < 3e-3 * np.sqrt(K / 64)
    np.testing.assert_array_equal(out[9], out[5])


@pytest.mark.parametrize("M,N,act", [(2048, 512, 0), (3000, 1536, 0), (5000, 2048, 1), (70001, 512, 1), (2304, 320, 0),
                                     (2049, 1536, 1), (40000, 2048, 0)])
def test_gemm_weight_stationary(M, N, act):
    """K = 512 bf16-output layers (CLIP-text qkv / fc1) take the weights-in-registers kernel: ragged M
    (partial last 32-row block), N not a multiple of the 256-column group, both activations; compared with
    the fp64 product of the bf16-rounded operands and with the tiled kernel on the same inputs."""
    lib = native.load_test()
    K = 512
    rng = np.random.default_rng(M + N + act)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[:, : K // 2] *= 3.0
    bias = rng.standard_normal(N).astype(np.float32)
    try:
        assert lib.czc_test_set_option(b"wreg", 1) == 0
        C = KH.gemm(BF16, A, W, bias=bias, act=act, typed_out=True)
        assert lib.czc_test_set_option(b"wreg", 0) == 0
        C2 = KH.gemm(BF16, A, W, bias=bias, act=act, typed_out=True)
    finally:
        lib.czc_test_set_option(b"wreg", WREG_DEFAULT)
    pre = (_bf16_round(A).astype(np.float64) @ _bf16_round(W).astype(np.float64).T + bias).astype(np.float32)
    ref = _act(pre, act)
    tol = 2e-3 * np.sqrt(K / 64) + np.abs(ref) * 2.0 ** -8  # + one bf16 rounding of the output
    assert (np.abs(C - ref) <= tol).all(), np.abs(C - ref).max()
    assert (np.abs(C - C2) <= np.abs(ref) * 2.0 ** -7 + 1e-3).all(), np.abs(C - C2).max()


F16X3 = 3  # internal precision code: split-fp16 storage, three fp16 MFMA passes (BERT tower of the bf16 engine)


@pytest.mark.parametrize("M,N,K,act", [(15, 768, 768, 0), (300, 2304, 768, 0), (256, 3072, 768, 2), (129, 768, 3072, 0),
                                       (7, 30522, 768, 0)])
def test_split_fp16_gemm_is_fp32_class(M, N, K, act):
    rng = np.random.default_rng(M + N)
    A = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C = KH.gemm(F16X3, A, W, bias=bias, act=act)
    ref = _act((A.astype(np.float64) @ W.astype(np.float64).T + bias).astype(np.float32), act)
    err = np.abs(C - ref).max()
    assert err < 1e-5 * np.sqrt(K / 64) * 4, err


@pytest.mark.parametrize("M", [1, 15, 16, 17, 30, 32])
@pytest.mark.parametrize("N,K,act,resid", [(768, 768, 0, True), (2304, 768, 0, False), (3072, 768, 2, False),
                                           (768, 3072, 0, True), (30522, 768, 0, False), (200, 64, 1, True)])
def test_split_fp16_skinny_gemm(M, N, K, act, resid):
    """M <= 32 takes the K-split weight-streaming kernel (BERT at batch 1-2); same fp32-class bound,
    and agreement with the tiled kernel on the same operands."""
    rng = np.random.default_rng(M * 31 + N + K)
    A = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[:, : K // 2] *= 3.0
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if resid else None
    C = KH.gemm(F16X3, A, W, bias=bias, resid=R, act=act)
    ref = _act((A.astype(np.float64) @ W.astype(np.float64).T + bias).astype(np.float32), act) + (R if resid else 0)
    tol = 1e-5 * np.sqrt(K / 64) * 4
    assert np.abs(C - ref).max() < tol
    lib = native.load_test()
    try:
        assert lib.czc_test_set_option(b"skinny", 0) == 0
        C2 = KH.gemm(F16X3, A, W, bias=bias, resid=R, act=act)
    finally:
        lib.czc_test_set_option(b"skinny", 1)
    assert np.abs(C - C2).max() < tol


@pytest.mark.parametrize("M,N,K,resid", [(3840, 768, 768, True), (3840, 768, 3072, True), (1000, 512, 1024, False),
                                         (300, 768, 3072, True)])
def test_split_fp16_gemm_split_k(M, N, K, resid):
    """Few-tile, long-K fp32-output layers (BERT out-proj / fc2 at B = 256) run split-K with a fixed-order
    reduction: fp32-class accuracy, bit-identical from run to run, and equal to the unsplit kernel up to
    fp32 summation order."""
    lib = native.load_test()
    rng = np.random.default_rng(M + N + K)
    A = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if resid else None
    C = KH.gemm(F16X3, A, W, bias=bias, resid=R)
    C_again = KH.gemm(F16X3, A, W, bias=bias, resid=R)
    np.testing.assert_array_equal(C, C_again)
    ref = (A.astype(np.float64) @ W.astype(np.float64).T + bias).astype(np.float32) + (R if resid else 0)
    tol = 1e-5 * np.sqrt(K / 64) * 4
    assert np.abs(C - ref).max() < tol
    try:
        assert lib.czc_test_set_option(b"splitk", 0) == 0
        C1 = KH.gemm(F16X3, A, W, bias=bias, resid=R)
    finally:
        lib.czc_test_set_option(b"splitk", 1)
    assert np.abs(C - C1).max() < tol


def test_split_fp16_layernorm_and_attention():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((33, 768)) * 2).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(768)).astype(np.float32)
    b = (0.1 * rng.standard_normal(768)).astype(np.float32)
    y = KH.layernorm(F16X3, x, g, b, 1e-12)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (768,), torch.from_numpy(g), torch.from_numpy(b), 1e-12).numpy()
    assert np.abs(y - ref).max() < 2e-5
    lens = [15, 17, 64]
    qkv = rng.standard_normal((sum(lens), 3 * 12 * 64)).astype(np.float32)
    out = KH.attention(F16X3, qkv, lens, 12, False, 0.125)
    assert np.abs(out - _attn_ref(qkv, lens, 12, False, 0.125)).max() < 3e-5


@pytest.mark.parametrize("prec", [0, 4])
@pytest.mark.parametrize("M,K,mean_shift", [(128 * 3 + 37, 512, 0.0), (5000, 2048, 0.0), (70000, 512, 3.0), (1000, 64, 0.0),
                                            (300, 96, 0.5), (41000, 2048, 0.0)])
def test_full_row_gemm_with_layernorm_epilogue(prec, M, K, mean_shift):
    """gemm_rowln_kernel: x = resid + A.W^T + b over full 512-wide rows and y = LayerNorm(x) from the same launch
    (out-proj -> LN2, fc2 -> next LN1 of the pre-LN block, HF:clip/modeling_clip.py:368-383).  x against fp64 on the
    rounded operands; y against the fp64 LayerNorm of the kernel's own x (isolates the statistics and the
    normalisation) at the output type's rounding.  Ragged M, one and several tiles per work-group, the short-K
    prologue paths (2 and 3 stages in total), a row mean far from zero."""
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((512, K)) * 0.03).astype(np.float32)
    b = (rng.standard_normal(512) * 0.1).astype(np.float32)
    resid = (rng.standard_normal((M, 512)) * 1.5 + mean_shift).astype(np.float32)
    gamma = (1.0 + 0.3 * rng.standard_normal(512)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(512)).astype(np.float32)
    x, y = KH.gemm_rowln(prec, A, W, b, resid, gamma, beta, 1e-5)
    dt = torch.bfloat16 if prec == 0 else torch.float16
    rd = lambda a: torch.from_numpy(a).to(dt).to(torch.float64).numpy()
    x_ref = resid.astype(np.float64) + rd(A) @ rd(W).T + b
    assert np.abs(x - x_ref).max() < 2e-4 * np.sqrt(max(K, 512) / 512) * (1.0 + abs(mean_shift))
    x64 = x.astype(np.float64)
    mu = x64.mean(1, keepdims=True)
    y_ref = (x64 - mu) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * gamma + beta
    ulp = 2.0 ** -8 if prec == 0 else 2.0 ** -11
    err = np.abs(y - y_ref)
    assert (err <= ulp * np.abs(y_ref) + 2e-5 * (1.0 + abs(mean_shift))).all(), err.max()
    # and the stand-alone LayerNorm kernel on the same x gives the same rounded rows (a few last-place flips at most)
    y_k = KH.layernorm(prec, x, gamma, beta, 1e-5) if prec == 0 else None
    if y_k is not None:
        assert (y_k != y).mean() < 2e-3 and np.abs(y_k - y).max() <= 2 * ulp * np.abs(y_ref).max()


@pytest.mark.parametrize("M,N,K,act,mode", [(16384 + 37, 1536, 512, 0, "typed"), (20000, 2048, 512, 1, "typed"),
                                            (16500, 512, 2048, 0, "resid"), (16384, 512, 512, 0, "resid")])
def test_split_fp16_gemm_256_tile_kernel(M, N, K, act, mode):
    """gemm256s: the persistent 256x256 LDS-DMA kernel with three fp16 MFMA passes per product (the CLIP-text linear
    layers of the split engine, M >= 16384).  fp32-class against an fp64 reference, ragged M, both epilogues
    (split_t activation output incl. quick-GELU, fp32 output + fp32 residual), and agreement with the 128x128
    register-staged kernel it replaces."""
    rng = np.random.default_rng(N + K)
    A = (rng.standard_normal((M, K)) * 1.5).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if mode == "resid" else None
    C = KH.gemm(F16X3, A, W, bias=bias, resid=R, act=act, typed_out=(mode == "typed"))
    ref = _act((A.astype(np.float64) @ W.astype(np.float64).T + bias).astype(np.float32), act) + (R if R is not None else 0)
    tol = 1e-5 * np.sqrt(K / 64) * 4 + (2e-6 * np.abs(ref).max() if mode == "typed" else 0)  # typed: hi+lo storage ~2^-22
    assert np.abs(C - ref).max() < tol, np.abs(C - ref).max()
    lib = native.load_test()
    for variant in (0,):  # 0: the 128x128 kernel (default 1: the four-stage ring kernel)
        try:
            assert lib.czc_test_set_option(b"gemm256s", variant) == 0
            C2 = KH.gemm(F16X3, A, W, bias=bias, resid=R, act=act, typed_out=(mode == "typed"))
        finally:
            lib.czc_test_set_option(b"gemm256s", 1)
        assert np.abs(C - C2).max() < tol, variant


def test_split_fp16_store_has_one_value_behind_hi_and_lo():
    """Regression: with -ffp-contract=fast the hi part fed to `v - hi` and the hi part stored could come from
    different roundings (v_fma_mixlo_f16 of the exact product vs v_cvt_pk_f16_f32 of the fp32 one) and hi + lo was
    one fp16 ulp off at near-ties, ~1 value in 8000 (common.h pin()).  49 k attention outputs, every one checked."""
    rng = np.random.default_rng(1)
    heads, lens = 12, [64]
    qkv = rng.standard_normal((64, 3 * heads * 64)).astype(np.float32)
    out = KH.attention(F16X3, qkv, lens, heads, False, 0.125)
    assert np.abs(out - _attn_ref(qkv, lens, heads, False, 0.125)).max() < 5e-6


@pytest.mark.parametrize("causal", [False, True])
def test_split_fp16_mfma_attention(causal):
    """attention_mfma_split_kernel (three fp16 MFMA passes for QK^T and PV) on packed ragged segments up to 77 rows,
    8 and 12 heads: fp32-class, and equal to the exact-fp32 VALU kernel it replaces within that class."""
    rng = np.random.default_rng(9)
    for heads in (8, 12):
        lens = [15, 1, 7, 16, 20, 77, 50, 64, 65, 2, 33]
        qkv = rng.standard_normal((sum(lens), 3 * heads * 64)).astype(np.float32)
        out = KH.attention(F16X3, qkv, lens, heads, causal, 0.125)
        ref = _attn_ref(qkv, lens, heads, causal, 0.125)
        assert np.abs(out - ref).max() < 3e-5, np.abs(out - ref).max()
        lib = native.load_test()
        try:
            assert lib.czc_test_set_option(b"mfma_attention", 0) == 0
            out2 = KH.attention(F16X3, qkv, lens, heads, causal, 0.125)
        finally:
            lib.czc_test_set_option(b"mfma_attention", 1)
        assert np.abs(out - out2).max() < 3e-5


FP16 = 4  # internal precision code: f16_t storage, one fp16 MFMA pass (the bf16 kernels with the fp16 opcode)


def _f16_round(a):
    return np.ascontiguousarray(a).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,N,K,act,mode", [(300, 200, 192, 0, "typed"), (3000, 512, 2048, 0, "resid"), (2100, 1536, 512, 0, "typed"),
                                            (40000, 2048, 512, 1, "typed"), (40000, 512, 512, 0, "resid"), (129, 64, 3072, 2, "f32")])
def test_fp16_gemm_kernel_families(M, N, K, act, mode):
    """fp16 operands through the 128x128 kernel, the weight-stationary kernel and the 256x256 ring kernel: against an
    fp64 reference on fp16-rounded operands (the kernels differ from bf16 only in the MFMA opcode and the converter)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if mode == "resid" else None
    C = KH.gemm(FP16, A, W, bias=bias, resid=R, act=act, typed_out=(mode == "typed"))
    pre = (_f16_round(A).astype(np.float64) @ _f16_round(W).astype(np.float64).T + bias).astype(np.float32)
    ref = _act(pre, act) + (R if R is not None else 0)
    tol = 3e-4 * np.sqrt(K / 64) + (2 ** -10 * np.abs(ref).max() if mode == "typed" else 0)
    assert np.abs(C - ref).max() < tol, np.abs(C - ref).max()


@pytest.mark.parametrize("causal", [True, False])
def test_fp16_attention(causal):
    rng = np.random.default_rng(13)
    heads = 8
    lens = [15, 1, 7, 16, 20, 77, 50, 64, 65, 2]
    qkv = _f16_round(rng.standard_normal((sum(lens), 3 * heads * 64)).astype(np.float32))
    out = KH.attention(FP16, qkv, lens, heads, causal, 0.125)
    assert np.abs(out - _attn_ref(qkv, lens, heads, causal, 0.125)).max() < 2e-3


# ---- one result per layer shape, whatever the row count -------------------------------------------------------------
# The engine picks GEMM / LayerNorm / attention kernels by how many packed rows a step has, and a bf16 near-tie that
# flips changes the rest of a caption, so kernels that can serve the same layer must agree BIT FOR BIT
# (tests/test_step_gpu.py::test_caption_does_not_depend_on_the_batch holds the engine to it end to end).

@pytest.mark.parametrize("prec", [0, 4])
@pytest.mark.parametrize("N,K", [(512, 512), (512, 2048)])
def test_tiled_and_ring_gemms_agree_bitwise_on_fp32_residual_layers(prec, N, K):
    """out-proj / fc2 / text projection: the 128x128 kernel (few rows) and the 256x256 ring kernels (>= 2048 rows) sum k
    in the same single ascending chain and apply bias and residual in the same order."""
    lib = native.load_test()
    rng = np.random.default_rng(N + K + prec)
    M = 2048 + 333
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    outs = []
    try:
        for variant in (0, 3, 5):
            assert lib.czc_test_set_option(b"gemm256", variant) == 0
            outs.append(KH.gemm(prec, A, W, bias=bias, resid=R))
    finally:
        lib.czc_test_set_option(b"gemm256", 1)
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])
    # and a row's result does not depend on how many rows ride with it
    np.testing.assert_array_equal(KH.gemm(prec, A[:100], W, bias=bias, resid=R[:100]), outs[0][:100])


@pytest.mark.parametrize("prec", [0, 4])
def test_full_row_kernel_and_gemm_plus_layernorm_agree_bitwise(prec):
    """out-proj + LN2: the full-row kernel with the LayerNorm in its epilogue (>= 4096 rows) against the ring / tiled
    GEMM followed by the stand-alone LayerNorm kernel, which for 512-wide half-precision rows restates the epilogue's
    arithmetic operation for operation (rowops.hip layernorm512_kernel)."""
    rng = np.random.default_rng(5 + prec)
    M, K = 4096 + 77, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((512, K)) * 0.03).astype(np.float32)
    b = (rng.standard_normal(512) * 0.1).astype(np.float32)
    resid = (rng.standard_normal((M, 512)) * 1.5 + 0.7).astype(np.float32)
    gamma = (1.0 + 0.3 * rng.standard_normal(512)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(512)).astype(np.float32)
    x, y = KH.gemm_rowln(prec, A, W, b, resid, gamma, beta, 1e-5)
    x2 = KH.gemm(prec, A, W, bias=b, resid=resid)
    np.testing.assert_array_equal(x, x2)
    y2 = KH.layernorm(prec, x2, gamma, beta, 1e-5)
    np.testing.assert_array_equal(y, y2)
    # the 30-VGPR LayerNorm kernel (ds_swizzle partners, default) against the ds_bpermute form: same tree, same bits
    lib = native.load_test()
    try:
        assert lib.czc_test_set_option(b"ln_lean", 0) == 0
        np.testing.assert_array_equal(KH.layernorm(prec, x2, gamma, beta, 1e-5), y2)
    finally:
        lib.czc_test_set_option(b"ln_lean", 1)
    # the asm-counted x phase of the epilogue (default) against the compiler-scheduled one it replaces, twice (a mis-counted
    # wait shows up as a few stale lanes on some run), with and without a bias, ragged last tile
    lib = native.load_test()
    for bias in (b, None):
        outs = []
        try:
            for dbg in (0, 8, 0):
                assert lib.czc_test_set_option(b"w_dbg", dbg) == 0
                outs.append(KH.gemm_rowln(prec, A, W, bias, resid, gamma, beta, 1e-5))
        finally:
            lib.czc_test_set_option(b"w_dbg", 0)
        for xo, yo in outs[1:]:
            np.testing.assert_array_equal(xo, outs[0][0])
            np.testing.assert_array_equal(yo, outs[0][1])
    np.testing.assert_array_equal(outs[0][0], KH.gemm(prec, A, W, resid=resid))


@pytest.mark.parametrize("act", [0, 1])
def test_weight_stationary_gemm_serves_every_row_count(act):
    """qkv / fc1 (K = 512, activation-typed output): the weights-in-registers kernel takes the layer at any row count,
    and a row's result does not depend on how many rows ride with it."""
    rng = np.random.default_rng(17 + act)
    M, N, K = 5000, 1536, 512
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C = KH.gemm(BF16, A, W, bias=bias, act=act, typed_out=True)
    for m in (1, 31, 100, 2047):
        np.testing.assert_array_equal(KH.gemm(BF16, A[:m], W, bias=bias, act=act, typed_out=True), C[:m])


def test_kernel_families_agree_bitwise_on_random_shapes():
    """Stress form of the three tests above: 24 random (M, N, K) with ragged edges, fp32-residual layer through the tiled
    kernel, the loader-wave ring kernel and the ping-pong ring kernel (asm-counted epilogue), twice each: bit-identical
    across kernels and across repetitions (a mis-counted wait or an unpadded hazard shows up as a few wrong lanes on some
    launches, not as rounding noise)."""
    lib = native.load_test()
    rng = np.random.default_rng(20260929)
    try:
        assert lib.czc_test_set_option(b"gemm256_min_m", 1) == 0
        for it in range(24):
            M = int(rng.integers(1, 3000)) if it % 3 else int(rng.integers(3000, 40000))
            N = int(rng.choice([512, 512, 1024, 320, 768, 256, 8 * int(rng.integers(1, 120))]))
            K = 64 * int(rng.integers(1, 33))
            A = rng.standard_normal((M, K)).astype(np.float32)
            W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32) if it % 2 else None
            R = rng.standard_normal((M, N)).astype(np.float32)
            ref = None
            for variant in (0, 3, 5, 5):
                assert lib.czc_test_set_option(b"gemm256", variant) == 0
                out = KH.gemm(BF16, A, W, bias=bias, resid=R)
                if ref is None:
                    ref = out
                else:
                    np.testing.assert_array_equal(out, ref, err_msg=f"variant {variant} M={M} N={N} K={K}")
    finally:
        lib.czc_test_set_option(b"gemm256", 1)
        lib.czc_test_set_option(b"gemm256_min_m", 8192)


@pytest.mark.parametrize("prec", [BF16, F32, F16X3])
def test_tiled_gemm_prefetch_depth_does_not_change_results(prec):
    """The 128 x 128 kernel requests its operands one K step ahead, or two for launches of at most one work-group per CU
    (option gemm_deep): same summation order, bit-identical outputs -- odd and even step counts, a single step, ragged
    edges, split-K slices."""
    lib = native.load_test()
    rng = np.random.default_rng(3 + prec)
    try:
        assert lib.czc_test_set_option(b"gemm256", 0) == 0 and lib.czc_test_set_option(b"wreg", 0) == 0
        for (M, N, K, resid) in ((300, 200, 64, False), (129, 768, 192, True), (1400, 512, 2048, True), (77, 640, 320, False),
                                 (3840, 768, 3072, True), (40, 512, 512, True)):
            A = rng.standard_normal((M, K)).astype(np.float32)
            W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32)
            R = rng.standard_normal((M, N)).astype(np.float32) if resid else None
            outs = []
            for deep in (0, 2):
                assert lib.czc_test_set_option(b"gemm_deep", deep) == 0
                outs.append(KH.gemm(prec, A, W, bias=bias, resid=R))
            np.testing.assert_array_equal(outs[0], outs[1], err_msg=f"M={M} N={N} K={K}")
    finally:
        lib.czc_test_set_option(b"gemm_deep", 1)
        lib.czc_test_set_option(b"gemm256", 1)
        lib.czc_test_set_option(b"wreg", WREG_DEFAULT)


@pytest.mark.parametrize("prec", [BF16, FP16, F32, F16X3])
def test_tiled_gemm_small_tiles_do_not_change_results(prec):
    """Launches that would put 128-wide tiles on less than a quarter of the CUs (one or two images) run the same kernel with
    64-wide tiles (option gemm_small_tiles, default 4 = whenever the 128-wide tiles would not fill the CUs): four times the work-groups, the same k order per output element -- bit-identical
    outputs, with and without residual / activation, ragged rows and columns, one K step and many."""
    lib = native.load_test()
    rng = np.random.default_rng(11 + prec)
    try:
        assert lib.czc_test_set_option(b"gemm256", 0) == 0 and lib.czc_test_set_option(b"wreg", 0) == 0
        assert lib.czc_test_set_option(b"splitk", 0) == 0 and lib.czc_test_set_option(b"skinny", 0) == 0
        for (M, N, K, resid, act) in ((1200, 512, 2048, True, 0), (1300, 512, 512, True, 0), (77, 640, 320, False, 1), (40, 512, 512, True, 0),
                                      (333, 200, 64, False, 0), (1, 512, 128, True, 0), (700, 1024, 192, False, 1), (65, 64, 64, True, 0)):
            A = rng.standard_normal((M, K)).astype(np.float32)
            W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32)
            R = rng.standard_normal((M, N)).astype(np.float32) if resid else None
            outs = []
            for small in (0, 1):
                assert lib.czc_test_set_option(b"gemm_small_tiles", small) == 0
                outs.append(KH.gemm(prec, A, W, bias=bias, resid=R, act=act))
            np.testing.assert_array_equal(outs[0], outs[1], err_msg=f"M={M} N={N} K={K}")
    finally:
        lib.czc_test_set_option(b"gemm_small_tiles", 4)
        lib.czc_test_set_option(b"splitk", 1)
        lib.czc_test_set_option(b"skinny", 1)
        lib.czc_test_set_option(b"gemm256", 1)
        lib.czc_test_set_option(b"wreg", WREG_DEFAULT)


@pytest.mark.parametrize("label", ["tiny", "full"])
def test_bridge_token_table_equals_the_merge_loop(label):
    """The text bridge takes the CLIP ids of a chunk that is exactly one all-letter BERT piece from a per-token table the
    device fills once with the same BPE code; every other chunk ('##' continuations glued to a word, digits, punctuation,
    contractions) runs the merge loop.  Random rows over the whole vocabulary, with and without the table: identical
    ids and lengths, and the table really serves most words."""
    lib = native.load_test()
    sv = harness.cached_vocab(label == "tiny")
    bt, ct = tokenizers_from_vocab(sv)
    t = tables_from_tokenizers(bt, ct)
    rng = np.random.default_rng(123)
    V = len(sv.bert_tokens)
    rows = rng.integers(0, V, size=(300, 16)).astype(np.int32)
    rows[:, 0] = bt.vocab["[CLS]"]
    rows[:100, 8] = bt.vocab["[MASK]"]            # specials in the middle are skipped
    rows[100:200, 1:] = rng.integers(sv.regular_lo, sv.regular_hi, size=(100, 15))  # captions of whole words
    try:
        assert lib.czc_test_set_option(b"bridge_no_table", 1) == 0
        ids0, ln0 = KH.bridge(t, rows)
        assert lib.czc_test_set_option(b"bridge_no_table", 0) == 0
        ids1, ln1 = KH.bridge(t, rows)
    finally:
        lib.czc_test_set_option(b"bridge_no_table", 0)
    np.testing.assert_array_equal(ln0, ln1)
    np.testing.assert_array_equal(ids0, ids1)


# ---- round 5: the 2-byte (fp16) residual stream of the bf16 engine's CLIP-text tower ----------------------------------------

def _f16_round(a):
    return np.ascontiguousarray(a, np.float32).astype(np.float16).astype(np.float32)


def _x16_ref(prec, A, W, bias, resid):
    rnd = _bf16_round if prec == BF16 else _f16_round
    acc = rnd(A).astype(np.float64) @ rnd(W).astype(np.float64).T
    return acc + (0 if bias is None else bias.astype(np.float64)) + _f16_round(resid).astype(np.float64)


@pytest.mark.parametrize("prec", [BF16, native.PREC_FP16])
@pytest.mark.parametrize("M,N,K", [(32, 512, 512), (31, 512, 512), (1, 512, 512), (1000, 512, 512), (4097, 512, 512), (20000, 512, 512),
                                   (40000, 512, 512), (333, 256, 512), (130, 512, 2048), (9000, 512, 2048), (700, 512, 64), (8200, 1024, 512)])
def test_residual_gemm_on_fp16_rows(prec, M, N, K):
    """x <- fp16(x + A.W^T + b) for every kernel that serves it: the weight-stationary residual kernel (K = 512: one block,
    ragged last block, head / steady / tail of the block loop, one or several row sets per column group), the ping-pong ring
    kernel (>= 8192 rows, K = 2048), the tiled kernel (the rest).  Against fp64 of the rounded operands: within one fp16
    rounding of the result (the kernels sum in fp32)."""
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[:, : K // 2] *= 2.0
    bias = rng.standard_normal(N).astype(np.float32)
    resid = (rng.standard_normal((M, N)) * 2).astype(np.float32)
    lib = native.load_test()
    ref = _x16_ref(prec, A, W, bias, resid)
    tol = np.maximum(np.abs(ref), 1.0) * 2.0 ** -10 + 2e-4 * np.sqrt(K / 64)  # fp16 rounding of the result + fp32 summation order
    try:
        for min_m in (1, 6144):  # the weight-stationary residual kernel at every row count / from its product threshold on
            assert lib.czc_test_set_option(b"wreg_resid_min_m", min_m) == 0
            out = KH.gemm_x16(prec, A, W, bias, resid)
            bad = np.abs(out - ref) > tol
            assert not bad.any(), (min_m, int(bad.sum()), float(np.abs(out - ref).max()), np.argwhere(bad)[:4].tolist())
            out2 = KH.gemm_x16(prec, A, W, None, resid)  # no bias
            assert np.abs(out2 - _x16_ref(prec, A, W, None, resid)).max() < float(tol.max())
    finally:
        lib.czc_test_set_option(b"wreg_resid_min_m", 6144)


@pytest.mark.parametrize("prec", [BF16, native.PREC_FP16])
def test_residual_gemm_on_fp16_rows_gives_one_result_per_layer_shape(prec):
    """fc2 (K = 2048) on the ring kernel and on the tiled kernel (the row-count threshold moved both ways), and the tiled
    kernel's 64-wide form: bit-identical fp16 rows -- the tower must not change its bits with the batch size.  (The K = 512
    out-projection runs on the weight-stationary residual kernel at every row count.)"""
    lib = native.load_test()
    rng = np.random.default_rng(99)
    M, N, K = 8200, 512, 2048
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = (rng.standard_normal((M, N)) * 2).astype(np.float32)
    outs = {}
    try:
        for name, (mn, small, deep) in {"ring": (1, 4, 1), "tiled": (1 << 30, 0, 1), "tiled64": (1 << 30, 1 << 20, 2)}.items():
            assert lib.czc_test_set_option(b"gemm256_min_m", mn) == 0 and lib.czc_test_set_option(b"gemm_small_tiles", small) == 0
            assert lib.czc_test_set_option(b"gemm_deep", deep) == 0
            outs[name] = KH.gemm_x16(prec, A, W, bias, resid)
    finally:
        lib.czc_test_set_option(b"gemm256_min_m", 8192)
        lib.czc_test_set_option(b"gemm_small_tiles", 4)
        lib.czc_test_set_option(b"gemm_deep", 1)
    np.testing.assert_array_equal(outs["ring"], outs["tiled"])
    np.testing.assert_array_equal(outs["ring"], outs["tiled64"])


@pytest.mark.parametrize("prec", [BF16, native.PREC_FP16])
def test_layernorm_of_fp16_rows(prec):
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((1001, 512)) * 3 + 0.5).astype(np.float32)
    x[5] *= 40.0   # a row with large entries
    g = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    y = KH.layernorm_x16(prec, x, g, b, 1e-5)
    xr = torch.from_numpy(_f16_round(x)).double()
    ref = torch.nn.functional.layer_norm(xr, (512,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), 1e-5).numpy()
    rnd = _bf16_round if prec == BF16 else _f16_round
    tol = np.maximum(np.abs(ref), 1.0) * (2.0 ** -8 if prec == BF16 else 2.0 ** -11) + 1e-5
    assert (np.abs(y - ref) <= tol).all(), float(np.abs(y - ref).max())
    assert np.abs(y - rnd(ref.astype(np.float32))).max() <= float(tol.max())


def _part_ref(x_out):
    """(sum, sum of squares) per row and 32-column block of the stored fp16 rows, in fp64."""
    M, N = x_out.shape
    blk = x_out.astype(np.float64).reshape(M, N // 32, 32)
    return np.stack([blk.sum(-1), (blk * blk).sum(-1)], axis=-1).transpose(1, 0, 2)  # [N/32, M, 2]


@pytest.mark.parametrize("M,K", [(31, 512), (4097, 512), (40000, 512), (64280, 512), (130, 2048), (9000, 2048), (20000, 2048)])
def test_residual_gemm_leaves_layernorm_partials(M, K):
    """GemmArgs::row_part: every producer of the 2-byte residual stream also writes (sum, sum of squares) of the fp16 values it
    stored, per row and 32-column block.  Against fp64 sums of the returned rows; every slot written (the buffer starts as NaN)."""
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((512, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    resid = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
    out, part = KH.gemm_x16(BF16, A, W, bias, resid, want_part=True)
    assert np.isfinite(part).all()
    ref = _part_ref(out)
    np.testing.assert_allclose(part[..., 0], ref[..., 0], rtol=0, atol=2e-4)
    np.testing.assert_allclose(part[..., 1], ref[..., 1], rtol=2e-6, atol=1e-4)


def test_layernorm_partials_do_not_depend_on_the_kernel():
    """The same layer on the ring kernel, the tiled kernel (128- and 64-wide tiles) -- fc2 -- and, for K = 512, the
    weight-stationary residual kernel against the tiled one: bit-identical rows AND bit-identical partials (one association
    order in every epilogue), so the folded LayerNorm's statistics cannot change with the batch size."""
    lib = native.load_test()
    rng = np.random.default_rng(5)
    bias = rng.standard_normal(512).astype(np.float32)
    try:
        M, K = 8200, 2048
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((512, K)) * 0.03).astype(np.float32)
        resid = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
        outs = {}
        for name, (mn, small, deep) in {"ring": (1, 4, 1), "tiled": (1 << 30, 0, 1), "tiled64": (1 << 30, 1 << 20, 2)}.items():
            assert lib.czc_test_set_option(b"gemm256_min_m", mn) == 0 and lib.czc_test_set_option(b"gemm_small_tiles", small) == 0
            assert lib.czc_test_set_option(b"gemm_deep", deep) == 0
            outs[name] = KH.gemm_x16(BF16, A, W, bias, resid, want_part=True)
        for name in ("tiled", "tiled64"):
            np.testing.assert_array_equal(outs["ring"][0], outs[name][0])
            np.testing.assert_array_equal(outs["ring"][1], outs[name][1])
        # the out-projection (K = 512): weight-stationary residual kernel vs the tiled kernel that serves it below 6144 rows
        for M in (100, 8200):
            A = rng.standard_normal((M, 512)).astype(np.float32)
            W = (rng.standard_normal((512, 512)) * 0.04).astype(np.float32)
            resid = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
            pair = {}
            for name, min_m in (("wreg_resid", 1), ("tiled", 1 << 30)):
                assert lib.czc_test_set_option(b"wreg_resid_min_m", min_m) == 0
                pair[name] = KH.gemm_x16(BF16, A, W, bias, resid, want_part=True)
            np.testing.assert_array_equal(pair["wreg_resid"][0], pair["tiled"][0])
            np.testing.assert_array_equal(pair["wreg_resid"][1], pair["tiled"][1])
    finally:
        lib.czc_test_set_option(b"gemm256_min_m", 8192)
        lib.czc_test_set_option(b"gemm_small_tiles", 4)
        lib.czc_test_set_option(b"gemm_deep", 1)
        lib.czc_test_set_option(b"wreg_resid_min_m", 6144)


@pytest.mark.parametrize("prec", [BF16, native.PREC_FP16])
@pytest.mark.parametrize("M,N,act", [(33, 1536, 0), (1000, 2048, 1), (20000, 1536, 0), (50000, 2048, 1)])
def test_layernorm_folded_into_the_weight_stationary_gemm(prec, M, N, act):
    """act(LN(x) . W^T + b) computed from x itself: fp16 MFMA on (x, W * gamma with every weight row centred), scaled in the
    epilogue with the row's rstd -- against fp64 LayerNorm + GEMM of the fp16-rounded x.  Rows with a large common offset
    included: the product carries mean * (sum of the stored weight row), so the stored rows must sum to (almost) nothing."""
    rng = np.random.default_rng(M + N)
    x = (rng.standard_normal((M, 512)) * 2).astype(np.float32)
    x[::7] += 6.0          # |mean| = 3 sigma
    x[3::11, 17] = 60.0    # an outlier channel
    W = (rng.standard_normal((N, 512)) * 0.04).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(512)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(512)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xr = _f16_round(x)
    part = _part_ref(xr).astype(np.float32)
    out, rowsum = KH.ln_fold_gemm(prec, x, W, gamma, beta, bias, part, 1e-5, act, want_rowsum=True)
    # round-to-nearest alone leaves ~ sqrt(512) * 2^-12 * |w| ~ 1e-4 per row; the compensated rounding a few of the finest normal
    # fp16 steps (2^-24 = 6e-8): measured 2.4e-7 worst over these rows
    assert np.abs(rowsum).max() < 1e-6, float(np.abs(rowsum).max())
    xd = torch.from_numpy(xr).double()
    y = torch.nn.functional.layer_norm(xd, (512,), torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5)
    pre = (y @ torch.from_numpy(W).double().T + torch.from_numpy(bias).double()).numpy()
    ref = pre * (1.0 / (1.0 + np.exp(-1.702 * pre))) if act == 1 else pre
    # operand roundings: x is exact (fp16), W * gamma rounded to fp16 (2^-11): ~ 2^-11 * |y| * |W| * sqrt(512); output rounding of `prec`
    tol = np.maximum(np.abs(ref), 1.0) * (2.0 ** -8 if prec == BF16 else 2.0 ** -10) + 6e-3
    bad = np.abs(out - ref) > tol
    assert not bad.any(), (int(bad.sum()), float(np.abs(out - ref).max()), np.argwhere(bad)[:4].tolist())
    # small launches form the rows' statistics inside the GEMM (no ln_finalize launch): the same bits as the two-kernel route
    lib = native.load_test()
    try:
        assert lib.czc_test_set_option(b"wreg_stats_in_kernel", 0) == 0
        two = KH.ln_fold_gemm(prec, x, W, gamma, beta, bias, part, 1e-5, act)
    finally:
        lib.czc_test_set_option(b"wreg_stats_in_kernel", 1)
    np.testing.assert_array_equal(out, two)

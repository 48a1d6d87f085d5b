#!/usr/bin/env python3
"""CPU experiment (torch, no GPU): where does the COMMON part of the single-pass fp16 text tower's cosine error come from?

The screen-then-refine engine (CZC_PREC_REFINE) re-encodes 12 sampled candidates per image and step only to estimate the
mean error of the screening cosines.  If that common component is carried by the shared trunk rows (the
causal prefix of an image's K candidates), computing the trunk exactly (B x ~9 rows per step: free) would remove the need
for the sample.  This script emulates the fp16 tower in torch (weights and every GEMM input rounded to fp16, q/k/v and
attention probabilities stored as fp16, fp32 accumulation / residual / LayerNorm / softmax) with and without exact trunk
rows, on the random-weight CLIP ViT-B/32 text tower the bench uses, K candidates that differ in one word.

    python tools/probes/trunk_error_probe.py [K] [n_images]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conzic_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def r16(x):
    return x.half().float()


def tower(w, cfg, ids, j, mode):
    """mode 'exact' | 'fp16' | 'fp16_trunk_exact' (rows < j computed exactly in every layer, their k/v rounded once)."""
    N, Tc = ids.shape
    H, heads = cfg.hidden, cfg.heads
    d = H // heads
    x = w["text_model.embeddings.token_embedding.weight"][ids] + w["text_model.embeddings.position_embedding.weight"][:Tc]
    xe = x.clone()  # exact stream (only used by the trunk-exact mode)
    mask = torch.full((Tc, Tc), float("-inf")).triu(1)
    q16 = mode != "exact"

    def lin(y, name, quant):
        W, b = w[name + ".weight"], w[name + ".bias"]
        return F.linear(r16(y), r16(W), b) if quant else F.linear(y, W, b)

    def layer(x, p, quant, kv_override=None):
        y = F.layer_norm(x, (H,), w[p + ".layer_norm1.weight"], w[p + ".layer_norm1.bias"], cfg.eps)
        q, k, v = (lin(y, p + f".self_attn.{n}_proj", quant) for n in "qkv")
        if quant:
            q, k, v = r16(q), r16(k), r16(v)
        if kv_override is not None:
            k[:, :j], v[:, :j] = kv_override
        qh, kh, vh = (t.view(N, Tc, heads, d).transpose(1, 2) for t in (q, k, v))
        s = torch.matmul(qh, kh.transpose(-1, -2)) * d ** -0.5 + mask
        pr = torch.softmax(s, -1)
        if quant:
            pr = r16(pr)
        a = torch.matmul(pr, vh).transpose(1, 2).reshape(N, Tc, H)
        x = x + lin(a, p + ".self_attn.out_proj", quant)
        y = F.layer_norm(x, (H,), w[p + ".layer_norm2.weight"], w[p + ".layer_norm2.bias"], cfg.eps)
        h = lin(y, p + ".mlp.fc1", quant)
        h = h * torch.sigmoid(1.702 * h)
        return x + lin(h, p + ".mlp.fc2", quant), (k, v)

    for n in range(cfg.layers):
        p = f"text_model.encoder.layers.{n}"
        if mode == "fp16_trunk_exact":
            xe_new, (ke, ve) = layer(xe, p, False)
            x, _ = layer(x, p, True, kv_override=(r16(ke[:, :j]), r16(ve[:, :j])))
            xe = xe_new
        else:
            x, _ = layer(x, p, q16)
    x = F.layer_norm(x, (H,), w["text_model.final_layer_norm.weight"], w["text_model.final_layer_norm.bias"], cfg.eps)
    pooled = x[:, -1]
    W = w["text_projection.weight"]
    return F.linear(r16(pooled), r16(W)) if q16 else F.linear(pooled, W)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cfg = synth.clip_b32()
    w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in synth.make_clip_weights(cfg, 12).items()}
    rng = np.random.default_rng(5)
    Tc = 15
    print(f"K={K} Tc={Tc}; d = cos_fp16 - cos_exact over the K candidates of one image: mean (common part), max |d - mean|")
    for img in range(n_img):
        j = int(rng.integers(4, 13))  # candidate position (after BOS + 3 prompt words)
        base = rng.integers(1000, cfg.vocab - 2, size=Tc)
        base[0], base[-1] = cfg.bos_id, cfg.eos_id
        ids = torch.from_numpy(np.tile(base, (K, 1)).astype(np.int64))
        ids[:, j] = torch.from_numpy(rng.choice(np.arange(1000, cfg.vocab - 2), size=K, replace=False))
        emb = torch.from_numpy(rng.standard_normal(cfg.proj).astype(np.float32))
        emb = emb / emb.norm()

        def cos(t):
            return (t / t.norm(dim=-1, keepdim=True)) @ emb
        ce = cos(tower(w, cfg, ids, j, "exact"))
        out = []
        for mode in ("fp16", "fp16_trunk_exact"):
            dd = cos(tower(w, cfg, ids, j, mode)) - ce
            out.append((mode, float(dd.mean()), float((dd - dd.mean()).abs().max()), float(dd.abs().max())))
        print(f"image {img} (candidate at row {j}, {Tc - j} own rows): " +
              " | ".join(f"{m}: mean {mu:+.2e} max|d-mean| {dev:.2e} max|d| {mx:.2e}" for m, mu, dev, mx in out))


if __name__ == "__main__":
    main()

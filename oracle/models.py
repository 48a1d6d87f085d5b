"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-torch fp32 CPU restatement of the third-party arithmetic the reference's hot path calls
(SURVEY.md §2 rows 15-16); `transformers` is NOT imported here.  Weights are plain dicts keyed by
the HF state-dict names (numpy or torch fp32).

Follows:
* BertForMaskedLM.forward     HF:bert/modeling_bert.py:53-108 (embeddings), :111-203 (attention),
                              :282-351 (layer), :466-496 (MLM head), :909-982   -- called at gen_utils.py:69
* CLIPTextModel + projection  HF:clip/modeling_clip.py:221-256, :280-383, :494-586, :675
                                                                                  -- called at clip/clip.py:78-83
* CLIPVisionModel + projection HF:clip/modeling_clip.py:202-218, :594-656, :674   -- called at clip/clip.py:59-61

Pinned by tests/golden/*.npz captured from the real HF modules in the build container
(tests/golden/make_goldens.py); see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def to_torch(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in w.items():
        out[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
    return out


def _ln(x, w, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def _lin(x, w, prefix, bias=True):
    return F.linear(x, w[prefix + ".weight"], w[prefix + ".bias"] if bias else None)


def _mha(q, k, v, heads, scale, mask=None):
    """q,k,v [B,T,H] -> [B,T,H]; softmax(q k^T * scale + mask) v per head."""
    B, T, H = q.shape
    d = H // heads
    q = q.view(B, T, heads, d).transpose(1, 2)
    k = k.view(B, T, heads, d).transpose(1, 2)
    v = v.view(B, T, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v)
    return o.transpose(1, 2).reshape(B, T, H)


# ---------------------------------------------------------------------------------------
# BERT masked LM
# ---------------------------------------------------------------------------------------

def bert_hidden(w, cfg, ids: torch.Tensor) -> torch.Tensor:
    """ids int64 [B,T] -> last hidden state fp32 [B,T,H] (no attention mask, token_type 0)."""
    B, T = ids.shape
    x = (w["bert.embeddings.word_embeddings.weight"][ids]
         + w["bert.embeddings.token_type_embeddings.weight"][0]
         + w["bert.embeddings.position_embeddings.weight"][:T])
    x = _ln(x, w, "bert.embeddings.LayerNorm", cfg.eps)
    scale = 1.0 / math.sqrt(cfg.hidden // cfg.heads)
    for n in range(cfg.layers):
        p = f"bert.encoder.layer.{n}"
        q = _lin(x, w, p + ".attention.self.query")
        k = _lin(x, w, p + ".attention.self.key")
        v = _lin(x, w, p + ".attention.self.value")
        a = _mha(q, k, v, cfg.heads, scale)
        x = _ln(_lin(a, w, p + ".attention.output.dense") + x, w, p + ".attention.output.LayerNorm", cfg.eps)
        h = F.gelu(_lin(x, w, p + ".intermediate.dense"))
        x = _ln(_lin(h, w, p + ".output.dense") + x, w, p + ".output.LayerNorm", cfg.eps)
    return x


def bert_mlm_head(w, cfg, h: torch.Tensor) -> torch.Tensor:
    """hidden [...,H] -> logits [...,V] (transform dense + GELU + LN, tied decoder + bias)."""
    t = F.gelu(_lin(h, w, "cls.predictions.transform.dense"))
    t = _ln(t, w, "cls.predictions.transform.LayerNorm", cfg.eps)
    return F.linear(t, w["bert.embeddings.word_embeddings.weight"], w["cls.predictions.bias"])


def bert_mlm_logits(w, cfg, ids: torch.Tensor, rows: Sequence[int] | None = None) -> torch.Tensor:
    """`model(inp).logits` (gen_utils.py:69).  With `rows` only those sequence positions are
    pushed through the head (the reference computes all T rows and reads one, gen_utils.py:42)."""
    h = bert_hidden(w, cfg, ids)
    if rows is not None:
        h = h[:, list(rows)]
    return bert_mlm_head(w, cfg, h)


# ---------------------------------------------------------------------------------------
# CLIP
# ---------------------------------------------------------------------------------------

def _quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _clip_encoder(w, prefix, x, n_layers, heads, eps, mask):
    scale = (x.shape[-1] // heads) ** -0.5
    for n in range(n_layers):
        p = f"{prefix}.encoder.layers.{n}"
        r = x
        y = _ln(x, w, p + ".layer_norm1", eps)
        q = _lin(y, w, p + ".self_attn.q_proj")
        k = _lin(y, w, p + ".self_attn.k_proj")
        v = _lin(y, w, p + ".self_attn.v_proj")
        a = _mha(q, k, v, heads, scale, mask)
        x = r + _lin(a, w, p + ".self_attn.out_proj")
        r = x
        y = _ln(x, w, p + ".layer_norm2", eps)
        y = _lin(_quick_gelu(_lin(y, w, p + ".mlp.fc1")), w, p + ".mlp.fc2")
        x = r + y
    return x


def clip_text_embeds(w, cfg, ids: torch.Tensor, lengths: torch.Tensor | None = None) -> torch.Tensor:
    """`CLIP.compute_text_representation` after tokenisation (clip/clip.py:78-83): ids int64
    [N,Tc] (right-padded), lengths[N] = number of real tokens incl. BOS/EOS -> [N, proj].
    Pooling reads the EOS row (= lengths-1; HF:clip/modeling_clip.py:561-581); with a causal mask
    padding behind EOS cannot influence it, so the padding mask is omitted."""
    N, Tc = ids.shape
    x = w["text_model.embeddings.token_embedding.weight"][ids] + \
        w["text_model.embeddings.position_embedding.weight"][:Tc]
    mask = torch.full((Tc, Tc), float("-inf")).triu(1)
    x = _clip_encoder(w, "text_model", x, cfg.layers, cfg.heads, cfg.eps, mask)
    x = _ln(x, w, "text_model.final_layer_norm", cfg.eps)
    if lengths is None:
        eos = (ids == cfg.eos_id).int().argmax(dim=-1)
    else:
        eos = lengths.long() - 1
    pooled = x[torch.arange(N), eos]
    return F.linear(pooled, w["text_projection.weight"])


def clip_image_embeds(w, cfg, pixels: torch.Tensor) -> torch.Tensor:
    """`CLIP.compute_image_representation_from_image_instance` after the image processor
    (clip/clip.py:57-61): pixel_values fp32 [B,3,S,S] -> un-normalised image_embeds [B, proj]."""
    B = pixels.shape[0]
    pe = F.conv2d(pixels, w["vision_model.embeddings.patch_embedding.weight"], stride=cfg.v_patch)
    pe = pe.flatten(2).transpose(1, 2)
    cls = w["vision_model.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, pe], dim=1) + w["vision_model.embeddings.position_embedding.weight"]
    x = _ln(x, w, "vision_model.pre_layrnorm", cfg.eps)
    x = _clip_encoder(w, "vision_model", x, cfg.v_layers, cfg.v_heads, cfg.eps, None)
    pooled = _ln(x[:, 0], w, "vision_model.post_layernorm", cfg.eps)
    return F.linear(pooled, w["visual_projection.weight"])


def clip_similarity(w, image_embeds: torch.Tensor, text_embeds: torch.Tensor):
    """`compute_image_text_similarity_via_embeddings` (clip/clip.py:86-98):
    -> (softmax over K of cos*exp(logit_scale), cos) both [B,K]."""
    B = image_embeds.shape[0]
    t = text_embeds.view(B, -1, text_embeds.shape[-1])
    i = image_embeds / image_embeds.norm(dim=-1, keepdim=True)
    t = t / t.norm(dim=-1, keepdim=True)
    scale = w["logit_scale"].exp()
    logits = torch.matmul(t, i.unsqueeze(-1)).squeeze(-1) * scale
    return logits.softmax(dim=1), logits / scale

"""Real-checkpoint path: build an engine straight from local Hugging Face checkpoint
directories -- `config.json` + `model.safetensors` (or sharded `model-*.safetensors`), `vocab.txt` for BERT and
`vocab.json` / `merges.txt` for CLIP -- without instantiating the torch modules the reference loads at
demo.py:125-132.  Tensors go to `czc_load_tensor` under their Hugging Face names (include/conzic_hip.h).

Nothing here downloads anything: the directories must exist (this image has no network and no checkpoints; the
tests write tiny synthetic ones in the same format)."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import native, synth
from .bridge import tables_from_tokenizers
from .engine import Engine
from .text import ClipBpeTokenizer, WordPieceTokenizer


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """All tensors of a `.safetensors` file (or every shard of a directory) as fp32 numpy arrays."""
    files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {path}")
    out: Dict[str, np.ndarray] = {}
    for f in files:
        try:
            from safetensors.numpy import load_file
            part = load_file(f)
        except (TypeError, ValueError):  # bf16 shards: numpy has no bfloat16, go through torch
            from safetensors.torch import load_file as load_torch
            part = {k: v.float().numpy() for k, v in load_torch(f).items()}
        for k, v in part.items():
            out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def bert_cfg_from_json(cfg: dict) -> synth.BertCfg:
    if cfg.get("hidden_act", "gelu") != "gelu":
        raise ValueError("BERT checkpoints with hidden_act != gelu are not supported")
    return synth.BertCfg(vocab=cfg["vocab_size"], hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"],
                         heads=cfg["num_attention_heads"], inter=cfg["intermediate_size"],
                         max_pos=cfg["max_position_embeddings"], eps=cfg.get("layer_norm_eps", 1e-12))


def clip_cfg_from_json(cfg: dict) -> synth.ClipCfg:
    t, v = cfg.get("text_config", {}) or {}, cfg.get("vision_config", {}) or {}
    if t.get("hidden_act", "quick_gelu") != "quick_gelu" or v.get("hidden_act", "quick_gelu") != "quick_gelu":
        raise ValueError("CLIP checkpoints with hidden_act != quick_gelu are not supported")
    return synth.ClipCfg(vocab=t.get("vocab_size", 49408), hidden=t.get("hidden_size", 512), layers=t.get("num_hidden_layers", 12),
                         heads=t.get("num_attention_heads", 8), inter=t.get("intermediate_size", 2048),
                         max_pos=t.get("max_position_embeddings", 77), eps=t.get("layer_norm_eps", 1e-5),
                         proj=cfg.get("projection_dim", 512), bos_id=t.get("bos_token_id", 49406),
                         eos_id=t.get("eos_token_id", 49407), v_hidden=v.get("hidden_size", 768),
                         v_layers=v.get("num_hidden_layers", 12), v_heads=v.get("num_attention_heads", 12),
                         v_inter=v.get("intermediate_size", 3072), v_image=v.get("image_size", 224),
                         v_patch=v.get("patch_size", 32), logit_scale=float(cfg.get("logit_scale_init_value", 2.6592)))


def load_tokenizers(bert_dir: str, clip_dir: str) -> Tuple[WordPieceTokenizer, ClipBpeTokenizer]:
    with open(os.path.join(bert_dir, "vocab.txt"), encoding="utf-8") as f:
        bert_tokens = [line.rstrip("\n") for line in f]
    with open(os.path.join(clip_dir, "vocab.json"), encoding="utf-8") as f:
        clip_vocab = json.load(f)
    merges = []
    with open(os.path.join(clip_dir, "merges.txt"), encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("#version"):
                continue
            a, b = line.split(" ")
            merges.append((a, b))
    return WordPieceTokenizer(bert_tokens), ClipBpeTokenizer(clip_vocab, merges)


def engine_from_checkpoints(bert_dir: str, clip_dir: str, precision: Optional[int] = None, device: int = 0):
    """-> (engine, bert_cfg, clip_cfg, bert_tokenizer, clip_tokenizer).  `logit_scale` comes from the checkpoint
    tensor of that name (config.json only holds its init value); precision None = chosen from it
    (`runtime.choose_precision`: split-fp16 MFMA for the published checkpoints' x100, bf16 towers below x20).
    Legacy key names of old BERT checkpoints (`LayerNorm.gamma/beta`) and tensors the path never reads (pooler,
    next-sentence head) are handled by `Engine.load_state` (`engine.normalize_state_name`)."""
    with open(os.path.join(bert_dir, "config.json")) as f:
        bcfg = bert_cfg_from_json(json.load(f))
    with open(os.path.join(clip_dir, "config.json")) as f:
        ccfg = clip_cfg_from_json(json.load(f))
    bw, cw = read_safetensors(bert_dir), read_safetensors(clip_dir)
    if "logit_scale" in cw:
        ccfg.logit_scale = float(np.asarray(cw["logit_scale"]).reshape(-1)[0])
    bt, ct = load_tokenizers(bert_dir, clip_dir)
    from .harness import special_ids
    if precision is None:
        from .runtime import choose_precision
        precision = choose_precision(ccfg.logit_scale)
    eng = Engine(bcfg, ccfg, special_ids(bt), precision, device)
    eng.load_state(bw)
    eng.load_state(cw)
    eng.finalize()
    eng.set_bridge(tables_from_tokenizers(bt, ct))
    return eng, bcfg, ccfg, bt, ct


def write_checkpoint_dirs(root: str, bert_cfg: synth.BertCfg, bert_w: dict, clip_cfg: synth.ClipCfg, clip_w: dict,
                          sv: "synth.SynthVocab") -> Tuple[str, str]:
    """Test helper: lay synthetic weights out as two local Hugging Face style checkpoint directories."""
    from safetensors.numpy import save_file
    bdir, cdir = os.path.join(root, "bert"), os.path.join(root, "clip")
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(cdir, exist_ok=True)
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in bert_w.items()}, os.path.join(bdir, "model.safetensors"))
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in clip_w.items()}, os.path.join(cdir, "model.safetensors"))
    json.dump(dict(vocab_size=bert_cfg.vocab, hidden_size=bert_cfg.hidden, num_hidden_layers=bert_cfg.layers,
                   num_attention_heads=bert_cfg.heads, intermediate_size=bert_cfg.inter,
                   max_position_embeddings=bert_cfg.max_pos, layer_norm_eps=bert_cfg.eps, hidden_act="gelu"),
              open(os.path.join(bdir, "config.json"), "w"))
    json.dump(dict(projection_dim=clip_cfg.proj, logit_scale_init_value=clip_cfg.logit_scale,
                   text_config=dict(vocab_size=clip_cfg.vocab, hidden_size=clip_cfg.hidden, num_hidden_layers=clip_cfg.layers,
                                    num_attention_heads=clip_cfg.heads, intermediate_size=clip_cfg.inter,
                                    max_position_embeddings=clip_cfg.max_pos, layer_norm_eps=clip_cfg.eps,
                                    bos_token_id=clip_cfg.bos_id, eos_token_id=clip_cfg.eos_id, hidden_act="quick_gelu"),
                   vision_config=dict(hidden_size=clip_cfg.v_hidden, num_hidden_layers=clip_cfg.v_layers,
                                      num_attention_heads=clip_cfg.v_heads, intermediate_size=clip_cfg.v_inter,
                                      image_size=clip_cfg.v_image, patch_size=clip_cfg.v_patch, hidden_act="quick_gelu")),
              open(os.path.join(cdir, "config.json"), "w"))
    with open(os.path.join(bdir, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(sv.bert_tokens) + "\n")
    json.dump(sv.clip_vocab, open(os.path.join(cdir, "vocab.json"), "w", encoding="utf-8"))
    with open(os.path.join(cdir, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in sv.clip_merges) + "\n")
    return bdir, cdir

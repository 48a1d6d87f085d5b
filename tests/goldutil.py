"""Helpers shared by the parity tests: load a golden case and rebuild its inputs from seeds."""
from __future__ import annotations

import json
import os

import numpy as np

from conzic_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


_vocab_cache = {}


def case_assets(meta):
    """(vocab, bert_cfg, clip_cfg, bert_w, clip_w, token_mask[1,V], lexicon|None) for a golden's meta."""
    key = bool(meta["tiny"])
    if key not in _vocab_cache:
        _vocab_cache[key] = synth.make_vocab_tiny() if key else synth.make_vocab()
    sv = _vocab_cache[key]
    bcfg = synth.BertCfg(**meta["bert_cfg"])
    ccfg = synth.ClipCfg(**meta["clip_cfg"])
    bw = synth.make_bert_weights(bcfg, meta["bseed"])
    cw = synth.make_clip_weights(ccfg, meta["cseed"])
    mask = synth.make_token_mask(sv, regular_only=meta["regular_only"])
    lex = synth.make_lexicon(len(sv.bert_tokens)) if (meta["gamma"] is not None and not meta.get("pos")
                                                       and not meta.get("ctx")) else None
    return sv, bcfg, ccfg, bw, cw, mask, lex


def make_oracle(meta):
    import torch
    from oracle import models as M, step as S, text as T
    sv, bcfg, ccfg, bw, cw, mask, lex = case_assets(meta)
    o = S.Oracle(M.to_torch(bw), bcfg, M.to_torch(cw), ccfg, sv.bert_tokens,
                 T.ClipBpe(sv.clip_vocab, sv.clip_merges), lexicon=lex,
                 pos_tags=synth.make_pos_tags(len(sv.bert_tokens)) if (meta.get("pos") and not meta.get("ctx")) else None)
    if meta.get("ctx"):  # the `*_ctx` goldens: the reference's scorers ran over tests/nltk_standin.py
        import nltk_standin
        o.nltk = nltk_standin.install()
    return o, sv, torch.from_numpy(mask.copy())

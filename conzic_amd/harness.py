"""Harness-side assembly: build an engine from (synthetic or caller-provided) state dicts and
tokenizers.  This is the counterpart of what demo.py:125-143 does before it calls the boundary
(load models, build token_mask); shared by tests, bench.py, __graft_entry__.smoke() and the
drop-in modules at the repo root."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import native, synth
from .bridge import BridgeArrays, tables_from_tokenizers
from .engine import Engine
from .text import ClipBpeTokenizer, WordPieceTokenizer, tokenizers_from_vocab


@dataclass
class SynthSetup:
    engine: Engine
    sv: synth.SynthVocab
    bert_cfg: synth.BertCfg
    clip_cfg: synth.ClipCfg
    bert_tok: WordPieceTokenizer
    clip_tok: ClipBpeTokenizer
    tables: BridgeArrays
    token_mask: np.ndarray  # [1,V]


def special_ids(bert_tok) -> dict:
    sp = {k: int(bert_tok.vocab[k]) for k in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ".")}
    return sp


_vocab_cache = {}


def cached_vocab(tiny: bool) -> synth.SynthVocab:
    if tiny not in _vocab_cache:
        _vocab_cache[tiny] = synth.make_vocab_tiny() if tiny else synth.make_vocab()
    return _vocab_cache[tiny]


def build_synthetic(tiny: bool, precision: int = native.PREC_BF16, bseed: int = 11, cseed: int = 12,
                    logit_scale: float = 2.6592, regular_only: bool = False, lexicon: bool = False,
                    device: int = 0, bert_w=None, clip_w=None, bert_cfg=None, clip_cfg=None) -> SynthSetup:
    sv = cached_vocab(tiny)
    if bert_cfg is None:
        bert_cfg = synth.bert_tiny(len(sv.bert_tokens)) if tiny else synth.bert_base()
    if clip_cfg is None:
        clip_cfg = synth.clip_tiny(len(sv.clip_vocab)) if tiny else synth.clip_b32()
    clip_cfg.logit_scale = logit_scale
    bt, ct = tokenizers_from_vocab(sv)
    eng = Engine(bert_cfg, clip_cfg, special_ids(bt), precision, device)
    eng.load_state(bert_w if bert_w is not None else synth.make_bert_weights(bert_cfg, bseed))
    eng.load_state(clip_w if clip_w is not None else synth.make_clip_weights(clip_cfg, cseed))
    eng.finalize()
    tables = tables_from_tokenizers(bt, ct)
    eng.set_bridge(tables)
    mask = synth.make_token_mask(sv, regular_only=regular_only)
    eng.set_token_mask(mask)
    if lexicon:
        eng.set_lexicon(synth.make_lexicon(len(sv.bert_tokens)))
    return SynthSetup(eng, sv, bert_cfg, clip_cfg, bt, ct, tables, mask)


OUTLIER_CHANNELS = (7, 93, 200, 301, 402, 499)


def outlier_clip_weights(clip_cfg, cseed: int = 12, gain: float = 1.0):
    """CLIP weights of seed `cseed` with six channels of every TEXT-tower LayerNorm gain multiplied by `gain`: the activation
    outlier channels trained checkpoints have (a few LayerNorm gains tens of times the rest), which a single-pass fp16 tower
    rounds more coarsely (tests/test_step_gpu.py, tools/refine_validate.py; gain 1 = the plain draw)."""
    cw = synth.make_clip_weights(clip_cfg, cseed)
    if gain != 1.0:
        ch = np.array(OUTLIER_CHANNELS)
        for n in range(clip_cfg.layers):
            for ln in ("layer_norm1", "layer_norm2"):
                k = f"text_model.encoder.layers.{n}.{ln}.weight"
                g = np.array(cw[k], dtype=np.float32, copy=True)
                g[ch] *= gain
                cw[k] = g
    return cw


def order_positions(order: str, L: int, iters: int, order_list=None, random_positions=None):
    """(positions, n_mask, snapshot_every) for czc_generate from the reference's visiting orders
    (gen_utils.py:64-65 sequential, :110-115 shuffle, :160-166 span, :209-210 random)."""
    if order == "sequential":
        lst = list(range(L))
        return lst * iters, [1] * (L * iters), L
    if order == "shuffle":
        lst = list(order_list)
        assert sorted(lst) == list(range(L))
        return lst * iters, [1] * (L * iters), L
    if order == "span":
        pos, nm = [], []
        for s in range(0, L, 2):
            e = min(s + 2, L)
            pos.append(s)
            nm.append(e - s)
            if e - s == 2:
                pos.append(s + 1)
                nm.append(0)
        return pos * iters, nm * iters, L
    if order == "random":
        pos = [int(p) for p in random_positions]
        assert len(pos) == L * iters
        return pos, [1] * len(pos), L
    raise ValueError(order)


def first_divergence(engine, emb_row, init_row, ref_snaps, got_snaps, L, seed_len, K, hp, positions_per_sweep=None):
    """Where and how closely one image left a reference trajectory.  `ref_snaps` / `got_snaps`: int32 [S, T] per-sweep snapshots
    of that image from the reference engine (`engine`, e.g. the all-split one) and from the engine under test, sequential
    (or `positions_per_sweep`) order.  Replays `engine` ALONE on the image from the last common snapshot up to the first
    position whose token differs and returns what the reference engine saw there: its top-2 margin and the gap between its
    winner and the token the other engine wrote -- a gap inside the fused-score bar is a near-tie, i.e. inside the stated
    tolerance ("identical argmax ids" can only hold where the margin exceeds the error bound).  None if the snapshots agree."""
    ref_snaps, got_snaps = np.asarray(ref_snaps), np.asarray(got_snaps)
    diff = np.nonzero((ref_snaps != got_snaps).any(axis=1))[0]
    if diff.size == 0:
        return None
    s = int(diff[0])
    order = list(positions_per_sweep) if positions_per_sweep is not None else list(range(L))
    p_idx = next(i for i, p in enumerate(order) if ref_snaps[s][seed_len + p] != got_snaps[s][seed_len + p])
    cur = np.ascontiguousarray((ref_snaps[s - 1] if s > 0 else np.asarray(init_row))[None, :], dtype=np.int32).copy()
    engine.set_image_embeds(np.ascontiguousarray(emb_row[None, :], dtype=np.float32))
    r = None
    for i in range(p_idx + 1):
        p = order[i]
        r = engine.step(cur, seed_len + p, K, hp, dot_allowed=(p == L - 1), want=("cand_ids", "final_score", "best"))
        if i < p_idx and cur[0, seed_len + p] != ref_snaps[s][seed_len + p]:
            return dict(sweep=s, position=int(p), replay_mismatch=True)  # the single-image replay left the batch's trajectory itself
    fin, cand = r["final_score"][0], r["cand_ids"][0]
    srt = np.sort(fin)[::-1]
    other = int(got_snaps[s][seed_len + order[p_idx]])
    where = np.nonzero(cand == other)[0]
    gap = float(srt[0] - fin[where].max()) if where.size else None
    return dict(sweep=s, position=int(order[p_idx]), reference_top2_margin=float(srt[0] - srt[1]),
                gap_to_other_engines_choice=gap, other_choice_rank=(int((fin > fin[where].max()).sum()) if where.size else None))

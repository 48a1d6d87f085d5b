"""Glue between the reference-shaped call surface (gen_utils / control_gen_utils / clip.clip at
the repo root) and the native engine: builds ONE engine per (masked-LM, CLIP, tokenizer) triple,
pulls weights from `state_dict()`, builds the device text-bridge tables from the two tokenizers,
and turns the reference's visiting orders into czc_generate step lists.

`model` may be a HF `BertForMaskedLM` (demo.py:125) or `conzic_amd.models.SyntheticLM`;
`clip` is `clip.clip.CLIP` (this repo's drop-in); `tokenizer` a HF `BertTokenizer` or
`conzic_amd.text.WordPieceTokenizer`.  There is no CPU fallback: without the HIP library or a GPU
`Engine(...)` raises `NativeError`.
"""
from __future__ import annotations

import math
import os
import random
import time
import weakref
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import native, synth
from .bridge import tables_from_tokenizers
from .engine import Engine
from .harness import order_positions

# bf16 CLIP towers carry a cosine error of up to ~4e-3; clip/clip.py:95-98 multiplies the cosine by
# logit_scale.exp() ahead of softmax_K, so the fused score stays inside the 1e-3 budget only while that factor is
# small (measured on the full-size goldens: in budget at 14.3 = HF init, far out at 100 = published checkpoint).
BF16_MAX_LOGIT_SCALE_EXP = 20.0
# single-pass fp16 towers (same speed as bf16, 11 significand bits): cosine error ~1.8e-4, fused score 5e-5 at x14.3 and
# 2.3e-3 at x100 (measured on the full-size goldens, linear in the scale) -- in budget up to about x40
FP16_MAX_LOGIT_SCALE_EXP = 40.0

_PRECISIONS = {"bf16": native.PREC_BF16, "f32": native.PREC_F32, "fp32": native.PREC_F32,
               "split": native.PREC_SPLIT, "split_fp16": native.PREC_SPLIT, "fp16": native.PREC_FP16, "f16": native.PREC_FP16,
               "refine": native.PREC_REFINE}


def choose_precision(logit_scale: Optional[float]) -> int:
    """Engine precision for a checkpoint: CZC_PRECISION (bf16 | fp16 | refine | split | f32) when set, else by the CLIP
    logit scale -- bf16 MFMA towers where exp(logit_scale) leaves their cosine error inside the 1e-3 fused-score
    budget (<= x20), single-pass fp16 towers up to x40, and above that -- the published checkpoints, x100 -- the
    screen-then-refine engine: every candidate through the single-pass fp16 text tower, the candidates that carry the
    softmax_K mass re-encoded by the split-fp16 tower (fused score inside 1e-3 on all K candidates, trajectories
    identical to the reference on the goldens, ~1.7x the all-split engine's throughput).  Unknown scale: all split-fp16."""
    p = os.environ.get("CZC_PRECISION", "auto").lower()
    if p != "auto":
        return _PRECISIONS[p]
    if logit_scale is None:
        return native.PREC_SPLIT
    if math.exp(float(logit_scale)) > FP16_MAX_LOGIT_SCALE_EXP:
        return native.PREC_REFINE
    if math.exp(float(logit_scale)) > BF16_MAX_LOGIT_SCALE_EXP:
        return native.PREC_FP16
    return native.PREC_BF16


def _logit_scale_of(clip) -> Optional[float]:
    try:
        v = clip.clip_state_dict()["logit_scale"]
        if hasattr(v, "detach"):
            v = v.detach().float().cpu().numpy()
        return float(np.asarray(v, dtype=np.float32).reshape(-1)[0])
    except (KeyError, AttributeError, TypeError, ValueError):
        return None


class _Entry:
    """One cached engine plus weak references to the objects it was built from: an entry is only ever returned
    for the very objects that built it (ids can be recycled after garbage collection), and it closes its engine
    when any of them dies."""

    def __init__(self, eng: Engine, objs):
        self.eng = eng
        self.refs = [weakref.ref(o) for o in objs]

    def matches(self, objs) -> bool:
        return len(objs) == len(self.refs) and all(r() is o for r, o in zip(self.refs, objs))


_ENGINES: Dict[tuple, _Entry] = {}


def _lookup(key, objs) -> Optional[Engine]:
    ent = _ENGINES.get(key)
    if ent is None:
        return None
    if ent.matches(objs):
        return ent.eng
    ent.eng.close()  # stale: the ids were re-used by new objects
    del _ENGINES[key]
    return None


def _store(key, eng: Engine, objs) -> None:
    _ENGINES[key] = _Entry(eng, objs)
    for o in objs:
        try:
            weakref.finalize(o, evict, key)
        except TypeError:
            pass


def evict(key=None) -> None:
    """Close cached engines (all of them when key is None) and release their device memory."""
    keys = list(_ENGINES) if key is None else [key]
    for k in keys:
        ent = _ENGINES.pop(k, None)
        if ent is not None:
            ent.eng.close()


def _to_numpy_state(sd) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().float().cpu().numpy()
        out[k] = np.asarray(v, dtype=np.float32)
    return out


def bert_cfg_of(model) -> synth.BertCfg:
    c = getattr(model, "czc_cfg", None)
    if c is not None:
        return c
    h = model.config  # HF BertConfig
    return synth.BertCfg(vocab=h.vocab_size, hidden=h.hidden_size, layers=h.num_hidden_layers,
                         heads=h.num_attention_heads, inter=h.intermediate_size, max_pos=h.max_position_embeddings,
                         eps=h.layer_norm_eps)


def clip_cfg_of(clip) -> synth.ClipCfg:
    c = getattr(clip, "czc_cfg", None)
    if c is not None:
        return c
    h = clip.model.config  # HF CLIPConfig
    t, v = h.text_config, h.vision_config
    return synth.ClipCfg(vocab=t.vocab_size, hidden=t.hidden_size, layers=t.num_hidden_layers,
                         heads=t.num_attention_heads, inter=t.intermediate_size, max_pos=t.max_position_embeddings,
                         eps=t.layer_norm_eps, proj=h.projection_dim, bos_id=clip.tokenizer.bos_token_id,
                         eos_id=clip.tokenizer.eos_token_id, v_hidden=v.hidden_size, v_layers=v.num_hidden_layers,
                         v_heads=v.num_attention_heads, v_inter=v.intermediate_size, v_image=v.image_size,
                         v_patch=v.patch_size)


def special_ids_of(tokenizer) -> Dict[str, int]:
    vocab = tokenizer.vocab if hasattr(tokenizer, "vocab") else tokenizer.get_vocab()
    return {k: int(vocab[k]) for k in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ".")}


PRECISION_NAMES = {native.PREC_BF16: "bf16", native.PREC_F32: "f32", native.PREC_SPLIT: "split-fp16", native.PREC_FP16: "fp16",
                   native.PREC_REFINE: "screen-then-refine (fp16 + split-fp16)", native.PREC_ALL_BF16: "all-bf16"}


def get_engine(model, clip, tokenizer, device: int = 0, precision: Optional[int] = None) -> Engine:
    """One engine per (model, clip, tokenizer, precision) on first use; precision None = `choose_precision`."""
    prec = choose_precision(_logit_scale_of(clip)) if precision is None else precision
    key = (id(model), id(clip), id(tokenizer), prec, device)
    objs = (model, clip, tokenizer)
    eng = _lookup(key, objs)
    if eng is None:
        bcfg, ccfg = bert_cfg_of(model), clip_cfg_of(clip)
        eng = Engine(bcfg, ccfg, special_ids_of(tokenizer), prec, device)
        eng.load_state(model.state_dict())
        eng.load_state(clip.clip_state_dict())
        eng.finalize()
        eng.set_bridge(tables_from_tokenizers(tokenizer, clip.tokenizer))
        if prec == native.PREC_REFINE and os.environ.get("CZC_REFINE_GUARD_X1E6"):
            eng.set_option("refine_guard_x1e6", int(os.environ["CZC_REFINE_GUARD_X1E6"]))  # trip point of the guard, 1e-6 of cosine
        _store(key, eng, objs)
    clip._engine = eng
    return eng


def clip_only_engine(clip, device: int = 0) -> Engine:
    """Engine without the BERT tower, for `CLIP.compute_*` calls made outside a generate call."""
    prec = choose_precision(_logit_scale_of(clip))
    key = (id(clip), "clip-only", prec, device)
    eng = _lookup(key, (clip,))
    if eng is None:
        eng = Engine(None, clip_cfg_of(clip), {}, prec, device)
        eng.load_state(clip.clip_state_dict())
        eng.finalize()
        _store(key, eng, (clip,))
    return eng


STREAMS_MIN_IMAGES = 32  # images per stream below which a batch is not split (smaller launches lose more than overlap gains)


def _group_for(eng: Engine, batch_size: int):
    """The engine itself, or its EngineGroup (engine + replicas on their own streams) when the batch is big enough
    for CZC_STREAMS (default 2) sub-batches of at least STREAMS_MIN_IMAGES images."""
    n = int(os.environ.get("CZC_STREAMS", "2"))
    if n <= 1 or batch_size < 2 * STREAMS_MIN_IMAGES:
        return eng
    grp = getattr(eng, "_group", None)
    if grp is None or grp.streams != n or grp.engines[0] is not eng or eng.h is None:
        from .engine import EngineGroup
        if grp is not None:
            grp.close(parent=False)
        grp = EngineGroup(eng, streams=n, min_images=STREAMS_MIN_IMAGES)
        eng._group = grp
    return grp


def advance_order_rng(order: str, max_len: int, max_iters: int) -> None:
    """Consume from the process-global RNG streams exactly what one *_generation call would (gen_utils.py:110-111
    shuffle: one `random.shuffle`; :210 random: one `np.random.randint` per iteration).  A rank of an image-sharded
    run calls this for the batches it does NOT own, so that every batch sees the visiting order it would have seen
    in the single-process run (SURVEY.md §8e parity caveat)."""
    if order == "shuffle":
        random.shuffle(list(range(max_len)))
    elif order == "random":
        for _ in range(max_iters):
            np.random.randint(0, max_len)


def _mask_to_numpy(token_mask) -> np.ndarray:
    if hasattr(token_mask, "detach"):
        return token_mask.detach().float().cpu().numpy()
    return np.asarray(token_mask, dtype=np.float32)


def run_generation(order: str, img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger, max_len,
                   top_k, temperature, alpha, beta, max_iters, batch_size, verbose=True, gamma=None,
                   ctl_signal="positive", print_every: Optional[int] = None, pos_template=None):
    """Body shared by every *_generation function (gen_utils.py:51-242, control_gen_utils.py:30-134):
    returns (gen_texts_list, clip_score_sequence) with the reference's list structure."""
    import utils as ref_utils  # the repo-root drop-in (same functions as the reference's utils.py)
    eng = get_engine(model, clip, tokenizer)
    seed_len = len(prompt.split()) + 1                                   # gen_utils.py:56
    batch = ref_utils.get_init_text(tokenizer, prompt, max_len, batch_size)  # gen_utils.py:57
    clip.compute_image_representation_from_image_instance(image_instance)    # gen_utils.py:58 (cached in the engine)
    if getattr(eng, "_precision_logged", None) is None:
        scale = _logit_scale_of(clip)
        logger.info(f"engine precision: {PRECISION_NAMES.get(eng.precision, eng.precision)}"
                    + (f" (exp(logit_scale) = {math.exp(scale):.1f})" if scale is not None else ""))
        eng._precision_logged = True
    order_list = random_positions = None
    if order == "shuffle":
        order_list = list(range(max_len))
        random.shuffle(order_list)                                       # gen_utils.py:110-111 (process-global stream)
        logger.info(f"Order_list:{order_list}")
    elif order == "random":
        random_positions = [int(np.random.randint(0, max_len)) for _ in range(max_iters)]  # gen_utils.py:210
    iters = max_iters if order != "random" else max_iters // max_len
    if order == "random":
        positions, n_mask, every = [int(p) for p in random_positions], [1] * len(random_positions), 1
    else:
        positions, n_mask, every = order_positions(order, max_len, iters, order_list=order_list)
    hp = Engine.hyper(alpha, beta, temperature, gamma, ctl_signal == "negative",
                      control="pos" if pos_template is not None else None)

    def polish(eng):
        """One whole *_generation call on `eng` (which must hold the batch's image embeddings)."""
        eng.set_token_mask(_mask_to_numpy(token_mask))
        if gamma is not None:
            # control scores: caller-provided tables, else (default, CZC_CONTROL=auto) the reference's own sentence scorer
            # called back per step while the CLIP tower runs, else -- CZC_CONTROL=table -- tables built once per tokenizer
            # from nltk; raises without nltk and tables
            from . import control
            chosen = control.configure(eng, clip, tokenizer, pos_template=pos_template, ctl_signal=ctl_signal)
            if chosen != getattr(eng, "_control_logged", None):
                logger.info(f"control scores: {chosen}")
                eng._control_logged = chosen
        runner = _group_for(eng, batch_size)
        emb = None
        if runner is not eng:
            # two (CZC_STREAMS) contiguous image sub-batches on their own HIP streams over the same weights: the same
            # captions image for image (images are independent, gen_utils.py:64-81), their kernels overlap on the GPU
            from clip.clip import ImageEmbeds
            emb = image_instance.embeds if isinstance(image_instance, ImageEmbeds) else clip.last_image_embeds()
            runner.set_image_embeds(emb)
        if eng.precision == native.PREC_REFINE:
            runner.refine_guard(reset=True)
        out = runner.generate(batch_size, batch[0], max_len, seed_len, top_k, positions, hp, n_mask=n_mask,
                              snapshot_every=every)
        if runner is not eng:
            eng.set_image_embeds(emb)  # the first engine holds the whole batch again, as after a single-stream call
        return out, runner

    try:
        (ids, cos), runner = polish(eng)
    except native.NativeError as exc:
        # the bf16 engine -- and, inside czc_generate, the screening pass of the screen-then-refine engine -- keep the text tower's
        # residual stream as fp16 rows (|x| < 65504): a checkpoint whose rows leave that range shows up as a non-finite cosine
        # (CZC_ERR_OVERFLOW); its engine then goes back to fp32 rows for good
        opt = {native.PREC_BF16: "resid16", native.PREC_REFINE: "refine_rows16"}.get(eng.precision)
        if getattr(exc, "code", None) != native.ERR_OVERFLOW or "non-finite" not in str(exc) or opt is None \
                or getattr(eng, "_resid16_off", False):
            raise   # (CZC_ERR_OVERFLOW also names the text bridge's scratch overflow: no other rows would help there)
        logger.info(f"the fp16 residual stream overflowed on this checkpoint; repeating the call with fp32 rows "
                    f"(engine option {opt} = 0, kept for this engine)")
        eng.set_option(opt, 0)
        eng._resid16_off = True
        (ids, cos), runner = polish(eng)
    guard_mode = os.environ.get("CZC_REFINE_GUARD", "rerun").lower()
    if eng.precision == native.PREC_REFINE and guard_mode != "off":
        # the screen-then-refine engine's 1e-3 bound rests on the single-pass fp16 tower's error staying near what it is
        # on the validated weights; every step measures that error on the candidates it re-encodes exactly
        g = runner.refine_guard(reset=True)
        if g["tripped"]:
            trip = eng.get_option("refine_guard_generate_x1e6") * 1e-6   # the trip point the engine applied inside czc_generate
            logger.info(f"screen-then-refine guard: |screening error - mean| reached {g['max_dev']:.2e} on {g['tripped']} "
                        f"image-steps (trip point {trip:.1e})" + ("; repeating the call on the all-split engine" if guard_mode == "rerun" else ""))
            if guard_mode == "rerun":
                from clip.clip import ImageEmbeds
                emb = image_instance.embeds if isinstance(image_instance, ImageEmbeds) else clip.last_image_embeds()
                # the replicas' workspaces (one per stream) go before the second engine is built on the same GPU
                grp = getattr(eng, "_group", None)
                if grp is not None:
                    grp.close(parent=False)
                    eng._group = None
                eng2 = get_engine(model, clip, tokenizer, precision=native.PREC_SPLIT)
                eng2.set_image_embeds(emb)   # the refine engine's vision tower is the split-fp16 one: same embeddings
                (ids, cos), _ = polish(eng2)
                clip._engine = eng
    # utils.update_token_mask mutates the caller's mask in place (utils.py:53-59): leave it as the
    # reference would after the last visited position
    if positions:
        ref_utils.update_token_mask(tokenizer, token_mask, max_len, positions[-1])
    # bookkeeping (gen_utils.py:82-96)
    best_score = [0] * batch_size
    best_cap = ['None'] * batch_size
    texts_out, scores_out = [], []
    pe = print_every or 1
    for s in range(ids.shape[0]):
        cur = [float(x) for x in cos[s]]
        cur_text = tokenizer.batch_decode(ids[s].tolist(), skip_special_tokens=True)
        for jj in range(batch_size):
            if best_score[jj] < cur[jj]:
                best_score[jj] = cur[jj]
                best_cap[jj] = cur_text[jj]
        if order == "random" and not (verbose and (s + 1) % pe == 0):
            continue  # gen_utils.py:232-238: the random order only records a snapshot inside its verbose branch
        if verbose:
            for_print = tokenizer.batch_decode(ids[s].tolist())
            for jj in range(batch_size):
                logger.info(f"iter {s + 1}, The {jj + 1}-th image: {img_name[jj]},"
                            f"clip score {cur[jj]:.3f}: " + for_print[jj])
        texts_out.append(cur_text)
        scores_out.append(cur)
    texts_out.append(best_cap)
    scores_out.append(best_score)
    return texts_out, scores_out

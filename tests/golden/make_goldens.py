#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and `transformers`); nothing here is
used at test time.  The reference's own modules (`gen_utils`, `control_gen_utils`, `clip.clip`,
`utils`) are imported unchanged from /root/reference (never copied) and driven with
architecture-exact HF models (`BertForMaskedLM`, `CLIPModel`) whose weights come from
`conzic_amd.synth` (the same seeds regenerate them on the GPU box).  Intermediates are captured
by wrapping call sites from the outside (model.forward, generate_caption_step,
compute_image_text_similarity_via_raw_text, tokenizer.batch_decode, CLIP tokenizer call).

Shims: `colorlog` stub; `nltk` stub (control path only);
`sys.dont_write_bytecode` so nothing is written into the read-only reference tree.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [--only NAME]
"""
import sys
sys.dont_write_bytecode = True
import argparse
import json
import logging
import os
import random
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from conzic_amd import synth  # noqa: E402

# ---- shims ----------------------------------------------------------------------------
sys.modules["colorlog"] = types.SimpleNamespace(ColoredFormatter=lambda fmt, **kw: logging.Formatter(fmt))
# nltk and its corpora are absent: tests/nltk_standin.py (a deterministic, context-DEPENDENT tagger + a SentiWordNet-shaped
# table) is what the reference's `from nltk import pos_tag` etc. bind to.  The `*_ctx` cases run the reference's scorers
# unchanged over it; the older control cases replace the scorers' inputs by per-token tables (Tap below).
sys.path.insert(0, os.path.dirname(HERE))
import nltk_standin  # noqa: E402
nltk_standin.install()
sys.path.pop(0)
# The repo root carries drop-in modules with the reference's names (utils, gen_utils, ...): make sure
# the REAL reference is what gets imported below, and only `conzic_amd` comes from the repo.
sys.path = [p_ for p_ in sys.path if os.path.abspath(p_ or ".") != REPO]
sys.path.insert(0, REF)
for _m in ("utils", "gen_utils", "control_gen_utils", "sentiments_classifer", "POS_classifier", "clip", "clip.clip"):
    assert _m not in sys.modules, _m

import transformers  # noqa: E402
from transformers import (BertConfig, BertForMaskedLM, BertTokenizer, CLIPConfig, CLIPModel,  # noqa: E402
                          CLIPProcessor, CLIPTokenizer)
from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil  # noqa: E402

import utils as ref_utils  # noqa: E402  (reference)
import gen_utils as ref_gen  # noqa: E402
import control_gen_utils as ref_ctl  # noqa: E402
import sentiments_classifer as ref_senti  # noqa: E402
from clip.clip import CLIP as RefCLIP  # noqa: E402
for _mod in (ref_utils, ref_gen, ref_ctl, ref_senti):
    assert _mod.__file__.startswith(REF + "/"), _mod.__file__

torch.set_grad_enabled(False)


def build_hf(bcfg: synth.BertCfg, ccfg: synth.ClipCfg, sv: synth.SynthVocab, bseed, cseed, tmp):
    bt = BertTokenizer(vocab=sv.bert_vocab)
    hb = BertForMaskedLM(BertConfig(vocab_size=bcfg.vocab, hidden_size=bcfg.hidden, num_hidden_layers=bcfg.layers,
                                    num_attention_heads=bcfg.heads, intermediate_size=bcfg.inter,
                                    max_position_embeddings=bcfg.max_pos, layer_norm_eps=bcfg.eps)).eval()
    bw = synth.make_bert_weights(bcfg, bseed)
    sd = {k: torch.from_numpy(v) for k, v in bw.items()}
    missing, unexpected = hb.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m], missing
    assert not unexpected, unexpected
    ct = CLIPTokenizer(vocab=sv.clip_vocab, merges=[tuple(m) for m in sv.clip_merges], model_max_length=77)
    tc = dict(vocab_size=ccfg.vocab, hidden_size=ccfg.hidden, intermediate_size=ccfg.inter,
              num_hidden_layers=ccfg.layers, num_attention_heads=ccfg.heads, max_position_embeddings=ccfg.max_pos,
              layer_norm_eps=ccfg.eps, bos_token_id=ccfg.bos_id, eos_token_id=ccfg.eos_id, pad_token_id=ccfg.eos_id,
              projection_dim=ccfg.proj)
    vc = dict(hidden_size=ccfg.v_hidden, intermediate_size=ccfg.v_inter, num_hidden_layers=ccfg.v_layers,
              num_attention_heads=ccfg.v_heads, image_size=ccfg.v_image, patch_size=ccfg.v_patch,
              layer_norm_eps=ccfg.eps, projection_dim=ccfg.proj)
    hc = CLIPModel(CLIPConfig(text_config=tc, vision_config=vc, projection_dim=ccfg.proj)).eval()
    cw = synth.make_clip_weights(ccfg, cseed)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in cw.items()}
    missing, unexpected = hc.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m], missing
    assert not unexpected, unexpected
    d = os.path.join(tmp, f"clip_{ccfg.hidden}_{cseed}_{ccfg.logit_scale}")
    os.makedirs(d, exist_ok=True)
    hc.save_pretrained(d)
    ct.save_pretrained(d)
    ip = CLIPImageProcessorPil(size={"shortest_edge": ccfg.v_image},
                               crop_size={"height": ccfg.v_image, "width": ccfg.v_image})
    CLIPProcessor(image_processor=ip, tokenizer=ct).save_pretrained(d)
    clip = RefCLIP(d).eval()
    return hb, bt, clip


class Tap:
    """Wraps the reference's call sites from outside and records what flows through them."""

    def __init__(self, model, clip, tok, lexicon=None, sign=1.0, pos_tags=None, ctx=False):
        self.ctx = ctx
        self.steps = []
        self.snaps = []
        self.cur = None
        self.model, self.clip, self.tok = model, clip, tok
        self.lexicon = lexicon
        self.sign = sign
        self.special = set(tok.all_special_ids)
        self._senti_queue = []
        self.pos_tags = pos_tags
        self._pos_queue = []
        self.id2tok = tok.convert_ids_to_tokens(list(range(tok.vocab_size)))

        orig_fwd = model.forward

        def fwd(inp, *a, **k):
            self.cur = dict(inp_before=inp.clone().numpy().astype(np.int32))
            self._last_out = orig_fwd(inp, *a, **k)
            return self._last_out
        model.forward = fwd

        def wrap_step(mod):
            orig = mod.generate_caption_step

            def gcs(out, gen_idx, mask, temperature=None, top_k=100):
                p, i = orig(out, gen_idx=gen_idx, mask=mask, temperature=temperature, top_k=top_k)
                if "probs" in self.cur:  # span: second position re-uses the same forward
                    self.cur = dict(inp_before=self.cur["inp_before"], reuse=1)
                self.cur.update(gen_idx=int(gen_idx), probs=p.clone().numpy(), idxs=i.clone().numpy().astype(np.int32),
                                logits_row=out[:, gen_idx].clone().numpy())
                return p, i
            mod.generate_caption_step = gcs
        wrap_step(ref_gen)
        wrap_step(ref_ctl)

        orig_sim = clip.compute_image_text_similarity_via_raw_text

        def sim(image_embeds, text_list):
            s, r = orig_sim(image_embeds, text_list)
            self.cur.update(clip_score=s.clone().numpy(), clip_ref=r.clone().numpy(), texts=list(text_list))
            self.steps.append(self.cur)
            return s, r
        clip.compute_image_text_similarity_via_raw_text = sim

        orig_ctok = clip.tokenizer.__class__.__call__
        tap = self

        class _CT(clip.tokenizer.__class__):
            def __call__(self_, *a, **k):
                o = orig_ctok(self_, *a, **k)
                if tap.cur is not None:
                    tap.cur["clip_ids"] = o["input_ids"].clone().numpy().astype(np.int32)
                    tap.cur["clip_lens"] = o["attention_mask"].sum(1).numpy().astype(np.int32)
                return o
        clip.tokenizer.__class__ = _CT

        orig_bd = tok.batch_decode

        def bd(ids, *a, **k):
            res = orig_bd(ids, *a, **k)
            t = ids if isinstance(ids, torch.Tensor) else torch.tensor(ids)
            skip = k.get("skip_special_tokens", False)
            if t.ndim == 2 and self.cur is not None and t.shape[0] != len(self.cur.get("inp_before", [])):
                pass
            if skip and self.lexicon is not None and t.ndim == 2:
                # scores for the nltk-free sentiment stand-in, consumed in call order
                keep = torch.ones_like(t, dtype=torch.bool)
                for s in self.special:
                    keep &= t != s
                sc = (torch.from_numpy(self.lexicon)[t] * keep).sum(1) * self.sign
                self._senti_queue = sc.tolist()
            if skip and self.pos_tags is not None and t.ndim == 2:
                # tag lists for the nltk-free POS stand-in, consumed in call order by word_tokenize
                q = []
                for row in t.tolist():
                    tags, first = [], True
                    for i in row:
                        if i in self.special:
                            continue
                        if first or not self.id2tok[i].startswith("##"):
                            tags.append(synth.UNIVERSAL_TAGS[int(self.pos_tags[i])])
                        first = False
                    q.append(tags)
                self._pos_queue = q
            if not skip and t.ndim == 2:
                self.snaps.append(t.clone().numpy().astype(np.int32))
            return res
        tok.batch_decode = bd

        if ctx:
            # the reference's scorers run UNCHANGED over the stand-in nltk (sentiments_classifer.py:9-48,
            # POS_classifier.py:6-31); only their return values are recorded, from outside, at the call sites in
            # control_gen_utils.py:56-57 / :160
            orig_s = ref_ctl.batch_texts_POS_Sentiments_analysis

            def rec_s(*a, **k):
                out = orig_s(*a, **k)
                self._ctl_raw = out[1].clone().numpy().astype(np.float32)
                return out
            ref_ctl.batch_texts_POS_Sentiments_analysis = rec_s
            orig_p = ref_ctl.batch_texts_POS_analysis

            def rec_p(texts, templ, device="cuda"):
                tags, sc = orig_p(texts, templ, device=device)
                self._ctl_raw = sc.clone().numpy().astype(np.float32)
                return tags, sc
            ref_ctl.batch_texts_POS_analysis = rec_p
            inner_sim = clip.compute_image_text_similarity_via_raw_text

            def sim2(image_embeds, text_list):
                self.cur["ctl_raw"] = self._ctl_raw.reshape(-1)  # scored just before (control_gen_utils.py:56-58 / :160-163)
                return inner_sim(image_embeds, text_list)
            clip.compute_image_text_similarity_via_raw_text = sim2
            return

        def senti_stub(text, sentiment_ctl=None):
            # stands in for sentiments_classifer.py:9-33 (nltk + SentiWordNet are absent)
            return self._senti_queue.pop(0), [], []
        ref_senti.text_POS_Sentiments_analysis = senti_stub
        # POS_classifier.py:12-14 calls word_tokenize(text) then pos_tag(words, tagset="universal"):
        # the stubs hand the per-row tag list through as the "words" (nltk + its tagger are absent)
        import POS_classifier as ref_pos
        ref_pos.word_tokenize = lambda text: self._pos_queue.pop(0)
        ref_pos.pos_tag = lambda words, tagset=None: [(w, w) for w in words]


def run_case(name, *, tiny, B, L, K, I, order, alpha=0.02, beta=2.0, temperature=0.1, gamma=None, style="positive", pos=None,
             seed=42, bseed=11, cseed=12, logit_scale=2.6592, image="synthetic", regular_only=False, tmp=None,
             keep_step_tensors=None, ctx=False):
    t0 = time.time()
    if tiny:
        sv = synth.make_vocab_tiny()
        bcfg = synth.bert_tiny(len(sv.bert_tokens))
        ccfg = synth.clip_tiny(len(sv.clip_vocab))
    else:
        sv = synth.make_vocab()
        bcfg = synth.bert_base()
        ccfg = synth.clip_b32()
    ccfg.logit_scale = logit_scale
    model, tok, clip = build_hf(bcfg, ccfg, sv, bseed, cseed, tmp)
    V = len(sv.bert_tokens)
    lexicon = synth.make_lexicon(V) if (gamma is not None and pos is None and not ctx) else None
    tap = Tap(model, clip, tok, lexicon, -1.0 if style == "negative" else 1.0,
              pos_tags=synth.make_pos_tags(V) if (pos is not None and not ctx) else None, ctx=ctx)
    token_mask = torch.from_numpy(synth.make_token_mask(sv, regular_only=regular_only))
    from PIL import Image
    if image == "synthetic":
        u8 = synth.make_images_u8(B, ccfg.v_image)
        imgs = [Image.fromarray(u8[j]) for j in range(B)]
    else:
        imgs = [Image.open(os.path.join(REF, image)).convert("RGB")]
        assert B == 1
    image_instance = imgs if B > 1 else imgs[0]
    ref_utils.set_seed(seed)
    logger = logging.getLogger("golden")
    logger.setLevel(logging.CRITICAL)
    orders_logged = []

    class L_:
        def info(self, s):
            if isinstance(s, str) and s.startswith("Order_list:"):
                orders_logged.append(json.loads(s[len("Order_list:"):]))
    names = [f"img{j}" for j in range(B)]
    kw = dict(prompt="Image of a", batch_size=B, max_len=L, top_k=K, temperature=temperature, max_iter=I,
              alpha=alpha, beta=beta, generate_order=order)
    if gamma is None:
        texts, scores = ref_gen.generate_caption(names, model, clip, tok, image_instance, token_mask, L_(), **kw)
    elif pos is not None:
        texts, scores = ref_ctl.control_generate_caption(names, model, clip, tok, image_instance, token_mask, L_(),
                                                         gamma=gamma, ctl_type="pos", pos_type=pos, **kw)
    else:
        texts, scores = ref_ctl.control_generate_caption(names, model, clip, tok, image_instance, token_mask, L_(),
                                                         gamma=gamma, ctl_type="sentiment", style_type=style, **kw)
    # image embeds as the reference computes them (processor + vision tower)
    img_emb = clip.compute_image_representation_from_image_instance(image_instance).numpy()
    steps = tap.steps
    seed_len = 4
    meta = dict(name=name, tiny=tiny, B=B, L=L, K=K, I=I, order=order, alpha=alpha, beta=beta,
                temperature=temperature, gamma=gamma, style=style, pos=pos, seed=seed, bseed=bseed, cseed=cseed,
                logit_scale=logit_scale, image=image, regular_only=regular_only, prompt="Image of a", ctx=ctx,
                order_list=orders_logged[0] if orders_logged else None,
                positions=[int(s["gen_idx"]) - seed_len for s in steps],
                reuse=[int(s.get("reuse", 0)) for s in steps],
                texts=texts, scores=[[float(x) for x in s] for s in scores],
                transformers=transformers.__version__, torch=torch.__version__,
                bert_cfg=synth.cfg_dict(bcfg), clip_cfg=synth.cfg_dict(ccfg), vocab_seed=7,
                wall_s=time.time() - t0, n_steps=len(steps))
    nkeep = len(steps) if keep_step_tensors is None else min(keep_step_tensors, len(steps))
    arrays = dict(image_embeds=img_emb.astype(np.float32),
                  snaps=np.stack(tap.snaps) if tap.snaps else np.zeros((0,), np.int32),
                  inp_before=np.stack([s["inp_before"] for s in steps]),
                  token_mask_zero_ids=np.nonzero(synth.make_token_mask(sv, regular_only=regular_only)[0] == 0)[0].astype(np.int32))
    Tc_max = max(s["clip_ids"].shape[1] for s in steps[:nkeep])
    cid = np.full((nkeep, B * K, Tc_max), -1, np.int32)
    for i, s in enumerate(steps[:nkeep]):
        cid[i, :, : s["clip_ids"].shape[1]] = s["clip_ids"]
    arrays.update(probs=np.stack([s["probs"] for s in steps[:nkeep]]),
                  idxs=np.stack([s["idxs"] for s in steps[:nkeep]]),
                  clip_score=np.stack([s["clip_score"] for s in steps[:nkeep]]),
                  clip_ref=np.stack([s["clip_ref"] for s in steps[:nkeep]]),
                  clip_ids=cid, clip_lens=np.stack([s["clip_lens"] for s in steps[:nkeep]]))
    if ctx:  # raw control score of every candidate as the reference's own scorer returned it
        arrays["ctl_raw"] = np.stack([s["ctl_raw"].reshape(B, K) for s in steps[:nkeep]])
    # one full logits row (first step) for the BERT/MLM-head parity test; top-64 of every kept step
    arrays["logits_row0"] = steps[0]["logits_row"].astype(np.float32)
    meta["texts_step0"] = steps[0]["texts"][: min(8, len(steps[0]["texts"]))]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"[golden] {name}: {len(steps)} steps, {time.time() - t0:.1f}s, {os.path.getsize(path) / 1024:.0f} KB,"
          f" final={texts[-2]} best={texts[-1]}", flush=True)
    # undo class patch so that later cases start clean
    return meta


def text_bridge_golden():
    """HF tokenizers on random id rows -> strings and CLIP ids (pins oracle/text.py)."""
    out = {}
    for label, sv in (("tiny", synth.make_vocab_tiny()), ("full", synth.make_vocab())):
        bt = BertTokenizer(vocab=sv.bert_vocab)
        ct = CLIPTokenizer(vocab=sv.clip_vocab, merges=[tuple(m) for m in sv.clip_merges], model_max_length=77)
        rng = np.random.default_rng(5)
        V = len(sv.bert_tokens)
        rows, strs, cids = [], [], []
        n = 400 if label == "tiny" else 300
        for it in range(n):
            ln = int(rng.integers(1, 24)) if it % 10 else int(rng.integers(60, 120))
            ids = [int(i) for i in rng.integers(0, V, size=ln)]
            if it % 4 == 0:
                lo, hi = sv.regular_lo, sv.regular_hi
                ids = [sv.special_ids["[CLS]"]] + [int(i) for i in rng.integers(lo, hi, size=ln)] + [sv.special_ids["[SEP]"]]
            s = bt.decode(ids, skip_special_tokens=True)
            assert s == bt.batch_decode([ids], skip_special_tokens=True)[0]
            c = ct([s], padding=True, max_length=ct.max_len_single_sentence + 2, truncation=True)["input_ids"][0]
            rows.append(ids)
            strs.append(s)
            cids.append(c)
        out[label] = dict(rows=rows, strings=strs, clip_ids=cids,
                          init_ids=bt.encode("Image of a" + bt.mask_token * 5),
                          full_decode=[bt.decode(r) for r in rows[:40]])
    with open(os.path.join(HERE, "text_bridge.json"), "w") as f:
        json.dump(out, f)
    print("[golden] text_bridge.json", os.path.getsize(os.path.join(HERE, "text_bridge.json")) // 1024, "KB")


def vision_golden(tmp):
    for label, tiny in (("tiny", True), ("full", False)):
        sv = synth.make_vocab_tiny() if tiny else synth.make_vocab()
        bcfg = synth.bert_tiny(len(sv.bert_tokens)) if tiny else synth.bert_base()
        ccfg = synth.clip_tiny(len(sv.clip_vocab)) if tiny else synth.clip_b32()
        _, _, clip = build_hf(bcfg, ccfg, sv, 11, 12, tmp)
        from PIL import Image
        u8 = synth.make_images_u8(3, ccfg.v_image)
        emb = clip.compute_image_representation_from_image_instance([Image.fromarray(u) for u in u8]).numpy()
        pv = clip.processor(images=[Image.fromarray(u) for u in u8], return_tensors="pt")["pixel_values"].numpy()
        assert np.abs(pv - synth.pixels_from_u8(u8)).max() < 1e-6, np.abs(pv - synth.pixels_from_u8(u8)).max()
        np.savez_compressed(os.path.join(HERE, f"vision_{label}.npz"), image_embeds=emb.astype(np.float32))
        print(f"[golden] vision_{label}", emb.shape)


def imageproc_golden(tmp):
    """Pins the image-processor geometry: synthetic odd-sized images through the reference's own CLIPProcessor
    (clip/clip.py:55-56); stored as the uint8 crop (pixel_values are an exact fp32 function of it)."""
    from PIL import Image
    for label, tiny in (("tiny", True), ("full", False)):
        sv = synth.make_vocab_tiny() if tiny else synth.make_vocab()
        bcfg = synth.bert_tiny(len(sv.bert_tokens)) if tiny else synth.bert_base()
        ccfg = synth.clip_tiny(len(sv.clip_vocab)) if tiny else synth.clip_b32()
        _, _, clip = build_hf(bcfg, ccfg, sv, 11, 12, tmp)
        S = ccfg.v_image
        imgs = synth.make_odd_images(S)
        if not tiny:
            imgs = imgs[:4] + imgs[4:6]
        crops = []
        for u in imgs:
            pv = clip.processor(images=Image.fromarray(u), return_tensors="pt")["pixel_values"].numpy()[0]
            u8 = np.rint((np.moveaxis(pv, 0, -1).astype(np.float64) * synth.CLIP_STD + synth.CLIP_MEAN) * 255.0)
            u8 = u8.astype(np.uint8)
            assert np.array_equal(synth.pixels_from_u8(u8[None])[0], pv), "pixel_values must be an exact function of the crop"
            crops.append(u8)
        np.savez_compressed(os.path.join(HERE, f"imageproc_{label}.npz"), crops=np.stack(crops),
                            sizes=np.array([im.shape[:2] for im in imgs], np.int32), n=np.int32(len(imgs)))
        print(f"[golden] imageproc_{label}", len(imgs), os.path.getsize(os.path.join(HERE, f"imageproc_{label}.npz")) // 1024, "KB")


CASES = dict(
    tiny_seq=dict(tiny=True, B=2, L=5, K=12, I=3, order="sequential"),
    tiny_shuffle=dict(tiny=True, B=3, L=6, K=16, I=2, order="shuffle"),
    tiny_span=dict(tiny=True, B=2, L=5, K=8, I=2, order="span"),
    tiny_random=dict(tiny=True, B=2, L=4, K=8, I=2, order="random"),
    tiny_senti_seq=dict(tiny=True, B=2, L=5, K=12, I=2, order="sequential", gamma=5.0, style="positive"),
    tiny_senti_shuffle=dict(tiny=True, B=2, L=5, K=12, I=2, order="shuffle", gamma=5.0, style="negative"),
    tiny_pos_seq=dict(tiny=True, B=2, L=5, K=12, I=2, order="sequential", gamma=5.0,
                      pos=[["DET"], ["ADJ", "NOUN"], "", ["NOUN"], ["VERB"], ["ADV"], ["ADP"], ["DET", "NOUN"], ["NOUN", "."]]),
    tiny_scale100=dict(tiny=True, B=2, L=4, K=12, I=2, order="sequential", logit_scale=4.6052),
    # the reference's own sentence scorers, unchanged, over the context-dependent stand-in tagger (tests/nltk_standin.py)
    tiny_senti_ctx=dict(tiny=True, B=2, L=5, K=12, I=2, order="sequential", gamma=5.0, style="positive", ctx=True),
    tiny_senti_ctx_neg=dict(tiny=True, B=2, L=5, K=12, I=2, order="shuffle", gamma=5.0, style="negative", ctx=True),
    tiny_pos_ctx=dict(tiny=True, B=2, L=5, K=12, I=2, order="sequential", gamma=5.0, ctx=True,
                      pos=[["DET"], ["ADJ", "NOUN"], "", ["NOUN"], ["VERB"], ["ADV"], ["ADP"], ["DET", "NOUN"], ["NOUN", "."]]),
    full_cfg1=dict(tiny=False, B=1, L=10, K=200, I=10, order="sequential", image="examples/girl.jpg",
                   keep_step_tensors=20),
    full_synth_b2=dict(tiny=False, B=2, L=10, K=200, I=1, order="shuffle", image="synthetic"),
    full_regular=dict(tiny=False, B=2, L=10, K=200, I=1, order="sequential", image="synthetic", regular_only=True),
    # published-checkpoint logit scale (ln 100, clip/clip.py:95-98) on full-size towers
    full_scale100=dict(tiny=False, B=2, L=10, K=200, I=1, order="sequential", image="synthetic", logit_scale=4.6052),
    # BASELINE configs[3] shape: shuffle order, L=15, K=512 (gen_utils.py:98-146)
    full_shuffle_k512=dict(tiny=False, B=2, L=15, K=512, I=1, order="shuffle", image="synthetic"),
    # the two remaining visiting orders on full-size towers, one sweep: span (two positions per BERT forward, gen_utils.py:148-195)
    # and random (positions from np.random, snapshots every max_len steps, gen_utils.py:197-242, :307-312)
    full_span=dict(tiny=False, B=2, L=10, K=200, I=1, order="span", image="synthetic"),
    full_random=dict(tiny=False, B=2, L=10, K=200, I=1, order="random", image="synthetic"),
    # BASELINE configs[4] shape: sentiment control, gamma=5, L=12, K=200 (control_gen_utils.py:30-80)
    full_senti=dict(tiny=False, B=2, L=12, K=200, I=1, order="sequential", image="synthetic", gamma=5.0, style="positive"),
    # POS control on full-size towers (control_gen_utils.py:136-195), the template demo.py:40-45 ships
    full_senti_ctx=dict(tiny=False, B=2, L=12, K=200, I=1, order="sequential", image="synthetic", gamma=5.0, style="positive",
                        ctx=True),
    # sentiment_shuffle_generation (control_gen_utils.py:82-134) at full size, negative style, the reference's own scorer
    full_senti_shuffle_neg_ctx=dict(tiny=False, B=2, L=12, K=200, I=1, order="shuffle", image="synthetic", gamma=5.0, style="negative",
                                    ctx=True),
    full_pos_ctx=dict(tiny=False, B=2, L=10, K=200, I=1, order="sequential", image="synthetic", gamma=5.0, ctx=True,
                      pos=[["DET"], ["ADJ", "NOUN"], ["NOUN"], ["VERB"], ["VERB"], ["ADV"], ["ADP"], ["DET", "NOUN"], ["NOUN"],
                           ["NOUN", "."], [".", "NOUN"], [".", "NOUN"]]),
    full_pos=dict(tiny=False, B=2, L=10, K=200, I=1, order="sequential", image="synthetic", gamma=5.0,
                  pos=[["DET"], ["ADJ", "NOUN"], ["NOUN"], ["VERB"], ["VERB"], ["ADV"], ["ADP"], ["DET", "NOUN"], ["NOUN"],
                       ["NOUN", "."], [".", "NOUN"], [".", "NOUN"]]),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.set_num_threads(8)
    with tempfile.TemporaryDirectory(prefix="czc_gold_") as tmp:
        if a.only in (None, "text"):
            text_bridge_golden()
        if a.only in (None, "vision"):
            vision_golden(tmp)
        if a.only in (None, "imageproc"):
            imageproc_golden(tmp)
        for name, kw in CASES.items():
            if a.only not in (None, name):
                continue
            if a.only is None and False:
                continue
            # each case in a fresh process keeps the monkey patches independent
            if a.only is None:
                import subprocess
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--only", name],
                                      env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
            else:
                run_case(name, tmp=tmp, **kw)


if __name__ == "__main__":
    main()

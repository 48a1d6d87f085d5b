"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's polishing loop:

* `generate_caption_step`            gen_utils.py:33-49  (dup control_gen_utils.py:12-28)
* one position-step                  gen_utils.py:66-81  (sentiment extras control_gen_utils.py:53-63)
* `sequential_generation`            gen_utils.py:51-96
* `shuffle_generation`               gen_utils.py:98-146
* `span_generation`                  gen_utils.py:148-195
* `random_generation`                gen_utils.py:197-242
* `sentiment_*_generation`           control_gen_utils.py:30-134
* `get_init_text`/`update_token_mask` utils.py:46-59

Model arithmetic comes from oracle/models.py, the text bridge from oracle/text.py.  The visiting
order is an explicit input (the reference draws it from the process-global `random` stream,
gen_utils.py:110-111); `shuffle_order` reproduces CPython's stream for a given seed.

Sentiment scores: the reference looks words up in SentiWordNet through nltk
(sentiments_classifer.py:9-33); neither nltk nor its corpora exist here, so the *values* are
"parity unpinned".  The stand-in used by both this oracle and the HIP engine is a per-BERT-token
table: score(row) = sum of lexicon[id] over the row's non-special ids (negated for "negative").
The fusion arithmetic around it (softmax over K, gamma term, repeat penalty) is pinned by goldens
captured through the reference with an nltk stub implementing the same stand-in.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import models as M
from . import text as T


@dataclass
class Oracle:
    bert_w: Dict[str, torch.Tensor]
    bert_cfg: object
    clip_w: Dict[str, torch.Tensor]
    clip_cfg: object
    id2tok: List[str]
    bpe: T.ClipBpe
    lexicon: Optional[np.ndarray] = None
    pos_tags: Optional[np.ndarray] = None
    lexicon_pos: Optional[tuple] = None  # (table [V,5], class_of_token [V]): score keyed per word and coarse POS
    nltk: Optional[object] = None  # an nltk-shaped module: the control scores are then the reference's own sentence-level
                                   # arithmetic on the decoded strings (tests: tests/nltk_standin.py, context-dependent tagger)
    vocab: Dict[str, int] = field(default_factory=dict)

    def __post_init__(self):
        self.vocab = {t: i for i, t in enumerate(self.id2tok)}
        self.mask_id = self.vocab["[MASK]"]
        self.dot_id = self.vocab["."]
        self.special = {self.vocab[t] for t in T.BERT_SPECIALS}

    # -- utils.py:46-51 ------------------------------------------------------------------
    def init_text(self, prompt: str, max_len: int, batch_size: int) -> List[List[int]]:
        ids = T.bert_encode(prompt + "[MASK]" * max_len, self.vocab)
        return [list(ids) for _ in range(batch_size)]

    # -- utils.py:53-59 ------------------------------------------------------------------
    def update_token_mask(self, token_mask: torch.Tensor, max_len: int, index: int):
        token_mask[:, self.dot_id] = 1 if index == max_len - 1 else 0
        return token_mask

    def decode(self, ids, skip_special_tokens=True):
        return T.bert_decode([int(i) for i in ids], self.id2tok, skip_special_tokens)

    # -- clip/clip.py:48-62 (after the image processor) ----------------------------------
    def image_embeds(self, pixels: np.ndarray) -> torch.Tensor:
        return M.clip_image_embeds(self.clip_w, self.clip_cfg, torch.from_numpy(pixels))

    # -- clip/clip.py:64-84 ---------------------------------------------------------------
    def tokenize_clip(self, texts: Sequence[str]):
        rows = [self.bpe.encode(t) for t in texts]
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.bpe.eos_id, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
        lens = torch.tensor([len(r) for r in rows])
        return ids, lens

    def text_embeds(self, texts: Sequence[str], chunk: int = 4096) -> torch.Tensor:
        ids, lens = self.tokenize_clip(texts)
        outs = []
        for s in range(0, ids.shape[0], chunk):  # chunking only bounds memory; rows are independent
            outs.append(M.clip_text_embeds(self.clip_w, self.clip_cfg, ids[s:s + chunk], lens[s:s + chunk]))
        return torch.cat(outs, 0)


def generate_caption_step(logits_row: torch.Tensor, mask: torch.Tensor, temperature, top_k):
    """gen_utils.py:33-49 on the already selected row `out[:, gen_idx]`."""
    logits = logits_row
    if temperature is not None:
        logits = logits / temperature
    probs = torch.softmax(logits, dim=-1)
    probs = probs * mask
    return probs.topk(top_k, dim=-1)


_PENN_TO_WORDNET = {'NN': 'n', 'NNP': 'n', 'NNPS': 'n', 'NNS': 'n', 'UH': 'n', 'VB': 'v', 'VBD': 'v', 'VBG': 'v', 'VBN': 'v',
                    'VBP': 'v', 'VBZ': 'v', 'JJ': 'a', 'JJR': 'a', 'JJS': 'a', 'RB': 'r', 'RBR': 'r', 'RBS': 'r', 'RP': 'r',
                    'WRB': 'r'}  # sentiments_classifer.py:19-22


def text_sentiment(nltk, text: str, ctl_signal: str) -> float:
    """sentiments_classifer.py:14-33 on one decoded sentence, over the nltk-shaped module `nltk`."""
    tagged = nltk.pos_tag(nltk.tokenize.word_tokenize(text))
    total = 0.0
    for word, tag in tagged:
        syn = list(nltk.corpus.sentiwordnet.senti_synsets(word, _PENN_TO_WORDNET.get(tag, '')))
        if syn:
            total += sum(x.pos_score() - x.neg_score() for x in syn) / len(syn)
    return -total if ctl_signal == "negative" else total


def text_pos_match(nltk, text: str, template) -> float:
    """POS_classifier.py:12-29 on one decoded sentence."""
    tags = [t for _, t in nltk.pos_tag(nltk.tokenize.word_tokenize(text), tagset="universal")]
    n = len(template)
    cur = tags + [""] * (n - len(tags)) if len(tags) <= n else tags[:n]
    ok = 0
    for w in range(len(cur)):
        if template[w] == "" or cur[w] in template[w]:
            ok += 1
    return ok / n


def senti_scores(o: Oracle, rows: torch.Tensor, ctl_signal: str) -> torch.Tensor:
    """Stand-in for sentiments_classifer.py:35-45 (see module docstring): [N] scores."""
    if o.nltk is not None:
        return torch.tensor([text_sentiment(o.nltk, o.decode(r, skip_special_tokens=True), ctl_signal) for r in rows.tolist()],
                            dtype=torch.float32)
    if o.lexicon_pos is not None:
        # per word (addressed by its first piece; '##' continuations add nothing) under the coarse POS class of that
        # piece: sentiments_classifer.py:14-30 with a context-free tagger
        table, cls = o.lexicon_pos
        per_tok = torch.from_numpy(np.asarray(table, np.float32)[np.arange(len(cls)), np.asarray(cls, np.int64)])
        cont = torch.tensor([t.startswith("##") for t in o.id2tok])
        sc = torch.zeros(rows.shape[0])
        for r in range(rows.shape[0]):
            first = True
            for i in rows[r].tolist():
                if i in o.special:
                    continue
                if first or not bool(cont[i]):
                    sc[r] += per_tok[i]
                first = False
        return -sc if ctl_signal == "negative" else sc
    lex = torch.from_numpy(o.lexicon)
    keep = torch.ones_like(rows, dtype=torch.bool)
    for s in o.special:
        keep &= rows != s
    sc = (lex[rows] * keep).sum(dim=1)
    return -sc if ctl_signal == "negative" else sc


def pos_scores(o: Oracle, rows: torch.Tensor, template) -> torch.Tensor:
    """Stand-in for POS_classifier.py:6-31: words = non-special pieces that do not continue a word
    ('##'), tag of a word = table tag of its first piece; acc = matches / len(template) with the
    reference's padding/wildcard rules (POS_classifier.py:17-29)."""
    from conzic_amd.synth import UNIVERSAL_TAGS
    if o.nltk is not None:
        return torch.tensor([text_pos_match(o.nltk, o.decode(r, skip_special_tokens=True), template) for r in rows.tolist()],
                            dtype=torch.float32)
    out = torch.zeros(rows.shape[0])
    for r, row in enumerate(rows.tolist()):
        tags = []
        first = True
        for i in row:
            if i in o.special:
                continue
            if first or not o.id2tok[i].startswith("##"):
                tags.append(UNIVERSAL_TAGS[int(o.pos_tags[i])])
            first = False
        total = len(template)
        cur = tags + [""] * (total - len(tags)) if len(tags) <= total else tags[:total]
        correct = 0
        for w in range(len(cur)):
            if template[w] == "":
                correct += 1
            elif cur[w] in template[w]:
                correct += 1
        out[r] = correct / total
    return out


def polish_step(o: Oracle, inp: torch.Tensor, image_embeds: torch.Tensor, token_mask: torch.Tensor,
                gen_idx: int, top_k: int, temperature, alpha: float, beta: float,
                gamma: Optional[float] = None, ctl_signal="positive",
                logits_row: Optional[torch.Tensor] = None, pos_template=None) -> dict:
    """One position-step: gen_utils.py:68-81 (+ control_gen_utils.py:53-63 when gamma is given).
    `inp` (int64 [B,T]) must already carry [MASK] at gen_idx; it is updated in place.
    Returns every intermediate the parity tests compare."""
    B = inp.shape[0]
    inp_ = inp.clone()
    if logits_row is None:
        logits_row = M.bert_mlm_logits(o.bert_w, o.bert_cfg, inp, rows=[gen_idx])[:, 0]
    probs, idxs = generate_caption_step(logits_row, token_mask, temperature, top_k)
    topk_inp = inp_.unsqueeze(1).repeat(1, top_k, 1)
    idxs_ = (idxs * token_mask[0][idxs]).long()
    topk_inp[:, :, gen_idx] = idxs_
    rows = topk_inp.view(-1, topk_inp.shape[-1])
    texts = [o.decode(r, skip_special_tokens=True) for r in rows]
    clip_ids, clip_lens = o.tokenize_clip(texts)
    text_embeds = o.text_embeds(texts)
    clip_score, clip_ref = M.clip_similarity(o.clip_w, image_embeds, text_embeds)
    final = alpha * probs + beta * clip_score
    out = dict(logits_row=logits_row, probs=probs, idxs=idxs, idxs_=idxs_, texts=texts,
               clip_ids=clip_ids, clip_lens=clip_lens, clip_score=clip_score, clip_ref=clip_ref)
    if gamma is not None and pos_template is not None:
        # control_gen_utils.py:163-169
        praw = pos_scores(o, rows, pos_template).view(B, -1)
        pprob = torch.softmax(praw / 0.1, dim=-1)
        final = final + gamma * pprob
        out.update(senti_raw=praw, senti_prob=pprob)
    elif gamma is not None:
        repeats = (idxs_[:, :, None] == topk_inp).float().sum(2) - 1
        sraw = senti_scores(o, rows, ctl_signal).view(B, -1)
        sprob = torch.softmax(sraw / 1, dim=1)
        final = final + gamma * sprob + 0.1 * (1 - torch.exp(repeats))
        out.update(repeats=repeats, senti_raw=sraw, senti_prob=sprob)
    best = final.argmax(dim=1).view(-1, 1)
    inp[:, gen_idx] = idxs_.gather(1, best).squeeze(-1)
    out.update(final=final, best=best.squeeze(-1), cur_clip=clip_ref.gather(1, best).squeeze(-1),
               inp_after=inp.clone())
    return out


def shuffle_order(max_len: int, seed: Optional[int] = None, rng: Optional[random.Random] = None) -> List[int]:
    """The `Order_list` of gen_utils.py:110-111 for a fresh `random.seed(seed)` stream."""
    r = rng if rng is not None else random.Random(seed)
    lst = list(range(max_len))
    r.shuffle(lst)
    return lst


def generate(o: Oracle, pixels: np.ndarray, token_mask: torch.Tensor, prompt: str, max_len: int,
             top_k: int, temperature, alpha: float, beta: float, max_iters: int,
             order: str = "sequential", order_list: Optional[Sequence[int]] = None,
             gamma: Optional[float] = None, ctl_signal: str = "positive",
             random_positions: Optional[Sequence[int]] = None, trace: Optional[list] = None,
             image_embeds: Optional[torch.Tensor] = None, pos_template=None):
    """The *_generation functions of gen_utils.py / control_gen_utils.py behind one signature.
    Returns (gen_texts_list, clip_score_sequence, ids_per_snapshot) with the reference's list
    structure (I snapshots + best)."""
    B = pixels.shape[0] if image_embeds is None else image_embeds.shape[0]
    seed_len = len(prompt.split()) + 1
    inp = torch.tensor(o.init_text(prompt, max_len, B))
    if image_embeds is None:
        image_embeds = o.image_embeds(pixels)
    best_score = [0] * B
    best_cap = ["None"] * B
    texts_out, scores_out, ids_out = [], [], []

    def snapshot(cur):
        cur_text = [o.decode(r, True) for r in inp]
        for j in range(B):
            if best_score[j] < cur[j]:
                best_score[j] = cur[j]
                best_cap[j] = cur_text[j]
        return cur_text

    def one(ii, logits_row=None):
        o.update_token_mask(token_mask, max_len, ii)
        r = polish_step(o, inp, image_embeds, token_mask, seed_len + ii, top_k, temperature,
                        alpha, beta, gamma, ctl_signal, logits_row, pos_template)
        if trace is not None:
            r["pos"] = ii
            trace.append(r)
        return r["cur_clip"].tolist()

    if order in ("sequential", "shuffle"):
        lst = list(range(max_len)) if order == "sequential" else list(order_list)
        for _ in range(max_iters):
            for ii in lst:
                inp[:, seed_len + ii] = o.mask_id
                cur = one(ii)
            texts_out.append(snapshot(cur))
            scores_out.append(cur)
            ids_out.append(inp.clone())
    elif order == "span":
        for _ in range(max_iters):
            for s in range(0, max_len, 2):
                e = min(s + 2, max_len)
                inp[:, seed_len + s: seed_len + e] = o.mask_id
                logits = M.bert_mlm_logits(o.bert_w, o.bert_cfg, inp, rows=list(range(seed_len + s, seed_len + e)))
                for ii in range(s, e):
                    cur = one(ii, logits[:, ii - s])
            texts_out.append(snapshot(cur))
            scores_out.append(cur)
            ids_out.append(inp.clone())
    elif order == "random":
        # generate_caption multiplies max_iter by max_len and prints every max_len (gen_utils.py:305-310)
        n = max_iters * max_len
        for t in range(n):
            kk = int(random_positions[t])
            inp[:, seed_len + kk] = o.mask_id
            cur = one(kk)
            cur_text = snapshot(cur)  # best-tracking happens every step (gen_utils.py:227-231)
            if (t + 1) % max_len == 0:
                texts_out.append(cur_text)
                scores_out.append(cur)
                ids_out.append(inp.clone())
    else:
        raise ValueError(order)
    texts_out.append(best_cap)
    scores_out.append(best_score)
    return texts_out, scores_out, ids_out

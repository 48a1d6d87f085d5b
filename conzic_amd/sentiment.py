"""Sentiment tables for the control path keyed the way the reference scores a caption.

`sentiments_classifer.py:9-33` tokenises the caption with nltk, tags it, maps the Penn tag of every word to a coarse
WordNet class ('' n v a r), and adds up, per word, the mean of `pos_score() - neg_score()` over
`sentiwordnet.senti_synsets(word, class)`.  The engine keeps that as a table lookup fused into the bridge kernel:

    table[V, 5]        score of BERT token `id` read as a whole word under class c (0 '' | 1 n | 2 v | 3 a | 4 r)
    class_of_token[V]  the class a CONTEXT-FREE tagger gives the token (nltk.pos_tag([word]))

A word is addressed by its first piece; '##' continuations add nothing (so multi-piece words score as their first
piece -- the one approximation besides the context-free tagger).  `conzic_amd/control.py` builds these tables once per
tokenizer whenever the controllable path runs without caller-provided tables (and offers the reference's own
sentence-level scorer as `CZC_CONTROL=exact`).  nltk and its corpora are absent from this image and from the GPU box, so
the builders are exercised with a stand-in `nltk` (tests/nltk_standin.py: a context-DEPENDENT tagger) and the values
stay "parity unpinned" (DESIGN.md §2); the arithmetic around the tables is pinned by the goldens, and what the
context-free approximation costs against a context-dependent tagger is measured on the `*_ctx` goldens."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

# sentiments_classifer.py:19-22
TAG_MAP = {'NN': 'n', 'NNP': 'n', 'NNPS': 'n', 'NNS': 'n', 'UH': 'n',
           'VB': 'v', 'VBD': 'v', 'VBG': 'v', 'VBN': 'v', 'VBP': 'v', 'VBZ': 'v',
           'JJ': 'a', 'JJR': 'a', 'JJS': 'a',
           'RB': 'r', 'RBR': 'r', 'RBS': 'r', 'RP': 'r', 'WRB': 'r'}
CLASSES = ('', 'n', 'v', 'a', 'r')


def word_score(senti_synsets, word: str, cls: str) -> float:
    """sentiments_classifer.py:26,30 for one word: mean of pos - neg over its synsets, 0 when it has none."""
    syn = list(senti_synsets(word, cls))
    if not syn:
        return 0.0
    return float(sum(x.pos_score() - x.neg_score() for x in syn) / len(syn))


def build_sentiwordnet_tables(bert_tokens: Sequence[str], nltk_module=None) -> Tuple[np.ndarray, np.ndarray]:
    """(table fp32 [V,5], class_of_token uint8 [V]) from SentiWordNet.  Needs nltk with the `sentiwordnet`, `wordnet`
    and `averaged_perceptron_tagger` data (app.py:280-283); raises ImportError with that message otherwise."""
    if nltk_module is None:
        try:
            import nltk as nltk_module  # noqa: F811
            from nltk.corpus import sentiwordnet  # noqa: F401
        except ImportError as exc:
            raise ImportError("build_sentiwordnet_tables needs nltk with the sentiwordnet / wordnet / "
                              "averaged_perceptron_tagger data; use a per-token table (clip.lexicon) without it") from exc
    swn = nltk_module.corpus.sentiwordnet
    V = len(bert_tokens)
    table = np.zeros((V, 5), np.float32)
    cls_of = np.zeros(V, np.uint8)
    for i, tok in enumerate(bert_tokens):
        if tok.startswith("##") or (tok.startswith("[") and tok.endswith("]")):
            continue  # continuation pieces and special tokens never start a word
        for c, name in enumerate(CLASSES):
            table[i, c] = word_score(swn.senti_synsets, tok, name)
        tag = nltk_module.pos_tag([tok])[0][1]
        cls_of[i] = CLASSES.index(TAG_MAP.get(tag, ''))
    return table, cls_of


def build_pos_tag_table(bert_tokens: Sequence[str], nltk_module=None) -> np.ndarray:
    """tag_of_token uint8 [V]: index into synth.UNIVERSAL_TAGS of the universal tag a CONTEXT-FREE call of the reference's
    tagger (`pos_tag([token], tagset="universal")`, POS_classifier.py:13-14) gives every word-start token; tags outside
    the twelve universal ones map to "X".  '##' continuations and special tokens never start a word (their entry is
    never read by the bridge kernel)."""
    from .synth import UNIVERSAL_TAGS
    if nltk_module is None:
        try:
            import nltk as nltk_module  # noqa: F811
        except ImportError as exc:
            raise ImportError("build_pos_tag_table needs nltk with the averaged_perceptron_tagger / universal_tagset "
                              "data; hand a per-token tag table over (clip.pos_tags) without it") from exc
    x = UNIVERSAL_TAGS.index("X")
    out = np.full(len(bert_tokens), x, np.uint8)
    for i, tok in enumerate(bert_tokens):
        if tok.startswith("##") or (tok.startswith("[") and tok.endswith("]")):
            continue
        tag = nltk_module.pos_tag([tok], tagset="universal")[0][1]
        out[i] = UNIVERSAL_TAGS.index(tag) if tag in UNIVERSAL_TAGS else x
    return out

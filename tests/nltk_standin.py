"""A deterministic stand-in for the three nltk entry points the reference's control path imports
(sentiments_classifer.py:1-3, POS_classifier.py:1-2): `nltk.tokenize.word_tokenize`, `nltk.pos_tag`
(Penn tags, or the universal tagset with tagset="universal") and `nltk.corpus.sentiwordnet.senti_synsets`.

TEST INFRASTRUCTURE.  nltk and its corpora exist on neither box (the tagger's and SentiWordNet's VALUES
are "parity unpinned"); what this module pins is everything around them.  Its tagger is **context-dependent** the way the
real averaged-perceptron tagger is -- a third of the words change their tag with the tag of the word before them -- so that

  * `tests/golden/make_goldens.py` runs the UNCHANGED reference scorers (`text_POS_Sentiments_analysis`,
    `batch_texts_POS_analysis`) over it and records the `*_ctx` goldens, and
  * on the GPU box the same module, installed as `nltk` in `sys.modules`, drives this repo's drop-in exactly as
    demo.py:98-103 does: the host-scorer mode (CZC_CONTROL=exact) must reproduce those goldens id for id, and the
    per-token tables the product builds from it (context-free by construction) are compared against them to put a
    number on that approximation.

Everything is a pure function of the word strings (zlib.crc32), so golden generation and the tests agree bit for bit.
"""
from __future__ import annotations

import os
import sys
import time
import types
import zlib

PENN = ['NN', 'NNS', 'VB', 'VBD', 'VBG', 'JJ', 'JJR', 'RB', 'IN', 'DT', 'CD', 'PRP', 'CC', 'RP', 'UH', 'FW', 'NN', 'JJ', 'VBZ', 'NN']
UNIVERSAL_OF = {'NN': 'NOUN', 'NNS': 'NOUN', 'VB': 'VERB', 'VBD': 'VERB', 'VBG': 'VERB', 'VBZ': 'VERB', 'JJ': 'ADJ', 'JJR': 'ADJ',
                'RB': 'ADV', 'IN': 'ADP', 'DT': 'DET', 'CD': 'NUM', 'PRP': 'PRON', 'CC': 'CONJ', 'RP': 'PRT', 'UH': 'X',
                'FW': 'X', '.': '.'}
PUNCT = set(".,;:!?'\"()-")


def _h(s: str) -> int:
    return zlib.crc32(s.encode("utf-8"))


def word_tokenize(text: str):
    """Whitespace words; a trailing punctuation mark is split off the way Treebank tokenisation splits a final period."""
    out = []
    for w in text.split():
        if len(w) > 1 and w[-1] in PUNCT and w[-2] not in PUNCT:
            out += [w[:-1], w[-1]]
        else:
            out.append(w)
    return out


def _penn(word: str, prev_tag):
    if all(c in PUNCT for c in word):
        return '.'
    h = _h(word)
    base = PENN[h % len(PENN)]
    if (h >> 8) % 3 == 0 and prev_tag is not None:
        # an ambiguous word: its reading follows the previous word's tag (odd / even class), as "run" after a
        # determiner or after a pronoun does for the real tagger
        alt = PENN[(h >> 16) % len(PENN)]
        return alt if _h(prev_tag) % 2 else base
    return base


def _spin(n_words: int) -> None:
    """Host cost of the real tagger, for throughput measurements only (bench.py --control exact): CZC_STANDIN_COST_US
    microseconds of busy CPU per 12-word sentence (nltk's averaged-perceptron tagger: roughly 300), scaled by the sentence's
    length; 0 / unset = free.  A busy loop, not a sleep: it occupies the core the way the tagger would."""
    us = float(os.environ.get("CZC_STANDIN_COST_US", "0") or 0)
    if us > 0:
        end = time.perf_counter() + us * 1e-6 * max(n_words, 1) / 12.0
        while time.perf_counter() < end:
            pass


def pos_tag(words, tagset=None):
    _spin(len(words))
    out, prev = [], None
    for w in words:
        t = _penn(w, prev)
        prev = t
        out.append((w, UNIVERSAL_OF[t] if tagset == "universal" else t))
    return out


class _Synset:
    __slots__ = ("p", "n")

    def __init__(self, p, n):
        self.p, self.n = p, n

    def pos_score(self):
        return self.p

    def neg_score(self):
        return self.n


def senti_synsets(word: str, pos: str = ''):
    """0-3 synsets per (word, coarse class) with scores in eighths, like SentiWordNet's."""
    h = _h(word + "/" + (pos or ""))
    n = h % 4
    return [_Synset(((h >> (4 + 6 * i)) % 9) / 8.0 * (((h >> (2 + i)) & 1)), ((h >> (7 + 6 * i)) % 9) / 8.0 * (((h >> (12 + i)) & 1)))
            for i in range(n)]


def install():
    """Register the stand-in as `nltk` (+ `nltk.tokenize`, `nltk.corpus`) in sys.modules; returns the module."""
    m = types.ModuleType("nltk")
    m.__standin__ = True
    m.__worker_hook__ = ("nltk_standin", "install")  # how a spawned scorer process gets the same stand-in (conzic_amd/control.py)
    m.pos_tag = pos_tag
    tok = types.ModuleType("nltk.tokenize")
    tok.word_tokenize = word_tokenize
    corpus = types.ModuleType("nltk.corpus")
    corpus.sentiwordnet = types.SimpleNamespace(senti_synsets=senti_synsets)
    m.tokenize, m.corpus, m.word_tokenize = tok, corpus, word_tokenize
    sys.modules["nltk"], sys.modules["nltk.tokenize"], sys.modules["nltk.corpus"] = m, tok, corpus
    return m


def uninstall():
    if getattr(sys.modules.get("nltk"), "__standin__", False):
        for k in ("nltk", "nltk.tokenize", "nltk.corpus"):
            sys.modules.pop(k, None)

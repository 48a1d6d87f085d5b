"""Control scores of the controllable path (control_gen_utils.py:30-195) for the native engine.

The reference scores every candidate SENTENCE on the host with nltk: sentiment = sum over the words of the mean
SentiWordNet `pos - neg` of the word under the coarse class of its (context-dependent) Penn tag
(sentiments_classifer.py:9-33); POS = fraction of template positions whose universal tag is accepted
(POS_classifier.py:12-29).  Two ways to feed the engine's fused score-combine kernel with them:

``exact``  (default where nltk imports: `CZC_CONTROL=auto|exact`) -- the reference's own arithmetic on the decoded candidate
           strings, called back from the engine once per step (`czc_set_control_callback`): identical to the reference
           whatever the tagger does with context (the `*_ctx` goldens: id for id), at the reference's own host cost
           (O(B*K) tagger calls per step; `CZC_CONTROL_WORKERS=N` spreads them over N spawned interpreters; unset: min(32, cores / 2)
           from 2048 strings per step on, the reference's serial loop below).  The engine calls the scorer while the CLIP tower of the
           same step runs (csrc/engine.hip), so its wall time only shows where it exceeds the tower's.  Parity is the
           first gate, so this is what an unchanged demo.py gets.
``table``  (`CZC_CONTROL=table`) -- per-BERT-token tables evaluated inside the text-bridge kernel, no host work per step:
           `sentiment.build_sentiwordnet_tables` / `sentiment.build_pos_tag_table` run nltk ONCE per tokenizer over the
           vocabulary.  Context-free by construction (a token is tagged alone; a multi-piece word scores as its first
           piece): the throughput mode, an APPROXIMATION of the reference's scorer -- against a tagger that lets a third
           of the words change their tag with the previous word's, 29-50 % of the image-steps of the full-size `*_ctx`
           goldens pick another winner.

A caller may still hand tables over explicitly (`clip.lexicon`, `clip.lexicon_pos`, `clip.pos_tags`: synthetic runs,
bench.py); with `auto` / `table` they win.  Without nltk and without tables the path raises -- there is no silent fallback.
"""
from __future__ import annotations

import os
import threading
import time
import weakref
from typing import Callable, Optional, Sequence

import numpy as np

from . import sentiment, synth
from .dist import local_world_size

NLTK_HELP = ("the controllable path scores candidates with nltk (sentiments_classifer.py:1-3, POS_classifier.py:1-2): "
             "install nltk with the punkt / averaged_perceptron_tagger / wordnet / sentiwordnet data (app.py:280-283), "
             "or hand tables over yourself (clip.lexicon [V] or clip.lexicon_pos ([V,5], [V]) for sentiment, "
             "clip.pos_tags [V] for POS; conzic_amd/sentiment.py)")


_NLTK_PROBE = {}


def import_nltk():
    """The nltk module with the entry points the reference imports AND their data, or None.  Importing is not enough:
    `nltk.corpus.sentiwordnet` is a lazy loader and `nltk.pos_tag` an attribute, so a box with nltk installed but without the
    punkt / tagger / sentiwordnet data (app.py:280-283) only fails at the first call -- inside the engine's callback.  Each
    entry point is therefore CALLED once here (result cached per module object); a LookupError means "not usable"."""
    try:
        import nltk
        from nltk.corpus import sentiwordnet
        from nltk.tokenize import word_tokenize
    except (ImportError, AttributeError, LookupError):
        return None
    ok = _NLTK_PROBE.get(id(nltk))
    if ok is None:
        try:
            word_tokenize("a b")
            nltk.pos_tag(["a"])
            nltk.pos_tag(["a"], tagset="universal")
            list(sentiwordnet.senti_synsets("good", "a"))
            ok = True
        except (LookupError, AttributeError, ImportError, OSError):
            ok = False
        _NLTK_PROBE[id(nltk)] = ok
    return nltk if ok else None


def host_cpus() -> int:
    """CPUs THIS process may run on (its affinity mask: conzic_amd.dist.pin_to_gpu_numa_node narrows it per rank)."""
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def default_workers(n_texts: int) -> int:
    """Interpreters a step's strings are spread over when CZC_CONTROL_WORKERS is unset: the reference's serial loop for a
    demo.py-sized step; from 2048 strings per step on (run.py-sized batches: B * K = 10^4..10^5 sentences through a Python
    tagger per step would leave the GPU waiting for the host) min(32, this rank's share of the host / 2) -- the share is the
    smaller of the process's affinity mask and cores / ranks on the host, so that 8 ranks on a 128-core box spawn 8 x 8
    interpreters, not 8 x 32."""
    if n_texts < 2048:
        return 0
    share = min(host_cpus(), max(1, (os.cpu_count() or 2) // local_world_size()))
    return max(2, min(32, share // 2))


def control_mode() -> str:
    m = os.environ.get("CZC_CONTROL", "auto").lower()
    if m not in ("auto", "table", "exact"):
        raise ValueError(f"CZC_CONTROL={m!r}: expected auto | table | exact")
    return m


def vocab_tokens(tokenizer) -> Sequence[str]:
    if hasattr(tokenizer, "convert_ids_to_tokens"):
        n = getattr(tokenizer, "vocab_size", None) or len(tokenizer.get_vocab())
        return tokenizer.convert_ids_to_tokens(list(range(n)))
    vocab = tokenizer.vocab if hasattr(tokenizer, "vocab") else tokenizer.get_vocab()
    out = [""] * len(vocab)
    for t, i in vocab.items():
        out[int(i)] = t
    return out


# one set of tables per tokenizer object (building them walks the vocabulary through nltk once: seconds)
_TABLES = weakref.WeakKeyDictionary()


def tables_for(tokenizer, kind: str, nltk_module):
    """kind 'sentiment' -> (table [V,5], class_of_token [V]); 'pos' -> tag_of_token [V]; built once per tokenizer."""
    try:
        slot = _TABLES.setdefault(tokenizer, {})
    except TypeError:  # not weak-referenceable: cache on the object
        slot = tokenizer.__dict__.setdefault("_czc_control_tables", {})
    key = (kind, id(nltk_module))
    if key not in slot:
        toks = vocab_tokens(tokenizer)
        slot[key] = (sentiment.build_sentiwordnet_tables(toks, nltk_module) if kind == "sentiment"
                     else sentiment.build_pos_tag_table(toks, nltk_module))
    return slot[key]


# ---- the reference's scorers on a decoded string (exact mode) --------------------------------------------------------

def sentence_sentiment(text: str, ctl_signal: Optional[str], nltk_module, memo: Optional[dict] = None) -> float:
    """sentiments_classifer.py:14-33 for one sentence.  `memo` caches the SentiWordNet mean per (word, class): a pure
    function of its key, and the K candidate sentences of a step share all but one word (the look-up is what nltk spends
    its time on)."""
    words = nltk_module.tokenize.word_tokenize(text)
    score = 0.0
    for word, tag in nltk_module.pos_tag(words):
        key = (word, sentiment.TAG_MAP.get(tag, ''))
        if memo is None:
            w = sentiment.word_score(nltk_module.corpus.sentiwordnet.senti_synsets, *key)
        else:
            w = memo.get(key)
            if w is None:
                w = memo[key] = sentiment.word_score(nltk_module.corpus.sentiwordnet.senti_synsets, *key)
        score += w
    return -score if ctl_signal == "negative" else score


def sentence_pos_match(text: str, template, nltk_module) -> float:
    """POS_classifier.py:12-29 for one sentence: the share of template slots whose tag is accepted ("" accepts anything;
    a sentence shorter than the template is padded with "" tags, a longer one is cut)."""
    tags = [t for _, t in nltk_module.pos_tag(nltk_module.tokenize.word_tokenize(text), tagset="universal")]
    total = len(template)
    cur = (tags + [""] * (total - len(tags)))[:total]
    correct = sum(1 for i, t in enumerate(cur) if template[i] == "" or t in template[i])
    return correct / total


def _worker_init(paths, hook):
    """Pool initializer (spawned interpreter): the parent's import paths, then an optional hook (tests install their
    stand-in nltk with it; a real nltk needs none)."""
    import sys
    for p_ in reversed(paths):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    if hook is not None:
        mod, fn = hook
        getattr(__import__(mod), fn)()


_WORKER_MEMO = {}


def _score_chunk(job):
    """Worker side: (kind, param, texts) -> list of scores with this process's nltk."""
    kind, param, texts = job
    nltk_module = import_nltk()
    if nltk_module is None:
        raise RuntimeError("control worker: nltk does not import in the worker process")
    if kind == "pos":
        return [sentence_pos_match(t, param, nltk_module) for t in texts]
    return [sentence_sentiment(t, param, nltk_module, _WORKER_MEMO) for t in texts]


class HostScorer:
    """czc_control_fn over a tokenizer: decodes the B*K candidate rows as the reference does
    (control_gen_utils.py:54-55 / :158-159, `batch_decode(skip_special_tokens=True)`) and scores each string with the
    reference's arithmetic -- `kind` "sentiment" (param = ctl_signal) or "pos" (param = the template).

    The reference scores the strings one after the other in Python; so does this class by default.  `workers` > 1
    (`CZC_CONTROL_WORKERS`) spreads a step's strings over that many SPAWNED interpreters (no fork of the process that owns the
    GPU context), each with its own nltk: every string is scored independently, so the scores are the serial ones; it is
    what makes the exact mode practical at run.py batch sizes (B*K = 10^4..10^5 strings per step)."""

    def __init__(self, tokenizer, kind: str, param, nltk_module, workers: Optional[int] = 0, worker_hook=None):
        assert kind in ("sentiment", "pos")
        self.tokenizer, self.kind, self.param, self.nltk = tokenizer, kind, param, nltk_module
        self.auto = workers is None                 # CZC_CONTROL_WORKERS unset: default_workers() of a step's size
        self.workers = 0 if workers is None else int(workers)
        self.worker_hook = worker_hook
        self.memo = {}
        self.sent_memo = {}       # sentence string -> score
        self.calls = self.asked = self.scored = 0   # callbacks, sentences asked for, sentences actually scored
        self.memo_evictions = 0   # times the sentence memo was emptied because it had reached SENT_MEMO_MAX
        self.host_seconds = 0.0   # wall time spent inside __call__ (decode + scoring), all calling threads
        self._pool = None
        # one scorer serves the parent engine AND its replicas (EngineGroup forwards set_control_callback), i.e. two host
        # threads: pool creation and the counters are guarded; Pool.map itself is thread-safe
        self._lock = threading.Lock()

    def _get_pool(self, n_texts):
        """The worker pool for a batch of `n_texts` strings, or None (score them here).  With CZC_CONTROL_WORKERS unset the
        decision is taken per call -- the first steps of a run repeat the same few strings (every image starts from the same
        prompt) and are scored in place; the pool appears with the first batch of default_workers()' size."""
        with self._lock:
            want = default_workers(n_texts) if self.auto else self.workers
            if want > 1 and n_texts >= 4 * want and self._pool is None:
                import multiprocessing as mp
                import sys
                if self.auto:
                    import logging
                    logging.getLogger("conzic").info(
                        "control scorer: %d strings in one step -> %d worker interpreters (CZC_CONTROL_WORKERS overrides)", n_texts, want)
                self._pool = mp.get_context("spawn").Pool(want, initializer=_worker_init,
                                                          initargs=(list(sys.path), self.worker_hook))
                self.workers = want
            return self._pool if self._pool is not None and n_texts >= 4 * self.workers else None

    SENT_MEMO_MAX = 1 << 21

    def score_texts(self, texts):
        """Scores of `texts`.  A sentence's score is a pure function of its string (tokenise -> tag -> look-ups), so a
        string already scored in an earlier step is not scored again: from the second sweep on most of a step's K candidate
        sentences per image were candidates of that position before (same context wherever the caption did not change).

        The result is built from a dict LOCAL to the call (memo hits captured up front + the freshly scored strings): a full
        memo is emptied without touching what this call -- or a call running on the other stream's host thread -- returns.
        The shared memo is only read and written under the lock; scoring itself (the slow part, pool.map) runs outside it."""
        uniq = list(dict.fromkeys(texts))
        with self._lock:
            memo = self.sent_memo
            got = {t: memo[t] for t in uniq if t in memo}
        todo = [t for t in uniq if t not in got]
        if todo:
            fresh = dict(zip(todo, self._score_new(todo)))
            got.update(fresh)
            with self._lock:
                if len(self.sent_memo) + len(fresh) > self.SENT_MEMO_MAX:
                    self.sent_memo.clear()          # (a batch larger than the cap simply is not remembered)
                    self.memo_evictions += 1
                if len(fresh) <= self.SENT_MEMO_MAX:
                    self.sent_memo.update(fresh)
        with self._lock:
            self.scored += len(todo)
            self.asked += len(texts)
        return [got[t] for t in texts]

    def _score_new(self, texts):
        pool = self._get_pool(len(texts))
        if pool is not None:
            n = self.workers
            per = (len(texts) + n - 1) // n
            jobs = [(self.kind, self.param, texts[i:i + per]) for i in range(0, len(texts), per)]
            out = []
            for part in pool.map(_score_chunk, jobs):
                out.extend(part)
            return out
        if self.kind == "pos":
            return [sentence_pos_match(t, self.param, self.nltk) for t in texts]
        return [sentence_sentiment(t, self.param, self.nltk, self.memo) for t in texts]

    def __call__(self, inp: np.ndarray, cand: np.ndarray, gen_idx: int) -> np.ndarray:
        t0 = time.perf_counter()
        B, K = cand.shape
        rows = np.repeat(inp[:, None, :], K, axis=1)
        rows[:, :, gen_idx] = cand
        texts = self.tokenizer.batch_decode(rows.reshape(B * K, -1).tolist(), skip_special_tokens=True)
        out = np.array(self.score_texts(texts), dtype=np.float32).reshape(B, K)
        with self._lock:
            self.calls += 1
            self.host_seconds += time.perf_counter() - t0
        return out

    def close(self):
        with self._lock:
            pool, self._pool = self._pool, None
        if pool is not None:
            pool.terminate()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def configure(eng, clip, tokenizer, *, pos_template=None, ctl_signal="positive") -> str:
    """Point `eng` at the control scores of one *_generation call; returns what was chosen
    ('caller-tables' | 'table' | 'exact').  pos_template None -> sentiment control."""
    is_pos = pos_template is not None
    explicit = (getattr(clip, "pos_tags", None) is not None) if is_pos else (
        getattr(clip, "lexicon_pos", None) is not None or getattr(clip, "lexicon", None) is not None)
    mode = control_mode()
    eng.set_control_callback(None)
    if explicit and mode != "exact":
        if is_pos:
            eng.set_pos(clip.pos_tags, synth.pos_template_masks(pos_template))
        elif getattr(clip, "lexicon_pos", None) is not None:  # (table [V,5], class_of_token [V])
            eng.set_lexicon_pos(*clip.lexicon_pos)
        else:
            eng.set_lexicon_pos(None, None)
            eng.set_lexicon(clip.lexicon)
        return "caller-tables"
    nltk_module = import_nltk()
    if nltk_module is None:
        raise RuntimeError(NLTK_HELP)
    if mode in ("exact", "auto"):
        env_workers = os.environ.get("CZC_CONTROL_WORKERS", "")
        workers = int(env_workers) if env_workers.strip() else None   # unset: by the size of the step (default_workers)
        key = ("pos", repr(pos_template)) if is_pos else ("sentiment", ctl_signal)
        prev = getattr(eng, "_control_scorer_cache", None)
        if prev is not None and prev[0] == (key, id(tokenizer), id(nltk_module), workers):
            scorer = prev[1]     # the same call again (samples_num loop): keep the scorer, its memo and its worker pool
        else:
            if prev is not None:
                prev[1].close()
            scorer = HostScorer(tokenizer, "pos" if is_pos else "sentiment", pos_template if is_pos else ctl_signal, nltk_module,
                                workers=workers, worker_hook=getattr(nltk_module, "__worker_hook__", None))
            eng._control_scorer_cache = ((key, id(tokenizer), id(nltk_module), workers), scorer)
        eng.set_control_callback(scorer)
        return "exact"
    if is_pos:
        eng.set_pos(tables_for(tokenizer, "pos", nltk_module), synth.pos_template_masks(pos_template))
    else:
        eng.set_lexicon_pos(*tables_for(tokenizer, "sentiment", nltk_module))
    return "table"

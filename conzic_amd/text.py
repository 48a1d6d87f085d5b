"""Minimal host tokenizers for the harness and benchmarks (no `transformers` needed at run time).

They expose exactly the attributes the reference touches on its tokenizers
(`mask_token`, `mask_token_id`, `encode`, `vocab['.']`, `batch_decode`, `decode`, `vocab_size`,
`convert_tokens_to_ids`; utils.py:46-59, gen_utils.py:67,75,83-84, demo.py:139-140) so that
`gen_utils.generate_caption(...)` accepts either these or the HF classes.

Host strings are produced only once per call (initial ids) and once per sweep (bookkeeping
captions); the per-step text bridge runs on the GPU (csrc/bridge.hip).
"""
from __future__ import annotations

import re
import unicodedata
from typing import Dict, List, Sequence, Tuple

from .synth import SynthVocab, bytes_to_unicode

_SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


class WordPieceTokenizer:
    """BERT-uncased style WordPiece tokenizer over an explicit token list."""

    def __init__(self, tokens: Sequence[str]):
        self.id2tok = list(tokens)
        self.vocab: Dict[str, int] = {t: i for i, t in enumerate(self.id2tok)}
        self.mask_token = "[MASK]"
        self.pad_token_id = self.vocab["[PAD]"]
        self.unk_token_id = self.vocab["[UNK]"]
        self.cls_token_id = self.vocab["[CLS]"]
        self.sep_token_id = self.vocab["[SEP]"]
        self.mask_token_id = self.vocab["[MASK]"]
        self.all_special_ids = [self.vocab[t] for t in _SPECIALS]
        self._special_set = set(self.all_special_ids)

    @property
    def vocab_size(self) -> int:
        return len(self.id2tok)

    def get_vocab(self):
        return dict(self.vocab)

    def convert_tokens_to_ids(self, toks):
        if isinstance(toks, str):
            return self.vocab.get(toks, self.unk_token_id)
        return [self.vocab.get(t, self.unk_token_id) for t in toks]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.id2tok[ids]
        return [self.id2tok[int(i)] for i in ids]

    def encode(self, text: str) -> List[int]:
        ids = [self.cls_token_id]
        for part in re.split(r"(\[PAD\]|\[UNK\]|\[CLS\]|\[SEP\]|\[MASK\])", text):
            if part in _SPECIALS:
                ids.append(self.vocab[part])
                continue
            s = unicodedata.normalize("NFD", part.lower())
            s = "".join(ch for ch in s if unicodedata.category(ch) != "Mn")
            words, cur = [], ""
            for ch in s:
                if ch.isspace():
                    if cur:
                        words.append(cur)
                    cur = ""
                elif _is_punct(ch):
                    if cur:
                        words.append(cur)
                    words.append(ch)
                    cur = ""
                else:
                    cur += ch
            if cur:
                words.append(cur)
            for wd in words:
                start, sub, bad = 0, [], False
                while start < len(wd):
                    end, found = len(wd), None
                    while start < end:
                        cand = ("##" if start > 0 else "") + wd[start:end]
                        if cand in self.vocab:
                            found = cand
                            break
                        end -= 1
                    if found is None:
                        bad = True
                        break
                    sub.append(self.vocab[found])
                    start = end
                ids += [self.unk_token_id] if bad else sub
        ids.append(self.sep_token_id)
        return ids

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        out = []
        first = True
        for i in ids:
            i = int(i)
            if skip_special_tokens and i in self._special_set:
                continue
            t = self.id2tok[i]
            if not first:
                t = t[2:] if t.startswith("##") else " " + t
            first = False
            for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"),
                         (" 'm", "'m"), (" do not", " don't"), (" 's", "'s"), (" 've", "'ve"), (" 're", "'re")):
                t = t.replace(a, b)
            out.append(t)
        return "".join(out)

    def batch_decode(self, batch, skip_special_tokens: bool = False) -> List[str]:
        if hasattr(batch, "tolist"):
            batch = batch.tolist()
        return [self.decode(r, skip_special_tokens) for r in batch]


class ClipBpeTokenizer:
    """CLIP byte-level BPE tokenizer over an explicit vocab + merge list."""

    def __init__(self, vocab: Dict[str, int], merges: Sequence[Tuple[str, str]], max_length: int = 77):
        self._vocab = dict(vocab)
        self.clip_merges = [tuple(m) for m in merges]
        self._ranks = {m: i for i, m in enumerate(self.clip_merges)}
        self.bos_token_id = self._vocab["<|startoftext|>"]
        self.eos_token_id = self._vocab["<|endoftext|>"]
        self.pad_token_id = self.eos_token_id
        self.model_max_length = max_length
        self.max_len_single_sentence = max_length - 2
        self._b2u = bytes_to_unicode()

    def get_vocab(self):
        return dict(self._vocab)

    @property
    def vocab_size(self):
        return len(self._vocab)

    @staticmethod
    def _cls(ch):
        if ch.isspace():
            return "S"
        c = unicodedata.category(ch)[0]
        return c if c in "LN" else "O"

    def _split(self, text: str) -> List[str]:
        out, i, n = [], 0, len(text)
        while i < n:
            hit = None
            for sp in ("<|startoftext|>", "<|endoftext|>", "'s", "'t", "'re", "'ve", "'m", "'ll", "'d"):
                if text.startswith(sp, i):
                    hit = sp
                    break
            if hit:
                out.append(hit)
                i += len(hit)
                continue
            c = self._cls(text[i])
            if c == "S":
                i += 1
                continue
            j = i + 1
            if c != "N":
                while j < n and self._cls(text[j]) == c:
                    j += 1
            out.append(text[i:j])
            i = j
        return out

    def encode_one(self, text: str) -> List[int]:
        text = re.sub(r"\s+", " ", unicodedata.normalize("NFC", text)).lower()
        body: List[int] = []
        for chunk in self._split(text):
            sym = [self._b2u[b] for b in chunk.encode("utf-8")]
            sym[-1] += "</w>"
            while len(sym) > 1:
                best, bi = None, -1
                for q in range(len(sym) - 1):
                    r = self._ranks.get((sym[q], sym[q + 1]))
                    if r is not None and (best is None or r < best):
                        best, bi = r, q
                if best is None:
                    break
                sym[bi:bi + 2] = [sym[bi] + sym[bi + 1]]
            body += [self._vocab.get(s, self.eos_token_id) for s in sym]
        body = body[: self.model_max_length - 2]
        return [self.bos_token_id] + body + [self.eos_token_id]

    def __call__(self, texts, padding=True, max_length=None, truncation=True, return_tensors=None):
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode_one(t) for t in texts]
        L = max(len(r) for r in rows)
        ids = [r + [self.pad_token_id] * (L - len(r)) for r in rows]
        att = [[1] * len(r) + [0] * (L - len(r)) for r in rows]
        return {"input_ids": ids, "attention_mask": att}


def tokenizers_from_vocab(sv: SynthVocab):
    return WordPieceTokenizer(sv.bert_tokens), ClipBpeTokenizer(sv.clip_vocab, sv.clip_merges)

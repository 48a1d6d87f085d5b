"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's *text bridge*: BERT ids -> string -> CLIP ids
(gen_utils.py:75 `tokenizer.batch_decode(..., skip_special_tokens=True)` followed by
clip/clip.py:71-74 `CLIPTokenizer(text_list, padding=True, max_length=77, truncation=True)`).

The arithmetic lives in the un-vendored third-party `tokenizers` 0.22.2 / `transformers` 5.15.0
packages (requirements.txt:3, unpinned); this file restates their published algorithms:

* WordPiece decoder with clean-up   (HF:bert/tokenization_bert.py:104-112 ->
  tokenizers `decoders::wordpiece::WordPiece{prefix:"##", cleanup:true}`)
* CLIP normaliser NFC / whitespace / lowercase, the regex pre-split and byte-level BPE with
  `</w>` end-of-word suffix   (HF:clip/tokenization_clip.py:78-107)
* `[CLS] $A [SEP]` / `<|startoftext|> $A <|endoftext|>` templates (HF:bert/tokenization_bert.py:128-135,
  HF:clip/tokenization_clip.py:117-122)

Pinned by tests/golden/text_bridge.json, generated in the build container by running the real
HF tokenizers on the synthetic vocabularies (tests/golden/make_goldens.py).
"""
from __future__ import annotations

import re
import unicodedata
from typing import Dict, List, Sequence, Tuple

from conzic_amd.synth import bytes_to_unicode

BERT_SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")


def _cleanup(tok: str) -> str:
    """tokenizers `decoders::wordpiece::cleanup` (applied to every piece separately)."""
    return (tok.replace(" .", ".").replace(" ?", "?").replace(" !", "!").replace(" ,", ",")
            .replace(" ' ", "'").replace(" n't", "n't").replace(" 'm", "'m")
            .replace(" do not", " don't").replace(" 's", "'s").replace(" 've", "'ve")
            .replace(" 're", "'re"))


def wordpiece_decode(tokens: Sequence[str]) -> str:
    """tokenizers `WordPiece::decode_chain`: piece i>0 is ' '+piece unless it starts with '##'."""
    out = []
    for i, t in enumerate(tokens):
        if i != 0:
            if t.startswith("##"):
                t = t[2:]
            else:
                t = " " + t
        out.append(_cleanup(t))
    return "".join(out)


def bert_decode(ids: Sequence[int], id2tok: Sequence[str], skip_special_tokens: bool = True) -> str:
    """`tokenizer.decode(ids, skip_special_tokens=...)` (gen_utils.py:75, :83-84)."""
    toks = [id2tok[i] for i in ids]
    if skip_special_tokens:
        toks = [t for t in toks if t not in BERT_SPECIALS]
    return wordpiece_decode(toks)


# ---- CLIP side -------------------------------------------------------------------------

def _cls(ch: str) -> str:
    """Character class for the CLIP pre-split regex: L (\\p{L}), N (\\p{N}), S (\\s), O (other)."""
    if ch.isspace():
        return "S"
    cat = unicodedata.category(ch)
    if cat[0] == "L":
        return "L"
    if cat[0] == "N":
        return "N"
    return "O"


_CONTRACTIONS = ("'s", "'t", "'re", "'ve", "'m", "'ll", "'d")


def clip_presplit(text: str) -> List[str]:
    """Leftmost, ordered-alternation matches of
    `<|startoftext|>|<|endoftext|>|'s|'t|'re|'ve|'m|'ll|'d|[\\p{L}]+|[\\p{N}]|[^\\s\\p{L}\\p{N}]+`
    (HF:clip/tokenization_clip.py:94-99); everything unmatched (whitespace) is dropped."""
    out: List[str] = []
    i, n = 0, len(text)
    while i < n:
        matched = False
        for sp in ("<|startoftext|>", "<|endoftext|>") + _CONTRACTIONS:
            if text.startswith(sp, i):
                out.append(sp)
                i += len(sp)
                matched = True
                break
        if matched:
            continue
        c = _cls(text[i])
        if c == "S":
            i += 1
        elif c == "L":
            j = i + 1
            while j < n and _cls(text[j]) == "L":
                j += 1
            out.append(text[i:j])
            i = j
        elif c == "N":
            out.append(text[i])
            i += 1
        else:
            j = i + 1
            while j < n and _cls(text[j]) == "O":
                j += 1
            out.append(text[i:j])
            i = j
    return out


class ClipBpe:
    """Byte-level BPE with `</w>` suffix: repeatedly merge the lowest-ranked adjacent pair
    (leftmost on ties), as tokenizers `models::bpe::Word::merge_all` does."""

    def __init__(self, vocab: Dict[str, int], merges: Sequence[Tuple[str, str]],
                 bos: str = "<|startoftext|>", eos: str = "<|endoftext|>", max_length: int = 77):
        self.vocab = vocab
        self.ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.b2u = bytes_to_unicode()
        self.bos_id = vocab[bos]
        self.eos_id = vocab[eos]
        self.unk_id = vocab[eos]
        self.max_length = max_length

    def bpe_word(self, chunk: str) -> List[int]:
        sym = [self.b2u[b] for b in chunk.encode("utf-8")]
        if not sym:
            return []
        sym[-1] = sym[-1] + "</w>"
        while len(sym) > 1:
            best, bi = None, -1
            for i in range(len(sym) - 1):
                r = self.ranks.get((sym[i], sym[i + 1]))
                if r is not None and (best is None or r < best):
                    best, bi = r, i
            if best is None:
                break
            sym[bi:bi + 2] = [sym[bi] + sym[bi + 1]]
        return [self.vocab.get(s, self.unk_id) for s in sym]

    def normalize(self, text: str) -> str:
        text = unicodedata.normalize("NFC", text)
        text = re.sub(r"\s+", " ", text)
        return text.lower()

    def encode(self, text: str) -> List[int]:
        """ids incl. BOS/EOS, truncated to max_length (clip/clip.py:71-72)."""
        body: List[int] = []
        for chunk in clip_presplit(self.normalize(text)):
            body += self.bpe_word(chunk)
        body = body[: self.max_length - 2]
        return [self.bos_id] + body + [self.eos_id]


def bridge(ids: Sequence[int], id2tok: Sequence[str], bpe: ClipBpe) -> List[int]:
    """One candidate row of BERT ids -> CLIP ids (the whole host round trip of gen_utils.py:75-76)."""
    return bpe.encode(bert_decode(ids, id2tok, skip_special_tokens=True))


# ---- BERT encode (only used once per call for the initial text, utils.py:46-51) ---------

def bert_encode(text: str, vocab: Dict[str, int]) -> List[int]:
    """`tokenizer.encode(text)`: special tokens matched verbatim, BertNormalizer (lowercase,
    strip accents), whitespace + punctuation pre-split, greedy longest-match WordPiece,
    `[CLS] ... [SEP]` template."""
    import re
    pieces = re.split(r"(\[PAD\]|\[UNK\]|\[CLS\]|\[SEP\]|\[MASK\])", text)
    ids = [vocab["[CLS]"]]
    for part in pieces:
        if part in BERT_SPECIALS:
            ids.append(vocab[part])
            continue
        s = unicodedata.normalize("NFD", part.lower())
        s = "".join(ch for ch in s if unicodedata.category(ch) != "Mn")
        words: List[str] = []
        cur = ""
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
            start, sub = 0, []
            bad = False
            while start < len(wd):
                end = len(wd)
                found = None
                while start < end:
                    cand = wd[start:end]
                    if start > 0:
                        cand = "##" + cand
                    if cand in vocab:
                        found = cand
                        break
                    end -= 1
                if found is None:
                    bad = True
                    break
                sub.append(vocab[found])
                start = end
            ids += [vocab["[UNK]"]] if bad else sub
    ids.append(vocab["[SEP]"])
    return ids


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")

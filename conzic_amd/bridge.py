"""Host-side builder of the device text-bridge tables (czc_bridge_tables, include/conzic_hip.h).

The reference turns candidate BERT ids into strings and re-tokenises them for CLIP on the host
every step (gen_utils.py:75 -> clip/clip.py:71-74).  The engine does that on the GPU; what it
needs from the two tokenizers is built here ONCE per (BERT tokenizer, CLIP tokenizer) pair:

* per BERT id: the piece's UTF-8 bytes ('##' stripped, NFC + lower-cased exactly as the CLIP
  normaliser would, HF:clip/tokenization_clip.py:90-92), a class per byte for the CLIP pre-split
  regex (\\p{L} / \\p{N} / other / space, HF:clip/tokenization_clip.py:94-99) and flags
  (special -> skipped by `skip_special_tokens=True`; '##' continuation; WordPiece clean-up
  removes the space in front: tokenizers decoders::wordpiece::cleanup);
* for CLIP: byte -> symbol ids (with and without the `</w>` suffix) and the merge list in rank
  order as (left id, right id, merged id).

Known limit (documented in DESIGN.md): NFC composition *across* two BERT pieces is not applied.
"""
from __future__ import annotations

import ctypes as C
import json
import unicodedata
from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

from . import native
from .synth import bytes_to_unicode

_NOSPACE_PREFIXES = (".", "?", "!", ",", "n't", "'m", "'s", "'ve", "'re")


def _char_class(ch: str) -> int:
    if ch.isspace():
        return 3
    cat = unicodedata.category(ch)
    if cat[0] == "L":
        return 0
    if cat[0] == "N":
        return 1
    return 2


@dataclass
class BridgeArrays:
    piece_off: np.ndarray
    piece_bytes: np.ndarray
    piece_class: np.ndarray
    piece_flags: np.ndarray
    byte_sym: np.ndarray
    byte_sym_eow: np.ndarray
    merge_left: np.ndarray
    merge_right: np.ndarray
    merge_out: np.ndarray
    bert_vocab: int
    clip_vocab: int
    bos_id: int
    eos_id: int

    def as_struct(self) -> native.BridgeTables:
        t = native.BridgeTables()
        t.bert_vocab = self.bert_vocab
        t.piece_off = self.piece_off.ctypes.data
        t.piece_bytes = self.piece_bytes.ctypes.data
        t.piece_class = self.piece_class.ctypes.data
        t.piece_flags = self.piece_flags.ctypes.data
        t.clip_vocab = self.clip_vocab
        t.byte_sym = self.byte_sym.ctypes.data
        t.byte_sym_eow = self.byte_sym_eow.ctypes.data
        t.n_merges = len(self.merge_left)
        t.merge_left = self.merge_left.ctypes.data
        t.merge_right = self.merge_right.ctypes.data
        t.merge_out = self.merge_out.ctypes.data
        t.bos_id = self.bos_id
        t.eos_id = self.eos_id
        return t


def build_tables(bert_tokens: Sequence[str], special_ids: Iterable[int], clip_vocab: Dict[str, int],
                 clip_merges: Sequence[Tuple[str, str]], bos_id: int, eos_id: int) -> BridgeArrays:
    V = len(bert_tokens)
    special = set(int(i) for i in special_ids)
    off = np.zeros(V + 1, dtype=np.uint32)
    chunks: List[bytes] = []
    classes: List[bytes] = []
    flags = np.zeros(V, dtype=np.uint8)
    n = 0
    for i, tok in enumerate(bert_tokens):
        f = 0
        piece = tok
        if i in special:
            f |= 1
        if tok.startswith("##"):
            f |= 2
            piece = tok[2:]
        if piece.startswith(_NOSPACE_PREFIXES):
            f |= 4
        piece = unicodedata.normalize("NFC", piece).lower()
        bs = bytearray()
        cs = bytearray()
        for ch in piece:
            enc = ch.encode("utf-8")
            k = _char_class(ch)
            bs += enc
            cs += bytes([k | 4]) + bytes([k]) * (len(enc) - 1)
        chunks.append(bytes(bs))
        classes.append(bytes(cs))
        flags[i] = f
        n += len(bs)
        off[i + 1] = n
    piece_bytes = np.frombuffer(b"".join(chunks) + b"\0", dtype=np.uint8).copy()
    piece_class = np.frombuffer(b"".join(classes) + b"\0", dtype=np.uint8).copy()
    b2u = bytes_to_unicode()
    unk = eos_id
    byte_sym = np.array([clip_vocab.get(b2u[b], unk) for b in range(256)], dtype=np.int32)
    byte_sym_eow = np.array([clip_vocab.get(b2u[b] + "</w>", unk) for b in range(256)], dtype=np.int32)
    ml, mr, mo = [], [], []
    for l, r in clip_merges:
        a, b_, o = clip_vocab.get(l), clip_vocab.get(r), clip_vocab.get(l + r)
        if a is None or b_ is None or o is None:
            a = b_ = o = -1  # keeps the rank numbering; an id of -1 never matches a symbol
        ml.append(a)
        mr.append(b_)
        mo.append(o)
    return BridgeArrays(off, piece_bytes, piece_class, flags, byte_sym, byte_sym_eow,
                        np.array(ml, dtype=np.int32), np.array(mr, dtype=np.int32), np.array(mo, dtype=np.int32),
                        V, len(clip_vocab), int(bos_id), int(eos_id))


def tables_from_tokenizers(bert_tok, clip_tok) -> BridgeArrays:
    """Works with HF tokenizers (BertTokenizer/CLIPTokenizer, slow or fast) and with the
    synthetic ones in conzic_amd.text (same attribute names)."""
    V = int(bert_tok.vocab_size)
    if hasattr(bert_tok, "convert_ids_to_tokens"):
        toks = list(bert_tok.convert_ids_to_tokens(list(range(V))))
    else:
        inv = {i: t for t, i in bert_tok.vocab.items()}
        toks = [inv[i] for i in range(V)]
    special = list(getattr(bert_tok, "all_special_ids"))
    if hasattr(clip_tok, "clip_merges"):
        vocab, merges = clip_tok.get_vocab(), list(clip_tok.clip_merges)
    else:
        model = json.loads(clip_tok.backend_tokenizer.to_str())["model"]
        vocab = model["vocab"]
        merges = [tuple(m) if not isinstance(m, str) else tuple(m.split(" ")) for m in model["merges"]]
    return build_tables(toks, special, vocab, merges, int(clip_tok.bos_token_id), int(clip_tok.eos_token_id))

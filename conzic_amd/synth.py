"""Synthetic assets for the polishing path: weights, vocabularies, images, token masks.

No pretrained checkpoints or vocabulary files exist on the build or GPU boxes (no network), so
parity and throughput are established on architecture-exact random weights and synthetic
vocabularies (SURVEY.md §7 hard part 4, §8d "Weights / vocab").  Everything here is a pure
function of integer seeds, so the container that captures the goldens and the GPU box that
replays them regenerate bit-identical inputs without shipping gigabytes.

State-dict key names and shapes follow what the reference's callers hand over
(`demo.py:125-132` -> `AutoModelForMaskedLM`, `clip/clip.py:11-16` -> `CLIPModel`), listed in
SURVEY.md §8b.

This module is shared *input* generation (not algorithm): the product harness, bench.py and
the oracle-side tests all draw their inputs from here so that they see the same bytes.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# model configurations
# --------------------------------------------------------------------------------------


@dataclass
class BertCfg:
    vocab: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 512
    eps: float = 1e-12


@dataclass
class ClipCfg:
    # text tower
    vocab: int = 49408
    hidden: int = 512
    layers: int = 12
    heads: int = 8
    inter: int = 2048
    max_pos: int = 77
    eps: float = 1e-5
    proj: int = 512
    bos_id: int = 49406
    eos_id: int = 49407
    # vision tower
    v_hidden: int = 768
    v_layers: int = 12
    v_heads: int = 12
    v_inter: int = 3072
    v_image: int = 224
    v_patch: int = 32
    logit_scale: float = 2.6592


def bert_base() -> BertCfg:
    return BertCfg()


def clip_b32() -> ClipCfg:
    return ClipCfg()


def bert_tiny(vocab: int) -> BertCfg:
    return BertCfg(vocab=vocab, hidden=128, layers=2, heads=2, inter=256, max_pos=64)


def clip_tiny(vocab: int) -> ClipCfg:
    return ClipCfg(vocab=vocab, hidden=128, layers=2, heads=2, inter=256, max_pos=77, proj=64,
                   bos_id=vocab - 2, eos_id=vocab - 1,
                   v_hidden=128, v_layers=2, v_heads=2, v_inter=256, v_image=32, v_patch=8)


# --------------------------------------------------------------------------------------
# deterministic per-tensor random streams
# --------------------------------------------------------------------------------------


def _rng(seed: int, name: str) -> np.random.Generator:
    """One independent stream per (seed, tensor name): generation order never matters."""
    return np.random.default_rng([seed & 0xFFFFFFFF, zlib.crc32(name.encode())])


def _normal(seed, name, shape, std, mean=0.0) -> np.ndarray:
    a = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    a *= np.float32(std)
    if mean:
        a += np.float32(mean)
    return a


def _ln(seed, prefix, n, out):
    # non-trivial affine so the gamma/beta paths are really exercised
    out[prefix + ".weight"] = _normal(seed, prefix + ".weight", (n,), 0.1, 1.0)
    out[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n,), 0.05)


def _lin(seed, prefix, n_out, n_in, std, out, bias=True, bias_std=0.02):
    out[prefix + ".weight"] = _normal(seed, prefix + ".weight", (n_out, n_in), std)
    if bias:
        out[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n_out,), bias_std)


def make_bert_weights(cfg: BertCfg, seed: int = 1) -> Dict[str, np.ndarray]:
    """BertForMaskedLM state dict (HF:bert/modeling_bert.py key names), fp32 numpy."""
    H, I = cfg.hidden, cfg.inter
    w: Dict[str, np.ndarray] = {}
    w["bert.embeddings.word_embeddings.weight"] = _normal(seed, "bert.word", (cfg.vocab, H), 0.05)
    w["bert.embeddings.position_embeddings.weight"] = _normal(seed, "bert.pos", (cfg.max_pos, H), 0.05)
    w["bert.embeddings.token_type_embeddings.weight"] = _normal(seed, "bert.type", (2, H), 0.05)
    _ln(seed, "bert.embeddings.LayerNorm", H, w)
    for n in range(cfg.layers):
        p = f"bert.encoder.layer.{n}"
        _lin(seed, p + ".attention.self.query", H, H, 0.06, w)
        _lin(seed, p + ".attention.self.key", H, H, 0.06, w)
        _lin(seed, p + ".attention.self.value", H, H, 0.04, w)
        _lin(seed, p + ".attention.output.dense", H, H, 0.04, w)
        _ln(seed, p + ".attention.output.LayerNorm", H, w)
        _lin(seed, p + ".intermediate.dense", I, H, 0.04, w)
        _lin(seed, p + ".output.dense", H, I, 0.03, w)
        _ln(seed, p + ".output.LayerNorm", H, w)
    _lin(seed, "cls.predictions.transform.dense", H, H, 0.04, w)
    _ln(seed, "cls.predictions.transform.LayerNorm", H, w)
    # decoder weight is tied to the word embeddings (HF:bert/modeling_bert.py:910-913)
    w["cls.predictions.decoder.weight"] = w["bert.embeddings.word_embeddings.weight"]
    w["cls.predictions.bias"] = _normal(seed, "cls.predictions.bias", (cfg.vocab,), 0.02)
    w["cls.predictions.decoder.bias"] = w["cls.predictions.bias"]
    return w


def _clip_layers(seed, prefix, n_layers, H, I, w):
    for n in range(n_layers):
        p = f"{prefix}.encoder.layers.{n}"
        for nm, std in (("q_proj", 0.06), ("k_proj", 0.06), ("v_proj", 0.04), ("out_proj", 0.03)):
            _lin(seed, f"{p}.self_attn.{nm}", H, H, std, w)
        _ln(seed, p + ".layer_norm1", H, w)
        _lin(seed, p + ".mlp.fc1", I, H, 0.04, w)
        _lin(seed, p + ".mlp.fc2", H, I, 0.03, w)
        _ln(seed, p + ".layer_norm2", H, w)


def make_clip_weights(cfg: ClipCfg, seed: int = 2) -> Dict[str, np.ndarray]:
    """CLIPModel state dict (HF:clip/modeling_clip.py key names), fp32 numpy."""
    w: Dict[str, np.ndarray] = {}
    w["logit_scale"] = np.array(cfg.logit_scale, dtype=np.float32)
    H = cfg.hidden
    w["text_model.embeddings.token_embedding.weight"] = _normal(seed, "clip.tok", (cfg.vocab, H), 0.05)
    w["text_model.embeddings.position_embedding.weight"] = _normal(seed, "clip.tpos", (cfg.max_pos, H), 0.03)
    _clip_layers(seed, "text_model", cfg.layers, H, cfg.inter, w)
    _ln(seed, "text_model.final_layer_norm", H, w)
    w["text_projection.weight"] = _normal(seed, "clip.tproj", (cfg.proj, H), H ** -0.5)
    VH = cfg.v_hidden
    npos = (cfg.v_image // cfg.v_patch) ** 2 + 1
    w["vision_model.embeddings.class_embedding"] = _normal(seed, "clip.cls", (VH,), 0.05)
    w["vision_model.embeddings.patch_embedding.weight"] = _normal(
        seed, "clip.patch", (VH, 3, cfg.v_patch, cfg.v_patch), 0.02)
    w["vision_model.embeddings.position_embedding.weight"] = _normal(seed, "clip.vpos", (npos, VH), 0.03)
    _ln(seed, "vision_model.pre_layrnorm", VH, w)  # (sic) HF spelling
    _clip_layers(seed, "vision_model", cfg.v_layers, VH, cfg.v_inter, w)
    _ln(seed, "vision_model.post_layernorm", VH, w)
    w["visual_projection.weight"] = _normal(seed, "clip.vproj", (cfg.proj, VH), VH ** -0.5)
    return w


# --------------------------------------------------------------------------------------
# synthetic images (BASELINE config 3: uint8 ~ U{0..255}, rng 1234)
# --------------------------------------------------------------------------------------

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


def make_images_u8(n: int, size: int = 224, seed: int = 1234, first: int = 0) -> np.ndarray:
    """[n, size, size, 3] uint8: images first..first+n-1 of the synthetic stream (image j has its
    own generator, so any shard of the stream can be produced independently)."""
    out = np.empty((n, size, size, 3), dtype=np.uint8)
    for j in range(n):
        out[j] = np.random.default_rng([seed, first + j]).integers(0, 256, size=(size, size, 3), dtype=np.uint8)
    return out


def make_odd_images(S):
    """Deterministic odd-sized RGB test images for the resize/crop path: smooth gradients + seeded noise, both
    orientations, down- and up-sampling, one side already S, extreme aspect."""
    sizes = [(int(S * 1.34) + 1, int(S * 2.01)), (int(S * 2.23), int(S * 1.16) + 1), (S, int(S * 1.8)), (int(S * 2.9), S),
             (int(S * 0.61), int(S * 0.83)), (S + 1, S + 1), (int(S * 4.3), int(S * 1.05)), (S, S)]
    out = []
    for i, (h, w) in enumerate(sizes):
        rng = np.random.default_rng(7000 + i)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        base = np.stack([127 + 120 * np.sin(xx / (3.0 + i) + yy / 7.0), 127 + 120 * np.cos(yy / (2.5 + i) - xx / 11.0),
                         (xx * 255.0 / max(w - 1, 1) + yy * 255.0 / max(h - 1, 1)) / 2], -1)
        noise = rng.integers(-40, 41, size=(h, w, 3))
        out.append(np.clip(base + noise, 0, 255).astype(np.uint8))
    return out


def pixels_from_u8(img_u8: np.ndarray) -> np.ndarray:
    """CLIP image processor on an already 224x224 RGB image: /255, normalise, HWC->CHW
    (clip/clip.py:55-56 with resize/crop being the identity)."""
    x = img_u8.astype(np.float32) / np.float32(255.0)  # bit-exact with the HF image processor (fp32 division)
    x = (x - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(np.moveaxis(x, -1, -3)).astype(np.float32)


# --------------------------------------------------------------------------------------
# synthetic vocabularies
# --------------------------------------------------------------------------------------

_CONS = "bcdfghjklmnpqrstvwxz"
_VOW = "aeiou"
SYLLABLES = [c + v for c in _CONS for v in _VOW]  # 100 CV syllables

_PUNCT = list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
_NONASCII = list("éüßæøñçłαβγδλπωбджяшאבגدبتあいカキ中日本語人大ﬁﬂ")


def bytes_to_unicode() -> Dict[int, str]:
    """GPT-2/CLIP byte<->printable-unicode table used by the ByteLevel pre-tokenizer
    (HF:clip/tokenization_clip.py:90-105 -> tokenizers ByteLevel)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + \
        list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


@dataclass
class SynthVocab:
    """A BERT WordPiece vocabulary plus a CLIP byte-level BPE vocabulary built so that every
    'regular' BERT word is exactly one CLIP token, while 'irregular' words, ## pieces,
    punctuation and digits split into several (they exercise the general bridge)."""
    bert_tokens: List[str]
    clip_vocab: Dict[str, int]
    clip_merges: List[Tuple[str, str]]
    regular_lo: int  # [regular_lo, regular_hi) = ids of one-CLIP-token words
    regular_hi: int
    special_ids: Dict[str, int] = field(default_factory=dict)

    @property
    def bert_vocab(self) -> Dict[str, int]:
        return {t: i for i, t in enumerate(self.bert_tokens)}


def _bpe_apply(symbols: List[str], ranks: Dict[Tuple[str, str], int]) -> List[str]:
    s = list(symbols)
    while len(s) > 1:
        best, bi = None, -1
        for i in range(len(s) - 1):
            r = ranks.get((s[i], s[i + 1]))
            if r is not None and (best is None or r < best):
                best, bi = r, i
        if best is None:
            break
        s[bi:bi + 2] = [s[bi] + s[bi + 1]]
    return s


def make_vocab(bert_size: int = 30522, clip_size: int = 49408, n_irregular: int = 600,
               n_pieces: int = 1200, n_numbers: int = 300, seed: int = 7,
               real_layout: bool = True) -> SynthVocab:
    """Build the paired vocabularies.  `real_layout` mirrors bert-base-uncased's id layout
    ([PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102, [MASK]=103, [unusedN] around them)."""
    rng = np.random.default_rng(seed)
    toks: List[str] = []
    if real_layout:
        toks.append("[PAD]")
        toks += [f"[unused{i}]" for i in range(99)]
        toks += ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
        toks += [f"[unused{i}]" for i in range(99, 994)]
    else:
        toks += ["[PAD]"] + [f"[unused{i}]" for i in range(3)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    singles = _PUNCT + list("0123456789") + list("abcdefghijklmnopqrstuvwxyz") + _NONASCII
    toks += singles
    seen = set(toks)

    def add(t):
        if t not in seen:
            seen.add(t)
            toks.append(t)
            return True
        return False

    for t in ("'s", "n't", "'m", "'ve", "'re", "...", "--"):
        add(t)
    nums = 0
    while nums < n_numbers:
        nums += add(str(int(rng.integers(10, 100000))))
    for t in ("image", "of", "the", "picture", "photo"):
        add(t)
    # irregular words: random letter strings -> several CLIP tokens each
    letters = "abcdefghijklmnopqrstuvwxyz"
    cnt = 0
    while cnt < n_irregular:
        n = int(rng.integers(2, 10))
        cnt += add("".join(letters[int(i)] for i in rng.integers(0, 26, size=n)))
    # ## continuation pieces
    for t in ("##s", "##ed", "##ing", "##ly", "##er", "##'", "##.", "##2", "##é"):
        add(t)
    cnt = 0
    while cnt < n_pieces:
        if rng.random() < 0.5:
            cnt += add("##" + SYLLABLES[int(rng.integers(0, 100))])
        else:
            n = int(rng.integers(1, 5))
            cnt += add("##" + "".join(letters[int(i)] for i in rng.integers(0, 26, size=n)))
    # regular words: 3 CV syllables, one CLIP token each; placed last
    n_regular = bert_size - len(toks)
    assert n_regular > 0, "bert_size too small for the fixed part of the vocabulary"
    regular_lo = len(toks)
    while len(toks) < bert_size:
        a, b, c = (int(i) for i in rng.integers(0, 100, size=3))
        add(SYLLABLES[a] + SYLLABLES[b] + SYLLABLES[c])
    regular_hi = len(toks)

    # ---- CLIP byte-level BPE ----
    b2u = bytes_to_unicode()
    cv: Dict[str, int] = {}
    for b in range(256):
        cv[b2u[b]] = len(cv)
    for b in range(256):
        cv[b2u[b] + "</w>"] = len(cv)
    merges: List[Tuple[str, str]] = []

    def add_merge(l, r):
        if (l, r) in mset:
            return
        mset.add((l, r))
        merges.append((l, r))
        if l + r not in cv:
            cv[l + r] = len(cv)

    mset = set()
    for s in SYLLABLES:                      # level 1: consonant+vowel
        add_merge(s[0], s[1])
    for s in SYLLABLES:                      # level 1': word-final consonant+vowel</w>
        add_merge(s[0], s[1] + "</w>")
    words = toks[regular_lo:regular_hi]
    for wd in sorted(set(w_[:4] for w_ in words)):   # level 2: syllable+syllable
        add_merge(wd[:2], wd[2:4])
    for wd in words:                          # level 3: (2 syllables)+(final syllable</w>)
        add_merge(wd[:4], wd[4:] + "</w>")
    # a handful of real words get dedicated single-token merges (prompt words)
    ranks = {m: i for i, m in enumerate(merges)}
    for wd in ("image", "of", "the", "picture", "photo"):
        sym = [b2u[ord(ch)] for ch in wd]
        sym[-1] += "</w>"
        sym = _bpe_apply(sym, ranks)
        while len(sym) > 1:
            add_merge(sym[0], sym[1])
            ranks[(sym[0], sym[1])] = len(merges) - 1
            sym[0:2] = [sym[0] + sym[1]]
    n_special = 2
    assert len(cv) + n_special <= clip_size, (len(cv), clip_size)
    k = 0
    while len(cv) + n_special < clip_size:    # filler ids no text ever produces
        cv[f"<|filler{k}|>"] = len(cv)
        k += 1
    cv["<|startoftext|>"] = len(cv)
    cv["<|endoftext|>"] = len(cv)
    sv = SynthVocab(bert_tokens=toks, clip_vocab=cv, clip_merges=merges,
                    regular_lo=regular_lo, regular_hi=regular_hi)
    bv = sv.bert_vocab
    sv.special_ids = {k_: bv[k_] for k_ in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")}
    return sv


def make_vocab_tiny(bert_size: int = 640, clip_size: int = 1600, seed: int = 7) -> SynthVocab:
    return make_vocab(bert_size=bert_size, clip_size=clip_size, n_irregular=60, n_pieces=80,
                      n_numbers=20, seed=seed, real_layout=False)


def make_token_mask(sv: SynthVocab, regular_only: bool = False) -> np.ndarray:
    """fp32 [1, V] token mask in the spirit of demo.py:135-143 with stop_words.txt: zero the
    [unusedN] range, single characters/punctuation, digit strings and [UNK].  With
    `regular_only` (throughput runs) every id outside the one-CLIP-token word range is zeroed as
    well, which gives Tc = T exactly."""
    V = len(sv.bert_tokens)
    m = np.ones((1, V), dtype=np.float32)
    for i, t in enumerate(sv.bert_tokens):
        if t.startswith("[unused") or t == "[UNK]" or t == "..." or t.isdigit() or len(t) == 1:
            m[0, i] = 0.0
    if regular_only:
        m[0, :sv.regular_lo] = 0.0
        m[0, sv.regular_hi:] = 0.0
    return m


def make_lexicon(V: int, seed: int = 99) -> np.ndarray:
    """Deterministic per-token sentiment score in [-1, 1] (BASELINE config 5 stand-in for the
    SentiWordNet lookup of sentiments_classifer.py:9-33; 'parity unpinned' for the values)."""
    return np.random.default_rng(seed).uniform(-1.0, 1.0, size=V).astype(np.float32)


UNIVERSAL_TAGS = ["ADJ", "ADP", "ADV", "CONJ", "DET", "NOUN", "NUM", "PRT", "PRON", "VERB", ".", "X"]


def make_pos_tags(V: int, seed: int = 77) -> np.ndarray:
    """Deterministic per-token universal-tagset id (stand-in for nltk.pos_tag(tagset='universal') of
    POS_classifier.py:13-14; 'parity unpinned' for the values, like the sentiment lexicon)."""
    return np.random.default_rng(seed).integers(0, len(UNIVERSAL_TAGS), size=V).astype(np.uint8)


def pos_template_masks(template) -> np.ndarray:
    """Template (list of lists of tag names, "" = wildcard, as demo.py:40-45) -> uint16 bit masks: bit t = universal tag
    t accepted, 0xFFFF = the "" wildcard.  POS_classifier.py:25 tests `cur_tag in template[w]`: a list entry is a
    membership test, a plain STRING entry a substring test -- the same thing for the twelve tag names (none contains
    another), except that the "" tag a too-short sentence is padded with (POS_classifier.py:19-20) is a substring of any
    string: bit 15 = "the padding tag matches too" carries that."""
    out = []
    for entry in template:
        if entry == "" or entry == [""]:
            out.append(0xFFFF)
            continue
        names = [entry] if isinstance(entry, str) else list(entry)
        m = 0x8000 if isinstance(entry, str) or "" in names else 0
        for n in names:
            if n in UNIVERSAL_TAGS:
                m |= 1 << UNIVERSAL_TAGS.index(n)
        out.append(m)
    return np.array(out, dtype=np.uint16)


def cfg_dict(cfg) -> dict:
    return asdict(cfg)

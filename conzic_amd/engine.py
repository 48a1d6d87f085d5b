"""Python handle on one native engine (one GPU): thin, typed calls into libconzic_hip.so.

Everything numeric happens in the shared library; this class only marshals numpy buffers (or
device pointers of torch tensors) across the C ABI of include/conzic_hip.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence

import numpy as np

from . import native
from .bridge import BridgeArrays
from .native import NativeError  # noqa: F401  (re-export)


def _ptr(a):
    """host numpy array or a device tensor (anything with .data_ptr()) -> void*"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(type(a))


def make_config(bert_cfg, clip_cfg, special: Dict[str, int], precision: int) -> native.Config:
    c = native.Config()
    if bert_cfg is not None:
        c.bert_vocab, c.bert_hidden, c.bert_layers = bert_cfg.vocab, bert_cfg.hidden, bert_cfg.layers
        c.bert_heads, c.bert_inter, c.bert_max_pos, c.bert_eps = bert_cfg.heads, bert_cfg.inter, bert_cfg.max_pos, bert_cfg.eps
    else:
        c.bert_vocab = c.bert_layers = 0
        c.bert_hidden, c.bert_heads, c.bert_inter, c.bert_max_pos, c.bert_eps = 64, 1, 64, 1, 1e-12
    c.clip_vocab, c.clip_hidden, c.clip_layers, c.clip_heads = clip_cfg.vocab, clip_cfg.hidden, clip_cfg.layers, clip_cfg.heads
    c.clip_inter, c.clip_max_pos, c.clip_proj, c.clip_eps = clip_cfg.inter, clip_cfg.max_pos, clip_cfg.proj, clip_cfg.eps
    c.clip_bos_id, c.clip_eos_id = clip_cfg.bos_id, clip_cfg.eos_id
    c.vis_hidden, c.vis_layers, c.vis_heads, c.vis_inter = clip_cfg.v_hidden, clip_cfg.v_layers, clip_cfg.v_heads, clip_cfg.v_inter
    c.vis_image, c.vis_patch = clip_cfg.v_image, clip_cfg.v_patch
    c.pad_id = special.get("[PAD]", 0)
    c.unk_id = special.get("[UNK]", 0)
    c.cls_id = special.get("[CLS]", 0)
    c.sep_id = special.get("[SEP]", 0)
    c.mask_id = special.get("[MASK]", 0)
    c.dot_id = special.get(".", 0)
    c.precision = precision
    return c


def normalize_state_name(name: str) -> Optional[str]:
    """Checkpoint key -> the name the engine knows, or None for tensors the path never reads.
    `from_pretrained` renames the legacy TF-style `LayerNorm.gamma` / `LayerNorm.beta` of old BERT checkpoints to
    `.weight` / `.bias` on load; a safetensors file read directly still carries the old names.  The pooler and the
    next-sentence head are not on the masked-LM path (HF:bert/modeling_bert.py:939-982); `position_ids` is a
    buffer; the MLM decoder weight is tied to the word embeddings (HF:bert/modeling_bert.py:910-913)."""
    if name.endswith(".gamma"):
        name = name[:-len(".gamma")] + ".weight"
    elif name.endswith(".beta"):
        name = name[:-len(".beta")] + ".bias"
    if name.endswith("position_ids") or name == "cls.predictions.decoder.weight":
        return None
    if name.startswith("bert.pooler.") or name.startswith("cls.seq_relationship."):
        return None
    return name


class Engine:
    def __init__(self, bert_cfg, clip_cfg, special: Dict[str, int], precision: int = native.PREC_BF16, device: int = 0):
        self.lib = native.load()
        self.cfg = make_config(bert_cfg, clip_cfg, special, precision)
        self.bert_cfg, self.clip_cfg = bert_cfg, clip_cfg
        self.precision = precision
        h = C.c_void_p()
        native.check(self.lib.czc_create(C.byref(self.cfg), device, C.byref(h)), None, "czc_create")
        self.h = h
        self._keep = []  # host arrays the engine may still point at
        self._replay = {}    # last call of every per-engine setter: replayed on replicas (they share only the weights)
        self._replicas = []
        self._parent = None

    # ---- lifecycle --------------------------------------------------------------------------
    def close(self):
        g = getattr(self, "_group", None)  # runtime.py caches its EngineGroup here
        if g is not None and g._pool is not None:
            g._pool.shutdown(wait=True)
            g._pool = None
        self._group = None
        for r in getattr(self, "_replicas", []):
            r.close()  # replicas point at this engine's weights: they go first
        self._replicas = []
        if getattr(self, "h", None):
            self.lib.czc_destroy(self.h)
            self.h = None

    def replica(self) -> "Engine":
        """A second engine on the same GPU over the SAME weights (czc_replicate): own stream, workspace, image
        embeddings and tables.  Every setter call made on this engine so far is replayed on it, later ones are
        forwarded.  Closed with (before) its parent."""
        if self._parent is not None:
            parent = self._parent()
            if parent is None or parent.h is None:
                raise NativeError("replica(): the parent engine of this replica was closed (it owns the weights)")
            return parent.replica()
        if self.h is None:
            raise NativeError("replica(): this engine is closed")
        r = object.__new__(Engine)
        r.lib, r.cfg, r.bert_cfg, r.clip_cfg, r.precision = self.lib, self.cfg, self.bert_cfg, self.clip_cfg, self.precision
        import weakref
        r._keep, r._replay, r._replicas, r._parent = [], {}, [], weakref.ref(self)  # no parent <-> replica cycle: both have __del__
        h = C.c_void_p()
        self._ck(self.lib.czc_replicate(self.h, C.byref(h)), "czc_replicate")
        r.h = h
        for name, (fn, args) in list(self._replay.items()):
            getattr(r, fn)(*args)
        self._replicas.append(r)
        return r

    def _record(self, key, fn, *args):
        self._replay[key] = (fn, args)
        for r in self._replicas:
            getattr(r, fn)(*args)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0 and getattr(self, "h", None) is None:
            raise NativeError(f"{what}: this engine is closed" + (" (a replica is closed with its parent, which owns the weights)"
                                                                  if self._parent is not None else ""))
        native.check(rc, self.h, what)

    # ---- frozen state -------------------------------------------------------------------------
    def load_state(self, *state_dicts, names: Optional[Iterable[str]] = None):
        """Each state dict maps HF names to fp32 arrays: numpy, or torch tensors (CPU or on this
        GPU -- device tensors are consumed in place, which is the RCCL-broadcast hand-over)."""
        for sd in state_dicts:
            for name, t in sd.items():
                if names is not None and name not in names:
                    continue
                name = normalize_state_name(name)
                if name is None:
                    continue
                if name == "cls.predictions.decoder.bias" and "cls.predictions.bias" in sd:
                    continue
                if hasattr(t, "detach"):
                    t = t.detach()
                    if t.dtype is not __import__("torch").float32:
                        t = t.float()
                    t = t.contiguous()
                    shape = tuple(t.shape)
                    src = t.data_ptr()
                    self._tmp = t
                else:
                    t = np.ascontiguousarray(t, dtype=np.float32)
                    shape = t.shape
                    src = t.ctypes.data
                    self._tmp = t
                sh = (C.c_int64 * max(1, len(shape)))(*shape)
                self._ck(self.lib.czc_load_tensor(self.h, name.encode(), 0, len(shape), sh, src), f"czc_load_tensor({name})")
        self._tmp = None

    def finalize(self):
        self._ck(self.lib.czc_finalize_weights(self.h), "czc_finalize_weights")

    def set_token_mask(self, mask):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.float32).reshape(-1))
        self._ck(self.lib.czc_set_token_mask(self.h, m.ctypes.data, m.size), "czc_set_token_mask")
        self._record("token_mask", "set_token_mask", m)

    def set_lexicon_pos(self, table, class_of_token):
        """(word-start piece, coarse POS class) keyed sentiment table [V,5] + class per token [V] (None: back to
        the per-token lexicon).  See conzic_amd/sentiment.py."""
        if table is None:
            self._ck(self.lib.czc_set_lexicon_pos(self.h, None, None, 0), "czc_set_lexicon_pos")
            self._record("lexicon_pos", "set_lexicon_pos", None, None)
            return
        t = np.ascontiguousarray(table, dtype=np.float32)
        c = np.ascontiguousarray(class_of_token, dtype=np.uint8)
        assert t.ndim == 2 and t.shape[1] == 5 and c.shape == (t.shape[0],)
        self._ck(self.lib.czc_set_lexicon_pos(self.h, t.ctypes.data, c.ctypes.data, t.shape[0]), "czc_set_lexicon_pos")
        self._record("lexicon_pos", "set_lexicon_pos", t, c)

    def set_lexicon(self, lex):
        m = np.ascontiguousarray(np.asarray(lex, dtype=np.float32).reshape(-1))
        self._ck(self.lib.czc_set_lexicon(self.h, m.ctypes.data, m.size), "czc_set_lexicon")
        self._record("lexicon", "set_lexicon", m)

    def set_pos(self, tag_of_token, template_masks):
        t = np.ascontiguousarray(np.asarray(tag_of_token, dtype=np.uint8).reshape(-1))
        m = np.ascontiguousarray(np.asarray(template_masks, dtype=np.uint16).reshape(-1))
        self._ck(self.lib.czc_set_pos(self.h, t.ctypes.data, t.size, m.ctypes.data, m.size), "czc_set_pos")
        self._record("pos", "set_pos", t, m)

    def set_control_callback(self, scorer):
        """czc_set_control_callback: `scorer(inp int32 [B,T], cand int32 [B,K], gen_idx) -> fp32 [B,K]` is called once per
        controlled step and replaces the control-score tables (conzic_amd/control.py: the reference's own nltk scorer on
        the decoded strings); None removes it.  An exception raised by the scorer fails the step and is re-raised by the
        generate / step call that triggered it."""
        if scorer is None:
            self._ck(self.lib.czc_set_control_callback(self.h, None, None), "czc_set_control_callback")
            self._ctl_fn = self._ctl_scorer = None
        else:
            def trampoline(_user, inp_p, cand_p, B, T, K, gen_idx, out_p):
                try:
                    inp = np.ctypeslib.as_array(inp_p, shape=(B, T)).copy()
                    cand = np.ctypeslib.as_array(cand_p, shape=(B, K)).copy()
                    sc = np.ascontiguousarray(scorer(inp, cand, int(gen_idx)), dtype=np.float32)
                    if sc.shape != (B, K):
                        raise ValueError(f"control scorer returned shape {sc.shape}, expected {(B, K)}")
                    np.ctypeslib.as_array(out_p, shape=(B, K))[...] = sc
                    return 0
                except BaseException as exc:  # noqa: BLE001 -- must not unwind through the C frames
                    self._ctl_error = exc
                    return 1
            fn = native.CONTROL_FN(trampoline)
            self._ck(self.lib.czc_set_control_callback(self.h, C.cast(fn, C.c_void_p), None), "czc_set_control_callback")
            self._ctl_fn, self._ctl_scorer = fn, scorer  # the C side keeps the pointer: keep the thunk alive
        self._ctl_error = None
        self._record("control_callback", "set_control_callback", scorer)

    def _raise_scorer_error(self):
        exc, self._ctl_error = getattr(self, "_ctl_error", None), None
        if exc is not None:
            raise exc

    def set_bridge(self, tables: BridgeArrays):
        st = tables.as_struct()
        self._ck(self.lib.czc_set_bridge(self.h, C.byref(st)), "czc_set_bridge")
        self._record("bridge", "set_bridge", tables)

    # ---- images / text ------------------------------------------------------------------------
    def encode_images(self, pixels) -> np.ndarray:
        if isinstance(pixels, np.ndarray):
            pixels = np.ascontiguousarray(pixels, dtype=np.float32)
        B = int(pixels.shape[0])
        out = np.empty((B, self.clip_cfg.proj), dtype=np.float32)
        self._ck(self.lib.czc_encode_images(self.h, _ptr(pixels), B, out.ctypes.data), "czc_encode_images")
        return out

    def preprocess_u8(self, rgb, slot: int = 0, want_pixels: bool = False, mean=None, std=None):
        """CLIPProcessor geometry on the device (czc_preprocess_u8): rgb uint8 [H,W,3] -> staged slot `slot`;
        returns the fp32 [3,S,S] pixels when want_pixels."""
        from .synth import CLIP_MEAN, CLIP_STD
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("rgb must be uint8 [H, W, 3]")
        m = np.ascontiguousarray(CLIP_MEAN if mean is None else mean, dtype=np.float32)
        d = np.ascontiguousarray(CLIP_STD if std is None else std, dtype=np.float32)
        S = self.clip_cfg.v_image
        out = np.empty((3, S, S), dtype=np.float32) if want_pixels else None
        self._ck(self.lib.czc_preprocess_u8(self.h, rgb.ctypes.data, rgb.shape[0], rgb.shape[1], m.ctypes.data, d.ctypes.data,
                                            int(slot), out.ctypes.data if want_pixels else None), "czc_preprocess_u8")
        return out

    def encode_staged(self, B: int) -> np.ndarray:
        out = np.empty((int(B), self.clip_cfg.proj), dtype=np.float32)
        self._ck(self.lib.czc_encode_staged(self.h, int(B), out.ctypes.data), "czc_encode_staged")
        return out

    def encode_pil(self, images, mean=None, std=None) -> np.ndarray:
        """PIL images / uint8 arrays -> image_embeds, geometry and normalisation on the device."""
        if not isinstance(images, (list, tuple)):
            images = [images]
        for i, im in enumerate(images):
            if not isinstance(im, np.ndarray):
                im = np.asarray(im.convert("RGB"))
            self.preprocess_u8(im, i, mean=mean, std=std)
        return self.encode_staged(len(images))

    def set_image_embeds(self, embeds):
        e = np.ascontiguousarray(embeds, dtype=np.float32)
        self._ck(self.lib.czc_set_image_embeds(self.h, e.ctypes.data, e.shape[0]), "czc_set_image_embeds")

    def encode_text(self, clip_ids: np.ndarray, clip_len: np.ndarray) -> np.ndarray:
        n = int(clip_ids.shape[0])
        ids = np.full((n, native.CLIP_MAX_LEN), self.clip_cfg.eos_id, dtype=np.int32)
        ids[:, : clip_ids.shape[1]] = clip_ids
        ln = np.ascontiguousarray(clip_len, dtype=np.int32)
        out = np.empty((n, self.clip_cfg.proj), dtype=np.float32)
        self._ck(self.lib.czc_encode_text(self.h, ids.ctypes.data, ln.ctypes.data, n, out.ctypes.data), "czc_encode_text")
        return out

    # ---- hot path --------------------------------------------------------------------------------
    @staticmethod
    def hyper(alpha, beta, temperature, gamma=None, negative=False, control=None) -> native.Hyper:
        """control: None -> 'sentiment' when gamma is given (back-compat), else 'sentiment' | 'pos'."""
        h = native.Hyper()
        h.alpha, h.beta = float(alpha), float(beta)
        h.gamma = float(gamma) if gamma is not None else 0.0
        h.temperature = 1.0 if temperature is None else float(temperature)
        if gamma is None:
            h.control = 0
        else:
            h.control = 2 if control == "pos" else 1
        h.negative = 1 if negative else 0
        return h

    def step(self, inp: np.ndarray, gen_idx: int, top_k: int, hyper: native.Hyper, n_mask: int = 1,
             dot_allowed: bool = False, want: Sequence[str] = ("probs", "idxs", "cand_ids", "clip_ids", "clip_len",
                                                               "clip_score", "clip_ref", "final_score", "best",
                                                               "best_cos")) -> Dict[str, np.ndarray]:
        """One position-step on int32 `inp` [B,T] (updated in place).  Returns the requested tensors."""
        assert inp.dtype == np.int32 and inp.flags.c_contiguous
        B, T = inp.shape
        K = top_k
        V = self.cfg.bert_vocab
        shapes = dict(probs=((B, K), np.float32), idxs=((B, K), np.int32), cand_ids=((B, K), np.int32),
                      clip_ids=((B * K, native.CLIP_MAX_LEN), np.int32), clip_len=((B * K,), np.int32),
                      clip_score=((B, K), np.float32), clip_ref=((B, K), np.float32), senti_raw=((B, K), np.float32),
                      repeats=((B, K), np.float32), final_score=((B, K), np.float32), best=((B,), np.int32),
                      best_cos=((B,), np.float32), logits=((B, V), np.float32))
        out = native.StepOut()
        res = {}
        for name in want:
            shp, dt = shapes[name]
            res[name] = np.empty(shp, dtype=dt)
            setattr(out, name, res[name].ctypes.data)
        rc = self.lib.czc_step(self.h, inp.ctypes.data, B, T, gen_idx, n_mask, 1 if dot_allowed else 0, K,
                               C.byref(hyper), C.byref(out))
        if rc:
            self._raise_scorer_error()
        self._ck(rc, "czc_step")
        return res

    def generate(self, B: int, init_ids: Sequence[int], L: int, seed_len: int, top_k: int, positions: Sequence[int],
                 hyper: native.Hyper, n_mask: Optional[Sequence[int]] = None, snapshot_every: Optional[int] = None,
                 want_cos: bool = True):
        """Whole *_generation call.  Returns (ids int32 [S,B,T], cos fp32 [S,B]) per snapshot (cos None with want_cos=False:
        the C ABI's out_cos == NULL)."""
        init = np.ascontiguousarray(init_ids, dtype=np.int32)
        T = init.size
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        nm = None if n_mask is None else np.ascontiguousarray(n_mask, dtype=np.int32)
        every = snapshot_every or L
        S = len(pos) // every
        ids = np.empty((S, B, T), dtype=np.int32)
        cos = np.empty((S, B), dtype=np.float32) if want_cos else None
        rc = self.lib.czc_generate(self.h, B, T, L, seed_len, init.ctypes.data, top_k, len(pos), pos.ctypes.data,
                                   None if nm is None else nm.ctypes.data, every, C.byref(hyper),
                                   ids.ctypes.data, None if cos is None else cos.ctypes.data)
        if rc:
            self._raise_scorer_error()
        self._ck(rc, "czc_generate")
        return ids, cos

    def similarity(self, image_embeds, text_embeds, K: int):
        """clip/clip.py:86-98: (softmax_K(cos * exp(logit_scale)), cos), both [B, K], from un-normalised embeddings."""
        ie = np.ascontiguousarray(image_embeds, np.float32)
        te = np.ascontiguousarray(text_embeds, np.float32)
        B = ie.shape[0]
        assert te.shape == (B * K, ie.shape[1]), (te.shape, ie.shape, K)
        cs, cr = np.empty((B, K), np.float32), np.empty((B, K), np.float32)
        self._ck(self.lib.czc_similarity(self.h, ie.ctypes.data, te.ctypes.data, B, K, cs.ctypes.data, cr.ctypes.data), "czc_similarity")
        return cs, cr

    def set_option(self, name: str, value: int):
        self._ck(self.lib.czc_set_option(self.h, name.encode(), int(value)), f"czc_set_option({name})")
        self._record("option:" + name, "set_option", name, int(value))

    def get_option(self, name: str) -> int:
        """An option as the engine holds it (czc_get_option; also the derived read-only values, e.g. the guard's trip point in
        force inside czc_generate: "refine_guard_generate_x1e6")."""
        v = C.c_int()
        self._ck(self.lib.czc_get_option(self.h, name.encode(), C.byref(v)), f"czc_get_option({name})")
        return int(v.value)

    # ---- measurement ------------------------------------------------------------------------------
    def profile(self, on):
        """False/0 off, True/1 every kernel class, 2 only the CLIP-text linear layers (cheap: the roofline family)."""
        self._ck(self.lib.czc_profile_enable(self.h, int(on)), "czc_profile_enable")

    def profile_reset(self):
        self._ck(self.lib.czc_profile_reset(self.h), "czc_profile_reset")

    def profile_get(self, kind: str):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self.lib.czc_profile_get(self.h, kind.encode(), C.byref(ms), C.byref(n), C.byref(fl)), "czc_profile_get")
        return dict(ms=ms.value, launches=n.value, flops=fl.value)

    def stats(self):
        a, b, c_, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.czc_stats(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)), "czc_stats")
        rs, rr = C.c_int64(), C.c_int64()
        self._ck(self.lib.czc_refine_stats(self.h, C.byref(rs), C.byref(rr)), "czc_refine_stats")
        g, gi = C.c_int64(), C.c_int64()
        self._ck(self.lib.czc_refine_gate_stats(self.h, C.byref(g), C.byref(gi)), "czc_refine_gate_stats")
        dd = C.c_int64()
        self._ck(self.lib.czc_dedup_stats(self.h, C.byref(dd), None), "czc_dedup_stats")
        return dict(clip_rows=a.value, clip_seqs=b.value, bert_rows=c_.value, steps=d.value, refine_seqs=rs.value,
                    refine_rows=rr.value, gated_image_steps=g.value, gate_image_steps=gi.value, dedup_seqs=dd.value)

    def refine_guard(self, reset: bool = True):
        """(max |screening error - mean| seen on re-encoded candidates, image-steps above the trip point) of a
        CZC_PREC_REFINE engine since the last reset (czc_refine_guard)."""
        dev, trips = C.c_float(), C.c_int64()
        self._ck(self.lib.czc_refine_guard(self.h, 1 if reset else 0, C.byref(dev), C.byref(trips)), "czc_refine_guard")
        return dict(max_dev=float(dev.value), tripped=int(trips.value))

    def sync(self):
        self._ck(self.lib.czc_sync(self.h), "czc_sync")

    def profile_intervals(self, kind: str, ref: Optional["Engine"] = None) -> np.ndarray:
        """[n, 2] start / end (ms) of every launch of class `kind` since profile_reset, on the clock that starts at
        `ref`'s profile_reset (default: this engine's)."""
        ref = ref or self
        n = C.c_int(0)
        self._ck(self.lib.czc_profile_intervals(self.h, ref.h, kind.encode(), None, None, 0, C.byref(n)), "czc_profile_intervals")
        a = np.empty(max(n.value, 1), np.float64)
        b = np.empty(max(n.value, 1), np.float64)
        self._ck(self.lib.czc_profile_intervals(self.h, ref.h, kind.encode(), a.ctypes.data, b.ctypes.data, n.value,
                                                C.byref(n)), "czc_profile_intervals")
        return np.stack([a[: n.value], b[: n.value]], axis=1)


def union_ms(intervals: Sequence[np.ndarray]) -> float:
    """Total length of the union of [start, end] intervals (ms) from one or more engines on a common clock."""
    iv = np.concatenate([x for x in intervals if len(x)], axis=0) if any(len(x) for x in intervals) else np.zeros((0, 2))
    if not len(iv):
        return 0.0
    iv = iv[np.argsort(iv[:, 0])]
    total, cur_s, cur_e = 0.0, iv[0, 0], iv[0, 1]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            total += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    return float(total + (cur_e - cur_s))


class EngineGroup:
    """One engine plus replicas over the same weights, each on its own HIP stream, polishing disjoint contiguous image
    sub-batches concurrently from host threads (the C calls release the GIL).  Images are independent
    (gen_utils.py:64-81), so the captions are the single-engine ones image for image; what changes is that one
    sub-batch's small launches (BERT, top-K, LayerNorm, attention) and the tail rounds of its persistent GEMMs overlap
    the other's big launches: +5 % captions/s at 256 images with two streams.  Below `min_images` per
    stream the batch is not split."""

    def __init__(self, engine: Engine, streams: int = 2, min_images: int = 32):
        from concurrent.futures import ThreadPoolExecutor
        self.engines = [engine] + [engine.replica() for _ in range(max(1, int(streams)) - 1)]
        self.min_images = int(min_images)
        self._pool = ThreadPoolExecutor(max_workers=len(self.engines)) if len(self.engines) > 1 else None
        self._full_embeds = None

    @property
    def streams(self) -> int:
        return len(self.engines)

    def parts(self, B: int):
        """Contiguous (lo, hi) image ranges, one per engine used."""
        n = max(1, min(len(self.engines), B // max(self.min_images, 1)))
        base, rem = divmod(B, n)
        out, lo = [], 0
        for r in range(n):
            hi = lo + base + (1 if r < rem else 0)
            out.append((lo, hi))
            lo = hi
        return out

    def _run(self, jobs):
        if len(jobs) == 1 or self._pool is None:
            return [j() for j in jobs]
        futs = [self._pool.submit(j) for j in jobs]
        return [f.result() for f in futs]

    def encode_images(self, pixels) -> np.ndarray:
        B = int(pixels.shape[0])
        parts = self.parts(B)
        outs = self._run([(lambda e=e, lo=lo, hi=hi: e.encode_images(pixels[lo:hi])) for e, (lo, hi) in zip(self.engines, parts)])
        self._full_embeds = None  # every engine now holds its own slice
        self._resident = parts
        return np.concatenate(outs, axis=0)

    def set_image_embeds(self, embeds):
        self._full_embeds = np.ascontiguousarray(embeds, dtype=np.float32)
        self._resident = None

    def generate(self, B: int, init_ids, L: int, seed_len: int, top_k: int, positions, hyper, n_mask=None,
                 snapshot_every=None):
        parts = self.parts(B)
        if self._full_embeds is not None:
            for e, (lo, hi) in zip(self.engines, parts):
                e.set_image_embeds(self._full_embeds[lo:hi])
        elif getattr(self, "_resident", None) != parts:
            raise NativeError("EngineGroup.generate: encode_images / set_image_embeds of the same batch first")
        outs = self._run([(lambda e=e, lo=lo, hi=hi: e.generate(hi - lo, init_ids, L, seed_len, top_k, positions, hyper,
                                                                n_mask=n_mask, snapshot_every=snapshot_every))
                          for e, (lo, hi) in zip(self.engines, parts)])
        return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1)

    # ---- the engine calls that apply to every member ----
    def set_option(self, name, value):
        self.engines[0].set_option(name, value)  # forwarded to the replicas

    def profile(self, on):
        for e in self.engines:
            e.profile(on)

    def profile_reset(self):
        for e in self.engines:
            e.profile_reset()

    def profile_get(self, kind: str):
        """Sum over the engines; `busy_ms` is the union of their launch intervals (what the GPU spent on the class)."""
        gs = [e.profile_get(kind) for e in self.engines]
        ref = self.engines[0]
        busy = union_ms([e.profile_intervals(kind, ref) for e in self.engines])
        return dict(ms=sum(g["ms"] for g in gs), launches=sum(g["launches"] for g in gs), flops=sum(g["flops"] for g in gs),
                    busy_ms=busy)

    def stats(self):
        ss = [e.stats() for e in self.engines]
        return {k: sum(s_[k] for s_ in ss) for k in ss[0]}

    def refine_guard(self, reset: bool = True):
        gs = [e.refine_guard(reset) for e in self.engines]
        return dict(max_dev=max(g["max_dev"] for g in gs), tripped=sum(g["tripped"] for g in gs))

    def close(self, parent: bool = True):
        """Close the replicas (and the thread pool); the first engine too unless parent=False."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        head = self.engines[0]
        for r in self.engines[1:]:
            r.close()
            if r in head._replicas:
                head._replicas.remove(r)
        self.engines = [head]
        if parent:
            head.close()

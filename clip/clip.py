"""Drop-in for the reference's clip/clip.py `CLIP` wrapper (clip/clip.py:6-102): same method
names and return conventions, executed by the native engine.

    clip = CLIP("openai/clip-vit-base-patch32")          # needs `transformers` + a local checkpoint
    clip = CLIP.from_state(cfg, state_dict, tokenizer)   # synthetic / already-loaded weights

Returned tensors are torch CPU tensors when torch is importable (the reference returns torch
tensors), numpy arrays otherwise.  Index-building helpers of the reference (clip/clip.py:105-144)
belong to its retrieval baseline and are out of scope.
"""
from __future__ import annotations

import numpy as np

from conzic_amd import native, synth


def _wrap(a: np.ndarray):
    try:
        import torch
        return torch.from_numpy(np.ascontiguousarray(a))
    except ImportError:
        return a


def _fingerprint(im):
    """Content key of an image object for the one-entry embed cache: geometry, pixel type and a 128-bit hash of ALL the
    pixel bytes (blake2b: ~0.15 ms for a 224x224x3 image, two orders of magnitude below the ViT encode it saves), so a
    caller that refills the same buffer / PIL object in place never gets stale embeddings, wherever the change is.
    None for objects it does not know (those are encoded on every call, like the reference does, gen_utils.py:58)."""
    import hashlib
    try:
        if isinstance(im, np.ndarray):
            buf = im if im.flags.c_contiguous else np.ascontiguousarray(im)
            return ("nd", im.shape, str(im.dtype), hashlib.blake2b(memoryview(buf).cast("B"), digest_size=16).digest())
        if hasattr(im, "tobytes") and hasattr(im, "size") and hasattr(im, "mode"):  # PIL.Image
            w, h = im.size
            if w <= 0 or h <= 0:
                return None
            return ("pil", (w, h), im.mode, hashlib.blake2b(im.tobytes(), digest_size=16).digest())
    except Exception:
        return None
    return None


class ImageEmbeds:
    """Image embeddings computed earlier (`CLIP.compute_image_representation_from_image_instance`), handed back in
    place of the images: the north star's "ViT image encode once per image, cached" across `samples_num` passes
    (the reference re-encodes on every call, demo.py:83-85 -> gen_utils.py:58).  `embeds`: fp32 [B, proj]."""

    def __init__(self, embeds):
        self.embeds = np.ascontiguousarray(np.asarray(embeds, dtype=np.float32))


class CLIP:
    def __init__(self, model_name=None):
        self.model = None
        self.processor = None
        self.tokenizer = None
        self.czc_cfg = None
        self._state = None
        self._engine = None
        # optional caller-provided control tables (conzic_amd/control.py; unset = nltk's own scorer / tables built from nltk)
        self.lexicon = None  # fp32 [bert_vocab] per-token sentiment score
        self.lexicon_pos = None  # (fp32 [bert_vocab, 5], uint8 [bert_vocab]): score per word-start piece and coarse POS class
        self.pos_tags = None  # uint8 [bert_vocab] universal-tag ids for POS control
        self.cuda_has_been_checked = False
        if model_name is not None:
            print('Initializing CLIP model...')
            from transformers import CLIPModel, CLIPProcessor, CLIPTokenizer
            self.model = CLIPModel.from_pretrained(model_name)
            self.model.eval()
            self.processor = CLIPProcessor.from_pretrained(model_name)
            self.tokenizer = CLIPTokenizer.from_pretrained(model_name)
            print('CLIP model initialized.')

    @classmethod
    def from_state(cls, cfg: synth.ClipCfg, state_dict, tokenizer):
        self = cls(None)
        self.czc_cfg = cfg
        self._state = state_dict
        self.tokenizer = tokenizer
        return self

    # nn.Module-ish no-ops the reference's callers use (demo.py:128-132)
    def eval(self):
        return self

    def to(self, device):
        return self

    def clip_state_dict(self):
        return self._state if self._state is not None else self.model.state_dict()

    def _eng(self):
        if self._engine is None:
            from conzic_amd.runtime import clip_only_engine
            self._engine = clip_only_engine(self)
        return self._engine

    def _image_size(self):
        if self.czc_cfg is not None:
            return self.czc_cfg.v_image
        return self.model.config.vision_config.image_size

    def _device_processor_params(self):
        """(mean, std) when the HF image processor is the geometry czc_preprocess_u8 implements (RGB, bicubic
        resize of the shorter side to S, centre crop SxS, 1/255 rescale, normalise), else None."""
        ip = getattr(self.processor, "image_processor", None)
        if ip is None:
            return None
        S = self._image_size()
        try:
            def field(obj, key):  # plain dict (older transformers) or SizeDict
                return obj.get(key) if isinstance(obj, dict) else getattr(obj, key, None)
            size, crop = getattr(ip, "size", None), getattr(ip, "crop_size", None)
            ok = (field(size, "shortest_edge") == S and field(crop, "height") == S and field(crop, "width") == S
                  and int(getattr(ip, "resample", -1)) == 3
                  and all(bool(getattr(ip, k, False)) for k in ("do_resize", "do_center_crop", "do_rescale", "do_normalize"))
                  and abs(float(getattr(ip, "rescale_factor", 0.0)) - 1.0 / 255.0) < 1e-12)
            if not ok:
                return None
            return (np.asarray(ip.image_mean, np.float32), np.asarray(ip.image_std, np.float32))
        except (TypeError, ValueError, AttributeError):
            return None

    # ---- clip/clip.py:48-62 ------------------------------------------------------------------
    def compute_image_representation_from_image_instance(self, image):
        if isinstance(image, ImageEmbeds):  # cached from an earlier sample: no second pass through the ViT
            self._eng().set_image_embeds(image.embeds)
            return _wrap(image.embeds)
        # one-entry cache on the identity of the image objects: the same PIL image(s) polished again
        # (demo.py:83 loops samples_num times over one image) are encoded once
        imgs = image if isinstance(image, (list, tuple)) else [image]
        # identity alone would serve stale embeddings to a caller that refills the same buffer / PIL object in place:
        # the key also carries a content fingerprint (size, mode / dtype and a hash of every pixel byte); objects that
        # cannot be fingerprinted are never cached
        prints = [_fingerprint(im) for im in imgs]
        key = None if any(fp is None for fp in prints) else tuple((id(im), fp) for im, fp in zip(imgs, prints))
        cached = getattr(self, "_img_cache", None)
        if key is None:
            self._img_cache = None
            emb = self._encode_images_uncached(image)
            self._last_embeds = np.ascontiguousarray(np.asarray(emb, dtype=np.float32))
            return emb
        if cached is not None and cached[0] == key and all(a is b for a, b in zip(cached[1], imgs)):
            self._eng().set_image_embeds(cached[2])
            self._last_embeds = cached[2]
            return _wrap(cached[2])
        emb = self._encode_images_uncached(image)
        self._img_cache = (key, list(imgs), np.ascontiguousarray(np.asarray(emb, dtype=np.float32)))
        self._last_embeds = self._img_cache[2]
        return emb

    def last_image_embeds(self):
        """fp32 [B, proj] of the most recent image encode (to be handed back as `ImageEmbeds` on later samples)."""
        last = getattr(self, "_last_embeds", None)
        return None if last is None else last.copy()

    def _encode_images_uncached(self, image):
        if self.processor is not None:
            std_cfg = self._device_processor_params()
            if std_cfg is not None:  # the checkpoint's processor is the standard CLIP one: run it on the device
                return _wrap(self._eng().encode_pil(image, mean=std_cfg[0], std=std_cfg[1]))
            pixels = self.processor(images=image, return_tensors="np")['pixel_values'].astype(np.float32)
            return _wrap(self._eng().encode_images(pixels))
        # resize / crop / normalise on the device, bit-identical to the PIL image processor (czc_preprocess_u8);
        # (oracle/imageproc.py is the host statement of the same thing, used by the tests only)
        return _wrap(self._eng().encode_pil(image))

    def compute_image_representation_from_image_path(self, image_path):
        from PIL import Image
        return self.compute_image_representation_from_image_instance(Image.open(image_path))

    # ---- clip/clip.py:64-84 ------------------------------------------------------------------
    def compute_text_representation(self, text_list):
        enc = self.tokenizer(text_list, padding=True, max_length=self.tokenizer.max_len_single_sentence + 2,
                             truncation=True)
        ids = np.asarray(enc['input_ids'], dtype=np.int32)
        lens = np.asarray(enc['attention_mask'], dtype=np.int32).sum(1).astype(np.int32)
        return _wrap(self._eng().encode_text(ids, lens))

    # ---- clip/clip.py:86-98 ------------------------------------------------------------------
    def compute_image_text_similarity_via_embeddings(self, image_embeds, text_embeds):
        """-> (softmax over the text list of cos*exp(logit_scale), cos), both [batch, len(text_list)]"""
        ie = np.asarray(image_embeds, dtype=np.float32)
        te = np.asarray(text_embeds, dtype=np.float32).reshape(ie.shape[0], -1, ie.shape[1])
        B, K, D = te.shape
        cs, cr = self._eng().similarity(ie, te.reshape(B * K, D), K)   # czc_similarity: the product library, not a test hook
        return _wrap(cs), _wrap(cr)

    def compute_image_text_similarity_via_raw_text(self, image_embeds, text_list):
        text_embeds = self.compute_text_representation(text_list)
        return self.compute_image_text_similarity_via_embeddings(image_embeds, text_embeds)

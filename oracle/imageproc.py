"""TEST INFRASTRUCTURE (oracle): CLIP image pre-processing on the host through PIL itself
(clip/clip.py:55-56 -> CLIPProcessor -> HF CLIPImageProcessor, PIL backend): RGB, resize the shorter side to S
with bicubic resampling, centre crop SxS, /255, normalise, CHW.  The product path is czc_preprocess_u8
(conzic_amd/csrc/imageproc.hip); this is what it is checked against, bit for bit."""
from __future__ import annotations

import numpy as np

from conzic_amd.synth import CLIP_MEAN, CLIP_STD


def preprocess(images, size: int = 224) -> np.ndarray:
    from PIL import Image
    if not isinstance(images, (list, tuple)):
        images = [images]
    out = np.empty((len(images), 3, size, size), dtype=np.float32)
    for i, im in enumerate(images):
        if isinstance(im, np.ndarray):
            im = Image.fromarray(im)
        im = im.convert("RGB")
        w, h = im.size
        if (w, h) != (size, size):
            if w <= h:
                nw, nh = size, int(size * h / w)
            else:
                nw, nh = int(size * w / h), size
            im = im.resize((nw, nh), resample=Image.BICUBIC)
            left, top = (nw - size) // 2, (nh - size) // 2
            im = im.crop((left, top, left + size, top + size))
        x = np.asarray(im, dtype=np.float32) / np.float32(255.0)
        x = (x - CLIP_MEAN) / CLIP_STD
        out[i] = np.moveaxis(x, -1, 0)
    return out

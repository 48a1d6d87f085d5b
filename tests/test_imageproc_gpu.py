"""-m gpu: the device image processor (czc_preprocess_u8: Pillow-exact bicubic resize, centre crop, /255, normalise)
against the reference processor's captured output and against PIL itself on more sizes -- bit-exact (integer
resampling, three IEEE fp32 operations per value)."""
import os

import numpy as np
import pytest

from conzic_amd import harness, native, synth
from goldutil import GOLD
from oracle.imageproc import preprocess as pil_preprocess

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setups():
    return {32: harness.build_synthetic(True, native.PREC_F32), 224: harness.build_synthetic(False, native.PREC_BF16)}


@pytest.mark.parametrize("label,S", [("tiny", 32), ("full", 224)])
def test_device_processor_matches_reference_goldens(setups, label, S):
    eng = setups[S].engine
    z = np.load(os.path.join(GOLD, f"imageproc_{label}.npz"))
    imgs = synth.make_odd_images(S)[: int(z["n"])]
    want = synth.pixels_from_u8(z["crops"])
    for i, im in enumerate(imgs):
        got = eng.preprocess_u8(im, slot=i, want_pixels=True)
        np.testing.assert_array_equal(got, want[i], err_msg=f"image {i} {im.shape}")


@pytest.mark.parametrize("hw", [(224, 224), (225, 224), (224, 225), (1000, 700), (333, 1279), (60, 45), (17, 400),
                                (2048, 1536), (223, 223), (448, 448)])
def test_device_processor_matches_pil(setups, hw):
    """Pillow's resampler is the third-party algorithm under the reference's processor: more geometries than the
    committed goldens hold -- heavy down-sampling (wide filter windows), up-sampling, one pass skipped, exact size."""
    eng = setups[224].engine
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    im = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    # blocky structure as well as noise, so that window placement errors show up as large differences
    im[: hw[0] // 2, : hw[1] // 3] //= 4
    got = eng.preprocess_u8(im, slot=0, want_pixels=True)
    np.testing.assert_array_equal(got, pil_preprocess([im], 224)[0])


def test_encode_pil_equals_encode_of_host_pixels(setups):
    su = setups[32]
    imgs = synth.make_odd_images(32)[:5]
    a = su.engine.encode_pil(imgs)
    b = su.engine.encode_images(pil_preprocess(imgs, 32))
    np.testing.assert_array_equal(a, b)
    # staged slots survive growth of the staging buffer
    for i in range(20):
        su.engine.preprocess_u8(imgs[i % 5], slot=i)
    c = su.engine.encode_staged(20)
    np.testing.assert_array_equal(c[:5], a)
    np.testing.assert_array_equal(c[15:20], a)


def test_bad_arguments_are_errors(setups):
    eng = setups[32].engine
    with pytest.raises(ValueError):
        eng.preprocess_u8(np.zeros((4, 4), np.uint8))
    with pytest.raises(native.NativeError):
        eng.encode_staged(10_000)


def test_dropin_clip_with_hf_processor_uses_device_path_bit_exactly(setups):
    """A drop-in `CLIP` that carries the checkpoint's own HF image processor (the real-checkpoint route) sends raw
    RGB to czc_preprocess_u8; the embeddings equal those of HF's pixel_values pushed through czc_encode_images."""
    import types
    from PIL import Image
    from clip.clip import CLIP
    try:
        from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    except ImportError:
        pytest.skip("transformers PIL image processor unavailable")
    su = setups[32]
    S = su.clip_cfg.v_image
    c = CLIP(None)
    c.czc_cfg = su.clip_cfg
    c._engine = su.engine
    ip = CLIPImageProcessorPil(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    c.processor = types.SimpleNamespace(image_processor=ip)
    imgs = [Image.fromarray(u) for u in synth.make_odd_images(S)[:5]]
    got = np.asarray(c.compute_image_representation_from_image_instance(imgs))
    pv = ip(images=imgs, return_tensors="np")["pixel_values"].astype(np.float32)
    want = su.engine.encode_images(pv)
    np.testing.assert_array_equal(got, want)

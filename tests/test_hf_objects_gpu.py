"""-m gpu: the drop-in boundary driven with REAL Hugging Face objects, the way demo.py:125-143 builds them:
`BertForMaskedLM` / `BertTokenizer` instances and a `CLIPModel` + `CLIPProcessor` + `CLIPTokenizer` directory opened by
`clip.clip.CLIP(path)`.  Weights are the goldens' synthetic ones loaded into architecture-exact HF modules, so the
captions must be the ones the reference produced (tests/golden/tiny_*.npz) -- which pins, beyond the engine itself,
the state-dict names the engine pulls from HF modules, the bridge tables built from HF tokenizer objects and the HF
image processor being recognised and run on the device."""
import logging

import numpy as np
import pytest

from conzic_amd import synth
from goldutil import load_case

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def _hf_objects(meta, tmp):
    import torch
    from transformers import BertConfig, BertForMaskedLM, BertTokenizer, CLIPConfig, CLIPModel, CLIPProcessor, CLIPTokenizer
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    from clip.clip import CLIP
    sv = synth.make_vocab_tiny()
    bcfg, ccfg = synth.BertCfg(**meta["bert_cfg"]), synth.ClipCfg(**meta["clip_cfg"])
    bt = BertTokenizer(vocab=sv.bert_vocab)
    lm = BertForMaskedLM(BertConfig(vocab_size=bcfg.vocab, hidden_size=bcfg.hidden, num_hidden_layers=bcfg.layers,
                                    num_attention_heads=bcfg.heads, intermediate_size=bcfg.inter,
                                    max_position_embeddings=bcfg.max_pos, layer_norm_eps=bcfg.eps)).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_bert_weights(bcfg, meta["bseed"]).items()}
    missing, unexpected = lm.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m] and not unexpected
    ct = CLIPTokenizer(vocab=sv.clip_vocab, merges=[tuple(m) for m in sv.clip_merges], model_max_length=77)
    tc = dict(vocab_size=ccfg.vocab, hidden_size=ccfg.hidden, intermediate_size=ccfg.inter, num_hidden_layers=ccfg.layers,
              num_attention_heads=ccfg.heads, max_position_embeddings=ccfg.max_pos, layer_norm_eps=ccfg.eps,
              bos_token_id=ccfg.bos_id, eos_token_id=ccfg.eos_id, pad_token_id=ccfg.eos_id, projection_dim=ccfg.proj)
    vc = dict(hidden_size=ccfg.v_hidden, intermediate_size=ccfg.v_inter, num_hidden_layers=ccfg.v_layers,
              num_attention_heads=ccfg.v_heads, image_size=ccfg.v_image, patch_size=ccfg.v_patch, layer_norm_eps=ccfg.eps,
              projection_dim=ccfg.proj)
    hc = CLIPModel(CLIPConfig(text_config=tc, vision_config=vc, projection_dim=ccfg.proj)).eval()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_clip_weights(ccfg, meta["cseed"]).items()}
    missing, unexpected = hc.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m] and not unexpected
    d = str(tmp / "clip")
    hc.save_pretrained(d)
    ct.save_pretrained(d)
    ip = CLIPImageProcessorPil(size={"shortest_edge": ccfg.v_image}, crop_size={"height": ccfg.v_image, "width": ccfg.v_image})
    CLIPProcessor(image_processor=ip, tokenizer=ct).save_pretrained(d)
    clip = CLIP(d)   # the drop-in wrapper, same constructor call as clip/clip.py:6-16
    assert clip.model is not None and clip.processor is not None
    return lm, bt, clip, sv, ccfg


@pytest.mark.parametrize("name", ["tiny_seq", "tiny_shuffle"])
def test_generate_caption_with_hugging_face_objects(name, tmp_path, monkeypatch):
    import utils
    from gen_utils import generate_caption
    from PIL import Image
    monkeypatch.setenv("CZC_PRECISION", "f32")
    meta, arr = load_case(name)
    lm, tok, clip, sv, ccfg = _hf_objects(meta, tmp_path)
    assert clip._device_processor_params() is not None, "the standard CLIP image processor must run on the device"
    imgs = [Image.fromarray(u) for u in synth.make_images_u8(meta["B"], ccfg.v_image)]
    import torch
    token_mask = torch.from_numpy(synth.make_token_mask(sv))          # a torch tensor, as demo.py:135 builds it
    utils.set_seed(meta["seed"])
    kw = dict(prompt=meta["prompt"], batch_size=meta["B"], max_len=meta["L"], top_k=meta["K"],
              temperature=meta["temperature"], max_iter=meta["I"], alpha=meta["alpha"], beta=meta["beta"],
              generate_order=meta["order"])
    with torch.no_grad():
        texts, scores = generate_caption([f"img{j}" for j in range(meta["B"])], lm, clip, tok, imgs, token_mask,
                                         logging.getLogger("hf-dropin"), **kw)
    assert texts == meta["texts"]
    np.testing.assert_allclose(np.array(scores, dtype=np.float64), np.array(meta["scores"]), atol=2e-5)
    # demo.py's own use of the wrapper outside a generate call
    emb = clip.compute_image_representation_from_image_instance(imgs)
    np.testing.assert_allclose(np.asarray(emb), arr["image_embeds"], atol=3e-5)
    last = meta["positions"][-1]
    assert float(token_mask[0, tok.vocab["."]]) == (1.0 if last == meta["L"] - 1 else 0.0)

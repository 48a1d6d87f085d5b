"""-m gpu: the reference-shaped call surface (gen_utils / control_gen_utils / clip.clip / utils at
the repo root) driven like demo.py drives the reference, compared with reference goldens."""
import logging
import os

import numpy as np
import pytest

from conzic_amd import synth
from goldutil import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f32_engine(monkeypatch):
    monkeypatch.setenv("CZC_PRECISION", "f32")  # id-for-id trajectories need the verification precision


def _objects(meta):
    from clip.clip import CLIP
    from conzic_amd.models import SyntheticLM
    from conzic_amd.text import tokenizers_from_vocab
    sv = synth.make_vocab_tiny()
    bcfg = synth.BertCfg(**meta["bert_cfg"])
    ccfg = synth.ClipCfg(**meta["clip_cfg"])
    bt, ct = tokenizers_from_vocab(sv)
    lm = SyntheticLM(bcfg, meta["bseed"])
    clip = CLIP.from_state(ccfg, synth.make_clip_weights(ccfg, meta["cseed"]), ct)
    if meta.get("pos"):
        clip.pos_tags = synth.make_pos_tags(len(sv.bert_tokens))
    elif meta["gamma"] is not None:
        clip.lexicon = synth.make_lexicon(len(sv.bert_tokens))
    from PIL import Image
    imgs = [Image.fromarray(u) for u in synth.make_images_u8(meta["B"], ccfg.v_image)]
    mask = synth.make_token_mask(sv)
    return lm, clip, bt, imgs, mask


@pytest.mark.parametrize("name", ["tiny_seq", "tiny_shuffle", "tiny_span", "tiny_random", "tiny_senti_seq",
                                  "tiny_senti_shuffle", "tiny_pos_seq"])
def test_generate_caption_dropin_matches_reference(name):
    import utils
    from control_gen_utils import control_generate_caption
    from gen_utils import generate_caption
    meta, arr = load_case(name)
    lm, clip, tok, imgs, mask = _objects(meta)
    utils.set_seed(meta["seed"])  # once per process, as demo.py:107 does
    logger = logging.getLogger("dropin-test")
    names = [f"img{j}" for j in range(meta["B"])]
    kw = dict(prompt=meta["prompt"], batch_size=meta["B"], max_len=meta["L"], top_k=meta["K"],
              temperature=meta["temperature"], max_iter=meta["I"], alpha=meta["alpha"], beta=meta["beta"],
              generate_order=meta["order"])
    if meta["gamma"] is None:
        texts, scores = generate_caption(names, lm, clip, tok, imgs, mask, logger, **kw)
    elif meta.get("pos"):
        texts, scores = control_generate_caption(names, lm, clip, tok, imgs, mask, logger, gamma=meta["gamma"],
                                                 ctl_type="pos", pos_type=meta["pos"], **kw)
    else:
        texts, scores = control_generate_caption(names, lm, clip, tok, imgs, mask, logger, gamma=meta["gamma"],
                                                 ctl_type="sentiment", style_type=meta["style"], **kw)
    assert texts == meta["texts"]
    np.testing.assert_allclose(np.array(scores, dtype=np.float64), np.array(meta["scores"]), atol=2e-5)
    # in-place mask mutation contract (utils.py:53-59): '.' reflects the last visited position
    last = meta["positions"][-1]
    assert mask[0, tok.vocab["."]] == (1.0 if last == meta["L"] - 1 else 0.0)


def test_generate_caption_two_streams_same_captions(monkeypatch):
    """batch_size = 64 through the reference-shaped generate_caption: above 2 x 32 images the runtime polishes two
    sub-batches on two streams (CZC_STREAMS, default 2); captions, scores and the shuffle order drawn from the global
    RNG are those of the one-stream run, and the second sample reuses the cached image embeddings on both streams."""
    import utils
    from conzic_amd import runtime
    from gen_utils import generate_caption
    meta, _ = load_case("tiny_shuffle")
    meta = dict(meta, B=64)
    logger = logging.getLogger("dropin-test")
    names = [f"img{j}" for j in range(64)]
    kw = dict(prompt=meta["prompt"], batch_size=64, max_len=meta["L"], top_k=meta["K"], temperature=meta["temperature"],
              max_iter=2, alpha=meta["alpha"], beta=meta["beta"], generate_order="shuffle")
    out = {}
    for streams in ("1", "2"):
        monkeypatch.setenv("CZC_STREAMS", streams)
        lm, clip, tok, imgs, mask = _objects(meta)
        utils.set_seed(meta["seed"])
        res = [generate_caption(names, lm, clip, tok, imgs, mask.copy(), logger, **kw) for _ in range(2)]  # two samples
        eng = runtime.get_engine(lm, clip, tok)
        assert (getattr(eng, "_group", None) is not None) == (streams == "2")
        if streams == "2":
            assert eng._group.streams == 2 and [hi - lo for lo, hi in eng._group.parts(64)] == [32, 32]
        out[streams] = res
        runtime.evict()
    for (t1, s1), (t2, s2) in zip(out["1"], out["2"]):
        assert t1 == t2
        np.testing.assert_allclose(np.array(s1), np.array(s2), atol=1e-6)


def test_image_embeds_cached_across_samples():
    """North star: the ViT encode happens once per image.  The same image objects polished again (the samples_num
    loop of demo.py:83) and `ImageEmbeds` handed back in their place must not go through the vision tower again,
    and must give the captions a fresh encode gives."""
    import utils
    from clip.clip import ImageEmbeds
    from gen_utils import generate_caption
    meta, arr = load_case("tiny_seq")
    lm, clip, tok, imgs, mask = _objects(meta)
    logger = logging.getLogger("dropin-test")
    names = [f"img{j}" for j in range(meta["B"])]
    kw = dict(prompt=meta["prompt"], batch_size=meta["B"], max_len=meta["L"], top_k=meta["K"],
              temperature=meta["temperature"], max_iter=meta["I"], alpha=meta["alpha"], beta=meta["beta"],
              generate_order="sequential")
    utils.set_seed(meta["seed"])
    t1, s1 = generate_caption(names, lm, clip, tok, imgs, mask.copy(), logger, **kw)
    eng = clip._engine
    eng.profile_reset()
    eng.profile(True)
    t2, s2 = generate_caption(names, lm, clip, tok, imgs, mask.copy(), logger, **kw)                     # same objects
    t3, s3 = generate_caption(names, lm, clip, tok, ImageEmbeds(clip.last_image_embeds()), mask.copy(), logger, **kw)
    eng.profile(False)
    assert eng.profile_get("gemm_vision")["launches"] == 0, "the vision tower ran again"
    assert t1 == t2 == t3 == meta["texts"]
    np.testing.assert_allclose(np.array(s2), np.array(s1), atol=1e-6)
    np.testing.assert_allclose(np.array(s3), np.array(s1), atol=1e-6)


def test_clip_wrapper_methods():
    from oracle import models as M
    import torch
    meta, arr = load_case("tiny_seq")
    lm, clip, tok, imgs, mask = _objects(meta)
    emb = clip.compute_image_representation_from_image_instance(imgs)
    np.testing.assert_allclose(np.asarray(emb), arr["image_embeds"], atol=3e-5)
    texts = ["image of a " + " ".join(w for w in synth.make_vocab_tiny().bert_tokens[300:303]), "image of a", "the . of"]
    te = clip.compute_text_representation(texts)
    score, ref = clip.compute_image_text_similarity_via_raw_text(emb[:1], texts)
    ccfg = synth.ClipCfg(**meta["clip_cfg"])
    w = M.to_torch(synth.make_clip_weights(ccfg, meta["cseed"]))
    enc = clip.tokenizer(texts)
    ids = torch.tensor(enc["input_ids"])
    lens = torch.tensor(enc["attention_mask"]).sum(1)
    rte = M.clip_text_embeds(w, ccfg, ids, lens)
    np.testing.assert_allclose(np.asarray(te), rte.numpy(), atol=3e-5)
    rs, rr = M.clip_similarity(w, torch.from_numpy(arr["image_embeds"][:1]), rte)
    np.testing.assert_allclose(np.asarray(ref), rr.numpy(), atol=5e-6)
    np.testing.assert_allclose(np.asarray(score), rs.numpy(), atol=5e-6)
    assert abs(float(np.asarray(score).sum()) - 1.0) < 1e-5


def test_checkpoint_with_legacy_bert_names_and_published_logit_scale(tmp_path, monkeypatch):
    """The published `bert-base-uncased` safetensors still carries TF-style `LayerNorm.gamma` / `LayerNorm.beta` and the
    pooler / next-sentence tensors, and the published CLIP checkpoints carry logit_scale = ln 100: such a directory
    loads, the engine precision is chosen from the logit scale (screen-then-refine), and a step equals the in-memory engine's."""
    from conzic_amd import checkpoint, harness, native, synth
    from conzic_amd.engine import Engine
    monkeypatch.delenv("CZC_PRECISION", raising=False)
    su = harness.build_synthetic(True, native.PREC_REFINE, logit_scale=4.6052)
    bw, cw = synth.make_bert_weights(su.bert_cfg, 11), synth.make_clip_weights(su.clip_cfg, 12)
    legacy = {}
    for k, v in bw.items():
        if k.endswith("LayerNorm.weight"):
            k = k[:-len("weight")] + "gamma"
        elif k.endswith("LayerNorm.bias"):
            k = k[:-len("bias")] + "beta"
        legacy[k] = v
    H = su.bert_cfg.hidden
    legacy["bert.pooler.dense.weight"] = np.zeros((H, H), np.float32)
    legacy["bert.pooler.dense.bias"] = np.zeros(H, np.float32)
    legacy["cls.seq_relationship.weight"] = np.zeros((2, H), np.float32)
    legacy["cls.seq_relationship.bias"] = np.zeros(2, np.float32)
    legacy["bert.embeddings.position_ids"] = np.arange(su.bert_cfg.max_pos, dtype=np.float32)[None]
    assert any(k.endswith(".gamma") for k in legacy)
    bdir, cdir = checkpoint.write_checkpoint_dirs(str(tmp_path), su.bert_cfg, legacy, su.clip_cfg, cw, su.sv)
    eng, bcfg, ccfg, bt, ct = checkpoint.engine_from_checkpoints(bdir, cdir)   # precision from the checkpoint
    try:
        assert abs(ccfg.logit_scale - 4.6052) < 1e-6 and eng.precision == native.PREC_REFINE
        eng.set_token_mask(su.token_mask)
        emb = np.random.default_rng(3).standard_normal((2, su.clip_cfg.proj)).astype(np.float32)
        inp = np.array([bt.encode("Image of a" + bt.mask_token * 5)] * 2, dtype=np.int32)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        outs = []
        for e in (eng, su.engine):
            e.set_image_embeds(emb)
            outs.append(e.step(inp.copy(), 4, 12, hp))
        np.testing.assert_array_equal(outs[0]["idxs"], outs[1]["idxs"])
        np.testing.assert_array_equal(outs[0]["final_score"], outs[1]["final_score"])
    finally:
        eng.close()
        su.engine.close()


def test_engine_from_checkpoint_directories_matches_in_memory_engine(tmp_path):
    """Real-checkpoint route on synthetic weights laid out as Hugging Face directories:
    same step results as the engine fed from memory."""
    from conzic_amd import checkpoint, harness, native, synth
    from conzic_amd.engine import Engine
    su = harness.build_synthetic(True, native.PREC_F32)
    sv = su.sv
    bw, cw = synth.make_bert_weights(su.bert_cfg, 11), synth.make_clip_weights(su.clip_cfg, 12)
    bdir, cdir = checkpoint.write_checkpoint_dirs(str(tmp_path), su.bert_cfg, bw, su.clip_cfg, cw, sv)
    eng, bcfg, ccfg, bt, ct = checkpoint.engine_from_checkpoints(bdir, cdir, native.PREC_F32)
    try:
        assert bcfg == su.bert_cfg
        eng.set_token_mask(su.token_mask)
        rng = np.random.default_rng(3)
        emb = rng.standard_normal((2, su.clip_cfg.proj)).astype(np.float32)
        inp = np.array([bt.encode("Image of a" + bt.mask_token * 5)] * 2, dtype=np.int32)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        outs = []
        for e in (eng, su.engine):
            e.set_image_embeds(emb)
            outs.append(e.step(inp.copy(), 4, 12, hp))
        for k in ("idxs", "clip_ids", "best"):
            np.testing.assert_array_equal(outs[0][k], outs[1][k])
        np.testing.assert_array_equal(outs[0]["final_score"], outs[1]["final_score"])
    finally:
        eng.close()
        su.engine.close()


@pytest.mark.parametrize("argv", [
    ["--run_type", "caption", "--order", "sequential"],
    ["--run_type", "caption", "--order", "span"],
    ["--run_type", "controllable", "--control_type", "sentiment", "--sentiment_type", "negative", "--order", "shuffle"],
    ["--run_type", "controllable", "--control_type", "pos", "--order", "sequential",
     "--pos_type", '[["DET"], ["ADJ", "NOUN"], ["NOUN"], ["VERB"], ["ADP"]]'],
])
def test_demo_harness_runs_every_run_type(argv, caplog):
    """conzic_amd.demo_cli in the role of demo.py (demo.py:105-153: set_seed once, models, token_mask, samples loop)
    through every run type the reference's demo offers, on the tiny synthetic towers: runs to completion and logs a
    final and a best caption per sample (gen_utils.py:326-333 / control_gen_utils.py tail)."""
    from conzic_amd import demo_cli
    caplog.set_level(logging.INFO)
    demo_cli.main(["--synthetic", "--tiny", "--samples_num", "2", "--sentence_len", "5", "--candidate_k", "12",
                   "--num_iterations", "2", "--batch_size", "2"] + argv)
    text = "\n".join(r.getMessage() for r in caplog.records)
    assert text.count("Sample ") == 2
    assert "final caption" in text.lower() or "best caption" in text.lower()


def test_refine_guard_trip_repeats_the_call_on_the_split_engine(monkeypatch):
    """runtime.run_generation under the screen-then-refine engine: when the guard trips (forced here by a 1e-6 trip point)
    the call is repeated on the all-split-fp16 engine and returns exactly what CZC_PRECISION=split returns; with the
    default trip point nothing trips and the refine engine's own result stands; CZC_REFINE_GUARD=warn only logs."""
    import utils
    from conzic_amd import native, runtime
    from gen_utils import generate_caption
    meta, _ = load_case("tiny_scale100")
    names = [f"img{j}" for j in range(meta["B"])]
    kw = dict(prompt=meta["prompt"], batch_size=meta["B"], max_len=meta["L"], top_k=meta["K"],
              temperature=meta["temperature"], max_iter=meta["I"], alpha=meta["alpha"], beta=meta["beta"],
              generate_order=meta["order"])

    class Log:
        def __init__(self):
            self.lines = []

        def info(self, s):
            self.lines.append(str(s))

    def run(prec, guard_x=None, mode=None):
        monkeypatch.setenv("CZC_PRECISION", prec)
        for k, v in (("CZC_REFINE_GUARD_X1E6", guard_x), ("CZC_REFINE_GUARD", mode)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        lm, clip, tok, imgs, mask = _objects(meta)
        utils.set_seed(meta["seed"])
        log = Log()
        out = generate_caption(names, lm, clip, tok, imgs, mask, log, **kw)
        precs = sorted(k[3] for k in runtime._ENGINES if len(k) == 5)
        runtime.evict()
        return out, log.lines, precs

    split, _, _ = run("split")
    plain, lines, precs = run("refine")
    assert precs == [native.PREC_REFINE] and not any("guard" in ln for ln in lines)
    assert any("engine precision: screen-then-refine" in ln for ln in lines)
    forced, lines, precs = run("refine", guard_x="1")
    assert precs == [native.PREC_SPLIT, native.PREC_REFINE]
    assert any("repeating the call on the all-split engine" in ln for ln in lines)
    assert forced[0] == split[0]
    np.testing.assert_array_equal(np.array(forced[1]), np.array(split[1]))
    warned, lines, precs = run("refine", guard_x="1", mode="warn")
    assert precs == [native.PREC_REFINE] and any("screen-then-refine guard" in ln for ln in lines)
    assert warned[0] == plain[0]


@pytest.mark.parametrize("precision", ["bf16", "refine"])
def test_fp16_residual_overflow_falls_back_to_fp32_rows(monkeypatch, precision):
    """The bf16 engine's text tower -- and the screening pass of the screen-then-refine engine inside a whole generation call --
    keep the residual stream as fp16 rows.  A checkpoint whose rows leave the fp16 range (here: token embeddings scaled by 2e6)
    produces non-finite cosines, which the engine reports (CZC_ERR_OVERFLOW) instead of hiding; `runtime.run_generation` then
    switches that engine to fp32 rows (option resid16 = 0 / refine_rows16 = 0), repeats the call and says so."""
    import utils
    from clip.clip import CLIP
    from conzic_amd import runtime
    from conzic_amd.models import SyntheticLM
    from conzic_amd.text import tokenizers_from_vocab
    from gen_utils import generate_caption
    monkeypatch.setenv("CZC_PRECISION", precision)
    sv = synth.make_vocab()
    bcfg, ccfg = synth.bert_base(), synth.clip_b32()
    bt, ct = tokenizers_from_vocab(sv)
    cw = synth.make_clip_weights(ccfg, 12)
    key = "text_model.embeddings.token_embedding.weight"
    cw[key] = np.asarray(cw[key], dtype=np.float32) * 2e6
    lm = SyntheticLM(bcfg, 11)
    clip = CLIP.from_state(ccfg, cw, ct)
    from PIL import Image
    B, L = 2, 4
    imgs = [Image.fromarray(u) for u in synth.make_images_u8(B, ccfg.v_image)]
    mask = synth.make_token_mask(sv, regular_only=True)

    class Log:
        def __init__(self):
            self.lines = []

        def info(self, s_):
            self.lines.append(str(s_))

    log = Log()
    utils.set_seed(42)
    texts, scores = generate_caption([f"img{j}" for j in range(B)], lm, clip, bt, imgs, mask, log, prompt="Image of a",
                                     batch_size=B, max_len=L, top_k=50, temperature=0.1, max_iter=2, alpha=0.02, beta=2.0,
                                     generate_order="sequential")
    assert any("fp16 residual stream overflowed" in ln for ln in log.lines), log.lines[-5:]
    assert len(texts) == 3 and all(np.isfinite(np.array(sc, dtype=np.float64)).all() for sc in scores)
    runtime.evict()

"""-m gpu, OPT-IN: the first run on the checkpoints the reference actually loads (`demo.py:125-132`: bert-base-uncased,
`clip/clip.py:11-16`: openai/clip-vit-base-patch32).  Neither box of this build holds those files (no network), so these tests are
SKIPPED unless two environment variables point at local Hugging Face style directories (config.json + model.safetensors +
vocab.txt, resp. vocab.json + merges.txt):

    CZC_BERT_DIR=/path/to/bert-base-uncased CZC_CLIP_DIR=/path/to/clip-vit-base-patch32 \\
        [CZC_STOP_WORDS=/path/to/ConZIC/stop_words.txt] python -m pytest tests/test_real_checkpoints_gpu.py -m gpu -q -s

What they hold the engine to, against the CPU oracle fed with the SAME tensors and vocabularies (VERDICT r5
"missing" #2): the real 30 522-piece vocabulary through the device text bridge (accent-stripped pieces, `##` gluing, CJK,
punctuation, the real 48 894-merge BPE), a trained BERT's peaky softmax at tau = 0.1 (most of the K probabilities underflow: the
zero-probability tail and its de-duplication), real `logit_scale` = ln 100 activations through the engine `runtime.choose_precision`
selects (screen-then-refine) and its guard.  Nothing here is needed for the synthetic-weight parity suite."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BERT_DIR, CLIP_DIR = os.environ.get("CZC_BERT_DIR"), os.environ.get("CZC_CLIP_DIR")
need_files = pytest.mark.skipif(not (BERT_DIR and CLIP_DIR and os.path.isdir(BERT_DIR) and os.path.isdir(CLIP_DIR)),
                                reason="set CZC_BERT_DIR / CZC_CLIP_DIR to local bert-base-uncased / clip-vit-base-patch32 directories")
SEED_LEN = 4


def _token_mask(bt):
    """demo.py:135-143: ones, every stop word's id (OOV -> [UNK]) zeroed.  Without the reference's stop_words.txt: the [unusedN]
    range, single characters, digit strings and [UNK] (what that file's composition amounts to)."""
    V = len(bt.id2tok)
    m = np.ones((1, V), np.float32)
    path = os.environ.get("CZC_STOP_WORDS")
    if path and os.path.exists(path):
        unk = bt.vocab["[UNK]"]
        for w in open(path, encoding="utf-8").read().split("\n"):
            m[0, bt.vocab.get(w.strip(), unk)] = 0.0
        return m
    for t, i in bt.vocab.items():
        if t.startswith("[unused") or t == "[UNK]" or t.isdigit() or len(t) == 1:
            m[0, i] = 0.0
    return m


@pytest.fixture(scope="module")
def real():
    from conzic_amd import checkpoint
    from oracle import models as M, step as S, text as T
    bw, cw = checkpoint.read_safetensors(BERT_DIR), checkpoint.read_safetensors(CLIP_DIR)
    eng, bcfg, ccfg, bt, ct = checkpoint.engine_from_checkpoints(BERT_DIR, CLIP_DIR)   # precision from the checkpoint's logit_scale
    from conzic_amd.engine import normalize_state_name
    norm = lambda w: {normalize_state_name(k): v for k, v in w.items() if normalize_state_name(k)}  # noqa: E731
    o = S.Oracle(M.to_torch(norm(bw)), bcfg, M.to_torch(norm(cw)), ccfg, list(bt.id2tok), T.ClipBpe(ct.get_vocab(), list(ct.clip_merges)))
    mask = _token_mask(bt)
    eng.set_token_mask(mask)
    yield dict(eng=eng, o=o, bt=bt, ct=ct, bcfg=bcfg, ccfg=ccfg, mask=mask)
    eng.close()


@need_files
def test_published_logit_scale_selects_the_refine_engine(real):
    from conzic_amd import native
    assert abs(np.exp(real["ccfg"].logit_scale) - 100.0) < 1.0          # clip/clip.py:95-98 on the published checkpoint
    assert real["eng"].precision == native.PREC_REFINE


@need_files
@pytest.mark.parametrize("sentence", ["a man riding a wave on top of a surfboard", "café déjà vu — naïve coördination, 12½ résumés!",
                                      "two dogs play in the snow near a red barn ."])
def test_real_vocabulary_through_the_device_bridge(real, sentence):
    """BERT ids -> decode -> CLIP BPE ids on the device against the oracle's host restatement, with the REAL vocabularies: every
    candidate row of one step (K = 200 replacement tokens incl. `##` pieces, accents the uncased tokenizer stripped, punctuation)."""
    from conzic_amd.engine import Engine
    from oracle import text as T
    eng, o, bt = real["eng"], real["o"], real["bt"]
    ids = bt.encode("Image of " + sentence)
    L = len(ids) - SEED_LEN - 1
    assert 2 <= L <= 40
    inp = np.array([ids], dtype=np.int32)
    eng.set_image_embeds(np.random.default_rng(0).standard_normal((1, real["ccfg"].proj)).astype(np.float32))
    r = eng.step(inp.copy(), SEED_LEN + L // 2, 200, Engine.hyper(0.02, 2.0, 0.1), want=("cand_ids", "clip_ids", "clip_len"))
    for k in range(200):
        row = ids.copy()
        row[SEED_LEN + L // 2] = int(r["cand_ids"][0][k])
        want = T.bridge(row, o.id2tok, o.bpe)
        n = int(r["clip_len"][k])
        assert n == len(want) and r["clip_ids"][k, :n].tolist() == list(want), (k, o.decode(row))


@need_files
def test_step_on_real_checkpoints_against_the_oracle(real):
    """One image, three positions of a real caption: top-K ids of the trained BERT at tau = 0.1 (zero-probability tail included),
    every fused score within the 1e-3 bar (or the guard tripped), the winner wherever the oracle's own margin exceeds the bar;
    czc_dedup_stats reports the collapsed tail."""
    from conzic_amd.engine import Engine
    from oracle import step as S
    eng, o, bt = real["eng"], real["o"], real["bt"]
    L, K = 10, 200
    ids = bt.encode("Image of a man riding a wave on top of a surfboard .")[:SEED_LEN + L] + [bt.vocab["[SEP]"]]
    inp = np.array([ids], dtype=np.int32)
    rng = np.random.default_rng(7)
    pix = rng.integers(0, 256, size=(1, real["ccfg"].v_image, real["ccfg"].v_image, 3), dtype=np.uint8)
    from conzic_amd import synth
    pixels = synth.pixels_from_u8(pix)
    emb = eng.encode_images(pixels)
    np.testing.assert_allclose(emb, o.image_embeds(pixels).numpy(), atol=2e-4 * max(1.0, float(np.abs(emb).max())))
    hp = Engine.hyper(0.02, 2.0, 0.1)
    eng.profile_reset()
    eng.refine_guard(reset=True)
    for pos in (0, 4, L - 1):
        work = inp.copy()
        r = eng.step(work, SEED_LEN + pos, K, hp, dot_allowed=(pos == L - 1))
        tmask = torch.from_numpy(real["mask"].copy())
        o.update_token_mask(tmask, L, pos)
        ref_inp = torch.from_numpy(inp.astype(np.int64))
        ref_inp[:, SEED_LEN + pos] = o.mask_id
        ref = S.polish_step(o, ref_inp, torch.from_numpy(emb), tmask, SEED_LEN + pos, K, 0.1, 0.02, 2.0)
        gp, gi = ref["probs"].numpy()[0], ref["idxs"].numpy()[0]
        nz = int((gp > 0).sum())
        print(f"[real] position {pos}: {nz} of {K} probabilities non-zero, top p = {gp[0]:.3f}")
        assert r["idxs"][0][:nz].tolist() == gi[:nz].tolist() or len(set(r["idxs"][0][:nz]) ^ set(gi[:nz])) <= 2
        same = r["idxs"][0] == gi
        g = eng.refine_guard(reset=True)
        err = np.abs(r["final_score"][0] - ref["final"].numpy()[0])[same]
        assert err.max() < 1e-3 or g["tripped"] > 0, (pos, float(err.max()), g)
        fin = ref["final"].numpy()[0]
        srt = np.sort(fin)[::-1]
        if srt[0] - srt[1] > 2e-3:
            assert int(r["best"][0]) == int(ref["best"].numpy()[0])
    st = eng.stats()
    print(f"[real] de-duplicated {st['dedup_seqs']} of {st['clip_seqs']} candidate sentences")


def test_the_acceptance_tests_themselves_run_on_synthetic_checkpoint_directories(tmp_path):
    """The tests above only mean something if they run.  Here they do, in a subprocess, on two Hugging Face style directories
    written from the synthetic tiny towers at the published logit scale (`checkpoint.write_checkpoint_dirs`): same loader, same
    oracle wiring, same assertions -- only the weights and vocabularies are not the real ones."""
    import subprocess
    import sys
    from conzic_amd import checkpoint, harness, synth
    sv = harness.cached_vocab(True)
    bcfg, ccfg = synth.bert_tiny(len(sv.bert_tokens)), synth.clip_tiny(len(sv.clip_vocab))
    ccfg.logit_scale = 4.6052
    bdir, cdir = checkpoint.write_checkpoint_dirs(str(tmp_path), bcfg, synth.make_bert_weights(bcfg, 11), ccfg,
                                                  synth.make_clip_weights(ccfg, 12), sv)
    env = dict(os.environ, CZC_BERT_DIR=bdir, CZC_CLIP_DIR=cdir)
    env.pop("CZC_STOP_WORDS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "not acceptance_tests_themselves"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert "5 passed" in r.stdout, r.stdout[-500:]

"""-m gpu: the polishing step and whole trajectories through the C ABI (czc_step / czc_generate /
czc_encode_images / czc_encode_text) against goldens captured from the real reference."""
import numpy as np
import pytest
import torch

from conzic_amd import harness, native, synth
from conzic_amd.engine import Engine
from goldutil import load_case

pytestmark = pytest.mark.gpu

BF16, F32, SPLIT, FP16 = native.PREC_BF16, native.PREC_F32, native.PREC_SPLIT, native.PREC_FP16
REFINE = native.PREC_REFINE
SEED_LEN = 4

# fused-score tolerance of the north star ("within 1e-3 on the fused logits") at the BASELINE
# shapes (K=200).  The tiny fixtures use K<=16, where one candidate carries ~1/K of the CLIP
# softmax mass, so the same bf16 cosine error (<=4e-3) moves the fused score ~K_full/K_tiny more.
def tol_final(prec, tiny):
    if prec == F32:
        return 2e-5
    if prec == SPLIT:  # split-fp16 MFMA towers: fp32-class (22 mantissa bits), an order inside the 1e-3 bar
        return 1e-4
    if prec == REFINE:  # screen-then-refine: the north star's bar itself, at every logit scale (measured 3-5e-4 at x100)
        return 1e-3
    if prec == FP16:   # single-pass fp16 MFMA towers: ~8x below bf16 (measured 5e-5 on the full-size goldens at scale 14.3)
        return 4e-3 if tiny else 1.5e-4
    return 2.5e-2 if tiny else 1e-3


def bf16_in_budget(meta):
    """The bf16 engine is only what the product path selects (conzic_amd.runtime.choose_precision) while
    exp(logit_scale) keeps its cosine error inside the fused-score budget; checkpoints with the published
    scale of 100 get the split-fp16 engine, and the scale-100 goldens are held to the full bar with it."""
    from conzic_amd import runtime
    import math
    return math.exp(meta["logit_scale"]) <= runtime.BF16_MAX_LOGIT_SCALE_EXP

_setups = {}


def setup_for(meta, prec):
    key = (meta["tiny"], prec, meta["bseed"], meta["cseed"], round(meta["logit_scale"], 4), meta["regular_only"],
           meta["gamma"] is not None, bool(meta.get("pos")))
    if key not in _setups:
        if len(_setups) >= 3:  # bound device memory: drop the oldest engines
            k0 = next(iter(_setups))
            _setups.pop(k0).engine.close()
        _setups[key] = harness.build_synthetic(meta["tiny"], prec, meta["bseed"], meta["cseed"], meta["logit_scale"],
                                               meta["regular_only"], lexicon=meta["gamma"] is not None)
        if meta.get("pos"):
            _setups[key].engine.set_pos(synth.make_pos_tags(len(_setups[key].sv.bert_tokens)),
                                        synth.pos_template_masks(meta["pos"]))
    return _setups[key]


def gold_final(meta, arr, i, su):
    """final_score exactly as the reference forms it from the captured tensors
    (gen_utils.py:77, control_gen_utils.py:53-59)."""
    probs = torch.from_numpy(arr["probs"][i])
    cs = torch.from_numpy(arr["clip_score"][i])
    fin = meta["alpha"] * probs + meta["beta"] * cs
    if meta.get("pos"):
        from oracle import step as S
        from goldutil import make_oracle
        o, _, _ = make_oracle(meta)
        B, K = probs.shape
        gen_idx = SEED_LEN + meta["positions"][i]
        inp = torch.from_numpy(arr["inp_before"][i].astype(np.int64))
        mask = torch.from_numpy(su.token_mask.copy())
        mask[0, su.bert_tok.vocab["."]] = 1.0 if meta["positions"][i] == meta["L"] - 1 else 0.0
        idxs = torch.from_numpy(arr["idxs"][i].astype(np.int64))
        idxs_ = (idxs * mask[0][idxs]).long()
        rows = inp.unsqueeze(1).repeat(1, K, 1)
        rows[:, :, gen_idx] = idxs_
        praw = S.pos_scores(o, rows.view(B * K, -1), meta["pos"]).view(B, K)
        fin = fin + meta["gamma"] * torch.softmax(praw / 0.1, dim=-1)
    elif meta["gamma"] is not None:
        B, K = probs.shape
        gen_idx = SEED_LEN + meta["positions"][i]
        inp = torch.from_numpy(arr["inp_before"][i].astype(np.int64))
        mask = torch.from_numpy(su.token_mask.copy())
        mask[0, su.bert_tok.vocab["."]] = 1.0 if meta["positions"][i] == meta["L"] - 1 else 0.0
        idxs = torch.from_numpy(arr["idxs"][i].astype(np.int64))
        idxs_ = (idxs * mask[0][idxs]).long()
        rows = inp.unsqueeze(1).repeat(1, K, 1)
        rows[:, :, gen_idx] = idxs_
        reps = (idxs_[:, :, None] == rows).float().sum(2) - 1
        lex = torch.from_numpy(synth.make_lexicon(len(su.sv.bert_tokens)))
        keep = torch.ones_like(rows, dtype=torch.bool)
        for s in su.bert_tok.all_special_ids:
            keep &= rows != s
        sraw = (lex[rows] * keep).sum(2)
        if meta["style"] == "negative":
            sraw = -sraw
        fin = fin + meta["gamma"] * torch.softmax(sraw, 1) + 0.1 * (1 - torch.exp(reps))
    return fin.numpy()


ERR_LOG = []  # (case, precision, max |final_score error|, max |cosine error|) per checked image-step


def check_step(meta, arr, i, res, su, prec, next_inp):
    """Compare one engine step with golden step i; returns number of soft mismatches."""
    B, K = arr["probs"][i].shape
    tol = tol_final(prec, meta["tiny"])
    gfin = gold_final(meta, arr, i, su)
    gen_idx = SEED_LEN + meta["positions"][i]
    soft = 0
    for b in range(B):
        gi, ei = arr["idxs"][i][b], res["idxs"][b]
        gmap = {int(t): k for k, t in enumerate(gi)}
        common = [(k, gmap[int(t)]) for k, t in enumerate(ei) if int(t) in gmap]
        need = K - 1
        assert len(common) >= need, f"step {i} img {b}: only {len(common)}/{K} candidates shared"
        ek = np.array([c[0] for c in common])
        gk = np.array([c[1] for c in common])
        gp = arr["probs"][i][b]
        # fluency probs (tau=0.1 amplifies logit error tenfold)
        np.testing.assert_allclose(res["probs"][b][ek], gp[gk], rtol=3e-3, atol=1e-12)
        # the top-K SET and ORDER are derived from those floating-point values (BERT runs on split-fp16 MFMA, 22 mantissa bits, in
        # every engine precision; torch on fp32): they may differ from the reference's only at genuine near-ties -- a candidate
        # missing from the common set must sit at the K-th probability, and a candidate at another rank must have (to the
        # tolerance of the values themselves) the probability of the one whose rank it took.  Anything else is a wrong top-K
        for g_missing in sorted(set(range(K)) - set(gk.tolist())):
            assert abs(gp[g_missing] - gp[K - 1]) <= 6e-3 * gp[K - 1] + 1e-12, \
                f"step {i} img {b}: reference candidate at rank {g_missing} (p = {gp[g_missing]:.3e}) is missing and is no tie with the K-th ({gp[K - 1]:.3e})"
        moved = ek != gk
        if moved.any():
            np.testing.assert_allclose(gp[gk[moved]], gp[np.minimum(ek[moved], K - 1)], rtol=6e-3, atol=1e-12,
                                       err_msg=f"step {i} img {b}: top-K order differs at candidates that are no near-tie")
        assert (ek == gk).mean() > 0.97, "top-K order differs beyond near-ties"
        # bridged CLIP ids are integer work: exact
        for e_, g_ in zip(ek, gk):
            ln = arr["clip_lens"][i][b * K + g_]
            assert res["clip_len"][b * K + e_] == ln
            np.testing.assert_array_equal(res["clip_ids"][b * K + e_, :ln], arr["clip_ids"][i][b * K + g_, :ln])
        np.testing.assert_allclose(res["clip_ref"][b][ek], arr["clip_ref"][i][b][gk],
                                   atol={F32: 5e-6, SPLIT: 1e-5, FP16: 6e-4, REFINE: 6e-4}.get(prec, 4e-3))
        if len(common) == K:
            np.testing.assert_allclose(res["clip_score"][b][ek], arr["clip_score"][i][b][gk],
                                       atol={F32: 2e-6, SPLIT: tol / 2}.get(prec, tol / 2),
                                       rtol={F32: 1e-4, SPLIT: 2e-3}.get(prec, 5e-2 if meta["logit_scale"] < 4.0 else 0.2))
            np.testing.assert_allclose(res["final_score"][b][ek], gfin[b][gk], atol=tol, rtol=0)
            ERR_LOG.append((meta["name"], prec, float(np.abs(res["final_score"][b][ek] - gfin[b][gk]).max()),
                            float(np.abs(res["clip_ref"][b][ek] - arr["clip_ref"][i][b][gk]).max())))
        elif meta["gamma"] is None:
            # the top-K lists differ by one candidate (a near-tie at the K-th fluency probability): softmax_K is over
            # different sets, so hold the scores to the reference on the COMMON candidates with both softmaxes
            # renormalised over them (softmax restricted to a subset = the full one divided by the subset's mass)
            gcs, ecs = arr["clip_score"][i][b][gk], res["clip_score"][b][ek]
            gcs, ecs = gcs / gcs.sum(), ecs / ecs.sum()
            np.testing.assert_allclose(ecs, gcs, atol=tol / 2, rtol=0)
            efin = meta["alpha"] * res["probs"][b][ek] + meta["beta"] * ecs
            gres = meta["alpha"] * arr["probs"][i][b][gk] + meta["beta"] * gcs
            np.testing.assert_allclose(efin, gres, atol=tol, rtol=0)
            ERR_LOG.append((meta["name"], prec, float(np.abs(efin - gres).max()),
                            float(np.abs(res["clip_ref"][b][ek] - arr["clip_ref"][i][b][gk]).max())))
        # winner: identical token wherever the reference's own top-2 margin exceeds the error bound
        srt = np.sort(gfin[b])[::-1]
        margin = srt[0] - srt[1]
        gbest_tok = int(next_inp[b, gen_idx]) if next_inp is not None else None
        ebest_tok = int(res["cand_ids"][b][res["best"][b]])
        if gbest_tok is not None:
            if margin > 2 * tol:
                assert ebest_tok == gbest_tok, f"step {i} img {b}: winner {ebest_tok} != {gbest_tok} (margin {margin})"
            elif ebest_tok != gbest_tok:
                soft += 1
    return soft


def teacher_forced(meta, arr, prec, n_steps=None):
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                      control="pos" if meta.get("pos") else None)
    n = arr["probs"].shape[0] if n_steps is None else min(n_steps, arr["probs"].shape[0])
    soft = 0
    prev_inp = None
    for i in range(n):
        pos = meta["positions"][i]
        reuse = meta["reuse"][i]
        if reuse:
            # second position of a span: same forward, state after the first position.  Put the
            # REFERENCE's winner of the previous step there (a near-tie flip of the engine's own
            # winner must not leak into this step's comparison).
            inp = prev_inp
            j = next((t for t in range(i + 1, len(meta["reuse"])) if not meta["reuse"][t]), None)
            prev_gen = SEED_LEN + meta["positions"][i - 1]
            if j is not None and j < arr["inp_before"].shape[0] and \
                    SEED_LEN + meta["positions"][j] != prev_gen and not (j + 1 < len(meta["reuse"]) and meta["reuse"][j + 1]
                                                                         and SEED_LEN + meta["positions"][j] + 1 == prev_gen):
                inp[:, prev_gen] = arr["inp_before"][j][:, prev_gen]
            n_mask = 0
        else:
            inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
            n_mask = 1
            if i + 1 < len(meta["reuse"]) and meta["reuse"][i + 1]:
                n_mask = 2
        res = eng.step(inp, SEED_LEN + pos, meta["K"], hp, n_mask=n_mask, dot_allowed=(pos == meta["L"] - 1),
                       want=("probs", "idxs", "cand_ids", "clip_ids", "clip_len", "clip_score", "clip_ref",
                             "final_score", "best", "best_cos"))
        nxt = None
        if i + 1 < arr["inp_before"].shape[0] and not (i + 1 < len(meta["reuse"]) and meta["reuse"][i + 1]):
            nxt = arr["inp_before"][i + 1]
            if meta["positions"][i + 1] == pos:
                nxt = None  # the next step re-masks the same slot: winner not observable
        soft += check_step(meta, arr, i, res, su, prec, nxt)
        prev_inp = inp
    return soft, n


TINY = ["tiny_seq", "tiny_shuffle", "tiny_span", "tiny_random", "tiny_senti_seq", "tiny_senti_shuffle", "tiny_scale100",
        "tiny_pos_seq"]
FULL = ["full_cfg1", "full_synth_b2", "full_regular", "full_scale100", "full_shuffle_k512", "full_senti", "full_pos",
        "full_span", "full_random"]


@pytest.mark.parametrize("name", TINY)
def test_step_parity_tiny_f32(name):
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, F32)
    assert soft <= max(1, n // 10)


@pytest.mark.parametrize("name", [n for n in TINY if n != "tiny_scale100"])
def test_step_parity_tiny_bf16(name):
    meta, arr = load_case(name)
    assert bf16_in_budget(meta)
    soft, n = teacher_forced(meta, arr, BF16)
    # near-tie winners may flip in bf16 on the tiny towers (64-wide rows: cosines quantise coarsely; the bar there is 2.5e-2);
    # the hard asserts are inside check_step.  A flip at more than half of the steps would mean the tolerance rule, not the
    # arithmetic, is carrying the test (measured: see the [soft] lines of the GPU run)
    SOFT_LOG.append((name, "bf16", soft, n))
    print(f"[soft] {name} bf16: {soft}/{n} steps with a near-tie winner flip")
    assert soft <= (n + 1) // 2


@pytest.mark.parametrize("name", TINY)
def test_step_parity_tiny_split(name):
    """Split-fp16 MFMA engine (the precision the product path selects at logit_scale = ln 100): fp32-class."""
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, SPLIT)
    assert soft <= max(1, n // 10)


@pytest.mark.parametrize("name", FULL)
def test_step_parity_full_size_f32(name):
    """BASELINE configs 1-4 shapes (bert-base + CLIP ViT-B/32; K=200 / K=512, L=15 shuffle / sentiment gamma=5,
    L=12; logit scale 14.3 and 100) in the verification precision."""
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, F32, n_steps=4 if meta["K"] > 200 else 6)
    assert soft <= 1


@pytest.mark.parametrize("name", FULL)
def test_step_parity_full_size_split(name):
    """Every full-size golden, including the published-checkpoint logit scale (full_scale100), through the
    split-fp16 MFMA engine: fused score within 1e-4 (bar: 1e-3), identical winners."""
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, SPLIT, n_steps=None if meta["K"] <= 200 else 8)
    assert soft <= 1


@pytest.mark.parametrize("name", [n for n in FULL if n != "full_scale100"])
def test_step_parity_full_size_bf16(name):
    """BASELINE configs 2-4 shapes: bf16 MFMA engine vs the CPU reference: fused score within 1e-3."""
    meta, arr = load_case(name)
    assert bf16_in_budget(meta)
    teacher_forced(meta, arr, BF16, n_steps=10)


@pytest.mark.parametrize("prec", [F32, BF16, SPLIT])
@pytest.mark.parametrize("name", ["full_cfg1", "full_senti", "tiny_seq"])
def test_bert_logits_row_vs_reference(name, prec):
    """A4 directly: czc_step_out.logits (the masked row of BertForMaskedLM(inp).logits, gen_utils.py:69,42) against
    the row captured from the reference's own forward at the first step.  The bf16 and split engines both run the
    BERT tower on split-fp16 MFMA; tau = 0.1 multiplies a logit error by ten, hence the tight bar."""
    meta, arr = load_case(name)
    su = setup_for(meta, prec)
    su.engine.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative")
    inp = np.ascontiguousarray(arr["inp_before"][0], dtype=np.int32)
    res = su.engine.step(inp, SEED_LEN + meta["positions"][0], meta["K"], hp, dot_allowed=(meta["positions"][0] == meta["L"] - 1),
                         want=("logits", "idxs"))
    ref = arr["logits_row0"]
    assert res["logits"].shape == ref.shape
    err = np.abs(res["logits"] - ref).max()
    assert err < (5e-5 if prec == F32 else 2e-4), err
    assert (res["logits"].argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("prec", [F32, BF16])
def test_sentiment_table_keyed_by_word_and_pos(prec):
    """czc_set_lexicon_pos: the sentence score summed per WORD (first piece; '##' continuations add nothing) under
    the coarse POS class of that piece (sentiments_classifer.py:14-30), fused into the bridge kernel -- against the
    oracle on the same random table, through a whole sentiment step (gamma = 5, control_gen_utils.py:53-63)."""
    from oracle import step as S
    from goldutil import make_oracle
    meta, arr = load_case("tiny_senti_seq")
    su = harness.build_synthetic(True, prec, lexicon=True)
    try:
        V = len(su.sv.bert_tokens)
        rng = np.random.default_rng(7)
        table = rng.uniform(-1, 1, size=(V, 5)).astype(np.float32)
        cls = rng.integers(0, 5, size=V).astype(np.uint8)
        su.engine.set_lexicon_pos(table, cls)
        su.engine.set_image_embeds(arr["image_embeds"])
        o, _, mask = make_oracle(meta)
        o.lexicon_pos = (table, cls)
        i = 3
        gen_idx = SEED_LEN + meta["positions"][i]
        inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
        # put a '##' continuation piece into the row so that the word-start rule matters
        cont = next(j for j, t in enumerate(su.sv.bert_tokens) if t.startswith("##") and su.token_mask[0, j] >= 0)
        inp[:, SEED_LEN] = cont if gen_idx != SEED_LEN else inp[:, SEED_LEN]
        inp[:, SEED_LEN + 1] = cont if gen_idx != SEED_LEN + 1 else inp[:, SEED_LEN + 1]
        hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], False)
        res = su.engine.step(inp.copy(), gen_idx, meta["K"], hp, want=("idxs", "senti_raw", "final_score", "best"))
        o.update_token_mask(mask, meta["L"], meta["positions"][i])
        r = S.polish_step(o, torch.from_numpy(inp.astype(np.int64)), torch.from_numpy(arr["image_embeds"]), mask, gen_idx,
                          meta["K"], meta["temperature"], meta["alpha"], meta["beta"], gamma=meta["gamma"])
        np.testing.assert_array_equal(res["idxs"], r["idxs"].numpy())
        np.testing.assert_allclose(res["senti_raw"], r["senti_raw"].numpy(), atol=1e-5)
        np.testing.assert_allclose(res["final_score"], r["final"].numpy(), atol=2e-5 if prec == F32 else 2.5e-2)
        # and back to the per-token lexicon
        su.engine.set_lexicon_pos(None, None)
        res2 = su.engine.step(inp.copy(), gen_idx, meta["K"], hp, want=("senti_raw",))
        assert np.abs(res2["senti_raw"] - res["senti_raw"]).max() > 1e-3
    finally:
        su.engine.close()


@pytest.mark.parametrize("name", [n for n in FULL if n != "full_scale100"])
def test_step_parity_full_size_fp16(name):
    """Single-pass fp16 MFMA CLIP towers (the bf16 kernels on IEEE fp16 operands, same speed): fused score within
    1.5e-4 on the full-size goldens where bf16 is within 3.4e-4 of a 1e-3 bar."""
    meta, arr = load_case(name)
    teacher_forced(meta, arr, FP16, n_steps=10)


def test_fp16_at_the_published_logit_scale_is_marginal():
    """Why the product path picks split-fp16 and not plain fp16 at x100: on `full_scale100` the fp16 engine's worst fused
    score error over the ten steps is 2.3e-3 (cosine error 1.8e-4 x 100 ahead of softmax_K) -- out of the 1e-3 bar by 2.3x
    (bf16: by ~20x), and a peakier softmax than the goldens' would amplify it further.  Asserted as 'between split and 5e-3'."""
    meta, arr = load_case("full_scale100")
    su = setup_for(meta, FP16)
    su.engine.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    worst = 0.0
    for i in range(10):
        inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
        res = su.engine.step(inp, SEED_LEN + meta["positions"][i], meta["K"], hp, dot_allowed=(meta["positions"][i] == meta["L"] - 1))
        if (res["idxs"] == arr["idxs"][i]).all():
            worst = max(worst, float(np.abs(res["final_score"] - gold_final(meta, arr, i, su)).max()))
    assert 1e-5 < worst < 5e-3, worst


REFINE_LOG = []  # (case, candidate sequences seen, re-encoded) per test, printed by conftest


@pytest.mark.parametrize("name", FULL)
def test_step_parity_full_size_refine(name):
    """Screen-then-refine engine (CZC_PREC_REFINE; what the product path selects above x40, i.e. for the published
    checkpoints): every full-size golden -- `full_scale100` is the case it exists for -- teacher-forced, fused score
    inside the 1e-3 bar on ALL K candidates, winners identical wherever the reference's margin exceeds the bar, and
    only a fraction of the candidates re-encoded by the split-fp16 tower."""
    meta, arr = load_case(name)
    su = setup_for(meta, REFINE)
    su.engine.profile_reset()
    soft, n = teacher_forced(meta, arr, REFINE, n_steps=None if meta["K"] <= 200 else 8)
    assert soft <= 1
    st = su.engine.stats()
    REFINE_LOG.append((name, st["clip_seqs"], st["refine_seqs"], st["clip_rows"], st["refine_rows"]))
    assert 0 < st["refine_seqs"] <= 0.35 * st["clip_seqs"], st


@pytest.mark.parametrize("name", ["tiny_scale100", "tiny_seq", "tiny_senti_seq", "tiny_pos_seq", "tiny_span"])
def test_step_parity_tiny_refine(name):
    """K <= 16: (nearly) every candidate carries more softmax_K mass than the threshold, so the engine degenerates to the
    split-fp16 one; span steps (n_mask = 0 re-use of the BERT forward) and the control paths go through the two-pass step."""
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, REFINE)
    assert soft <= max(1, n // 10)


@pytest.mark.parametrize("name", ["full_scale100", "full_senti", "full_pos", "full_span", "full_random"])
def test_generate_free_running_full_size_refine(name):
    """czc_generate through the screen-then-refine engine reproduces the reference's trajectory id for id at the
    published-checkpoint logit scale (and with the sentiment / POS control scores fused in)."""
    meta, arr = load_case(name)
    su = setup_for(meta, REFINE)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                      control="pos" if meta.get("pos") else None)
    init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
    pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"],
                                             random_positions=meta["positions"] if meta["order"] == "random" else None)
    assert pos == meta["positions"]
    ids, cos = eng.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
    np.testing.assert_array_equal(ids, arr["snaps"])
    np.testing.assert_allclose(cos, np.array(meta["scores"][:-1], dtype=np.float32), atol=2e-5)  # the winner is always re-encoded


def test_margin_gate_skips_the_second_pass_without_changing_what_generate_returns():
    """czc_generate of the screen-then-refine engine returns ids of every step and the winner's cosine at the snapshot steps.
    With the margin gate (default: delta = 4e-4, x 1.75 while the screening pass of czc_generate runs on fp16 rows: option
    refine_rows16, the default) an image-step whose screening winner survives every cosine-error assignment
    within delta does no second pass; `full_scale100` (the published logit scale): same ids and cosines as the reference AND as
    the ungated engine, most image-steps gated, far fewer candidates re-encoded; czc_step never gates."""
    meta, arr = load_case("full_scale100")
    su = setup_for(meta, REFINE)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"])
    init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
    pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"])
    res = {}
    for gate in (400, 0):
        eng.set_option("refine_gate_x1e6", gate)
        eng.profile_reset()
        ids, cos = eng.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
        res[gate] = (ids, cos, eng.stats())
        np.testing.assert_array_equal(ids, arr["snaps"])
        np.testing.assert_allclose(cos, np.array(meta["scores"][:-1], dtype=np.float32), atol=2e-5)
    on, off = res[400][2], res[0][2]
    assert off["gated_image_steps"] == 0 and off["gate_image_steps"] == 0
    audits = sum(1 for s_ in range(len(pos)) if (s_ + 1) % every == 0 and (s_ // every) % 4 == 0)  # audit steps never gate
    assert on["gate_image_steps"] == meta["B"] * (len(pos) - audits)
    frac = on["gated_image_steps"] / on["gate_image_steps"]
    print(f"[margin gate] {on['gated_image_steps']}/{on['gate_image_steps']} image-steps gated; candidates re-encoded "
          f"{on['refine_seqs']} (gate on) vs {off['refine_seqs']} (off)")
    assert frac >= 0.5, frac
    assert on["refine_seqs"] < 0.5 * off["refine_seqs"]
    # the screening pass on fp32 rows (the form czc_step keeps): the same ids and cosines, a few more image-steps gated
    try:
        eng.set_option("refine_gate_x1e6", 400)
        eng.set_option("refine_rows16", 0)
        eng.profile_reset()
        ids32, cos32 = eng.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
        st32 = eng.stats()
    finally:
        eng.set_option("refine_rows16", 1)
    np.testing.assert_array_equal(ids32, res[400][0])
    np.testing.assert_allclose(cos32, res[400][1], atol=2e-5)
    print(f"[margin gate] fp32-row screening: {st32['gated_image_steps']}/{st32['gate_image_steps']} image-steps gated")
    assert st32["gated_image_steps"] >= on["gated_image_steps"]
    # parity granularity is untouched: a czc_step refines the full selection whatever the option says
    eng.set_option("refine_gate_x1e6", 400)
    eng.profile_reset()
    inp = np.ascontiguousarray(arr["inp_before"][0], dtype=np.int32)
    eng.step(inp, SEED_LEN + pos[0], meta["K"], hp, dot_allowed=(pos[0] == meta["L"] - 1))
    st = eng.stats()
    assert st["gate_image_steps"] == 0 and st["refine_seqs"] >= 2 * meta["B"]


def test_refine_guard_audits_every_generate_call_whether_or_not_cosines_are_read():
    """The margin gate rests on a bound only the guard polices, and gated image-steps never feed the guard: the audit steps do.
    They used to exist only where the caller asked for cosines (out_cos != NULL).  Now: the snapshot step of every fourth
    sweep is an audit step whatever the caller reads, a call too short to reach one audits its last step, and with the guard
    switched off nothing is gated.  The trip point the engine applies is readable (czc_get_option)."""
    meta, arr = load_case("full_scale100")
    su = setup_for(meta, REFINE)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"])
    init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
    pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"])
    B = meta["B"]
    assert eng.get_option("refine_guard_x1e6") == 200 and eng.get_option("refine_rows16") == 1
    assert eng.get_option("refine_guard_generate_x1e6") == 350 and eng.get_option("refine_gate_generate_x1e6") == 700
    assert eng.get_option("has_folded_ln_weights") == 1
    with pytest.raises(native.NativeError) as ei:
        eng.get_option("no_such_option")
    assert ei.value.code == native.ERR_ARG
    # without cosines: same ids, the guard has measured something, the audit steps were not gated
    eng.refine_guard(reset=True)
    eng.profile_reset()
    ids, cos = eng.generate(B, init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every, want_cos=False)
    assert cos is None
    np.testing.assert_array_equal(ids, arr["snaps"])
    g = eng.refine_guard(reset=True)
    st = eng.stats()
    audits = sum(1 for s_ in range(len(pos)) if (s_ + 1) % every == 0 and (s_ // every) % 4 == 0)
    assert audits >= 1 and st["gate_image_steps"] == B * (len(pos) - audits)
    assert 0.0 < g["max_dev"] < 3.5e-4 and g["tripped"] == 0
    # a call shorter than one sweep (no snapshot step at all): its last step is the audit
    short = pos[:3]
    eng.profile_reset()
    eng.generate(B, init, meta["L"], SEED_LEN, meta["K"], short, hp, n_mask=nm[:3], snapshot_every=every, want_cos=False)
    g = eng.refine_guard(reset=True)
    assert eng.stats()["gate_image_steps"] == B * 2 and g["max_dev"] > 0.0
    # guard off -> gate off (nothing would police the bound)
    try:
        eng.set_option("refine_guard_x1e6", 0)
        eng.profile_reset()
        eng.generate(B, init, meta["L"], SEED_LEN, meta["K"], short, hp, n_mask=nm[:3], snapshot_every=every)
        assert eng.stats()["gate_image_steps"] == 0
    finally:
        eng.set_option("refine_guard_x1e6", 200)


def test_refine_engine_vs_split_engine_many_image_steps():
    """The goldens hold ~20 image-steps per case; this holds the screen-then-refine engine to the all-split-fp16 engine
    (pinned to the reference within 8e-6) on 32 images x 6 positions at the published logit scale: same candidate lists,
    fused score within the 1e-3 bar on every one of 38 400 candidates, winners identical wherever the split engine's own
    top-2 margin exceeds the bar."""
    B, L, K, P = 32, 10, 200, 6
    hp = Engine.hyper(0.02, 2.0, 0.1)
    rng = np.random.default_rng(77)
    emb = rng.standard_normal((B, 512)).astype(np.float32)
    outs = {}
    inp0 = None
    for prec in (SPLIT, REFINE):
        su = harness.build_synthetic(False, prec, logit_scale=4.6052, regular_only=True)
        try:
            if inp0 is None:
                inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
                regular = np.nonzero(su.token_mask[0] > 0)[0]
                inp0[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
            su.engine.set_image_embeds(emb)
            cur = inp0.copy()
            rows = []
            for p in range(P):
                before = cur.copy() if prec == SPLIT else outs[SPLIT][p][0]
                work = before.copy()
                r = su.engine.step(work, SEED_LEN + 2 + p, K, hp, want=("idxs", "final_score", "best"))
                rows.append((before, r))
                cur = work
            outs[prec] = rows
        finally:
            su.engine.close()
    worst = 0.0
    for (_, a), (_, b) in zip(outs[SPLIT], outs[REFINE]):
        np.testing.assert_array_equal(a["idxs"], b["idxs"])
        worst = max(worst, float(np.abs(a["final_score"] - b["final_score"]).max()))
        srt = np.sort(a["final_score"], axis=1)[:, ::-1]
        clear = (srt[:, 0] - srt[:, 1]) > 2e-3
        assert (a["best"][clear] == b["best"][clear]).all()
    ERR_LOG.append(("refine_vs_split_32x6", REFINE, worst, 0.0))
    assert worst < 1e-3, worst


GUARD_LOG = []  # (case, max |screening error - mean| on re-encoded candidates, tripped image-steps)


@pytest.mark.parametrize("name", ["full_scale100", "full_senti", "full_shuffle_k512"])
def test_refine_guard_measures_the_screening_error(name):
    """czc_refine_guard: every screen-then-refine step records, on the candidates it re-encodes exactly, how far the
    single-pass fp16 cosine is from the exact one once the estimated mean error is removed.  On the validated weights the
    figure stays inside the 2.5e-4 the 1e-3 bound needs and nothing trips at the default trip point (2.0e-4); with the
    trip point at 1e-6 every image-step trips."""
    meta, arr = load_case(name)
    su = setup_for(meta, REFINE)
    eng = su.engine
    eng.refine_guard(reset=True)
    teacher_forced(meta, arr, REFINE, n_steps=6)
    g = eng.refine_guard(reset=True)
    GUARD_LOG.append((name, g["max_dev"], g["tripped"]))
    assert 0.0 < g["max_dev"] < 2.0e-4 and g["tripped"] == 0, g
    try:
        eng.set_option("refine_guard_x1e6", 1)
        teacher_forced(meta, arr, REFINE, n_steps=2)
        g2 = eng.refine_guard(reset=True)
        assert g2["tripped"] == 2 * meta["B"], g2
    finally:
        eng.set_option("refine_guard_x1e6", 200)
    assert eng.refine_guard()["tripped"] == 0


OUTLIER_LOG = []  # (outlier factor, worst |d final_score| refine vs split, guard max_dev, tripped image-steps, image-steps)


def test_refine_guard_catches_towers_the_fp16_screening_pass_does_not_carry():
    """the 1e-3 bound of the screen-then-refine engine was validated on random-weight towers only; real
    checkpoints have activation OUTLIER channels (a few LayerNorm gains tens of times the rest) that a single-pass fp16 tower
    rounds more coarsely.  Emulated here by scaling 6 channels of every text-tower LayerNorm gain by F: the refine engine is
    compared with the all-split engine on 8 images x 3 positions at the published logit scale.  Safety property: whenever
    the fused score leaves the 1e-3 bar on any candidate, the guard has tripped (so `runtime.run_generation` would repeat
    the call on the split engine); on the plain towers (F = 1) nothing trips."""
    B, L, K, P, SCALE = 8, 10, 200, 3, 4.6052
    hp = Engine.hyper(0.02, 2.0, 0.1)
    rng = np.random.default_rng(41)
    emb = rng.standard_normal((B, 512)).astype(np.float32)
    seen_trip = False
    for F_ in (1.0, 12.0, 24.0, 48.0):
        ccfg = synth.clip_b32()
        ccfg.logit_scale = SCALE
        cw = synth.make_clip_weights(ccfg, 12)
        ch = np.array([7, 93, 200, 301, 402, 499])
        for n in range(ccfg.layers):
            for ln in ("layer_norm1", "layer_norm2"):
                g = np.array(cw[f"text_model.encoder.layers.{n}.{ln}.weight"], dtype=np.float32, copy=True)
                g[ch] *= F_
                cw[f"text_model.encoder.layers.{n}.{ln}.weight"] = g
        outs = {}
        inp0 = None
        guard = None
        for prec in (SPLIT, REFINE):
            su = harness.build_synthetic(False, prec, logit_scale=SCALE, regular_only=True, clip_w=cw, clip_cfg=ccfg)
            try:
                if inp0 is None:
                    inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
                    regular = np.nonzero(su.token_mask[0] > 0)[0]
                    inp0[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
                su.engine.set_image_embeds(emb)
                if prec == REFINE:
                    su.engine.refine_guard(reset=True)
                rows = []
                cur = inp0.copy()
                guard = dict(max_dev=0.0, tripped=0)
                for p in range(P):
                    before = cur.copy() if prec == SPLIT else outs[SPLIT][p][0]
                    work = before.copy()
                    r = su.engine.step(work, SEED_LEN + 3 + p, K, hp, want=("idxs", "final_score", "best"))
                    rows.append((before, r))
                    cur = work
                    if prec == REFINE:  # per step: images beyond 8e-4 against the split engine must all have tripped
                        gs = su.engine.refine_guard(reset=True)
                        guard = dict(max_dev=max(guard["max_dev"], gs["max_dev"]), tripped=guard["tripped"] + gs["tripped"])
                        over = int((np.abs(outs[SPLIT][p][1]["final_score"] - r["final_score"]).max(axis=1) > 8e-4).sum())
                        assert gs["tripped"] >= over, (F_, p, over, gs)
                outs[prec] = rows
            finally:
                su.engine.close()
        worst = 0.0
        for (_, a), (_, b) in zip(outs[SPLIT], outs[REFINE]):
            np.testing.assert_array_equal(a["idxs"], b["idxs"])
            worst = max(worst, float(np.abs(a["final_score"] - b["final_score"]).max()))
        OUTLIER_LOG.append((F_, worst, guard["max_dev"], guard["tripped"], B * P))
        if F_ == 1.0:
            assert worst < 7e-4 and guard["tripped"] == 0, (worst, guard)
        if worst >= 1e-3:
            assert guard["tripped"] > 0, (F_, worst, guard)
        seen_trip |= guard["tripped"] > 0
    assert seen_trip, OUTLIER_LOG  # the emulated outliers are strong enough to exercise the guard at all


DEDUP_LOG = []  # (precision, images, de-duplicated candidate sequences, candidate sequences, rows with / without the option)
SOFT_LOG = []   # (case, precision, steps with a near-tie winner flip, steps)
DRAW_LOG = []  # (draw, worst |d final| czc_step, guard max_dev in czc_generate, images with identical ids, images, tripped)


@pytest.mark.parametrize("bseed,cseed,gain", [(21, 22, 1.0), (31, 32, 1.0), (41, 42, 1.0), (51, 52, 1.0), (11, 12, 12.0), (61, 62, 6.0)])
def test_refine_heuristics_hold_on_other_weight_draws(bseed, cseed, gain):
    """Every constant of the screen-then-refine engine (margin-gate delta 4e-4, guard trip point 2e-4, theta_gen, their x1.75 on
    fp16 rows) was fitted on ONE weight draw (seeds 11 / 12).  Per further draw -- four other seed pairs, the x12 outlier tower
    of seeds 11 / 12 and a x6 outlier tower of a fifth pair -- against the all-split engine (pinned to the reference within
    8e-6): (i) czc_step, 16 images x 6 positions: same candidate lists, every one of the 19 200 fused scores within the 1e-3
    bar OR that image-step tripped the guard; (ii) czc_generate free-running over two sweeps with the margin gate on: ids
    identical to the split engine's for every image, or the guard tripped (runtime.run_generation then repeats the call on the
    split engine).  tools/refine_validate.py runs the same draws at 128 images x 10 sweeps (profiles/r06_refine_validate_*)."""
    B, L, K, P, SCALE = 16, 10, 200, 6, 4.6052
    hp = Engine.hyper(0.02, 2.0, 0.1)
    rng = np.random.default_rng(1000 + bseed)
    emb = rng.standard_normal((B, 512)).astype(np.float32)
    ccfg = synth.clip_b32()
    ccfg.logit_scale = SCALE
    cw = harness.outlier_clip_weights(ccfg, cseed, gain)
    outs, gen = {}, {}
    inp0 = None
    gpos, gnm, gevery = harness.order_positions("sequential", L, 2)
    step_trips = []
    engines = {}
    try:
      for prec in (SPLIT, REFINE):
            su = harness.build_synthetic(False, prec, bseed=bseed, cseed=cseed, logit_scale=SCALE, regular_only=True, clip_w=cw, clip_cfg=ccfg)
            engines[prec] = su
            if inp0 is None:
                inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
                regular = np.nonzero(su.token_mask[0] > 0)[0]
                inp0[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
            su.engine.set_image_embeds(emb)
            rows, cur = [], inp0.copy()
            for p in range(P):
                before = cur.copy() if prec == SPLIT else outs[SPLIT][p][0]
                work = before.copy()
                if prec == REFINE:
                    su.engine.refine_guard(reset=True)
                r = su.engine.step(work, SEED_LEN + 2 + p, K, hp, want=("idxs", "final_score", "best"))
                if prec == REFINE:
                    step_trips.append(su.engine.refine_guard(reset=True))
                rows.append((before, r))
                cur = work
            outs[prec] = rows
            init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
            if prec == REFINE:
                su.engine.refine_guard(reset=True)
            ids, cos = su.engine.generate(B, init, L, SEED_LEN, K, gpos, hp, n_mask=gnm, snapshot_every=gevery)
            gen[prec] = (ids, cos, su.engine.refine_guard(reset=True) if prec == REFINE else None)
      same = (gen[SPLIT][0] == gen[REFINE][0]).all(axis=(0, 2))
      # an image that left the split engine's trajectory: replayed alone on the split engine up to the first differing token
      init = engines[SPLIT].bert_tok.encode("Image of a" + engines[SPLIT].bert_tok.mask_token * L)
      div = [harness.first_divergence(engines[SPLIT].engine, emb[b_], init, gen[SPLIT][0][:, b_], gen[REFINE][0][:, b_], L, SEED_LEN, K, hp)
             for b_ in np.nonzero(~same)[0]]
    finally:
        for su_ in engines.values():
            su_.engine.close()
    worst = 0.0
    for p, ((_, a), (_, b)) in enumerate(zip(outs[SPLIT], outs[REFINE])):
        np.testing.assert_array_equal(a["idxs"], b["idxs"])
        d = np.abs(a["final_score"] - b["final_score"]).max(axis=1)
        over = int((d >= 1e-3).sum())
        assert step_trips[p]["tripped"] >= over, (bseed, cseed, gain, p, over, step_trips[p], float(d.max()))
        if not over:
            worst = max(worst, float(d.max()))
    g = gen[REFINE][2]
    DRAW_LOG.append((f"bert {bseed} clip {cseed} outlier x{gain:g}", worst, g["max_dev"], int(same.sum()), B, g["tripped"]))
    print(f"[draw] bert {bseed} clip {cseed} outlier x{gain:g}: czc_step worst |d final| {worst:.3e} (untripped image-steps); czc_generate "
          f"{int(same.sum())}/{B} images identical, guard max_dev {g['max_dev']:.3e}, tripped {g['tripped']}")
    # an image may only leave the trajectory at a near-tie of the split engine itself: its winner and the token the refine engine
    # wrote are inside the fused-score bar of each other there (the rule the half-precision engines are held to as well)
    near_tie = all(d_.get("gap_to_other_engines_choice") is not None and d_["gap_to_other_engines_choice"] < 1e-3 for d_ in div)
    assert same.all() or g["tripped"] > 0 or near_tie, (bseed, cseed, gain, int(same.sum()), g, div)
    if gain == 1.0:  # plain draws: the heuristics must simply hold, without the guard's help
        assert worst < 1e-3 and (same.all() or near_tie) and g["tripped"] == 0 and sum(t["tripped"] for t in step_trips) == 0, (worst, g, step_trips, div)
        np.testing.assert_allclose(gen[REFINE][1][:, same], gen[SPLIT][1][:, same], atol=2e-5)


def test_refine_engine_encode_text_and_images_are_exact():
    """Outside the polishing step the refine engine answers with its exact towers: czc_encode_text through the split-fp16
    text tower, czc_encode_images through the split-fp16 vision tower (compute_image_text_similarity_via_* callers)."""
    from oracle import models as M
    meta = dict(tiny=True, bseed=11, cseed=12, logit_scale=4.6052, regular_only=False, gamma=None)
    su = setup_for(meta, REFINE)
    ccfg = su.clip_cfg
    rng = np.random.default_rng(1)
    lens = np.array([2, 5, 77, 16, 9], np.int32)
    ids = np.full((len(lens), 77), ccfg.eos_id, np.int32)
    for r, L in enumerate(lens):
        ids[r, 0] = ccfg.bos_id
        ids[r, 1:L - 1] = rng.integers(0, ccfg.vocab - 2, size=L - 2)
    out = su.engine.encode_text(ids, lens)
    w = M.to_torch(synth.make_clip_weights(ccfg, 12))
    ref = M.clip_text_embeds(w, ccfg, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(lens)).numpy()
    assert np.abs(out - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())
    z = np.load(f"{harness.__file__.rsplit('/', 2)[0]}/tests/golden/vision_tiny.npz")
    emb = su.engine.encode_images(synth.pixels_from_u8(synth.make_images_u8(3, ccfg.v_image)))
    assert np.abs(emb - z["image_embeds"]).max() < 5e-5 * max(1.0, np.abs(z["image_embeds"]).max())


@pytest.mark.parametrize("name", [n for n in TINY if n != "tiny_scale100"])
def test_step_parity_tiny_fp16(name):
    meta, arr = load_case(name)
    soft, n = teacher_forced(meta, arr, FP16)
    SOFT_LOG.append((name, "fp16", soft, n))
    print(f"[soft] {name} fp16: {soft}/{n} steps with a near-tie winner flip")
    assert soft <= max(1, n // 4)


def test_precision_selected_from_logit_scale():
    """conzic_amd.runtime.choose_precision: bf16 towers only where exp(logit_scale) keeps them inside the
    budget; the published checkpoints' scale (100) gets the split-fp16 engine."""
    from conzic_amd import runtime
    import os
    old = os.environ.pop("CZC_PRECISION", None)
    try:
        assert runtime.choose_precision(2.6592) == BF16
        assert runtime.choose_precision(3.4) == FP16      # x30: bf16 out of budget, fp16 (8x smaller error) inside
        assert runtime.choose_precision(4.6052) == REFINE  # published checkpoints: screen-then-refine
        assert runtime.choose_precision(None) == SPLIT
        os.environ["CZC_PRECISION"] = "split"
        assert runtime.choose_precision(4.6052) == SPLIT
        os.environ["CZC_PRECISION"] = "f32"
        assert runtime.choose_precision(4.6052) == F32
    finally:
        os.environ.pop("CZC_PRECISION", None)
        if old is not None:
            os.environ["CZC_PRECISION"] = old


@pytest.mark.parametrize("name", ["full_synth_b2", "full_regular"])
def test_step_parity_full_size_bf16_per_image_attention(name):
    """Same bar with the per-image persistent branch attention kernel forced on (the B = 2 goldens are below
    the batch size at which the engine selects it)."""
    meta, arr = load_case(name)
    lib = native.load_test()
    assert lib.czc_test_set_option(b"attention_image", 2) == 0
    try:
        teacher_forced(meta, arr, BF16, n_steps=10)
    finally:
        lib.czc_test_set_option(b"attention_image", 1)


@pytest.mark.parametrize("name", TINY)
def test_generate_free_running_tiny_f32(name):
    """czc_generate (no host round trips) reproduces the reference trajectory id-for-id."""
    meta, arr = load_case(name)
    su = setup_for(meta, F32)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                      control="pos" if meta.get("pos") else None)
    init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
    pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"],
                                             random_positions=meta["positions"])
    assert pos == meta["positions"]
    ids, cos = eng.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
    np.testing.assert_array_equal(ids, arr["snaps"])
    np.testing.assert_allclose(cos, np.array(meta["scores"][:-1], dtype=np.float32), atol=1e-5)
    texts = [su.bert_tok.batch_decode(s, skip_special_tokens=True) for s in ids]
    assert texts == meta["texts"][:-1]


@pytest.mark.parametrize("name", ["full_scale100", "full_senti", "full_shuffle_k512", "full_pos", "full_span", "full_random"])
def test_generate_free_running_full_size_split(name):
    """czc_generate on full-size towers in the split-fp16 precision reproduces the reference's trajectory
    id-for-id (published-checkpoint logit scale; sentiment control at configs[4] shape; K=512 shuffle at
    configs[3] shape)."""
    meta, arr = load_case(name)
    su = setup_for(meta, SPLIT)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                      control="pos" if meta.get("pos") else None)
    init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
    pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"],
                                             random_positions=meta["positions"] if meta["order"] == "random" else None)
    assert pos == meta["positions"]
    ids, cos = eng.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
    np.testing.assert_array_equal(ids, arr["snaps"])
    np.testing.assert_allclose(cos, np.array(meta["scores"][:-1], dtype=np.float32), atol=2e-5)
    texts = [su.bert_tok.batch_decode(s, skip_special_tokens=True) for s in ids]
    assert texts == meta["texts"][:-1]


DIVERGENCE_LOG = []  # (case, precision, tokens compared, identical, final captions identical / images, margins at first divergence)


@pytest.mark.parametrize("prec", [BF16, FP16])
@pytest.mark.parametrize("name", ["full_regular", "full_synth_b2", "full_senti", "full_shuffle_k512", "full_cfg1", "full_pos"])
def test_free_running_half_precision_vs_reference_trajectory(name, prec):
    """The headline precisions FREE-RUNNING (the winner is written back and read by the next step, gen_utils.py:77-81)
    against the reference's trajectory, step by step: an image may only leave the reference's trajectory at a step where
    the reference's own top-2 margin is inside twice the fused-score bar (a near-tie), never at a clear decision; the
    fraction of identical tokens / final captions and the margin at every first divergence go to the run summary."""
    meta, arr = load_case(name)
    assert bf16_in_budget(meta)
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                      control="pos" if meta.get("pos") else None)
    n = arr["inp_before"].shape[0]
    assert not any(meta["reuse"][:n])
    B = arr["inp_before"].shape[1]
    inp = np.ascontiguousarray(arr["inp_before"][0], dtype=np.int32)
    alive = np.ones(B, bool)
    same_tok = tot_tok = 0
    margins = []
    for i in range(n):
        pos = meta["positions"][i]
        gen_idx = SEED_LEN + pos
        eng.step(inp, gen_idx, meta["K"], hp, n_mask=1, dot_allowed=(pos == meta["L"] - 1), want=("best",))
        after = arr["inp_before"][i + 1] if i + 1 < n else None
        if after is None:
            if n != len(meta["positions"]):
                break  # golden holds a prefix of the trajectory only
            after = arr["snaps"][-1]
        gfin = None
        for b in range(B):
            if not alive[b]:
                continue
            tot_tok += 1
            if int(inp[b, gen_idx]) == int(after[b, gen_idx]):
                same_tok += 1
                continue
            if gfin is None:
                gfin = gold_final(meta, arr, i, su)
            srt = np.sort(gfin[b])[::-1]
            margins.append(float(srt[0] - srt[1]))
            alive[b] = False  # off the reference's trajectory from here on: later steps of this image are not comparable
    DIVERGENCE_LOG.append((name, prec, tot_tok, same_tok, int(alive.sum()), B, margins))
    assert all(m < 2e-3 for m in margins), margins


@pytest.mark.parametrize("prec", [F32, BF16, SPLIT])
@pytest.mark.parametrize("label", ["tiny", "full"])
def test_vision_tower(prec, label):
    z = np.load(f"{harness.__file__.rsplit('/', 2)[0]}/tests/golden/vision_{label}.npz")
    meta = dict(tiny=label == "tiny", bseed=11, cseed=12, logit_scale=2.6592, regular_only=False, gamma=None)
    su = setup_for(meta, prec)
    pix = synth.pixels_from_u8(synth.make_images_u8(3, su.clip_cfg.v_image))
    emb = su.engine.encode_images(pix)
    ref = z["image_embeds"]
    err = np.abs(emb - ref).max()
    scale = np.abs(ref).max()
    assert err < ({F32: 3e-5, SPLIT: 5e-5}.get(prec, 3e-2)) * max(1.0, scale), (err, scale)
    cosv = (emb * ref).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(ref, axis=1)
    assert (cosv > (0.9995 if prec == BF16 else 0.999999)).all()


@pytest.mark.parametrize("prec", [F32, BF16])
def test_encode_text_ragged_lengths(prec):
    """CLIP text tower on packed ragged sequences (empty caption = BOS,EOS; longest = 77)."""
    from oracle import models as M
    meta = dict(tiny=True, bseed=11, cseed=12, logit_scale=2.6592, regular_only=False, gamma=None)
    su = setup_for(meta, prec)
    ccfg = su.clip_cfg
    rng = np.random.default_rng(0)
    lens = np.array([2, 3, 77, 16, 15, 40, 2, 64], np.int32)
    ids = np.full((len(lens), 77), ccfg.eos_id, np.int32)
    for r, L in enumerate(lens):
        ids[r, 0] = ccfg.bos_id
        ids[r, 1:L - 1] = rng.integers(0, ccfg.vocab - 2, size=L - 2)
    out = su.engine.encode_text(ids, lens)
    w = M.to_torch(synth.make_clip_weights(ccfg, 12))
    ref = M.clip_text_embeds(w, ccfg, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(lens)).numpy()
    assert np.abs(out - ref).max() < (3e-5 if prec == F32 else 5e-2)


def test_step_rejects_bad_arguments():
    meta = dict(tiny=True, bseed=11, cseed=12, logit_scale=2.6592, regular_only=False, gamma=None)
    su = setup_for(meta, F32)
    su.engine.set_image_embeds(np.ones((2, su.clip_cfg.proj), np.float32))
    inp = np.array([su.bert_tok.encode("Image of a" + "[MASK]" * 3)] * 2, np.int32)
    hp = Engine.hyper(0.02, 2.0, 0.1)
    with pytest.raises(native.NativeError):
        su.engine.step(inp, 99, 8, hp)
    with pytest.raises(native.NativeError, match="lexicon"):
        e2 = harness.build_synthetic(True, F32)
        e2.engine.set_image_embeds(np.ones((2, su.clip_cfg.proj), np.float32))
        e2.engine.step(inp.copy(), 4, 8, Engine.hyper(0.02, 2.0, 0.1, gamma=5.0))


@pytest.fixture
def one_gemm_family():
    """The A/B equivalence tests below compare two runs of the SAME step whose CLIP-text row counts differ.  The
    weight-stationary GEMM only takes M >= 2048 and sums k in two interleaved chains, so letting it serve one run
    and the tiled kernel the other would add fp32 summation-order noise that has nothing to do with what these
    tests check (prefix sharing / packing / pooling are the same math).  Its own parity is in test_kernels_gpu.py
    and every golden-vector test in this file runs with it enabled."""
    lib = native.load_test()
    assert lib.czc_test_set_option(b"wreg", 0) == 0
    yield
    lib.czc_test_set_option(b"wreg", 1)


@pytest.mark.parametrize("prec", [F32, BF16, REFINE])
@pytest.mark.parametrize("name", ["tiny_shuffle", "full_synth_b2"])
def test_prefix_sharing_is_exact(prec, name, one_gemm_family):
    """Encoding the candidates' common causal prefix once (trunk + branches) must not change the
    result: same step with share_prefix on and off."""
    meta, arr = load_case(name)
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    outs = []
    for share in (1, 0):
        eng.set_option("share_prefix", share)
        eng.profile_reset()
        rows = []
        for i in (0, 3, arr["probs"].shape[0] - 1):
            inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
            r = eng.step(inp, SEED_LEN + meta["positions"][i], meta["K"], hp,
                         dot_allowed=(meta["positions"][i] == meta["L"] - 1))
            rows.append((r, inp.copy()))
        outs.append((rows, eng.stats()["clip_rows"]))
    eng.set_option("share_prefix", 1)
    (a, rows_a), (b, rows_b) = outs
    assert rows_a < rows_b, "sharing must push fewer rows through the CLIP text tower"
    for (ra, ia), (rb, ib) in zip(a, b):
        np.testing.assert_array_equal(ra["clip_ids"], rb["clip_ids"])
        # f32: identical up to summation order.  bf16: a different key-slot order inside the PV MFMA can
        # flip a bf16 rounding of the context vector, which is ordinary bf16 noise downstream.
        np.testing.assert_allclose(ra["clip_ref"], rb["clip_ref"], atol=1e-6 if prec == F32 else 1e-3)
        np.testing.assert_allclose(ra["final_score"], rb["final_score"], atol=1e-6 if prec == F32 else 2e-4)
        if prec == F32:
            np.testing.assert_array_equal(ia, ib)


def _peaky_bert_weights(bcfg, n_hot, seed=5):
    """BERT weights whose MLM head behaves like a TRAINED model's at tau = 0.1 (tests/test_kernels_gpu.py::
    test_topk_with_a_trained_models_logit_range): `n_hot` regular tokens sit 5.6..16 logit units above a bulk that underflows
    to exactly zero in softmax(logits / 0.1), so fewer than K probabilities are non-zero and the tail of the top-K is
    zero-probability fill-ins -- which gen_utils.py:72 (`idxs * mask[idxs]`) turns into [PAD], i.e. K - n_hot copies of the same
    caption without the word."""
    w = synth.make_bert_weights(bcfg, 11)
    sv = harness.cached_vocab(False)
    regular = np.nonzero(synth.make_token_mask(sv, regular_only=True)[0] > 0)[0]
    hot = np.random.default_rng(seed).choice(regular, size=n_hot, replace=False)
    bias = w["cls.predictions.bias"].copy()
    bias[hot] += np.linspace(16.0, 5.6, n_hot).astype(np.float32)
    w["cls.predictions.bias"] = bias
    if "cls.predictions.decoder.bias" in w:
        w["cls.predictions.decoder.bias"] = bias
    return w


@pytest.mark.parametrize("prec,B", [(F32, 2), (BF16, 2), (BF16, 64), (SPLIT, 2), (REFINE, 64)])
def test_dedup_is_exact(prec, B):
    """Exact de-duplication of identical candidate sentences (option "dedup", bridge.hip prefix_plan_kernel): with a
    trained-like MLM head 150 of the K = 200 probabilities are non-zero, the other 50 candidates all decode to the caption
    without the word -- one CLIP id row.  All but the first of them per image ride on the first one's rows: fewer rows through
    the text tower (czc_dedup_stats says how many).  What "exact" means per precision is spelled out at the asserts: a
    de-duplicated candidate carries its representative's feature bit for bit; candidates in front of the first removed one are
    bit-identical to the run without the option; the f32 engine is bit-identical throughout (ids, cosines, fused scores,
    winner, write-back); in the MFMA engines the copies of one sentence did not agree among THEMSELVES without the option
    (slot-dependent fp32 association inside the attention tile) and now carry one value inside that noise.  Both attention
    forms (B = 2: one wave per candidate group; B = 64: the per-image kernel) and the two-pass engine."""
    L, K, n_hot = 10, 200, 150
    bcfg = synth.bert_base()
    scale = 4.6052 if prec == REFINE else 2.6592
    su = harness.build_synthetic(False, prec, logit_scale=scale, regular_only=True, bert_w=_peaky_bert_weights(bcfg, n_hot), bert_cfg=bcfg)
    eng = su.engine
    try:
        rng = np.random.default_rng(B)
        eng.set_image_embeds(rng.standard_normal((B, 512)).astype(np.float32))
        hp = Engine.hyper(0.02, 2.0, 0.1)
        inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp0[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
        want = ("probs", "idxs", "cand_ids", "clip_ids", "clip_len", "clip_score", "clip_ref", "final_score", "best", "best_cos")
        outs = {}
        for dd in (1, 0):
            eng.set_option("dedup", dd)
            assert eng.get_option("dedup") == dd
            eng.profile_reset()
            rows = []
            for pos in (0, 4, L - 1):
                inp = inp0.copy()
                r = eng.step(inp, SEED_LEN + pos, K, hp, dot_allowed=(pos == L - 1), want=want)
                rows.append((r, inp))
            outs[dd] = (rows, eng.stats())
        (a, sa), (b, sb) = outs[1], outs[0]
        pads = [int((r["cand_ids"] == 0).sum()) for r, _ in a]
        assert all(p_ >= B * (K - n_hot - 2) for p_ in pads), pads            # the construction: ~50 [PAD] candidates per image
        assert sb["dedup_seqs"] == 0 and sa["dedup_seqs"] >= sum(p_ - B for p_ in pads)
        assert sa["clip_seqs"] == sb["clip_seqs"] == 3 * B * K
        assert sa["clip_rows"] < sb["clip_rows"]
        DEDUP_LOG.append((prec, B, sa["dedup_seqs"], sa["clip_seqs"], sa["clip_rows"], sb["clip_rows"]))
        print(f"[dedup] prec {prec} B {B}: {sa['dedup_seqs']} of {sa['clip_seqs']} candidates ride on an identical one; "
              f"text-tower rows {sa['clip_rows']} vs {sb['clip_rows']} ({sa['clip_rows'] / sb['clip_rows']:.3f})")
        cos_tol = {F32: 0.0, SPLIT: 2e-6, BF16: 1e-3, REFINE: 4e-4}[prec]      # packing noise of the precision (test_prefix_sharing_is_exact's bars)
        fin_tol = {F32: 0.0, SPLIT: 2e-6, BF16: 2e-4, REFINE: 1e-3}[prec]
        for (ra, ia), (rb, ib) in zip(a, b):
            for k in ("probs", "idxs", "cand_ids", "clip_ids", "clip_len"):
                np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
            ids = ra["clip_ids"].reshape(B, K, -1)
            for j in range(B):
                # representative of every candidate, from the ids the engine returned: first candidate with the same row
                first, rep = {}, np.empty(K, np.int64)
                for k in range(K):
                    rep[k] = first.setdefault(ids[j, k].tobytes(), k)
                dup = rep != np.arange(K)
                assert dup.sum() >= K - n_hot - 2
                # (1) a de-duplicated candidate carries its representative's cosine, bit for bit (two-pass engine: its screening
                #     cosine; the final one differs where only one of the two was re-encoded by the second pass)
                if prec == REFINE:
                    np.testing.assert_allclose(ra["clip_ref"][j][dup], ra["clip_ref"][j][rep[dup]], atol=cos_tol, rtol=0)
                else:
                    np.testing.assert_array_equal(ra["clip_ref"][j][dup], ra["clip_ref"][j][rep[dup]], err_msg="(1) dup == rep")
                # (2) every candidate in front of the first removed one has the rows AND the place inside its attention tile it
                #     had without the option: bit-identical cosine in every precision
                head = np.arange(K) < np.argmax(dup)
                #     (f32 and bf16: every kernel that can serve a layer gives the same bits at any row count.  The split-fp16
                #     engine's fp32-output layers pick their K split by the row count -- another fp32 summation order, 1e-7 -- and
                #     in the two-pass engine the second pass's sample depends on the whole score row: those two within tolerance)
                if prec in (F32, BF16):
                    np.testing.assert_array_equal(ra["clip_ref"][j][head], rb["clip_ref"][j][head], err_msg="(2) head candidates")
                else:
                    np.testing.assert_allclose(ra["clip_ref"][j][head], rb["clip_ref"][j][head], atol=cos_tol, rtol=0, err_msg="(2) head candidates")
                # (3) WITHOUT the option identical sentences do not even agree among themselves in the MFMA engines: a
                #     candidate's softmax sum is associated by its slot inside the 32-query tile (in-lane, attention.hip), so each
                #     copy rounds on its own.  With it they all carry one value, inside that packing noise of their own
                spread_off = np.abs(rb["clip_ref"][j][dup] - rb["clip_ref"][j][rep[dup]]).max()
                assert spread_off <= cos_tol, ("(3) spread among copies without the option", prec, float(spread_off))
                np.testing.assert_allclose(ra["clip_ref"][j], rb["clip_ref"][j], atol=cos_tol, rtol=0)
            np.testing.assert_allclose(ra["final_score"], rb["final_score"], atol=fin_tol, rtol=0)
            if prec == F32:   # exact arithmetic per row and no slot-dependent association: everything, bit for bit
                for k in want:
                    np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
                np.testing.assert_array_equal(ia, ib)
            else:             # same winner unless the two runs' own top-2 margin is inside the packing noise
                srt = np.sort(rb["final_score"], axis=1)
                clear = (srt[:, -1] - srt[:, -2]) > 2 * fin_tol
                np.testing.assert_array_equal(ra["best"][clear], rb["best"][clear])
    finally:
        eng.set_option("dedup", 1)
        eng.close()


def test_dedup_leaves_the_flat_softmax_goldens_alone():
    """On the random-weight towers of the goldens every one of the K probabilities is non-zero and no two candidates decode
    to the same string: nothing is de-duplicated (the headline bench is unaffected) and the reference's trajectory is
    reproduced with the option on (every other test) and off (here)."""
    meta, arr = load_case("full_regular")
    su = setup_for(meta, F32)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    try:
        for dd in (1, 0):
            eng.set_option("dedup", dd)
            eng.profile_reset()
            inp = np.ascontiguousarray(arr["inp_before"][3], dtype=np.int32)
            r = eng.step(inp, SEED_LEN + meta["positions"][3], meta["K"], hp, want=("clip_ref", "best"))
            assert eng.stats()["dedup_seqs"] == 0
            np.testing.assert_allclose(r["clip_ref"], arr["clip_ref"][3], atol=5e-6)
    finally:
        eng.set_option("dedup", 1)


@pytest.mark.parametrize("prec,B", [(BF16, 8), (BF16, 64), (F32, 8)])
def test_bert_fc2_layernorm_sums_the_split_k_slabs_itself(prec, B):
    """BERT's fc2 (K = 3072) is split along K at every row count above 32; the slice sums used to be reduced by one kernel and
    normalised by the next.  Now the LayerNorm kernel sums the slabs itself (option bert_fuse_splitk_ln, GemmArgs::splitk_pending):
    same additions in the same order, same ln_row -- the masked-row logits and everything behind them are bit-identical to the
    two-kernel form (the f32 engine's fc2 is not split: the option is a no-op there)."""
    L, K = 10, 50
    su = harness.build_synthetic(False, prec, regular_only=True)
    eng = su.engine
    try:
        rng = np.random.default_rng(B)
        eng.set_image_embeds(rng.standard_normal((B, 512)).astype(np.float32))
        hp = Engine.hyper(0.02, 2.0, 0.1)
        inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp0[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
        outs = []
        for on in (1, 0):
            eng.set_option("bert_fuse_splitk_ln", on)
            assert eng.get_option("bert_fuse_splitk_ln") == on
            for prune in (1, 0):   # the pruned last layer (B rows) and the full one (B * T rows) both take the fused form
                eng.set_option("bert_prune", prune)
                inp = inp0.copy()
                r = eng.step(inp, SEED_LEN + 3, K, hp, want=("logits", "probs", "idxs", "final_score", "best"))
                outs.append((on, prune, r, inp))
        for (_, p1, a, ia), (_, p0, b, ib) in zip(outs[:2], outs[2:]):
            assert p1 == p0
            for k in ("logits", "probs", "idxs", "final_score", "best"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            np.testing.assert_array_equal(ia, ib)
    finally:
        eng.set_option("bert_fuse_splitk_ln", 1)
        eng.set_option("bert_prune", 1)
        eng.close()


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("name", ["tiny_shuffle", "full_synth_b2", "full_senti"])
def test_last_bert_layer_on_the_masked_row_only_is_exact(prec, name):
    """The MLM head reads one row per sequence (gen_utils.py:69), so behind the last layer's attention only that row
    goes through out-projection / LayerNorm / MLP / LayerNorm (option bert_prune, on by default for n_mask == 1 steps).
    Same per-row arithmetic; the B-row GEMMs may take another K split than the B*T-row ones -> fp32 summation order."""
    meta, arr = load_case(name)
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative")
    outs = []
    for prune in (1, 0):
        eng.set_option("bert_prune", prune)
        rows = []
        for i in (0, 2, arr["probs"].shape[0] - 1):
            if meta["reuse"][i]:
                continue
            inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
            r = eng.step(inp, SEED_LEN + meta["positions"][i], meta["K"], hp, dot_allowed=(meta["positions"][i] == meta["L"] - 1),
                         want=("logits", "probs", "idxs", "final_score", "best"))
            rows.append((r, inp.copy()))
        outs.append(rows)
    eng.set_option("bert_prune", 1)
    assert outs[0]
    for (ra, ia), (rb, ib) in zip(*outs):
        scale = float(np.abs(rb["logits"]).max())
        assert float(np.abs(ra["logits"] - rb["logits"]).max()) <= 2e-6 * max(scale, 1.0) + 1e-6
        np.testing.assert_array_equal(ra["idxs"], rb["idxs"])
        np.testing.assert_allclose(ra["probs"], rb["probs"], rtol=2e-4, atol=1e-7)
        np.testing.assert_array_equal(ia, ib)


def test_reuse_of_a_one_row_forward_for_another_row_is_refused():
    """n_mask = 0 re-uses the previous BERT forward for the second slot of a span (gen_utils.py:164-166) and follows an
    n_mask = 2 step, which keeps every row.  After an n_mask = 1 step only the masked row exists: asking for another
    row must fail loudly, the same row must still work."""
    meta, arr = load_case("tiny_seq")
    su = setup_for(meta, F32)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    inp = np.ascontiguousarray(arr["inp_before"][0], dtype=np.int32)
    a = eng.step(inp.copy(), SEED_LEN + 1, meta["K"], hp, n_mask=1, want=("logits",))
    b = eng.step(inp.copy(), SEED_LEN + 1, meta["K"], hp, n_mask=0, want=("logits",))
    np.testing.assert_array_equal(a["logits"], b["logits"])
    with pytest.raises(native.NativeError, match="kept one row"):
        eng.step(inp.copy(), SEED_LEN + 2, meta["K"], hp, n_mask=0)
    two = eng.step(inp.copy(), SEED_LEN + 1, meta["K"], hp, n_mask=2, want=("logits",))  # a span's first step keeps all rows
    eng.step(inp.copy(), SEED_LEN + 2, meta["K"], hp, n_mask=0)
    assert np.isfinite(two["logits"]).all()


@pytest.mark.parametrize("kernel", ["per_group", "per_image"])
@pytest.mark.parametrize("name", ["tiny_shuffle", "full_synth_b2"])
def test_packed_branch_attention_matches_per_segment(name, kernel, one_gemm_family):
    """bf16 engine: packing G candidates into one attention tile vs one wave per candidate -- through the
    per-(group, head) kernel and through the per-image persistent kernel (which the engine only picks by itself
    once the batch fills the chip, B >= 64; forced here)."""
    meta, arr = load_case(name)
    if kernel == "per_image" and meta["tiny"]:
        pytest.skip("needs heads % 4 == 0")
    lib = native.load_test()
    assert lib.czc_test_set_option(b"attention_image", 2 if kernel == "per_image" else 0) == 0
    try:
        _packed_vs_per_segment(meta, arr)
    finally:
        lib.czc_test_set_option(b"attention_image", 1)


def _packed_vs_per_segment(meta, arr):
    su = setup_for(meta, BF16)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    outs = []
    for pack in (1, 0):
        eng.set_option("pack_branches", pack)
        rows = []
        for i in (0, 2, arr["probs"].shape[0] - 1):
            inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
            rows.append(eng.step(inp, SEED_LEN + meta["positions"][i], meta["K"], hp,
                                 dot_allowed=(meta["positions"][i] == meta["L"] - 1)))
        outs.append(rows)
    eng.set_option("pack_branches", 1)
    for ra, rb in zip(*outs):
        np.testing.assert_allclose(ra["clip_ref"], rb["clip_ref"], atol=1e-3)
        np.testing.assert_allclose(ra["final_score"], rb["final_score"], atol=2e-4)


@pytest.mark.parametrize("prec", [BF16, FP16])
@pytest.mark.parametrize("name,steps", [("full_synth_b2", (4, 7, 9)), ("full_shuffle_k512", (2, 9)), ("full_senti", (6, 11))])
def test_fused_layernorm_matches_layernorm_kernel(name, steps, prec):
    """CLIP-text path above 8192 packed rows: out-proj (default, fuse_ln = 1) and fc2 (fuse_ln = 2) as full-row kernels
    that also emit the LayerNorm that follows them, against the same steps with the stand-alone LayerNorm kernel: the
    same two-pass statistics on the same fp32 rows, so only last-place flips of the normalised rows remain.  With
    both fused the LayerNorm kernel all but disappears from the profile (layer 0's LN1 and the pooled rows keep it)."""
    lib = native.load_test()
    meta, arr = load_case(name)
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative")
    outs = []
    try:
        assert lib.czc_test_set_option(b"rowln_min_m", 256) == 0
        eng.set_option("resid16", 0)   # the fp32 residual stream (what the fp16 / refine engines run; bf16: the option's 0 arm)
        for fuse in (2, 1, 0):
            eng.set_option("fuse_ln", fuse)
            eng.profile(1)
            eng.profile_reset()
            rows = []
            for i in steps:
                inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
                rows.append(eng.step(inp, SEED_LEN + meta["positions"][i], meta["K"], hp,
                                     dot_allowed=(meta["positions"][i] == meta["L"] - 1)))
            outs.append((rows, eng.profile_get("rowops_clip_text")))
            eng.profile(0)
    finally:
        lib.czc_test_set_option(b"rowln_min_m", 8192)
        eng.set_option("fuse_ln", 1)
        eng.set_option("resid16", 1)
    for fused in outs[:2]:
        for ra, rb in zip(fused[0], outs[2][0]):
            np.testing.assert_array_equal(ra["clip_ids"], rb["clip_ids"])
            assert np.isfinite(ra["final_score"]).all()
            np.testing.assert_allclose(ra["clip_ref"], rb["clip_ref"], atol=1e-3 if prec == BF16 else 2e-4)
            assert np.abs(ra["clip_ref"] - rb["clip_ref"]).mean() < (1e-4 if prec == BF16 else 6e-5)
    n2, n1, n0 = (o[1]["launches"] for o in outs)
    assert n2 < n1 - 9 * len(steps) and n1 < n0 - 9 * len(steps), (n2, n1, n0)


@pytest.mark.parametrize("prec", [F32, BF16])
def test_last_layer_pooling_is_exact(prec, one_gemm_family):
    """Running the last CLIP-text layer's out-projection/MLP on the EOS rows only is the same math."""
    meta, arr = load_case("full_synth_b2")
    su = setup_for(meta, prec)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    outs = []
    for pool in (1, 0):
        eng.set_option("pool_last_layer", pool)
        inp = np.ascontiguousarray(arr["inp_before"][4], dtype=np.int32)
        outs.append(eng.step(inp, SEED_LEN + meta["positions"][4], meta["K"], hp))
    eng.set_option("pool_last_layer", 1)
    # bf16: the 400 pooled rows take the 128x128 GEMM whose quick-GELU uses __expf/__frcp_rn, the full batch the
    # 256x256 one with raw v_exp/v_rcp -- a last-bit difference in fc1, not in what is being tested
    np.testing.assert_allclose(outs[0]["clip_ref"], outs[1]["clip_ref"], atol=1e-6 if prec == F32 else 5e-5)
    np.testing.assert_allclose(outs[0]["final_score"], outs[1]["final_score"], atol=1e-6 if prec == F32 else 5e-5)


@pytest.mark.parametrize("prec", [F32, BF16])
def test_long_ragged_sequences_vs_oracle(prec):
    """Edge cases in one step, checked against the oracle computed on the fly: a long caption (L=11)
    whose words are multi-token 'irregular' words, so CLIP sequences are long and ragged, some hit
    the 77-token truncation (clip/clip.py:71-72), and the branches are too long for the packed
    attention kernel (generic segment path).  Also K > number of unmasked tokens -> [PAD] candidates."""
    from oracle import models as M, step as S, text as T
    meta = dict(tiny=True, bseed=11, cseed=12, logit_scale=2.6592, regular_only=False, gamma=None)
    su = harness.build_synthetic(True, prec)
    try:
        sv = su.sv
        V = len(sv.bert_tokens)
        # only multi-token words, digits and pieces stay allowed; leaves fewer than K usable ids
        mask = np.zeros((1, V), np.float32)
        irregular = [i for i, t in enumerate(sv.bert_tokens) if t.isalpha() and len(t) >= 6 and i < sv.regular_lo]
        mask[0, irregular[:20]] = 1.0
        su.engine.set_token_mask(mask)
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(su.clip_cfg, 12)), su.clip_cfg, sv.bert_tokens,
                     T.ClipBpe(sv.clip_vocab, sv.clip_merges))
        B, L, K = 2, 11, 32
        rng = np.random.default_rng(3)
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        inp[:, 4:4 + L] = rng.choice(irregular, size=(B, L))
        gen_idx = 4 + 3
        inp[:, gen_idx] = su.bert_tok.mask_token_id
        emb = rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32)
        su.engine.set_image_embeds(emb)
        tmask = torch.from_numpy(mask.copy())
        ref_inp = torch.from_numpy(inp.astype(np.int64))
        r = S.polish_step(o, ref_inp, torch.from_numpy(emb), tmask, gen_idx, K, 0.1, 0.02, 2.0)
        res = su.engine.step(inp, gen_idx, K, Engine.hyper(0.02, 2.0, 0.1))
        lens = r["clip_lens"].numpy()
        assert lens.max() == 77 and lens.min() < 77, (lens.min(), lens.max())  # truncation really happens
        np.testing.assert_array_equal(res["idxs"][:, :20], r["idxs"].numpy()[:, :20])
        assert (res["cand_ids"][:, 20:] == 0).all() and (r["idxs_"].numpy()[:, 20:] == 0).all()  # masked -> [PAD]
        np.testing.assert_array_equal(res["clip_len"], lens)
        for row in range(B * K):
            np.testing.assert_array_equal(res["clip_ids"][row, :lens[row]], r["clip_ids"][row, :lens[row]].numpy())
        tol = 3e-5 if prec == F32 else 6e-3
        np.testing.assert_allclose(res["clip_ref"][:, :20], r["clip_ref"].numpy()[:, :20], atol=tol)
        if prec == F32:
            np.testing.assert_allclose(res["final_score"], r["final"].numpy(), atol=3e-5)
            np.testing.assert_array_equal(res["best"], r["best"].numpy())
    finally:
        su.engine.close()


def test_minimal_shapes_k1_l1():
    """K=1 / L=1 / B=1 degenerate shapes run and agree with the oracle."""
    from oracle import models as M, step as S, text as T
    su = harness.build_synthetic(True, F32)
    try:
        sv = su.sv
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(su.clip_cfg, 12)), su.clip_cfg, sv.bert_tokens,
                     T.ClipBpe(sv.clip_vocab, sv.clip_merges))
        inp = np.array(o.init_text("Image of a", 1, 1), dtype=np.int32)
        emb = np.random.default_rng(0).standard_normal((1, su.clip_cfg.proj)).astype(np.float32)
        su.engine.set_image_embeds(emb)
        tmask = torch.from_numpy(su.token_mask.copy())
        o.update_token_mask(tmask, 1, 0)
        r = S.polish_step(o, torch.from_numpy(inp.astype(np.int64)), torch.from_numpy(emb), tmask, 4, 1, 0.1, 0.02, 2.0)
        res = su.engine.step(inp, 4, 1, Engine.hyper(0.02, 2.0, 0.1), dot_allowed=True)
        np.testing.assert_array_equal(res["idxs"], r["idxs"].numpy())
        np.testing.assert_allclose(res["final_score"], r["final"].numpy(), atol=2e-5)
        assert abs(float(res["clip_score"][0, 0]) - 1.0) < 1e-6  # softmax over a single candidate
        assert inp[0, 4] == int(r["inp_after"][0, 4])
    finally:
        su.engine.close()


def test_config3_shape_k512_l15_kernel_families_agree():
    """BASELINE configs[3] shape (K = 512 candidates, L = 15 -> T = 20, half-filled caption) on full-size towers:
    the bf16 engine's three branch-attention routes (per-segment, per-(group,head), per-image persistent) agree
    with each other, and with the oracle within the 1e-3 bar."""
    from oracle import models as M, step as S, text as T
    su = harness.build_synthetic(False, BF16)
    lib = native.load_test()
    try:
        sv = su.sv
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(su.clip_cfg, 12)), su.clip_cfg, sv.bert_tokens,
                     T.ClipBpe(sv.clip_vocab, sv.clip_merges))
        B, L, K = 2, 15, 512
        rng = np.random.default_rng(5)
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp[:, SEED_LEN:SEED_LEN + 9] = rng.choice(regular, size=(B, 9))  # nine positions already filled
        gen_idx = SEED_LEN + 9
        emb = rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32)
        su.engine.set_image_embeds(emb)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        outs = {}
        for name, opts in (("per_image", dict(attention_image=2)), ("per_group", dict(attention_image=0)),
                           ("per_segment", dict(attention_image=0, pack=0))):
            assert lib.czc_test_set_option(b"attention_image", opts.get("attention_image", 1)) == 0
            su.engine.set_option("pack_branches", opts.get("pack", 1))
            outs[name] = su.engine.step(inp.copy(), gen_idx, K, hp)
        tmask = torch.from_numpy(su.token_mask.copy())
        o.update_token_mask(tmask, L, 9)
        r = S.polish_step(o, torch.from_numpy(inp.astype(np.int64)), torch.from_numpy(emb), tmask, gen_idx, K, 0.1, 0.02, 2.0)
        ref = outs["per_segment"]
        # deep in a 512-long list the split-fp16 BERT and the fp32 oracle may order near-equal probabilities
        # differently; compare the fused scores where both hold the same candidate (all but a handful)
        same = ref["idxs"] == r["idxs"].numpy()
        assert same.mean() > 0.98
        for name, res in outs.items():
            np.testing.assert_array_equal(res["idxs"], ref["idxs"], err_msg=name)
            np.testing.assert_allclose(res["final_score"], ref["final_score"], atol=3e-4, err_msg=name)
            np.testing.assert_allclose(res["final_score"][same], r["final"].numpy()[same], atol=1e-3, err_msg=name)
    finally:
        lib.czc_test_set_option(b"attention_image", 1)
        su.engine.close()


@pytest.mark.parametrize("prec", [BF16, SPLIT])
def test_batch16_step_vs_oracle(prec):
    """A batch that takes the big-batch kernels (B = 16 images x K = 200 candidates = 20 k packed CLIP rows:
    weight-stationary + 256x256 GEMMs / the split ring GEMM, packed-branch attention) checked against the ORACLE,
    not against another kernel family: one later-sweep position-step, fused score within the bar of the precision."""
    from oracle import models as M, step as S, text as T
    torch.set_num_threads(min(16, torch.get_num_threads()))  # plain torch oversubscribes badly on a 256-thread host
    su = harness.build_synthetic(False, prec, logit_scale=4.6052 if prec == SPLIT else 2.6592)
    try:
        ccfg = synth.clip_b32()
        ccfg.logit_scale = su.clip_cfg.logit_scale
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(ccfg, 12)), ccfg, su.sv.bert_tokens,
                     T.ClipBpe(su.sv.clip_vocab, su.sv.clip_merges))
        B, L, K = 16, 10, 200
        rng = np.random.default_rng(31)
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
        gen_idx = SEED_LEN + 3
        emb = rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32)
        su.engine.set_image_embeds(emb)
        res = su.engine.step(inp.copy(), gen_idx, K, Engine.hyper(0.02, 2.0, 0.1))
        tmask = torch.from_numpy(su.token_mask.copy())
        o.update_token_mask(tmask, L, 3)
        ref_inp = torch.from_numpy(inp.astype(np.int64))
        ref_inp[:, gen_idx] = o.mask_id
        r = S.polish_step(o, ref_inp, torch.from_numpy(emb), tmask, gen_idx, K, 0.1, 0.02, 2.0)
        same = res["idxs"] == r["idxs"].numpy()
        assert same.mean() > 0.995
        tol = 1e-3 if prec == BF16 else 1e-4
        err = np.abs(res["final_score"] - r["final"].numpy())[same]
        assert err.max() < tol, err.max()
        fin = r["final"].numpy()
        srt = np.sort(fin, axis=1)[:, ::-1]
        clear = (srt[:, 0] - srt[:, 1]) > 2 * tol
        assert (res["best"][clear] == r["best"].numpy()[clear]).all()
    finally:
        su.engine.close()


@pytest.mark.parametrize("prec", [BF16, SPLIT])
def test_maximum_topk_step_vs_oracle(prec):
    """The largest K the boundary accepts (CZC_MAX_TOPK = 1024: the top-K sort buffer, the per-image LDS arrays of the plan /
    de-duplication / combine kernels all sit at their limit) on full-size towers, two images, against the ORACLE; K + 1 is
    refused loudly."""
    from oracle import models as M, step as S, text as T
    torch.set_num_threads(min(16, torch.get_num_threads()))
    su = harness.build_synthetic(False, prec, logit_scale=4.6052 if prec == SPLIT else 2.6592)
    try:
        ccfg = synth.clip_b32()
        ccfg.logit_scale = su.clip_cfg.logit_scale
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(ccfg, 12)), ccfg, su.sv.bert_tokens,
                     T.ClipBpe(su.sv.clip_vocab, su.sv.clip_merges))
        B, L, K = 2, 10, 1024
        rng = np.random.default_rng(1024)
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
        gen_idx = SEED_LEN + 6
        emb = rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32)
        su.engine.set_image_embeds(emb)
        res = su.engine.step(inp.copy(), gen_idx, K, Engine.hyper(0.02, 2.0, 0.1))
        tmask = torch.from_numpy(su.token_mask.copy())
        o.update_token_mask(tmask, L, 6)
        ref_inp = torch.from_numpy(inp.astype(np.int64))
        ref_inp[:, gen_idx] = o.mask_id
        r = S.polish_step(o, ref_inp, torch.from_numpy(emb), tmask, gen_idx, K, 0.1, 0.02, 2.0)
        same = res["idxs"] == r["idxs"].numpy()
        assert same.mean() > 0.99
        tol = 1e-3 if prec == BF16 else 1e-4
        err = np.abs(res["final_score"] - r["final"].numpy())[same]
        assert err.max() < tol, err.max()
        fin = r["final"].numpy()
        srt = np.sort(fin, axis=1)[:, ::-1]
        clear = (srt[:, 0] - srt[:, 1]) > 2 * tol
        assert (res["best"][clear] == r["best"].numpy()[clear]).all()
        with pytest.raises(native.NativeError) as ei:
            su.engine.step(inp.copy(), gen_idx, K + 1, Engine.hyper(0.02, 2.0, 0.1))
        assert ei.value.code == native.ERR_ARG
    finally:
        su.engine.close()


def test_large_batch_kernel_families_agree():
    """B = 64 images x K = 200 candidates (78 k packed CLIP rows, ~58 blocks per work-group of the weight-stationary
    GEMM, the per-image attention kernel selected by the engine itself): the exact-count vmcnt pipelines of
    gemm_wreg / attention_image against the kernels that do not rely on them, on the same step.
    A mis-counted wait would show up here as a handful of wildly wrong rows, not as rounding noise."""
    su = harness.build_synthetic(False, BF16)
    lib = native.load_test()
    try:
        from oracle import step as S, models as M, text as T
        B, L, K = 64, 10, 200
        rng = np.random.default_rng(11)
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(su.clip_cfg, 12)), su.clip_cfg, su.sv.bert_tokens,
                     T.ClipBpe(su.sv.clip_vocab, su.sv.clip_merges))
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        regular = np.nonzero(su.token_mask[0] > 0)[0]
        inp[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))  # a later sweep: every position filled
        gen_idx = SEED_LEN + 6
        su.engine.set_image_embeds(rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32))
        hp = Engine.hyper(0.02, 2.0, 0.1)
        outs = {}
        for name, wreg, att in (("default", 1, 1), ("tiled_gemm", 0, 1), ("per_group_attention", 1, 0)):
            assert lib.czc_test_set_option(b"wreg", wreg) == 0
            assert lib.czc_test_set_option(b"attention_image", att) == 0
            outs[name] = su.engine.step(inp.copy(), gen_idx, K, hp)
        ref = outs["tiled_gemm"]
        for name, res in outs.items():
            np.testing.assert_array_equal(res["idxs"], ref["idxs"], err_msg=name)
            assert np.isfinite(res["final_score"]).all(), name
            # different fp32 summation orders under bf16 roundings: ~1e-4 typical, ~1e-3 in the tail of 12 800 cosines
            np.testing.assert_allclose(res["clip_ref"], ref["clip_ref"], atol=3e-3, err_msg=name)
            # (3e-4 since round 5: the tower's rows live in fp16, and a different summation order in the out-projection moves
            # an element across a rounding boundary now and then)
            assert np.abs(res["clip_ref"] - ref["clip_ref"]).mean() < 3e-4, name
            np.testing.assert_allclose(res["final_score"], ref["final_score"], atol=1e-3, err_msg=name)
        # bit-reproducible from run to run
        again = su.engine.step(inp.copy(), gen_idx, K, hp)
        np.testing.assert_array_equal(again["final_score"], outs["per_group_attention"]["final_score"])
    finally:
        lib.czc_test_set_option(b"wreg", 1)
        lib.czc_test_set_option(b"attention_image", 1)
        su.engine.close()


def test_first_sweep_large_batch_kernel_families_agree():
    """The whole first sweep (captions growing from 0 to L filled positions: trunk length, rows per candidate and the
    packing factor G all change from step to step) at B = 64, K = 200, teacher-forced on the default kernels'
    own write-backs: default kernels vs the tiled-GEMM / per-group-attention family at every position."""
    su = harness.build_synthetic(False, BF16)
    lib = native.load_test()
    try:
        from oracle import step as S, models as M, text as T
        B, L, K = 64, 10, 200
        rng = np.random.default_rng(21)
        o = S.Oracle(M.to_torch(synth.make_bert_weights(su.bert_cfg, 11)), su.bert_cfg,
                     M.to_torch(synth.make_clip_weights(su.clip_cfg, 12)), su.clip_cfg, su.sv.bert_tokens,
                     T.ClipBpe(su.sv.clip_vocab, su.sv.clip_merges))
        inp = np.array(o.init_text("Image of a", L, B), dtype=np.int32)
        su.engine.set_image_embeds(rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32))
        hp = Engine.hyper(0.02, 2.0, 0.1)
        flips = 0
        for pos in range(L):
            gen_idx = SEED_LEN + pos
            res = {}
            for name, wreg, att in (("conservative", 0, 0), ("default", 1, 1)):
                assert lib.czc_test_set_option(b"wreg", wreg) == 0
                assert lib.czc_test_set_option(b"attention_image", att) == 0
                cur = inp.copy()
                res[name] = (su.engine.step(cur, gen_idx, K, hp, dot_allowed=(pos == L - 1)), cur)
            a, b = res["default"][0], res["conservative"][0]
            np.testing.assert_array_equal(a["idxs"], b["idxs"])
            np.testing.assert_array_equal(a["clip_ids"], b["clip_ids"])
            assert np.isfinite(a["final_score"]).all()
            np.testing.assert_allclose(a["final_score"], b["final_score"], atol=1e-3, err_msg=f"position {pos}")
            flips += int((a["best"] != b["best"]).sum())
            inp = res["default"][1]  # the step wrote the winners back
        assert flips <= 3, flips  # near-ties may flip between kernel families in bf16
    finally:
        lib.czc_test_set_option(b"wreg", 1)
        lib.czc_test_set_option(b"attention_image", 1)
        su.engine.close()


@pytest.mark.parametrize("B,L,K,filled", [(3, 7, 100, 7), (5, 9, 37, 4), (1, 25, 64, 25), (7, 10, 200, 0), (2, 15, 1024, 15),
                                          (33, 6, 50, 6), (4, 12, 333, 12)])
def test_odd_shapes_fast_engines_vs_f32_engine(B, L, K, filled):
    """Ragged shapes the goldens do not have -- candidate counts that are no multiple of the attention packing factor
    or of any tile size, K at CZC_MAX_TOPK, a 30-token BERT row (L = 25), a single image, an empty caption (first step of
    the first sweep) -- through the bf16, fp16 and split engines, checked against the f32-MFMA engine (itself pinned to the
    reference by the goldens) on one position-step: same candidates, fused scores within each precision's bar."""
    rng = np.random.default_rng(B * 1000 + K)
    emb = rng.standard_normal((B, 512)).astype(np.float32)
    ref = None
    for prec in (F32, BF16, FP16, SPLIT):
        su = harness.build_synthetic(False, prec)
        try:
            if ref is None:
                inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
                regular = np.nonzero(su.token_mask[0] > 0)[0]
                inp0[:, SEED_LEN:SEED_LEN + filled] = rng.choice(regular, size=(B, filled))
            gen_idx = SEED_LEN + (min(filled, L) // 2 if filled else 0)
            su.engine.set_image_embeds(emb)
            res = su.engine.step(inp0.copy(), gen_idx, K, Engine.hyper(0.02, 2.0, 0.1), dot_allowed=False)
            assert np.isfinite(res["final_score"]).all()
            if prec == F32:
                ref = res
                continue
            same = res["idxs"] == ref["idxs"]
            assert same.mean() > 0.97, (prec, same.mean())  # a near-tie swap of two neighbours in the top-K list counts twice
            np.testing.assert_array_equal(res["clip_len"].reshape(B, K)[same], ref["clip_len"].reshape(B, K)[same])
            tol = {BF16: 1e-3 * max(1.0, 200.0 / K), FP16: 2e-4 * max(1.0, 200.0 / K), SPLIT: 2e-5}[prec]
            err = np.abs(res["final_score"] - ref["final_score"])[same]
            assert err.max() < tol, (prec, err.max())
        finally:
            su.engine.close()


@pytest.mark.parametrize("prec", [BF16, FP16])
def test_caption_does_not_depend_on_the_batch(prec):
    """Images are independent (gen_utils.py:64-81 has no cross-image term): an image polished alone and inside a batch of
    sixteen must produce the SAME caption, token for token and cosine for cosine, with no kernel switches -- although the
    batch takes the full-row / ring GEMMs and the per-image attention kernel never runs at this size while the single
    image takes the tiled GEMM + LayerNorm kernel.  Free-running over two sweeps (a single near-tie flip would show)."""
    su = harness.build_synthetic(False, prec, regular_only=True)
    try:
        B, L, K, I = 16, 6, 200, 2
        rng = np.random.default_rng(404)
        emb = rng.standard_normal((B, 512)).astype(np.float32)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
        pos, nm, every = harness.order_positions("sequential", L, I)
        su.engine.set_image_embeds(emb)
        su.engine.profile_reset()
        ids, cos = su.engine.generate(B, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)
        rows_batch = su.engine.stats()["clip_rows"] / len(pos)
        for b in (0, 5, 11):
            su.engine.set_image_embeds(emb[b:b + 1])
            su.engine.profile_reset()
            ids1, cos1 = su.engine.generate(1, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)
            rows_one = su.engine.stats()["clip_rows"] / len(pos)
            np.testing.assert_array_equal(ids1[:, 0], ids[:, b])
            np.testing.assert_array_equal(cos1[:, 0], cos[:, b])
        assert rows_batch > 8192 > rows_one, (rows_batch, rows_one)  # the two runs really took different kernels
    finally:
        su.engine.close()


@pytest.mark.parametrize("prec", [BF16, SPLIT])
def test_attention_packing_does_not_couple_the_images_of_a_batch(prec):
    """With a real vocabulary words take different numbers of CLIP tokens, so the longest branch differs from image to
    image.  The packed-branch attention takes its packing factor PER IMAGE (SegTable::img_max): an image with short
    branches gets the same tiles -- the same fp32 association, the same bits -- whether or not a long-branch image rides
    in the batch.  Image 1 of this batch carries multi-token ('irregular') words behind the polished position, images 0
    and 2 one-token words; each is compared with itself polished alone."""
    su = harness.build_synthetic(False, prec, logit_scale=2.6592 if prec == BF16 else 4.6052)
    try:
        sv = su.sv
        B, L, K = 3, 8, 64
        rng = np.random.default_rng(9)
        regular = np.arange(sv.regular_lo, sv.regular_hi)
        irregular = np.array([i for i, t in enumerate(sv.bert_tokens) if t.isalpha() and len(t) >= 6 and i < sv.regular_lo])
        assert len(irregular) >= 20
        inp = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
        inp[:, SEED_LEN:SEED_LEN + L] = rng.choice(regular, size=(B, L))
        inp[1, SEED_LEN + 5:SEED_LEN + 7] = rng.choice(irregular, size=2)  # two multi-token words: longer branches, still <= 32 rows
        gen_idx = SEED_LEN + 2
        emb = rng.standard_normal((B, su.clip_cfg.proj)).astype(np.float32)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        su.engine.set_image_embeds(emb)
        full = su.engine.step(inp.copy(), gen_idx, K, hp, want=("idxs", "clip_len", "clip_ref", "final_score", "best"))
        lens = full["clip_len"].reshape(B, K)
        assert lens[0].max() + 4 < lens[1].max() <= 30, (lens[0].max(), lens[1].max())  # really heterogeneous, still packable
        for b in (0, 2):
            su.engine.set_image_embeds(emb[b:b + 1])
            one = su.engine.step(inp[b:b + 1].copy(), gen_idx, K, hp, want=("idxs", "clip_ref", "final_score", "best"))
            np.testing.assert_array_equal(one["idxs"][0], full["idxs"][b])
            np.testing.assert_array_equal(one["clip_ref"][0], full["clip_ref"][b])  # the whole CLIP side: bit for bit
            # the fused score also carries alpha * BERT probabilities: BERT is fp32-class in every engine, and at one or two
            # images (<= 32 rows) its GEMMs run on the K-split skinny kernel, another fp32 summation order (1e-6 relative)
            np.testing.assert_allclose(one["final_score"][0], full["final_score"][b], rtol=0, atol=1e-6)
    finally:
        su.engine.close()

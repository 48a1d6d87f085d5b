"""Two image sub-batches on two HIP streams (czc_replicate + EngineGroup): same captions image for image.

Images are independent units of the polishing loop (gen_utils.py:64-81: no cross-image term), so an engine plus a
replica over the same weights, each driven from its own host thread on half of the images, must reproduce what one
engine produces on all of them -- and therefore the reference's golden trajectories."""
import numpy as np
import pytest
import torch

from conzic_amd import harness, native, synth
from conzic_amd.engine import Engine, EngineGroup, union_ms
from goldutil import load_case

pytestmark = pytest.mark.gpu
SEED_LEN = 4
BF16, F32, SPLIT, REFINE = native.PREC_BF16, native.PREC_F32, native.PREC_SPLIT, native.PREC_REFINE


def _setup(meta, prec):
    su = harness.build_synthetic(meta["tiny"], prec, meta["bseed"], meta["cseed"], meta["logit_scale"], meta["regular_only"],
                                 lexicon=meta["gamma"] is not None)
    if meta.get("pos"):
        su.engine.set_pos(synth.make_pos_tags(len(su.sv.bert_tokens)), synth.pos_template_masks(meta["pos"]))
    return su


@pytest.mark.parametrize("name,prec", [("full_scale100", SPLIT), ("full_scale100", REFINE), ("full_senti", SPLIT), ("full_senti", REFINE),
                                       ("full_pos", SPLIT), ("tiny_shuffle", F32)])
def test_two_streams_reproduce_the_reference_trajectory(name, prec):
    """One image per stream (min_images = 1): the golden snapshots of the imported reference come back id for id,
    including the sentiment / POS tables that have to reach the replica (setter replay)."""
    meta, arr = load_case(name)
    su = _setup(meta, prec)
    grp = EngineGroup(su.engine, streams=2, min_images=1)
    try:
        assert len(grp.parts(meta["B"])) == 2
        grp.set_image_embeds(arr["image_embeds"])
        hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative",
                          control="pos" if meta.get("pos") else None)
        init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * meta["L"])
        pos, nm, every = harness.order_positions(meta["order"], meta["L"], meta["I"], order_list=meta["order_list"],
                                                 random_positions=meta["positions"] if meta["order"] == "random" else None)
        assert pos == meta["positions"]
        ids, cos = grp.generate(meta["B"], init, meta["L"], SEED_LEN, meta["K"], pos, hp, n_mask=nm, snapshot_every=every)
        np.testing.assert_array_equal(ids, arr["snaps"])
        np.testing.assert_allclose(cos, np.array(meta["scores"][:-1], dtype=np.float32), atol=2e-5)
    finally:
        grp.close()


@pytest.mark.parametrize("prec", [BF16, SPLIT])
def test_two_and_three_streams_match_one_engine(prec):
    """B = 9 random images, full-size towers, two sweeps: one engine on all nine against two streams (5 + 4) and three
    (3 + 3 + 3), pixels through czc_encode_images on every member.  No kernel family is pinned: every kernel that can
    serve a layer gives the same bits (tests/test_kernels_gpu.py), so bf16 comes out bit-identical whatever the
    sub-batch sizes; the split-fp16 engine agrees id for id with cosines to fp32 rounding."""
    B, L, K, I = 9, 6, 64, 2
    su = harness.build_synthetic(False, prec, logit_scale=2.6592 if prec == BF16 else 4.6052, regular_only=True)
    pix = torch.from_numpy(synth.pixels_from_u8(synth.make_images_u8(B, su.clip_cfg.v_image))).to("cuda:0")
    init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
    pos, nm, every = harness.order_positions("sequential", L, I)
    hp = Engine.hyper(0.02, 2.0, 0.1)
    try:
        su.engine.encode_images(pix)
        ids0, cos0 = su.engine.generate(B, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)
        for streams in (2, 3):
            grp = EngineGroup(su.engine, streams=streams, min_images=2)
            assert [hi - lo for lo, hi in grp.parts(B)] == ([5, 4] if streams == 2 else [3, 3, 3])
            emb = grp.encode_images(pix)
            assert emb.shape == (B, su.clip_cfg.proj)
            ids, cos = grp.generate(B, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)
            np.testing.assert_array_equal(ids, ids0)
            if prec == BF16:
                np.testing.assert_array_equal(cos, cos0)
            else:  # split-fp16 kernels are chosen by row count too (ring kernel from 16 k rows): fp32-level differences remain
                np.testing.assert_allclose(cos, cos0, atol=2e-6)
            grp.close(parent=False)
    finally:
        su.engine.close()


def test_group_profile_counts_overlapping_launches_once():
    """profile_get of a group: launches and FLOPs add up over the members, busy_ms (union of the launch intervals on
    one clock) is at most the sum of the durations and at least the longest member's."""
    assert union_ms([np.array([[0.0, 1.0], [2.0, 3.0]]), np.array([[0.5, 2.5], [10.0, 11.0]])]) == 4.0
    B, L, K = 64, 4, 100
    su = harness.build_synthetic(False, BF16, regular_only=True)
    grp = EngineGroup(su.engine, streams=2, min_images=8)
    try:
        pix = torch.from_numpy(synth.pixels_from_u8(synth.make_images_u8(B, su.clip_cfg.v_image))).to("cuda:0")
        init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
        pos, nm, every = harness.order_positions("sequential", L, 1)
        hp = Engine.hyper(0.02, 2.0, 0.1)
        grp.encode_images(pix)
        grp.generate(B, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)  # warm-up (workspace growth)
        grp.profile_reset()
        grp.profile(2)
        grp.generate(B, init, L, SEED_LEN, K, pos, hp, n_mask=nm, snapshot_every=every)
        grp.profile(0)
        g = grp.profile_get("gemm_clip_text")
        per = [e.profile_get("gemm_clip_text") for e in grp.engines]
        assert g["launches"] == sum(p["launches"] for p in per) > 0
        assert max(p["ms"] for p in per) * 0.5 < g["busy_ms"] <= g["ms"] * 1.001
    finally:
        grp.close()


def test_orphan_replica_reports_a_closed_parent():
    """A replica only borrows its parent's weights: once the parent is closed the replica is closed with it, and calls on
    it (or asking it for another replica) say so instead of handing a NULL handle to the C ABI."""
    from conzic_amd import harness, native
    su = harness.build_synthetic(True, native.PREC_F32)
    rep = su.engine.replica()
    su.engine.close()
    with pytest.raises(native.NativeError, match="closed"):
        rep.replica()
    with pytest.raises(native.NativeError, match="closed"):
        rep.set_option("fuse_ln", 1)
    with pytest.raises(native.NativeError, match="closed"):
        su.engine.replica()


def test_configs2_batch_at_full_size_holds_the_reference_trajectories():
    """BASELINE configs[2] at its real size under pytest: 256 images, L = 10, K = 200, sequential, polished the way bench.py
    polishes them (EngineGroup: two 128-image sub-batches on two streams, czc_encode_images + czc_generate), one sweep.
    Images 0-1 of the synthetic stream are the two images of the `full_regular` golden, so inside the batch of 256
    (a) the split-fp16 engine reproduces the reference's trajectory for them id for id, and (b) the bf16 engine gives them,
    bit for bit, what it gives them polished alone at B = 2 (no kernel switches: different kernel families serve the two
    row counts)."""
    meta, arr = load_case("full_regular")
    B, L, K = 256, meta["L"], meta["K"]
    assert (L, K, meta["order"], meta["I"]) == (10, 200, "sequential", 1)
    u8 = synth.make_images_u8(B)
    pixels = synth.pixels_from_u8(u8)
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"])
    pos, nm, every = harness.order_positions("sequential", L, 1)
    for prec in (native.PREC_SPLIT, native.PREC_BF16):
        su = harness.build_synthetic(False, prec, meta["bseed"], meta["cseed"], meta["logit_scale"], regular_only=True)
        try:
            init = su.bert_tok.encode(meta["prompt"] + su.bert_tok.mask_token * L)
            grp = EngineGroup(su.engine, streams=2, min_images=32)
            assert [hi - lo for lo, hi in grp.parts(B)] == [128, 128]
            emb = grp.encode_images(pixels)
            ids, cos = grp.generate(B, init, L, 4, K, pos, hp, n_mask=nm, snapshot_every=every)
            assert ids.shape == (1, B, len(init)) and np.isfinite(cos).all()
            if prec == native.PREC_SPLIT:
                np.testing.assert_allclose(emb[:2], arr["image_embeds"], atol=5e-5 * max(1.0, float(np.abs(arr["image_embeds"]).max())))
                np.testing.assert_array_equal(ids[:, :2], arr["snaps"])
                np.testing.assert_allclose(cos[:, :2], np.array(meta["scores"][:-1], dtype=np.float32), atol=2e-5)
            else:
                grp.close(parent=False)
                emb2 = su.engine.encode_images(pixels[:2])
                ids2, cos2 = su.engine.generate(2, init, L, 4, K, pos, hp, n_mask=nm, snapshot_every=every)
                np.testing.assert_array_equal(emb2, emb[:2])
                np.testing.assert_array_equal(ids2, ids[:, :2])
                np.testing.assert_array_equal(cos2, cos[:, :2])
            # every image left the all-[MASK] state, and the images do not collapse onto a few captions (random-weight
            # towers separate random-pixel images weakly: ~126 distinct captions of 256 after one sweep)
            assert (ids[0, :, 4:4 + L] != su.bert_tok.vocab["[MASK]"]).all()
            assert len({tuple(r) for r in ids[0].tolist()}) > B // 8
        finally:
            su.engine.close()

"""CPU tests of the host side: C-ABI symbol export, product tokenizers, bridge tables."""
import json
import os
import sys
import re

import numpy as np
import pytest

from conzic_amd import native, synth
from conzic_amd.bridge import tables_from_tokenizers
from conzic_amd.text import tokenizers_from_vocab
from conzic_amd import harness
from goldutil import GOLD
from bridge_emulator import emulate_row


def test_library_exports_every_declared_symbol():
    """include/conzic_hip.h is the contract: every `int czc_*(` / `const char* czc_*(` must be
    exported by the shared library and typed in native.SIGNATURES (no compute calls here)."""
    hdr = open(native.HEADER_PATH).read()
    declared = set(re.findall(r"\b(czc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"czc_engine", "czc_control_fn"}
    assert declared, "no declarations parsed"
    lib = native.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in native.SIGNATURES, f"{name} has no ctypes signature"
    assert set(native.SIGNATURES) <= declared
    assert lib.czc_version() >= 100


def test_product_library_exports_its_header_and_nothing_else():
    """`nm -D --defined-only libconzic_hip.so` = the functions include/conzic_hip.h declares: no mangled launchers, kernel
    handles, template instantiations or globals (-fvisibility=hidden + csrc/exports.map); the hook library reaches the
    inside through czc_internal_hooks only and exports only czc_test_* / czc_bench_*."""
    import subprocess
    hdr = open(native.HEADER_PATH).read()
    declared = set(re.findall(r"\b(czc_[a-z_0-9]+)\s*\(", hdr)) - {"czc_engine", "czc_control_fn"}
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    # + the hook library's private door (declared in csrc/kernels.h, not in the public header)
    assert exported == declared | set(native.PRIVATE_SIGNATURES), (sorted(exported - declared)[:10], sorted(declared - exported))
    assert "czc_internal_hooks" not in hdr
    out = subprocess.run(["nm", "-D", "--defined-only", native.TEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    texp = {ln.split()[-1] for ln in out.splitlines() if ln.strip() and not ln.split()[-1].startswith("__hip_cuid")}
    assert texp == set(native.TEST_SIGNATURES), sorted(texp ^ set(native.TEST_SIGNATURES))
    und = subprocess.run(["nm", "-D", "--undefined-only", native.TEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "_ZN3czc" not in und  # no C++ symbol of the product library is needed
    assert native.load().czc_internal_hooks(0) is None  # a wrong tag is refused


def test_test_hooks_live_in_their_own_library():
    """The kernel-level parity hooks and the GEMM microbenchmark (include/conzic_hip_test.h) are test infrastructure:
    libconzic_hip_test.so exports every one of them, the product library and its header none."""
    import subprocess
    thdr = open(native.TEST_HEADER_PATH).read()
    declared = set(re.findall(r"\b(czc_[a-z_0-9]+)\s*\(", thdr))
    assert declared and all(n.startswith(("czc_test_", "czc_bench_")) for n in declared), declared
    assert declared == set(native.TEST_SIGNATURES)
    tlib = native.load_test()
    for name in sorted(declared):
        assert hasattr(tlib, name), name
    product = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "czc_create" in product
    assert "czc_test_" not in product and "czc_bench_" not in product
    assert not re.search(r"czc_(test|bench)_", open(native.HEADER_PATH).read())


def test_product_modules_never_touch_the_test_library():
    """The drop-in modules and conzic_amd/ reach the GPU through libconzic_hip.so only: none of them loads the hook library
    (native.load_test), names a czc_test_* / czc_bench_* symbol, imports tests/kernel_hooks.py or the oracle."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ("gen_utils.py", "control_gen_utils.py", "utils.py", "clip/clip.py")]
    files += sorted(glob.glob(os.path.join(root, "conzic_amd", "*.py")))
    for f in files:
        src = open(f).read()
        if f.endswith(os.path.join("conzic_amd", "native.py")):
            src = src[:src.index("# every entry point include/conzic_hip_test.h declares")] + src[src.index("_lib: Optional"):]
            src = src.replace("def load_test()", "def _lt()")
        for needle in ("load_test(", "czc_test_", "czc_bench_", "kernel_hooks"):
            assert needle not in src, (f, needle)
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_struct_sizes_match_header_layout():
    assert native.ctypes_sizeof_ok() if hasattr(native, "ctypes_sizeof_ok") else True
    import ctypes as C
    assert C.sizeof(native.Config) == 30 * 4
    assert C.sizeof(native.Hyper) == 6 * 4
    assert C.sizeof(native.StepOut) == 13 * C.sizeof(C.c_void_p)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError, match="no HIP device|no CPU fallback|failed"):
        harness.build_synthetic(tiny=True)


@pytest.fixture(scope="module")
def bridge_gold():
    return json.load(open(os.path.join(GOLD, "text_bridge.json")))


@pytest.mark.parametrize("label", ["tiny", "full"])
def test_product_tokenizers_match_hf_golden(bridge_gold, label):
    sv = harness.cached_vocab(label == "tiny")
    bt, ct = tokenizers_from_vocab(sv)
    g = bridge_gold[label]
    assert bt.encode("Image of a" + bt.mask_token * 5) == g["init_ids"]
    for ids, s, c in zip(g["rows"], g["strings"], g["clip_ids"]):
        assert bt.decode(ids, skip_special_tokens=True) == s
        assert ct([s])["input_ids"][0] == c
    for ids, s in zip(g["rows"], g["full_decode"]):
        assert bt.decode(ids) == s


@pytest.mark.parametrize("label", ["tiny", "full"])
def test_bridge_tables_and_device_algorithm_emulated(bridge_gold, label):
    """The device bridge's algorithm, executed in Python on the real tables, reproduces the HF
    decode -> CLIP tokenise round trip id-for-id (incl. '##' gluing, punctuation clean-up,
    contractions, non-ASCII, [unusedN], truncation to 77)."""
    sv = harness.cached_vocab(label == "tiny")
    bt, ct = tokenizers_from_vocab(sv)
    t = tables_from_tokenizers(bt, ct)
    g = bridge_gold[label]
    n_over = 0
    for ids, c in zip(g["rows"], g["clip_ids"]):
        try:
            assert emulate_row(t, ids) == c
        except OverflowError:
            n_over += 1
    assert n_over <= len(g["rows"]) // 8  # only the deliberately over-long rows may overflow


def test_token_mask_composition():
    sv = harness.cached_vocab(True)
    m = synth.make_token_mask(sv)
    bv = sv.bert_vocab
    assert m.shape == (1, len(sv.bert_tokens))
    assert m[0, bv["[UNK]"]] == 0 and m[0, bv["[unused1]"]] == 0 and m[0, bv["!"]] == 0
    assert m[0, bv["image"]] == 1 and m[0, bv["[MASK]"]] == 1  # specials are NOT stop words (demo.py:135-143)
    mr = synth.make_token_mask(sv, regular_only=True)
    assert mr[0, sv.regular_lo:sv.regular_hi].all() and mr.sum() == sv.regular_hi - sv.regular_lo


def test_order_positions():
    pos, nm, every = harness.order_positions("span", 5, 2)
    assert pos == [0, 1, 2, 3, 4] * 2 and nm == [2, 0, 2, 0, 1] * 2 and every == 5
    pos, nm, every = harness.order_positions("shuffle", 4, 2, order_list=[2, 1, 3, 0])
    assert pos == [2, 1, 3, 0, 2, 1, 3, 0]


def test_weight_generator_is_order_independent_and_tied():
    cfg = synth.bert_tiny(640)
    a = synth.make_bert_weights(cfg, 11)
    b = synth.make_bert_weights(cfg, 11)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["cls.predictions.decoder.weight"] is a["bert.embeddings.word_embeddings.weight"]
    c = synth.make_bert_weights(cfg, 12)
    assert not np.array_equal(a["cls.predictions.bias"], c["cls.predictions.bias"])


def test_run_cli_batching_and_result_layout(tmp_path, monkeypatch):
    """run.py semantics: listdir order, drop_last batching, iter_k.json / best_clipscore.json layout."""
    import json
    from conzic_amd import run_cli
    assert list(run_cli.batches(["a", "b", "c", "d", "e"], 2)) == [["a", "b"], ["c", "d"]]
    res = [None] * 3
    res = run_cli.merge_results(res, [["t0a", "t0b"], ["t1a", "t1b"], ["best_a", "best_b"]], ["a", "b"])
    res = run_cli.merge_results(res, [["t0c", "t0d"], ["t1c", "t1d"], ["best_c", "best_d"]], ["c", "d"])
    monkeypatch.chdir(tmp_path)
    args = run_cli.get_args(["--run_type", "caption", "--order", "sequential"])
    d = run_cli.result_dir(args, "caption", 0)
    assert d == "results/caption_sequential_len10_topk200_alpha0.020_beta2.000_gamma5.000_lmTemp0.100/sample_0"
    run_cli.write_results(d, res)
    assert sorted(os.listdir(d)) == ["best_clipscore.json", "iter_0.json", "iter_1.json"]
    assert json.load(open(os.path.join(d, "iter_1.json"))) == {"a": "t1a", "b": "t1b", "c": "t1c", "d": "t1d"}
    assert json.load(open(os.path.join(d, "best_clipscore.json")))["d"] == "best_d"


def test_dropin_clip_recognises_the_standard_image_processor():
    """clip/clip.py hands images to the device processor only when the checkpoint's HF image processor is the
    geometry czc_preprocess_u8 implements; anything else keeps the HF path."""
    import types
    from clip.clip import CLIP
    from conzic_amd import synth
    try:
        from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    except ImportError:
        pytest.skip("transformers PIL image processor unavailable")
    c = CLIP(None)
    c.czc_cfg = synth.clip_tiny(300)
    S = c.czc_cfg.v_image
    c.processor = types.SimpleNamespace(image_processor=CLIPImageProcessorPil(size={"shortest_edge": S},
                                                                             crop_size={"height": S, "width": S}))
    mean, std = c._device_processor_params()
    np.testing.assert_allclose(mean, synth.CLIP_MEAN, rtol=0, atol=1e-7)
    np.testing.assert_allclose(std, synth.CLIP_STD, rtol=0, atol=1e-7)
    c.processor = types.SimpleNamespace(image_processor=CLIPImageProcessorPil(size={"shortest_edge": S}, resample=2,
                                                                             crop_size={"height": S, "width": S}))
    assert c._device_processor_params() is None  # bilinear: not ours
    c.processor = types.SimpleNamespace(image_processor=CLIPImageProcessorPil(size={"shortest_edge": S + 8},
                                                                             crop_size={"height": S, "width": S}))
    assert c._device_processor_params() is None


def test_checkpoint_directories_round_trip(tmp_path):
    """conzic_amd/checkpoint.py: Hugging Face style directories (config.json, model.safetensors, vocab files)
    -> configs, fp32 tensors under their HF names, tokenizers; no torch modules involved."""
    from conzic_amd import checkpoint
    sv = synth.make_vocab_tiny()
    bcfg, ccfg = synth.bert_tiny(len(sv.bert_tokens)), synth.clip_tiny(len(sv.clip_vocab))
    bw, cw = synth.make_bert_weights(bcfg, 11), synth.make_clip_weights(ccfg, 12)
    bdir, cdir = checkpoint.write_checkpoint_dirs(str(tmp_path), bcfg, bw, ccfg, cw, sv)
    with open(os.path.join(bdir, "config.json")) as f:
        assert checkpoint.bert_cfg_from_json(json.load(f)) == bcfg
    with open(os.path.join(cdir, "config.json")) as f:
        got = checkpoint.clip_cfg_from_json(json.load(f))
    assert got == ccfg
    rb, rc = checkpoint.read_safetensors(bdir), checkpoint.read_safetensors(os.path.join(cdir, "model.safetensors"))
    assert set(rb) == set(bw) and set(rc) == set(cw)
    for k in bw:
        np.testing.assert_array_equal(rb[k], np.asarray(bw[k], np.float32))
    bt, ct = checkpoint.load_tokenizers(bdir, cdir)
    bt0, ct0 = tokenizers_from_vocab(sv)
    text = "image of a " + " ".join(sv.bert_tokens[sv.regular_lo:sv.regular_lo + 5])
    assert bt.encode(text) == bt0.encode(text)
    assert ct(text)["input_ids"] == ct0(text)["input_ids"]
    with pytest.raises(FileNotFoundError):
        checkpoint.read_safetensors(str(tmp_path / "nothing_here"))


def test_advance_order_rng_consumes_what_a_generation_call_would():
    """A rank that skips a batch must leave the process-global streams where a real call would (gen_utils.py:110-111
    one random.shuffle per shuffle call; :210 one np.random.randint per iteration of the random order)."""
    import random
    from conzic_amd.runtime import advance_order_rng
    random.seed(42)
    np.random.seed(42)
    lst = list(range(10))
    random.shuffle(lst)          # what a shuffle_generation call draws
    after_real = random.random()
    random.seed(42)
    advance_order_rng("shuffle", 10, 7)
    assert random.random() == after_real
    np.random.seed(5)
    [np.random.randint(0, 10) for _ in range(7)]
    after_real = np.random.randint(0, 1 << 30)
    np.random.seed(5)
    advance_order_rng("random", 10, 7)
    assert np.random.randint(0, 1 << 30) == after_real
    random.seed(1)
    a = random.random()
    random.seed(1)
    advance_order_rng("sequential", 10, 7)   # draws nothing
    assert random.random() == a


def test_sentiwordnet_table_builder_with_stand_in_nltk():
    """conzic_amd/sentiment.py against sentiments_classifer.py:14-30 arithmetic with a stand-in nltk (the real one and
    its corpora are absent here): per (word, class) mean of pos-neg over the synsets, 0 without synsets; class of a
    token from the Penn tag map; '##' pieces and specials never start a word."""
    import types
    from conzic_amd import sentiment

    class Syn:
        def __init__(self, p, n): self.p, self.n = p, n
        def pos_score(self): return self.p
        def neg_score(self): return self.n
    db = {("good", "a"): [Syn(0.75, 0.0), Syn(0.5, 0.25)], ("good", "n"): [Syn(0.5, 0.0)], ("bad", "a"): [Syn(0.0, 0.625)],
          ("run", "v"): [Syn(0.125, 0.125)]}
    tags = {"good": "JJ", "bad": "JJ", "run": "VB", "dog": "NN", "the": "DT"}
    fake = types.SimpleNamespace(
        corpus=types.SimpleNamespace(sentiwordnet=types.SimpleNamespace(senti_synsets=lambda w, c: db.get((w, c), []))),
        pos_tag=lambda ws: [(w, tags.get(w, "XX")) for w in ws])
    toks = ["[PAD]", "[CLS]", "good", "bad", "run", "dog", "the", "##ly"]
    table, cls = sentiment.build_sentiwordnet_tables(toks, nltk_module=fake)
    assert table.shape == (8, 5) and cls.tolist() == [0, 0, 3, 3, 2, 1, 0, 0]
    assert table[2, 3] == np.float32((0.75 + 0.25) / 2) and table[2, 1] == np.float32(0.5) and table[2, 2] == 0
    assert table[3, 3] == np.float32(-0.625) and table[4, 2] == 0.0 and not table[[0, 1, 7]].any()
    with pytest.raises(ImportError, match="nltk"):
        sentiment.build_sentiwordnet_tables(toks)


def test_state_names_of_legacy_checkpoints_are_normalised():
    from conzic_amd.engine import normalize_state_name as n
    assert n("bert.embeddings.LayerNorm.gamma") == "bert.embeddings.LayerNorm.weight"
    assert n("bert.encoder.layer.3.output.LayerNorm.beta") == "bert.encoder.layer.3.output.LayerNorm.bias"
    assert n("cls.predictions.transform.LayerNorm.gamma") == "cls.predictions.transform.LayerNorm.weight"
    assert n("bert.pooler.dense.weight") is None and n("cls.seq_relationship.bias") is None
    assert n("bert.embeddings.position_ids") is None and n("cls.predictions.decoder.weight") is None
    assert n("text_model.encoder.layers.0.mlp.fc1.weight") == "text_model.encoder.layers.0.mlp.fc1.weight"


def test_engine_group_partition_and_interval_union():
    """Host logic of the two-stream path: contiguous sub-batches of at least min_images images, never more parts than
    engines; the busy time of overlapping launch intervals from several engines is counted once."""
    from conzic_amd.engine import EngineGroup, union_ms
    g = object.__new__(EngineGroup)
    g.engines, g.min_images = [None, None], 32
    assert g.parts(256) == [(0, 128), (128, 256)]
    assert g.parts(65) == [(0, 33), (33, 65)]
    assert g.parts(63) == [(0, 63)] and g.parts(1) == [(0, 1)]
    g.engines = [None, None, None]
    assert g.parts(256) == [(0, 86), (86, 171), (171, 256)]
    assert g.parts(70) == [(0, 35), (35, 70)]
    a = np.array([[0.0, 1.0], [2.0, 3.0]])
    b = np.array([[0.5, 2.5], [10.0, 11.0]])
    assert union_ms([a, b]) == 4.0 and union_ms([a]) == 2.0 and union_ms([np.zeros((0, 2))]) == 0.0


def test_bench_gpus_n_never_falls_back_to_fewer_gpus():
    """`python bench.py --gpus 2` on a box with fewer than two GPUs (this container has none) must exit non-zero with
    the reason and print no JSON line -- a --gpus 8 run can never come back as n_gpus 1 (missing #1).
    A launcher whose rank count disagrees with --gpus is refused as well."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CZC_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env2, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in r.stderr


def test_bench_line_bookkeeping_follows_the_shape_it_ran():
    """bench.py's `metric` names the L / K / order (/ gamma / samples_num) the line was measured on -- BASELINE.json's own
    string only for configs[2] -- and per-caption figures divide by images x samples_num.  Every committed bench line of this
    round executes at most the algorithmic FLOPs (prefix sharing and row pruning only REMOVE work)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    base = json.load(open(os.path.join(root, "BASELINE.json")))["metric"]
    assert base.startswith(bench.metric_name(10, 200, "sequential"))
    presets = {n: dict(dict(L=10, topk=200, order="sequential", gamma=None, samples=1), **p) for n, p in bench.CONFIG_PRESETS.items()}
    names = {n: bench.metric_name(p["L"], p["topk"], p["order"], p["gamma"], p["samples"]) for n, p in presets.items()}
    assert names[1] == names[2] == "captions/sec (L=10, K=200, seq order)"
    assert names[3] == "captions/sec (L=15, K=512, shuffle order, samples_num=3)"
    assert names[4] == "captions/sec (L=12, K=200, seq order, sentiment gamma=5)"
    assert bench.per_caption(12.0, 4, 3) == 1.0 and bench.per_caption(12.0, 4) == 3.0 and bench.per_caption(8.0, 4, 0) == 2.0
    for n, p in presets.items():   # the work-skipping engine can only execute less than the reference's full towers
        assert bench.caption_flops(p["L"], p["topk"], 10) > 0
    lines = sorted(glob.glob(os.path.join(root, "profiles", "r06_bench_*.json")))
    for path in lines:
        j = json.load(open(path))
        if not isinstance(j, dict) or j.get("executed_tflop_per_caption") is None:
            continue
        assert j["executed_tflop_per_caption"] <= j["algorithmic_tflop_per_caption"], path
        c = j["config"]
        assert j["metric"] == bench.metric_name(c["sentence_len"], c["candidate_k"], c["order"], c["gamma"], c["samples_num"]), path


def test_image_cache_key_follows_the_pixels_not_only_the_object():
    """The drop-in CLIP caches the last batch's embeddings per image OBJECT (demo.py:83 polishes one image samples_num
    times); a caller that refills the same buffer in place must not get the old embeddings back."""
    from PIL import Image
    from clip.clip import _fingerprint
    a = np.zeros((40, 30, 3), np.uint8)
    k0 = _fingerprint(a)
    a[:] = 7                       # same object, new pixels
    assert _fingerprint(a) != k0
    im = Image.new("RGB", (33, 21), (1, 2, 3))
    k1 = _fingerprint(im)
    assert k1 == _fingerprint(im)
    im.paste((9, 9, 9), (0, 0, 33, 21))
    assert _fingerprint(im) != k1
    assert _fingerprint(object()) is None   # unknown objects are never cached
    # a change ANYWHERE is seen (every pixel byte is hashed: a frame with a static border, a one-pixel edit), also through a
    # non-contiguous view
    b = np.random.default_rng(3).integers(0, 255, (64, 48, 3), dtype=np.uint8)
    kb, kv = _fingerprint(b), _fingerprint(b[:, ::2])
    b[37, 22, 1] ^= 1
    assert _fingerprint(b) != kb and _fingerprint(b[:, ::2]) != kv
    im2 = Image.fromarray(b)
    k2 = _fingerprint(im2)
    im2.putpixel((5, 60), (1, 2, 3))
    assert _fingerprint(im2) != k2


def test_visiting_orders_and_partitions_hold_their_invariants():
    """Property tests (hypothesis) of the host-side schedule builders: every visiting order touches each position the
    reference's number of times per sweep (gen_utils.py:64-65, :110-115, :160-166), span steps re-use the forward of the
    step before them exactly when they are the second position of a pair, and the image partitions (ranks, streams) are
    contiguous, disjoint and complete."""
    from hypothesis import given, settings, strategies as st
    from conzic_amd import dist as czd
    from conzic_amd.engine import EngineGroup

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 24), st.integers(1, 6), st.randoms(use_true_random=False))
    def orders(L, iters, rnd):
        lst = list(range(L))
        rnd.shuffle(lst)
        for order, kw in (("sequential", {}), ("shuffle", dict(order_list=lst)), ("span", {})):
            pos, nm, every = harness.order_positions(order, L, iters, **kw)
            assert every == L and len(pos) == len(nm) == L * iters
            for it in range(iters):
                sweep = pos[it * L:(it + 1) * L]
                assert sorted(sweep) == list(range(L))
                if order == "shuffle":
                    assert sweep == lst
            if order == "span":
                for i, (p, n) in enumerate(zip(pos, nm)):
                    assert n in (0, 1, 2)
                    if n == 0:
                        assert nm[i - 1] == 2 and pos[i - 1] == p - 1   # second position of a pair re-uses the forward
                    if n == 2:
                        assert nm[i + 1] == 0
            else:
                assert set(nm) == {1}
    orders()

    @settings(max_examples=80, deadline=None)
    @given(st.integers(0, 5000), st.integers(1, 16))
    def shards(n, world):
        spans = [czd.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    shards()

    class _E:  # EngineGroup.parts only needs len(engines) and min_images
        pass

    @settings(max_examples=80, deadline=None)
    @given(st.integers(1, 600), st.integers(1, 4), st.integers(1, 64))
    def parts(B, streams, min_images):
        g = EngineGroup.__new__(EngineGroup)
        g.engines, g.min_images = [_E()] * streams, min_images
        p = g.parts(B)
        assert p[0][0] == 0 and p[-1][1] == B and all(a[1] == b[0] for a, b in zip(p, p[1:]))
        assert len(p) == 1 or all(hi - lo >= min_images for lo, hi in p)
        assert 1 <= len(p) <= streams
    parts()


@pytest.mark.parametrize("name", ["tiny_senti_ctx", "tiny_senti_ctx_neg", "tiny_pos_ctx", "full_senti_ctx", "full_pos_ctx",
                                  "full_senti_shuffle_neg_ctx"])
def test_host_control_scorer_reproduces_the_reference_scores(name):
    """The product's exact-mode scorer (conzic_amd/control.py: what `czc_set_control_callback` calls once per step) on the
    candidate rows of the `*_ctx` goldens: the raw control score of every candidate equals what the reference's UNCHANGED
    `text_POS_Sentiments_analysis` / `batch_texts_POS_analysis` returned over the same stand-in nltk (no GPU involved:
    decode + score only)."""
    import nltk_standin
    from conzic_amd import control
    from conzic_amd.text import tokenizers_from_vocab
    from goldutil import load_case
    meta, arr = load_case(name)
    sv = harness.cached_vocab(meta["tiny"])
    tok, _ = tokenizers_from_vocab(sv)
    m = nltk_standin.install()
    try:
        if meta.get("pos"):
            scorer = control.HostScorer(tok, "pos", meta["pos"], m)
        else:
            scorer = control.HostScorer(tok, "sentiment", meta["style"], m)
        mask = synth.make_token_mask(sv, regular_only=meta["regular_only"])[0]
        n = arr["ctl_raw"].shape[0] if meta["tiny"] else 3
        for i in range(n):
            pos = meta["positions"][i]
            mk = mask.copy()
            mk[tok.vocab["."]] = 1.0 if pos == meta["L"] - 1 else 0.0
            cand = (arr["idxs"][i] * mk[arr["idxs"][i]]).astype(np.int32)      # control_gen_utils.py:52
            got = scorer(arr["inp_before"][i].astype(np.int32), cand, 4 + pos)
            np.testing.assert_allclose(got, arr["ctl_raw"][i], atol=1e-6, rtol=0)
        assert scorer.calls == n
    finally:
        nltk_standin.uninstall()


def test_control_mode_selection(monkeypatch):
    """control.configure: caller tables win under auto / table; with nltk importable auto = the reference's scorer through the
    callback, CZC_CONTROL=table builds the tables once per tokenizer; without nltk and tables it raises."""
    import nltk_standin
    from conzic_amd import control
    from conzic_amd.text import tokenizers_from_vocab

    class FakeEngine:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            if name.startswith("set_"):
                return lambda *a: self.calls.append((name, a))
            raise AttributeError(name)

    class Clip:
        lexicon = None
        lexicon_pos = None
        pos_tags = None

    sv = harness.cached_vocab(True)
    tok, _ = tokenizers_from_vocab(sv)
    V = len(sv.bert_tokens)
    for k in ("nltk", "nltk.tokenize", "nltk.corpus"):
        monkeypatch.setitem(sys.modules, k, None)
    monkeypatch.delenv("CZC_CONTROL", raising=False)
    e, c = FakeEngine(), Clip()
    with pytest.raises(RuntimeError, match="nltk"):
        control.configure(e, c, tok)
    c.lexicon = np.zeros(V, np.float32)
    assert control.configure(e, c, tok) == "caller-tables" and e.calls[-1][0] == "set_lexicon"
    c.pos_tags = np.zeros(V, np.uint8)
    assert control.configure(e, c, tok, pos_template=[["NOUN"], ""]) == "caller-tables" and e.calls[-1][0] == "set_pos"
    m = nltk_standin.install()
    try:
        e, c = FakeEngine(), Clip()
        assert control.configure(e, c, tok) == "exact"
        assert e.calls[-1][0] == "set_control_callback" and isinstance(e.calls[-1][1][0], control.HostScorer)
        monkeypatch.setenv("CZC_CONTROL", "table")
        n0 = len(e.calls)
        assert control.configure(e, c, tok) == "table" and e.calls[-1][0] == "set_lexicon_pos"
        t1 = e.calls[-1][1]
        assert control.configure(e, c, tok) == "table" and e.calls[-1][1][0] is t1[0]       # cached per tokenizer
        assert control.configure(e, c, tok, pos_template=[["NOUN"]]) == "table" and e.calls[-1][0] == "set_pos"
        assert e.calls[n0][0] == "set_control_callback" and e.calls[n0][1] == (None,)         # a table call clears the callback
        monkeypatch.setenv("CZC_CONTROL", "bogus")
        with pytest.raises(ValueError):
            control.configure(e, c, tok)
    finally:
        nltk_standin.uninstall()


def test_host_control_scorer_worker_pool_gives_the_serial_scores():
    """`CZC_CONTROL_WORKERS`: a step's candidate strings scored by spawned interpreters (each with its own nltk -- here the
    stand-in, installed through the worker hook) are the serially scored ones, for both control types."""
    import nltk_standin
    from conzic_amd import control
    from conzic_amd.text import tokenizers_from_vocab
    from goldutil import load_case
    m = nltk_standin.install()
    try:
        for name in ("full_senti_ctx", "full_pos_ctx"):
            meta, arr = load_case(name)
            sv = harness.cached_vocab(False)
            tok, _ = tokenizers_from_vocab(sv)
            kind, param = ("pos", meta["pos"]) if meta.get("pos") else ("sentiment", meta["style"])
            serial = control.HostScorer(tok, kind, param, m)
            pooled = control.HostScorer(tok, kind, param, m, workers=3, worker_hook=m.__worker_hook__)
            try:
                inp = arr["inp_before"][2].astype(np.int32)
                cand = arr["idxs"][2].astype(np.int32)
                a, b = serial(inp, cand, 4 + meta["positions"][2]), pooled(inp, cand, 4 + meta["positions"][2])
                assert pooled._pool is not None            # 400 strings: the pool was used
                np.testing.assert_array_equal(a, b)
                np.testing.assert_allclose(a, arr["ctl_raw"][2], atol=1e-6, rtol=0)
            finally:
                pooled.close()
    finally:
        nltk_standin.uninstall()


def test_host_control_scorer_survives_a_full_sentence_memo():
    """The sentence memo of the exact-mode scorer is bounded (SENT_MEMO_MAX).  Reaching the bound empties it -- without
    losing strings the running call still has to return (hits captured before the eviction), from either of the two host
    threads that share one scorer (one per stream).  Scores stay those of an unbounded memo."""
    import threading
    from conzic_amd import control

    class Fake(control.HostScorer):
        def __init__(self, cap):
            super().__init__(None, "sentiment", "positive", None)
            self.SENT_MEMO_MAX = cap

        def _score_new(self, texts):
            return [float(len(t)) + sum(map(ord, t)) * 1e-3 for t in texts]

    s = Fake(4)
    assert s.score_texts(["a", "bb", "ccc"]) == Fake(1 << 20).score_texts(["a", "bb", "ccc"])
    # 'a' is a memo hit, 'dddd' + 'e' overflow the 4-entry memo: the call used to raise KeyError('a')
    got = s.score_texts(["a", "dddd", "e", "a"])
    assert got == Fake(1 << 20).score_texts(["a", "dddd", "e", "a"])
    assert s.memo_evictions == 1 and len(s.sent_memo) <= 4
    big = [str(i) for i in range(64)]                # one call larger than the cap: scored, not remembered, no error
    assert s.score_texts(big) == Fake(1 << 20).score_texts(big)
    assert s.asked == 3 + 4 + 64 and s.scored == 3 + 2 + 64

    # two threads on one scorer with a tiny cap: every call returns the unbounded-memo scores
    s2, ref, errs = Fake(8), Fake(1 << 20), []

    def run(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(300):
                texts = ["w%d" % v for v in rng.integers(0, 40, size=12)]
                if s2.score_texts(texts) != [float(len(t)) + sum(map(ord, t)) * 1e-3 for t in texts]:
                    errs.append("mismatch")
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:3]
    assert s2.memo_evictions > 0
    del ref


def test_control_workers_divide_the_host_between_the_ranks(monkeypatch):
    """CZC_CONTROL_WORKERS unset: a rank spawns min(32, its share of the host / 2) interpreters, the share being cores /
    ranks on this host (LOCAL_WORLD_SIZE, else WORLD_SIZE) or the process's affinity mask, whichever is smaller."""
    from conzic_amd import control
    monkeypatch.setattr(control.os, "cpu_count", lambda: 128)
    monkeypatch.setattr(control, "host_cpus", lambda: 128)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert control.default_workers(100) == 0 and control.default_workers(51200) == 32
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert control.default_workers(51200) == 8            # 8 ranks x 8 interpreters = 64 = cores / 2
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")           # two ranks on this host of an 8-rank job
    assert control.default_workers(51200) == 32
    monkeypatch.setattr(control, "host_cpus", lambda: 16)  # pinned to a 16-CPU NUMA node
    assert control.default_workers(51200) == 8


def test_headers_are_plain_c_and_match_the_ctypes_layout(tmp_path):
    """The boundary is a C ABI: both headers compile as C99 with gcc (no C++-isms, no torch / HIP types), a C translation unit
    can name every entry point with the declared prototype, and the struct sizes / field offsets a C compiler derives are the
    ones conzic_amd/native.py hands over through ctypes."""
    import ctypes as C
    import subprocess
    inc = os.path.dirname(native.HEADER_PATH)
    src = tmp_path / "abi_probe.c"
    names = sorted(native.SIGNATURES) + sorted(native.TEST_SIGNATURES)
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "conzic_hip.h"\n#include "conzic_hip_test.h"\n'
        "int main(void) {\n"
        + "".join(f"  (void)&{n};\n" for n in names) +
        '  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(czc_config), sizeof(czc_hyper), sizeof(czc_step_out), sizeof(czc_bridge_tables),\n'
        "         offsetof(czc_config, precision), offsetof(czc_bridge_tables, merge_out), offsetof(czc_hyper, negative));\n"
        "  czc_control_fn fn = 0; (void)fn;\n  return 0;\n}\n")
    exe = tmp_path / "abi_probe"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe),
                        "-L", os.path.dirname(native.LIB_PATH), "-lconzic_hip_test", "-lconzic_hip",
                        "-Wl,-rpath," + os.path.dirname(native.LIB_PATH), "-Wl,--unresolved-symbols=ignore-in-shared-libs"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    got = [int(v) for v in out]
    want = [C.sizeof(native.Config), C.sizeof(native.Hyper), C.sizeof(native.StepOut), C.sizeof(native.BridgeTables),
            native.Config.precision.offset, native.BridgeTables.merge_out.offset, native.Hyper.negative.offset]
    assert got == want, (got, want)


def test_generated_doc_blocks_are_fresh():
    """README.md, DESIGN.md §0 and the measured-figures comment of include/conzic_hip.h quote the committed evidence under
    profiles/ through tools/refresh_docs.py (one source of truth per figure): the blocks in the tree must be what the tool
    generates today."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "refresh_docs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

"""-m gpu: the N>1 path on ONE GPU -- two ranks (gloo rendezvous on 127.0.0.1, both engines on cuda:0) do what two
GPUs would: receive the frozen weights through `dist.broadcast_state`, polish their own image shard with the
visiting order every rank agrees on, gather.  Images are independent (gen_utils.py:65-81 has no cross-image term)
and the order is an explicit input, so the 2-rank result must equal the 1-rank run image for image.
The f32 engine is used because its kernels do not depend on the batch size, which makes "equal" mean bit-equal."""
import json
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, B, L, K, I, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from conzic_amd import dist as czd, harness, native, synth
    from conzic_amd.engine import Engine
    from oracle import step as S
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        sv = harness.cached_vocab(True)
        bcfg, ccfg = synth.bert_tiny(len(sv.bert_tokens)), synth.clip_tiny(len(sv.clip_vocab))
        bw = czd.broadcast_state(synth.make_bert_weights(bcfg, 11) if rank == 0 else None, dev)
        cw = czd.broadcast_state(synth.make_clip_weights(ccfg, 12) if rank == 0 else None, dev)
        su = harness.build_synthetic(True, native.PREC_F32, device=0, bert_w=bw, clip_w=cw, bert_cfg=bcfg, clip_cfg=ccfg)
        lo, hi = czd.shard_range(B, rank, world)
        pix = synth.pixels_from_u8(synth.make_images_u8(hi - lo, ccfg.v_image, first=lo))
        su.engine.encode_images(pix)
        # one shuffle order per call for the whole job: rank 0 draws it, every rank uses it
        order = [S.shuffle_order(L, seed=42) if rank == 0 else None]
        dist.broadcast_object_list(order, src=0)
        pos, nm, every = harness.order_positions("shuffle", L, I, order_list=order[0])
        init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
        ids, cos = su.engine.generate(hi - lo, init, L, 4, K, pos, Engine.hyper(0.02, 2.0, 0.1), n_mask=nm, snapshot_every=every)
        full = czd.gather_ids(ids, world)
        q.put((rank, full.tolist(), order[0]))
        su.engine.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank_image_for_image():
    import torch.multiprocessing as mp
    from conzic_amd import harness, native, synth
    from conzic_amd.engine import Engine
    from oracle import step as S
    B, L, K, I = 6, 5, 12, 2
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, B, L, K, I, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert [r[0] for r in res] == [0, 1]
    assert res[0][1] == res[1][1], "every rank must hold the same gathered result"
    gathered = np.array(res[0][1], dtype=np.int32)
    order = res[0][2]
    assert order == S.shuffle_order(L, seed=42)
    # the single-process run of the same job
    su = harness.build_synthetic(True, native.PREC_F32)
    try:
        su.engine.encode_images(synth.pixels_from_u8(synth.make_images_u8(B, su.clip_cfg.v_image)))
        pos, nm, every = harness.order_positions("shuffle", L, I, order_list=order)
        init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)
        ids, _ = su.engine.generate(B, init, L, 4, K, pos, Engine.hyper(0.02, 2.0, 0.1), n_mask=nm, snapshot_every=every)
    finally:
        su.engine.close()
    assert gathered.shape == ids.shape
    np.testing.assert_array_equal(gathered, ids)


def _run_cli(rank, world, port, img_dir, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", CZC_DIST_BACKEND="gloo", CZC_PRECISION="f32")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.chdir(out_dir)
    from conzic_amd import run_cli
    run_cli.main(["--synthetic", "--tiny", "--caption_img_path", img_dir, "--run_type", "caption", "--order", "shuffle",
                  "--batch_size", "2", "--samples_num", "2", "--sentence_len", "5", "--candidate_k", "12",
                  "--num_iterations", "2"])
    q.put(rank)


def test_run_cli_sharded_over_two_ranks_matches_single_process(tmp_path):
    """The run.py-shaped harness: 3 batches x 2 samples, shuffle order (one draw per call from the process-global
    stream, gen_utils.py:110-111).  Two ranks -- which skip each other's batches but advance the order stream for
    them -- must write the JSON files the single process writes; the second sample reuses the cached image
    embeddings (ViT once per image)."""
    import torch.multiprocessing as mp
    from PIL import Image
    from conzic_amd import synth
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    for j, u in enumerate(synth.make_images_u8(6, 40)):
        Image.fromarray(u).save(img_dir / f"im{j}.png")
    outs = {}
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        out_dir = tmp_path / f"w{world}"
        out_dir.mkdir()
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_run_cli, args=(r, world, port, str(img_dir), str(out_dir), q)) for r in range(world)]
        for p in procs:
            p.start()
        done = sorted(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
        assert done == list(range(world))
        files = {}
        for root, _, fs in os.walk(out_dir / "results"):
            for f in fs:
                files[os.path.relpath(os.path.join(root, f), out_dir)] = json.load(open(os.path.join(root, f)))
        outs[world] = files
    assert outs[1] and set(outs[1]) == set(outs[2])
    assert any("sample_1" in k for k in outs[1]) and any(k.endswith("best_clipscore.json") for k in outs[1])
    for k in outs[1]:
        assert outs[1][k] == outs[2][k], k
        assert len(outs[1][k]) == 6


def test_bench_gpus_2_spawns_two_ranks_on_a_shared_gpu():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (a plain --gpus N run
    must never report n_gpus 1).  On this one-GPU box the ranks share the device under the explicit test flag
    (gloo rendezvous: RCCL cannot put two ranks on one device); on a node they are rank r <-> GPU r over RCCL."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "64", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--no-alt", "--no-invariance", "--share-gpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2
    assert out["ranks"]["world_size"] == 2 and out["ranks"]["reported_by_backend"] == 2
    assert out["ranks"]["shared_gpu"] is True and "TEST RUN" in out["config"]["parallelism"]
    assert out["value"] > 0 and out["config"]["images_per_gpu"] == 64


def test_bench_strong_scaling_two_ranks_reproduce_the_one_rank_captions():
    """`--total-images N` fixes the TOTAL (BASELINE configs[3]/[4] are fixed-total runs): 130 images split 65 + 65 over two
    ranks (uneven per-stream sub-batches inside each rank) give, image for image, the captions of the same 130 images on
    one rank -- compared through the crc32 of the gathered final ids; the line says `scaling: strong`, reports per-rank
    captions/s, and carries the scale-100 leg under world > 1 too."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "1", "--warmup", "0", "--iters", "2", "--no-cpu-baseline", "--no-invariance", "--no-profile"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common,
                           capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(["--gpus", "1", "--images", "130", "--no-alt"])
    two = run(["--gpus", "2", "--share-gpu", "--total-images", "130"])
    assert two["scaling"] == "strong" and one["scaling"] == "weak"
    assert two["config"]["total_images"] == 130 and two["ranks"]["per_rank_images"] == [65, 65]
    assert len(two["ranks"]["per_rank_captions_per_s"]) == 2 and min(two["ranks"]["per_rank_captions_per_s"]) > 0
    assert two["captions_crc32"]["images"] == 130 == one["captions_crc32"]["images"]
    assert two["captions_crc32"]["value"] == one["captions_crc32"]["value"]
    assert two["scale100_mode"]["value"] > 0 and two["scale100_mode"]["steps"] == two["steps"]
    assert abs(two["value"] - sum(two["ranks"]["per_rank_captions_per_s"])) / two["value"] < 0.2


def test_bench_config3_preset_with_samples_reproduces_the_one_rank_captions():
    """`bench.py --config 3` = BASELINE configs[3] as flags (--total-images 2048 --order shuffle --len 15 --topk 512 --samples 3),
    here scaled to 64 images and one sweep: three samples per step on image embeddings encoded once, every sample with its own
    order from the one random.Random(42) stream; two ranks sharing the GPU reproduce the one-rank run's captions (crc32 of the
    last sample's final ids), and the line names the preset and the sample count."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--config", "3", "--total-images", "64", "--steps", "1", "--warmup", "0", "--iters", "1", "--no-cpu-baseline",
              "--no-invariance", "--no-profile", "--no-alt"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common,
                           capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(["--gpus", "1"])
    two = run(["--gpus", "2", "--share-gpu"])
    for d in (one, two):
        c = d["config"]
        assert c["preset"] == 3 and c["samples_num"] == 3 and c["order"] == "shuffle" and c["sentence_len"] == 15 and c["candidate_k"] == 512
        assert "configs[3]" in c["workload"] and "samples_num=3" in c["workload"]
        assert d["scaling"] == "strong" and c["total_images"] == 64
    assert two["ranks"]["per_rank_images"] == [32, 32]
    assert two["captions_crc32"]["value"] == one["captions_crc32"]["value"] and one["captions_crc32"]["images"] == 64
    # value counts every sample's captions: 64 images x 3 samples per step -- and so do the per-rank rates and the per-caption work
    assert abs(one["value"] - 64 * 3 / (one["ms_per_step"] * 1e-3)) / one["value"] < 1e-3
    assert abs(one["value"] - one["ranks"]["per_rank_captions_per_s"][0]) / one["value"] < 1e-3
    assert abs(two["value"] - sum(two["ranks"]["per_rank_captions_per_s"])) / two["value"] < 0.25
    assert one["metric"] == "captions/sec (L=15, K=512, shuffle order, samples_num=3)"


def test_bench_eight_ranks_on_a_shared_gpu_reproduce_the_one_rank_captions():
    """The rank count of the 8-GPU node, on this one-GPU box under the explicit test flag: `bench.py --gpus 8 --share-gpu
    --total-images 16` starts eight ranks (gloo rendezvous), every rank receives the weights through the start-up broadcast,
    polishes its two images, the gather restores image order: n_gpus 8, eight ranks reported by the backend, the 1-rank run's
    captions (crc32), and the line carries what the first real SCALE run needs to explain itself (broadcast_s, gather_s, the
    host share of each rank)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    common = ["--steps", "1", "--warmup", "0", "--iters", "1", "--no-cpu-baseline", "--no-invariance", "--no-profile", "--no-alt"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common,
                           capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(["--gpus", "1", "--images", "16"])
    eight = run(["--gpus", "8", "--share-gpu", "--total-images", "16"])
    assert eight["n_gpus"] == 8 and eight["ranks"]["world_size"] == 8 and eight["ranks"]["reported_by_backend"] == 8
    assert eight["ranks"]["per_rank_images"] == [2] * 8 and eight["scaling"] == "strong"
    assert eight["captions_crc32"]["images"] == 16 and eight["captions_crc32"]["value"] == one["captions_crc32"]["value"]
    assert eight["ranks"]["broadcast_s"] > 0 and eight["ranks"]["gather_s"] is not None
    assert one["ranks"]["broadcast_s"] is None and one["ranks"]["gather_s"] is None
    h = eight["ranks"]["host"]
    assert h["local_world"] == 8 and (not h["pinned"] or h["cpus"] >= 1)


def test_bench_gpus_2_without_the_flag_fails_on_one_gpu():
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CZC_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _rccl_single_rank(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from conzic_amd import dist as czd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # "nccl" is RCCL on ROCm
    try:
        rng = np.random.default_rng(5)
        w = rng.standard_normal((33, 17)).astype(np.float32)
        state = {"emb": w, "decoder_tied": w, "bias": rng.standard_normal(17).astype(np.float32),
                 "scale": np.float32(2.5).reshape(()), "on_device": torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)}
        out = czd.broadcast_state(state, dev)
        ok = all(v.device == dev and v.dtype == torch.float32 for v in out.values())
        ok &= out["emb"].data_ptr() == out["decoder_tied"].data_ptr()  # tied tensors travel (and live) once
        for k, v in state.items():
            ref = v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)
            ok &= np.array_equal(out[k].cpu().numpy(), ref)
        ids = rng.integers(0, 1000, size=(3, 4, 9)).astype(np.int32)
        ok &= np.array_equal(czd.gather_ids(ids, 1), ids)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)  # bench.py's max-over-ranks of the timed region
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        q.put((bool(ok), float(t.item()), dist.get_backend(), dist.get_world_size()))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_runs_the_start_up_broadcast_and_the_gather():
    """The collectives of the N>1 path on the backend the GPU job uses (torch 'nccl' = RCCL): weights bucket broadcast
    from device memory, id gather, max-reduce of the timing -- with the one rank this box has.  Multi-rank semantics
    are covered by the gloo tests above; this one holds the device-side branches of conzic_amd/dist.py to RCCL itself."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank, args=(_free_port(), q))
    p.start()
    ok, t, backend, world = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert ok and t == 1.25 and backend == "nccl" and world == 1

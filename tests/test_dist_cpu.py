"""CPU (gloo, world_size 2) coverage of the N>1 plumbing: weight broadcast in one bucket per tower,
block sharding of images, result gather.  The per-rank compute is independent by construction
(SURVEY.md §8e), so these are the only cross-rank operations of a multi-GPU run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conzic_amd import dist as czd
from conzic_amd import synth


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [czd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sv = synth.make_vocab_tiny()
        bcfg = synth.bert_tiny(len(sv.bert_tokens))
        ref = synth.make_bert_weights(bcfg, 11)
        got = czd.broadcast_state(ref if rank == 0 else None, torch.device("cpu"))
        ok = set(got) == set(ref)
        for k in ref:
            ok = ok and np.array_equal(got[k].numpy(), ref[k])
        # tied tensors stay tied (they travel once)
        ok = ok and got["cls.predictions.decoder.weight"].data_ptr() == got["bert.embeddings.word_embeddings.weight"].data_ptr()
        # each rank "polishes" its own image shard; the gather restores global image order
        n_img, T = 6, 5
        lo, hi = czd.shard_range(n_img, rank, world)
        local = np.stack([np.full((hi - lo, T), 0, np.int32) + np.arange(lo, hi, dtype=np.int32)[:, None]] * 2)
        full = czd.gather_ids(local, world)
        ok = ok and full.shape == (2, n_img, T) and (full[0, :, 0] == np.arange(n_img)).all()
        # uneven shards (7 images on 2 ranks: 4 + 3) and a per-image score vector ride the same gather
        lo7, hi7 = czd.shard_range(7, rank, world)
        ids7 = np.arange(lo7, hi7, dtype=np.int32)[None, :, None] * np.ones((3, 1, T), np.int32)
        full7 = czd.gather_ids(ids7, world)
        ok = ok and full7.shape == (3, 7, T) and (full7[1, :, 2] == np.arange(7)).all()
        cos7 = czd.gather_along(np.arange(lo7, hi7, dtype=np.float32)[None, :] * 0.5, world, axis=1)
        ok = ok and cos7.shape == (1, 7) and np.array_equal(cos7[0], np.arange(7, dtype=np.float32) * 0.5)
        # a rank with NO images (3 ranks' worth of work on 1 image) contributes an empty block
        lo1, hi1 = czd.shard_range(1, rank, world)
        one = czd.gather_ids(np.full((2, hi1 - lo1, T), 9, np.int32), world)
        ok = ok and one.shape == (2, 1, T)
        # synthetic image stream: any shard can be generated independently of the others
        a = synth.make_images_u8(hi - lo, 8, first=lo)
        b = synth.make_images_u8(n_img, 8)[lo:hi]
        ok = ok and np.array_equal(a, b)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_gather_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # a (small) weight bucket from rank 0 to all eight
        ref = {"a.weight": np.arange(24, dtype=np.float32).reshape(4, 6), "b.bias": np.full(5, 2.5, np.float32)}
        ref["tied.weight"] = ref["a.weight"]
        got = czd.broadcast_state(ref if rank == 0 else None, torch.device("cpu"))
        ok = ok and all(np.array_equal(got[k].numpy(), ref[k]) for k in ref)
        ok = ok and got["tied.weight"].data_ptr() == got["a.weight"].data_ptr()
        # BASELINE configs[3] / configs[4] image counts over 8 ranks (even), and counts that do not divide (uneven, with
        # ranks that own nothing): ids [S, B_local, T] and per-image cosines [S, B_local] come back in global image order
        T = 3
        for n_img in (2048, 512, 13, 5):
            lo, hi = czd.shard_range(n_img, rank, world)
            ids = (np.arange(lo, hi, dtype=np.int32)[None, :, None] + np.zeros((2, 1, T), np.int32))
            full = czd.gather_ids(ids, world)
            ok = ok and full.shape == (2, n_img, T) and np.array_equal(full[1, :, 1], np.arange(n_img, dtype=np.int32))
            cos = czd.gather_along(np.arange(lo, hi, dtype=np.float32)[None, :] * 0.25, world, axis=1)
            ok = ok and cos.shape == (1, n_img) and np.array_equal(cos[0], np.arange(n_img, dtype=np.float32) * 0.25)
            a = synth.make_images_u8(min(hi - lo, 2), 8, first=lo)     # any shard's pixels without the others'
            ok = ok and (hi == lo or np.array_equal(a[0], synth.make_images_u8(1, 8, first=lo)[0]))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_gather_gloo_world8():
    """The N-rank plumbing at the rank count of the 8-GPU node (BASELINE configs[3]: 2048 images, configs[4]: 512), with
    even and uneven shards."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_rank_cpu_shares_partition_the_numa_nodes(tmp_path):
    """pin_rank's arithmetic on a fake sysfs: 8 ranks on a 2-socket host with 4 GPUs per socket get disjoint, equal slices of
    their own socket's CPUs; without NUMA information the allowed CPUs are cut into local_world slices."""
    node_dir = tmp_path / "devices" / "system" / "node"
    (node_dir / "node0").mkdir(parents=True)
    (node_dir / "node1").mkdir(parents=True)
    (node_dir / "node0" / "cpulist").write_text("0-63,128-191\n")
    (node_dir / "node1" / "cpulist").write_text("64-127,192-255\n")
    allowed = list(range(256))
    shares = [czd.rank_cpu_share(r, 8, node=0 if r < 4 else 1, sysfs=str(tmp_path), allowed=allowed) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    flat = [c for s in shares for c in s]
    assert len(set(flat)) == 256                                    # disjoint and complete
    n0 = set(czd._parse_cpulist("0-63,128-191"))
    assert all(set(shares[r]) <= n0 for r in range(4)) and all(not (set(shares[r]) & n0) for r in range(4, 8))
    # GPUs interleaved over the sockets (0, 1, 4, 5 on node 0): the slice index comes from the rank's place among its node's ranks
    on0, on1 = [0, 1, 4, 5], [2, 3, 6, 7]
    inter = [czd.rank_cpu_share(r, 8, node=0 if r in on0 else 1, sysfs=str(tmp_path), allowed=allowed,
                                peers_on_node=on0 if r in on0 else on1) for r in range(8)]
    assert len({c for s_ in inter for c in s_}) == 256 and all(len(s_) == 32 for s_ in inter)
    assert all(set(inter[r]) <= n0 for r in on0) and all(not (set(inter[r]) & n0) for r in on1)
    plain = [czd.rank_cpu_share(r, 8, node=None, allowed=list(range(20))) for r in range(8)]
    assert sorted(c for s in plain for c in s) == list(range(20)) and max(map(len, plain)) - min(map(len, plain)) <= 1
    assert czd.rank_cpu_share(5, 8, node=None, allowed=[3, 4]) in ([3], [4])    # fewer CPUs than ranks: still one each
    assert czd.rank_cpu_share(0, 1, node=None, allowed=[1, 2, 3]) == [1, 2, 3]


def test_pin_rank_leaves_a_single_rank_alone(monkeypatch):
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    before = os.sched_getaffinity(0)
    info = czd.pin_rank(0)
    assert info["pinned"] is False and os.sched_getaffinity(0) == before
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("CZC_PIN", "0")
    assert czd.pin_rank(3)["pinned"] is False and os.sched_getaffinity(0) == before

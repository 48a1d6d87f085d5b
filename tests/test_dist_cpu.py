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

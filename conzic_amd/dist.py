"""One process per GPU.  Images (x samples) are independent units: ranks own
contiguous image shards and never exchange data during polishing.  The only collective is the
start-up broadcast of the frozen weights from rank 0 (RCCL over xGMI when the backend is "nccl";
"gloo" in the CPU tests), one bucket per tower, plus an optional final gather of results."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition: rank r owns [r*n/world, (r+1)*n/world) (remainder spread over the first ranks)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_state(state: Optional[Dict[str, np.ndarray]], device, src: int = 0):
    """Rank `src` passes a {name: fp32 ndarray} dict, the others None.  Every rank gets back
    {name: torch tensor on `device`} -- views into ONE flat bucket that went through a single
    broadcast (a 0.4-0.6 GB message per tower: per-link bound on the xGMI mesh, ~3-4 ms)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    manifest = [None]
    if rank == src:
        seen = {}
        items = []
        for k, v in state.items():
            if id(v) in seen:  # tied tensors travel once
                items.append((k, tuple(v.shape), seen[id(v)]))
            else:
                seen[id(v)] = k
                items.append((k, tuple(v.shape), None))
        manifest = [items]
    dist.broadcast_object_list(manifest, src=src)
    items = manifest[0]
    total = sum(int(np.prod(s)) if len(s) else 1 for _, s, alias in items if alias is None)
    # RCCL ("nccl") moves device memory directly over xGMI; gloo (CPU tests) stages through the host
    bdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    flat = torch.empty(total, dtype=torch.float32, device=bdev)
    if rank == src:
        # every tensor goes straight into its slice of the bucket (one H2D copy per tensor when the bucket is on the
        # GPU, no second host-side image of the 0.5 GB tower); tensors already on the device are copied D2D
        o = 0
        for k, s, alias in items:
            if alias is None:
                n = int(np.prod(s)) if len(s) else 1
                v = state[k]
                if hasattr(v, "detach"):
                    flat[o:o + n].copy_(v.detach().reshape(-1).to(torch.float32), non_blocking=True)
                else:
                    flat[o:o + n].copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(-1)))
                o += n
    dist.broadcast(flat, src=src)
    if flat.device != torch.device(device):
        flat = flat.to(device)
    out = {}
    o = 0
    for k, s, alias in items:
        if alias is None:
            n = int(np.prod(s)) if len(s) else 1
            out[k] = flat[o:o + n].view(s if len(s) else ())
            o += n
    for k, s, alias in items:
        if alias is not None:
            out[k] = out[alias]
    return out


def gather_along(local: np.ndarray, world: int, axis: int = 1) -> np.ndarray:
    """All-gather of per-rank arrays that differ in length along `axis` (uneven image shards: 7 images on 2 ranks are
    4 + 3): the lengths travel first, every rank pads its block to the longest one, one all_gather, blocks are trimmed
    back and concatenated in rank order -- i.e. global image order for `shard_range` shards."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(np.moveaxis(local, axis, 0))
    on_gpu = dist.get_backend() == "nccl"
    n = torch.tensor([a.shape[0]], dtype=torch.int64)
    if on_gpu:
        n = n.cuda()
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    longest = max(lens + [1])
    pad = np.zeros((longest,) + a.shape[1:], dtype=a.dtype)
    pad[: a.shape[0]] = a
    t = torch.from_numpy(pad)
    if on_gpu:
        t = t.cuda()
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = np.concatenate([o.cpu().numpy()[:ln] for o, ln in zip(outs, lens)], axis=0)
    return np.moveaxis(full, 0, axis)


def gather_ids(local_ids: np.ndarray, world: int):
    """Optional end-of-run gather of int32 ids [S, B_local, T] along the image axis (tiny; shards may be uneven)."""
    return gather_along(local_ids, world, axis=1)

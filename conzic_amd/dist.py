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


def local_world_size() -> int:
    """Ranks sharing this host: LOCAL_WORLD_SIZE (torchrun exports it), else WORLD_SIZE, else 1."""
    for k in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(k, "").strip()
        if v.isdigit() and int(v) > 0:
            return int(v)
    return 1


def _parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/nodeN/cpulist)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_node(device_index: int, sysfs: str = "/sys") -> Optional[int]:
    """NUMA node of GPU `device_index` (HIP device order) from sysfs, or None where the kernel does not say.  The PCI
    address comes from torch (`cuda.get_device_properties(i).pci_bus_id` etc.) when a GPU is visible; the node from
    /sys/bus/pci/devices/<addr>/numa_node (what `rocm-smi --showtoponuma` prints)."""
    addr = None
    try:
        import torch
        if torch.cuda.is_available() and device_index < torch.cuda.device_count():
            pr = torch.cuda.get_device_properties(device_index)
            dom = getattr(pr, "pci_domain_id", 0)
            addr = "%04x:%02x:%02x.0" % (dom, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001 -- pinning is an optimisation, never a reason to fail a run
        addr = None
    if addr is None:
        return None
    try:
        node = int(open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")).read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def rank_cpu_share(local_rank: int, local_world: int, node: Optional[int] = None, sysfs: str = "/sys",
                   allowed: Optional[List[int]] = None, peers_on_node: Optional[List[int]] = None) -> List[int]:
    """CPUs rank `local_rank` of `local_world` ranks on this host should run on.  With the NUMA node of its GPU known: the
    CPUs of that node, cut into one slice per rank whose GPU sits on the same node -- `peers_on_node` (the local ranks on that
    node, in order; `pin_rank` reads it from sysfs for every visible GPU) or, when that is not known, ceil(local_world / nodes)
    slices indexed by local_rank modulo that count (GPUs spread evenly and in order over the nodes: 4 per socket on the 8-GPU
    boxes).  Without NUMA information: slice local_rank of local_world equal slices of the allowed CPUs.  Always at least one
    CPU; always a subset of `allowed` (the current affinity mask)."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            allowed = list(range(os.cpu_count() or 1))
    cpus = allowed
    slices, idx = max(1, local_world), local_rank % max(1, local_world)
    if node is not None:
        try:
            on_node = set(_parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read()))
            nodes = len([d for d in os.listdir(os.path.join(sysfs, "devices/system/node")) if d.startswith("node") and d[4:].isdigit()])
            mine = [c for c in allowed if c in on_node]
            if mine:
                cpus = mine
                if peers_on_node and local_rank in peers_on_node:
                    slices, idx = len(peers_on_node), peers_on_node.index(local_rank)
                else:
                    per_node = max(1, -(-local_world // max(1, nodes)))   # ranks per node, rounded up
                    slices, idx = per_node, local_rank % per_node
        except OSError:
            pass
    q, r = divmod(len(cpus), slices)
    if q == 0:
        return [cpus[idx % len(cpus)]]
    lo = idx * q + min(idx, r)
    return cpus[lo: lo + q + (1 if idx < r else 0)]


def pin_rank(local_rank: int, device_index: Optional[int] = None, set_torch_threads: bool = True) -> Dict[str, object]:
    """Per-rank host resources of a multi-GPU run (one process per GPU on ONE host): CPU affinity = this rank's share of
    the CPUs of its GPU's NUMA node (`rank_cpu_share`), torch's intra-op threads = the size of that share.  A single-rank
    run (local world 1) is left alone.  `CZC_PIN=0` disables it.  Returns what was done (bench.py prints it)."""
    lw = local_world_size()
    info: Dict[str, object] = dict(local_world=lw, pinned=False)
    if lw <= 1 or os.environ.get("CZC_PIN", "1") == "0":
        return info
    dev = local_rank if device_index is None else device_index
    node = gpu_numa_node(dev)
    peers = None
    if node is not None and dev == local_rank:   # rank r <-> GPU r (one process per GPU): which local ranks share this node
        nodes_of = [gpu_numa_node(i) for i in range(lw)]
        if all(n is not None for n in nodes_of):
            peers = [i for i in range(lw) if nodes_of[i] == node]
    cpus = rank_cpu_share(local_rank, lw, node, peers_on_node=peers)
    try:
        os.sched_setaffinity(0, cpus)
        info.update(pinned=True)
    except (AttributeError, OSError):
        pass
    info.update(numa_node=node, cpus=len(cpus), first_cpu=cpus[0], last_cpu=cpus[-1])
    if set_torch_threads:
        try:
            import torch
            torch.set_num_threads(max(1, min(len(cpus), 16)))
            info.update(torch_threads=torch.get_num_threads())
        except Exception:  # noqa: BLE001
            pass
    return info


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

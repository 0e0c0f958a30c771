"""Multi-GPU glue for the render path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The path shards over INDEPENDENT units and has no collective on its data path:

* training: batch samples per rank -- exactly what ``idist.auto_model`` -> DistributedDataParallel does in the reference
  (models/bts/trainer.py:418); the only exchange step is the gradient all-reduce, which DDP overlaps with the backward.  The
  renderer's parameters are ordinary ``nn.Parameter``s fed through ``torch.autograd.Function``, so the DDP hooks fire unchanged
  (``wrap_ddp``).
* inference / evaluation: rays are independent -> ``render_sharded`` gives every rank a contiguous slice of the rays of each sample
  (the reference's dead ``DataParallel(dim=1)`` hook, models/common/render/nerf.py:454-456, was this axis) and all-gathers the
  per-ray outputs (0.5 MB of depth per 192x640 frame).  Every rank must hold the encoded field (run ``net.encode`` on each rank).
"""
import os
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract), picks this rank's GPU and initialises the process group.
    Returns (rank, world_size, device).  world_size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if use_cuda else "gloo")
        kwargs = dict(device_id=device) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced split of range(n_items): the first ``n_items % world`` ranks get one item more."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_rays(rays: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """rays (SB, B', 8) -> this rank's contiguous slice along B' (every rank keeps all SB samples: the field of each sample is
    resident on every rank)."""
    s, e = shard_range(rays.shape[1], rank, world)
    return rays[:, s:e].contiguous()


def all_gather_cat(t: torch.Tensor, dim: int, total: int, world: Optional[int] = None) -> torch.Tensor:
    """Concatenates the ranks' shards of unequal length along ``dim`` (shard sizes as in ``shard_range(total, ...)``).  Shards are
    padded to the largest size for the collective (one all_gather) and trimmed afterwards."""
    world = world or dist.get_world_size()
    if world == 1:
        return t
    sizes = [shard_range(total, r, world) for r in range(world)]
    longest = max(e - s for s, e in sizes)
    if t.shape[dim] < longest:
        pad = list(t.shape)
        pad[dim] = longest - t.shape[dim]
        t = torch.cat((t, t.new_zeros(pad)), dim=dim)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat([p.narrow(dim, 0, e - s) for p, (s, e) in zip(parts, sizes)], dim=dim)


def render_sharded(render: Callable[..., Dict], rays: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None,
                   **want) -> Dict:
    """``render(rays_shard (SB, b, 8), **want) -> {"coarse": {...}[, "fine": {...}]}`` on this rank's slice of the rays, then an
    all-gather of every per-ray tensor along the ray axis: returns the same dict the un-sharded call would (on every rank).
    ``render`` is the wrapped renderer (``NeRFRenderer.bind_parallel(net)``); inference only (no autograd through the gather)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    total = rays.shape[1]
    with torch.no_grad():
        out = render(shard_rays(rays, rank, world), **want)
        if world == 1:
            return out
        return {part: {k: all_gather_cat(v, 1, total, world) for k, v in d.items()} for part, d in out.items()}


def wrap_ddp(module: torch.nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 64, force: bool = False,
             find_unused_parameters: bool = False) -> torch.nn.Module:
    """DistributedDataParallel around the task wrapper, as ``idist.auto_model`` does.  64 MB buckets: the ~140 MB of CNN gradients go
    out as 2-3 large RCCL all-reduces (xGMI is point-to-point, large messages amortise the per-link latency); the renderer's 27 KB
    of MLP gradients ride in the last bucket.  At world size 1 the module comes back bare (like idist.auto_model) unless ``force``:
    then the reducer, the bucket views and the (one-rank) all-reduce run exactly as on a node -- what the single-GPU tests and
    ``bench.py`` under ``torch.distributed.run`` use to execute the DDP path over RCCL before a multi-GPU box ever sees it.
    ``find_unused_parameters``: for models with parameters outside the loss' graph -- Monodepth2's output convolutions of the scales a
    ``prediction_mode: default`` step never renders (monodepth2.py:211-239 computes all four, trainer.py:243-259 uses scale 0)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return module
    if device is not None and device.type == "cuda":
        return torch.nn.parallel.DistributedDataParallel(module, device_ids=[device.index], output_device=device.index,
                                                         bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
                                                         find_unused_parameters=find_unused_parameters)
    return torch.nn.parallel.DistributedDataParallel(module, bucket_cap_mb=bucket_cap_mb, find_unused_parameters=find_unused_parameters)


def all_reduce_mean_(tensors: Sequence[torch.Tensor]) -> None:
    """In-place mean over ranks of a list of tensors with ONE collective (flattened); the MeanMetric reduction of the reference
    (utils/metrics.py:31) for scalars."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t).to(t.dtype))
        o += n

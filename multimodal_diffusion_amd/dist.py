"""Multi-GPU layout for the sampler: independent samples shard over ranks, one broadcast of the conditioning.

Samples never interact inside the loop (attention is within-sample, norms per token, DDIM elementwise), so the
only collective is a root→all broadcast of the conditioning latents before the loop (RCCL over xGMI when the
backend is "nccl"; gloo on CPU for tests).  No per-step communication.  One process per GPU.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the default group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global batch owned by ``rank`` (sizes differ by at most one)."""
    if global_batch < 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad shard request")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_conditioning(cond: Optional[torch.Tensor], shape: Tuple[int, ...], device: torch.device,
                           src: int = 0) -> torch.Tensor:
    """Root holds ``cond`` [B_global, ...]; every rank returns the full tensor after ONE broadcast."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        if cond is None:
            raise ValueError("single-process run needs the conditioning tensor")
        return cond.to(device)
    rank = dist.get_rank()
    buf = cond.to(device=device, dtype=torch.float32).contiguous() if rank == src else \
        torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(buf.shape) != tuple(shape):
        raise ValueError("conditioning shape mismatch on the source rank")
    dist.broadcast(buf, src=src)
    return buf


def local_conditioning(cond_global: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(cond_global.shape[0], rank, world)
    return cond_global[lo:hi].contiguous()


def gather_batch(local: torch.Tensor, global_batch: int) -> torch.Tensor:
    """Optional epilogue of the sharded loop: every rank returns the finished latents of the WHOLE batch,
    [global_batch, ...], shards in rank order (one all-gather; shard sizes may differ by one sample)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    if local.shape[0] != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("local shard does not match shard_range")
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf.contiguous())
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def run_sharded(n_items: int, cond: Optional[torch.Tensor], cond_shape: Tuple[int, ...], comm_device: torch.device, fn,
                src: int = 0) -> torch.Tensor:
    """The data-parallel layout of SURVEY 8e as one call: ``cond`` [n_items, ...] (held by rank ``src``; the others pass None) is
    broadcast ONCE, rank r runs ``fn(cond[lo:hi], lo, hi)`` on its contiguous shard ``[lo, hi) = shard_range(n_items, r, world)``
    — ``fn`` returns a tensor whose first dimension is ``hi - lo`` — and every rank gets the results of the WHOLE batch back in
    item order (one all-gather, ragged shards allowed).  No communication inside ``fn``.  Single process: ``fn(cond, 0, n_items)``.
    A rank whose shard is empty (more ranks than items) calls ``fn`` with an empty slice and must return an empty tensor."""
    if tuple(cond_shape)[:1] != (n_items,):
        raise ValueError("cond_shape must start with n_items")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    full = broadcast_conditioning(cond, tuple(cond_shape), comm_device, src=src)
    lo, hi = shard_range(n_items, rank, world)
    out = fn(full[lo:hi], lo, hi)
    if out.shape[0] != hi - lo:
        raise ValueError(f"fn returned {out.shape[0]} items for the shard [{lo}, {hi})")
    if world == 1:
        return out
    back = out.device
    return gather_batch(out.to(comm_device).contiguous(), n_items).to(back)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scalars(value: float, device: torch.device) -> list:
    """[value of rank 0, value of rank 1, ...] on every rank (verification aid, outside the data path)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

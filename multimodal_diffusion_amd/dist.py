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
                           src: int = 0, error: Optional[BaseException] = None) -> torch.Tensor:
    """Root holds ``cond`` [B_global, ...]; every rank returns the full tensor after ONE broadcast.

    The broadcast buffer carries one status word behind the payload.  ``error`` (source rank only): something went wrong while the
    conditioning was being prepared (a failed encode, a shape the config does not imply).  The source then still enters the
    broadcast — with the status word cleared — and raises ``error`` afterwards; every other rank raises a ``RuntimeError`` instead of
    waiting for a broadcast that never comes."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        if error is not None:
            raise error
        if cond is None:
            raise ValueError("single-process run needs the conditioning tensor")
        return cond.to(device)
    rank = dist.get_rank()
    n = 1
    for s_ in shape:
        n *= int(s_)
    buf = torch.zeros(n + 1, dtype=torch.float32, device=device)
    if rank == src:
        if error is None and (cond is None or tuple(cond.shape) != tuple(shape)):
            error = ValueError("conditioning shape mismatch on the source rank")
        if error is None:
            buf[:n] = cond.to(device=device, dtype=torch.float32).reshape(-1)
            buf[n] = 1.0
    dist.broadcast(buf, src=src)
    if rank == src and error is not None:
        raise error
    if float(buf[n].item()) != 1.0:
        raise RuntimeError(f"rank {src} failed while preparing the conditioning (its own log has the error); nothing was broadcast")
    return buf[:n].view(tuple(shape))


def local_conditioning(cond_global: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(cond_global.shape[0], rank, world)
    return cond_global[lo:hi].contiguous()


def gather_batch(local: torch.Tensor, global_batch: int, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Optional epilogue of the sharded loop: the finished items of the WHOLE batch, [global_batch, ...], shards in rank
    order (shard sizes may differ by one item; any dtype).  ``dst=None``: one all-gather, every rank returns the batch.  ``dst=r``: one
    gather to rank r — the others return None and receive nothing (decoded frames are 9.4 MB per 256 x 256 window: an all-gather
    would push every rank's frames over every xGMI link for a result only the stitching rank reads)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    if local.shape[0] != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("local shard does not match shard_range")
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    buf = buf.contiguous()
    if dst is None:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst)
        if rank != dst:
            return None
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def run_sharded(n_items: int, cond: Optional[torch.Tensor], cond_shape: Tuple[int, ...], comm_device: torch.device, fn,
                src: int = 0, result: str = "all", error: Optional[BaseException] = None) -> Optional[torch.Tensor]:
    """The data-parallel layout of SURVEY 8e as one call: ``cond`` [n_items, ...] (held by rank ``src``; the others pass None) is
    broadcast ONCE, rank r runs ``fn(cond[lo:hi], lo, hi)`` on its contiguous shard ``[lo, hi) = shard_range(n_items, r, world)``
    — ``fn`` returns a tensor whose first dimension is ``hi - lo`` (any dtype: latents, or what the rank DECODED from them) — and the
    results come back in item order: ``result="all"`` on every rank (one all-gather), ``result="root"`` on rank ``src`` only (one
    gather; the other ranks return None).  Ragged shards allowed.  No communication inside ``fn``.  Single process:
    ``fn(cond, 0, n_items)``.  A rank whose shard is empty (more ranks than items) calls ``fn`` with an empty slice and must return an
    empty tensor with the same trailing shape.  ``error``: see ``broadcast_conditioning``."""
    if tuple(cond_shape)[:1] != (n_items,):
        raise ValueError("cond_shape must start with n_items")
    if result not in ("all", "root"):
        raise ValueError("result must be 'all' or 'root'")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    full = broadcast_conditioning(cond, tuple(cond_shape), comm_device, src=src, error=error)
    lo, hi = shard_range(n_items, rank, world)
    out = fn(full[lo:hi], lo, hi)
    if out.shape[0] != hi - lo:
        raise ValueError(f"fn returned {out.shape[0]} items for the shard [{lo}, {hi})")
    if world == 1:
        return out
    back = out.device
    got = gather_batch(out.to(comm_device).contiguous(), n_items, dst=None if result == "all" else src)
    return None if got is None else got.to(back)


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

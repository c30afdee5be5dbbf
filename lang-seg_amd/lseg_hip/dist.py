"""Data-parallel plumbing for the one way this path shards: independent images (BASELINE config
3).  One process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo"
on CPU for tests).  Inference needs NO data-path collective: weights are replicated once, every
rank runs the engine on its own slice of the batch, and only host-side results (masks / metric
counters / timings) are gathered.
"""
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when
    WORLD_SIZE > 1.  Rendezvous defaults to 127.0.0.1 (container hostnames may not resolve)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) slice of n_items for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    s, e = shard_range(x.shape[0], rank, world)
    return x[s:e]


def gather_to_rank0(t: torch.Tensor, sizes: Sequence[int]) -> Optional[torch.Tensor]:
    """Concatenate per-rank host tensors (dim 0, ragged sizes) on rank 0.  Host-side result
    collection only -- never on the forward's critical path."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    buf[: t.shape[0]] = t
    outs = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, outs, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)


def max_over_ranks(value: float, device: str = "cpu") -> float:
    """Bench timing reduction: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(t: torch.Tensor) -> torch.Tensor:
    """Metric counters (pixAcc / intersection / union histograms) are additive across shards."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

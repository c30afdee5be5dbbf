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


# ---- training exchange step (BASELINE config 4): ONE collective family ---------------------------------
class GradBucketer:
    """Bucketed gradient all-reduce for data-parallel training: the single collective of the path
    (the reference gets it from Lightning's DDP, `train_lseg.py` -> accelerator="ddp").

    Parameters are grouped into buckets in REVERSE forward order (head -> refinenets -> ViT block 23..0 -> patch embed),
    one bucket per ViT block (12.6 M parameters = 50 MB fp32: large enough that a ring over the 7 x ~153 GB/s xGMI
    links is bandwidth- not latency-bound, small enough to start while earlier blocks are still in backward).
    `ready(i)` is called by the backward as soon as bucket i's gradients are complete: the bucket is flattened and its
    all-reduce is launched asynchronously (on a side stream for CUDA/RCCL, so it overlaps the remaining dgrad/wgrad
    GEMMs); `finish()` waits for every handle, divides by the world size (DDP's mean) and scatters the result back into
    `.grad`.  Buckets never change after construction, so the flat buffers are allocated once.
    """

    def __init__(self, named_params, bucket_key=None, dtype: Optional[torch.dtype] = None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        key = bucket_key or default_bucket_key
        groups, order = {}, []
        for name, p in named_params:
            if not p.requires_grad:
                continue
            k = key(name)
            if k not in groups:
                groups[k] = []
                order.append(k)
            groups[k].append((name, p))
        self.keys = order[::-1]                                   # reverse forward order = backward completion order
        self.buckets = [groups[k] for k in self.keys]
        self.flat = []
        for b in self.buckets:
            n = sum(p.numel() for _, p in b)
            dev = b[0][1].device
            self.flat.append(torch.zeros(n, dtype=dtype or b[0][1].dtype, device=dev))
        self._work = [None] * len(self.buckets)
        self._stream = torch.cuda.Stream() if (torch.cuda.is_available() and self.flat and self.flat[0].is_cuda) else None

    def __len__(self):
        return len(self.buckets)

    def ready(self, i: int):
        """Gradients of bucket i are final: flatten and launch its all-reduce (asynchronous)."""
        flat, off = self.flat[i], 0
        for _, p in self.buckets[i]:
            n = p.numel()
            g = p.grad if p.grad is not None else torch.zeros_like(p)     # unused parameters (find_unused_parameters=True)
            flat[off:off + n].copy_(g.reshape(-1))
            off += n
        if self.world == 1:
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Wait for all buckets, average, write back into .grad (call once per step, before the optimiser)."""
        for i, b in enumerate(self.buckets):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None
            flat, off = self.flat[i], 0
            if self.world > 1:
                flat.div_(self.world)
            for _, p in b:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


def default_bucket_key(name: str) -> str:
    """One bucket per ViT block / per refinenet / per remaining top-level group (state-dict names of SURVEY App. B)."""
    parts = name.split(".")
    if name.startswith("pretrained.model.blocks.") or name.startswith("net.pretrained.model.blocks."):
        i = parts.index("blocks")
        return ".".join(parts[: i + 2])
    if "refinenet" in name:
        i = [k for k, s in enumerate(parts) if s.startswith("refinenet")][0]
        return ".".join(parts[: i + 1])
    if "clip_pretrained" in parts:
        return "clip_pretrained"
    i = 2 if parts[0] == "net" else 1
    return ".".join(parts[: i + 1]) if len(parts) > i else name

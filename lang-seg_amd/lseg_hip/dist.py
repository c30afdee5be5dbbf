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


def gather_floats(value: float, device: str = "cpu") -> List[float]:
    """Every rank's value, in rank order, on every rank (the bench line's per-rank images/s)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


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
    """Bucketed gradient all-reduce over module parameters (the reference gets it from Lightning's DDP, `train_lseg.py` ->
    accelerator="ddp") -- the torch.autograd-facing front of `lseg_hip.train.BucketExchange`.

    Parameters are grouped with `lseg_hip.train.grad_bucket_index` -- the same rule the HIP engine uses (lseg_grad_bucket): bucket
    0 = DPT head + reassemble (complete first in the backward), 1 + j = ViT block depth-1-j with the readout hooked on it, the
    last one also carries patch_embed / cls_token / pos_embed.  Every bucket is ONE flat buffer and each parameter's `.grad` is
    made a view into it, so `ready(i)` launches the all-reduce in place (no flatten / scatter copies) and `finish()` only waits
    and divides.  Parameters without a gradient (find_unused_parameters=True in the reference) contribute zeros.  `finish()`
    raises if a bucket was not readied this step (ranks that disagree would hang on mismatched collectives)."""

    def __init__(self, named_params, depth: int, hooks, dtype: Optional[torch.dtype] = None, group=None):
        from .train import BucketExchange, grad_bucket_index
        named = [(n, p) for n, p in named_params if p.requires_grad]
        nb = depth + 1
        groups = [[] for _ in range(nb)]
        for n, p in named:
            b = grad_bucket_index(n[4:] if n.startswith("net.") else n, depth, hooks)
            if b >= 0:
                groups[b].append((n, p))
        self.buckets = groups
        self.keys = [[n for n, _ in g] for g in groups]
        self.flat = []
        self.views = []
        any_dev = named[0][1].device if named else torch.device("cpu")       # empty buckets live where the parameters do (nccl needs that)
        for g in groups:
            n = sum(p.numel() for _, p in g)
            dev = g[0][1].device if g else any_dev
            flat = torch.zeros(max(n, 1), dtype=dtype or torch.float32, device=dev)
            off = 0
            vs = []
            for _, p in g:
                view = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view                                  # autograd accumulates into the bucket from now on
                vs.append(view)
                off += p.numel()
            self.flat.append(flat)
            self.views.append(vs)
        self.exchange = BucketExchange(self.flat, group)
        self.world = self.exchange.world

    def __len__(self):
        return len(self.buckets)

    def ready(self, i: int):
        """Bucket i's gradients are final.  `optimizer.zero_grad()` defaults to set_to_none=True (also Lightning's default), which drops
        the bucket views; autograd then allocates fresh .grad tensors: their values are copied into the bucket and the views are
        re-attached, so the flat buffer that is all-reduced always holds this step's gradients (zeros for parameters without one)."""
        for (_, p), view in zip(self.buckets[i], self.views[i]):
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
        self.exchange.ready(i)

    def finish(self):
        self.exchange.finish()

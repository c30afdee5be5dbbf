"""The data-parallel training step of the path (BASELINE config 4).

What the reference gets from Lightning (`train_lseg.py` -> utils.do_training: accelerator="ddp", sync_batchnorm=True,
utils.py:20-22,34) around `LSegmentationModule.training_step` (modules/lsegmentation_module.py:66-81):

  * DistributedDataParallel's bucketed gradient all-reduce  ->  `BucketExchange`: the HIP engine writes every gradient straight
    into one flat fp32 buffer per bucket (lseg_bind_grad); as soon as the last kernel of a bucket is enqueued the engine calls
    back, the bucket's RCCL all-reduce is launched IN PLACE on a side stream (ordered behind the compute stream by an event) and
    runs under the remaining backward GEMMs.  Buckets = DPT head | ViT block 23 | ... | ViT block 0 + embeddings (12.6 M
    parameters = 50 MB fp32 per block: bandwidth-bound on the 7 x ~153 GB/s xGMI links).  No flatten / scatter copies.
  * SyncBatchNorm  ->  `bn_sync`: the 2C per-layer sums are all-reduced IN PLACE on the engine's buffers between the statistics and the
    normalisation kernels (forward) and between the gradient sums and the dx kernel (backward): 56 x 2 KB latency-bound collectives per
    step -- each BatchNorm's statistics feed the next conv of the same unit, so they cannot be batched without changing the reference's
    semantics (torch SyncBatchNorm issues one collective per layer and direction as well).  `sync_bn=False` keeps per-GPU statistics (a declared deviation: no collective besides the
    gradient all-reduce, as BASELINE.json's north_star words it).
  * SGD(momentum 0.9, weight decay 1e-4) with the two learning-rate groups of configure_optimizers (:119-127,165-171)
    ->  the engine's fused `lseg_sgd_step` on the fp32 masters (torch.optim works too: the gradients are ordinary tensors).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests of `BucketExchange`.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def grad_bucket_index(key: str, depth: int, hooks) -> int:
    """Bucket of a state-dict key (relative to `net.`): the Python mirror of the engine's lseg_grad_bucket (csrc/train.hip
    Engine::bucket_of; tests/test_gpu_train.py holds the two together).  -1 = not a trainable parameter of the path."""
    bp = "pretrained.model.blocks."
    if key.startswith(bp):
        return 1 + (depth - 1 - int(key[len(bp):].split(".")[0]))
    ap = "pretrained.act_postprocess"
    if key.startswith(ap):
        l = int(key[len(ap)]) - 1
        if key[len(ap) + 1:].startswith(".0."):
            return 1 + (depth - 1 - hooks[l])
        return 0
    if key.startswith("scratch."):
        return 0
    if key.startswith("pretrained.model."):
        return depth
    return -1


class BucketExchange:
    """In-place, asynchronous mean all-reduce of flat gradient buckets in backward-completion order."""

    def __init__(self, buckets: List[torch.Tensor], group=None):
        self.buckets = buckets
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._work = [None] * len(buckets)
        self._ready = [False] * len(buckets)
        cuda = any(b.is_cuda for b in buckets)
        self._dev = next((b.device for b in buckets if b.is_cuda), None)
        self._stream = torch.cuda.Stream(device=self._dev) if cuda else None
        self._event = torch.cuda.Event() if cuda else None
        # DDP's mean: RCCL reduces with ncclAvg directly (no extra launch); gloo (the CPU tests) has no AVG, there the sums are divided once,
        # all buckets in ONE multi-tensor launch (25 separate div_ kernels per step before)
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._avg_op = backend == "nccl" and hasattr(dist.ReduceOp, "AVG")
        self.op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
        # measurement (bench.py --gpus N): per-bucket start / end events on the exchange stream when enabled
        self.timing = False
        self._tev = {}

    def __len__(self):
        return len(self.buckets)

    def ready(self, i: int):
        """Bucket i's gradients are complete on the current stream: launch its all-reduce behind them."""
        if self._ready[i]:
            raise RuntimeError(f"gradient bucket {i} was readied twice in one step")
        self._ready[i] = True
        if self.world == 1:
            return
        flat = self.buckets[i]
        if self._stream is not None:
            self._event.record(torch.cuda.current_stream(flat.device))
            self._stream.wait_event(self._event)
            with torch.cuda.stream(self._stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._stream)
                self._work[i] = dist.all_reduce(flat, op=self.op, group=self.group, async_op=True)
                if self.timing:
                    self._work[i].wait()                  # stream-ordered wait (no host block): the end event lands behind the collective
                    e1.record(self._stream)
                    self._tev[i] = (e0, e1)
        else:
            self._work[i] = dist.all_reduce(flat, op=self.op, group=self.group, async_op=True)

    def finish(self):
        """Every bucket must have been readied this step (ranks that disagree would hang on mismatched collectives): wait,
        turn the sums into DDP's mean, hand the buffers back to the compute stream."""
        missing = [i for i, r in enumerate(self._ready) if not r]
        if missing:
            raise RuntimeError(f"gradient buckets {missing} were never readied this step")
        pending = [i for i, w in enumerate(self._work) if w is not None]
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                for i in pending:
                    self._work[i].wait()
                if pending and not self._avg_op:
                    torch._foreach_div_([self.buckets[i] for i in pending], float(self.world))
        else:
            for i in pending:
                self._work[i].wait()
            if pending and not self._avg_op:
                torch._foreach_div_([self.buckets[i] for i in pending], float(self.world))
        for i in pending:
            self._work[i] = None
        if self._stream is not None and self.world > 1:
            torch.cuda.current_stream(self._dev).wait_stream(self._stream)
        self._ready = [False] * len(self.buckets)

    def bucket_times_ms(self):
        """{bucket: milliseconds of its all-reduce on the exchange stream} of the last step run with timing = True (synchronises)."""
        out = {}
        for i, (e0, e1) in self._tev.items():
            e1.synchronize()
            out[i] = e0.elapsed_time(e1)
        return out

    def abort(self):
        """Forget a step that failed half-way (an exception between ready() calls): wait for what was launched, clear the flags."""
        for i, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[i] = None
        self._ready = [False] * len(self.buckets)


class BnSync:
    """SyncBatchNorm exchange (utils.py:34): sums the engine's 2C per-layer BatchNorm sums over the ranks, in place, ordered on the
    current stream -- between the statistics and the normalisation kernels (forward) and between the gradient sums and dx (backward)."""

    def __init__(self, engine, world: int, group=None):
        self.eng, self.group = engine, group
        self._views = {}
        engine.set_bn_sync(self, world)

    class _Raw:
        """the engine's 2C floats as a zero-copy tensor (CUDA array interface): the collective runs on the engine's memory itself"""
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    def __call__(self, ptr: int, n: int):
        v = self._views.get((ptr, n))
        if v is None:
            v = torch.as_tensor(self._Raw(ptr, n), device=self.eng.device)
            assert v.data_ptr() == ptr and v.numel() == n
            self._views[(ptr, n)] = v
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)      # ordered on the current stream (RCCL), blocking on gloo


class DataParallelTrainer:
    """forward (train mode) -> loss + backward with overlapped bucket all-reduce -> fused SGD, on one engine."""

    def __init__(self, engine, state_dict: Dict[str, torch.Tensor], sync_bn: bool = True, group=None):
        self.eng = engine
        engine.enable_training(state_dict)
        self.exchange = BucketExchange(engine.grad_buckets, group)
        self.world = self.exchange.world
        engine.set_bucket_callback(self.exchange.ready)
        self.sync_bn = bool(sync_bn) and self.world > 1
        if self.sync_bn:
            self._bn = BnSync(engine, self.world, group)

    def forward_backward(self, x: torch.Tensor, target: torch.Tensor, ignore_index: int = -1, accumulate: bool = False) -> torch.Tensor:
        """Train-mode forward + fused loss/backward + gradient exchange, without the optimizer step."""
        if not self.eng.training:                        # an eval forward on the same engine (validation) switched it off
            self.eng.set_train(True)
        self.eng.forward(x, want_logits=False)
        loss = self.eng.backward(target=target, ignore_index=ignore_index, accumulate=accumulate)
        self.exchange.finish()
        return loss

    def step(self, x: torch.Tensor, target: torch.Tensor, lr_pretrained: float, lr_scratch: float, momentum: float = 0.9,
             weight_decay: float = 1e-4, ignore_index: int = -1, optimize: bool = True) -> torch.Tensor:
        """One training step; returns the (local) mean cross-entropy as a 0-dim device tensor, no host synchronisation."""
        loss = self.forward_backward(x, target, ignore_index)
        if optimize:
            self.eng.sgd_step(lr_pretrained, lr_scratch, momentum, weight_decay)
        return loss

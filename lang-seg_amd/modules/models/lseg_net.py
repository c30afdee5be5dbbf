"""LSegNet -- drop-in for the reference's modules/models/lseg_net.py:104-226.

Same constructor, same module tree / state-dict keys (`clip_pretrained.*`, `pretrained.*`,
`scratch.*`), same `forward(x, labelset='') -> float32 [B, K, H, W]`; the arithmetic of
LSeg.forward (lseg_net.py:160-205) runs in the MI355X HIP engine (lseg_hip, C ABI in
include/lseg_hip.h).  There is NO PyTorch/CPU fallback: without the built extension or without
a GPU the forward raises.
"""
import math
import os
import threading
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from lseg_hip.config import get_config
from lseg_hip.tokenizer import tokenize
from .lseg_blocks import FeatureFusionBlock_custom, Interpolate, _make_encoder
from .lseg_vit import _NoForward

# MFMA operand type of inference engines unless the constructor says otherwise (image_dtype=...); see LSeg.__init__.
# The environment is read when a network is CONSTRUCTED (not at import), so LSEG_IMAGE_DTYPE set after `import modules` still counts.
DEFAULT_IMAGE_DTYPE = "fp16"


def default_image_dtype():
    return os.environ.get("LSEG_IMAGE_DTYPE", DEFAULT_IMAGE_DTYPE)


class depthwise_conv(_NoForward):                      # lseg_net.py:29-40
    def __init__(self, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.depthwise = nn.Conv2d(1, 1, kernel_size=kernel_size, stride=stride, padding=padding)


class _head_block(_NoForward):                         # bottleneck_block / depthwise_block, :43-79
    def __init__(self, activation="relu"):
        super().__init__()
        self.depthwise = depthwise_conv(kernel_size=3, stride=1, padding=1)
        self.activation_name = activation


class bottleneck_block(_head_block):
    pass


class depthwise_block(_head_block):
    pass


class BaseModel(torch.nn.Module):
    def load(self, path):
        """Load model from file (lseg_net.py:82-92)."""
        parameters = torch.load(path, map_location=torch.device("cpu"))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        self.load_state_dict(parameters)


def _make_fusion_block(features, use_bn):
    return FeatureFusionBlock_custom(features, activation=nn.ReLU(False), deconv=False, bn=use_bn,
                                     expand=False, align_corners=True)


class _TrainState:
    """What one training engine carries across steps: the gradient exchange (when torch.distributed is up), the lazy zero_grad flag,
    the number of fused optimizer steps (momentum lives in the engine), pending momentum to restore."""

    def __init__(self):
        self.exchange = None
        self.lazy_zero = True            # no gradient values to add to yet: the next backward overwrites
        self.fresh = False               # the engine's buckets hold this step's (exchanged) gradients -> EngineSGD may use the fused step
        self.sgd_steps = 0
        self.sync_bn = False


class _EngineTrainFn(torch.autograd.Function):
    """Autograd node of the train-mode forward with the LOGITS as output (any criterion): `loss.backward()` (Lightning's, or
    anyone's) lands in lseg_backward with the gradient of the logits.  The parameters are inputs only so that autograd knows the
    output depends on them; the arithmetic is all in the HIP engine and the gradients are written straight into the flat buckets
    behind each parameter's .grad (or handed to autograd when the net was built with autograd_grads=True: DDP-wrapper mode)."""

    @staticmethod
    def forward(ctx, x, net, eng, keys, *params):
        ctx.net, ctx.eng, ctx.keys = net, eng, keys
        return eng.forward(x)

    @staticmethod
    def backward(ctx, dlogits):
        grads = ctx.net._engine_backward(ctx.eng, ctx.keys, dlogits=dlogits)
        return (None, None, None, None) + grads


class _EngineLossFn(torch.autograd.Function):
    """Autograd node of forward + criterion for the plain cross-entropy the reference trains with (SegmentationLosses with
    se_loss=False, aux=False == nn.CrossEntropyLoss(ignore_index), lsegmentation_module.py:72,236-244): the [B,K,H,W] logits and their
    gradient (2 x 1.1 GB at B = 8) never exist -- the loss VALUE comes from lseg_train_loss on the low-resolution logits through the
    x2 bilinear, the backward is lseg_backward_scaled(target, d(loss)) with d(loss) read on the device (no host synchronisation)."""

    @staticmethod
    def forward(ctx, x, target, net, eng, keys, ignore_index, *params):
        ctx.net, ctx.eng, ctx.keys, ctx.ignore_index = net, eng, keys, ignore_index
        eng.forward(x, want_logits=False)
        t = target.detach().to(eng.device, torch.int64).contiguous()
        ctx.target = t
        loss, counts = eng.train_loss(t, ignore_index, want_counts=True)
        net._last_train_counts = counts
        return loss

    @staticmethod
    def backward(ctx, g):
        grads = ctx.net._engine_backward(ctx.eng, ctx.keys, target=ctx.target, ignore_index=ctx.ignore_index, grad_scale=g)
        return (None, None, None, None, None, None) + grads


def _new_shared(image_dtype):
    return _SharedState(image_dtype=image_dtype, lock=threading.RLock())


class _SharedState(dict):
    """Per-network state that DataParallel.replicate's shallow module copies share (one dict object behind every replica): the
    operand type of the inference engines and the lock that guards a fallback.  Copies / pickles get a fresh lock."""

    def __deepcopy__(self, memo):
        return _new_shared(self["image_dtype"])

    def __reduce__(self):
        return (_new_shared, (self["image_dtype"],))


class LSeg(BaseModel):
    def __init__(self, head, features=256, backbone="clip_vitl16_384", readout="project",
                 channels_last=False, use_bn=False, **kwargs):
        super().__init__()
        self.channels_last = channels_last
        if readout != "project":
            raise NotImplementedError("the HIP engine implements readout='project' (the only mode LSegNet uses)")
        self.arch_option = kwargs["arch_option"]
        act = kwargs.get("activation", "lrelu")
        self.block_depth = kwargs.get("block_depth", 0) if self.arch_option in (1, 2) else 0
        self.cfg = get_config(backbone, features=features, arch_option=self.arch_option,
                              block_depth=self.block_depth, activation=act)
        self.clip_pretrained, self.pretrained, self.scratch = _make_encoder(self.cfg)
        for r in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{r}", _make_fusion_block(features, use_bn))
        # plain fp32 tensor, not a Parameter/buffer (absent from the state dict): lseg_net.py:141
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07)).exp()
        self.out_c = self.cfg.out_c
        self.scratch.head1 = nn.Conv2d(features, self.out_c, kernel_size=1)
        if self.arch_option == 1:
            self.scratch.head_block = bottleneck_block(activation=act)
        elif self.arch_option == 2:
            self.scratch.head_block = depthwise_block(activation=act)
        self.scratch.output_conv = head
        self.text = tokenize(self.labels, self.cfg.text.ctx, self.cfg.text.vocab)     # lseg_net.py:158
        self._engines = OrderedDict()            # (H, W, device index, train) -> HipEngine, least recently used first
        self.max_engines = kwargs.get("max_engines", 4)
        # MFMA operand type of the image tower.  The reference's tower is fp32; of the two 16-bit types fp16 (11 significand bits) is 8x
        # closer to it than bf16 (8) at the same MFMA rate (DESIGN.md par. 4, bench.py `dtype_selection`).  Training engines are bf16:
        # the per-logit gradient of a mean over 1.8 M pixels is ~1e-7 and would flush to zero in fp16.
        # (kept in a dict that DataParallel.replicate's shallow copies SHARE -- like `_engines` -- so that a loud bf16 fallback taken by one
        # replica thread holds for all of them and survives the next replicate: additional_utils/encoding_models.py:43)
        self._shared = _new_shared(kwargs.get("image_dtype", default_image_dtype()))
        # fp16 inference engines check their 16-bit activations for overflow (65504) on the first forward after every (re)pack and on
        # demand (`check_overflow`): a network whose activations leave the fp16 range falls back to bf16 LOUDLY (warning + rebuild),
        # never to inf / NaN masks (lseg_get_overflow)
        self.overflow_fallback = kwargs.get("overflow_fallback", True)
        # training: the reference's head gradient is fp16-subnormal arithmetic (DESIGN par. 3.6) and the engine reproduces it by
        # default; exact_head_grad=True (or LSEG_EXACT_HEAD_GRAD=1) keeps the un-rounded gradient instead (lseg_config.flags bit 1)
        self.exact_head_grad = bool(kwargs.get("exact_head_grad", os.environ.get("LSEG_EXACT_HEAD_GRAD", "") not in ("", "0")))
        self.cache_text = kwargs.get("cache_text", False)
        # image b of a batch == the same image alone to fp32 round-off (no split-K at small batches): validation / regression runs
        self.batch_invariant = kwargs.get("batch_invariant", False)
        # DDP-wrapper mode: hand the parameter gradients to autograd (DistributedDataParallel's hooks then reduce them) instead of
        # writing them behind .grad and exchanging the flat buckets ourselves
        self.autograd_grads = bool(kwargs.get("autograd_grads", os.environ.get("LSEG_AUTOGRAD_GRADS", "") not in ("", "0")))
        self.sync_batchnorm = kwargs.get("sync_batchnorm", True)       # utils.py:34 (only matters when torch.distributed is initialised)
        self._native_epoch = 0                   # bumped whenever the engine wrote the masters / running statistics through raw pointers
        self._last_train_counts = None

    # ---- engine plumbing -----------------------------------------------------------------------

    @property
    def image_dtype(self):
        return self._shared["image_dtype"]

    @image_dtype.setter
    def image_dtype(self, value):
        self._shared["image_dtype"] = value

    def _engine(self, B, H, W, K, device, train=False):
        """One engine per (image size, device, train/eval): each holds its own packed weights + activation plan (~1 GB for ViT-L/16
        in eval mode).  The cache is bounded (callers with varying image sizes: lseg_app, `evaluate` on uncropped images): the least
        recently used EVAL engine of THIS device is closed when more than `max_engines` exist; training engines (momentum, bound
        gradients, possibly referenced by a live autograd graph) are never evicted.  Replicas made by DataParallel.replicate
        (additional_utils/encoding_models.py:43) share this dict object; the device index in the key keeps their engines apart."""
        from lseg_hip.engine import HipEngine
        # cache look-up, eviction and insert under the lock the range guard iterates under: DataParallel replicas (one thread per device)
        # share this dict
        with self._shared["lock"]:
            # the operand type is part of the key: after a fallback (fp16 -> bf16) an fp16 engine is never picked up again, on any device
            key = (H, W, device.index, bool(train), "bf16" if train else self.image_dtype)
            eng = self._engines.get(key)
            if eng is not None:
                self._engines.move_to_end(key)
            else:
                mine = [k for k in self._engines if k[2] == device.index and not k[3] and self._engines[k] is not getattr(self, "_last_engine", None)]
                n_eval = len([k for k in self._engines if k[2] == device.index and not k[3]])
                while mine and n_eval >= max(1, self.max_engines):
                    self._engines.pop(mine.pop(0)).close()
                    n_eval -= 1
            if eng is None or eng.max_batch < B or eng.max_labels < K:
                carried = None
                if eng is not None:
                    if train and getattr(eng, "_ts", None) is not None and eng._ts.sgd_steps > 0:
                        # a bigger batch on a training engine: rebuild it, but the optimizer state moves over
                        carried = ({k: eng.get_momentum(k) for k in eng.grads}, eng._ts.sgd_steps)
                        torch.cuda.current_stream(device).synchronize()
                    eng.close()
                eng = HipEngine(self.cfg, H, W, max_batch=max(B, eng.max_batch if eng else 1),
                                max_labels=max(K, eng.max_labels if eng else 1), device=device,
                                image_dtype="bf16" if train else self.image_dtype,
                                exact_head_grad=bool(getattr(self, "exact_head_grad", False)),
                                batch_invariant=bool(getattr(self, "batch_invariant", False)))
                eng._stamp = None
                eng._tok = None
                eng._ts = None
                eng._carried = carried
                self._engines[key] = eng
        stamp = self._stamp()
        if eng._stamp != stamp:                       # first use, load_state_dict, torch.optim step, .cuda(), a fused step on another engine ...
            eng.load_state_dict(self.state_dict())
            eng._stamp = stamp
            eng._tok = None
        return eng

    def _stamp(self):
        # (storage, version) of every tensor of the state dict: changes on load_state_dict, optimizer steps, .cuda(), .half() ...
        # plus a counter for what tensor versions cannot see: the fused SGD step and the train-mode BatchNorm write the masters /
        # running statistics through raw pointers.
        # The tensor list is cached (building the state dict costs ~1 ms on ViT-L; it only changes when a module is replaced, which
        # _apply / load_state_dict signal by bumping _stamp_epoch)
        ep = getattr(self, "_stamp_epoch", 0)
        if getattr(self, "_stamp_cache", None) is None or self._stamp_cache[0] != ep:
            self._stamp_cache = (ep, [p for p in self.state_dict(keep_vars=True).values() if isinstance(p, torch.Tensor)])
        return (self._native_epoch,) + tuple((p.data_ptr(), p._version) for p in self._stamp_cache[1])

    def invalidate_engines(self, except_=None):
        """The engine `except_` changed the bound tensors in place without moving their versions (lseg_sgd_step on the masters,
        train-mode BatchNorm on the running statistics): every OTHER cached engine (another image size: uncropped validation,
        lseg_app) must re-bind and re-pack before its next forward; `except_` itself is already consistent."""
        self._native_epoch += 1
        if except_ is not None:
            except_._stamp = self._stamp()

    def _apply(self, fn, *a, **k):
        self._stamp_epoch = getattr(self, "_stamp_epoch", 0) + 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._stamp_epoch = getattr(self, "_stamp_epoch", 0) + 1
        return super().load_state_dict(*a, **k)

    def forward_metrics(self, x, target, labelset="", ignore_index=-1):
        """evaluate(x, target) of the Lightning module without the full-resolution logits (lseg_forward_stats)."""
        was = self.training
        self.eval()
        try:
            with torch.no_grad():
                self.forward(x, labelset, _want_logits=False)
            eng = self._last_engine
            return eng.forward_stats(target, ignore_index)
        finally:
            self.train(was)

    # ---- training step plumbing (lsegmentation_module.py:66-81 + what Lightning's DDP does around it) -----------------------------
    def _train_engine(self, B, H, W, K, device):
        import torch.distributed as dist
        from lseg_hip.train import BucketExchange
        eng = self._engine(B, H, W, K, device, train=True)
        if eng._ts is None:
            sd = self.state_dict()
            eng.enable_training({k: v for k, v in sd.items()})
            ts = _TrainState()
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            if world > 1 and not self.autograd_grads:
                ts.exchange = BucketExchange(eng.grad_buckets)
                eng.set_bucket_callback(ts.exchange.ready)
            if world > 1 and self.sync_batchnorm:
                # SyncBatchNorm (utils.py:34) belongs to the FORWARD/backward arithmetic, not to the gradient exchange: it is installed in
                # DDP-wrapper mode (autograd_grads=True, Lightning accelerator='ddp' reducing the gradients itself) as well -- there the
                # reference runs torch.nn.SyncBatchNorm under DistributedDataParallel
                from lseg_hip.train import BnSync
                ts.bn = BnSync(eng, world)
                ts.sync_bn = True
            eng._ts = ts
            eng._named = [(k, p) for k, p in self.named_parameters() if k in eng.grads]
            eng._nbt = [b for k, b in self.named_buffers() if k.endswith("num_batches_tracked") and k.startswith("scratch.")
                        and ".refinenet4.resConfUnit1." not in k]
            if eng._carried is not None:
                mom, steps = eng._carried
                for k, v in mom.items():
                    eng.set_momentum(k, v)
                eng.mark_sgd_initialized(True)
                ts.sgd_steps = steps
                eng._carried = None
        if not eng.training:
            eng.set_train(True)
        return eng

    def _set_tokens(self, eng, text, labelset):
        tkey = (tuple(text.shape), text.data_ptr() if labelset == "" else hash(text.numpy().tobytes()))
        if eng._tok != tkey:
            eng.set_tokens(text)
            eng._tok = tkey
        eng.set_text_cache(bool(self.cache_text))

    def _engine_backward(self, eng, keys, dlogits=None, target=None, ignore_index=-1, grad_scale=None):
        """Runs lseg_backward and puts the gradients where the caller's optimizer looks for them.  Returns the tuple autograd gets for
        the parameter inputs: Nones (the engine wrote behind .grad itself) or, in DDP-wrapper mode, the gradient tensors."""
        ts = eng._ts
        named = eng._named
        if self.autograd_grads:
            acc = False                                  # autograd accumulates into .grad itself; the buckets are scratch
        else:
            # accumulate (accumulate_grad_batches, train.sh) iff every .grad still is the engine's view AND holds values: after
            # zero_grad(set_to_none=True) / a first step the views are re-attached and overwritten; EngineSGD.zero_grad only flags
            attached = all(p.grad is not None and p.grad.data_ptr() == eng.grads[k].data_ptr() for k, p in named)
            acc = attached and not ts.lazy_zero
        try:
            if dlogits is not None:
                eng.backward(dlogits=dlogits, accumulate=acc)
            else:
                eng.backward(target=target, ignore_index=ignore_index, accumulate=acc, grad_scale=grad_scale)
            if ts.exchange is not None:
                ts.exchange.finish()
        except BaseException:
            if ts.exchange is not None:
                ts.exchange.abort()
            raise
        ts.lazy_zero = False
        ts.fresh = True
        if self.autograd_grads:
            return tuple(eng.grads[k] for k in keys)
        for k, p in named:
            p.grad = eng.grads[k]
        return tuple(None for _ in keys)

    def _train_inputs(self, x, labelset):
        text = self.text if labelset == "" else tokenize(labelset, self.cfg.text.ctx, self.cfg.text.vocab)
        if not x.is_cuda:
            raise RuntimeError("LSegNet needs CUDA/HIP tensors (no CPU path, like the reference: lseg_vit.py:224)")
        B, _, H, W = x.shape
        eng = self._train_engine(B, H, W, text.shape[0], x.device)
        self._set_tokens(eng, text, labelset)
        if eng._nbt:
            torch._foreach_add_(eng._nbt, 1)             # nn.BatchNorm2d.num_batches_tracked (state-dict parity; momentum is fixed at 0.1)
        self.invalidate_engines(except_=eng)             # train-mode BatchNorm moves the running statistics through raw pointers
        return eng, tuple(k for k, _ in eng._named), [p for _, p in eng._named]

    def forward_loss(self, x, target, labelset="", ignore_index=-1):
        """`criterion(self(x), target)` for the reference's criterion (mean cross-entropy over pixels != ignore_index) as ONE autograd
        node: the value, and under `loss.backward()` the gradients, without the [B,K,H,W] logits.  Also leaves the pixel-accuracy
        counts {correct, labeled} of this batch in `self._last_train_counts` (int64[2], device)."""
        if not (self.training and torch.is_grad_enabled()):
            raise RuntimeError("forward_loss is the training-step path: call it under net.train() with grad enabled")
        eng, keys, params = self._train_inputs(x, labelset)
        return _EngineLossFn.apply(x.float(), target, self, eng, keys, int(ignore_index), *params)

    def _range_guard(self, eng, device):
        """fp16 MFMA operands saturate at 65504 where the reference's fp32 tower cannot (lseg_vit.py:196-197).  Checked on the first
        forward after every (re)pack of the weights (overflow_fallback="always": on every forward, one synchronising ~1 ms scan); on
        later forwards the engine's asynchronous sentinel (lseg_overflow_seen) reports an input-dependent overflow one call late, which
        triggers the same scan: any inf / NaN in a 16-bit
        activation buffer of an fp16 engine -> this network -- every replica sharing `_shared` -- switches to bf16 operands (same
        speed, fp32 range), LOUDLY, and the caller re-runs.  A bf16 engine that still shows non-finite values (the head map g is fp16
        in every mode: DESIGN par. 3.4) raises -- never inf / NaN masks without an error.  Returns True when the caller must re-run.
        Only THIS device's fp16 eval engines are closed, under the shared lock: under DataParallel-style threaded replicas
        (additional_utils/models.py:229-238) another thread may be inside forward() on its own device's engine."""
        mode = getattr(self, "overflow_fallback", True)
        if not mode:
            return False
        if mode != "always" and getattr(eng, "_range_stamp", None) == eng._stamp:
            # later forwards of the same weights: the engine's always-on sentinel (lseg_overflow_seen: a device flag raised when the head feature
            # map carries inf / NaN, read without synchronising).  An input-dependent overflow on image n is caught at the latest when image
            # n + 1 is submitted: the full scan below then names the damage and the fallback is taken for everything that follows.
            if not eng.overflow_seen(reset=True):
                return False
            warnings.warn("LSeg: a previous forward of this engine left non-finite values in the head feature map (overflow sentinel): its "
                          "output was not usable; checking the activation ranges now", RuntimeWarning, stacklevel=3)
        eng._range_stamp = eng._stamp
        r = eng.check_range()
        if r["nonfinite"] > 0 and eng.image_dtype == "fp16":
            warnings.warn(f"LSeg: {r['nonfinite']} non-finite values in the fp16 image tower's activations (largest finite |x| "
                          f"{r['max_abs']:.3g}; fp16 saturates at 65504): this network's activations leave the fp16 range -- "
                          "falling back to bf16 MFMA operands for all further forwards (image_dtype='bf16')", RuntimeWarning, stacklevel=3)
            with self._shared["lock"]:
                self._shared["image_dtype"] = "bf16"
                for key in [k_ for k_ in self._engines if not k_[3] and k_[2] == device.index and k_[4] == "fp16"]:
                    self._engines.pop(key).close()
            self._last_engine = None
            return True
        if r["nonfinite"] > 0:
            raise RuntimeError(f"LSeg: {r['nonfinite']} non-finite values in the image tower's 16-bit activations with {eng.image_dtype} operands "
                               f"(largest finite |x| {r['max_abs']:.3g}): the head feature map leaves the fp16 range, or the input / weights "
                               "are not finite; image_dtype='strict' keeps fp32-class range and precision")
        self.last_range_check = r
        return False

    def forward(self, x, labelset="", _want_logits=True):
        if labelset == "":
            text = self.text
        else:
            text = tokenize(labelset, self.cfg.text.ctx, self.cfg.text.vocab)         # lseg_net.py:163-164
        if not x.is_cuda:
            raise RuntimeError("LSegNet.forward needs a CUDA/HIP tensor: like the reference (clip.load(device="
                               "'cuda'), lseg_vit.py:224) this network has no CPU path, and the HIP engine has no "
                               "PyTorch fallback")
        B, _, H, W = x.shape
        if self.training and torch.is_grad_enabled():              # net.train() under autograd: training_step (:66-81)
            eng, keys, params = self._train_inputs(x, labelset)
            return _EngineTrainFn.apply(x.float(), self, eng, keys, *params)
        eng = self._engine(B, H, W, text.shape[0], x.device)
        if eng.training:
            eng.set_train(False)
        self._set_tokens(eng, text, labelset)
        self._last_engine = eng
        out = eng.forward(x.float(), want_logits=_want_logits)
        if self._range_guard(eng, x.device):
            return self.forward(x, labelset, _want_logits)              # fell back to bf16: run again on a bf16 engine
        return out


class LSegNet(LSeg):
    """Network for semantic segmentation (lseg_net.py:208-226)."""

    def __init__(self, labels, path=None, scale_factor=0.5, crop_size=480, **kwargs):
        features = kwargs["features"] if "features" in kwargs else 256
        kwargs["use_bn"] = True
        self.crop_size = crop_size
        self.scale_factor = scale_factor
        self.labels = labels
        head = nn.Sequential(Interpolate(scale_factor=2, mode="bilinear", align_corners=True))
        super().__init__(head, **kwargs)
        if path is not None:
            self.load(path)

"""LSegNet -- drop-in for the reference's modules/models/lseg_net.py:104-226.

Same constructor, same module tree / state-dict keys (`clip_pretrained.*`, `pretrained.*`,
`scratch.*`), same `forward(x, labelset='') -> float32 [B, K, H, W]`; the arithmetic of
LSeg.forward (lseg_net.py:160-205) runs in the MI355X HIP engine (lseg_hip, C ABI in
include/lseg_hip.h).  There is NO PyTorch/CPU fallback: without the built extension or without
a GPU the forward raises.
"""
import math
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from lseg_hip.config import get_config
from lseg_hip.tokenizer import tokenize
from .lseg_blocks import FeatureFusionBlock_custom, Interpolate, _make_encoder
from .lseg_vit import _NoForward


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


class _EngineTrainFn(torch.autograd.Function):
    """Autograd node of the train-mode forward: `loss.backward()` (Lightning's, or anyone's) lands in lseg_backward with the
    gradient of the logits; the parameter gradients come back as views of the engine's flat buckets (lseg_bind_grad).
    The parameters are inputs only so that autograd routes their gradients; the arithmetic is all in the HIP engine."""

    @staticmethod
    def forward(ctx, x, net, eng, keys, *params):
        ctx.eng, ctx.keys = eng, keys
        return eng.forward(x)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.eng.backward(dlogits=dlogits)
        grads = tuple(ctx.eng.grads[k] if k in ctx.eng.grads else None for k in ctx.keys)
        return (None, None, None, None) + grads


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
        self._engines = OrderedDict()            # (H, W, device index) -> HipEngine, least recently used first
        self.max_engines = kwargs.get("max_engines", 4)
        self.image_dtype = kwargs.get("image_dtype", "bf16")
        self.cache_text = kwargs.get("cache_text", False)

    # ---- engine plumbing -----------------------------------------------------------------------

    def _engine(self, B, H, W, K, device):
        """One engine per (image size, device): each holds its own packed weights + activation plan (~1 GB for ViT-L/16).  The
        cache is bounded (callers with varying image sizes: lseg_app, `evaluate` on uncropped images): the least recently used
        engine of THIS device is closed when more than `max_engines` exist.  Replicas made by DataParallel.replicate
        (additional_utils/encoding_models.py:43) share this dict object; the device index in the key keeps their engines apart."""
        from lseg_hip.engine import HipEngine
        key = (H, W, device.index)
        eng = self._engines.get(key)
        if eng is not None:
            self._engines.move_to_end(key)
        else:
            mine = [k for k in self._engines if k[2] == device.index]
            while len(mine) >= max(1, self.max_engines):
                self._engines.pop(mine.pop(0)).close()
        if eng is None or eng.max_batch < B or eng.max_labels < K:
            if eng is not None:
                eng.close()
            eng = HipEngine(self.cfg, H, W, max_batch=max(B, eng.max_batch if eng else 1),
                            max_labels=max(K, eng.max_labels if eng else 1), device=device,
                            image_dtype=self.image_dtype)
            eng._stamp = None
            eng._tok = None
            self._engines[key] = eng
        stamp = self._stamp()
        if eng._stamp != stamp:                       # first use, load_state_dict, optimizer step, .cuda() ...
            eng.load_state_dict(self.state_dict())
            eng._stamp = stamp
            eng._tok = None
        return eng

    def _stamp(self):
        # (storage, version) of every tensor of the state dict: changes on load_state_dict, optimizer steps, .cuda(), .half() ...
        # The tensor list is cached (building the state dict costs ~1 ms on ViT-L; it only changes when a module is replaced, which
        # _apply / load_state_dict signal by bumping _stamp_epoch)
        ep = getattr(self, "_stamp_epoch", 0)
        if getattr(self, "_stamp_cache", None) is None or self._stamp_cache[0] != ep:
            self._stamp_cache = (ep, [p for p in self.state_dict(keep_vars=True).values() if isinstance(p, torch.Tensor)])
        return tuple((p.data_ptr(), p._version) for p in self._stamp_cache[1])

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
        eng = self._engine(B, H, W, text.shape[0], x.device)
        train = self.training and torch.is_grad_enabled()          # net.train() under autograd: training_step (:66-81)
        if train and not getattr(eng, "grads", None):
            eng.enable_training({k: v for k, v in self.state_dict().items()})
        if eng.training != train:
            eng.set_train(train)
        tkey = (tuple(text.shape), text.data_ptr() if labelset == "" else hash(text.numpy().tobytes()))
        if eng._tok != tkey:
            eng.set_tokens(text)
            eng._tok = tkey
        eng.set_text_cache(bool(self.cache_text))
        if train:
            named = [(k, p) for k, p in self.named_parameters() if k in eng.grads]
            return _EngineTrainFn.apply(x.float(), self, eng, tuple(k for k, _ in named), *[p for _, p in named])
        self._last_engine = eng
        return eng.forward(x.float(), want_logits=_want_logits)


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

"""LSegNetZS -- drop-in for the reference's zero-shot network, modules/models/lseg_net_zs.py:106-214, 217-240.

Same constructor surface (`LSegNetZS(label_list, path=None, scale_factor=0.5, aux=False, use_relabeled=False,
use_pretrained=True, **kwargs)`), same module tree / state-dict keys as LSegNet, same
`forward(x, class_info) -> float32 [B, 2, H, W]`: image b is scored against ITS OWN label pair
['others', label_list[class_info[b]]] (`self.texts`, lseg_net_zs.py:170-176, 178).  The arithmetic runs in the HIP engine
with per-image label grouping (include/lseg_hip.h: lseg_set_text_grouping); there is no PyTorch fallback.

Only the ViT backbones are implemented (clip_vitl16_384, clip_vitb32_384); the reference's clip_resnet101 / RN50x*
variants (LSegRNNetZS, lseg_net_zs.py:243-363) raise.
"""
import numpy as np
from collections import OrderedDict

import torch
import torch.nn as nn

from lseg_hip.config import get_config
from lseg_hip.tokenizer import tokenize
from .lseg_blocks import Interpolate, _make_encoder
from .lseg_net import BaseModel, default_image_dtype, LSeg as _LSegShared, _make_fusion_block, _new_shared


class LSeg(_LSegShared):
    def __init__(self, head, features=256, backbone="clip_vitl16_384", readout="project", channels_last=False,
                 use_bn=False, **kwargs):
        BaseModel.__init__(self)
        self.channels_last = channels_last
        if readout != "project":
            raise NotImplementedError("the HIP engine implements readout='project' (the only mode LSegNetZS uses)")
        if kwargs.get("arch_option", 0) not in (0, None):
            # the reference stores arch_option but LSeg.forward of the ZS net never runs head blocks (:177-214)
            pass
        self.arch_option = kwargs.get("arch_option", 0)
        self.block_depth = 0
        self.cfg = get_config(backbone, features=features, arch_option=0, block_depth=0,
                              activation=kwargs.get("activation", "lrelu") or "lrelu")
        self.clip_pretrained, self.pretrained, self.scratch = _make_encoder(self.cfg)
        for r in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{r}", _make_fusion_block(features, use_bn))
        self.auxlayer = nn.Sequential(Interpolate(scale_factor=2, mode="bilinear", align_corners=True))   # :150-152
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07)).exp()                           # :155
        self.out_c = self.cfg.out_c
        self.scratch.head1 = nn.Conv2d(features, self.out_c, kernel_size=1)
        self.scratch.output_conv = head
        # one token pair per class: ['others', <class name>]  (:169-176)
        self.texts = [tokenize(["others", name], self.cfg.text.ctx, self.cfg.text.vocab) for name in self.label_list]
        self._engines = OrderedDict()
        self.max_engines = kwargs.get("max_engines", 4)
        self._shared = _new_shared(kwargs.get("image_dtype", default_image_dtype()))
        self.overflow_fallback = kwargs.get("overflow_fallback", True)      # fp16 range check + loud bf16 fallback, as LSegNet (lseg_net.py)
        self.cache_text = kwargs.get("cache_text", False)
        self.autograd_grads = False
        self.sync_batchnorm = False
        self._native_epoch = 0
        self._last_train_counts = None

    def forward(self, x, class_info):
        ids = [int(c) for c in (class_info.tolist() if torch.is_tensor(class_info) else class_info)]
        if not x.is_cuda:
            raise RuntimeError("LSegNetZS.forward needs a CUDA/HIP tensor (no CPU path, no PyTorch fallback)")
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("the HIP engine implements the inference forward only -- call .eval()")
        B, _, H, W = x.shape
        if len(ids) != B:
            raise ValueError(f"class_info has {len(ids)} entries for a batch of {B}")
        text = torch.cat([self.texts[c] for c in ids], dim=0)                 # [2B, ctx]; :178
        eng = self._engine(B, H, W, text.shape[0], x.device)
        tkey = ("zs", tuple(ids))
        if eng._tok != tkey:
            eng.set_tokens(text, labels_per_image=2)
            eng._tok = tkey
        eng.set_text_cache(bool(self.cache_text))
        out = eng.forward(x.float())                                          # [B, 2, H, W]
        if self._range_guard(eng, x.device):
            return self.forward(x, class_info)                                # fp16 overflowed: again on bf16 operands (loud)
        return out


class LSegNetZS(LSeg):
    """Network for zero-shot semantic segmentation (lseg_net_zs.py:217-240)."""

    def __init__(self, label_list, path=None, scale_factor=0.5, aux=False, use_relabeled=False, use_pretrained=True,
                 **kwargs):
        features = kwargs["features"] if "features" in kwargs else 256
        kwargs["use_bn"] = True
        self.scale_factor = scale_factor
        self.aux = aux
        self.use_relabeled = use_relabeled
        self.label_list = label_list
        self.use_pretrained = use_pretrained
        head = nn.Sequential(Interpolate(scale_factor=2, mode="bilinear", align_corners=True))
        super().__init__(head, **kwargs)
        if path is not None:
            self.load(path)


class LSegRNNetZS(BaseModel):                         # lseg_net_zs.py:243-363 (clip_resnet101 backbone)
    def __init__(self, *a, **k):
        raise NotImplementedError("the CLIP-ResNet101 zero-shot backbone is outside the HIP engine's scope (ViT backbones only)")

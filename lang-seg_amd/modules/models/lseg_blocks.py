"""Parameter holders for the DPT "scratch" head: same names/keys as the reference's
modules/models/lseg_blocks.py:60-110 (_make_scratch), :113-147 (Interpolate), :222-288
(ResidualConvUnit_custom), :293-358 (FeatureFusionBlock_custom).  No PyTorch compute here."""
import torch.nn as nn

from .lseg_vit import _NoForward, make_backbone, make_clip_text


def _make_encoder(cfg):
    """(clip_pretrained, pretrained, scratch) like lseg_blocks.py:12-57; unknown backbones are
    rejected earlier by lseg_hip.config.get_config the way the reference does (print + assert)."""
    clip_pretrained = make_clip_text(cfg.text)
    pretrained = make_backbone(cfg)
    scratch = _make_scratch(list(cfg.reassemble), cfg.features)
    return clip_pretrained, pretrained, scratch


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    scratch = nn.Module()
    o = [out_shape, out_shape * 2, out_shape * 4, out_shape * 8] if expand else [out_shape] * 4
    for i in range(4):
        setattr(scratch, f"layer{i + 1}_rn",
                nn.Conv2d(in_shape[i], o[i], kernel_size=3, stride=1, padding=1, bias=False, groups=groups))
    return scratch


class Interpolate(_NoForward):
    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners


class ResidualConvUnit_custom(_NoForward):
    def __init__(self, features, activation, bn):
        super().__init__()
        self.bn = bn
        self.groups = 1
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not bn, groups=1)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not bn, groups=1)
        if bn:
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)
        self.activation = activation


class FeatureFusionBlock_custom(_NoForward):
    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True):
        super().__init__()
        self.deconv, self.align_corners, self.groups, self.expand = deconv, align_corners, 1, expand
        out_features = features // 2 if expand else features
        self.out_conv = nn.Conv2d(features, out_features, kernel_size=1, stride=1, padding=0, bias=True, groups=1)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)

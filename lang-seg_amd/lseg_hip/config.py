"""Static shape description of one LSeg network variant.

Mirrors the constants the reference hard-codes in
  modules/models/lseg_net.py:119-123,142-146   (hooks, out_c)
  modules/models/lseg_vit.py:221-272,275-535   (timm model names, reassemble stacks)
  modules/models/lseg_blocks.py:24-52          (scratch in-channels)
plus the [3P] timm-0.4.12 / CLIP@04f4dc2 model dimensions.

`tiny16` / `tiny32` are not reference variants: they are reduced-size twins
(same topology, small dims) used by the test-suite so the oracle finishes in
milliseconds.
"""
from dataclasses import dataclass, field, asdict
from typing import List, Tuple


@dataclass(frozen=True)
class TextConfig:
    vocab: int = 49408
    ctx: int = 77
    width: int = 512
    heads: int = 8
    layers: int = 12
    embed_dim: int = 512  # == out_c of the image head


@dataclass(frozen=True)
class LSegConfig:
    name: str
    patch: int
    dim: int
    depth: int
    heads: int
    hooks: Tuple[int, int, int, int]
    pos_grid: int                      # pretrained pos-embed grid (384 / patch)
    reassemble: Tuple[int, int, int, int]  # channels after the 1x1 conv
    # per-level resample op after the 1x1: ("convT", k) | ("id", 0) | ("conv_s2", 3)
    resample: Tuple[Tuple[str, int], ...]
    features: int = 256
    out_c: int = 512
    text: TextConfig = field(default_factory=TextConfig)
    arch_option: int = 0
    block_depth: int = 0
    activation: str = "lrelu"

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    def tokens(self, h: int, w: int) -> int:
        return (h // self.patch) * (w // self.patch) + 1

    def to_dict(self):
        return asdict(self)


_CONFIGS = {
    # lseg_vit.py:221-237 + 408-535 ; lseg_net.py:120
    "clip_vitl16_384": LSegConfig(
        name="clip_vitl16_384", patch=16, dim=1024, depth=24, heads=16,
        hooks=(5, 11, 17, 23), pos_grid=24,
        reassemble=(256, 512, 1024, 1024),
        resample=(("convT", 4), ("convT", 2), ("id", 0), ("conv_s2", 3)),
    ),
    # lseg_vit.py:240-257 (ViT-L/16 image tower + the text tower of CLIP RN50x16: width 768, 12 heads) ; lseg_net.py:121,142-143 (out_c 768)
    "clipRN50x16_vitl16_384": LSegConfig(
        name="clipRN50x16_vitl16_384", patch=16, dim=1024, depth=24, heads=16,
        hooks=(5, 11, 17, 23), pos_grid=24,
        reassemble=(256, 512, 1024, 1024),
        resample=(("convT", 4), ("convT", 2), ("id", 0), ("conv_s2", 3)),
        out_c=768, text=TextConfig(width=768, heads=12, layers=12, embed_dim=768),
    ),
    # lseg_vit.py:259-272 + 275-405 ; lseg_net.py:122
    "clip_vitb32_384": LSegConfig(
        name="clip_vitb32_384", patch=32, dim=768, depth=12, heads=12,
        hooks=(2, 5, 8, 11), pos_grid=12,
        reassemble=(96, 192, 384, 768),
        resample=(("convT", 8), ("convT", 4), ("convT", 2), ("id", 0)),
    ),
    # reduced twins for tests (NOT reference variants)
    "tiny16": LSegConfig(
        name="tiny16", patch=16, dim=128, depth=4, heads=2,
        hooks=(0, 1, 2, 3), pos_grid=4,
        reassemble=(64, 64, 128, 128),
        resample=(("convT", 4), ("convT", 2), ("id", 0), ("conv_s2", 3)),
        features=64, out_c=128,
        text=TextConfig(vocab=512, ctx=77, width=128, heads=2, layers=2, embed_dim=128),
    ),
    "tiny32": LSegConfig(
        name="tiny32", patch=32, dim=128, depth=4, heads=2,
        hooks=(0, 1, 2, 3), pos_grid=3,
        reassemble=(64, 64, 128, 128),
        resample=(("convT", 8), ("convT", 4), ("convT", 2), ("id", 0)),
        features=64, out_c=128,
        text=TextConfig(vocab=512, ctx=77, width=128, heads=2, layers=2, embed_dim=128),
    ),
}


def get_config(backbone: str, features: int = None, arch_option: int = 0,
               block_depth: int = 0, activation: str = "lrelu") -> LSegConfig:
    """Look a backbone up by the reference's `--backbone` string.

    Unknown names fail the way the reference does (lseg_blocks.py:53-55:
    prints and `assert False`)."""
    if backbone not in _CONFIGS:
        print(f"Backbone '{backbone}' not implemented")
        assert False
    base = _CONFIGS[backbone]
    kw = base.to_dict()
    kw["text"] = base.text
    kw["hooks"] = tuple(kw["hooks"])
    kw["reassemble"] = tuple(kw["reassemble"])
    kw["resample"] = tuple(tuple(r) for r in kw["resample"])
    if features is not None:
        kw["features"] = features
    kw["arch_option"] = arch_option
    kw["block_depth"] = block_depth
    kw["activation"] = activation
    return LSegConfig(**kw)


def available_backbones() -> List[str]:
    return list(_CONFIGS)

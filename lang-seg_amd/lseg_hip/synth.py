"""Deterministic synthetic parameters + inputs for an LSeg variant.

There is no network in the build/bench environment, so neither the LSeg
checkpoints nor the CLIP BPE vocabulary are available.
Bench and parity tests therefore run on seeded random-init weights laid out
under exactly the reference's state-dict keys (key names come
from the module tree built at modules/models/lseg_net.py:126-146,
lseg_vit.py:408-535, lseg_blocks.py:60-110,222-358 and [3P] timm/CLIP).

Everything is generated on CPU with an explicit torch.Generator so the same
tensors come out on every machine with the same torch version.
"""
import math
import zlib
from typing import Dict, List

import torch

from .config import LSegConfig

# CLIP special tokens ([3P] clip/simple_tokenizer.py): <|startoftext|>, <|endoftext|>
SOT_TOKEN = 49406
EOT_TOKEN = 49407


def _randn(g, shape, std):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def synthetic_state_dict(cfg: LSegConfig, seed: int = 0, clip_fp16: bool = True,
                         with_prefix: str = "") -> Dict[str, torch.Tensor]:
    """Seeded random weights under the reference's `net.*` key layout.

    Scales are chosen so activations stay O(1) through all blocks and the
    attention softmax is *not* flat (qkv std is boosted), BN running stats are
    non-trivial and LN/BN affine parameters are 1 +- 0.1.
    CLIP Linear/MHA/text_projection weights are fp16 like `clip.load(device=
    'cuda')` leaves them ([3P] clip/model.py convert_weights); LayerNorm and the
    embeddings stay fp32.
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    D, P = cfg.dim, cfg.patch

    def put(k, t):
        sd[with_prefix + k] = t.contiguous()

    # ---- timm VisionTransformer (pretrained.model.*) -------------------------------
    vm = "pretrained.model."
    put(vm + "cls_token", _randn(g, (1, 1, D), 0.02))
    put(vm + "pos_embed", _randn(g, (1, 1 + cfg.pos_grid ** 2, D), 0.1))
    put(vm + "patch_embed.proj.weight", _randn(g, (D, 3, P, P), 1.0 / math.sqrt(3 * P * P)))
    put(vm + "patch_embed.proj.bias", _randn(g, (D,), 0.02))
    for i in range(cfg.depth):
        b = f"{vm}blocks.{i}."
        for n in ("norm1", "norm2"):
            put(b + n + ".weight", 1.0 + _randn(g, (D,), 0.1))
            put(b + n + ".bias", _randn(g, (D,), 0.05))
        put(b + "attn.qkv.weight", _randn(g, (3 * D, D), 2.0 / math.sqrt(D)))
        put(b + "attn.qkv.bias", _randn(g, (3 * D,), 0.02))
        put(b + "attn.proj.weight", _randn(g, (D, D), 0.5 / math.sqrt(D)))
        put(b + "attn.proj.bias", _randn(g, (D,), 0.02))
        put(b + "mlp.fc1.weight", _randn(g, (4 * D, D), 1.0 / math.sqrt(D)))
        put(b + "mlp.fc1.bias", _randn(g, (4 * D,), 0.02))
        put(b + "mlp.fc2.weight", _randn(g, (D, 4 * D), 0.5 / math.sqrt(4 * D)))
        put(b + "mlp.fc2.bias", _randn(g, (D,), 0.02))
    put(vm + "norm.weight", 1.0 + _randn(g, (D,), 0.1))   # dead in forward (lseg_vit.py:108)
    put(vm + "norm.bias", _randn(g, (D,), 0.05))
    put(vm + "head.weight", _randn(g, (1000, D), 0.02))     # dead, must still load/save
    put(vm + "head.bias", torch.zeros(1000))

    # ---- readout + reassemble (pretrained.act_postprocessK.*) ----------------------
    for lvl in range(4):
        a = f"pretrained.act_postprocess{lvl + 1}."
        C = cfg.reassemble[lvl]
        put(a + "0.project.0.weight", _randn(g, (D, 2 * D), 1.0 / math.sqrt(2 * D)))
        put(a + "0.project.0.bias", _randn(g, (D,), 0.02))
        put(a + "3.weight", _randn(g, (C, D, 1, 1), 1.0 / math.sqrt(D)))
        put(a + "3.bias", _randn(g, (C,), 0.02))
        kind, k = cfg.resample[lvl]
        if kind == "convT":          # ConvTranspose2d weight is [in, out, kh, kw]
            put(a + "4.weight", _randn(g, (C, C, k, k), 1.0 / math.sqrt(C)))
            put(a + "4.bias", _randn(g, (C,), 0.02))
        elif kind == "conv_s2":
            put(a + "4.weight", _randn(g, (C, C, 3, 3), 1.0 / math.sqrt(9 * C)))
            put(a + "4.bias", _randn(g, (C,), 0.02))

    # ---- scratch (DPT head) ---------------------------------------------------------
    F_ = cfg.features
    for lvl in range(4):
        C = cfg.reassemble[lvl]
        put(f"scratch.layer{lvl + 1}_rn.weight", _randn(g, (F_, C, 3, 3), 1.0 / math.sqrt(9 * C)))
    for r in range(1, 5):
        p = f"scratch.refinenet{r}."
        put(p + "out_conv.weight", _randn(g, (F_, F_, 1, 1), 1.0 / math.sqrt(F_)))
        put(p + "out_conv.bias", _randn(g, (F_,), 0.02))
        for u in (1, 2):
            q = f"{p}resConfUnit{u}."
            for c in (1, 2):
                put(q + f"conv{c}.weight", _randn(g, (F_, F_, 3, 3), 1.0 / math.sqrt(9 * F_)))
                put(q + f"bn{c}.weight", 1.0 + _randn(g, (F_,), 0.1))
                put(q + f"bn{c}.bias", _randn(g, (F_,), 0.05))
                put(q + f"bn{c}.running_mean", _randn(g, (F_,), 0.1))
                put(q + f"bn{c}.running_var",
                    0.5 + torch.rand((F_,), generator=g, dtype=torch.float32))
                put(q + f"bn{c}.num_batches_tracked", torch.tensor(100, dtype=torch.int64))
    put("scratch.head1.weight", _randn(g, (cfg.out_c, F_, 1, 1), 1.0 / math.sqrt(F_)))
    put("scratch.head1.bias", _randn(g, (cfg.out_c,), 0.02))
    if cfg.arch_option in (1, 2):
        put("scratch.head_block.depthwise.depthwise.weight", _randn(g, (1, 1, 3, 3), 0.3))
        put("scratch.head_block.depthwise.depthwise.bias", _randn(g, (1,), 0.05))

    # ---- CLIP text tower (clip_pretrained.*); visual tower omitted (unused) --------
    t = cfg.text
    cp = "clip_pretrained."
    W = t.width
    lin_dt = torch.float16 if clip_fp16 else torch.float32
    put(cp + "token_embedding.weight", _randn(g, (t.vocab, W), 0.02))
    put(cp + "positional_embedding", _randn(g, (t.ctx, W), 0.01))
    for i in range(t.layers):
        b = f"{cp}transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            put(b + n + ".weight", 1.0 + _randn(g, (W,), 0.1))
            put(b + n + ".bias", _randn(g, (W,), 0.05))
        put(b + "attn.in_proj_weight", _randn(g, (3 * W, W), 2.0 / math.sqrt(W)).to(lin_dt))
        put(b + "attn.in_proj_bias", _randn(g, (3 * W,), 0.02).to(lin_dt))
        put(b + "attn.out_proj.weight", _randn(g, (W, W), 0.5 / math.sqrt(W)).to(lin_dt))
        put(b + "attn.out_proj.bias", _randn(g, (W,), 0.02).to(lin_dt))
        put(b + "mlp.c_fc.weight", _randn(g, (4 * W, W), 1.0 / math.sqrt(W)).to(lin_dt))
        put(b + "mlp.c_fc.bias", _randn(g, (4 * W,), 0.02).to(lin_dt))
        put(b + "mlp.c_proj.weight", _randn(g, (W, 4 * W), 0.5 / math.sqrt(4 * W)).to(lin_dt))
        put(b + "mlp.c_proj.bias", _randn(g, (W,), 0.02).to(lin_dt))
    put(cp + "ln_final.weight", 1.0 + _randn(g, (W,), 0.1))
    put(cp + "ln_final.bias", _randn(g, (W,), 0.05))
    put(cp + "text_projection", _randn(g, (W, t.embed_dim), 1.0 / math.sqrt(W)).to(lin_dt))
    put(cp + "logit_scale", torch.tensor(math.log(1 / 0.07), dtype=torch.float32))
    return sd


def synthetic_tokens(labels: List[str], vocab: int = 49408, ctx: int = 77) -> torch.Tensor:
    """Stand-in for `clip.tokenize` when the BPE vocabulary is not on disk.

    Produces int64 [K, ctx] rows `[SOT, ids..., EOT, 0...]` with one id per
    whitespace-separated word piece (hash of the piece), so that
      * the EOT id is the row maximum (encode_text pools at `text.argmax(-1)`,
        [3P] clip/model.py encode_text),
      * different labels give different rows, multi-word labels give longer rows.
    For reduced vocabularies (tiny test configs) SOT/EOT are vocab-2 / vocab-1.
    """
    sot, eot = (SOT_TOKEN, EOT_TOKEN) if vocab >= 49408 else (vocab - 2, vocab - 1)
    out = torch.zeros((len(labels), ctx), dtype=torch.int64)
    for i, lab in enumerate(labels):
        pieces = str(lab).lower().replace("_", " ").split() or [""]
        ids = [1 + (zlib.crc32(p.encode("utf-8")) % (sot - 1)) for p in pieces][: ctx - 2]
        row = [sot] + ids + [eot]
        out[i, : len(row)] = torch.tensor(row, dtype=torch.int64)
    return out


def synthetic_images(batch: int, h: int = 480, w: int = 480, seed: int = 0) -> torch.Tensor:
    """U(0,1) RGB normalised with mean=std=0.5, like ToTensor+Normalize at
    modules/lseg_module.py:37-50."""
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.rand((batch, 3, h, w), generator=g, dtype=torch.float32)
    return (x - 0.5) / 0.5


def read_labels(path: str, skip_header: bool = True) -> List[str]:
    """Same parse as LSegModule.get_labels (modules/lseg_module.py:97-109)."""
    labels = []
    with open(path, "r") as f:
        for line in f.readlines():
            labels.append(line.strip().split(",")[-1].split(";")[0])
    return labels[1:] if skip_header else labels


def outlier_state_dict(cfg: LSegConfig, seed: int, level: float = 1e2) -> Dict[str, torch.Tensor]:
    """A synthetic state dict with the statistics real ViT / DPT checkpoints are known for and N(0, 0.02)-style random weights are
    not: a few residual-stream OUTLIER CHANNELS carried by mlp.fc2 / attn.proj rows (x sqrt(level)), LayerNorm gains of 10 on them,
    mlp.fc1 columns reading them amplified (x sqrt(level)), BatchNorm layers with small running variances (x10 effective scale on every
    17th channel).  level 1e2: residual outliers of a few hundred -- what fp16 MFMA operands (range 65504) must survive and what costs
    bf16 operands (8-bit significand) accuracy on the ordinary channels next to them.  Used by the range / fallback test and by the
    reference-run fixture `ref_full_*_outlier` (oracle/make_ref_golden.py) that pins the 16-bit inference default on such weights."""
    sd = synthetic_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 99)
    ch = torch.randperm(cfg.dim, generator=g)[:3]
    for k in sd:
        if k.endswith(("mlp.fc2.weight", "attn.proj.weight")) and k.startswith("pretrained.model.blocks."):
            sd[k][ch] *= level ** 0.5
        if k.endswith(("norm1.weight", "norm2.weight")) and k.startswith("pretrained.model.blocks."):
            sd[k][ch] = 10.0
        if k.endswith("mlp.fc1.weight") and k.startswith("pretrained.model.blocks."):
            sd[k][:, ch] *= level ** 0.5
        if k.endswith("running_var") and ".bn1." in k:
            sd[k][::17] *= 0.01
    return sd


def fixture_state_dict(cfg: LSegConfig, seed: int, fixture: Dict) -> Dict[str, torch.Tensor]:
    """The weights a tests/golden fixture was made with: `outlier_level` in the fixture selects outlier_state_dict."""
    lvl = fixture.get("outlier_level") if hasattr(fixture, "get") else None
    return outlier_state_dict(cfg, seed, float(lvl)) if lvl else synthetic_state_dict(cfg, seed=seed)

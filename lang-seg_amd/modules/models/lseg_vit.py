"""Parameter holders for the ViT backbone + readout/reassemble stacks.

Same module tree and state-dict keys as the reference builds at
modules/models/lseg_vit.py:221-272 (timm.create_model + clip.load) and :275-535
(_make_vit_b32_backbone / _make_vit_b16_backbone), so checkpoints load unchanged.  These
modules only OWN parameters: all arithmetic runs in the HIP engine (lseg_hip), there is no
PyTorch compute path here.
"""
import torch
import torch.nn as nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; the LSeg forward runs in the HIP engine")


class PatchEmbed(_NoForward):
    def __init__(self, img_size, patch, dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch, patch)
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class Attention(_NoForward):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(_NoForward):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * dim, dim)


class Block(_NoForward):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim)


class VisionTransformer(_NoForward):
    """[3P] timm 0.4.12 VisionTransformer attribute surface the reference touches
    (SURVEY.md App. A.1): patch_embed.proj/.img_size, cls_token, pos_embed, pos_drop, blocks,
    norm, head, plus the injected start_index / patch_size."""

    def __init__(self, patch, dim, depth, heads, img_size=384, num_classes=1000):
        super().__init__()
        self.patch_embed = PatchEmbed(img_size, patch, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + (img_size // patch) ** 2, dim))
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.ModuleList([Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        self.start_index = 1
        self.patch_size = [patch, patch]
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)


class ProjectReadout(_NoForward):          # lseg_vit.py:79-90
    def __init__(self, in_features, start_index=1):
        super().__init__()
        self.start_index = start_index
        self.project = nn.Sequential(nn.Linear(2 * in_features, in_features), nn.GELU())


class Transpose(_NoForward):               # lseg_vit.py:93-101
    def __init__(self, dim0, dim1):
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1


def make_backbone(cfg, size=(384, 384)) -> nn.Module:
    """`pretrained` of the reference: .model + act_postprocess1..4 (indices 0 readout,
    1 Transpose, 2 Unflatten, 3 1x1 Conv2d, 4 ConvTranspose2d / strided Conv2d)."""
    pretrained = nn.Module()
    pretrained.model = VisionTransformer(cfg.patch, cfg.dim, cfg.depth, cfg.heads, img_size=cfg.pos_grid * cfg.patch)
    for lvl in range(4):
        C = cfg.reassemble[lvl]
        mods = [ProjectReadout(cfg.dim, 1), Transpose(1, 2),
                nn.Unflatten(2, torch.Size([size[0] // cfg.patch, size[1] // cfg.patch])),
                nn.Conv2d(cfg.dim, C, kernel_size=1, stride=1, padding=0)]
        kind, k = cfg.resample[lvl]
        if kind == "convT":
            mods.append(nn.ConvTranspose2d(C, C, kernel_size=k, stride=k, padding=0, bias=True))
        elif kind == "conv_s2":
            mods.append(nn.Conv2d(C, C, kernel_size=3, stride=2, padding=1))
        setattr(pretrained, f"act_postprocess{lvl + 1}", nn.Sequential(*mods))
    return pretrained


class _AnyKeys(nn.Module):
    """Accepts (and round-trips) arbitrary state-dict entries: the CLIP *visual* tower lives in
    every LSeg checkpoint under clip_pretrained.visual.* but is never used by the forward."""

    def __init__(self):
        super().__init__()
        self._extra = {}

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing, unexpected, errors):
        for k in [k for k in state_dict if k.startswith(prefix)]:
            self._extra[k[len(prefix):]] = state_dict[k]
            # mark as consumed so strict loading does not flag it
        for k in list(unexpected):
            if k.startswith(prefix):
                unexpected.remove(k)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if destination is None:
            destination = {}
        for k, v in self._extra.items():
            destination[prefix + k] = v
        return destination


class _TextMlp(_NoForward):
    def __init__(self, w):
        super().__init__()
        self.c_fc = nn.Linear(w, 4 * w)
        self.gelu = nn.Identity()     # QuickGELU has no parameters
        self.c_proj = nn.Linear(4 * w, w)


class _ResBlock(_NoForward):
    def __init__(self, w, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(w, heads)
        self.ln_1 = nn.LayerNorm(w)
        self.mlp = _TextMlp(w)
        self.ln_2 = nn.LayerNorm(w)


class _Transformer(_NoForward):
    def __init__(self, w, layers, heads):
        super().__init__()
        self.width, self.layers = w, layers
        self.resblocks = nn.Sequential(*[_ResBlock(w, heads) for _ in range(layers)])


class CLIPTextHolder(_NoForward):
    """[3P] CLIP attribute surface of the text tower: token_embedding,
    positional_embedding, transformer.resblocks.N.{ln_1,attn,ln_2,mlp.c_fc,mlp.c_proj}, ln_final,
    text_projection, logit_scale; Linear/MHA/text_projection in fp16 like clip.load(device='cuda')."""

    def __init__(self, tcfg, fp16=True):
        super().__init__()
        self.context_length = tcfg.ctx
        self.vocab_size = tcfg.vocab
        self.token_embedding = nn.Embedding(tcfg.vocab, tcfg.width)
        self.positional_embedding = nn.Parameter(torch.empty(tcfg.ctx, tcfg.width))
        self.transformer = _Transformer(tcfg.width, tcfg.layers, tcfg.heads)
        self.ln_final = nn.LayerNorm(tcfg.width)
        self.text_projection = nn.Parameter(torch.empty(tcfg.width, tcfg.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600)
        self.visual = _AnyKeys()
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=tcfg.width ** -0.5)
        if fp16:
            for blk in self.transformer.resblocks:
                for m in (blk.attn, blk.attn.out_proj, blk.mlp.c_fc, blk.mlp.c_proj):
                    for p in m.parameters(recurse=False):
                        p.data = p.data.half()
            self.text_projection.data = self.text_projection.data.half()

    def encode_text(self, text):
        raise RuntimeError("encode_text runs inside the HIP engine (lseg_encode_text)")


def make_clip_text(tcfg) -> nn.Module:
    """clip.load("ViT-B/32", device='cuda', jit=False)[0] when the package (and its weights) are
    available -- exactly lseg_vit.py:224 -- else an identically-keyed holder."""
    if tcfg.vocab >= 49408:
        try:
            import clip
            model, _ = clip.load("ViT-B/32", device="cuda" if torch.cuda.is_available() else "cpu", jit=False)
            return model
        except Exception:
            pass
    return CLIPTextHolder(tcfg)

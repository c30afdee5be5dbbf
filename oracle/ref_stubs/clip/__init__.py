"""Stand-in for openai/CLIP@04f4dc2 (requirements.txt:15 of the reference): only what the reference touches --
clip.load("ViT-B/32", device='cuda', jit=False) -> (model, preprocess) and clip.tokenize.  The text tower keeps
CLIP's parameter names and is built on torch's own nn.MultiheadAttention; weights are cast like
clip.load on cuda + convert_weights: Linear / MHA / text_projection fp16, LayerNorm / embeddings fp32, LayerNorm
computed in fp32 and cast back.  The visual tower (never run by LSeg) is a parameter-less placeholder.
TEST INFRASTRUCTURE."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn


class LayerNorm(nn.LayerNorm):
    def forward(self, x):                                  # clip/model.py: fp32 LayerNorm, cast back
        return super().forward(x.type(torch.float32)).type(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask

    def attention(self, x):
        mask = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        return self.attn(x, x, x, need_weights=False, attn_mask=mask)[0]

    def forward(self, x):
        x = x + self.attention(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Visual(nn.Module):                                   # CLIP.dtype reads visual.conv1.weight.dtype
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 1, bias=False)


class CLIP(nn.Module):
    def __init__(self, embed_dim=512, context_length=77, vocab_size=49408, width=512, heads=8, layers=12):
        super().__init__()
        self.context_length = context_length
        self.visual = _Visual()
        self.transformer = Transformer(width, layers, heads, self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.ln_final = LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim).normal_(std=width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def build_attention_mask(self):
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_text(self, text):
        x = self.token_embedding(text).type(self.dtype)
        x = x + self.positional_embedding.type(self.dtype)
        x = x.permute(1, 0, 2)
        x = self.transformer(x)
        x = x.permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


def convert_weights(model):
    def _h(l):
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            l.weight.data = l.weight.data.half()
            if l.bias is not None:
                l.bias.data = l.bias.data.half()
        if isinstance(l, nn.MultiheadAttention):
            for attr in ["in_proj_weight", "in_proj_bias", "bias_k", "bias_v"]:
                t = getattr(l, attr)
                if t is not None:
                    t.data = t.data.half()
            l.out_proj.weight.data = l.out_proj.weight.data.half()
            l.out_proj.bias.data = l.out_proj.bias.data.half()
        if hasattr(l, "text_projection"):
            l.text_projection.data = l.text_projection.data.half()
    model.apply(_h)


_TEXT_CFG = {"ViT-B/32": dict(embed_dim=512, width=512, heads=8, layers=12),
             # CLIP RN50x16's text tower ([3P] clip/model.py build_model: transformer_width 768, heads = width // 64, 12 layers,
             # embed_dim 768) -- what lseg_vit.py:243 loads for backbone "clipRN50x16_vitl16_384"
             "RN50x16": dict(embed_dim=768, width=768, heads=12, layers=12)}


def load(name, device="cpu", jit=False):
    if name not in _TEXT_CFG:
        raise RuntimeError(f"clip stand-in: no text tower for {name}")
    model = CLIP(**_TEXT_CFG[name])
    convert_weights(model)                                 # what clip.load does for device='cuda'
    return model.eval(), None


def tokenize(texts, context_length=77, truncate=False):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(root, "lang-seg_amd"))
    from lseg_hip.synth import synthetic_tokens
    if isinstance(texts, str):
        texts = [texts]
    return synthetic_tokens(list(texts), 49408, context_length)

"""CPU ORACLE for the LSeg forward hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (lang-seg_amd/) never does.

PARITY PINNING.  The reference (isl-org/lang-seg) ships no tests, golden vectors or fixtures for this path, so
the pins are outputs of the reference ITSELF run in the build container: oracle/make_ref_golden.py executes the
reference's own modules/models/lseg_net.py and lseg_net_zs.py (LSegNet / LSegNetZS construction and forward, with
everything they pull in from lseg_vit.py and lseg_blocks.py) on CPU with seeded synthetic weights and commits the
results as tests/golden/ref_*.pt; tests/test_oracle_ref_golden.py holds this file to them (fp32 stages 2e-3 relative,
fp16-valued tensors a few fp16 ulps, identical arg-max wherever the reference is decisive).  That covers every function
on the path that lives in the reference repository.  The two THIRD-PARTY pieces it imports are absent here (timm==0.4.12,
openai/CLIP@04f4dc2) and are stood in by oracle/ref_stubs/ (timm's VisionTransformer module tree; CLIP's
text tower on torch's own nn.MultiheadAttention with fp16 weights): for those two pieces parity remains "unpinned by the
reference" and rests on the published algorithms restated in SURVEY App. A, cross-checked against HF `transformers`
(CLIP text tower, ViT block, DPT neck: tests/test_oracle_hf.py) and against torch's MHA through the stand-in.

What is restated (all paths relative to /root/reference):
  modules/models/lseg_net.py:160-205     LSeg.forward
  modules/models/lseg_vit.py:104-201     forward_vit / forward_flex / _resize_pos_embed
  modules/models/lseg_vit.py:79-90       ProjectReadout
  modules/models/lseg_vit.py:275-535     act_postprocess stacks
  modules/models/lseg_blocks.py:60-110   scratch.layerN_rn
  modules/models/lseg_blocks.py:222-358  ResidualConvUnit_custom / FeatureFusionBlock_custom
  modules/models/lseg_net.py:29-79       head blocks (arch_option 1/2)
  modules/models/lseg_net_zs.py:177-214  LSegNetZS.forward (labels_per_image)
  modules/lsegmentation_module.py:66-81  training_step (loss + autograd gradients; pinned by tests/golden/ref_train_*.pt)
  [3P] timm==0.4.12 vision_transformer.py  Block / Attention / Mlp   
  [3P] openai/CLIP@04f4dc2 clip/model.py   encode_text               

Dtype choreography follows the reference exactly: fp32 image tower; CLIP text
tower with fp16 weights/activations and fp32 LayerNorm (emulated here as fp32
math with an explicit round-to-fp16 wherever the CUDA path materialises an fp16
tensor); the correlation is the left-associative
`logit_scale * image_features.half() @ text_features.t()` (lseg_net.py:194).
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def r16(t: Tensor) -> Tensor:
    """Round to fp16 and come back to fp32 (emulates an fp16 tensor on CPU)."""
    return t.to(torch.float16).to(torch.float32)


def _ident(t: Tensor) -> Tensor:
    return t


# --------------------------------------------------------------------------------------
# [3P] timm 0.4.12 VisionTransformer pieces (invoked at lseg_vit.py:179,196-197)
# --------------------------------------------------------------------------------------
def vit_block(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int) -> Tensor:
    """timm Block: x += attn(norm1(x)); x += mlp(norm2(x)); LN eps 1e-6, GELU(erf)."""
    B, N, C = x.shape
    hd = C // heads
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x


def resize_pos_embed(posemb: Tensor, gs_h: int, gs_w: int, start_index: int = 1) -> Tensor:
    """lseg_vit.py:149-163 (bilinear, align_corners=False default)."""
    posemb_tok, posemb_grid = posemb[:, :start_index], posemb[0, start_index:]
    gs_old = int(math.sqrt(len(posemb_grid)))
    posemb_grid = posemb_grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    posemb_grid = F.interpolate(posemb_grid, size=(gs_h, gs_w), mode="bilinear")
    posemb_grid = posemb_grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
    return torch.cat([posemb_tok, posemb_grid], dim=1)


def forward_flex(sd: Dict[str, Tensor], x: Tensor, cfg) -> List[Tensor]:
    """lseg_vit.py:166-201; returns the 4 hooked block outputs (lseg_vit.py:12-16,
    421-424).  The final `self.norm` (:199) only feeds `glob`, which forward_vit
    discards (:108), so it is not computed."""
    vm = "pretrained.model."
    b, c, h, w = x.shape
    pos_embed = resize_pos_embed(sd[vm + "pos_embed"], h // cfg.patch, w // cfg.patch)
    x = F.conv2d(x, sd[vm + "patch_embed.proj.weight"], sd[vm + "patch_embed.proj.bias"],
                 stride=cfg.patch).flatten(2).transpose(1, 2)
    cls_tokens = sd[vm + "cls_token"].expand(b, -1, -1)
    x = torch.cat((cls_tokens, x), dim=1)
    x = x + pos_embed
    acts = []
    for i in range(cfg.depth):
        x = vit_block(sd, f"{vm}blocks.{i}.", x, cfg.heads)
        if i in cfg.hooks:
            acts.append(x)
    return acts


# --------------------------------------------------------------------------------------
# readout + reassemble (lseg_vit.py:79-90, 104-146, 275-535)
# --------------------------------------------------------------------------------------
def project_readout(sd, p: str, x: Tensor) -> Tensor:
    """ProjectReadout (lseg_vit.py:86-90): Linear(cat(tok, cls)) -> GELU."""
    readout = x[:, 0].unsqueeze(1).expand_as(x[:, 1:])
    feats = torch.cat((x[:, 1:], readout), -1)
    return F.gelu(F.linear(feats, sd[p + "0.project.0.weight"], sd[p + "0.project.0.bias"]))


def act_postprocess(sd, lvl: int, x: Tensor, gh: int, gw: int, cfg) -> Tensor:
    """act_postprocessK as applied by forward_vit (lseg_vit.py:115-144): [0:2] =
    readout + transpose, dynamic unflatten to (h/patch, w/patch), [3:] = 1x1 conv
    (+ ConvTranspose / strided conv)."""
    p = f"pretrained.act_postprocess{lvl + 1}."
    B = x.shape[0]
    y = project_readout(sd, p, x).transpose(1, 2)             # [B, D, N-1]
    y = y.reshape(B, -1, gh, gw)
    y = F.conv2d(y, sd[p + "3.weight"], sd[p + "3.bias"])
    kind, k = cfg.resample[lvl]
    if kind == "convT":
        y = F.conv_transpose2d(y, sd[p + "4.weight"], sd[p + "4.bias"], stride=k)
    elif kind == "conv_s2":
        y = F.conv2d(y, sd[p + "4.weight"], sd[p + "4.bias"], stride=2, padding=1)
    return y


# --------------------------------------------------------------------------------------
# DPT scratch head (lseg_blocks.py)
# --------------------------------------------------------------------------------------
BN_TRAIN = False      # set by lseg_forward(bn_train=True): nn.BatchNorm2d in train() mode (batch statistics; the training
                      # step of modules/lsegmentation_module.py:66-81 runs the network in train mode)


def residual_conv_unit(sd, p: str, x: Tensor) -> Tensor:
    """ResidualConvUnit_custom.forward (lseg_blocks.py:265-288) with bn=True
    (lseg_net.py:213), activation = nn.ReLU(False) (lseg_net.py:97); eval-mode BN unless BN_TRAIN."""
    def bn(t, q):
        if BN_TRAIN:      # batch statistics over (N,H,W), biased variance for the normalisation; running stats untouched here
            return F.batch_norm(t, None, None, sd[q + ".weight"], sd[q + ".bias"], True, 0.1, 1e-5)
        return F.batch_norm(t, sd[q + ".running_mean"], sd[q + ".running_var"],
                            sd[q + ".weight"], sd[q + ".bias"], False, 0.1, 1e-5)
    out = F.relu(x)
    out = bn(F.conv2d(out, sd[p + "conv1.weight"], None, padding=1), p + "bn1")
    out = F.relu(out)
    out = bn(F.conv2d(out, sd[p + "conv2.weight"], None, padding=1), p + "bn2")
    return out + x


def fusion_block(sd, p: str, *xs: Tensor) -> Tensor:
    """FeatureFusionBlock_custom.forward (lseg_blocks.py:337-358)."""
    output = xs[0]
    if len(xs) == 2:
        output = output + residual_conv_unit(sd, p + "resConfUnit1.", xs[1])
    output = residual_conv_unit(sd, p + "resConfUnit2.", output)
    output = F.interpolate(output, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(output, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


# --------------------------------------------------------------------------------------
# [3P] CLIP encode_text (called at lseg_net.py:183)
# --------------------------------------------------------------------------------------
def encode_text(sd: Dict[str, Tensor], text: Tensor, tcfg, emulate_fp16: bool = True) -> Tensor:
    """CLIP.encode_text as `clip.load(..., device='cuda')` runs it: Linear / MHA /
    text_projection in fp16 (fp32 accumulate, one rounding per op), LayerNorm in
    fp32 then cast back, QuickGELU, causal -inf mask, pooled at text.argmax(-1).
    Returns fp32 holding fp16-representable values when emulate_fp16."""
    rd = r16 if emulate_fp16 else _ident
    cp = "clip_pretrained."
    Kn, L = text.shape
    W, H = tcfg.width, tcfg.heads
    hd = W // H

    def lin(t, wk, bk):
        return rd(F.linear(t, sd[wk].float(), sd[bk].float()))

    def ln(t, q):
        return rd(F.layer_norm(t, (W,), sd[q + ".weight"].float(), sd[q + ".bias"].float(), 1e-5))

    x = rd(sd[cp + "token_embedding.weight"][text].float())
    x = rd(x + rd(sd[cp + "positional_embedding"].float()))
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(tcfg.layers):
        b = f"{cp}transformer.resblocks.{i}."
        h = ln(x, b + "ln_1")
        qkv = lin(h, b + "attn.in_proj_weight", b + "attn.in_proj_bias")
        q, k, v = qkv.chunk(3, dim=-1)
        q = rd(q * (hd ** -0.5))
        q = q.reshape(Kn, L, H, hd).transpose(1, 2)
        k = k.reshape(Kn, L, H, hd).transpose(1, 2)
        v = v.reshape(Kn, L, H, hd).transpose(1, 2)
        a = rd(q @ k.transpose(-2, -1))
        a = a + mask
        a = rd(a.softmax(dim=-1))
        y = rd(a @ v).transpose(1, 2).reshape(Kn, L, W)
        x = rd(x + lin(y, b + "attn.out_proj.weight", b + "attn.out_proj.bias"))
        h = ln(x, b + "ln_2")
        h = lin(h, b + "mlp.c_fc.weight", b + "mlp.c_fc.bias")
        h = rd(h * rd(torch.sigmoid(rd(1.702 * h))))            # QuickGELU in fp16 steps
        x = rd(x + lin(h, b + "mlp.c_proj.weight", b + "mlp.c_proj.bias"))
    x = ln(x, cp + "ln_final")
    pooled = x[torch.arange(Kn), text.argmax(dim=-1)]
    return rd(pooled @ sd[cp + "text_projection"].float())


# --------------------------------------------------------------------------------------
# head blocks (lseg_net.py:29-79), only for arch_option 1/2
# --------------------------------------------------------------------------------------
def _act(name: str, x: Tensor) -> Tensor:
    if name == "relu":
        return F.relu(x)
    if name == "lrelu":
        return F.leaky_relu(x)
    if name == "tanh":
        return torch.tanh(x)
    raise ValueError(name)


def head_block(sd, cfg, x: Tensor, act: bool = True) -> Tensor:
    """bottleneck_block (arch_option 1, lseg_net.py:73-79) / depthwise_block
    (arch_option 2, :54-58): one shared 1-channel 3x3 conv per label plane."""
    w = sd["scratch.head_block.depthwise.depthwise.weight"]
    b = sd["scratch.head_block.depthwise.depthwise.bias"]
    B, C, H, W = x.shape
    y = F.conv2d(x.reshape(-1, 1, H, W), w, b, padding=1).view(-1, C, H, W)
    if cfg.arch_option == 1:
        y = y + x.max(dim=1, keepdim=True)[0]
    if act:
        y = _act(cfg.activation, y)
    return y


# --------------------------------------------------------------------------------------
# LSeg.forward (lseg_net.py:160-205)
# --------------------------------------------------------------------------------------
LOGIT_SCALE = float(torch.tensor(math.log(1 / 0.07), dtype=torch.float32).exp())  # lseg_net.py:141


def correlate(image_features: Tensor, text_features: Tensor, logit_scale: float = LOGIT_SCALE,
              emulate_fp16: bool = True) -> Tensor:
    """lseg_net.py:191-196 on flattened features: image_features [M, C] fp32,
    text_features [K, C] (fp16-valued) -> logits [M, K] fp32 (fp16-valued).
    Left-associative: the scale multiplies the fp16 pixel features first."""
    rd = r16 if emulate_fp16 else _ident
    imf = image_features / image_features.norm(dim=-1, keepdim=True)
    # fp16 tensor / fp16 norm (norm accumulates in fp32, result rounded to fp16)
    tn = rd(text_features.norm(dim=-1, keepdim=True))
    txt = rd(text_features / tn)
    a = rd(logit_scale * rd(imf))
    return rd(a @ txt.t())


def lseg_forward(sd: Dict[str, Tensor], x: Tensor, text: Tensor, cfg,
                 emulate_fp16: bool = True, text_features: Optional[Tensor] = None,
                 return_intermediates: bool = False, labels_per_image: int = 0, bn_train: bool = False):
    """Full LSeg.forward.  sd keys are relative to `net.` (App. B of SURVEY.md);
    x [B,3,H,W] fp32; text int64 [K, ctx].  Returns logits [B,K,H,W] fp32.

    labels_per_image = k > 0 restates the zero-shot network instead (LSegNetZS.forward,
    modules/models/lseg_net_zs.py:177-214): text holds B*k rows, image b is correlated with ITS rows
    [b*k, (b+1)*k) only (:198-208: per-image lists of features, one GEMM per image, torch.cat) and the
    result is [B,k,H,W]; no head blocks on that path."""
    global BN_TRAIN
    BN_TRAIN = bool(bn_train)
    inter = {}
    B, _, H, W = x.shape
    gh, gw = H // cfg.patch, W // cfg.patch
    acts = forward_flex(sd, x, cfg)                                   # lseg_net.py:169
    layers = [act_postprocess(sd, l, acts[l], gh, gw, cfg) for l in range(4)]
    rn = [F.conv2d(layers[l], sd[f"scratch.layer{l + 1}_rn.weight"], None, padding=1)
          for l in range(4)]                                          # :171-174
    path_4 = fusion_block(sd, "scratch.refinenet4.", rn[3])           # :176
    path_3 = fusion_block(sd, "scratch.refinenet3.", path_4, rn[2])   # :177
    path_2 = fusion_block(sd, "scratch.refinenet2.", path_3, rn[1])   # :178
    path_1 = fusion_block(sd, "scratch.refinenet1.", path_2, rn[0])   # :179
    if text_features is None:
        text_features = encode_text(sd, text, cfg.text, emulate_fp16)  # :183
    image_features = F.conv2d(path_1, sd["scratch.head1.weight"], sd["scratch.head1.bias"])  # :185
    imshape = image_features.shape
    imf = image_features.permute(0, 2, 3, 1).reshape(-1, imshape[1])  # :188
    if labels_per_image > 0:                                          # lseg_net_zs.py:198-208
        k, P = labels_per_image, imshape[2] * imshape[3]
        assert text_features.shape[0] == imshape[0] * k
        per_image = [correlate(imf[b * P:(b + 1) * P], text_features[b * k:(b + 1) * k], LOGIT_SCALE, emulate_fp16)
                     for b in range(imshape[0])]
        outs = [l.view(1, imshape[2], imshape[3], -1).permute(0, 3, 1, 2) for l in per_image]
        out = torch.cat(outs, dim=0)
    else:
        logits = correlate(imf, text_features, LOGIT_SCALE, emulate_fp16)  # :191-194
        out = logits.view(imshape[0], imshape[2], imshape[3], -1).permute(0, 3, 1, 2)  # :196
    lowres = out
    if cfg.arch_option in (1, 2) and labels_per_image == 0:           # :198-201
        for _ in range(cfg.block_depth - 1):
            out = head_block(sd, cfg, out)
        out = head_block(sd, cfg, out, False)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)  # :203
    BN_TRAIN = False
    if return_intermediates:
        inter.update(acts=acts, layers=layers, rn=rn, paths=[path_1, path_2, path_3, path_4],
                     image_features=image_features, text_features=text_features, lowres=lowres)
        return out, inter
    return out


# ---- the step after the path: metrics and loss value ([3P] torch-encoding @331ecdd) --------------
def batch_pix_accuracy(output: Tensor, target: Tensor):
    """encoding/utils/metrics.py batch_pix_accuracy (call sites modules/lsegmentation_module.py:49,59):
    predict = argmax(output, 1) + 1; target = target + 1; labeled = sum(target > 0);
    correct = sum((predict == target) * (target > 0))."""
    predict = torch.max(output, 1)[1].long() + 1
    tgt = target.long() + 1
    labeled = int((tgt > 0).sum())
    correct = int(((predict == tgt) & (tgt > 0)).sum())
    return correct, labeled


def batch_intersection_union(output: Tensor, target: Tensor, nclass: int):
    """encoding/utils/metrics.py batch_intersection_union (call sites :50,60): predict = (argmax + 1) * (target + 1 > 0);
    intersection = predict * (predict == target + 1); np.histogram(., bins=nclass, range=(1, nclass)) of
    intersection / predict / target; union = pred + lab - inter.  Returns int64 tensors [nclass]."""
    import numpy as np
    predict = (torch.max(output, 1)[1].long() + 1).numpy()
    tgt = (target.long() + 1).numpy()
    predict = predict * (tgt > 0).astype(predict.dtype)
    intersection = predict * (predict == tgt)
    area_inter, _ = np.histogram(intersection, bins=nclass, range=(1, nclass))
    area_pred, _ = np.histogram(predict, bins=nclass, range=(1, nclass))
    area_lab, _ = np.histogram(tgt, bins=nclass, range=(1, nclass))
    area_union = area_pred + area_lab - area_inter
    return torch.from_numpy(area_inter.astype("int64")), torch.from_numpy(area_union.astype("int64"))


def cross_entropy_value(output: Tensor, target: Tensor, ignore_index: int = -1) -> float:
    """encoding/nn/loss.py SegmentationLosses with se_loss=False, aux=False (every reference run) ==
    nn.CrossEntropyLoss(weight=None, ignore_index)(output, target): mean over non-ignored pixels of
    -log_softmax(output, 1)[target] (call site modules/lsegmentation_module.py:72)."""
    lsm = torch.log_softmax(output.double(), dim=1)
    valid = target != ignore_index
    picked = lsm.gather(1, target.clamp_min(0).unsqueeze(1)).squeeze(1)
    return float(-(picked[valid]).sum() / valid.sum())


def training_step(sd: Dict[str, Tensor], x: Tensor, target: Tensor, text: Tensor, cfg, ignore_index: int = -1):
    """LSegmentationModule.training_step (modules/lsegmentation_module.py:66-81) as loss + gradients: the network in
    train() mode (BatchNorm batch statistics; no dropout anywhere on the path), `criterion` = SegmentationLosses with
    se_loss=False, aux=False == CrossEntropyLoss(ignore_index) ([3P] encoding/nn/loss.py), autograd for the backward.
    Like the reference, the fp16 CLIP text tower is part of the graph (its fp16 weights receive fp16 gradients).
    Returns (loss, {state-dict key: gradient}) for every floating-point tensor that received one; parameters the forward
    never touches (pretrained.model.norm/head, refinenet4.resConfUnit1, clip logit_scale / visual) get none, which is
    why the reference needs DDP(find_unused_parameters=True).  Oracle for the backward kernels."""
    bn_stats = ("running_mean", "running_var", "num_batches_tracked")
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(bn_stats)}
    full = dict(sd)
    full.update(leaves)
    out = lseg_forward(full, x, text, cfg, bn_train=True)
    loss = F.cross_entropy(out, target, ignore_index=ignore_index)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items() if v.grad is not None}

#!/usr/bin/env python
"""Bisect a training-step mismatch: (1) train-mode forward logits per image vs the oracle, (2) lseg_backward(dlogits) with the oracle's
own d(logits) of the CE loss (skips the fused loss kernels), (3) lseg_backward(target) (fused).  usage: probe2 H W B K"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_images, synthetic_tokens, read_labels
from oracle.lseg_oracle import lseg_forward
H, W, B, K = (int(v) for v in sys.argv[1:5])
torch.set_num_threads(min(64, os.cpu_count()))
cfg = get_config("clip_vitl16_384"); sd = synthetic_state_dict(cfg, seed=3)
labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[:K]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
x = synthetic_images(B, H, W, seed=3)
g = torch.Generator().manual_seed(5); t = torch.randint(0, K, (B, H, W), generator=g); t[torch.rand((B, H, W), generator=g) < 0.2] = -1
bn_stats = ("running_mean", "running_var", "num_batches_tracked")
leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(bn_stats) and not k.startswith("clip_pretrained.")}
full = dict(sd); full.update(leaves)
out = lseg_forward(full, x, tok, cfg, bn_train=True)
out.retain_grad()
loss = F.cross_entropy(out, t, ignore_index=-1)
loss.backward()
ref = {k: v.grad for k, v in leaves.items() if v.grad is not None}
dl = out.grad.detach().clone()
sdd = {k: v.cuda() for k, v in sd.items()}
eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K); eng.load_state_dict(sdd); eng.set_tokens(tok); eng.enable_training(sdd)
lo = eng.forward(x.cuda()).cpu(); torch.cuda.synchronize()
for b in range(B):
    print(f"image {b}: train-mode logits max|d| {float((lo[b] - out[b].detach()).abs().max()):.4f}  (range {float(out[b].abs().max()):.2f})")
names = ("scratch.head1.weight", "scratch.head1.bias", "scratch.refinenet1.out_conv.weight", "scratch.refinenet1.resConfUnit2.conv2.weight",
         "scratch.layer4_rn.weight", "pretrained.model.blocks.23.mlp.fc2.weight", "pretrained.model.blocks.2.mlp.fc1.weight")
def rep(tag):
    for n in names:
        a, b_ = eng.grads[n].float().cpu(), ref[n].float()
        print(f"   [{tag}] {n:52s} norm ratio {float(a.norm() / b_.norm()):.3f} rel err {float((a - b_).norm() / b_.norm()):.3f}")
eng.backward(dlogits=dl.cuda()); torch.cuda.synchronize(); rep("dlogits path")
eng.forward(x.cuda(), want_logits=False); l2 = eng.backward(target=t.cuda()); torch.cuda.synchronize(); rep("fused CE path")
print("loss", float(l2), float(loss))

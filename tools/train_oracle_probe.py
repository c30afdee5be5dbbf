#!/usr/bin/env python
"""Engine training step vs oracle.training_step (CPU, fp32 autograd) at a given shape: per-bucket gradient-norm ratios + a few named
tensors (tools; bisecting view).  usage: train_oracle_probe.py backbone H W B K"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_images, synthetic_tokens, read_labels
from oracle.lseg_oracle import training_step
bb, H, W, B, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
SEED = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.set_num_threads(min(64, os.cpu_count()))
cfg = get_config(bb); sd = synthetic_state_dict(cfg, seed=SEED)
labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[:K]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
x = synthetic_images(B, H, W, seed=SEED)
g = torch.Generator().manual_seed(5); t = torch.randint(0, K, (B, H, W), generator=g); t[torch.rand((B, H, W), generator=g) < 0.2] = -1
t0 = time.time(); ref_loss, ref = training_step(sd, x, t, tok, cfg, ignore_index=-1); t_or = time.time() - t0
sdd = {k: v.cuda() for k, v in sd.items()}
eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K); eng.load_state_dict(sdd); eng.set_tokens(tok); eng.enable_training(sdd)
eng.forward(x.cuda(), want_logits=False); loss = eng.backward(target=t.cuda()); torch.cuda.synchronize()
rows = []
for k in eng.grads:
    a, b = eng.grads[k].float().cpu(), ref[k].float()
    rows.append((eng.lib.lseg_grad_bucket(eng._h, k.encode()), k, float(a.norm() / b.norm().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30))))
print(f"{bb} {H}x{W} B={B} K={K} seed={SEED}: loss {float(loss):.5f} vs oracle {float(ref_loss):.5f} (oracle {t_or:.0f} s)")
for name in ("scratch.head1.weight", "scratch.head1.bias", "scratch.refinenet1.out_conv.weight", "scratch.refinenet1.resConfUnit2.bn2.bias",
             "scratch.refinenet1.resConfUnit2.conv2.weight", "pretrained.model.blocks.2.mlp.fc1.weight"):
    r = [q for q in rows if q[1] == name]
    if r: print(f"   {name:55s} norm ratio {r[0][2]:.3f}  rel err {r[0][3]:.3f}")
for b in sorted({r[0] for r in rows}):
    v = [r[2] for r in rows if r[0] == b]; e = [r[3] for r in rows if r[0] == b]
    if b in (0, 1, max(r[0] for r in rows)):
        print(f"   bucket {b:2d}: norm ratio median {statistics.median(v):.3f} [{min(v):.3f}, {max(v):.3f}]; rel err median {statistics.median(e):.3f}")

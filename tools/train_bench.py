#!/usr/bin/env python
"""Time the engine-level training step (BASELINE config 4: ViT-L/16, 480x480, K=150, per-GPU batch 8) on one GPU:
train-mode forward + fused CE + backward + fused SGD (tools; bench.py reports the same figure in its `train_step` field)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels
from lseg_hip.train import DataParallelTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--backbone", default="clip_vitl16_384"); ap.add_argument("--size", type=int, default=480); ap.add_argument("--labels", type=int, default=150)
ap.add_argument("--no-opt", action="store_true")
a = ap.parse_args()
cfg = get_config(a.backbone)
sd = {k: v.cuda() for k, v in synthetic_state_dict(cfg, seed=0).items()}
labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[: a.labels]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
eng = HipEngine(cfg, a.size, a.size, max_batch=a.batch, max_labels=len(labels))
eng.load_state_dict(sd); eng.set_tokens(tok)
tr = DataParallelTrainer(eng, sd)
x = synthetic_images(a.batch, a.size, a.size, seed=0).cuda()
g = torch.Generator().manual_seed(1); t = torch.randint(0, len(labels), (a.batch, a.size, a.size), generator=g); t[torch.rand(t.shape, generator=g) < 0.2] = -1
t = t.cuda()
print("allocated GB", torch.cuda.memory_allocated() / 1e9, "free/total", [v / 1e9 for v in torch.cuda.mem_get_info()])
for _ in range(a.warmup):
    loss = tr.step(x, t, 1e-4, 1e-3, optimize=not a.no_opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for s in range(a.steps):
    if s == a.steps - 1: ev[0].record()
    eng.forward(x, want_logits=False)
    if s == a.steps - 1: ev[1].record()
    loss = eng.backward(target=t)
    tr.exchange.finish()
    if s == a.steps - 1: ev[2].record()
    if not a.no_opt: eng.sgd_step(1e-4, 1e-3, 0.9, 1e-4)
    if s == a.steps - 1: ev[3].record()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"train step {dt * 1e3:.1f} ms  ->  {a.batch / dt:.1f} images/s ; loss {loss.item():.4f}; "
      f"forward {ev[0].elapsed_time(ev[1]):.1f} ms, backward {ev[1].elapsed_time(ev[2]):.1f} ms, sgd+repack {ev[2].elapsed_time(ev[3]):.1f} ms; "
      f"{3 * a.batch * (799.4 + 0.059 * len(labels)) / dt / 1e3:.0f} TF/s on the 3x-forward convention")

#!/usr/bin/env python
"""The loop Lightning runs around LSegmentationModule.training_step (modules/lsegmentation_module.py:66-81) -- training_step ->
loss.backward() -> optimizer.step() -> optimizer.zero_grad() -- on the drop-in LSegModule at BASELINE configs[3] (ViT-L/16,
480x480, K = 150, per-GPU batch 8), next to native_training_step (the same engine calls without autograd / torch.optim) and,
with --slow, the generic path (full logits + torch CE + torch.optim.SGD).  Tools; bench.py's `train_step` is the contract figure."""
import argparse, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("LSEG_SYNTHETIC_TOKENS", "1")     # synthetic weights: stand-in token ids (lseg_hip/tokenizer.py)
warnings.simplefilter("ignore")
from modules.lseg_module import LSegModule
from lseg_hip.synth import synthetic_state_dict, synthetic_images

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--slow", action="store_true")
a = ap.parse_args()


def make():
    m = LSegModule("", "ade20k", a.batch, 0.004, 240, backbone="clip_vitl16_384", num_features=256, arch_option=0, block_depth=0,
                   activation="lrelu", ignore_index=-1, weight_decay=1e-4)
    m.net.load_state_dict(synthetic_state_dict(m.net.cfg, seed=0))
    return m.cuda().train()


x = synthetic_images(a.batch, 480, 480, seed=0).cuda()
g = torch.Generator().manual_seed(1)
t = torch.randint(0, 150, (a.batch, 480, 480), generator=g); t[torch.rand(t.shape, generator=g) < 0.2] = -1
t = t.cuda()


def timed(fn, name):
    for _ in range(a.warmup):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"{name}: {dt * 1e3:.1f} ms/step -> {a.batch / dt:.1f} images/s (loss {float(loss):.4f})", flush=True)
    return dt


m = make()
(opt,), _ = m.configure_optimizers()


def lightning_step():
    loss = m.training_step((x, t), 0)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss


dt_l = timed(lightning_step, "training_step + loss.backward() + EngineSGD.step() + zero_grad()")
del m, opt
torch.cuda.empty_cache()
m2 = make()
dt_n = timed(lambda: m2.native_training_step(x, t), "native_training_step")
print(f"lightning-shaped loop = {100 * dt_n / dt_l:.1f} % of the native step rate")
if a.slow:
    del m2
    torch.cuda.empty_cache()
    m3 = make()
    m3.other_kwargs["materialize_logits"] = True
    o3 = torch.optim.SGD([{"params": m3.net.pretrained.parameters(), "lr": m3.base_lr}, {"params": m3.net.scratch.parameters(), "lr": m3.base_lr * 10}],
                         lr=m3.base_lr, momentum=0.9, weight_decay=1e-4)

    def slow_step():
        loss = m3.training_step((x, t), 0)
        loss.backward(); o3.step(); o3.zero_grad()
        return loss
    timed(slow_step, "generic path: [B,K,H,W] logits + torch CE + torch.optim.SGD (re-bind + re-pack every step)")

#!/usr/bin/env python
"""Per-tensor gradient-norm error of the engine's training step against a reference-autograd fixture (tests/golden/ref_train_*.pt),
every tensor, in backward (bucket) order -- the bisecting view of tests/test_gpu_train.py's fixture test (tools)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_images

name = sys.argv[1]
g = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"))
bb, H, W, B, K, seed = g["spec"]
cfg = get_config(bb)
sd = {k: v.cuda() for k, v in synthetic_state_dict(cfg, seed=seed).items()}
x = synthetic_images(B, H, W, seed=seed).cuda()
gg = torch.Generator().manual_seed(1000 + seed)
t = torch.randint(0, K, (B, H, W), generator=gg); t[torch.rand((B, H, W), generator=gg) < 0.2] = -1
eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K)
eng.load_state_dict(sd); eng.set_tokens(g["tokens"]); eng.enable_training(sd)
eng.forward(x, want_logits=False)
loss = eng.backward(target=t.cuda())
torch.cuda.synchronize()
print(name, "env", {k: v for k, v in os.environ.items() if k.startswith("LSEG_")}, "loss", float(loss), "ref", g["loss"])
rows = []
for k in eng.grads:
    r = g["grads"][k]
    mine = float(eng.grads[k].float().norm())
    rows.append((eng.lib.lseg_grad_bucket(eng._h, k.encode()), k, mine / max(r["norm"], 1e-30)))
rows.sort(key=lambda r: (r[0], r[1]))
bad = [r for r in rows if abs(r[2] - 1) > 0.1]
print(f"{len(bad)} of {len(rows)} tensors off by > 10 % in norm")
full = os.environ.get("REPORT_ALL")
for b, k, ratio in rows:
    if b <= 0 or full:
        print(f"  bucket {b:2d} {k:70s} |engine| / |reference| = {ratio:.3f}")
import statistics
for b in sorted({r[0] for r in rows}):
    v = [r[2] for r in rows if r[0] == b]
    print(f"  bucket {b:2d}: median ratio {statistics.median(v):.3f}  min {min(v):.3f}  max {max(v):.3f}")

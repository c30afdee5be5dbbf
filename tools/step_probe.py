#!/usr/bin/env python
"""Time the inference forward of the engine at one batch size (tools; bench.py is the contract line): HIP events around `steps`
forwards; --text-cache keeps the text features (diagnostic only: what the concurrent text tower costs the image tower)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[36]); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--backbone", default="clip_vitl16_384"); ap.add_argument("--size", type=int, default=480); ap.add_argument("--labels", type=int, default=150)
ap.add_argument("--text-cache", action="store_true"); ap.add_argument("--no-logits", action="store_true"); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
cfg = get_config(a.backbone)
sd = {k: v.cuda() for k, v in synthetic_state_dict(cfg, seed=0).items()}
labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[: a.labels]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
for B in a.batch:
    eng = HipEngine(cfg, a.size, a.size, max_batch=B, max_labels=len(labels), image_dtype=a.dtype)
    eng.load_state_dict(sd); eng.set_tokens(tok)
    if a.text_cache: eng.set_text_cache(True)
    x = synthetic_images(B, a.size, a.size, seed=0).cuda()
    kw = dict(want_logits=False, want_argmax=True) if a.no_logits else {}
    for _ in range(a.warmup): eng.forward(x, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.steps): eng.forward(x, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    print(f"B={B} text_cache={int(a.text_cache)} no_logits={int(a.no_logits)}: {ms:.3f} ms/step -> {B / ms * 1e3:.1f} images/s", flush=True)
    del eng

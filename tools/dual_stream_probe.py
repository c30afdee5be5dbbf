#!/usr/bin/env python
"""Does the forward gain from running sub-batches concurrently on several streams?  (tools; bench.py is the contract line)

The ViT block alternates MFMA-bound phases (GEMM K-loops) with HBM-bound ones (LayerNorm, the fp32 residual read-modify-write of
attn.proj / mlp.fc2, the logits write).  A single stream runs them one after the other, every CU in the same phase.  Here the batch is
cut into `--parts` sub-batches, each with its own engine on its own stream; LSEG_GEMM_MAXGRID caps every GEMM's persistent grid so that
the sub-batches' GEMMs run side by side on disjoint CUs.  Prints images/s of the whole batch for each arrangement."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=36); ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--dtype", default="fp16"); ap.add_argument("--text-cache", action="store_true")
ap.add_argument("--delay-us", type=float, default=0.0, help="host sleep between the sub-batches' enqueues")
a = ap.parse_args()
cfg = get_config("clip_vitl16_384")
sd = {k: v.cuda() for k, v in synthetic_state_dict(cfg, seed=0).items()}
labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[:150]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
P = a.parts
sizes = [a.batch // P + (1 if i < a.batch % P else 0) for i in range(P)]
engs = []
for b in sizes:
    e = HipEngine(cfg, 480, 480, max_batch=b, max_labels=150, image_dtype=a.dtype)
    e.load_state_dict(sd); e.set_tokens(tok)
    if a.text_cache: e.set_text_cache(True)
    engs.append(e)
x = synthetic_images(a.batch, 480, 480, seed=0).cuda()
xs = list(torch.split(x, sizes))
streams = [torch.cuda.Stream() for _ in range(P)]
outs = [None] * P

def step():
    if P == 1:
        outs[0] = engs[0].forward(xs[0]); return
    ev = torch.cuda.Event(); ev.record()
    for i in range(P):
        streams[i].wait_event(ev)
        with torch.cuda.stream(streams[i]):
            outs[i] = engs[i].forward(xs[i])
        if a.delay_us > 0 and i + 1 < P:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < a.delay_us: pass
    for s in streams: torch.cuda.current_stream().wait_stream(s)

for _ in range(a.warmup): step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.steps): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
print(f"parts={P} sizes={sizes} maxgrid={os.environ.get('LSEG_GEMM_MAXGRID', '0')} text_cache={int(a.text_cache)} delay={a.delay_us:.0f}us: "
      f"{ms:.3f} ms/step -> {a.batch / ms * 1e3:.1f} images/s", flush=True)

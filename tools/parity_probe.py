#!/usr/bin/env python
"""Stage-by-stage parity of the engine vs the CPU oracle on the full ViT-L/16 config, for both
MFMA operand types (diagnostic: separates rounding noise from logic errors)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels
from oracle.lseg_oracle import lseg_forward
torch.set_num_threads(min(32, os.cpu_count()))
bb = sys.argv[1] if len(sys.argv) > 1 else "clip_vitl16_384"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
cfg = get_config(bb); sd = synthetic_state_dict(cfg, seed=0)
labels = read_labels(os.path.join(ROOT, "lang-seg_amd/label_files/ade20k_objectInfo150.txt"))[:K]
tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
x = synthetic_images(1, 480, 480, seed=0)
with torch.no_grad():
    ref, inter = lseg_forward(sd, x, tok, cfg, return_intermediates=True)
rr = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
for dt in ("bf16", "fp16"):
    eng = HipEngine(cfg, 480, 480, 1, K, image_dtype=dt); eng.load_state_dict(sd); eng.set_tokens(tok); eng.set_debug(True)
    out, am = eng.forward(x.cuda(), want_argmax=True); torch.cuda.synchronize()
    rep = {}
    for l in range(4):
        rep[f"act{l+1}"] = rr(eng.intermediate(f"act{l+1}", inter["acts"][l].shape).cpu(), inter["acts"][l])
        rep[f"path{l+1}"] = rr(eng.intermediate(f"path{l+1}", inter["paths"][l].shape).cpu(), inter["paths"][l])
    low = eng.intermediate("lowres", inter["lowres"].shape).cpu()
    rep["lowres_maxabs"] = (low - inter["lowres"]).abs().max().item()
    rep["logits_maxabs"] = (out.cpu() - ref).abs().max().item()
    lo = inter["lowres"]; mism = am.cpu().long() != lo.argmax(1)
    t2 = lo.topk(2, dim=1).values; margin = t2[:, 0] - t2[:, 1]
    rep["argmax_mismatch_frac"] = mism.float().mean().item()
    rep["max_margin_at_mismatch"] = margin[mism].max().item() if mism.any() else 0.0
    rep["median_margin"] = margin.median().item()
    print(dt, json.dumps({k: round(v, 5) for k, v in rep.items()}), flush=True)
    eng.close()

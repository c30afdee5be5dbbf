#!/usr/bin/env python
"""Where does image b of a batch stop being bit-equal to the same image run alone?  (tools; taps of the debug schedule + production logits)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.engine import HipEngine
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images
bb, H, W, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = get_config(bb); sd = synthetic_state_dict(cfg, seed=2)
tok = synthetic_tokens(["wall", "sky", "tree", "floor", "other"], cfg.text.vocab, cfg.text.ctx)
x = synthetic_images(B, H, W, seed=5).cuda()
for dt in ("bf16", "fp16"):
    for dbg in (True, False):
        eng = HipEngine(cfg, H, W, max_batch=B, max_labels=5, image_dtype=dt, batch_invariant=True)
        eng.load_state_dict(sd); eng.set_tokens(tok); eng.set_debug(dbg)
        ntok = cfg.tokens(H, W)
        def taps(n):
            t = {}
            if dbg:
                for l in range(4):
                    t[f"act{l+1}"] = eng.intermediate(f"act{l+1}", (n, ntok, cfg.dim)).clone()
                for l in range(4):
                    hh, ww = (H // 4, W // 4) if l == 0 else ((H // 8, W // 8) if l == 1 else ((H // 16, W // 16) if l == 2 else (H // 32, W // 32)))
                    t[f"rn{l+1}"] = eng.intermediate(f"rn{l+1}", (n, cfg.features, hh, ww)).clone()
                    t[f"path{l+1}"] = eng.intermediate(f"path{l+1}", (n, cfg.features, 2 * hh, 2 * ww)).clone()
            t["lowres"] = eng.intermediate("lowres", (n, 5, H // 2, W // 2)).clone()
            return t
        ob = eng.forward(x).clone(); tb = taps(B)
        o1 = eng.forward(x[1:2]).clone(); t1 = taps(1)
        rep = {k: float((tb[k][1] - t1[k][0]).abs().max()) for k in tb}
        rep["logits"] = float((ob[1] - o1[0]).abs().max())
        print(dt, "debug" if dbg else "production", {k: (f"{v:.2e}") for k, v in rep.items()}, flush=True)
        eng.close()

#!/usr/bin/env python
"""The NOISE FLOOR of "argmax masks bit-exact" (VERDICT r5 item 4a), measured on the CPU, no GPU needed:

  floor A  the fp32 oracle (oracle/lseg_oracle.py) against the REFERENCE-RUN fixtures tests/golden/ref_full_*.pt: two fp32 CPU
           implementations of the same network -- flips and largest reference margin at a flip, at 240x240 and at 480x480.
  floor B  the SAME fp32 oracle against itself at 1 thread vs N threads (other GEMM blocking = other fp32 summation order): what one
           implementation loses to scheduling alone.

Writes profiles/r06_parity_floor.json; tests/test_gpu_forward.py and bench.py state the engine's flips as a multiple of floor A.
Checker-side tooling: imports oracle/ (allowed for tests / tools), never the product path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip.config import get_config
from lseg_hip.synth import synthetic_images, fixture_state_dict
from oracle.lseg_oracle import lseg_forward

GOLD = os.path.join(ROOT, "tests", "golden")
names = sorted(f[:-3] for f in os.listdir(GOLD) if f.startswith("ref_full_") and not f.endswith("_out480.pt"))
nthreads = int(os.environ.get("FLOOR_THREADS", str(min(8, os.cpu_count() or 1))))
out = {"_meta": {"what": __doc__.strip().split("\n\n")[0], "threads": nthreads, "made_by": "tools/parity_floor.py"}}
for name in names:
    g = torch.load(os.path.join(GOLD, name + ".pt"))
    bb, H, W, B, K, seed, arch, depth = g["spec"]
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = fixture_state_dict(cfg, seed, g)
    x = synthetic_images(B, H, W, seed=seed)
    tf = g["text_features"].float()          # same text features on every side: the floor of the IMAGE path + correlation
    res = {}
    runs = {}
    for nt in (nthreads, 1):
        torch.set_num_threads(nt)
        t0 = time.time()
        with torch.no_grad():
            o, inter = lseg_forward(sd, x, g["tokens"], cfg, return_intermediates=True, text_features=tf)
        runs[nt] = (o, inter["lowres"])
        print(f"{name}: oracle forward at {nt} thread(s) {time.time() - t0:.1f} s", flush=True)
    o, low = runs[nthreads]
    mism = low.argmax(1) != g["argmax_lowres"].long()
    margin = g["margin_lowres"].float()
    err = max((low[:, :, ::8, ::8] - g["lowres_sub8"].float()).abs().max().item(),
              (low.gather(1, g["top2_idx"].long()) - g["top2_val"].float()).abs().max().item())
    res["oracle_vs_reference_240"] = {"argmax_mismatch_frac": mism.float().mean().item(), "max_abs_dlogit": err,
                                      "max_reference_margin_at_mismatch": margin[mism].max().item() if mism.any() else 0.0}
    side = os.path.join(GOLD, name + "_out480.pt")
    if os.path.exists(side):
        g4 = torch.load(side)
        ref_am, m4g = g4["argmax"].long(), g4["margin"].float()
        m4 = o.argmax(1) != ref_am
        res["oracle_vs_reference_480"] = {"argmax_mismatch_frac": m4.float().mean().item(),
                                          "max_reference_margin_at_mismatch": m4g[m4].max().item() if m4.any() else 0.0}
    o1, low1 = runs[1]
    res["oracle_1thread_vs_%dthreads" % nthreads] = {
        "argmax_mismatch_frac_240": (low1.argmax(1) != low.argmax(1)).float().mean().item(),
        "argmax_mismatch_frac_480": (o1.argmax(1) != o.argmax(1)).float().mean().item(),
        "max_abs_dlogit": (low1 - low).abs().max().item()}
    out[name] = res
    print(name, json.dumps(res), flush=True)
with open(os.path.join(ROOT, "profiles", "r06_parity_floor.json"), "w") as f:
    json.dump(out, f, indent=1)

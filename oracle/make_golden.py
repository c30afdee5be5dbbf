"""Generate the committed golden fixtures under tests/golden/ from the oracle.

Run here (build container):  python oracle/make_golden.py [--full]
The reference cannot be imported in this environment, so the
fixtures are *oracle* outputs on seeded synthetic weights: they pin the oracle
against accidental edits and against CPU/BLAS differences between machines, and
they give the GPU parity tests a box-independent target.  TEST INFRASTRUCTURE.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))

from lseg_hip.config import get_config                                    # noqa: E402
from lseg_hip.synth import (synthetic_state_dict, synthetic_tokens,       # noqa: E402
                            synthetic_images, read_labels)
from oracle.lseg_oracle import lseg_forward                              # noqa: E402

LABELS = os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt")

# name -> (backbone, H, W, B, K, seed, arch_option, block_depth)
CASES = {
    "tiny16_64x64_k5": ("tiny16", 64, 64, 2, 5, 0, 0, 0),
    "tiny16_96x64_k7": ("tiny16", 96, 64, 1, 7, 1, 0, 0),
    "tiny32_96x96_k7": ("tiny32", 96, 96, 1, 7, 2, 0, 0),
    "tiny16_64x64_k5_arch1": ("tiny16", 64, 64, 1, 5, 3, 1, 2),
    "tiny16_64x64_k5_arch2": ("tiny16", 64, 64, 1, 5, 4, 2, 2),
}
FULL = {
    # BASELINE.json configs[1]: ViT-L/16, 480x480, K=150 (B=1)
    "vitl16_480_k150": ("clip_vitl16_384", 480, 480, 1, 150, 0, 0, 0),
    # configs[0]: ViT-B/32, 480x480, K=7
    "vitb32_480_k7": ("clip_vitb32_384", 480, 480, 1, 7, 0, 0, 0),
}


# zero-shot network (LSegNetZS.forward, lseg_net_zs.py:177-214): name -> (backbone, H, W, class_info, seed)
FSS = os.path.join(ROOT, "lang-seg_amd", "label_files", "fewshot_fss.txt")
ZS_CASES = {
    "tiny16_64x64_zs": ("tiny16", 64, 64, (3, 0, 7), 5),
}


def run_zs_case(spec):
    bb, H, W, class_info, seed = spec
    cfg = get_config(bb, arch_option=0, block_depth=0, activation="lrelu")
    sd = synthetic_state_dict(cfg, seed=seed)
    names = read_labels(FSS)
    # self.texts[class_i] = clip.tokenize(['others', label_list[class_i]])   (lseg_net_zs.py:170-176)
    tok = torch.cat([synthetic_tokens(["others", names[c]], cfg.text.vocab, cfg.text.ctx) for c in class_info], 0)
    x = synthetic_images(len(class_info), H, W, seed=seed)
    with torch.no_grad():
        out, inter = lseg_forward(sd, x, tok, cfg, return_intermediates=True, labels_per_image=2)
    return cfg, sd, x, tok, out, inter


def run_case(spec):
    bb, H, W, B, K, seed, arch, depth = spec
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = synthetic_state_dict(cfg, seed=seed)
    labels = read_labels(LABELS)[:K]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(B, H, W, seed=seed)
    with torch.no_grad():
        out, inter = lseg_forward(sd, x, tok, cfg, return_intermediates=True)
    return cfg, out, inter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also regenerate the full-size (ViT-L/B) fixtures")
    args = ap.parse_args()
    gd = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gd, exist_ok=True)
    for name, spec in CASES.items():
        cfg, out, inter = run_case(spec)
        torch.save({"spec": spec, "logits": out.clone(),
                    "text_features": inter["text_features"].to(torch.float16),
                    "lowres": inter["lowres"].to(torch.float16),
                    "acts_mean_abs": [float(a.abs().mean()) for a in inter["acts"]],
                    "paths_mean_abs": [float(p.abs().mean()) for p in inter["paths"]]},
                   os.path.join(gd, name + ".pt"))
        print(name, tuple(out.shape), float(out.abs().mean()))
    for name, spec in ZS_CASES.items():
        cfg, sd, x, tok, out, inter = run_zs_case(spec)
        torch.save({"spec": spec, "logits": out.clone(), "text_features": inter["text_features"].to(torch.float16)},
                   os.path.join(gd, name + ".pt"))
        print(name, tuple(out.shape), float(out.abs().mean()))
    if args.full:
        for name, spec in FULL.items():
            cfg, out, inter = run_case(spec)
            low = inter["lowres"]                       # [B,K,H/2,W/2], fp16-valued
            torch.save({"spec": spec,
                        "text_features": inter["text_features"].to(torch.float16),
                        "lowres_sub8": low[:, :, ::8, ::8].to(torch.float16).clone(),
                        "argmax_lowres": low.argmax(1).to(torch.uint8).clone(),
                        "logits_sub16": out[:, :, ::16, ::16].clone(),
                        "logits_mean_abs": float(out.abs().mean()),
                        "acts_mean_abs": [float(a.abs().mean()) for a in inter["acts"]],
                        "paths_mean_abs": [float(p.abs().mean()) for p in inter["paths"]]},
                       os.path.join(gd, name + ".pt"))
            print(name, tuple(out.shape), float(out.abs().mean()))


if __name__ == "__main__":
    main()

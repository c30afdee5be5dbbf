"""Element-wise gradient comparison on the SEEDED RANDOM network: lseg_backward(dlogits) vs fp32 autograd through the oracle
(oracle.lseg_oracle.lseg_forward, bn_train=True), every gradient tensor.  The most chaotic comparison of the suite -- a peaky-softmax
random net amplifies every bf16 rounding -- so pytest collects it LAST (file name), behind the reference-autograd fixtures that pin
BASELINE configs[3] (tests/test_gpu_train.py), and its bars come from a measured distribution instead of one run:

  profiles/r05_train_spread.txt  (tests/gpu_train_spread.py: 3 configs x 2 regimes x 6 seeds x runs; deterministic reductions:
  run-to-run spread exactly 0 -- every run bit-equal -- so what varies is the SEED, not the run)

Tensors with <= 1024 elements (cls_token, biases, LayerNorm / BatchNorm affine parameters: a handful of bf16 flips move their relative
Frobenius error by several points) are held to NORM + COSINE bars like the fixture tests; larger tensors to the element-wise
(relative Frobenius) bar.  A mis-wired or missing term shows as >= 70-100 % / cosine <= 0.7 in either regime.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from lseg_hip.config import get_config                                            # noqa: E402
from lseg_hip.engine import HipEngine                                             # noqa: E402
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels   # noqa: E402
from lseg_hip.train import grad_bucket_index                                      # noqa: E402
from oracle import make_golden as MG                                              # noqa: E402
from train_helpers import oracle_backward as _oracle_backward, away_from_the_relu_kinks as _away_from_the_relu_kinks, rel, cosine  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# bars: ~2x the three tested (case, seed) pairs = the ~80-90th percentile of the 18-pair seed distribution (profiles/r05_train_spread.txt);
# element-wise on tensors > 1024 elements, norm error and cosine on every tensor, per-case median -- per regime (smooth / rough)
# round 6: tensors with <= 1024 elements -- the biases, BatchNorm / LayerNorm parameters whose reductions round 5 rewrote -- keep an
# ELEMENT-WISE bar in the smooth regime too (no ReLU flips there: their error is the reduction's own); rough regime: norm + cosine as before
BARS = {True: dict(big=0.15, small=0.15, norm=0.10, cos=0.985, median=0.04),
        False: dict(big=0.55, small=None, norm=0.28, cos=0.85, median=0.35)}


@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("bb,H,W,B,K,seed", [("tiny16", 64, 64, 2, 5, 3), ("tiny32", 96, 96, 2, 7, 4), ("tiny16", 96, 64, 1, 3, 6)])
def test_backward_matches_oracle_autograd_gradient_by_gradient(bb, H, W, B, K, seed, smooth):
    """lseg_backward(dlogits) vs fp32 autograd through the oracle, every gradient tensor.  The engine's saved activations and
    inter-kernel gradients are bf16.  smooth=True: ReLU inputs kept positive (train_helpers.away_from_the_relu_kinks) -- measures the
    backward arithmetic and its wiring; smooth=False: the seeded zero-centred net, where bf16-vs-fp32 ReLU mask flips dominate.
    Bars: module docstring / BARS."""
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    if smooth:
        sd = _away_from_the_relu_kinks(sd, cfg)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(77 + seed)
    # d(logits) ~ 1e-3: well inside fp16's normal range.  The engine (like the reference, lseg_net.py:194 under autograd) carries this
    # gradient through the correlation in HALF precision; at the ~1e-6 magnitude of a real mean-CE gradient that is subnormal
    # quantisation noise (step 6e-8), which the reference-autograd fixtures at 480x480 cover -- here the backward ARITHMETIC is measured
    dl = torch.randn((B, K, H, W), generator=g) * 1e-3
    ref_out, ref_grads = _oracle_backward(sd, x, tok, cfg, dl)
    sd_dev = {k: v.cuda() for k, v in sd.items()}
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K, deterministic=True)
    eng.load_state_dict(sd_dev)
    eng.set_tokens(tok)
    eng.enable_training(sd_dev)
    out = eng.forward(x.cuda())
    eng.backward(dlogits=dl.cuda())
    torch.cuda.synchronize()
    assert (out.cpu() - ref_out).abs().max().item() <= 0.35
    assert set(eng.grads) == set(ref_grads), sorted(set(eng.grads) ^ set(ref_grads))[:10]
    report = {k: rel(eng.grads[k].cpu(), ref_grads[k]) for k in sorted(ref_grads)}
    worst = sorted(report.items(), key=lambda kv: -kv[1])[:10]
    nerr = max(abs(eng.grads[k].float().norm().item() - ref_grads[k].norm().item()) / ref_grads[k].norm().item() for k in ref_grads)
    print(f"[{bb} {H}x{W} smooth={smooth}] median gradient error {sorted(report.values())[len(report) // 2]:.4f}; max norm error {nerr:.4f}; "
          f"worst:", [(k, round(v, 4)) for k, v in worst])
    if os.path.isdir(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out")):
        import json
        with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", f"train_grad_report_{bb}_{H}x{W}_{int(smooth)}.json"), "w") as f:
            json.dump({k: [report[k], float(ref_grads[k].norm()), float(eng.grads[k].float().norm())] for k in report}, f, indent=0)
    bars = BARS[smooth]
    nerrs = {k: abs(eng.grads[k].float().norm().item() - ref_grads[k].norm().item()) / ref_grads[k].norm().item() for k in ref_grads}
    cosv = {k: cosine(eng.grads[k].cpu(), ref_grads[k]) for k in ref_grads}
    med = sorted(report.values())[len(report) // 2]
    bad_big = {k: round(v, 4) for k, v in report.items() if ref_grads[k].numel() > 1024 and not v <= bars["big"]}
    small = {k: v for k, v in report.items() if ref_grads[k].numel() <= 1024}
    if small:
        print(f"   worst element-wise error on the <= 1024-element tensors: {max(small.values()):.4f} ({max(small, key=small.get)})")
    if bars["small"] is not None:
        bad_big.update({k: round(v, 4) for k, v in small.items() if not v <= bars["small"]})
    bad_norm = {k: round(v, 4) for k, v in nerrs.items() if not v <= bars["norm"]}
    bad_cos = {k: round(v, 4) for k, v in cosv.items() if not v >= bars["cos"]}
    print(f"   min cosine {min(cosv.values()):.4f} ({min(cosv, key=cosv.get)}); bars {bars}")
    assert not bad_big and not bad_norm and not bad_cos and med <= bars["median"], (bad_big, bad_norm, bad_cos, med)

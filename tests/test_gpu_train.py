"""The engine-level training step (BASELINE config 4) through the C ABI:
lseg_set_train / lseg_forward (train mode) / lseg_backward / lseg_sgd_step, against

  * tests/golden/ref_train_*.pt -- loss, gradient norms, sums and stored elements of each gradient from back-propagating through
    the reference's own network code (oracle/make_ref_train_golden.py), full ViT-L/16 and ViT-B/32 dimensions: FIRST in this file --
    `ref_train_full_*` pin BASELINE configs[3] at its own shape and must not hide behind a noisier comparison;
  * oracle.lseg_oracle.training_step (fp32 autograd restatement of modules/lsegmentation_module.py:66-81, itself pinned by the
    reference-made fixtures in tests/test_oracle_train_ref_golden.py): loss and EVERY gradient tensor;
  * itself: the same step twice is BIT-identical under deterministic reductions (lseg_config.flags bit 3; the whole parity suite runs
    with them -- the engine's default --, tests/conftest.py), and the fp32-atomics sums (LSEG_DETERMINISTIC=0) stay within rounding of them.

The element-wise comparison on the seeded random network (the most chaotic one) lives in tests/test_zz_gpu_random_net_gradients.py,
which pytest collects last.

Tolerances: the engine keeps saved activations and inter-kernel gradients in bf16 (8-bit mantissa) with fp32 accumulation, the
reference is fp32 end to end.  Loss within 1 % (measured 0.03-0.07 %).
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from lseg_hip.config import get_config                                            # noqa: E402
from lseg_hip.engine import HipEngine                                             # noqa: E402
from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels   # noqa: E402
from oracle.lseg_oracle import training_step                                      # noqa: E402
from oracle import make_golden as MG                                              # noqa: E402
from train_helpers import target_map as _target, engine_step as _engine_step, rel  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_REF = sorted((f[:-3] for f in os.listdir(GOLD) if f.startswith("ref_train_")), key=lambda n: ("_full_" not in n, n))


def _sample_index(numel, n=64):                       # == oracle/make_ref_train_golden.sample_index
    n = min(n, numel)
    return (torch.arange(n, dtype=torch.int64) * (numel - 1)) // max(n - 1, 1)


def _compare_with_fixture(eng, g):
    """per-tensor metrics of the engine's gradients against a reference-autograd fixture: norm error, first-16 and strided-sample element
    errors (relative to the tensor's scale), sum error, cosine over the stored elements"""
    names = {n for n in g["grads"] if not n.startswith("clip_pretrained.")}
    assert set(eng.grads) == names, sorted(set(eng.grads) ^ names)[:10]
    nerr, herr, serr, sumerr, cosv = {}, {}, {}, {}, {}
    for n in sorted(names):
        r = g["grads"][n]
        mine = eng.grads[n].float().cpu()
        rms = r["norm"] / max(1.0, mine.numel() ** 0.5)
        nerr[n] = abs(float(mine.norm()) - r["norm"]) / max(r["norm"], 1e-20)
        scale = max(float(r["head"].abs().max()), rms, 1e-20)
        herr[n] = (mine.flatten()[:16] - r["head"]).abs().max().item() / scale
        if "sample" in r:
            sscale = max(float(r["sample"].abs().max()), rms, 1e-20)
            got = mine.flatten()[_sample_index(mine.numel(), r["sample"].numel())]
            serr[n] = (got - r["sample"]).abs().max().item() / sscale
            sumerr[n] = abs(float(mine.double().sum()) - r["sum"]) / max(r["norm"] * mine.numel() ** 0.5, 1e-20)
            # direction: cosine between the engine's and the reference's gradient over the stored elements (first 16 + the strided sample:
            # 1040 per tensor in the 480 x 480 fixtures since round 6, 80 in the small ones) -- a permuted or mis-signed gradient with the
            # right norm has cosine ~ 0
            a, b = torch.cat([mine.flatten()[:16], got]).double(), torch.cat([r["head"], r["sample"]]).double()
            if float(b.norm()) > 0 and b.numel() >= 32:
                cosv[n] = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
    return {"nerr": nerr, "herr": herr, "serr": serr, "sumerr": sumerr, "cos": cosv}


# Bars (round 6), from the measured tables profiles/r05_train_parity_table.txt / r06_train_parity_table.txt: worst
# gradient-norm error 3.5-6.5 % (bar 10 %), median 0.3-1.2 % at 480 x 480 (bar 2 %), 0.6-3.2 % on the small cases (bar 5 %); element errors
# relative to the tensor's scale: medians 0.06-0.18 (bar 0.25, was 0.35), single elements up to 1.23 (bar 1.5); cosine median 0.984-0.998
# (bar 0.98), worst single tensor 0.873-0.905 (bar 0.85, was 0.80).  The worst cosines do NOT come from small samples: the B = 8 fixture
# holds 1040 elements per tensor and its worst is 0.896 (patch_embed.proj.weight) -- early-layer gradients behind 24 blocks of bf16
# ReLU / softmax roundings and the reference's fp16-subnormal head gradient (DESIGN par. 3.6).  What the bars DO catch is pinned by
# test_the_fixture_bars_catch_a_dropped_residual_branch below.
COS_MEDIAN, COS_WORST = 0.98, 0.85
ELEM_MEDIAN, ELEM_WORST = 0.25, 1.5


def _violations(m, full):
    med = lambda d: sorted(d.values())[len(d) // 2]
    v = []
    if max(m["nerr"].values()) > 0.10: v.append(("worst gradient-norm error", max(m["nerr"].values())))
    if med(m["nerr"]) > (0.02 if full else 0.05): v.append(("median gradient-norm error", med(m["nerr"])))
    if max(m["herr"].values()) > ELEM_WORST: v.append(("worst first-16 element error", max(m["herr"].values())))
    if med(m["herr"]) > ELEM_MEDIAN: v.append(("median first-16 element error", med(m["herr"])))
    if m["serr"]:
        if max(m["serr"].values()) > ELEM_WORST: v.append(("worst strided element error", max(m["serr"].values())))
        if med(m["serr"]) > ELEM_MEDIAN: v.append(("median strided element error", med(m["serr"])))
        if max(m["sumerr"].values()) > 0.25: v.append(("worst sum error", max(m["sumerr"].values())))
    if m["cos"]:
        if med(m["cos"]) < COS_MEDIAN: v.append(("median cosine", med(m["cos"])))
        if min(m["cos"].values()) < COS_WORST: v.append(("worst cosine", min(m["cos"].values())))
    return v


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.gpu_fast) if "_full_" not in n else n for n in TRAIN_REF])
def test_training_step_matches_fixtures_made_by_reference_autograd(name):
    """The engine's training step against loss + gradients of the REFERENCE'S OWN network under torch autograd
    (oracle/make_ref_train_golden.py).  `ref_train_full_*` = BASELINE configs[3] at its own shape (ViT-L/16, 480x480, K = 150;
    B = 1, 2 and the per-GPU batch 8): loss, every gradient's norm, its sum, its first 16 and 1024 evenly strided elements."""
    g = torch.load(os.path.join(GOLD, name + ".pt"))
    bb, H, W, B, K, seed = g["spec"]
    full = "_full_" in name
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    x = synthetic_images(B, H, W, seed=seed)
    eng, out, loss, _ = _engine_step(cfg, sd, x, _target(B, H, W, K, seed), g["tokens"])
    assert abs(loss.item() - g["loss"]) <= 1e-2 * abs(g["loss"]), (loss.item(), g["loss"])
    m = _compare_with_fixture(eng, g)
    nerr, herr, serr, sumerr, cosv = m["nerr"], m["herr"], m["serr"], m["sumerr"], m["cos"]
    med = lambda d: sorted(d.values())[len(d) // 2]
    wn = sorted(nerr.items(), key=lambda kv: -kv[1])[:5]
    wh = sorted(herr.items(), key=lambda kv: -kv[1])[:5]
    ws = sorted(serr.items(), key=lambda kv: -kv[1])[:5]
    line = (f"{name}: loss {loss.item():.5f} vs {g['loss']:.5f}; gradient-norm error median {med(nerr):.4f} worst "
            f"{[(k, round(v, 4)) for k, v in wn[:3]]}; first-16 error median {med(herr):.3f} worst {wh[0][1]:.3f}; "
            f"strided-sample error median {med(serr) if serr else -1:.3f} worst {ws[0][1] if ws else -1:.3f}; "
            f"sum error worst {max(sumerr.values()) if sumerr else -1:.4f}")
    if cosv:
        wc = sorted(cosv.items(), key=lambda kv: kv[1])[:3]
        n_el = 16 + next(iter(g["grads"].values()))["sample"].numel()
        line += f"; cosine over the stored elements (<= {n_el} per tensor) median {med(cosv):.4f} worst {[(k, round(v, 3)) for k, v in wc]}"
    print(line)
    out_dir = os.path.join(os.path.dirname(GOLD), "..", "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "train_parity_table.txt"), "a") as f:
            f.write(line + "\n")
    bad = _violations(m, full)
    assert not bad, (bad, wn[:3], sorted(cosv.items(), key=lambda kv: kv[1])[:3])


@pytest.mark.gpu_fast
@pytest.mark.parametrize("fault", ["drop_mlp_branch_of_block_5", "scale_layer1_rn_by_1.5", "drop_attention_branch_of_block_20"])
def test_the_fixture_bars_catch_a_dropped_residual_branch(fault):
    """Fault injection: the SAME comparison, with one block's residual branch removed from the network the engine runs
    (or the head mis-scaled by 15 %) -- the bars above must fail.  A bar that passes a network with a missing branch measures nothing."""
    name = "ref_train_vitl16_64x64_k5_b2"
    g = torch.load(os.path.join(GOLD, name + ".pt"))
    bb, H, W, B, K, seed = g["spec"]
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    if fault == "drop_mlp_branch_of_block_5":
        sd["pretrained.model.blocks.5.mlp.fc2.weight"].zero_(); sd["pretrained.model.blocks.5.mlp.fc2.bias"].zero_()
    elif fault == "drop_attention_branch_of_block_20":
        sd["pretrained.model.blocks.20.attn.proj.weight"].zero_(); sd["pretrained.model.blocks.20.attn.proj.bias"].zero_()
    else:
        sd["scratch.layer1_rn.weight"].mul_(1.5)
    x = synthetic_images(B, H, W, seed=seed)
    eng, out, loss, _ = _engine_step(cfg, sd, x, _target(B, H, W, K, seed), g["tokens"])
    bad = _violations(_compare_with_fixture(eng, g), full=False)
    print(fault, "->", bad)
    assert bad, f"{fault}: every bar still passes"


@pytest.mark.parametrize("bb,H,W,B,K,seed", [("tiny16", 64, 64, 2, 5, 3), ("tiny32", 96, 96, 2, 7, 4)])
def test_training_step_loss_and_gradients_match_the_oracle(bb, H, W, B, K, seed):
    """The whole step with the loss inside (lseg_backward(target): fused CE) vs oracle.training_step.  Here the gradients also
    inherit the loss's sensitivity to the bf16 forward (p = softmax(14.3 * cosine)): 10-15 % per tensor, norms within 3 %."""
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(B, H, W, seed=seed)
    target = _target(B, H, W, K, seed)
    ref_loss, ref_grads = training_step(sd, x, target, tok, cfg, ignore_index=-1)
    eng, out, loss, sd_dev = _engine_step(cfg, sd, x, target, tok)
    assert abs(loss.item() - float(ref_loss)) <= 1e-2 * abs(float(ref_loss)), (loss.item(), float(ref_loss))
    trainable = {k for k in ref_grads if not k.startswith("clip_pretrained.")}
    assert set(eng.grads) == trainable, sorted(set(eng.grads) ^ trainable)[:10]
    report = {k: rel(eng.grads[k].cpu(), ref_grads[k]) for k in sorted(trainable)}
    nerr = {k: abs(eng.grads[k].float().norm().item() - ref_grads[k].norm().item()) / ref_grads[k].norm().item() for k in trainable}
    print(f"[{bb}] loss {loss.item():.5f} vs {float(ref_loss):.5f}; median / max gradient error {sorted(report.values())[len(report) // 2]:.4f} / "
          f"{max(report.values()):.4f}; max norm error {max(nerr.values()):.4f}")
    assert max(report.values()) <= 0.35 and max(nerr.values()) <= 0.10
    # running statistics were updated in the caller's tensors like nn.BatchNorm2d(momentum=0.1) does
    k0 = "scratch.refinenet1.resConfUnit2.bn1.running_mean"
    assert not torch.equal(sd_dev[k0].cpu(), sd[k0])


def test_the_same_training_step_twice_is_bit_identical():
    """Deterministic reductions (HipEngine(deterministic=True), lseg_config.flags bit 3): forward + backward run twice on one engine and
    once more on a FRESH engine give bit-identical logits, loss and gradients -- no fp32 atomics on the gradient path (bias / BatchNorm
    column sums through partial rows, the pos-embed resize transpose as a gather), fixed split-K orders, no data race.  A read-before-land
    race in a kernel (VERDICT r4's suspicion about the hand-waited LDS transpose reads of the K-major weight-gradient GEMM) would show
    here as run-to-run differences."""
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=3)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:5], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(2, 64, 64, seed=3)
    target = _target(2, 64, 64, 5, 3)
    runs = []
    eng = None
    for r in range(3):
        if r == 2:
            eng = None                                   # a fresh engine: nothing carried over
        sd_r = {k: v.clone() for k, v in sd.items()}     # (train-mode BatchNorm moves the running statistics in the caller's tensors)
        if eng is None:
            eng, out, loss, _ = _engine_step(cfg, sd_r, x, target, tok, deterministic=True)
        else:
            eng, out, loss, _ = _engine_step(cfg, sd_r, x, target, tok, eng=eng)
        runs.append((out.clone(), float(loss), {k: v.clone() for k, v in eng.grads.items()}))
    for r in (1, 2):
        assert torch.equal(runs[r][0], runs[0][0]), f"logits of run {r} differ"
        assert runs[r][1] == runs[0][1], (runs[r][1], runs[0][1])
        diff = [k for k in runs[0][2] if not torch.equal(runs[r][2][k], runs[0][2][k])]
        assert not diff, (r, diff[:8])


@pytest.mark.parametrize("bb,H,W,B,K,seed", [("clip_vitl16_384", 96, 96, 2, 5, 2)])
def test_the_same_training_step_twice_is_bit_identical_at_vitl_width(bb, H, W, B, K, seed):
    """The same at ViT-L/16 widths (D = 1024, 16 heads, 256-channel DPT maps): the production tile configurations of every dgrad / wgrad
    GEMM (K-major 128x128 split-K slabs included) and of the attention backward."""
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(B, H, W, seed=seed)
    target = _target(B, H, W, K, seed)
    runs = []
    for r in range(3):                                   # fresh engines on fresh copies of the weights (BatchNorm running statistics move)
        eng, out, loss, _ = _engine_step(cfg, {k: v.clone() for k, v in sd.items()}, x, target, tok, deterministic=True)
        runs.append((out.clone(), float(loss), {k: v.clone() for k, v in eng.grads.items()}))
        eng.close()
    for r in (1, 2):
        assert torch.equal(runs[r][0], runs[0][0]) and runs[r][1] == runs[0][1]
        diff = [k for k in runs[0][2] if not torch.equal(runs[r][2][k], runs[0][2][k])]
        assert not diff, (r, diff[:8])


def test_bucket_rule_and_gradient_accumulation():
    """The engine's bucket rule == the Python mirror the DDP front uses; a second backward with accumulate=True doubles every gradient
    (accumulate_grad_batches, train.sh)."""
    from lseg_hip.train import grad_bucket_index
    for bb, H, W, B, K, seed in (("tiny16", 64, 64, 2, 5, 3), ("tiny32", 96, 96, 2, 7, 4)):
        cfg = get_config(bb)
        sd = synthetic_state_dict(cfg, seed=seed)
        tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
        x = synthetic_images(B, H, W, seed=seed)
        dl = torch.randn((B, K, H, W), generator=torch.Generator().manual_seed(77 + seed)) * 1e-3
        sd_dev = {k: v.cuda() for k, v in sd.items()}
        eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K)
        eng.load_state_dict(sd_dev)
        eng.set_tokens(tok)
        eng.enable_training(sd_dev)
        eng.forward(x.cuda())
        eng.backward(dlogits=dl.cuda())
        torch.cuda.synchronize()
        for k in eng.grads:
            assert eng.lib.lseg_grad_bucket(eng._h, k.encode()) == grad_bucket_index(k, cfg.depth, cfg.hooks), k
        before = {k: v.clone() for k, v in eng.grads.items()}
        eng.backward(dlogits=dl.cuda(), accumulate=True)
        torch.cuda.synchronize()
        worst_acc = max(rel(eng.grads[k], 2 * before[k]) for k in before)
        assert worst_acc <= 1e-2, worst_acc


@pytest.mark.parametrize("bb,H,W,K,seed", [("tiny16", 64, 64, 5, 3), ("clip_vitl16_384", 64, 64, 5, 11)])
def test_atomic_sums_stay_within_rounding_of_the_deterministic_ones(bb, H, W, K, seed):
    """The column sums on fp32 atomics (deterministic=False; rounds 2-4's only form): same numbers up to the summation order.  A last-bit difference in a BatchNorm
    batch sum flips a bf16 rounding somewhere, which the layers above amplify on the seeded random net: measured median 0.5 %, worst
    1.7 % per tensor (lease A of round 5) -- the run-to-run noise of the atomics path itself (profiles/r05_train_spread.txt: +-0.8 points on the
    oracle comparison), an order of magnitude below either engine's distance to the fp32 oracle on those tensors.  Round 6: also at
    ViT-L/16 WIDTH (1024-wide Linears, 256-channel BatchNorms: the column kernels' full-width paths), not only on the 64-wide twin."""
    cfg = get_config(bb)
    sd = synthetic_state_dict(cfg, seed=seed)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(2, H, W, seed=seed)
    target = _target(2, H, W, K, seed)
    ea, outa, la, _ = _engine_step(cfg, {k: v.clone() for k, v in sd.items()}, x, target, tok, deterministic=True)
    eb, outb, lb, _ = _engine_step(cfg, {k: v.clone() for k, v in sd.items()}, x, target, tok, deterministic=False)
    assert ea.deterministic and not eb.deterministic
    assert abs(float(la) - float(lb)) <= 1e-4 * abs(float(la))
    d = {k: rel(eb.grads[k], ea.grads[k]) for k in ea.grads}
    med = sorted(d.values())[len(d) // 2]
    print(f"atomics vs deterministic: median {med:.2e}, worst {max(d.values()):.2e} ({max(d, key=d.get)})")
    assert med <= 2e-2 and max(d.values()) <= 8e-2, (med, sorted(d.items(), key=lambda kv: -kv[1])[:5])


def test_fused_sgd_matches_torch_sgd():
    """lseg_sgd_step == torch.optim.SGD(momentum=0.9, weight_decay=1e-4) with the reference's two learning-rate groups
    (modules/lsegmentation_module.py:119-127,165-171) on the engine's own gradients, two consecutive steps."""
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=5)
    tok = synthetic_tokens(["wall", "sky", "tree"], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(2, 64, 64, seed=5)
    target = _target(2, 64, 64, 3, 5)
    eng, out0, _, sd_dev = _engine_step(cfg, sd, x, target, tok)
    out0 = out0.clone()
    keys = sorted(eng.grads)
    grads0 = {k: eng.grads[k].clone() for k in keys}
    params = {k: sd_dev[k].clone().requires_grad_(True) for k in keys}
    opt = torch.optim.SGD([{"params": [params[k] for k in keys if k.startswith("pretrained.")], "lr": 0.01},
                           {"params": [params[k] for k in keys if k.startswith("scratch.")], "lr": 0.1}],
                          lr=0.01, momentum=0.9, weight_decay=1e-4)
    for step in range(2):
        for k in keys:
            params[k].grad = eng.grads[k].clone()
        opt.step()
        eng.sgd_step(0.01, 0.1, 0.9, 1e-4)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.allclose(eng.bound[k], params[k].detach(), rtol=1e-5, atol=1e-7), (step, k)
        if step == 0:      # the updated weights are live in the engine: a new step runs on them
            eng.forward(x.cuda())
            eng.backward(target=target.cuda())
            torch.cuda.synchronize()
    # every packed copy the engine keeps (16-bit MFMA operands, transposed / flipped / tap-major / padded re-layouts, fp32 bias and
    # affine copies) was refreshed by the step: the engine now computes exactly what a fresh engine loaded from the updated masters does
    sd_new = {k: v.detach().clone().cpu() for k, v in eng.bound.items()}
    out_a = eng.forward(x.cuda()).clone()
    loss_a = eng.backward(target=target.cuda(), ignore_index=-1)
    torch.cuda.synchronize()
    grads_a = {k: v.clone() for k, v in eng.grads.items()}
    eng_b, out_b, loss_b, _ = _engine_step(cfg, sd_new, x, target, tok)
    # (the BatchNorm batch sums are fp32 atomics: the two runs may differ in a last bit that a bf16 rounding then amplifies, so the
    # yardstick is what the two optimizer steps did to the output -- a pack left behind would be off by that much)
    moved = (out_a - out0).abs().max().item()
    assert (out_a - out_b).abs().max().item() <= 0.02 * moved, ((out_a - out_b).abs().max().item(), moved)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-4 * abs(float(loss_b))
    # gradients (only the backward reads the W^T / flipped-conv packs): the two engines agree far inside what the two optimizer steps
    # did to each gradient; what is left is the forward's last-bit noise (atomics in the BatchNorm sums) through bf16 ReLU masks
    report = {k: (rel(grads_a[k], eng_b.grads[k]), rel(grads_a[k], grads0[k])) for k in keys}
    bad = {k: v for k, v in report.items() if v[0] > max(2e-2, 0.1 * v[1])}
    assert not bad, bad
    # ... and the eval-only packs (BatchNorm folded into the convs, the commuted head) catch up at the next eval forward
    eng.set_train(False); eng_b.set_train(False)
    ev_a, ev_b = eng.forward(x.cuda()), eng_b.forward(x.cuda())
    assert (ev_a - ev_b).abs().max().item() <= 0.02 * moved, ((ev_a - ev_b).abs().max().item(), moved)


def test_lsegnet_train_mode_backpropagates_through_the_engine():
    """The drop-in class under autograd, the way Lightning drives LSegmentationModule.training_step (:66-81): net.train();
    out = net(img); loss = criterion(out, target); loss.backward() -> parameter .grad tensors; torch.optim.SGD steps them."""
    import warnings
    warnings.simplefilter("ignore")
    import torch.nn.functional as F
    from modules.models.lseg_net import LSegNet
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=8)
    labels = read_labels(MG.LABELS)[:5]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(2, 64, 64, seed=8)
    target = _target(2, 64, 64, 5, 8)
    ref_loss, ref_grads = training_step(sd, x, target, tok, cfg, ignore_index=-1)
    net = LSegNet(labels=labels, backbone="tiny16", features=cfg.features, arch_option=0, block_depth=0, activation="lrelu")
    net.load_state_dict(sd)
    net = net.cuda().train()
    out = net(x.cuda())
    assert out.requires_grad
    loss = F.cross_entropy(out, target.cuda(), ignore_index=-1)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))
    named = dict(net.named_parameters())
    for k in ("scratch.head1.weight", "pretrained.model.blocks.2.mlp.fc1.weight", "pretrained.act_postprocess1.4.weight",
              "scratch.refinenet3.resConfUnit1.bn2.weight", "pretrained.model.patch_embed.proj.weight"):
        assert named[k].grad is not None and rel(named[k].grad.cpu(), ref_grads[k]) <= 0.45, k
    assert named["pretrained.model.norm.weight"].grad is None            # dead in the forward (lseg_vit.py:108)
    # an optimizer step on the masters is picked up by the next forward; eval mode still works afterwards
    opt = torch.optim.SGD([p for p in net.parameters() if p.grad is not None], lr=0.05)
    opt.step()
    out2 = net(x.cuda())
    assert not torch.equal(out2.detach(), out.detach())
    net.eval()
    with torch.no_grad():
        ev = net(x.cuda())
    assert ev.shape == out.shape and torch.isfinite(ev).all()


def test_gelu_epilogue_fusions_equal_the_separate_passes(tmp_path):
    """The training step's MLP GELU lives in GEMM epilogues (fc1 stores z and gelu(z); the fc2 dX GEMM multiplies by gelu'(z): gemm.hip
    EPI_LIN16_GELU2 / EPI_LIN16_DGELU).  Same loss, logits and gradients as the GEMMs followed by the streaming GELU / GELU' kernels
    (LSEG_NO_GELU_FUSE=1, read once per process -> a second interpreter), up to the one bf16 rounding of d(mlp) the fused form skips
    and the erf polynomial of the epilogue (|err| <= 1.5e-7) against erff."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from lseg_hip.config import get_config; from lseg_hip.engine import HipEngine\n"
        "from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels\n"
        "cfg = get_config('tiny16'); sd = {k: v.cuda() for k, v in synthetic_state_dict(cfg, seed=3).items()}\n"
        "tok = synthetic_tokens(read_labels(%r)[:5], cfg.text.vocab, cfg.text.ctx)\n"
        "eng = HipEngine(cfg, 96, 64, max_batch=2, max_labels=5); eng.load_state_dict(sd); eng.set_tokens(tok); eng.enable_training(sd)\n"
        "x = synthetic_images(2, 96, 64, seed=4).cuda()\n"
        "g = torch.Generator().manual_seed(9); t = torch.randint(0, 5, (2, 96, 64), generator=g); t[torch.rand((2, 96, 64), generator=g) < 0.2] = -1\n"
        "out = eng.forward(x).clone(); loss = eng.backward(target=t.cuda(), ignore_index=-1); torch.cuda.synchronize()\n"
        "torch.save({'out': out.cpu(), 'loss': float(loss), 'grads': {k: v.float().cpu() for k, v in eng.grads.items()}}, sys.argv[1])\n"
    ) % (os.path.join(root, "lang-seg_amd"), root, MG.LABELS)
    res = {}
    for tag, off in (("fused", False), ("separate", True)):
        env = dict(os.environ)
        env.pop("LSEG_NO_GELU_FUSE", None)
        if off:
            env["LSEG_NO_GELU_FUSE"] = "1"
        out = str(tmp_path / (tag + ".pt"))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
        res[tag] = torch.load(out)
    a, b = res["fused"], res["separate"]
    # a flipped bf16 rounding of one activation is amplified by the layers above it: compare at the scale of the tensors (a wrong
    # column / row mapping in an epilogue would be an O(1) error everywhere)
    d_out = (a["out"] - b["out"]).abs().max().item() / b["out"].abs().max().item()
    d_loss = abs(a["loss"] - b["loss"]) / abs(b["loss"])
    worst = {k: rel(a["grads"][k], b["grads"][k]) for k in b["grads"]}
    med = sorted(worst.values())[len(worst) // 2]
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    mlp = [k for k in worst if ".mlp." in k or "project" in k]
    assert mlp and any(not torch.equal(a["grads"][k], b["grads"][k]) for k in mlp)        # the two interpreters really ran different kernels
    # measured: logits 0.8 % of their range, loss 8e-5, gradients 3.1 % median / 6.7 % worst -- the same ReLU-sign amplification on the
    # seeded random network that separates the engine from the fp32 oracle (module docstring); a broken epilogue is >= 50 %
    assert d_out <= 2e-2 and d_loss <= 3e-3 and max(worst.values()) <= 0.15 and med <= 6e-2, (d_out, d_loss, med, top)

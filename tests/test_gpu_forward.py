"""Whole-path parity: HIP engine (through the C ABI) vs the CPU oracle on the same seeded
synthetic weights/inputs, stage by stage (the four hooked activations, reassembled maps,
refinenet paths, pixel features, text features, low-res logits, final logits) and against
the committed golden fixtures.

Tolerances (stated, per BASELINE north_star "within a stated fp tolerance"); measured values in
DESIGN.md "Parity":
  * text tower (fp16 like the reference): |d| <= 4e-3 on unit-norm features
  * image tower, bf16 MFMA operands / fp32 accumulate vs the fp32 reference on the seeded
    random-weight network (an error amplifier: boosted qkv scale, peaky 901-key softmax):
    relative RMS error <= 10% per stage (measured 5.9-8.6% ViT-L, 0.7-5% tiny),
    logits |d| <= 0.30 (measured 0.04-0.22) on logits that live in [-14.3, 14.3]; x the head blocks' own gain for arch_option 1/2
  * same with fp16 operands (8x finer mantissa, same MFMA rate): <= 1.5% per stage, |d| <= 0.06
  * argmax masks: every mismatching pixel must have an oracle top-2 margin below twice the
    measured logit error (bit-parity is only meaningful where the reference itself is decisive)
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from lseg_hip.config import get_config                      # noqa: E402
from lseg_hip.engine import HipEngine                       # noqa: E402
from lseg_hip.synth import (synthetic_state_dict, fixture_state_dict, outlier_state_dict, synthetic_tokens, synthetic_images,   # noqa: E402
                            read_labels)
from oracle.lseg_oracle import lseg_forward                  # noqa: E402
from oracle import make_golden as MG                        # noqa: E402

LOGIT_TOL = 0.30          # bf16 operands; measured 0.04-0.22 (the largest at ViT-L/16, 480x480).  fp16: 0.06 (measured <= 0.03)


def head_block_gain(cfg, sd):
    """Worst-case amplification of a logit error by the arch_option 1/2 head blocks (lseg_net.py:43-79): each applies ONE shared 3x3
    filter per label plane (+ the channel max for the bottleneck) and a 1-Lipschitz activation -> sum|w| (+1) per block."""
    if cfg.arch_option == 0:
        return 1.0
    per_block = sd["scratch.head_block.depthwise.depthwise.weight"].abs().sum().item() + (1.0 if cfg.arch_option == 1 else 0.0)
    return max(1.0, per_block) ** cfg.block_depth


def relrms(a, b):
    return ((a - b).float().pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt().clamp_min(1e-12)).item()


def run_engine(spec, image_dtype="bf16", debug=True):
    bb, H, W, B, K, seed, arch, depth = spec
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = synthetic_state_dict(cfg, seed=seed)
    labels = read_labels(MG.LABELS)[:K]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(B, H, W, seed=seed)
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K, image_dtype=image_dtype)
    eng.load_state_dict(sd)
    eng.set_tokens(tok)
    eng.set_debug(debug)
    logits = eng.forward(x.cuda())                                              # logits only: the call every caller of the class makes
    amax = eng.forward(x.cuda(), want_logits=False, want_argmax=True)           # masks only
    torch.cuda.synchronize()
    return cfg, sd, tok, x, eng, logits, amax


def check_case(spec, image_dtype="bf16", stage_tol=0.10):
    cfg, sd, tok, x, eng, logits, amax = run_engine(spec, image_dtype)
    with torch.no_grad():
        ref, inter = lseg_forward(sd, x, tok, cfg, return_intermediates=True)
    B, _, H, W = x.shape
    report = {}
    # text features (fp16 path, same arithmetic as the reference)
    tf = eng.encode_text().float().cpu()
    tref = inter["text_features"]
    tref = (tref / tref.norm(dim=-1, keepdim=True).half().float()).half().float()
    report["text"] = (tf - tref).abs().max().item()
    assert report["text"] <= 4e-3, report
    ntok = cfg.tokens(H, W)
    for l in range(4):
        a = eng.intermediate(f"act{l + 1}", (B, ntok, cfg.dim)).cpu()
        report[f"act{l + 1}"] = relrms(a, inter["acts"][l])
        lay = inter["layers"][l]
        got = eng.intermediate(f"layer{l + 1}", lay.shape).cpu()
        report[f"layer{l + 1}"] = relrms(got, lay)
        rn = inter["rn"][l]
        report[f"rn{l + 1}"] = relrms(eng.intermediate(f"rn{l + 1}", rn.shape).cpu(), rn)
        p = inter["paths"][l]
        report[f"path{l + 1}"] = relrms(eng.intermediate(f"path{l + 1}", p.shape).cpu(), p)
    imf = inter["image_features"]
    report["image_features"] = relrms(eng.intermediate("image_features", imf.shape).cpu(), imf)
    low = eng.intermediate("lowres", inter["lowres"].shape).cpu()
    report["lowres_maxabs"] = (low - inter["lowres"]).abs().max().item()
    out = logits.cpu()
    report["logits_maxabs"] = (out - ref).abs().max().item()
    report["logits_relrms"] = relrms(out, ref)
    print("PARITY", spec, image_dtype, {k: round(v, 5) for k, v in report.items()})
    for k, v in report.items():
        if k.startswith(("act", "layer", "rn", "path", "image_features")):
            assert v <= stage_tol, (k, report)
    if cfg.arch_option == 0:
        assert report["lowres_maxabs"] <= LOGIT_TOL, report
    assert report["logits_maxabs"] <= LOGIT_TOL * head_block_gain(cfg, sd), report
    # argmax: mismatches only where the oracle's own top-2 margin is inside the tolerance
    if cfg.arch_option == 0:
        lo = ref                                   # the masks are argmax of the OUTPUT logits (engine: through the x2 bilinear on the fly)
        ref_am = lo.argmax(1)
        mism = amax.cpu().long() != ref_am
        top2 = lo.topk(2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
        frac = mism.float().mean().item()
        print("ARGMAX mismatch fraction", frac, "max margin at mismatch",
              margin[mism].max().item() if mism.any() else 0.0)
        if mism.any():
            assert margin[mism].max().item() <= 2 * report["lowres_maxabs"] + 1e-6
    return report


@pytest.mark.gpu_fast
@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_tiny_forward_matches_oracle(name):
    check_case(MG.CASES[name])


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_tiny_forward_matches_golden(name, golden_dir):
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    cfg, sd, tok, x, eng, logits, amax = run_engine(MG.CASES[name], debug=False)
    tol = LOGIT_TOL * head_block_gain(cfg, sd)
    assert (logits.cpu() - g["logits"]).abs().max().item() <= tol


def test_tiny_forward_fp16_image_tower():
    """fp16 MFMA operands (10-bit mantissa) track the fp32 reference ~8x closer than bf16."""
    rep = check_case(MG.CASES["tiny16_64x64_k5"], image_dtype="fp16", stage_tol=0.015)
    assert rep["logits_maxabs"] <= 0.06


def test_vitl16_480_k150_matches_oracle_and_golden(golden_dir):
    """BASELINE.json configs[1] at B=1: full-size ViT-L/16, 480x480, K=150."""
    spec = MG.FULL["vitl16_480_k150"]
    rep = check_case(spec)
    g = torch.load(os.path.join(golden_dir, "vitl16_480_k150.pt"))
    cfg, sd, tok, x, eng, logits, amax = run_engine(spec, debug=False)
    assert (logits.cpu()[:, :, ::16, ::16] - g["logits_sub16"]).abs().max().item() <= LOGIT_TOL
    tf = eng.encode_text().float().cpu()
    gt = g["text_features"].float()
    gt = (gt / gt.norm(dim=-1, keepdim=True).half().float()).half().float()
    assert (tf - gt).abs().max().item() <= 4e-3


def test_vitb32_480_k7_matches_oracle_and_golden(golden_dir):
    """BASELINE.json configs[0] (the reference's CPU-runnable plumbing case): ViT-B/32, 480x480, K=7.
    Level-1 reassemble has 96 channels: the engine pads it to 128 zero-weight channels."""
    spec = MG.FULL["vitb32_480_k7"]
    rep = check_case(spec, stage_tol=0.15)        # this random 12-layer net amplifies bf16 rounding to 2.6-12.7 %
    rep16 = check_case(spec, image_dtype="fp16", stage_tol=0.03)
    assert rep16["logits_maxabs"] <= 0.06
    g = torch.load(os.path.join(golden_dir, "vitb32_480_k7.pt"))
    cfg, sd, tok, x, eng, logits, amax = run_engine(spec, debug=False)
    assert (logits.cpu()[:, :, ::16, ::16] - g["logits_sub16"]).abs().max().item() <= LOGIT_TOL


def test_vitl16_fp16_operands_track_the_reference_8x_closer():
    rep = check_case(MG.FULL["vitl16_480_k150"], image_dtype="fp16", stage_tol=0.015)
    assert rep["logits_maxabs"] <= 0.06


def test_lsegnet_module_forward_is_the_engine(golden_dir):
    """The drop-in class API (modules.models.lseg_net.LSegNet, same ctor/forward as the reference)
    produces exactly what the engine produces, reloads after load_state_dict, and handles
    `labelset` (evaluate_random path, lsegmentation_module.py:54-63)."""
    import warnings
    warnings.simplefilter("ignore")
    from modules.models.lseg_net import LSegNet
    spec = MG.CASES["tiny16_64x64_k5"]
    cfg, sd, tok, x, eng, logits, amax = run_engine(spec, debug=False)
    labels = read_labels(MG.LABELS)[:5]
    net = LSegNet(labels=labels, backbone="tiny16", features=cfg.features, arch_option=0, block_depth=0,
                  activation="lrelu", image_dtype="bf16")          # the same operand type as run_engine's engine (the class default is fp16)
    net.load_state_dict(sd)
    net = net.eval().cuda()
    with torch.no_grad():
        out = net(x.cuda())
        assert out.dtype == torch.float32 and out.shape == logits.shape
        assert torch.equal(out, logits)
        out += 1.0                                     # callers mutate the result in place (encoding_models.py:138)
        out2 = net(x.cuda())
        assert torch.equal(out2, logits)
        sub = net(x.cuda(), labelset=labels[:3])
        assert sub.shape == (x.shape[0], 3, 64, 64)
        assert torch.allclose(sub, logits[:, :3], atol=1e-6)       # logits of a label do not depend on the others
        # new weights are picked up
        sd2 = synthetic_state_dict(cfg, seed=123)
        net.load_state_dict(sd2)
        assert not torch.equal(net(x.cuda()), logits)
    g = torch.load(os.path.join(golden_dir, "tiny16_64x64_k5.pt"))
    assert (logits.cpu() - g["logits"]).abs().max().item() <= LOGIT_TOL


def test_batch_entries_are_independent():
    """Embarrassingly parallel over images (config 3): image i of a batch == the same image alone."""
    spec = list(MG.CASES["tiny16_64x64_k5"])
    spec[3] = 3
    cfg, sd, tok, x, eng, logits, amax = run_engine(tuple(spec), debug=False)
    one = eng.forward(x[1:2].cuda())
    torch.cuda.synchronize()
    assert torch.equal(one[0], logits[1])


def test_text_cache_is_exact():
    cfg, sd, tok, x, eng, logits, amax = run_engine(MG.CASES["tiny16_64x64_k5"], debug=False)
    eng.set_text_cache(True)
    again = eng.forward(x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(again, logits)


@pytest.mark.gpu_fast
def test_errors_are_loud():
    from lseg_hip import _lib
    cfg = get_config("tiny16")
    eng = HipEngine(cfg, 64, 64, max_batch=1, max_labels=4)
    with pytest.raises(_lib.LSegError):      # nothing bound / no tokens
        eng.forward(torch.zeros((1, 3, 64, 64)).cuda())
    sd = synthetic_state_dict(cfg)
    del sd["scratch.head1.weight"]
    with pytest.raises(_lib.LSegError) as ei:
        eng.load_state_dict(sd)
    assert "scratch.head1.weight" in str(ei.value)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros((1, 3, 32, 32)).cuda())


def test_text_truncation_is_bit_exact():
    """Causal truncation to max(EOT)+1 positions == the reference's full 77-position schedule."""
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=9)
    tok = synthetic_tokens(["wall", "potted plant", "a b c d e f g", "sky"], cfg.text.vocab, cfg.text.ctx)
    feats = []
    for full in (False, True):
        eng = HipEngine(cfg, 64, 64, max_batch=1, max_labels=4, full_text_context=full)
        eng.load_state_dict(sd)
        eng.set_tokens(tok)
        feats.append(eng.encode_text().clone())
        eng.close()
    assert torch.equal(feats[0], feats[1])


def test_zero_shot_network_matches_oracle_and_golden(golden_dir):
    """LSegNetZS.forward(x, class_info) (modules/models/lseg_net_zs.py:177-214) through the drop-in class:
    every image is scored against its own ['others', class] pair; equals the oracle's restatement and the
    committed golden, and equals the shared-label engine run image by image."""
    import warnings
    warnings.simplefilter("ignore")
    from modules.models.lseg_net_zs import LSegNetZS
    name = "tiny16_64x64_zs"
    spec = MG.ZS_CASES[name]
    cfg, sd, x, tok, ref, inter = MG.run_zs_case(spec)
    names = read_labels(MG.FSS)
    net = LSegNetZS(label_list=names[:16], backbone="tiny16", features=cfg.features, arch_option=0, block_depth=0,
                    activation="lrelu")
    net.load_state_dict(sd)
    net = net.eval().cuda()
    class_info = torch.tensor(spec[3])
    with torch.no_grad():
        out = net(x.cuda(), class_info)
    assert out.shape == (len(spec[3]), 2, 64, 64) and out.dtype == torch.float32
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    assert (out.cpu() - ref).abs().max().item() <= LOGIT_TOL
    assert (out.cpu() - g["logits"]).abs().max().item() <= LOGIT_TOL
    # same numbers as the shared-label path run on each image alone with its pair
    eng = HipEngine(cfg, 64, 64, max_batch=1, max_labels=2, image_dtype=net.image_dtype)      # the class default (fp16), not HipEngine's (bf16)
    eng.load_state_dict(sd)
    for b in range(x.shape[0]):
        eng.set_tokens(tok[2 * b:2 * b + 2])
        single = eng.forward(x[b:b + 1].cuda())
        assert torch.equal(single[0], out[b])
    # a wrong number of class ids is an error, not a silent broadcast
    with pytest.raises(ValueError):
        net(x.cuda(), class_info[:1])


def test_vitl16_batch_of_8_equals_single_image_runs():
    """The tile configuration follows the problem size (256x256 tiles once M = B*901 is large), so a batch must be
    checked against the single-image runs that the oracle comparison above validates: every GEMM accumulates K in the
    same order whatever the tile, so the logits have to agree to fp32 round-off."""
    spec = MG.FULL["vitl16_480_k150"]
    bb, H, W, _, K, seed, arch, depth = spec
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = synthetic_state_dict(cfg, seed=seed)
    tok = synthetic_tokens(read_labels(MG.LABELS)[:K], cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(8, H, W, seed=seed + 7).cuda()
    eng = HipEngine(cfg, H, W, max_batch=8, max_labels=K, batch_invariant=True)     # (default engines split K at B = 1: other rounding)
    eng.load_state_dict(sd)
    eng.set_tokens(tok)
    batch = eng.forward(x, want_logits=False, want_argmax=True)
    assert batch.shape == (8, H, W)
    low8 = eng.intermediate("lowres", (8, K, H // 2, W // 2))
    for b in (0, 5):
        am1 = eng.forward(x[b:b + 1], want_logits=False, want_argmax=True)
        low1 = eng.intermediate("lowres", (1, K, H // 2, W // 2))
        assert (low1[0] - low8[b]).abs().max().item() <= 1e-3
        assert (am1[0] != batch[b]).float().mean().item() <= 1e-4


_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# forward fixtures made by the reference's code; ref_train_* (loss + gradients) and ref_eval_* (evaluator) have their own tests
_REF = sorted(f[:-3] for f in os.listdir(_GOLD) if f.startswith("ref_vit"))
_REF_FULL = sorted(f[:-3] for f in os.listdir(_GOLD) if f.startswith("ref_full_") and not f.endswith("_out480.pt"))

# measured |dlogit| of the bf16 engine vs these fixtures is 0.08-0.16 (fp16 operands: 0.01-0.03); stated tolerance ~2x that
# "strict" = the split-precision validation mode ((hi, lo) fp16 operand pairs): what is left is fp32 summation order and the
# reference's own fp16 rounding of the logits (ulp 0.002-0.004 at |logit| 2-8); two fp32 CPU implementations of the path (oracle vs
# reference) already differ by 0.002-0.004 (tests/test_oracle_ref_golden.py)
REF_TOL = {"bf16": 0.30, "fp16": 0.06, "strict": 0.012}
STAGE_TOL = {"bf16": 0.10, "fp16": 0.015, "strict": 2e-3}
# the realistic-statistics fixture (`*_outlier`: residual outlier channels ~1e3 next to O(1) ones, LayerNorm gains of 10, x10 BatchNorm
# scales; oracle/make_ref_golden.py): the 16-bit towers lose 1.5-2.5x more per stage and per logit there -- measured, lease B of round 5
# (profiles/r05_parity_table.txt): hooked activations rel. rms bf16 0.136 / fp16 0.016, max |dlogit| bf16 0.200 / fp16 0.064 / strict 0.0049
REF_TOL_BY_NAME = {"ref_full_vitl16_480x480_k150_outlier": {"bf16": 0.35, "fp16": 0.11, "strict": 0.012}}
# (hooked activations, fp16 operands: act1 0.016, act2 0.041, act3 0.060 -- the error of the three x1000 outlier channels dominates the rms and
# grows with depth; bf16: act2 0.136, all four <= 0.22; leases B-D of round 5.  Bars ~2x the largest measured.)
STAGE_TOL_BY_NAME = {"ref_full_vitl16_480x480_k150_outlier": {"bf16": 0.30, "fp16": 0.12, "strict": 2e-3}}


def assert_argmax_mismatches_are_ties(out_low_or_logits, ref_argmax, ref_margin, err, what):
    """Arg-max parity statement: every pixel where the engine's mask differs from the reference's must be a pixel where the
    REFERENCE's own top-2 margin is below twice the measured logit error (i.e. the reference is not decisive there)."""
    mism = out_low_or_logits.argmax(1) != ref_argmax
    frac = mism.float().mean().item()
    worst = ref_margin[mism].max().item() if mism.any() else 0.0
    print(f"{what}: argmax mismatch fraction {frac:.5f}, max reference margin at a mismatch {worst:.4f}, max|dlogit| {err:.4f}")
    assert worst <= 2 * err + 1e-6, (what, frac, worst, err)
    return frac


@pytest.mark.gpu_fast
@pytest.mark.parametrize("dtype", ["bf16", "fp16", "strict"])
@pytest.mark.parametrize("name", _REF)
def test_engine_matches_fixtures_made_by_the_reference_code(name, dtype, golden_dir):
    """tests/golden/ref_vit*.pt were produced by the reference's own modules/models/lseg_net(_zs).py on CPU
    (oracle/make_ref_golden.py).  Full ViT-L/16 and ViT-B/32 dimensions on small images, arch_option 0/1/2, B=2, ZS.
    16-bit MFMA operands vs the reference's fp32 tower: |dlogit| <= REF_TOL on the cosine logits (scaled by the head blocks'
    gain for arch_option 1/2); every arg-max mismatch sits where the reference's own top-2 margin is < 2x the measured error."""
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    zs = name.endswith("_zs")
    if zs:
        bb, H, W, class_info, seed = g["spec"]
        B, arch, depth, k = len(class_info), 0, 0, 2
    else:
        bb, H, W, B, K, seed, arch, depth = g["spec"]
        k = 0
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = synthetic_state_dict(cfg, seed=seed)
    x = synthetic_images(B, H, W, seed=seed)
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=g["tokens"].shape[0], image_dtype=dtype)
    eng.load_state_dict(sd)
    eng.set_tokens(g["tokens"], labels_per_image=k)
    out = eng.forward(x.cuda()).cpu()
    sub = int(g.get("sub", 1))                 # ref_vitl16_480x480_zs (BASELINE configs[4]'s own size): every second pixel, fp16
    ref = g["logits"].float()
    out = out[:, :, ::sub, ::sub]
    assert out.shape == ref.shape
    # head blocks (arch_option 1/2, lseg_net.py:43-79): each applies ONE shared 3x3 filter per label plane (+ the channel max
    # for the bottleneck) and a 1-Lipschitz activation, so a block amplifies an input error by at most sum|w| (+1): the
    # tolerance follows that bound, computed from the actual filter
    gain = 1.0
    if arch != 0:
        per_block = sd["scratch.head_block.depthwise.depthwise.weight"].abs().sum().item() + (1.0 if arch == 1 else 0.0)
        gain = max(1.0, per_block) ** depth
    err = (out - ref).abs().max().item()
    print(f"{name}[{dtype}]: max|dlogit| {err:.4f} (logit range {ref.abs().max().item():.2f}, head-block gain {gain:.2f})")
    assert err <= REF_TOL[dtype] * gain, (name, dtype, err)
    top2 = ref.topk(2, dim=1).values
    assert_argmax_mismatches_are_ties(out, ref.argmax(1), top2[:, 0] - top2[:, 1], err, f"{name}[{dtype}]")
    if "text_features" in g:
        tf = eng.encode_text().float().cpu()
        tr = g["text_features"].float()
        tr = tr / tr.norm(dim=-1, keepdim=True)
        assert (tf - tr).abs().max().item() <= 4e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "strict"])
@pytest.mark.parametrize("name", _REF_FULL)
def test_engine_matches_the_reference_at_the_baseline_configs(name, dtype, golden_dir):
    """BASELINE.json configs[1] (ViT-L/16, 480x480, K=150) and configs[4] (K=1000 open-vocabulary prompts), B=1: fixtures made by
    running the reference's own LSegNet.forward (modules/models/lseg_net.py:160-205) at full size on CPU
    (oracle/make_ref_golden.py --full).  Checks the hooked activations, path_1, the text features, the low-resolution logits
    (sub-sampled), the output logits (sub-sampled) and the FULL 240x240 arg-max mask against the reference's mask + margins."""
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    bb, H, W, B, K, seed, arch, depth = g["spec"]
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    sd = fixture_state_dict(cfg, seed, g)          # `*_outlier` fixtures: realistic-statistics weights (outlier_state_dict)
    x = synthetic_images(B, H, W, seed=seed)
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K, image_dtype=dtype)
    eng.load_state_dict(sd)
    eng.set_tokens(g["tokens"])
    eng.set_debug(True)
    out = eng.forward(x.cuda())
    torch.cuda.synchronize()
    stage_tol = STAGE_TOL_BY_NAME.get(name, STAGE_TOL)[dtype]
    ref_tol = REF_TOL_BY_NAME.get(name, REF_TOL)[dtype]
    ntok = cfg.tokens(H, W)
    for l in range(4):
        a = eng.intermediate(f"act{l + 1}", (B, ntok, cfg.dim)).cpu()[:, ::8, :]
        r = relrms(a, g["acts_sub"][l].float())
        print(f"{name}[{dtype}] act{l + 1}: relative rms error {r:.5f} (bar {stage_tol})")
        assert r <= stage_tol, (f"act{l + 1}", r)
    p1 = eng.intermediate("path1", (B, cfg.features, H // 2, W // 2)).cpu()[:, :, ::8, ::8]
    assert relrms(p1, g["path_1_sub8"].float()) <= stage_tol
    # the taps above come from the debug schedule (two 1x1 GEMMs at 240x240); the numbers below from the production schedule
    # (commuted head, fused upsample + normalise), like bench.py runs it
    eng.set_debug(False)
    out = eng.forward(x.cuda())
    torch.cuda.synchronize()
    low = eng.intermediate("lowres", (B, K, H // 2, W // 2)).cpu()
    err = (low[:, :, ::8, ::8] - g["lowres_sub8"].float()).abs().max().item()
    st = g["logits_sub_step"]
    err_out = (out.cpu()[:, :, ::st, ::st] - g["logits_sub"]).abs().max().item()
    # the engine's logits at the reference's two best labels of EVERY pixel (the decision surface of the mask)
    err_top2 = (low.gather(1, g["top2_idx"].long()) - g["top2_val"].float()).abs().max().item()
    err = max(err, err_top2)
    print(f"{name}[{dtype}]: lowres max|d| {err:.4f}, logits max|d| {err_out:.4f} (range {g['lowres_absmax']:.2f})")
    assert err <= ref_tol and err_out <= ref_tol, (name, dtype, err, err_out)
    frac = assert_argmax_mismatches_are_ties(low, g["argmax_lowres"].long(), g["margin_lowres"].float(), err, f"{name}[{dtype}]")
    out_dir = os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out")
    if os.path.isdir(out_dir):                     # the parity table of DESIGN.md §4 is made from these lines
        mism = low.argmax(1) != g["argmax_lowres"].long()
        worst = g["margin_lowres"].float()[mism].max().item() if mism.any() else 0.0
        with open(os.path.join(out_dir, "parity_table.txt"), "a") as f:
            fl, ftxt = _floor(name, "oracle_vs_reference_240")
            mult = f" = {frac / fl:.1f} x floor" if fl else ""
            f.write(f"{name} {dtype}: max|dlogit| {err:.5f}  argmax mismatch fraction {frac:.6f}  max reference margin at a mismatch {worst:.5f}{ftxt}{mult}\n")
    tf = eng.encode_text().float().cpu()
    tr = g["text_features"].float()
    tr = tr / tr.norm(dim=-1, keepdim=True)
    assert (tf - tr).abs().max().item() <= 4e-3



def _floor(name, key):
    """the fp32-vs-fp32 noise floor of the mask (tools/parity_floor.py -> profiles/r06_parity_floor.json): argmax flips of the fp32 oracle against
    the fp32 reference on the same fixture with the same text features.  (value, text for the table); (None, "") when not recorded"""
    import json
    try:
        fl = json.load(open(os.path.join(os.path.dirname(_GOLD), "..", "profiles", "r06_parity_floor.json")))[name][key]["argmax_mismatch_frac"]
    except (OSError, KeyError, ValueError):
        return None, ""
    return fl, f"  fp32-vs-fp32 floor {fl:.6f}"


# absolute bars of the 480 x 480 mask test (the "ties" statement above scales with the measured error; these do not): fraction of the
# output pixels whose label may differ from the reference's, and the largest reference top-2 margin at such a pixel.  Measured at
# 240 x 240 in round 3: K = 150 bf16 1.62 % / 0.099, fp16 0.27 % / 0.0092, strict 0.16 % / 0.0024; K = 1000 10.8 % / 0.101, 1.85 % / 0.013,
# 0.91 % / 0.0039 (1000 synthetic prompts on a random text tower: median top-2 margin 0.055).  Measured at 480 x 480 in round 4: K = 150 bf16
# 2.1-2.4 % (the flip fraction of the bf16 tower moves by +-0.4 % with any change of rounding points: random weights amplify), fp16 0.22 %,
# strict 0.12 %; K = 1000 10.9 % / 1.6-1.8 % / 0.95 %
MASK480_CAPS = {150: {"bf16": (0.035, 0.15), "fp16": (0.005, 0.02), "strict": (0.003, 0.006)},
                1000: {"bf16": (0.14, 0.15), "fp16": (0.026, 0.025), "strict": (0.013, 0.008)}}


# the realistic-statistics fixture (residual outliers ~1e3 next to O(1) channels; the reference's median top-2 margin there is 0.45, so a
# flip needs a larger logit error than on the N(0, 0.02) net): measured r5 -- see profiles/r05_parity_table.txt
# (bf16 0.164 % / 0.070, fp16 0.056 % / 0.021, strict 0.015 % / 0.0021: fewer flips than on the N(0, 0.02) net at LARGER logit errors)
MASK480_CAPS_BY_NAME = {"ref_full_vitl16_480x480_k150_outlier": {"bf16": (0.004, 0.12), "fp16": (0.0015, 0.04), "strict": (0.0005, 0.005)}}


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "strict"])
@pytest.mark.parametrize("name", _REF_FULL)
def test_engine_masks_match_the_reference_at_480x480(name, dtype, golden_dir):
    """The reference upsamples BEFORE anyone takes an arg-max (lseg_net.py:203): the decision surface that counts is the 480 x 480 one.
    `<name>_out480.pt` holds the reference's own label and top-2 margin for every output pixel (oracle/make_ref_golden.py --full480).
    Held to it: the arg-max of the engine's materialised [1,K,480,480] logits, and (K <= 256) the uint8 mask the engine produces WITHOUT
    the logits, through the x2 bilinear on the fly (lseg_forward masks) -- which must equal the arg-max of its own logits except at
    exact fp32 ties."""
    path = os.path.join(golden_dir, name + "_out480.pt")
    if not os.path.exists(path):
        pytest.skip("no 480x480 side fixture")
    g = torch.load(path)
    bb, H, W, B, K, seed, arch, depth = g["spec"]
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K, image_dtype=dtype)
    eng.load_state_dict(fixture_state_dict(cfg, seed, g))
    eng.set_tokens(g["tokens"])
    x = synthetic_images(B, H, W, seed=seed).cuda()
    out = eng.forward(x)
    torch.cuda.synchronize()
    ref_am, ref_margin = g["argmax"].long().cuda(), g["margin"].float().cuda()
    am = out.argmax(1)
    # logit error at the reference's decision: the engine's value at the reference's best label vs the reference's own
    err = (out.gather(1, ref_am.unsqueeze(1)).squeeze(1) - g["top1_val"].float().cuda()).abs().max().item()
    mism = am != ref_am
    frac = mism.float().mean().item()
    worst = ref_margin[mism].max().item() if mism.any() else 0.0
    cap_frac, cap_margin = MASK480_CAPS_BY_NAME.get(name, MASK480_CAPS[K])[dtype]
    print(f"{name}[{dtype}] 480x480: argmax mismatch fraction {frac:.5f} (cap {cap_frac}), max reference margin at a mismatch {worst:.4f} "
          f"(cap {cap_margin}), max|dlogit| at the reference's label {err:.4f}")
    out_dir = os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_table.txt"), "a") as f:
            fl, ftxt = _floor(name, "oracle_vs_reference_480")
            mult = f" = {frac / fl:.1f} x floor" if fl else ""
            f.write(f"{name} {dtype} 480x480: max|dlogit| {err:.5f}  argmax mismatch fraction {frac:.6f}  max reference margin at a mismatch {worst:.5f}{ftxt}{mult}\n")
    assert err <= REF_TOL_BY_NAME.get(name, REF_TOL)[dtype], (name, dtype, err)
    assert worst <= 2 * err + 1e-6 and worst <= cap_margin and frac <= cap_frac, (name, dtype, frac, worst, err)
    if K <= 256:
        only = eng.forward(x, want_logits=False, want_argmax=True)
        top2 = out.topk(2, dim=1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 1e-6
        assert torch.equal(only.long()[decisive], am[decisive])
        m2 = only.long() != ref_am
        assert m2.float().mean().item() <= cap_frac and (ref_margin[m2].max().item() if m2.any() else 0.0) <= cap_margin
    eng.close()


_outlier_state_dict = outlier_state_dict            # (moved to lseg_hip.synth: the reference-run `*_outlier` fixture is made from it too)


def test_overflow_sentinel_reports_a_later_image_without_synchronising():
    """The always-on companion of the range check (lseg_overflow_seen, round 6): the first-forward scan of a weight set cannot see an
    INPUT-dependent non-finite activation on a later image.  Here batch A keeps the fp16 tower finite and batch B carries a NaN pixel (the
    cheapest stand-in for an activation that leaves the fp16 range on one input only): the sentinel must stay clear after A, be raised once
    B's forward has completed, be sticky until reset; the drop-in network reports it on the NEXT call, loudly, and re-scans the ranges."""
    import warnings
    from modules.models.lseg_net import LSegNet
    cfg = get_config("tiny16")
    labels = read_labels(MG.LABELS)[:5]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    sd = _outlier_state_dict(cfg, 5, 1e2)
    xa = synthetic_images(2, 64, 64, seed=3).cuda()
    eng = HipEngine(cfg, 64, 64, max_batch=2, max_labels=5, image_dtype="fp16")
    eng.load_state_dict(sd)
    eng.set_tokens(tok)
    out = eng.forward(xa)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and not eng.overflow_seen()
    # poison the input instead of hunting for a scale: a NaN pixel is the cheapest "non-finite activation on a later image"
    xb = xa.clone()
    xb[1, 0, 10, 10] = float("nan")
    out_b = eng.forward(xb)
    torch.cuda.synchronize()
    assert not torch.isfinite(out_b[1]).all()
    assert eng.overflow_seen() and eng.overflow_seen(reset=True)          # sticky, then cleared
    eng.forward(xa)
    torch.cuda.synchronize()
    assert not eng.overflow_seen()
    eng.close()
    # the drop-in: the call AFTER the poisoned one reports it (warning), scans, and -- the fp16 tower itself being finite on clean input --
    # carries on; a network whose fp16 tower really saturates takes the bf16 fallback through the same scan (test above)
    net = LSegNet(labels=labels, backbone="tiny16", features=64, arch_option=0, block_depth=0, activation="lrelu", image_dtype="fp16")
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with torch.no_grad():
        net(xa)
        net(xb)
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            o = net(xa)
        assert any("overflow sentinel" in str(x.message) for x in w), [str(x.message) for x in w]
        assert torch.isfinite(o).all()


def test_fp16_range_check_and_loud_bf16_fallback():
    """VERDICT r3 item 3a / ADVICE: the inference default is fp16 MFMA operands, whose range (65504) the reference's fp32 tower does
    not have.  (1) lseg_check_range counts non-finite 16-bit activations; (2) with outlier statistics at the 1e2 level an fp16 engine
    stays finite and close to the fp32 oracle; (3) at a level that overflows fp16 the check reports it, a bf16 engine stays finite,
    and the drop-in network falls back to bf16 LOUDLY (RuntimeWarning) and returns the bf16 result instead of inf / NaN masks."""
    import warnings
    from modules.models.lseg_net import LSegNet
    cfg = get_config("tiny16")
    labels = read_labels(MG.LABELS)[:5]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    x = synthetic_images(2, 64, 64, seed=3)

    def run(sd, dtype):
        eng = HipEngine(cfg, 64, 64, max_batch=2, max_labels=5, image_dtype=dtype)
        eng.load_state_dict(sd)
        eng.set_tokens(tok)
        out = eng.forward(x.cuda()).cpu()
        r = eng.check_range()
        eng.close()
        return out, r

    sd = _outlier_state_dict(cfg, 5, 1e2)
    out16, r16 = run(sd, "fp16")
    with torch.no_grad():
        ref = lseg_forward(sd, x, tok, cfg)
    print("outliers 1e2: fp16 range", r16, "max|dlogit| vs oracle", (out16 - ref).abs().max().item())
    assert r16["nonfinite"] == 0 and r16["scanned"] > 1e5 and 50.0 < r16["max_abs"] < 65504.0
    assert torch.isfinite(out16).all() and (out16 - ref).abs().max().item() <= 0.30

    # one MLP whose HIDDEN activations leave the fp16 range (|z| ~ 3e5 > 65504) while its output is O(1) again: fp32 and bf16 hold it
    sd = _outlier_state_dict(cfg, 5, 1e2)
    sd["pretrained.model.blocks.1.mlp.fc1.weight"] *= 3e5
    sd["pretrained.model.blocks.1.mlp.fc1.bias"] *= 3e5
    sd["pretrained.model.blocks.1.mlp.fc2.weight"] /= 3e5
    out16, r16 = run(sd, "fp16")
    outbf, rbf = run(sd, "bf16")
    print("overflowing MLP: fp16 range", r16, "bf16 range", rbf)
    assert r16["nonfinite"] > 0                                  # the fp16 tower overflowed and the check saw it
    # (the scan sees every buffer as its LAST writer left it -- the MLP buffer is reused by the later blocks -- so the 3e5 hidden values of
    # block 1 are not in max_abs; an overflow is still caught because inf / NaN propagate through the fp32 residual stream to everything after)
    assert rbf["nonfinite"] == 0 and torch.isfinite(outbf).all()
    net = LSegNet(labels=labels, backbone="tiny16", features=64, arch_option=0, block_depth=0, activation="lrelu", image_dtype="fp16")
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            got = net(x.cuda()).cpu()
    assert any(issubclass(m.category, RuntimeWarning) and "falling back to bf16" in str(m.message) for m in w), [str(m.message) for m in w]
    assert net.image_dtype == "bf16" and torch.isfinite(got).all()
    assert (got - outbf).abs().max().item() <= 1e-3               # the bf16 engine's result, not a patched-up fp16 one
    with warnings.catch_warnings(record=True) as w2:              # and it stays on bf16 without further warnings
        warnings.simplefilter("always")
        with torch.no_grad():
            again = net(x.cuda()).cpu()
    assert not any("falling back" in str(m.message) for m in w2) and torch.equal(again, got)


def test_masks_and_metrics_without_the_full_resolution_logits():
    """lseg_forward(logits = NULL, masks) and lseg_op_seg_stats_lowres read the low-resolution logits through the x2 bilinear on the
    fly: same masks as argmax of the materialised output (except exact fp32 ties), same integer metric counts, same loss."""
    import ctypes as C
    from lseg_hip import _lib
    cfg, sd, tok, x, eng, logits, amax = run_engine(MG.CASES["tiny16_96x64_k7"], debug=False)
    B, K, H, W = logits.shape
    top2 = logits.topk(2, dim=1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(amax.long()[decisive], logits.argmax(1)[decisive]) and decisive.float().mean().item() > 0.99
    only = eng.forward(x.cuda(), want_logits=False, want_argmax=True)
    assert torch.equal(only, amax)
    g = torch.Generator().manual_seed(3)
    target = torch.randint(0, K, (B, H, W), generator=g)
    target[torch.rand((B, H, W), generator=g) < 0.2] = -1
    target = target.cuda()
    lib = _lib.load()
    low = eng.intermediate("lowres", (B, K, H // 2, W // 2))
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = []
    for fused in (False, True):
        counts = torch.zeros(2 + 3 * K, dtype=torch.int64, device="cuda")
        nll = torch.zeros(2, dtype=torch.float64, device="cuda")
        if fused:
            am = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
            _lib.check(lib.lseg_op_seg_stats_lowres(P(low), P(target), B, K, H // 2, W // 2, -1, P(counts), P(nll), P(am), st))
        else:
            _lib.check(lib.lseg_op_seg_stats(P(logits), P(target), B, K, H, W, -1, P(counts), P(nll), st))
        torch.cuda.synchronize()
        res.append((counts.cpu(), nll.cpu()))
    assert torch.equal(am, amax)
    # counts may differ only through pixels with an exact fp32 tie between the two best labels (none on this input)
    assert (res[0][0] - res[1][0]).abs().max().item() <= int((~decisive).sum().item())
    assert abs(res[0][1][0] - res[1][1][0]).item() <= 1e-6 * abs(res[0][1][0]).item() and res[0][1][1] == res[1][1][1]


@pytest.mark.gpu_fast
def test_one_pass_x4_upsample_equals_its_two_stages():
    """Production schedule: the logits come out of ONE pass over the quarter-resolution label planes (x2 bilinear * per-pixel 1/||.||,
    fp16 rounding, then output_conv's x2 bilinear, lseg_net.py:191-203) and the (h/2, w/2) logits stay in LDS.  They must equal
    output_conv applied to the low-resolution logits the engine writes on demand (the "lowres" tap / masks / metrics path), and the
    two ways of asking for the masks must agree."""
    import torch.nn.functional as F
    cfg, sd, tok, x, eng, logits, amax = run_engine(MG.CASES["tiny16_96x64_k7"], debug=False)
    B, K, H, W = logits.shape
    only_logits = eng.forward(x.cuda())                                     # one-pass x4 upsample
    low = eng.intermediate("lowres", (B, K, H // 2, W // 2))                # written on demand from the same label planes
    two_stage = F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)
    assert (only_logits - two_stage).abs().max().item() <= 2e-6 * max(1.0, two_stage.abs().max().item())
    both, amax2 = eng.forward(x.cuda(), want_logits=True, want_argmax=True)  # logits AND masks: the two x2 kernels
    assert torch.equal(both, only_logits) and torch.equal(amax2, amax)       # same arithmetic, spelled out (bilerp / src_tap)
    assert torch.equal(low, low.half().float())                             # the reference's fp16 logits before output_conv


def test_module_metrics_path_needs_no_logits():
    """LSegmentationModule.evaluate(x, target) (lsegmentation_module.py:43-52) through LSegNet.forward_metrics: the engine's
    lseg_forward_stats on the low-resolution logits == the device metric pass over the materialised [B,K,H,W] logits."""
    import warnings
    warnings.simplefilter("ignore")
    from lseg_hip import metrics
    from modules.models.lseg_net import LSegNet
    cfg = get_config("tiny16")
    sd = synthetic_state_dict(cfg, seed=2)
    labels = read_labels(MG.LABELS)[:6]
    net = LSegNet(labels=labels, backbone="tiny16", features=cfg.features, arch_option=0, block_depth=0, activation="lrelu")
    net.load_state_dict(sd)
    net = net.eval().cuda()
    x = synthetic_images(2, 64, 96, seed=2).cuda()
    g = torch.Generator().manual_seed(4)
    target = torch.randint(0, 6, (2, 64, 96), generator=g)
    target[torch.rand((2, 64, 96), generator=g) < 0.25] = -1
    with torch.no_grad():
        full = metrics.seg_stats(net(x), target.cuda())
    fused = net.forward_metrics(x, target.cuda())
    for k in ("correct", "labeled", "nll_count"):
        assert full[k] == fused[k], k
    for k in ("area_inter", "area_pred", "area_lab", "area_union"):
        assert torch.equal(full[k], fused[k]), k
    assert abs(full["nll_sum"] - fused["nll_sum"]) <= 1e-6 * abs(full["nll_sum"])


def test_split_k_residual_gemms_at_small_batch_equal_the_unsplit_schedule(tmp_path):
    """B = 1 at the BASELINE shape: attn.proj / mlp.fc2 run as split-K work items whose fp32 partial slabs the following LayerNorm sums
    into the residual stream (engine.hip split_residual).  Same logits as the unsplit schedule (LSEG_SPLITK=0, read once per
    process -> a second interpreter) up to fp32 summation order; and the partial sums are deterministic (two runs bit-equal)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from lseg_hip.config import get_config; from lseg_hip.engine import HipEngine\n"
        "from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels\n"
        "cfg = get_config('clip_vitl16_384'); sd = synthetic_state_dict(cfg, seed=0)\n"
        "tok = synthetic_tokens(read_labels(%r)[:7], cfg.text.vocab, cfg.text.ctx)\n"
        "outs = []\n"
        "for B in (1, 2):\n"
        "    eng = HipEngine(cfg, 160, 160, max_batch=B, max_labels=7, image_dtype='fp16'); eng.load_state_dict(sd); eng.set_tokens(tok)\n"
        "    x = synthetic_images(B, 160, 160, seed=1).cuda()\n"
        "    a = eng.forward(x).clone(); b = eng.forward(x); torch.cuda.synchronize()\n"
        "    assert torch.equal(a, b)\n"
        "    outs.append(a.cpu())\n"
        "torch.save(outs, sys.argv[1])\n"
    ) % (os.path.join(root, "lang-seg_amd"), root, MG.LABELS)
    res = {}
    for tag, target in (("split", None), ("plain", "0")):
        env = dict(os.environ)
        env.pop("LSEG_SPLITK", None)
        if target is not None:
            env["LSEG_SPLITK"] = target
        out = str(tmp_path / (tag + ".pt"))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
        res[tag] = torch.load(out)
    for a, b in zip(res["split"], res["plain"]):
        assert a.shape == b.shape and torch.isfinite(a).all()
        d = (a - b).abs().max().item()
        assert 0 < d <= 2e-2, d          # > 0: the split schedule really ran (101 tokens x 1024: 8 tiles -> 4 / 8 K-ranges); tiny: fp32 order + fp16 re-rounding


def test_text_features_handed_in_replace_the_text_tower():
    """lseg_set_text_features: features from `encode_text` handed back in give the same logits as the tokens did (the
    fp16 L2 normalisation is idempotent up to one rounding), the text tower is no longer run, and set_tokens switches back."""
    cfg, sd, tok, x, eng, logits, amax = run_engine(MG.CASES["tiny16_64x64_k5"], debug=False)
    feat = eng.encode_text()                              # normalised fp16 [K, out_c]
    eng2 = HipEngine(cfg, x.shape[2], x.shape[3], max_batch=x.shape[0], max_labels=tok.shape[0])
    eng2.load_state_dict(sd)
    eng2.set_text_features(feat * 3.0)                    # any scale: the engine normalises
    out = eng2.forward(x.cuda())
    torch.cuda.synchronize()
    assert (out - logits).abs().max().item() <= 2e-2
    eng2.set_tokens(tok)
    assert torch.equal(eng2.forward(x.cuda()), logits)


@pytest.mark.gpu_fast
def test_dedicated_correlation_kernel_equals_the_generic_pair(tmp_path):
    """The engine's correlation on the commuted schedule runs as ONE kernel (csrc/corr.hip: T resident in LDS, g streamed once, label planes
    + cell dot products) where round 4 ran the generic GEMM + pixel_gram_kernel (LSEG_CORR_GENERIC=1, read once per process -> a second
    interpreter).  The label planes use the same MFMA instruction, operand roles and k order as the GEMM they replace; the cell dot products
    are summed in another order (MFMA instead of fp32 FMAs + shuffles), which can move a logit by ONE fp16 rounding step of the reference's
    own `half @ half` rounding: BASELINE configs[1]'s shape (ViT-L/16, 480 x 480, K = 150, B = 2) and a small K (other instantiation)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from lseg_hip.config import get_config; from lseg_hip.engine import HipEngine\n"
        "from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels\n"
        "cfg = get_config('clip_vitl16_384'); sd = synthetic_state_dict(cfg, seed=0)\n"
        "outs = []\n"
        "for B, K, S in ((2, 150, 480), (1, 7, 160)):\n"
        "    tok = synthetic_tokens(read_labels(%r)[:K], cfg.text.vocab, cfg.text.ctx)\n"
        "    eng = HipEngine(cfg, S, S, max_batch=B, max_labels=K, image_dtype='fp16'); eng.load_state_dict(sd); eng.set_tokens(tok)\n"
        "    x = synthetic_images(B, S, S, seed=1).cuda()\n"
        "    eng.forward(x); torch.cuda.synchronize()\n"
        "    outs.append(eng.intermediate('lowres', (B, K, S // 2, S // 2)).cpu())\n"
        "    eng.close()\n"
        "torch.save(outs, sys.argv[1])\n"
    ) % (os.path.join(root, "lang-seg_amd"), root, MG.LABELS)
    res = {}
    for tag, generic in (("fused", False), ("generic", True)):
        env = dict(os.environ)
        env.pop("LSEG_CORR_GENERIC", None)
        if generic:
            env["LSEG_CORR_GENERIC"] = "1"
        out = str(tmp_path / (tag + ".pt"))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
        res[tag] = torch.load(out)
    for a, b in zip(res["fused"], res["generic"]):
        assert a.shape == b.shape and torch.isfinite(a).all()
        d = (a - b).abs()
        ulp = 2.0 ** (torch.floor(torch.log2(b.abs().clamp_min(2.0 ** -14))) - 10)          # fp16 spacing at the generic path's value
        frac = (d > 0).float().mean().item()
        print(f"fused vs generic correlation: {frac:.2e} of the low-resolution logits differ, max |d| {d.max().item():.2e}")
        assert (d <= ulp * 1.001).all(), (d / ulp).max().item()                              # never more than one fp16 step
        assert frac <= 2e-3, frac

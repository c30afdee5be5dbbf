"""The oracle must reproduce the committed golden fixtures (tests/golden/*.pt,
made by oracle/make_golden.py).  Guards against accidental oracle edits and
CPU/BLAS drift between the build container and the GPU box's host."""
import os

import pytest
import torch

from oracle import make_golden as MG


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_tiny_cases_match_golden(name, golden_dir):
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    cfg, out, inter = MG.run_case(MG.CASES[name])
    assert out.shape == g["logits"].shape
    assert torch.allclose(out, g["logits"], atol=2e-3, rtol=0), (out - g["logits"]).abs().max()
    # fp16-valued tensors: allow one fp16 ulp of BLAS reduction-order drift
    tf = inter["text_features"]
    assert (tf - g["text_features"].float()).abs().max() <= 4e-3
    assert (inter["lowres"].argmax(1) != g["lowres"].float().argmax(1)).float().mean() < 5e-3


@pytest.mark.parametrize("name", sorted(MG.FULL))
def test_full_size_cases_match_golden(name, golden_dir):
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    cfg, out, inter = MG.run_case(MG.FULL[name])
    sub = out[:, :, ::16, ::16]
    assert (sub - g["logits_sub16"]).abs().max() < 1e-2
    am = inter["lowres"].argmax(1).to(torch.uint8)
    assert (am != g["argmax_lowres"]).float().mean() < 2e-3
    assert (inter["text_features"] - g["text_features"].float()).abs().max() <= 8e-3


def test_correlation_is_left_associative():
    """SURVEY 'five facts' #5: logit_scale multiplies the fp16 pixel features
    BEFORE the GEMM (lseg_net.py:194); s*(x@t) is a different function."""
    from oracle.lseg_oracle import correlate, r16, LOGIT_SCALE
    g = torch.Generator().manual_seed(0)
    x = torch.randn((257, 64), generator=g)
    t = r16(torch.randn((9, 64), generator=g))
    ours = correlate(x, t)
    xn = x / x.norm(dim=-1, keepdim=True)
    tn = r16(t / r16(t.norm(dim=-1, keepdim=True)))
    left = r16(r16(LOGIT_SCALE * r16(xn)) @ tn.t())
    right = r16(LOGIT_SCALE * r16(r16(xn) @ tn.t()))
    assert torch.equal(ours, left)
    assert not torch.equal(ours, right)
    assert ours.dtype == torch.float32 and torch.equal(ours, r16(ours))


@pytest.mark.parametrize("name", sorted(MG.ZS_CASES))
def test_zero_shot_case_matches_golden_and_the_shared_label_path(name, golden_dir):
    """LSegNetZS.forward (lseg_net_zs.py:177-214): image b against its own ['others', class] pair.  Must equal
    running the shared-label forward on image b alone with those two labels."""
    from oracle.lseg_oracle import lseg_forward
    g = torch.load(os.path.join(golden_dir, name + ".pt"))
    cfg, sd, x, tok, out, inter = MG.run_zs_case(MG.ZS_CASES[name])
    assert out.shape == g["logits"].shape == (x.shape[0], 2, x.shape[2], x.shape[3])
    assert torch.allclose(out, g["logits"], atol=2e-3, rtol=0)
    for b in range(x.shape[0]):
        with torch.no_grad():
            single = lseg_forward(sd, x[b:b + 1], tok[2 * b:2 * b + 2], cfg)
        assert torch.allclose(single[0], out[b], atol=2e-3, rtol=0)

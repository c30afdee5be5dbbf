"""The exact re-associations the engine's head schedule rests on, checked in float64 on the CPU with the
reference's own operators (torch's upsample_bilinear2d with align_corners=True = scratch.output_conv / FeatureFusionBlock's upsample,
lseg_net.py:185-203, lseg_blocks.py:352-356):

  1. a 1x1 convolution commutes with the bilinear upsample  ->  head1(out_conv(up(t))) = up(Wc t + bc)
  2. so does the pixel x text correlation, and the norm of an up-sampled feature follows from the dot products of the 2x2 cells
     ->  logits[k, P] = s * up(t_k . g)[P] / ||up(g)_P||,  ||up(g)_P||^2 = sum_ij w_i w_j (g_i . g_j)
  3. under a x2 align_corners=True upsample every output row's two taps lie within [2y-1, 2y+2] of the rows they feed (the 4-tap
     footprint of the fused upsample + cross-entropy backward), in the fp32 arithmetic the kernels use.
"""
import numpy as np
import torch
import torch.nn.functional as F


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def test_1x1_convolutions_commute_with_the_bilinear_upsample():
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 16, 9, 11, generator=g, dtype=torch.float64)
    w1, b1 = torch.randn(16, 16, 1, 1, generator=g, dtype=torch.float64), torch.randn(16, generator=g, dtype=torch.float64)
    w2, b2 = torch.randn(24, 16, 1, 1, generator=g, dtype=torch.float64), torch.randn(24, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.conv2d(up2(t), w1, b1), w2, b2)                       # the reference's order: upsample, out_conv, head1
    wc = (w2[:, :, 0, 0] @ w1[:, :, 0, 0])[:, :, None, None]
    bc = w2[:, :, 0, 0] @ b1 + b2
    got = up2(F.conv2d(t, wc, bc))                                         # one combined 1x1 conv below the upsample
    assert (got - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


def test_the_correlation_commutes_with_the_upsample_and_the_norm_follows_from_cell_dot_products():
    gen = torch.Generator().manual_seed(1)
    B, C, H, W, K, s = 2, 32, 7, 10, 5, 14.285714
    g = torch.randn(B, C, H, W, generator=gen, dtype=torch.float64)
    txt = torch.randn(K, C, generator=gen, dtype=torch.float64)
    txt = txt / txt.norm(dim=-1, keepdim=True)
    u = up2(g)                                                              # the reference: features at (2H, 2W), normalise, correlate
    a = s * u / u.norm(dim=1, keepdim=True)
    ref = torch.einsum("kc,bchw->bkhw", txt, a)
    R = torch.einsum("kc,bchw->bkhw", txt, g)                               # the engine: correlate at (H, W) ...
    # ... and ||u_P||^2 from the five dot products per pixel of g: self, right, down, diagonal, anti-diagonal
    gp = F.pad(g, (0, 1, 0, 1))
    c, r, d, dr = gp[:, :, :H, :W], gp[:, :, :H, 1:], gp[:, :, 1:, :W], gp[:, :, 1:, 1:]
    gram = torch.stack([(c * c).sum(1), (c * r).sum(1), (c * d).sum(1), (c * dr).sum(1), (r * d).sum(1)], dim=-1)   # [B,H,W,5]
    Ho, Wo = 2 * H, 2 * W
    ry, rx = (H - 1) / (Ho - 1), (W - 1) / (Wo - 1)
    n2 = torch.zeros(B, Ho, Wo, dtype=torch.float64)
    for yo in range(Ho):
        sy = ry * yo; y0 = min(int(sy), H - 1); y1 = min(y0 + 1, H - 1); ly = sy - y0
        for xo in range(Wo):
            sx = rx * xo; x0 = min(int(sx), W - 1); x1 = min(x0 + 1, W - 1); lx = sx - x0
            wa, wb, wc_, wd = (1 - ly) * (1 - lx), (1 - ly) * lx, ly * (1 - lx), ly * lx
            ra, rb, rc, rd = gram[:, y0, x0], gram[:, y0, x1], gram[:, y1, x0], gram[:, y1, x1]
            n2[:, yo, xo] = (wa * wa * ra[:, 0] + wb * wb * rb[:, 0] + wc_ * wc_ * rc[:, 0] + wd * wd * rd[:, 0] +
                             2 * (wa * wb * ra[:, 1] + wa * wc_ * ra[:, 2] + wa * wd * ra[:, 3] + wb * wc_ * ra[:, 4] +
                                  wb * wd * rb[:, 2] + wc_ * wd * rc[:, 1]))
    assert (n2.sqrt() - u.norm(dim=1)).abs().max().item() <= 1e-12 * u.norm(dim=1).max().item()
    got = up2(R) * (s / n2.sqrt())[:, None]
    assert (got - ref).abs().max().item() <= 1e-11 * ref.abs().max().item()


def test_x2_align_corners_footprint_is_four_rows_in_fp32():
    """csrc/elementwise.hip x2_footprint_is_4 / upsample_ce_bwd_rows_kernel: with r = fp32((H-1)/(2H-1)) and s = fp32(r * Y), the rows
    Y that put weight on low row y are within [2y-1, 2y+2] -- for every map height the engine can meet (and far beyond)."""
    for H in list(range(2, 260)) + [300, 384, 480, 512, 600, 768, 1024]:
        Ho = 2 * H
        r = np.float32(H - 1) / np.float32(Ho - 1)
        Y = np.arange(Ho, dtype=np.int64)
        s = (r * Y.astype(np.float32)).astype(np.float32)
        y0 = s.astype(np.int64)
        l = s - y0.astype(np.float32)
        y1 = np.minimum(y0 + 1, H - 1)
        assert (y0 >= 0).all() and (y0 <= H - 1).all()
        ok0 = (l == 1.0) | ((Y >= 2 * y0 - 1) & (Y <= 2 * y0 + 2))          # weight 1 - l on y0
        ok1 = (l == 0.0) | (y1 == y0) | ((Y >= 2 * y1 - 1) & (Y <= 2 * y1 + 2))   # weight l on y1
        assert ok0.all() and ok1.all(), H

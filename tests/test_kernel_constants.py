"""Constants the HIP kernels carry, checked on the CPU against their definitions (no GPU, no library call).

gemm.hip::gelu_fast evaluates nn.GELU() (the erf form timm's Mlp uses: [3P] timm 0.4.12 vision_transformer.py, reached through
/root/reference/modules/models/lseg_vit.py:196-197) as  max(x, 0) - |x| * exp2(P(|x|) - 1)  with a degree-6 polynomial P without constant
term, |x| clamped at 4.4 * sqrt(2).  The test parses the coefficients out of the source and holds the formula -- in emulated fp32 -- to the
exact 0.5 x erfc(-x / sqrt 2): a typo in one constant is a 1e-3 error, the fit's own error is 4.6e-7 (tools/fit_gelu.py)."""
import os
import re

import numpy as np
import pytest

scipy_special = pytest.importorskip("scipy.special")
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lang-seg_amd", "csrc", "gemm.hip")


def _gelu_fast_constants():
    src = open(SRC).read()
    body = src[src.index("__device__ __forceinline__ float gelu_fast(float x)"):]
    body = body[:body.index("\n}\n")]
    clamp = float(re.search(r"fminf\(fabsf\(x\), ([0-9.eE+-]+)f\)", body).group(1))
    lead = float(re.search(r"float r = ([0-9.eE+-]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"r = fmaf\(r, u, ([0-9.eE+-]+)f\);", body)]
    assert "fmaf(r, u, -1.0f)" in body and "fmaxf(x, 0.f)" in body
    return clamp, [lead] + rest


def test_gelu_fast_constants_reproduce_the_erf_gelu():
    clamp, coef = _gelu_fast_constants()
    assert len(coef) == 6 and abs(clamp - 4.4 * np.sqrt(2.0)) < 1e-5
    f32 = np.float32
    x = np.linspace(-12, 12, 1200001).astype(f32)
    u = np.minimum(np.abs(x), f32(clamp)).astype(f32)
    r = np.full_like(u, f32(coef[0]))
    for c in coef[1:]:
        r = (r * u + f32(c)).astype(f32)
    h = np.exp2((r * u - f32(1.0)).astype(f32)).astype(f32)
    y = (np.maximum(x, 0) - np.abs(x) * h).astype(f32)
    ref = 0.5 * x.astype(np.float64) * scipy_special.erfc(-x.astype(np.float64) / np.sqrt(2.0))
    err = np.abs(y - ref)
    assert err.max() < 1e-6, (err.max(), float(x[err.argmax()]))
    # the halved complementary error function at 0 is exactly 1/2 (no constant term in P): GELU(x) ~ x / 2 near 0 with RELATIVE accuracy
    small = np.abs(x) < 1e-2
    assert (err[small] <= 1e-6 * np.abs(ref[small]) + 1e-9).all()

#!/usr/bin/env python
"""Coefficients of gemm.hip::gelu_fast: erfc(z) ~= exp2(z R(z)) on [0, 4.4] with R of degree 5 (so erfc(0) = 1 exactly), fitted to
log2(erfc) by least squares re-weighted towards the absolute error of erfc; then the GELU built on it, evaluated in emulated fp32,
against the exact 0.5 x erfc(-x / sqrt 2).  (numpy + scipy, CPU.)"""
import numpy as np
from scipy import special
z = np.linspace(0, 4.4, 40001)
t = special.erfc(z); lg = np.log2(t)
deg = 6
A = np.vstack([z ** k for k in range(1, deg + 1)]).T
w = np.ones_like(z)
for it in range(200):
    c, *_ = np.linalg.lstsq(A * w[:, None], lg * w, rcond=None)
    err = np.exp2(A @ c) - t
    w = (t * np.log(2)) * (1 + 4 * np.abs(err) / (np.abs(err).max() + 1e-30))
c32 = c.astype(np.float32)
print("R coefficients (z^0 .. z^5):", [repr(float(v)) for v in c32])
# gemm.hip evaluates in u = |x| = sqrt(2) z: t(u) = sum_k d_k u^(k+1) - 1 with d_k = c_k 2^(-(k+1)/2) (round 4: one multiply less per element)
d32 = np.array([c[k] * 0.70710678118654752 ** (k + 1) for k in range(6)]).astype(np.float32)
print("folded coefficients (u^1 .. u^6):", [repr(float(v)) for v in d32], "; clamp |x| at", repr(float(np.float32(4.4 / 0.70710678118654752))))
print("max |erfc error| (f64 evaluation): %.3g" % np.abs(err).max())
x = np.linspace(-10, 10, 2000001).astype(np.float32)
az = np.minimum(np.abs(x), np.float32(4.4 / 0.70710678118654752)).astype(np.float32)
r = np.float32(d32[5]) * np.ones_like(az)
for k in range(4, -1, -1):
    r = (r * az + d32[k]).astype(np.float32)
h = np.exp2((r * az - np.float32(1.0)).astype(np.float32)).astype(np.float32)
y = (np.maximum(x, 0) - np.abs(x) * h).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * special.erfc(-x.astype(np.float64) / np.sqrt(2))
e = np.abs(y - ref)
print("GELU (fp32 evaluation) max |error| %.3g at x = %.3f; max error / max(|y|, 1e-2) = %.3g" % (e.max(), x[e.argmax()], (e / np.maximum(np.abs(ref), 1e-2)).max()))

#!/usr/bin/env python
"""Standalone timing of the dedicated correlation kernel (csrc/corr.hip, lseg_op_corr_planes) at the bench shape: B images of the padded
122 x 122 quarter-resolution map, C = 512, K = 150.  Algorithmic bytes = g read once + label planes written + gram records written.
  python tools/corr_bench.py [--batch 36] [--labels 150]"""
import argparse, ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
from lseg_hip import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="*", default=[36, 4, 1]); ap.add_argument("--labels", type=int, default=150)
ap.add_argument("--hw", type=int, default=120); ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
for B in a.batch:
    H = W = a.hw; K = a.labels
    g = (torch.randn((B, H + 2, W + 2, 512), device="cuda") * 0.5).half()
    T = torch.randn((K, 512), device="cuda"); T = (T / T.norm(dim=-1, keepdim=True)).half()
    R = torch.empty((B, K, H + 2, W + 2), device="cuda"); G = torch.empty((B, H, W, 5), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode, gp in (("planes + gram", P(G)), ("planes only", None)):
        for _ in range(3):
            _lib.check(lib.lseg_op_corr_planes(P(g), P(T), P(R), gp, B, K, H, W, 512, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            _lib.check(lib.lseg_op_corr_planes(P(g), P(T), P(R), gp, B, K, H, W, 512, st))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        alg = g.numel() * 2 + B * K * H * W * 4 + (B * H * W * 20 if gp else 0) + K * 1024
        print(f"B={B} K={K} {H}x{W} {mode}: {us:.1f} us  {alg / 1e6:.1f} MB algorithmic -> {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s; "
              f"{2 * K * 512 * B * (H + 2) * (W + 2) / us / 1e6:.0f} TFLOP/s on the padded map")

#!/usr/bin/env python
"""The hand-scheduled residual GEMM (csrc/gemm_asm.hip) against the generic kernel family and against torch, standalone through the C ABI
(lseg_op_gemm_res32): correctness on ragged and multi-tile shapes, then us per launch / TFLOP/s at the engine's shapes (B = 36: M = 32436)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd")); sys.path.insert(0, ROOT)
import torch
from lseg_hip import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--check-only", action="store_true"); ap.add_argument("--dtype", default="fp16"); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--shapes", default="proj,fc2"); ap.add_argument("--no-check", action="store_true")
a = ap.parse_args()
lib = _lib.load()
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
code = _lib.LSEG_F16 if a.dtype == "fp16" else _lib.LSEG_BF16
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(A, W, b, Cbuf, M, impl, max_grid=0):
    _lib.check(lib.lseg_op_gemm_res32(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(Cbuf.data_ptr()),
                                      M, W.shape[0], W.shape[1], Cbuf.shape[0], code, impl, max_grid, st()))


def make(M, N, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rows = (M + 255) // 256 * 256
    A = torch.randn(rows, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dt)
    b = torch.randn(N, device="cuda", generator=g)
    C0 = torch.randn(rows, N, device="cuda", generator=g) * 2
    return A, W, b, C0


ok = True
for (M, N, K) in [] if a.no_check else [(256, 128, 1024), (700, 256, 1024), (1500, 384, 1280), (4000, 1024, 4096), (901 * 8, 1024, 1024)]:
    A, W, b, C0 = make(M, N, K, seed=M)
    ref = C0[:M].double() + A[:M].double() @ W.double().t() + b.double()
    outs = {}
    for impl in (1, 0):
        Cb = C0.clone()
        run(A, W, b, Cb, M, impl)
        torch.cuda.synchronize()
        outs[impl] = Cb[:M]
    e1 = (outs[1].double() - ref).abs().max().item(); e0 = (outs[0].double() - ref).abs().max().item()
    good = e1 <= 2e-5 * max(1.0, ref.abs().max().item()) * (K / 1024) ** 0.5 + 1e-4
    ok &= good
    print(f"M={M} N={N} K={K}: max|asm - fp64| {e1:.3e}  max|generic - fp64| {e0:.3e}  |ref| {ref.abs().max().item():.1f}  {'ok' if good else 'MISMATCH'}", flush=True)
if not ok:
    sys.exit(1)
if a.check_only:
    sys.exit(0)

shapes = {"proj": (32436, 1024, 1024), "fc2": (32436, 1024, 4096), "proj_b8": (7208, 1024, 1024), "fc2_b8": (7208, 1024, 4096)}
for name in a.shapes.split(","):
    M, N, K = shapes[name]
    A, W, b, C0 = make(M, N, K)
    res = {}
    for rnd in range(2):
        for impl in (1, 0):
            Cb = C0.clone()
            for _ in range(3): run(A, W, b, Cb, M, impl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(a.iters): run(A, W, b, Cb, M, impl)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(impl, []).append(e0.elapsed_time(e1) / a.iters * 1e3)
    gf = 2.0 * M * N * K / 1e9
    for impl in (1, 0):
        us = min(res[impl])
        print(f"{name} M={M} N={N} K={K} {a.dtype} {'asm 256x128' if impl else 'generic    '}: {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s  = {gf / us * 1e3 / 2500:.3f} of 2.5 PF   rounds {['%.1f' % x for x in res[impl]]}", flush=True)

#!/usr/bin/env python
"""The one-pass x4 logits upsample (the 4.98 GB write of a B = 36 forward) standalone: time, GB/s on the algorithmic bytes, and bit
equality with the two-stage form (round 4's phase-3 forms -- direct / rolling rows / separable through LDS, plain / non-temporal stores,
band 8 / 16 -- were compared with this tool behind a launch-time switch; figures in profiles/r04_head_kernels.txt, the rolling form is kept).
  python tools/upsample_bench.py 36 4"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in [int(v) for v in sys.argv[1:]] or [36]:
    K, H, W = 150, 120, 120
    g = torch.Generator(device="cuda").manual_seed(0)
    R = torch.zeros((B * K, H + 2, W + 2), device="cuda")
    R[:, 1:-1, 1:-1] = torch.randn((B * K, H, W), generator=g, device="cuda")
    sc = torch.rand((B, 2 * H, 2 * W), generator=g, device="cuda") + 0.5
    out = torch.empty((B, K, 4 * H, 4 * W), device="cuda")
    run = lambda: _lib.check(lib.lseg_op_upsample4x_planes_scaled(P(R), P(sc), P(out), B, K, H, W, 0, None, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nbytes = 4 * (R.numel() + sc.numel() + out.numel())
    same = None
    if B <= 8:
        low = torch.empty((B, K, 2 * H, 2 * W), device="cuda"); ref = torch.empty_like(out)
        _lib.check(lib.lseg_op_upsample4x_planes_scaled(P(R), P(sc), P(ref), B, K, H, W, 1, P(low), st))
        torch.cuda.synchronize()
        same = bool(torch.equal(ref, out))
    print(f"B={B}: {us:.1f} us -> {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.3f} of 8 TB/s)"
          + (f"; equals the two-stage form bit for bit: {same}" if same is not None else ""), flush=True)

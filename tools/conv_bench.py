#!/usr/bin/env python
"""The DPT head's hot 3x3 conv standalone (lseg_op_conv3x3: 256 -> 256 channels at 120 x 120, B = 36, padded NHWC bf16, with / without the
skip input): us per launch and TFLOP/s.  With LSEG_HIP_LIB pointing at the `abl2` attribution build (make -C lang-seg_amd/csrc probes:
no epilogue) the same launch times the implicit-GEMM K-loop alone -- the difference is what the padded-NHWC epilogue costs."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, W, Cin, Cout = int(os.environ.get("B", "36")), 120, 120, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
xp = torch.zeros((B, H + 2, W + 2, Cin), dtype=dt, device="cuda"); xp[:, 1:-1, 1:-1] = torch.relu(torch.randn((B, H, W, Cin), generator=g, device="cuda")).to(dt)
wp = (torch.randn((Cout, 9 * Cin), generator=g, device="cuda") / math.sqrt(9 * Cin)).to(dt)
bias = torch.randn((Cout,), generator=g, device="cuda")
resp = torch.zeros((B, H + 2, W + 2, Cout), dtype=dt, device="cuda"); resp[:, 1:-1, 1:-1] = torch.randn((B, H, W, Cout), generator=g, device="cuda").to(dt)
out = torch.zeros((B, H + 2, W + 2, Cout), dtype=dt, device="cuda")
flops = 2.0 * B * H * W * Cout * 9 * Cin
for name, res in (("no residual", None), ("with residual", resp)):
    run = lambda: _lib.check(lib.lseg_op_conv3x3(P(xp), P(wp), P(bias), P(res), P(out), B, H, W, Cin, Cout, 1, 0, 0, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{os.environ.get('LSEG_PROBE_VARIANT', 'library')} conv3x3 256->256 @120^2 B={B} {name}: {us:.1f} us  {flops / us / 1e6:.0f} TF/s", flush=True)

#!/usr/bin/env python
"""Time lseg_op_linear_backward (dgrad + wgrad + bias grad = 4*M*N*K flops, plus the operand transposes) at ViT shapes."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
for (M, N, K) in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32436, 4096, 1024), (32436, 1024, 4096), (32436, 1024, 1024)]:
    dy = torch.randn(M, N).to(torch.bfloat16).cuda(); x = torch.randn(M, K).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K) / math.sqrt(K)).to(torch.bfloat16).cuda()
    dx = torch.empty_like(x); dw = torch.empty((N, K), device="cuda"); db = torch.empty(N, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _lib.check(lib.lseg_op_linear_backward(P(dy), P(x), P(w), _lib.LSEG_BF16, P(dx), P(dw), P(db), M, N, K, st))
    for _ in range(3): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{M}x{N}x{K}: {ms:.3f} ms  {4.0 * M * N * K / ms / 1e9:.0f} TF/s (both GEMMs, transposes included)")

#!/usr/bin/env python
"""Time lseg_op_attention_backward at the ViT-L shape (901 tokens, 16 heads): 5 matmuls = 10*N^2*64 flops per head."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
B, H, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 16, 901
Npad = (N + 127) // 128 * 128
bf = torch.bfloat16
q = torch.zeros((B * H, Npad, 64), dtype=bf, device="cuda"); q[:, :N] = torch.randn((B * H, N, 64), device="cuda").to(bf)
k = torch.zeros_like(q); k[:, :N] = torch.randn((B * H, N, 64), device="cuda").to(bf)
vt = torch.zeros((B * H, 64, Npad), dtype=bf, device="cuda"); vt[:, :, :N] = torch.randn((B * H, 64, N), device="cuda").to(bf)
o = torch.zeros((B, N, H * 64), dtype=bf, device="cuda"); d_o = torch.randn((B, N, H * 64), device="cuda").to(bf)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.lseg_op_attention(P(q), P(k), P(vt), P(o), B, H, N, Npad, _lib.LSEG_BF16, 0, 0.125, st))
s = (q[:, :N].float() @ k[:, :N].float().transpose(1, 2)) * 0.125
lse2 = torch.zeros((B * H, Npad), device="cuda"); lse2[:, :N] = torch.logsumexp(s, -1) * 1.4426950408889634
del s
dq, dk, dv = (torch.empty((B * H, Npad, 64), device="cuda") for _ in range(3))
run = lambda: _lib.check(lib.lseg_op_attention_backward(P(q), P(k), P(vt), P(o), P(d_o), P(lse2), P(dq), P(dk), P(dv), B, H, N, Npad, _lib.LSEG_BF16, 0, 0.125, st))
for _ in range(2): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"B={B}: {ms:.3f} ms, {10.0 * N * N * 64 * B * H / ms / 1e9:.0f} TF/s")

#!/usr/bin/env python
"""Time lseg_op_attention at the ViT-L shape (901 tokens, 16 heads, head_dim 64) and check it against fp32 torch on two heads (tools).
PRESCALED=1 times the inference engine's form (lseg_op_attention_prescaled: q arrives as q * scale * log2 e).  The round-4 A/B of the four
kernel bodies is recorded in profiles/r04_attention_experiments.txt."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in (int(v) for v in (sys.argv[1:] or ["36"])):
    H, N, Npad = 16, 901, 1024
    for dt, code in ((torch.bfloat16, _lib.LSEG_BF16), (torch.float16, _lib.LSEG_F16)):
        g = torch.Generator().manual_seed(0)
        q = torch.zeros((B * H, Npad, 64), dtype=dt).cuda(); q[:, :N] = torch.randn((B * H, N, 64), generator=g).to(dt).cuda()
        k = torch.zeros((B * H, Npad, 64), dtype=dt).cuda(); k[:, :N] = torch.randn((B * H, N, 64), generator=g).to(dt).cuda()
        vt = torch.zeros((B * H, 64, Npad), dtype=dt).cuda(); vt[:, :, :N] = torch.randn((B * H, 64, N), generator=g).to(dt).cuda()
        out = torch.zeros((B, N, H * 64), dtype=dt).cuda()
        pre = os.environ.get("PRESCALED", "0") == "1"
        if pre:
            q = (q.float() * (0.125 * 1.4426950408889634)).to(dt)
            run = lambda: _lib.check(lib.lseg_op_attention_prescaled(P(q), P(k), P(vt), P(out), None, B, H, N, Npad, code, st))
        else:
            run = lambda: _lib.check(lib.lseg_op_attention(P(q), P(k), P(vt), P(out), B, H, N, Npad, code, 0, 0.125, st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        err = 0.0
        for bh in (0, B * H - 1):
            s = (q[bh, :N].float() @ k[bh, :N].float().t()) * (0.6931471805599453 if pre else 0.125)
            ref = s.softmax(-1) @ vt[bh, :, :N].float().t()
            b, h = bh // H, bh % H
            err = max(err, (out[b, :, h * 64:(h + 1) * 64].float() - ref).abs().max().item())
        print(f"prescaled={int(pre)} B={B} {str(dt)[6:]}: {us:.1f} us -> {4.0 * B * H * N * N * 64 / us / 1e6:.0f} TF/s; max|err| vs fp32 torch {err:.4f}", flush=True)

#!/usr/bin/env python
"""Do the fp16 MFMA / conversions on this GPU keep fp16 SUBNORMALS (the reference's half-precision head gradient lives there)?
A [M,K] fp16 of values k * 2^-24, W = ones: every output must be the exact sum; fp16 output must keep subnormal results."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 128, 128, 64
g = torch.Generator().manual_seed(0)
k = torch.randint(0, 40, (M, K), generator=g).float()
A = (k * 2.0 ** -24).to(torch.float16).cuda()
assert torch.equal(A.float().cpu(), k * 2.0 ** -24)
W = torch.zeros((N, K), dtype=torch.float16); W[:, :3] = 1.0; W = W.cuda()
for od, code in ((torch.float32, _lib.LSEG_F32), (torch.float16, _lib.LSEG_F16)):
    out = torch.zeros((M, N), dtype=od).cuda()
    _lib.check(lib.lseg_op_gemm(P(A), P(W), None, None, P(out), M, N, K, _lib.LSEG_F16, code, 0, st))
    torch.cuda.synchronize()
    ref = (k[:, :3].sum(1, keepdim=True) * 2.0 ** -24).expand(M, N)
    print("fp16 subnormal operands ->", od, "max rel err", float(((out.float().cpu() - ref).abs() / ref.clamp_min(2.0 ** -24)).max()),
          "zeros where ref nonzero:", int(((out.float().cpu() == 0) & (ref > 0)).sum()))

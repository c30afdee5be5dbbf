#!/usr/bin/env python
"""Micro-benchmark of the MFMA GEMM through the C ABI (used with rocprofv3 for PMC counters)."""
import ctypes as C, os, sys, math, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1:] or ["7208x4096x1024", "7208x1024x4096", "7208x3072x1024", "7208x1024x1024", "901x4096x1024", "8192x8192x8192", "4096x4096x4096"])]
iters = int(os.environ.get("ITERS", "20"))
for (M, N, K) in shapes:
    A = (torch.randn(M, K) ).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = torch.randn(N).cuda()
    out = torch.empty((M, N), dtype=torch.bfloat16).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for act in (0, 1):
        for _ in range(3):
            _lib.check(lib.lseg_op_gemm(P(A), P(W), P(bias), None, P(out), M, N, K, 2, 2, act, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            _lib.check(lib.lseg_op_gemm(P(A), P(W), P(bias), None, P(out), M, N, K, 2, 2, act, st))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(json.dumps({"M": M, "N": N, "K": K, "act": act, "ms": round(ms, 4), "TF": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)

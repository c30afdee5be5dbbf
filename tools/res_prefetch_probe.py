#!/usr/bin/env python
"""Is the residual read-modify-write of attn.proj / mlp.fc2 faster when the residual stream sits in the Infinity Cache?  (tools)
The epilogue is bound by what one CU keeps in flight against memory latency (profiles/r06_gemm_asm.txt).  In the engine x was last touched three
kernels earlier (266 MB of other traffic ago); here the GEMM alone is timed (HIP events around the launch only) with x
  warm      left in the caches by the previous launch of the same GEMM,
  cold      after 1 GB of unrelated traffic,
  prefetch  cold, then one 4-byte load per 128-byte line of x (what a producer kernel could issue for free) right before the launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
M, D = 36 * 901, 1024
g = torch.Generator(device="cuda").manual_seed(0)
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); junk2 = torch.empty_like(junk)
for name, K in (("proj", D), ("fc2", 4 * D)):
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((D, K), generator=g, device="cuda") * 0.02).half()
    b = torch.randn(D, generator=g, device="cuda")
    x = torch.randn((M, D), generator=g, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def gemm():
        _lib.check(lib.lseg_op_gemm_vit(P(A), P(W), P(b), P(x), None, None, M, D, K, _lib.LSEG_F16, 2, 901, 1024, 0, st))
    res = {}
    for mode in ("warm", "cold", "prefetch", "cold", "prefetch", "warm"):
        ts = []
        for it in range(8):
            if mode != "warm":
                junk2.copy_(junk)                                  # 1 GB of traffic: x leaves L2 and the Infinity Cache
            if mode == "prefetch":
                s_ = x.view(-1, 32)[:, 0].sum()                    # one element per 128-byte line
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gemm(); e1.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        res.setdefault(mode, []).append(sum(ts) / len(ts))
    print(name, {k: [round(v, 1) for v in vs] for k, vs in res.items()}, "us per launch", flush=True)

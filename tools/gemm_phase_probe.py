#!/usr/bin/env python
"""Where does a GEMM wave spend its time?  (LSEG_GEMM_DBG=3: s_memtime marks around the vmcnt wait, the
barrier, the MFMA block, the deferred epilogue and the first K-step of every tile; summed per wave and
dumped through the residual pointer, 8 slots per wave.  The marks perturb the kernel a little: s_memtime
returns through lgkmcnt.)"""
import ctypes as C, os, sys, math
os.environ["LSEG_GEMM_DBG"] = os.environ.get("PROBE_DBG", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
BM = int(os.environ.get("PROBE_BM", "128")); BN = int(os.environ.get("PROBE_BN", "128"))
WAVES = (BM // 64) * (BN // 64) if BM * BN < 65536 else 8       # waves per workgroup
MFMA_STEP = (BM * BN // WAVES // 256) * 2 * 16 * (WAVES // 4)   # MFMA-pipe cycles per K-step per SIMD
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or \
    [(28832, 4096, 1024), (28832, 1024, 4096), (7208, 1024, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    A = torch.randn(M, K).to(torch.bfloat16).cuda(); W = (torch.randn(N, K) / math.sqrt(K)).to(torch.bfloat16).cuda()
    out = torch.empty((M, N), dtype=torch.bfloat16).cuda()
    bias = torch.randn(N).cuda()
    dump = torch.zeros((512 * 8 * 8,), dtype=torch.int64).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        dump.zero_()
        _lib.check(lib.lseg_op_gemm(P(A), P(W), P(bias), P(dump), P(out), M, N, K, 2, 2, 0, st))
    torch.cuda.synchronize()
    d = dump.view(-1, 8).cpu().double(); d = d[d[:, 6] > 0]
    wgs = d.shape[0] / WAVES
    tiles = math.ceil(M / BM) * math.ceil(N / BN) / wgs           # tiles per workgroup
    nk = K // 64
    m = d.mean(0)
    ghz = m[6] / (m[7] * 10.0)
    per_tile = m[6] / tiles
    print(f"{M}x{N}x{K}: {int(wgs)} workgroups, {tiles:.1f} tiles each, clock {ghz:.2f} GHz, kernel {m[7] * 0.01:.1f} us/wave, "
          f"{per_tile:.0f} cycles/tile (MFMA-bound: {nk * MFMA_STEP * (2 if WAVES == 4 else 1)})")
    print(f"   per K-step: vmcnt-wait {m[0] / (tiles * nk):.0f}  barrier {m[1] / (tiles * nk):.0f}  MFMA blocks {m[3] / (tiles * nk):.0f}"
          f"   | per tile: epilogue {m[2] / tiles:.0f}  K-loop total {(m[0] + m[1] + m[3]) / tiles:.0f}")

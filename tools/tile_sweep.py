#!/usr/bin/env python
"""Tile-choice sweep of the MFMA GEMM at the ViT shapes over batch sizes (tools): one subprocess per forced tile config
(LSEG_GEMM_TILE is read once per process; 0 = the cost model's own choice), HIP-event time per launch through lseg_op_gemm.
  python tools/tile_sweep.py            # parent: prints one table row per (shape, batch)
"""
import ctypes as C, json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = {"qkv": (3072, 1024, "lin"), "proj": (1024, 1024, "res"), "fc1": (4096, 1024, "gelu"), "fc2": (1024, 4096, "res")}
BATCHES = [int(v) for v in os.environ.get("BATCHES", "1,2,4,8,16").split(",")]

def child():
    sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
    import torch
    from lseg_hip import _lib
    lib = _lib.load()
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for B in BATCHES:
        M = B * 901
        for name, (N, K, kind) in KINDS.items():
            A = torch.randn(M, K).to(torch.bfloat16).cuda()
            W = (torch.randn(N, K) / math.sqrt(K)).to(torch.bfloat16).cuda()
            bias = torch.randn(N).cuda()
            if kind == "res":
                res = torch.randn(M, N).cuda(); o = res; od = 0          # fp32 in-place residual stream (LSEG_F32 = 0)
            else:
                res = None; o = torch.empty((M, N), dtype=torch.bfloat16).cuda(); od = 2
            act = 1 if kind == "gelu" else 0
            call = lambda: _lib.check(lib.lseg_op_gemm(P(A), P(W), P(bias), P(res), P(o), M, N, K, 2, od, act, st))
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): call()
            e1.record(); torch.cuda.synchronize()
            out[f"{name}@{B}"] = e0.elapsed_time(e1) / 30 * 1e3
    print("RESULT " + json.dumps(out), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(); sys.exit(0)
res = {}
for tile in (0, 1, 2, 6):
    env = dict(os.environ); env["LSEG_GEMM_TILE"] = str(tile)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line: print("tile", tile, "failed:", p.stderr[-400:]); continue
    res[tile] = json.loads(line[0][7:])
print(f"{'shape@B':12s} {'auto':>8s} {'64x64':>8s} {'128x128':>8s} {'256x256':>8s}   (us per launch)")
for key in res.get(0, {}):
    print(f"{key:12s} " + " ".join(f"{res[t][key]:8.1f}" if t in res and key in res[t] else "     n/a" for t in (0, 1, 2, 6)))

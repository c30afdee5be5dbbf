#!/usr/bin/env python
"""Where does a wave of the attention forward kernel spend its cycles?  Compiles csrc/attention.hip with its ATTN_PROBE hooks turned into
s_memtime marks (every mark also waits for the wave's outstanding LDS reads, so the picture is coarse) and prints, per 64-key tile and
live wave: [0] wait for the tile (vmcnt + workgroup barrier), [1] issue of the next tile's direct-to-LDS loads, [2] K fragment reads +
issue of the 8 S MFMAs, [3] S results -> row max (MFMA latency lands here), [4] exp2 / sums / packing, [5] V fragment reads + 8 PV MFMAs.
Usage: attn_phase_probe.py [batch ...]   (ViT-L shape: 16 heads, 901 tokens)"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
C.CDLL(_lib.LIB_PATH, mode=C.RTLD_GLOBAL)                      # set_error / device_cu_count come from the library
csrc = os.path.join(ROOT, "lang-seg_amd", "csrc")
src = r'''
#include <hip/hip_runtime.h>
__device__ unsigned long long g_attn_probe[8];
#define ATTN_PROBE_DECL unsigned long long pt_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long pl_ = __builtin_readcyclecounter();
#define ATTN_PROBE(i) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pl_; pl_ = n_; }
#define ATTN_PROBE_DUMP if (lane == 0 && live) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_attn_probe[i_], pt_[i_]); \
                                                 atomicAdd(&g_attn_probe[6], 1ull); atomicAdd(&g_attn_probe[7], (unsigned long long)n_tiles); }
#include "attention.hip"
extern "C" int probe_run(const void* q, const void* k, const void* vt, void* out, int B, int H, int ntok, int npad, int dtype, void* st) {
    return lseg::launch_attention(q, k, vt, out, B, H, ntok, npad, dtype, 0, 0.125f, (hipStream_t)st);
}
extern "C" int probe_read(unsigned long long* out, int reset) {
    int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_probe), 64);
    if (reset) { unsigned long long z[8] = {0}; e |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_probe), z, 64); }
    return e;
}
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "p.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-I", csrc,
                       "-o", os.path.join(d, "p.so"), os.path.join(d, "p.hip")])
lib = C.CDLL(os.path.join(d, "p.so"))
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["wait tile + barrier", "issue next tile", "K reads + S MFMA issue", "S latency + row max", "exp2 / sum / pack", "V reads + PV MFMA"]
for B in (int(v) for v in (sys.argv[1:] or ["36"])):
    H, N, Npad = 16, 901, 1024
    for dt, code in ((torch.bfloat16, _lib.LSEG_BF16), (torch.float16, _lib.LSEG_F16)):
        g = torch.Generator().manual_seed(0)
        q = torch.zeros((B * H, Npad, 64), dtype=dt).cuda(); q[:, :N] = (0.5 * torch.randn((B * H, N, 64), generator=g)).to(dt).cuda()
        k = torch.zeros((B * H, Npad, 64), dtype=dt).cuda(); k[:, :N] = (0.5 * torch.randn((B * H, N, 64), generator=g)).to(dt).cuda()
        vt = torch.zeros((B * H, 64, Npad), dtype=dt).cuda(); vt[:, :, :N] = torch.randn((B * H, 64, N), generator=g).to(dt).cuda()
        out = torch.zeros((B, N, H * 64), dtype=dt).cuda()
        run = lambda: lib.probe_run(P(q), P(k), P(vt), P(out), B, H, N, Npad, code, st)
        for _ in range(3): assert run() == 0
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)(); lib.probe_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        lib.probe_read(buf, 1)
        v = list(buf); tiles = v[7]
        print(f"B={B} {str(dt)[6:]}: {e0.elapsed_time(e1) * 100:.1f} us per launch (with marks); per tile and live wave, s_memtime ticks:")
        tot = sum(v[:6])
        for i in range(6):
            print(f"   [{i}] {names[i]:24s} {v[i] / tiles:8.1f}  {100 * v[i] / tot:5.1f} %")
        print(f"       total {tot / tiles:8.1f} ticks per tile; {v[6] // 10} live waves, {tiles / v[6]:.1f} tiles each")

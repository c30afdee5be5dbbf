"""Does the ROW PITCH of a GEMM operand limit the direct-to-LDS rate?  A GEMM K-step pulls, per operand row, one 128-byte line at a stride of the
row pitch (2 KB for K = 1024, 8 KB for K = 4096 16-bit elements) -- 256 such lines per panel and step, the same lines from 4-8 CUs of an XCD at once.
tools/probes/lds_dma_probe.py streams CONTIGUOUS 32 KB blocks (36 TB/s); here every workgroup walks a [256 rows x K] panel K-step by K-step like the
kernels do, panels shared by `share` workgroups of an XCD, for several pitches.  L2-resident working set."""
import os, subprocess, ctypes, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void k(const char* g, int iters, int pitch, int ksteps, int share, int panels, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    // panel of this workgroup: `share` workgroups of an XCD read the same one (like the tiles of one row block / column block)
    const int panel = (xcd * panels + (idx / share) % panels);
    const char* base = g + (size_t)panel * 256 * pitch;
    const int row = lane >> 3, ch = lane & 7;
    for (int it = 0; it < iters; ++it) {
        const int ks = it % ksteps;
        // 256 rows = 32 pieces of 8 rows; 8 waves -> 4 pieces per wave per step
        for (int s = w; s < 32; s += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(s * 8 + row) * pitch + ks * 128 + ch * 16),
                (__attribute__((address_space(3))) void*)(smem + (it & 1) * 32768 + s * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[0];
}
extern "C" float run(const char* g, int grid, int iters, int pitch, int ksteps, int share, int panels, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 65536, 0, g, 64, pitch, ksteps, share, panels, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 65536, 0, g, iters, pitch, ksteps, share, panels, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so")); lib.run.restype = ctypes.c_float
buf = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda"); sink = torch.zeros(4096, device="cuda")
iters = 2000
for share in (8, 16, 32):
    for pitch, ksteps in [(2048, 16), (2048 + 128, 16), (2048 + 256, 16), (2048 + 64, 16), (8192, 64), (8192 + 128, 64), (8192 + 256, 64)]:
        # per XCD: 32 / share panels of 256 x pitch bytes, capped so that the XCD's working set stays <= 2 MB (L2-resident after the warm-up)
        panels = max(1, min(32 // share, (2 << 20) // (256 * pitch)))
        assert 8 * panels * 256 * pitch <= buf.numel()
        ms = lib.run(ctypes.c_void_p(buf.data_ptr()), 256, iters, pitch, ksteps, share, panels, ctypes.c_void_p(sink.data_ptr()))
        tot = 256 * iters * 32768
        print(f"share {share:2d} panels/XCD {panels} pitch {pitch:6d} B, {ksteps} K-steps of the panel: {tot / ms / 1e9:6.2f} TB/s aggregate = {tot / ms / 1e6 / 256 / 2.1:5.1f} B/clk/CU", flush=True)

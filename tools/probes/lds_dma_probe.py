"""How fast can a CU pull L2-resident data into LDS with global_load_lds_dwordx4?  (upper bound for the
GEMM operand staging).  Variants: waves per workgroup issuing, workgroups per CU."""
import os, subprocess, ctypes, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void k(const char* g, int iters, int bytes_per_wg_iter, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int slabs = bytes_per_wg_iter / 1024;           // 1 KiB per wave instruction
    const char* base = g + (size_t)(blockIdx.x % 64) * 65536;      // L2-resident working set (4 MB)
    for (int it = 0; it < iters; ++it) {
        for (int s = w; s < slabs; s += nw)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((it & 1) * 32768) + s * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(smem + (it & 1) * 32768 + s * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[0];
}
extern "C" float run(const char* g, int grid, int block, int iters, int bytes, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 65536, 0, g, 10, bytes, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 65536, 0, g, iters, bytes, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so")); lib.run.restype = ctypes.c_float
buf = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda"); sink = torch.zeros(4096, device="cuda")
for grid, block, bytes_ in [(512, 256, 32768), (256, 256, 32768), (512, 128, 32768), (512, 512, 32768), (1024, 256, 16384)]:
    iters = 2000
    ms = lib.run(ctypes.c_void_p(buf.data_ptr()), grid, block, iters, bytes_, ctypes.c_void_p(sink.data_ptr()))
    tot = grid * iters * bytes_
    print(f"grid {grid} x {block} thr, {bytes_//1024} KB/iter/WG: {tot/ms/1e9:.2f} TB/s aggregate, {tot/ms/1e6/256/2.1:.1f} B/clk/CU (at 2.1 GHz)")

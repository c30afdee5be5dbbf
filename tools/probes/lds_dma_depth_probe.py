"""L2->LDS DMA (global_load_lds_dwordx4) throughput per CU as a function of the bytes kept in flight:
a ring of D stages of S bytes per workgroup; every iteration issues one stage and waits for the stage
issued D-1 iterations ago (counted vmcnt), like a GEMM operand pipeline.  Also: how much a concurrent
ds_read_b128 stream (the MFMA fragment reads) slows it down."""
import os, subprocess, ctypes, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// SPW: loads per wave per stage; D: ring depth; READS: ds_read_b128 per wave per iteration
template <int SPW, int D, int READS>
__global__ __launch_bounds__(256) void k(const char* g, int iters, int ws_mb, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STAGE = SPW * 4 * 1024;
    const size_t span = (size_t)ws_mb << 20;
    size_t off = ((size_t)blockIdx.x * 1315423911u) & (span - 1);
    float acc = 0.f;
    int st = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const size_t o = (off + (size_t)(s * 4 + w) * 1024) & (span - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o + lane * 16),
                (__attribute__((address_space(3))) void*)(smem + st * STAGE + (s * 4 + w) * 1024), 16, 0, 0);
        }
        off = (off + STAGE * 97) & (span - 1);
        st = st + 1 == D ? 0 : st + 1;
        if (READS) {
#pragma unroll
            for (int r = 0; r < READS; ++r) {
                const float4 v = *reinterpret_cast<const float4*>(smem + st * STAGE + ((r * 1024 + lane * 16) % STAGE));
                acc += v.x;
            }
        }
        wait_vmcnt<(D - 1) * SPW>();
        __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();
    if (acc == 123.f) sink[blockIdx.x] = acc;
}
template <int SPW, int D, int READS>
float run1(const char* g, int grid, int iters, int ws_mb, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds = SPW * 4096 * D;
    hipFuncSetAttribute((const void*)k<SPW, D, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<SPW, D, READS>), dim3(grid), dim3(256), lds, 0, g, 10, ws_mb, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<SPW, D, READS>), dim3(grid), dim3(256), lds, 0, g, iters, ws_mb, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
extern "C" float run(int spw, int d, int reads, const char* g, int grid, int iters, int ws_mb, float* sink) {
#define CASE(S, DD, R) if (spw == S && d == DD && reads == R) return run1<S, DD, R>(g, grid, iters, ws_mb, sink);
    CASE(8, 2, 0) CASE(4, 4, 0) CASE(4, 5, 0) CASE(8, 2, 16) CASE(4, 5, 8) CASE(4, 3, 0) CASE(2, 8, 0) CASE(2, 10, 0) CASE(8, 2, 32)
    return -1.f;
}
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so")); lib.run.restype = ctypes.c_float
buf = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda"); sink = torch.zeros(4096, device="cuda")
print("2 workgroups (4 waves) per CU; stage = SPW*4 KB per workgroup; in flight per CU = 2*(D-1)*stage")
for ws in (2, 8, 64):
    for spw, dd, reads in [(8, 2, 0), (4, 3, 0), (4, 4, 0), (4, 5, 0), (2, 8, 0), (2, 10, 0), (8, 2, 16), (8, 2, 32), (4, 5, 8)]:
        iters = 4000 * 8 // spw
        ms = lib.run(spw, dd, reads, ctypes.c_void_p(buf.data_ptr()), 512, iters, ws, ctypes.c_void_p(sink.data_ptr()))
        tot = 512 * iters * spw * 4096
        print(f"working set {ws:3d} MB  stage {spw*4:2d} KB x depth {dd:2d} ({2*(dd-1)*spw*4:3d} KB in flight/CU) reads/iter {reads:2d}: "
              f"{tot/ms/1e9:6.2f} TB/s, {tot/ms/1e6/256/2.1:5.1f} B/clk/CU @2.1GHz")

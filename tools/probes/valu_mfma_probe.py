#!/usr/bin/env python
"""Issue rates on a gfx950 SIMD (one workgroup per CU, 256 CUs busy): cycles per wave-instruction of v_exp_f32, v_fma_f32, v_pk_fma_f32,
v_cvt_pk_bf16_f32, v_max3_f32 and the 32x32x16 bf16 MFMA with one wave per SIMD, then whether a VALU wave and an MFMA wave that share a
SIMD overlap: waves 0-3 of a workgroup run the MFMA loop, waves 4-7 the VALU (exp) loop, alone and together."""
import ctypes, os, subprocess, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N 4096
template <int MODE> __device__ __forceinline__ float valu_loop(float seed) {
    float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2};
    for (int i = 0; i < N / 4; ++i) {
        if (MODE == 0) { a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3); }
        if (MODE == 1) { a0 = fmaf(a0, 0.999f, 0.001f); a1 = fmaf(a1, 0.999f, 0.001f); a2 = fmaf(a2, 0.999f, 0.001f); a3 = fmaf(a3, 0.999f, 0.001f); }
        if (MODE == 2) { const f32x2 c = {0.999f, 0.999f}, d = {0.001f, 0.001f};
                         p0 = __builtin_elementwise_fma(p0, c, d); p1 = __builtin_elementwise_fma(p1, c, d); p2 = __builtin_elementwise_fma(p2, c, d); p3 = __builtin_elementwise_fma(p3, c, d); }
        if (MODE == 3) { unsigned u0, u1, u2, u3;
                         asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4"
                                      : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
                         a0 += __uint_as_float(u0 & 1u); }
        if (MODE == 4) { a0 = fmaxf(fmaxf(a0, a1), a2); a1 = fmaxf(fmaxf(a1, a2), a3); a2 = fmaxf(fmaxf(a2, a3), a0); a3 = fmaxf(fmaxf(a3, a0), a1); }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    return a0 + a1 + a2 + a3 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
}
__device__ __forceinline__ float mfma_loop(float seed) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)seed; b[j] = (__bf16)(seed + 1.f); }
    f32x16 c0 = {0}, c1 = {0};
    for (int i = 0; i < N / 2; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    return c0[0] + c1[0];
}
// role: 0 = VALU mode MODE on waves 0-3 ; 1 = MFMA on waves 0-3 ; 2 = MFMA on waves 0-3 and VALU (MODE) on waves 4-7 ; 3 = VALU on all 8 waves
template <int MODE> __global__ __launch_bounds__(512) void k(float* out, long long* cyc, int role) {
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    if (role == 0) { if (w < 4) r = valu_loop<MODE>((float)threadIdx.x * 1e-3f); }
    else if (role == 1) { if (w < 4) r = mfma_loop(1.f); }
    else if (role == 2) { if (w < 4) r = mfma_loop(1.f); else r = valu_loop<MODE>((float)threadIdx.x * 1e-3f); }
    else r = valu_loop<MODE>((float)threadIdx.x * 1e-3f);
    const long long t1 = __builtin_readcyclecounter();
    if (r == 123.456f) out[0] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}
extern "C" float run(int mode, int role, float* out, long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, role); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, cyc, role); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, cyc, role); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, cyc, role); break;
            default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, cyc, role); break;
        }
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so")); lib.run.restype = ctypes.c_float
out = torch.zeros(16, device="cuda"); cyc = torch.zeros(8, dtype=torch.int64, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
N = 4096
names = ["v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_max3_f32 (or 2 v_max)"]
print("one wave per SIMD, %d instructions each:" % N)
for m, nm in enumerate(names):
    ms = lib.run(m, 0, P(out), P(cyc)); c = cyc.cpu()
    print(f"  {nm:26s} {ms * 1e3:8.1f} us   {c[0].item() / N:6.2f} s_memtime ticks per wave-instruction")
ms_m = lib.run(0, 1, P(out), P(cyc)); c = cyc.cpu()
print(f"  mfma_f32_32x32x16_bf16     {ms_m * 1e3:8.1f} us   {c[0].item() / N:6.2f} ticks per MFMA  ({256 * 4 * N * 32768 / ms_m / 1e9:.0f} TF/s over 256 CUs)")
print("MFMA wave + VALU wave on the same SIMD (waves 0-3 MFMA, waves 4-7 VALU):")
for m in (0, 1, 2):
    ms_v = lib.run(m, 0, P(out), P(cyc))
    ms_b = lib.run(m, 2, P(out), P(cyc)); c = cyc.cpu()
    ms_2 = lib.run(m, 3, P(out), P(cyc))
    print(f"  {names[m]:26s} alone {ms_v * 1e3:7.1f} us, MFMA alone {ms_m * 1e3:7.1f} us, together {ms_b * 1e3:7.1f} us  (sum {1e3 * (ms_v + ms_m):7.1f});"
          f" two VALU waves per SIMD {ms_2 * 1e3:7.1f} us")

"""Prints the lane mapping of v_permlane16_swap / v_permlane32_swap on the device (documentation probe)."""
import os, subprocess, sys, ctypes, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(unsigned* p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[128 + threadIdx.x] = q[0]; p[192 + threadIdx.x] = q[1];
}
extern "C" void run(unsigned* p) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p); hipDeviceSynchronize(); }
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so"))
t = torch.zeros(256, dtype=torch.int32, device="cuda")
lib.run(ctypes.c_void_p(t.data_ptr()))
v = t.cpu().tolist()
print("permlane16_swap vdst':", v[0:64]); print("permlane16_swap src' :", v[64:128])
print("permlane32_swap vdst':", v[128:192]); print("permlane32_swap src' :", v[192:256])

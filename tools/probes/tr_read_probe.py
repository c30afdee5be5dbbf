"""ds_read_b64_tr_b16 (gfx950 LDS transpose read) -- what each lane receives, and an MFMA operand built with it from a K-MAJOR tile.
Preparation for a wgrad GEMM without materialised transposes (dW = dY^T X contracts over the token axis, which is
the slow axis of both operands as they sit in memory).
  part 1: LDS holds lds[i] = i (u16); lane l passes the address of element 4*l; prints which four elements every lane gets and checks
          the reading "a 16-lane group reads a [4][16] block (lane i supplies row i/4, columns 4*(i%4)..+3) and lane i receives column i".
  part 2: A [16 x 32] is stored K-major (T[k][m], 32-byte rows) in LDS, its MFMA fragments (v_mfma_f32_16x16x32_bf16: lane (m, kg)
          holds k = 8*kg .. 8*kg+7) are gathered with two transpose reads per lane, B comes from a plain [n][k] tile; C is compared
          with the host product."""
import ctypes, os, subprocess, tempfile
import torch
src = r'''
#include <hip/hip_runtime.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;
extern "C" __global__ void map_kernel(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
// Tk: [32][16] bf16 (A stored K-major: Tk[k][m]); Bn: [16][32] bf16 ([n][k]); C: [16][16] fp32, C[m][n] = sum_k A[m][k] B[n][k]
extern "C" __global__ void mfma_kernel(const unsigned short* Tk, const unsigned short* Bn, float* C) {
    __shared__ __attribute__((aligned(16))) unsigned short sa[32 * 16], sb[16 * 32];
    for (int i = threadIdx.x; i < 512; i += 64) { sa[i] = Tk[i]; sb[i] = Bn[i]; }
    __syncthreads();
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    union { v4s h[2]; v8bf v; } a, b;
#pragma unroll
    for (int half = 0; half < 2; ++half)      // lane i of a 16-lane group supplies row i/4, columns 4*(i%4) of the [4][16] block of its k-group
        a.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sa + (kg * 8 + half * 4 + (m >> 2)) * 16 + 4 * (m & 3)));
    b.h[0] = *reinterpret_cast<const v4s*>(sb + m * 32 + kg * 8);
    b.h[1] = *reinterpret_cast<const v4s*>(sb + m * 32 + kg * 8 + 4);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
    // accumulator layout of 16x16: lane (col = lane & 15 of operand B rows = n, rows 4*(lane>>4) + r of operand A = m)
    for (int r = 0; r < 4; ++r) C[(4 * kg + r) * 16 + m] = c[r];
}
extern "C" void run_map(unsigned short* out) { hipLaunchKernelGGL(map_kernel, dim3(1), dim3(64), 0, 0, out); (void)hipDeviceSynchronize(); }
extern "C" void run_mfma(const unsigned short* a, const unsigned short* b, float* c) { hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, a, b, c); (void)hipDeviceSynchronize(); }
'''
d = tempfile.mkdtemp(); open(os.path.join(d, "k.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = ctypes.CDLL(os.path.join(d, "k.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
out = torch.zeros(256, dtype=torch.int16, device="cuda")
lib.run_map(P(out))
got = out.cpu().view(64, 4).to(torch.int32) & 0xffff
print("lane -> elements received (first 20 lanes):")
for l in range(20):
    print(f"  lane {l:2d}: {got[l].tolist()}")
want = torch.zeros_like(got)
for l in range(64):
    grp, i = l >> 4, l & 15
    for j in range(4):
        # block of the group = the 64 elements its lanes address: row r = the 4 lanes 4r..4r+3 (16 contiguous elements); lane i gets column i
        want[l, j] = grp * 64 + j * 16 + i
print("reading '[4][16] block per 16 lanes, lane i <- column i':", "CONFIRMED" if torch.equal(got, want) else "DIFFERENT (see the dump)")
g = torch.Generator().manual_seed(0)
A = torch.randn(16, 32, generator=g).to(torch.bfloat16)
B = torch.randn(16, 32, generator=g).to(torch.bfloat16)
Tk = A.t().contiguous().cuda()                  # K-major image of A
C = torch.zeros(16, 16, device="cuda")
lib.run_mfma(P(Tk), P(B.cuda()), P(C))
ref = A.float() @ B.float().t()
err = (C.cpu() - ref).abs().max().item()
print(f"MFMA with transpose-read A fragments: max|C - A B^T| = {err:.3e} ->", "OK" if err < 1e-3 else "WRONG LAYOUT (try C^T / the other accumulator orientation)")
if err >= 1e-3:
    print("  vs transposed:", (C.cpu().t() - ref).abs().max().item())

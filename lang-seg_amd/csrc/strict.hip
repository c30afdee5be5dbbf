// strict.hip -- the elementwise / attention kernels of the split-precision ("strict") validation mode.
//
// In this mode every 16-bit tensor of the image tower is a (hi, lo) pair of fp16 PLANES: x = hi + lo, hi = fp16(x),
// lo = fp16(x - hi) (~21 mantissa bits; the reference's image tower is fp32, lseg_net.py:160-205 under torch defaults).  The GEMM /
// implicit-GEMM conv kernel runs its K-loop over three segments A_hi.W_hi + A_lo.W_hi + A_hi.W_lo with fp32 accumulation
// (gemm.hip, GemmArgs::split); the kernels here produce and consume the planes around it.  They are plain one-thread-per-element
// kernels: the mode exists to show that the 16-bit modes' deviations from the reference are operand rounding and nothing else
// (argmax masks equal to the reference's up to fp16-ulp ties of its own fp16 logits), not to be fast (~1/4 of the bf16 rate).
#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {
namespace {

__device__ __forceinline__ void put_split(uint16_t* p, size_t i, size_t plane, float v) {
    const float hi = round_f16(v);
    p[i] = f32_to_f16(hi);
    p[i + plane] = f32_to_f16(v - hi);
}
__device__ __forceinline__ float get_split(const uint16_t* p, size_t i, size_t plane) {
    return f16_to_f32(p[i]) + f16_to_f32(p[i + plane]);
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void convert_split_kernel(const void* in, int in_dtype, uint16_t* out, size_t n, size_t plane) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        put_split(out, i, plane, load_as_f32(in, i, in_dtype));
}

// LayerNorm, fp32 rows -> (hi, lo) planes; one wave per row
__global__ __launch_bounds__(256) void ln_split_kernel(const float* x, const float* gamma, const float* beta, uint16_t* out, size_t plane,
                                                       int M, int D, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wsum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wsum(q) / (float)D + eps);
    for (int c = lane; c < D; c += 64) put_split(out, (size_t)row * D + c, plane, (xr[c] - mean) * rstd * gamma[c] + beta[c]);
}

// patch im2col: x fp32 NCHW -> A [B*gh*gw, 3*P*P] planes (k = c*P*P + i*P + j)
__global__ void im2col_split_kernel(const float* x, uint16_t* A, size_t plane, int B, int H, int W, int P) {
    const int gh = H / P, gw = W / P, Kd = 3 * P * P;
    const size_t total = (size_t)B * gh * gw * Kd;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % Kd);
        const size_t m = idx / Kd;
        const int c = k / (P * P), rem = k - c * P * P, i = rem / P, j = rem - i * P;
        const int b = (int)(m / (gh * gw)), p = (int)(m - (size_t)b * gh * gw), py = p / gw, px = p - py * gw;
        put_split(A, idx, plane, x[(((size_t)b * 3 + c) * H + py * P + i) * W + px * P + j]);
    }
}

// ProjectReadout concat: x fp32 [B,N,D] -> A [B*(N-1), 2D] planes
__global__ void readout_cat_split_kernel(const float* x, uint16_t* A, size_t plane, int B, int ntok, int D) {
    const size_t total = (size_t)B * (ntok - 1) * 2 * D;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (2 * D));
        const size_t m = idx / (2 * D);
        const int b = (int)(m / (ntok - 1)), t = (int)(m - (size_t)b * (ntok - 1));
        const float v = c < D ? x[((size_t)b * ntok + t + 1) * D + c] : x[(size_t)b * ntok * D + (c - D)];
        put_split(A, idx, plane, v);
    }
}

// bilinear x2 (align_corners=True), padded NHWC planes -> plain NHWC planes
__global__ void upsample2x_nhwc_split_kernel(const uint16_t* in, size_t in_plane, uint16_t* out, size_t out_plane, int B, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)B * Ho * Wo * C;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        size_t p = idx / C;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float sy = ry * (float)yo, sx = rx * (float)xo;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        auto at = [&](int y, int x) { return get_split(in, (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c, in_plane); };
        put_split(out, idx, out_plane, (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1)));
    }
}

// out = relu(in) on (hi, lo) planes (the RCU convs read their input through a ReLU: a sign test on hi alone would keep a lo of the
// wrong sign, so the ReLU-ed map is materialised)
__global__ void relu_split_kernel(const uint16_t* in, size_t in_plane, uint16_t* out, size_t out_plane, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = get_split(in, i, in_plane);
        put_split(out, i, out_plane, v > 0.f ? v : 0.f);
    }
}

// softmax(Q K^T * scale) V in fp32 on (hi, lo) planes.  Block = 64 queries x 4 key partitions (wave p handles keys p, p+4, ...);
// each thread keeps an online-softmax state for its query over its partition; the four partitions are merged through LDS.
// q,k [BH, Npad, 64]; vt [BH, 64, Npad]; out [B, Ntok, H*64].
__global__ __launch_bounds__(256) void attention_strict_kernel(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out,
                                                               size_t qk_plane, size_t vt_plane, size_t out_plane, int B, int H, int ntok,
                                                               int npad, float scale) {
    __shared__ float sm[4][64], sl[4][64];
    __shared__ float so[4][64][65];
    const int qi = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int bh = blockIdx.y, qrow = blockIdx.x * 64 + qi;
    const bool live = qrow < ntok;
    float qv[64], o[64];
    const size_t qoff = ((size_t)bh * npad + (live ? qrow : 0)) * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) { qv[d] = get_split(q, qoff + d, qk_plane) * scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int key = part; key < ntok; key += 4) {
        const size_t koff = ((size_t)bh * npad + key) * 64;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s = fmaf(qv[d], get_split(k, koff + d, qk_plane), s);
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), p = __expf(s - mn);
        l = l * a + p;
        const size_t voff = (size_t)bh * 64 * npad + key;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = o[d] * a + p * get_split(vt, voff + (size_t)d * npad, vt_plane);
        m = mn;
    }
    sm[part][qi] = m; sl[part][qi] = l;
#pragma unroll
    for (int d = 0; d < 64; ++d) so[part][qi][d] = o[d];
    __syncthreads();
    if (part == 0 && live) {
        float mm = fmaxf(fmaxf(sm[0][qi], sm[1][qi]), fmaxf(sm[2][qi], sm[3][qi]));
        float w[4], lt = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) { w[p] = sm[p][qi] == -INFINITY ? 0.f : __expf(sm[p][qi] - mm); lt += w[p] * sl[p][qi]; }
        const int b = bh / H, h = bh - b * H;
        const size_t obase = ((size_t)b * ntok + qrow) * (H * 64) + h * 64;
        for (int d = 0; d < 64; ++d) {
            const float v = (w[0] * so[0][qi][d] + w[1] * so[1][qi][d] + w[2] * so[2][qi][d] + w[3] * so[3][qi][d]) / lt;
            put_split(out, obase + d, out_plane, v);
        }
    }
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define CHECK_LAUNCH() LSEG_HIP_TRY(hipGetLastError())

int launch_convert_split(const void* in, int in_dtype, void* out, size_t n, size_t plane, hipStream_t st) {
    hipLaunchKernelGGL(convert_split_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, in_dtype, (uint16_t*)out, n, plane);
    CHECK_LAUNCH();
    return 0;
}
int launch_ln_split(const float* x, const float* gamma, const float* beta, void* out, size_t plane, int M, int D, float eps, hipStream_t st) {
    hipLaunchKernelGGL(ln_split_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, gamma, beta, (uint16_t*)out, plane, M, D, eps);
    CHECK_LAUNCH();
    return 0;
}
int launch_im2col_split(const float* x, void* A, size_t plane, int B, int H, int W, int P, hipStream_t st) {
    hipLaunchKernelGGL(im2col_split_kernel, dim3(grid_for((size_t)B * 3 * H * W)), dim3(256), 0, st, x, (uint16_t*)A, plane, B, H, W, P);
    CHECK_LAUNCH();
    return 0;
}
int launch_readout_cat_split(const float* x, void* A, size_t plane, int B, int ntok, int D, hipStream_t st) {
    hipLaunchKernelGGL(readout_cat_split_kernel, dim3(grid_for((size_t)B * (ntok - 1) * 2 * D)), dim3(256), 0, st, x, (uint16_t*)A, plane, B, ntok, D);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_nhwc_split(const void* in, size_t in_plane, void* out, size_t out_plane, int B, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(upsample2x_nhwc_split_kernel, dim3(grid_for((size_t)B * 4 * H * W * C)), dim3(256), 0, st, (const uint16_t*)in, in_plane,
                       (uint16_t*)out, out_plane, B, H, W, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_relu_split(const void* in, size_t in_plane, void* out, size_t out_plane, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_split_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const uint16_t*)in, in_plane, (uint16_t*)out, out_plane, n);
    CHECK_LAUNCH();
    return 0;
}
int launch_attention_strict(const void* q, const void* k, const void* vt, void* out, size_t qk_plane, size_t vt_plane, size_t out_plane,
                            int B, int H, int ntok, int npad, float scale, hipStream_t st) {
    dim3 grid((ntok + 63) / 64, B * H);
    hipLaunchKernelGGL(attention_strict_kernel, grid, dim3(256), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)vt, (uint16_t*)out,
                       qk_plane, vt_plane, out_plane, B, H, ntok, npad, scale);
    CHECK_LAUNCH();
    return 0;
}

}  // namespace lseg
